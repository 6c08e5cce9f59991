/*
 * wsis_hip.h -- C ABI of the MI355X (gfx950) hot path of 3D-WSIS.
 *
 * This is the drop-in boundary.  Two shared libraries export it:
 *
 *   libwsis_host.so  (g++ only, never touches the HIP runtime -- safe in forked DataLoader workers)
 *       wsis_host_*   : voxelization_idx, bfs_cluster, reference-order helpers
 *   libwsis_hip.so   (hipcc --offload-arch=gfx950)
 *       wsis_*        : every device operator
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory, h_* is HOST memory; the caller (PyTorch) allocates
 *     every input, output and workspace buffer and passes raw pointers + sizes.  The library never
 *     allocates device memory that outlives a call and never frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - every function returns int: 0 = ok, <0 = error; wsis_last_error()/wsis_host_last_error()
 *     return a thread-local message.  No C++ exception crosses the ABI.
 *   - variable-size outputs use count-then-fill or documented upper bounds with the true count
 *     written to a device int32 the caller reads back.
 *   - feature rows are row-major fp32 [rows, C]; indices are int32 [M,4] = (batch, c0, c1, c2).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * fpthink/3D-WSIS tree).  [UPSTREAM] marks semantics of an un-vendored dependency (spconv v1.0
 * llijiang fork, PointGroup lib/pointgroup_ops, torch_scatter 2.0.x) restated in SURVEY.md App. A.
 */
#ifndef WSIS_HIP_H_
#define WSIS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSIS_ABI_VERSION 2

/* error codes */
#define WSIS_OK 0
#define WSIS_ERR_ARG (-1)      /* bad argument (null pointer, bad size, unsupported shape) */
#define WSIS_ERR_HIP (-2)      /* a HIP runtime call failed (message has hipGetErrorString) */
#define WSIS_ERR_OVERFLOW (-3) /* a caller-provided capacity was too small */

/* ------------------------------------------------------------------------------------------ */
/* libwsis_host.so : host-only operators                                                      */
/* ------------------------------------------------------------------------------------------ */

int wsis_host_version(void);
const char* wsis_host_last_error(void);

/* pointgroup_ops.voxelization_idx(coords, batchsize, mode)  [UPSTREAM PG_OP.voxelize_idx]
 * call sites: modules/datasets/scannetv2_dataset.py:449,528  test_scannetv2.py:389
 * Pass 1: scan points in order, voxel id = order of first occurrence of the (b,x,y,z) row.
 *   h_coords int64 [N,4]; writes h_p2v int32 [N]; *M = #voxels; *max_active = longest point list. */
int wsis_host_voxelize_idx_map(const int64_t* h_coords, int64_t N, int32_t* h_p2v, int64_t* M,
                               int32_t* max_active);
/* Pass 2: h_voxel_locs int64 [M,4] = coords of each voxel's first point;
 *   h_v2p int32 [M, 1+max_active] = [count, p0, p1, ... ascending, 0 padded]. */
int wsis_host_voxelize_idx_fill(const int64_t* h_coords, int64_t N, const int32_t* h_p2v, int64_t M,
                                int32_t max_active, int64_t* h_voxel_locs, int32_t* h_v2p);

/* pointgroup_ops.bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold)
 * [UPSTREAM PG_OP.bfs_cluster; no call site in the reference, named by BASELINE.json north_star]
 * Pass 1: FIFO BFS, seeds ascending, neighbour accepted iff same label & unvisited.  Writes
 *   h_point_cluster int32 [N] = kept-cluster id or -1 and h_order int32 [N] = discovery rank of the
 *   point inside its cluster; *n_clusters, *n_points = totals over kept clusters (size >= threshold). */
int wsis_host_bfs_cluster_count(const int32_t* h_semantic, const int32_t* h_ball_idx,
                                const int32_t* h_start_len, int64_t N, int32_t threshold,
                                int32_t* h_point_cluster, int32_t* h_order, int64_t* n_clusters,
                                int64_t* n_points);
/* Pass 2: h_cluster_idxs int32 [n_points,2] = (cluster id, point idx) in discovery order,
 *   h_cluster_offsets int32 [n_clusters+1]. */
int wsis_host_bfs_cluster_fill(const int32_t* h_point_cluster, const int32_t* h_order, int64_t N,
                               int64_t n_clusters, int64_t n_points, int32_t* h_cluster_idxs,
                               int32_t* h_cluster_offsets);

/* Superpoint-graph BFS of the test-time grouping, test_scannetv2.py:312-340 (`BFS`) driven by the seed loop of
 * :372-381.  Seeds ascending over superpoints whose class is valid and that are unvisited; a
 * neighbour joins iff same class as the seed, unvisited, and ||centre[cur] - centre[nb]||_2 < 0.25*ins_size[seed]
 * (fp32).  h_adj_off int32 [S+1], h_adj int32 [nnz] = igraph `neighbors(mode='all')` lists; h_class_valid
 * uint8 [n_class].  h_group int32 [S] = group id in seed order or -1. */
int wsis_host_graph_bfs(const int32_t* h_label, const uint8_t* h_class_valid, int32_t n_class,
                        const float* h_centre, const float* h_ins_size, const int32_t* h_adj_off,
                        const int32_t* h_adj, int64_t S, int32_t* h_group, int64_t* n_groups);

/* ------------------------------------------------------------------------------------------ */
/* libwsis_hip.so : device operators                                                          */
/* ------------------------------------------------------------------------------------------ */

int wsis_version(void);
const char* wsis_last_error(void);
/* number of visible HIP devices (0 when none) -- does not create a context */
int wsis_device_count(void);

/* ---- a2: pointgroup_ops.voxelization(feats, map_rule, mode)  train_scannetv2.py:189 ---------
 * out[m,:] = sum_{i<n_m} w * feats[v2p[m,1+i],:], w = 1/n_m for mode 4 else 1, accumulated in
 * list order in fp32.  d_v2p int32 [M, stride] with stride = 1+max_active. */
int wsis_voxelize_fwd(const float* d_feats, const int32_t* d_v2p, float* d_out, int64_t M, int32_t C,
                      int32_t stride, int32_t mode, void* stream);
/* d_dfeats [N,C] must be zero-initialised by the caller; dfeats[p,:] = w * dout[m,:]. */
int wsis_voxelize_bwd(const float* d_dout, const int32_t* d_v2p, float* d_dfeats, int64_t M, int32_t C,
                      int32_t stride, int32_t mode, void* stream);

/* ---- GPU voxelization_idx (a1 off the CPU workers) ------------------------------------------------
 * Same contract as wsis_host_voxelize_idx_* (first-occurrence voxel ids, ascending point lists), bit-exact.
 * d_coords int64 [N,4] device.  Pass 1 writes d_p2v int32 [N] and d_counts2 int32[2] = {M, max_active}; the caller
 * reads them back, allocates voxel_locs int64 [M,4] and v2p int32 [M,1+max_active], and calls pass 2 with the SAME
 * workspace (it holds the points sorted by voxel id). */
int64_t wsis_voxelize_idx_workspace_bytes(int64_t N);
int wsis_voxelize_idx_map(const int64_t* d_coords, int64_t N, int32_t* d_p2v, int32_t* d_counts2, void* d_ws,
                          int64_t ws_bytes, void* stream);
int wsis_voxelize_idx_fill(const int64_t* d_coords, int64_t N, int64_t M, int32_t max_active,
                           int64_t* d_voxel_locs, int32_t* d_v2p, void* d_ws, int64_t ws_bytes, void* stream);

/* ---- a5/a6: rulebooks [UPSTREAM spconv getIndicePair]  sparse_unet3d.py:130,261,292 ---------
 * Native rulebook format = gather table nbr int32 [K, M_rows]: nbr[k][r] = row of the OTHER side
 * paired with row r under flat kernel offset k (row-major over the 3 kernel dims), or -1.
 * Hash table: open addressing, d_keys int64 [cap] (cap power of two >= 2*M), d_vals int32 [cap];
 * key = linear index ((b*S0+c0)*S1+c1)*S2+c2. */
int wsis_hash_build(const int32_t* d_indices, int64_t M, const int32_t* h_shape3, int64_t* d_keys,
                    int32_t* d_vals, int64_t cap, void* stream);
/* SubMConv3d: out rows == in rows.  nbr[k][o] = i with coord[i] = coord[o] - pad + kappa.
 * d_mask uint32 [M] (optional, may be null, needs K<=32): bit k set iff nbr[k][o] >= 0. */
int wsis_rulebook_subm(const int32_t* d_indices, int64_t M, const int32_t* h_shape3,
                       const int32_t* h_ksize3, const int32_t* h_pad3, const int64_t* d_keys,
                       const int32_t* d_vals, int64_t cap, int32_t* d_nbr, uint32_t* d_mask,
                       void* stream);
/* SparseConv3d (stride s, pad p, dilation 1): out_shape_j = floor((in_j + 2p_j - (k_j-1) - 1)/s_j)+1.
 * Step 1: candidate keys + sort + unique -> d_out_keys int64 [<= n_cand] ascending linear index,
 * count in d_count int32[1].  n_cand = M_in when k==s && p==0, else M_in*K. */
int64_t wsis_rulebook_down_ncand(int64_t M_in, const int32_t* h_ksize3, const int32_t* h_stride3,
                                 const int32_t* h_pad3);
int64_t wsis_rulebook_down_workspace_bytes(int64_t n_cand);
int wsis_rulebook_down_keys(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3,
                            const int32_t* h_out_shape3, const int32_t* h_ksize3,
                            const int32_t* h_stride3, const int32_t* h_pad3, int64_t* d_cand,
                            int64_t* d_out_keys, int32_t* d_count, void* d_ws, int64_t ws_bytes,
                            void* stream);
/* Step 2 (after the caller read M_out): decode out indices [M_out,4], build the coarse hash table,
 * fill nbr_down int32 [K, M_out] (coarse row -> fine row) and nbr_up int32 [K, M_in]
 * (fine row -> coarse row); optional masks as in wsis_rulebook_subm. */
int wsis_rulebook_down_fill(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3,
                            const int32_t* h_out_shape3, const int32_t* h_ksize3,
                            const int32_t* h_stride3, const int32_t* h_pad3,
                            const int64_t* d_out_keys, int64_t M_out, int32_t* d_indices_out,
                            int64_t* d_keys, int32_t* d_vals, int64_t cap, int32_t* d_nbr_down,
                            int32_t* d_nbr_up, uint32_t* d_mask_down, uint32_t* d_mask_up,
                            void* stream);
/* Tile ordering for the implicit GEMM: d_order int32 [M] = stable argsort of d_mask (rows with the
 * same set of active offsets become neighbours so whole tiles skip inactive offsets). */
int64_t wsis_mask_order_workspace_bytes(int64_t M);
/* Spatial + mask tile ordering: stable sort by (batch, Morton code of the row's block of 2^block_shift voxels
 * per side, offset mask).  Rows of one block become neighbours, so the gathers of a tile hit the same L1/L2
 * lines; inside a block equal offset sets stay adjacent.  d_mask may be NULL (pure spatial order). */
int64_t wsis_tile_order_workspace_bytes(int64_t M);
int wsis_tile_order(const int32_t* d_indices, const uint32_t* d_mask, int64_t M, int32_t block_shift,
                    int32_t* d_order, void* d_ws, int64_t ws_bytes, void* stream);
int wsis_mask_order(const uint32_t* d_mask, int64_t M, int32_t* d_order, void* d_ws, int64_t ws_bytes,
                    void* stream);
/* The tile orders of up to 16 gather tables (every table of a UNet pyramid) from ONE sort: h_indices[t] /
 * h_mask[t] are host arrays of device pointers (int32 [M_t,4] / uint32 [M_t] or NULL), h_M the row counts.
 * d_order_all int32 [sum M_t]: table t's order (row numbers local to the table, identical to what
 * wsis_tile_order returns for it) at offset sum_{u<t} M_u.  The batch index of every row must be < 16
 * (batch_size, checked on the host side); N = sum M_t for the workspace query. */
int64_t wsis_tile_order_batch_workspace_bytes(int64_t N);
int wsis_tile_order_batch(int32_t n, const void* const* h_indices, const void* const* h_mask, const int64_t* h_M,
                          int32_t block_shift, int32_t batch_size, int32_t* d_order_all, void* d_ws,
                          int64_t ws_bytes, void* stream);

/* Packed gather table for the convolution kernels: nbr_packed[k][t] = nbr[k][order[t]] (columns in tile
 * order, so a 128-row tile reads K coalesced 512-byte pieces).  With d_order == NULL the kernels take the
 * unpacked table as is. */
int wsis_rulebook_pack(const int32_t* d_nbr, const int32_t* d_order, int32_t* d_nbr_packed, int64_t M,
                       int32_t K, void* stream);
/* the same for up to 16 tables in one launch: host arrays of device pointers / sizes, one entry per table */
int wsis_rulebook_pack_batch(int32_t n, const void* const* h_nbr, const void* const* h_order,
                             void* const* h_nbr_packed, const int64_t* h_M, const int32_t* h_K, void* stream);

/* The whole rulebook pyramid of a UBlock from ONE call (SubM k3 p1 table per level, SparseConv3d k2 s2 tables between
 * levels: sparse_unet3d.py:130,261,292) when every level's row count is known on the host (h_counts[l-1] = rows of
 * level l, e.g. from the loader's host-side coordinates): no device read-back, one arena, the same entry points in the
 * same order as the per-table build (identical tables).  wsis_rulebook_pyramid_layout returns the arena size and
 * fills h_layout [n_levels][WSIS_PYR_FIELDS] with byte offsets into the arena (-1: not present; ROWS / CAP are values):
 * the packed tables + tile orders the convolutions take, the unpacked tables, masks, coordinate hashes, the coarse
 * levels' indices, and the device's own output count per strided level (int32 at COUNT: compare with h_counts). */
enum {
  WSIS_PYR_ROWS = 0, WSIS_PYR_CAP, WSIS_PYR_INDICES, WSIS_PYR_KEYS, WSIS_PYR_VALS, WSIS_PYR_SUBM_NBR, WSIS_PYR_SUBM_MASK,
  WSIS_PYR_SUBM_ORDER, WSIS_PYR_SUBM_NBR_P, WSIS_PYR_CAND, WSIS_PYR_OUT_KEYS, WSIS_PYR_COUNT, WSIS_PYR_DOWN_WS,
  WSIS_PYR_DOWN_NBR, WSIS_PYR_UP_NBR, WSIS_PYR_DOWN_MASK, WSIS_PYR_UP_MASK, WSIS_PYR_DOWN_ORDER, WSIS_PYR_UP_ORDER,
  WSIS_PYR_DOWN_NBR_P, WSIS_PYR_UP_NBR_P, WSIS_PYR_FIELDS
};
int64_t wsis_rulebook_pyramid_layout(int64_t M0, const int64_t* h_counts, int32_t n_levels, int64_t* h_layout);
int wsis_rulebook_pyramid(const int32_t* d_indices0, int64_t M0, const int32_t* h_shape3, const int64_t* h_counts,
                          int32_t n_levels, int32_t batch_size, int32_t block_shift, void* d_arena, int64_t arena_bytes,
                          void* stream);

/* ---- a7-a11: sparse convolution [UPSTREAM spconv indiceConv / indiceConvBackward] -----------
 * out[r,:] = sum_k X[nbr[k][r],:] @ W[k]  (rows with nbr<0 contribute nothing), fp32.
 *   SubMConv3d fwd      : X=in,   nbr=subm table,  W=weight [K,Cin,Cout]
 *   SubMConv3d dIn      : X=dOut, nbr=subm table,  W=wsis_weight_transpose(weight, flip=1)
 *   SparseConv3d fwd    : X=in,   nbr=nbr_down,    W=weight
 *   SparseConv3d dIn    : X=dOut, nbr=nbr_up,      W=transpose(weight, flip=0)
 *   SparseInverseConv3d : X=in,   nbr=nbr_up,      W=weight ; dIn: X=dOut, nbr=nbr_down, W^T
 * d_nbr is the PACKED table when d_order (int32 [M_out], tile ordering from wsis_mask_order) is given,
 * the plain table when d_order is NULL; d_bias optional [Cout];
 * d_residual optional [M_out,Cout] added in the epilogue (sparse_unet3d.py:170 fused).
 * K==1 with d_nbr==null is the dense 1x1 shortcut (sparse_unet3d.py:115-119). */
/* Small levels split the K offsets over blockIdx.z and reduce partial slabs from d_ws in a fixed order
 * (size from the query; 256 bytes when no split is used). */
int64_t wsis_spconv_fwd_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout);
int wsis_spconv_fwd(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_W,
                    const float* d_bias, const float* d_residual, float* d_out, int64_t M_in,
                    int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes,
                    void* stream);
/* The same product with the weights given as B^T, d_WT [K,Cout,Cin] (row = output channel, Cin contiguous) -- the
 * wave-autonomous LDS-DMA kernel (csrc/spconv2.hip) that every UNet layer with channel counts that are multiples of 32
 * runs on (wsis_spconv_fwd_t_supported; K <= 32).  Replaces the same upstream call sites as wsis_spconv_fwd:
 *   forward (sparse_unet3d.py:130,261,292) : d_WT = wsis_weight_transpose(weight, flip=0), flip = 0
 *   dIn (autograd backward)    : d_WT = the layer's own weight [K,Cin_fwd,Cout_fwd] (its rows ARE the dIn
 *                                            output channels), flip = 1 for submanifold tables (offset k uses
 *                                            slice K-1-k), 0 for the coupled strided / inverse tables
 * with NW = 1 bit-identical to wsis_spconv_fwd; small levels split the offsets over the waves of a workgroup (added
 * through LDS in wave order) and, below that, over blockIdx.z into partial slabs in d_ws (fixed order).  d_sync: a
 * sync slot (wsis_sync_bytes; may be NULL; reserved for in-launch reductions: wsis_spconv_fwd_f). */
int32_t wsis_spconv_fwd_t_supported(int32_t K, int32_t Cin, int32_t Cout);
/* number of offset slabs the launch plan of wsis_spconv_fwd_t uses for this shape (1 = one kernel, no second pass) */
int32_t wsis_spconv_fwd_t_slabs(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout);
int64_t wsis_spconv_fwd_t_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout);
/* d_stats (optional): BatchNorm partials of the finished output, [ceil(M_out / 32)][2][Cout] floats per 32-row slice
 * in tile order: (sum, sum of squared deviations from the SLICE's own mean), fixed order of additions -- the
 * statistics pass of the BatchNorm that consumes this tensor (sparse_unet3d.py:128-137) fused into the producer's
 * epilogue; combined in fp64 (Chan) by wsis_bn_stats_finalize, so nothing cancels in fp32. */
int wsis_spconv_fwd_t(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT, int32_t flip,
                      const float* d_bias, const float* d_residual, float* d_out, float* d_stats, int64_t M_in,
                      int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes,
                      void* d_sync, void* stream);
/* the same product for a dIn pass whose output dy [M_out, Cout] feeds the backward of a BatchNorm(+ReLU) with input
 * d_bn_x [M_out, Cout] (reference: spconv's dIn followed by torch's batch_norm backward, sparse_unet3d.py:128-137):
 * besides d_out the epilogue writes, per 32-row slice, (sum dz, sum dz * xhat) with xhat = (x - mean) rsqrt(var + eps)
 * and dz = dy masked where relu && xhat * gamma + beta <= 0, into d_partials [ceil(M_out / 32)][2][Cout] -- the
 * reduction wsis_bn_bwd would otherwise make in a pass of its own (wsis_bn_bwd_from_partials consumes them). */
int wsis_spconv_fwd_t_bn(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT, int32_t flip,
                         float* d_out, float* d_partials, const float* d_bn_x, const float* d_bn_mean,
                         const float* d_bn_var, const float* d_bn_gamma, const float* d_bn_beta, float eps, int32_t relu,
                         int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes,
                         void* d_sync, void* stream);
/* ---- EXPERIMENTAL build only (make -C 3d-wsis_amd/csrc EXPERIMENTAL=1; wsis_experimental() says which build a library
 * is): the retired designs of DESIGN.md section 8 -- measured slower than the default path, kept buildable and tested.
 * The default library does not export the entry points inside these guards. */
/* The fused form of wsis_spconv_fwd_t for one layer of `BatchNorm1d -> ReLU -> conv` chains (sparse_unet3d.py:127-143):
 *   bn_in   (optional) BatchNorm(+ReLU) of the INPUT applied while the gathered rows are read: the activation
 *           relu(bn(x)) is never written; mean / var are the batch statistics (training) or the running statistics;
 *   targets (optional, needs d_stats and a sync slot) the statistics of the OUTPUT are FINISHED inside the launch: the
 *           last workgroups to arrive add the slice partials (the order of wsis_bn_stats_finalize, bit-identical) and
 *           write mean / var and update the running statistics of up to two BatchNorm layers that normalise this tensor
 *           (the UNet's skip connection feeds a second one).
 * One launch whatever the level: up to 16 waves of a workgroup split a work item's offsets, no slabs, no second kernel.
 * With both NULL it is wsis_spconv_fwd_t without slabs.  Workspace: wsis_spconv_fwd_f_workspace_bytes. */
typedef struct wsis_bn_in {
  const float* mean;
  const float* var;
  const float* gamma; /* may be NULL (1) */
  const float* beta;  /* may be NULL (0) */
  float eps;
  int32_t relu;
} wsis_bn_in;
typedef struct wsis_stat_target {
  float* mean;
  float* var;
  float* running_mean; /* may be NULL (with running_var) */
  float* running_var;
  float momentum;
  int32_t reserved;
} wsis_stat_target;
#if defined(WSIS_EXPERIMENTAL) && WSIS_EXPERIMENTAL
int64_t wsis_spconv_fwd_f_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout);
int wsis_spconv_fwd_f(const float* d_X, const wsis_bn_in* bn_in, const int32_t* d_nbr, const int32_t* d_order,
                      const float* d_WT, int32_t flip, const float* d_bias, const float* d_residual, float* d_out,
                      float* d_stats, const wsis_stat_target* targets, int32_t n_targets, int64_t M_in, int64_t M_out,
                      int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream);
#endif /* WSIS_EXPERIMENTAL */
/* WT[k'] = W[k]^T with k' = (flip ? K-1-k : k); W [K,Cin,Cout] -> WT [K,Cout,Cin]. */
int wsis_weight_transpose(const float* d_W, float* d_WT, int32_t K, int32_t Cin, int32_t Cout,
                          int32_t flip, void* stream);
/* dW[k] = sum_r X[nbr[k][r],:]^T (x) dY[r,:]   -> d_dW [K,Cin,Cout] (overwritten); d_nbr/d_order as above.
 * Deterministic: per-workgroup partial slabs in d_ws reduced in a fixed order. */
int64_t wsis_spconv_dw_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout);
int wsis_spconv_dw(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY,
                   float* d_dW, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws,
                   int64_t ws_bytes, void* stream);

/* The same weight gradient for a convolution whose INPUT is BatchNorm(+ReLU)(d_X) applied on the fly (the activation
 * relu(bn(x)) of sparse_unet3d.py:128-137 is never materialised: the forward pass normalises the gathered rows as it
 * reads them, wsis_spconv_fwd_f): dW[k] = sum_pairs relu(bn(X[i]))^T (x) dY[o].  "Own rows" form: the slices run over
 * the convolution's INPUT rows, which are normalised once per slice (d_mean / d_var [Cin], optional d_gamma / d_beta;
 * d_mean == NULL: X as it stands), and the paired dY rows are gathered through the dIn table d_nbr_b / d_order_b
 * (packed, rows = inputs; flip = 1 for submanifold tables, whose dIn offset k pairs with forward offset K - 1 - k).
 * Cin % 32 == 0, Cout % 32 == 0, K <= 32.  Fixed summation order, no atomics. */
int32_t wsis_spconv_dw_bn_supported(int32_t K, int32_t Cin, int32_t Cout);
int64_t wsis_spconv_dw_bn_workspace_bytes(int64_t M_in, int32_t K, int32_t Cin, int32_t Cout);
int wsis_spconv_dw_bn(const float* d_X, const float* d_mean, const float* d_var, const float* d_gamma,
                      const float* d_beta, float eps, int32_t relu, const int32_t* d_nbr_b, const int32_t* d_order_b,
                      int32_t flip, const float* d_dY, float* d_dW, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin,
                      int32_t Cout, void* d_ws, int64_t ws_bytes, void* stream);

/* Live timing of the dominant kernels (bench.py roofline): with profiling enabled every forward / dIn convolution
 * (which = 0) and every weight-gradient product (which = 1) is bracketed by HIP events on its launch stream: one in
 * front of the main kernel, one behind it, and -- where the product is finished by a second launch (the fixed-order sum
 * of offset slabs / workgroup slabs) -- one behind that launch.  wsis_prof_summary synchronises them, returns the
 * summed duration INCLUDING the finishing launches and the count, and clears the list; wsis_prof_records returns the
 * per-product durations in issue order instead (h_main_ms: main kernel only, h_total_ms: with the finishing launch).
 * which = 2: the BatchNorm ops of wsis_run_ops (forward and backward), one event pair around ALL launches of an op
 * (statistics finish + apply): h_main_ms == h_total_ms. */
int wsis_prof_enable(int32_t on);
int wsis_prof_summary(int32_t which, double* total_ms, int64_t* launches);
int wsis_prof_records(int32_t which, double* h_main_ms, double* h_total_ms, int64_t cap, int64_t* n);

/* ---- sync slots of the one-launch reductions ------------------------------------------------------------------
 * Operators that finish a two-level reduction in the SAME launch (last-arrival tickets; published flag for the
 * finish + apply forms) keep their cross-workgroup words in CALLER memory: d_sync points at a 4-KiB slot that the
 * caller zero-fills ONCE (e.g. torch.zeros); every launch leaves its slot zero again, so consecutive launches on one
 * stream may share a slot, launches that may overlap on different streams need different slots.  d_sync == NULL selects
 * the multi-launch form of the operator.  wsis_run_ops takes a block of wsis_sync_bytes() bytes (64 slots, one per op
 * index modulo 64, + one slot for the barrier words of the resident deep-level launches, which the library itself
 * zeroes in front of every such launch).  A wait that does not see its producers within 2 s sets word 19 of the slot
 * (nothing hangs).  One process drives one GPU: the launch plans cache device properties (CU count, LDS
 * limits) of the first device they ran on. */
int64_t wsis_sync_bytes(void);

/* ---- a12: BatchNorm1d(+ReLU) over the active voxels  sparse_unet3d.py:128-137, backbone_3D_WSIS.py:47,52-55 ---
 * Training statistics with a fixed reduction tree (deterministic).  d_ws from wsis_bn_workspace_bytes.
 * wsis_bn_stats: d_mean/d_var [C] (biased var); running stats (optional pair) updated with `momentum` and the
 * unbiased variance, as torch.nn.BatchNorm1d does. */
int64_t wsis_bn_workspace_bytes(int64_t M, int32_t C);
int wsis_bn_stats(const float* d_x, int64_t M, int32_t C, float* d_mean, float* d_var, float* d_running_mean,
                  float* d_running_var, float momentum, void* d_ws, int64_t ws_bytes, void* stream);
/* mean / biased variance (and the running-statistics update) of C channels from the n_part = ceil(M / 32) rows of
 * (sum, centred sum of squares) partials with row pitch 2*C floats written by wsis_spconv_fwd_t(d_stats): fp64,
 * fixed order, two levels (chunk sums in d_ws, then one thread per channel). */
int64_t wsis_bn_stats_finalize_workspace_bytes(int64_t n_part, int32_t C);
int wsis_bn_stats_finalize(const float* d_partials, int64_t n_part, int64_t M, int32_t C, float* d_mean, float* d_var,
                           float* d_running_mean, float* d_running_var, float momentum, void* d_ws, int64_t ws_bytes,
                           void* d_sync, void* stream);
/* wsis_bn_stats_finalize followed by wsis_bn_apply (y = relu?((x - mean) * gamma / sqrt(var + eps) + beta)) as ONE
 * launch where that is possible (a sync slot, C % 4 == 0, a grid of <= 512 resident workgroups): the workgroups that
 * finish the statistics publish them, all wait for that (bounded) and apply.  Identical results to the two calls, which
 * it makes itself otherwise (d_sync == NULL, WSIS_BN_FUSED_APPLY=0, odd shapes).  Same workspace as
 * wsis_bn_stats_finalize. */
int wsis_bn_stats_finalize_apply(const float* d_partials, int64_t n_part, int64_t M, int32_t C, float* d_mean,
                                 float* d_var, float* d_running_mean, float* d_running_var, float momentum,
                                 const float* d_x, const float* d_gamma, const float* d_beta, float eps, int32_t relu,
                                 float* d_y, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream);
/* backward of the fused BN(+ReLU) when dy was produced by wsis_spconv_fwd_t_bn: d_partials holds the n_part =
 * ceil(M / 32) rows of (sum dz, sum dz * xhat) slice partials (pitch 2*C floats) that the convolution's epilogue wrote,
 * so the reduction pass over x and dy of wsis_bn_bwd is replaced by an fp64 sum of the partials (fixed order, chunk
 * sums in d_ws: wsis_bn_stats_finalize_workspace_bytes); then dx as in wsis_bn_bwd (training mode). */
int wsis_bn_bwd_from_partials(const float* d_partials, int64_t n_part, const float* d_x, const float* d_dy,
                              const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta,
                              float eps, int32_t relu, float* d_dx, float* d_dgamma, float* d_dbeta,
                              const float* d_addend, int64_t M, int32_t C, void* d_ws, int64_t ws_bytes, void* d_sync,
                              void* stream);
/* (with d_dx the reduction finish and the apply pass are one launch where wsis_bn_stats_finalize_apply's conditions
 * hold; identical results) */
/* y = relu?( (x-mean)*rsqrt(var+eps)*gamma + beta )   (gamma/beta may be NULL = 1/0) */
int wsis_bn_apply(const float* d_x, const float* d_mean, const float* d_var, const float* d_gamma,
                  const float* d_beta, float eps, int32_t relu, float* d_y, int64_t M, int32_t C, void* stream);
/* backward of the fused BN(+ReLU): dgamma = sum dz*xhat, dbeta = sum dz (dz = dy masked by the ReLU),
 * dx = gamma*rstd*(dz - dbeta/M - xhat*dgamma/M) when training, gamma*rstd*dz otherwise; d_dx may be NULL.
 * d_addend (optional, [M,C]) is added to dx in the same pass: the gradient arriving over the residual skip
 * connection of sparse_unet3d.py:164-172, which autograd would otherwise add with one more kernel. */
int wsis_bn_bwd(const float* d_x, const float* d_dy, const float* d_mean, const float* d_var,
                const float* d_gamma, const float* d_beta, float eps, int32_t relu, int32_t training,
                float* d_dx, float* d_dgamma, float* d_dbeta, const float* d_addend, int64_t M, int32_t C, void* d_ws,
                int64_t ws_bytes, void* stream);

/* the apply pass of wsis_bn_bwd alone, from reduced sums: dx = gamma*rstd*(dz - d_sum_dz/M - xhat*d_sum_dz_xhat/M)
 * (+ addend).  For BatchNorm statistics shared across ranks (torch.nn.SyncBatchNorm, train_scannetv2.py:734-736:
 * mean / var over all ranks' rows) the caller all-reduces the two sum vectors of wsis_bn_bwd (d_dx = NULL) and passes
 * them scaled by M / N_global. */
int wsis_bn_bwd_apply(const float* d_x, const float* d_dy, const float* d_mean, const float* d_var,
                      const float* d_gamma, const float* d_beta, const float* d_sum_dz_xhat, const float* d_sum_dz,
                      float eps, int32_t relu, float* d_dx, const float* d_addend, int64_t M, int32_t C, void* stream);

/* ---- a14/a15: row gather and torch_scatter.scatter  backbone_3D_WSIS.py:179,188,225,232,244 --
 * CSR of a (possibly unsorted) index vector: d_perm int32 [N] = stable argsort(index),
 * d_offsets int32 [S+1].  d_index is int64 [N] (torch_scatter takes LongTensor). */
int64_t wsis_segment_csr_workspace_bytes(int64_t N, int64_t S);
int wsis_segment_csr(const int64_t* d_index, int64_t N, int64_t S, int32_t* d_perm, int32_t* d_offsets,
                     void* d_ws, int64_t ws_bytes, void* stream);
/* The CSRs of up to 8 index vectors from ONE sort (a batch of 3D-WSIS needs six: superpoint ids, p2v map, the two
 * directions of the affinity graph and of the ECC graph): h_index[t] device int64 [h_N[t]], h_S[t] segments.  Table t's
 * perm is d_perm_all[sum_{u<t} h_N[u] ...] (row numbers local to the table), its offsets d_offsets_all[sum_{u<t}
 * (h_S[u] + 1) ...]; identical to wsis_segment_csr per table. */
int64_t wsis_segment_csr_batch_workspace_bytes(int64_t N_all);
int wsis_segment_csr_batch(int32_t n, const void* const* h_index, const int64_t* h_N, const int64_t* h_S,
                           int32_t* d_perm_all, int32_t* d_offsets_all, void* d_ws, int64_t ws_bytes, void* stream);
/* reduce: 0 = sum, 1 = mean (sum / max(count,1)), 2 = max (empty segments -> 0).
 * out [S,C]; d_argmax int32 [S,C] only for max (row index into src, -1 for empty). Points are
 * accumulated in ascending original position => deterministic. */
int wsis_segment_reduce_fwd(const float* d_src, const int32_t* d_perm, const int32_t* d_offsets,
                            float* d_out, int32_t* d_argmax, int64_t N, int64_t S, int32_t C,
                            int32_t reduce, void* stream);
/* dsrc [N,C]: sum -> dout[index[p]]; mean -> dout[index[p]]/max(cnt,1); max -> scattered to argmax
 * rows (dsrc must be zero-initialised for max). */
int wsis_segment_reduce_bwd(const float* d_dout, const int64_t* d_index, const int32_t* d_offsets,
                            const int32_t* d_argmax, float* d_dsrc, int64_t N, int64_t S, int32_t C,
                            int32_t reduce, void* stream);
/* out[p,:] = src[idx[p],:] (int32 or int64 index selected by idx_is_64). */
int wsis_gather_rows(const float* d_src, const void* d_idx, int32_t idx_is_64, float* d_out, int64_t N,
                     int32_t C, void* stream);

/* Position encoding of the edge affinity (backbone_3D_WSIS.py:54-58 fc_position = Linear(3,16) -> ReLU -> Linear(16,1),
 * :222-224 applied to centre[u_e] - centre[v_e]): pos [E] in one launch; backward (no gradient for the centres: they are
 * data) writes every non-NULL dW1 [16,3] / db1 [16] / dW2 [1,16] / db2 [1] in two launches, fixed summation order. */
int64_t wsis_pos_enc_workspace_bytes(int64_t E);
int wsis_pos_enc_fwd(const float* d_centre, const int64_t* d_eu, const int64_t* d_ev, const float* d_W1, const float* d_b1,
                     const float* d_W2, const float* d_b2, float* d_pos, int64_t E, void* stream);
int wsis_pos_enc_bwd(const float* d_centre, const int64_t* d_eu, const int64_t* d_ev, const float* d_W1, const float* d_b1,
                     const float* d_W2, const float* d_dpos, float* d_dW1, float* d_db1, float* d_dW2, float* d_db2, int64_t E,
                     void* d_ws, int64_t ws_bytes, void* stream);

/* ---- a16: edge affinity attention  backbone_3D_WSIS.py:218-249 ------------------------------
 * logit_e = (q[u_e].k[v_e]) * scale * pos_enc_e ; a = segment softmax over edges sharing u ;
 * res[u,:] = sum_e a_e v[v_e,:].  CSR over u: d_perm_u/d_off_u from wsis_segment_csr(edge_u).
 * res [Su,D] with Su = max(u)+1 (rows without edges -> 0). */
int wsis_edge_affinity_fwd(const float* d_q, const float* d_k, const float* d_v, const float* d_pos,
                           const int64_t* d_eu, const int64_t* d_ev, const int32_t* d_perm_u,
                           const int32_t* d_off_u, float scale, float* d_aff, float* d_res, int64_t E,
                           int64_t Su, int32_t D, void* stream);
/* backward: inputs dAff [E] (may be null), dRes [Su,D].  Outputs dq [S,D] (rows >= Su and rows
 * without edges zero), dk, dv [S,D], dpos [E].  Needs CSR over v as well.  d_tmp float [2*E]. */
int wsis_edge_affinity_bwd(const float* d_q, const float* d_k, const float* d_v, const float* d_pos,
                           const float* d_aff, const int64_t* d_eu, const int64_t* d_ev,
                           const int32_t* d_perm_u, const int32_t* d_off_u, const int32_t* d_perm_v,
                           const int32_t* d_off_v, float scale, const float* d_daff,
                           const float* d_dres, float* d_dq, float* d_dk, float* d_dv, float* d_dpos,
                           float* d_tmp, int64_t E, int64_t S, int64_t Su, int32_t D, void* stream);

/* ---- a21: edge-conditioned message passing of the superpoint GNN --------------------------
 * modules/model/spg_modules.py:97-121,168-183 (PyG NNConv, flow=target_to_source, aggr='mean', vv=False):
 *   out[s,:] = mean_{e: src_e = s} x[dst_e,:] @ W_e,  W_e = d_w[e] in R^{C x C}, C <= 32.
 * CSR over the sources (forward) and over the targets (backward) from wsis_segment_csr.  Backward writes
 * dx [S,C] and the per-edge filter gradient dw [E,C,C] (every entry written). */
/* The same messages WITHOUT the per-edge [E, C*C] filter tensor: the filter is affine in the fnet hidden
 * state h_e in R^64 (graphnet.py:19-36: W_e = reshape(Wl h_e + bl)), so m_e = x_t @ W_e = sum_c h_e[c] U_t[c,:] + U_t[64,:]
 * with the per-NODE tensor U [S, 65*32], U_t[c,b] = sum_a x_t[a] Wl[a*32+b, c], row 64 = the bias term (one small GEMM
 * per GRU step, done by the caller).  d_h [E,64], d_U [S,65*32], CSR over the targets; forward writes m [E,32]
 * (every in-edge of every target), backward writes dU [S,65*32] (zeros for targets without in-edge) and dh [E,64].
 * C = 32, hidden width 64 (the model's 'gru_7_0' configuration); one wavefront per target, fixed order. */
int wsis_ecc_contract_fwd(const float* d_h, const float* d_U, const int32_t* d_perm_dst, const int32_t* d_off_dst,
                          float* d_m, int64_t S, int64_t E, void* stream);
int wsis_ecc_contract_bwd(const float* d_h, const float* d_U, const float* d_dm, const int32_t* d_perm_dst,
                          const int32_t* d_off_dst, float* d_dU, float* d_dh, int64_t S, int64_t E, void* stream);
/* accumulate != 0: dh += (every edge row is owned by one wavefront: still fixed order); the R repeats of the
 * recurrence share h, so its gradient is summed in place instead of by R - 1 extra launches */
int wsis_ecc_contract_bwd_acc(const float* d_h, const float* d_U, const float* d_dm, const int32_t* d_perm_dst,
                              const int32_t* d_off_dst, float* d_dU, float* d_dh, int64_t S, int64_t E,
                              int32_t accumulate, void* stream);
/* the same with the backward of the segmented mean that follows the messages folded in (spg_modules.py:97-121,
 * aggr='mean'): dm[e,:] = d_dinp[src_e,:] / out-degree(src_e) is formed on the fly from the source index / CSR offsets */
int wsis_ecc_contract_bwd_mean(const float* d_h, const float* d_U, const float* d_dinp, const int64_t* d_src_index,
                               const int32_t* d_off_src, const int32_t* d_perm_dst, const int32_t* d_off_dst, float* d_dU,
                               float* d_dh, int64_t S, int64_t E, int32_t accumulate, void* stream);
/* the dense product in front of the contraction (graphnet.py:19-36 folded as above): U [S,65*32] = hx [S,32] @ W' [32,65*32] */
int wsis_ecc_u_fwd(const float* d_hx, const float* d_W, float* d_U, int64_t S, void* stream);
int wsis_ecc_message_fwd(const float* d_x, const float* d_w, const int64_t* d_dst, const int32_t* d_perm_src,
                         const int32_t* d_off_src, float* d_out, int64_t S, int64_t E, int32_t C, void* stream);
int wsis_ecc_message_bwd(const float* d_x, const float* d_w, const float* d_dout, const int64_t* d_src,
                         const int32_t* d_perm_dst, const int32_t* d_off_dst, const int32_t* d_off_src,
                         float* d_dx, float* d_dw, int64_t S, int64_t E, int32_t C, void* stream);

/* Column sums out[c] = sum_r x[r, c] of x [M, C] (C % 4 == 0, C <= 1024): the bias gradient of the point-level
 * Linear layers (backbone_3D_WSIS.py:59-64, `dy.sum(0)` in torch's AddmmBackward).  One launch, two levels, fixed
 * summation order (run-to-run identical); inputs of more than one chunk need a sync slot (d_sync, see
 * wsis_sync_bytes). */
int64_t wsis_colsum_workspace_bytes(int64_t M, int32_t C);
int wsis_colsum(const float* d_x, int64_t M, int32_t C, float* d_out, void* d_ws, int64_t ws_bytes, void* d_sync,
                void* stream);

/* Prediction heads over the superpoint rows: replaces the per-layer module chains of
 * modules/model/backbone_3D_WSIS.py:59-64 (`head(cin, cout)` = Linear(cin,cin) -> BatchNorm1d -> ReLU -> Linear(cin,cout)),
 * :195-204 (sp_sem_seg / sp_offset_vector_head / sp_occupancy_head / sp_ins_size_head on the GNN output), :210-216 (the
 * bias-free w_qs / w_ks / w_vs on the same rows) and :253 (feature_term), cin == 64.  All blocks of one call read the same
 * input x [S,64]: blocks 0 .. n_heads-1 are heads (cout <= 32), blocks n_heads .. n_heads+n_lin-1 plain Linear(64,64)
 * layers without bias.  Forward = 3 launches, backward = 4 (csrc/heads.hip), fixed summation orders, exact fp32.
 * The tables hold DEVICE pointers in torch's layouts (Linear weight [out,in]); the struct itself is host memory.
 *   forward   reads W1,b1,gamma,beta,W2,b2 (+ running_* in eval mode), writes hidden[p] [S,64] (heads: the pre-BatchNorm
 *             activations, kept for backward; plain layers: their OUTPUT), out[p] [S,cout[p]], d_saved [n_heads,2,64]
 *             (mean, 1/sqrt(var+eps)); training != 0 uses batch statistics and updates running_* (NULL: not tracked)
 *   backward  reads dout[p] (heads: [S,cout[p]]; plain layers: [S,64]; NULL = zero), hidden, d_saved; writes d_dx [S,64]
 *             and every non-NULL dW1 [64,64] / db1 [64] / dgamma / dbeta [64] / dW2 [cout,64] / db2 [cout] (overwritten) */
#define WSIS_HEADS_MAX 8
typedef struct wsis_heads {
  int32_t n_heads, n_lin;
  int32_t cout[WSIS_HEADS_MAX];
  const float* W1[WSIS_HEADS_MAX];
  const float* b1[WSIS_HEADS_MAX];
  const float* gamma[WSIS_HEADS_MAX];
  const float* beta[WSIS_HEADS_MAX];
  const float* W2[WSIS_HEADS_MAX];
  const float* b2[WSIS_HEADS_MAX];
  float* running_mean[WSIS_HEADS_MAX];
  float* running_var[WSIS_HEADS_MAX];
  float* hidden[WSIS_HEADS_MAX];
  float* out[WSIS_HEADS_MAX];
  const float* dout[WSIS_HEADS_MAX];
  float* dW1[WSIS_HEADS_MAX];
  float* db1[WSIS_HEADS_MAX];
  float* dgamma[WSIS_HEADS_MAX];
  float* dbeta[WSIS_HEADS_MAX];
  float* dW2[WSIS_HEADS_MAX];
  float* db2[WSIS_HEADS_MAX];
} wsis_heads;
int64_t wsis_heads_workspace_bytes(int64_t S, int32_t n_heads, int32_t n_lin);
int wsis_heads_fwd(const wsis_heads* h, const float* d_x, int64_t S, float eps, float momentum, int32_t training,
                   float* d_saved, void* d_ws, int64_t ws_bytes, void* stream);
int wsis_heads_bwd(const wsis_heads* h, const float* d_x, int64_t S, int32_t training, const float* d_saved, float* d_dx,
                   void* d_ws, int64_t ws_bytes, void* stream);

/* GRUCellEx of the superpoint GNN (modules/model/spg_modules.py:207-253: GRU cell + input gate + per-row
 * normalisation of the gate pre-activations), C == 32: one kernel forward, one backward + a fixed-order reduce of
 * the parameter gradients.  Weights in torch.nn.GRUCell layout: Wih/Whh [3C,C] (r,z,n blocks), Wig [C,C]. */
int64_t wsis_gru_cell_workspace_bytes(int64_t S);
int wsis_gru_cell_fwd(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big,
                      const float* d_Wih, const float* d_Whh, const float* d_bih, const float* d_bhh, float* d_hy,
                      int64_t S, int32_t C, void* stream);
int wsis_gru_cell_bwd(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big,
                      const float* d_Wih, const float* d_Whh, const float* d_bih, const float* d_bhh,
                      const float* d_dhy, float* d_dx, float* d_dh, float* d_dWig, float* d_dbig, float* d_dWih,
                      float* d_dWhh, float* d_dbih, float* d_dbhh, int64_t S, int32_t C, void* d_ws,
                      int64_t ws_bytes, void* stream);
/* The same backward as evaluation `slot` of a sequence of `n_slots` evaluations that share the six parameter
 * tensors (the R repeats of spg_modules.py:152-185): every evaluation leaves its parameter-gradient slabs in its own
 * region of d_ws (n_slots x the single-call workspace); the call with finish != 0 reduces ALL regions in one
 * fixed-order launch into d_dW* / d_db* (= the sum over the sequence; may be NULL on the other calls).
 * The upstream gradient of the evaluation is d_dhy + d_dhy2 (either may be NULL): d_dhy2 [S, pitch >= 32] is the output
 * gradient of this hidden state inside the concatenated sequence (cat_all), added on load instead of by a launch. */
int wsis_gru_cell_bwd_seq(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big,
                          const float* d_Wih, const float* d_Whh, const float* d_bih, const float* d_bhh,
                          const float* d_dhy, const float* d_dhy2, int64_t dhy2_pitch, float* d_dx, float* d_dh, float* d_dWig,
                          float* d_dbig, float* d_dWih, float* d_dWhh, float* d_dbih, float* d_dbhh, int64_t S, int32_t C,
                          int32_t slot, int32_t n_slots, int32_t finish, void* d_ws, int64_t ws_bytes, void* stream);
/* wsis_gru_cell_fwd whose input is the segmented MEAN of the edge messages d_m [E,32] over the CSR (d_perm, d_off) of the
 * rows (spg_modules.py:97-121 aggr='mean' followed by the cell, :168-183): the mean is formed per row in the order of
 * wsis_segment_reduce_fwd and written to d_x_out [S,32] (kept for the backward) -- one launch instead of two. */
int wsis_gru_cell_fwd_mean(const float* d_m, const int32_t* d_perm, const int32_t* d_off, float* d_x_out, const float* d_h,
                           const float* d_Wig, const float* d_big, const float* d_Wih, const float* d_Whh, const float* d_bih,
                           const float* d_bhh, float* d_hy, int64_t S, int32_t C, void* stream);

/* ---- a17: dense inter-superpoint affinity + label propagation -------------------------------
 * train_scannetv2.py:562-570, modules/datasets/scannetv2_dataset.py:664-721 (fp64, host numpy).
 * A [S,S] fp64 zero-filled then A[u_e, v_e] = aff_e (edge order, later edges win). */
int wsis_affinity_dense_build(const int64_t* d_eu, const int64_t* d_ev, const float* d_aff, int64_t E,
                              double* d_A, int64_t S, void* stream);
/* T0 = rownorm(A * adj * sem_c) for one class c:
 *   sem[r][j] = (m[r] && m[j]) || (r==j && label[r]==c), m[r] = (pred[r]==c && conf[r]>thr);
 *   adj [S,S] uint8 (adjacency + I); rows summing to 0 are divided by 1. */
int wsis_affinity_transition(const double* d_A, const uint8_t* d_adj, const int32_t* d_pred,
                             const float* d_conf, const int32_t* d_label, int32_t cls, float thr,
                             double* d_T0, int64_t S, void* stream);
/* C = A @ B, fp64 [S,S] row-major on the f64 matrix cores (v_mfma_f64_16x16x4_f64). */
int wsis_dgemm(const double* d_A, const double* d_B, double* d_C, int64_t M, int64_t N, int64_t Kd,
               void* stream);
/* column-wise max / first argmax of T restricted to rows with label[r]==c (other rows count as 0):
 * scores fp64 [S], arg int32 [S]  (np.max/np.argmax(axis=0) semantics, first max wins). */
int wsis_affinity_colmax(const double* d_T, const int32_t* d_label, int32_t cls, double* d_scores,
                         int32_t* d_arg, int64_t S, void* stream);
/* The whole per-scene propagation of scannetv2_dataset.py:689-721 for ALL present classes in one call, exploiting that
 * T0_c = rownorm(A * adj * sem_c) is as sparse as the edge list and that only the rows of the superpoints labelled c of
 * T0_c^(iterations+1) are looked at: W0 = A * adj as CSR (d_col / d_val: caller buffers of nnz_cap >= nnz(A) entries),
 * per class row sums, then per labelled superpoint a chain of (row vector) x (sparse matrix) products in LDS (fp64,
 * k-ascending sums = the order of a dot product) and the column max / first argmax of wsis_affinity_colmax.
 * d_cls_of int32 [n_present] = the present classes ascending, d_ci_of_cls int32 [class_num] = index into d_cls_of or -1;
 * d_scores fp64 [n_present, S], d_arg int32 [n_present, S].  S <= 8188 (two rows of S doubles in LDS); beyond that use
 * wsis_affinity_transition / wsis_dgemm / wsis_affinity_colmax. */
int64_t wsis_affinity_propagate_sparse_workspace_bytes(int64_t S, int32_t n_present);
int wsis_affinity_propagate_sparse(const double* d_A, const uint8_t* d_adj, const int32_t* d_pred, const float* d_conf,
                                   const int32_t* d_label, const int32_t* d_cls_of, const int32_t* d_ci_of_cls,
                                   int32_t n_present, int32_t class_num, float thr, int32_t iterations, int64_t S,
                                   int32_t* d_col, double* d_val, int64_t nnz_cap, double* d_scores, int32_t* d_arg,
                                   void* d_ws, int64_t ws_bytes, void* stream);

/* ---- a19/a20: ballquery_batch_p / bfs_cluster [UPSTREAM PG_OP] ------------------------------
 * For point p: ascending indices k of same-batch points with |x_p-x_k|^2 < r^2 (strict, incl. p),
 * capped at 1000.  Deterministic offsets = exclusive prefix sum of the counts.
 * Pass 1 writes d_start_len int32 [N,2] and d_total int32[1]; pass 2 fills d_idx int32 [total] and must get
 * the SAME workspace back untouched (it holds the uniform grid: points radix-sorted by cell of size ~radius and
 * a cell hash; each point visits 27 cells instead of its whole batch item).  B < 1024, |coordinate| < 2^17 cells. */
int64_t wsis_ballquery_workspace_bytes(int64_t N);
int wsis_ballquery_count(const float* d_xyz, const int32_t* d_batch_idx, const int32_t* d_batch_off,
                         int64_t N, int32_t B, float radius, int32_t* d_start_len, int32_t* d_total,
                         void* d_ws, int64_t ws_bytes, void* stream);
int wsis_ballquery_fill(const float* d_xyz, const int32_t* d_batch_idx, const int32_t* d_batch_off,
                        int64_t N, int32_t B, float radius, const int32_t* d_start_len, int32_t* d_idx,
                        int64_t total, void* d_ws, int64_t ws_bytes, void* stream);

/* ---- a20 on the device: pointgroup_ops.bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold) [UPSTREAM
 * PointGroup, host code] with CUDA tensors.  wsis_cc_same_label: connected components of the ball-query graph
 * restricted to equal labels, d_root[i] = smallest point index of i's component (the host walk's seed),
 * d_size[root] = its point count (d_parent: int32 [N] scratch).  The caller keeps the components of >= threshold
 * points, numbers them in seed order (d_seeds, exclusive prefix d_offsets [n+1]) and wsis_bfs_order writes
 * cluster_idxs [sum, 2] = (cluster id, point) in the host walk's FIFO discovery order: one workgroup per cluster,
 * d_pos int32 [N] preset to -1, d_stamp int32 [N] preset to INT32_MAX. */
int wsis_cc_same_label(const int32_t* d_sem, const int32_t* d_idx, const int32_t* d_start_len, int64_t N,
                       int32_t* d_parent, int32_t* d_root, int32_t* d_size, void* stream);
int wsis_bfs_order(const int32_t* d_idx, const int32_t* d_start_len, const int32_t* d_sem, const int32_t* d_seeds,
                   const int32_t* d_offsets, int64_t n_clusters, int32_t* d_pos, int32_t* d_stamp,
                   int32_t* d_cluster_idxs, void* stream);

/* ---- a18 (point-level part): semantic loss of MultiTaskLoss.forward (losses_3D_WSIS.py:52-67 of the reference):
 * CrossEntropyLoss(ignore_index) + mean_c(1 - dice_c) with dice_c = (2 sum p_c y_c + 1e-5) / (sum p_c^2 + sum y_c
 * + 1e-4 + 1e-5) over the rows whose label != ignore_label, p = softmax(scores).  d_scores fp32 [N,C] (C <= 32),
 * d_labels int64 [N].  fwd: d_out2[0] = loss, d_out2[1] = number of kept rows; d_saved fp32 [2C+1] feeds bwd.
 * bwd: d_dscores [N,C] = *d_grad_loss * dloss/dscores (zero rows for ignored labels).  Deterministic. */
int64_t wsis_semantic_loss_workspace_bytes(int64_t N);
int wsis_semantic_loss_fwd(const float* d_scores, const int64_t* d_labels, int64_t N, int32_t C, int64_t ignore_label,
                           float* d_out2, float* d_saved, void* d_ws, int64_t ws_bytes, void* stream);
int wsis_semantic_loss_bwd(const float* d_scores, const int64_t* d_labels, int64_t N, int32_t C, int64_t ignore_label,
                           const float* d_saved, const float* d_grad_loss, float* d_dscores, void* stream);

/* Superpoint semantic term (losses_3D_WSIS.py:72-74: CrossEntropyLoss(ignore_index) on the [S,C] superpoint scores, mean
 * over the kept rows) with the logged scores.sum(): d_out3 = {loss, sum of all scores, n_kept}; one launch each way (the
 * forward finishes its workgroup sums by a ticket in the caller's sync slot, d_sync: see wsis_sync_bytes; S >= 1, C <= 32).
 * wsis_loss_sum: out = t_0 + t_1 + ... in order (the weighted sum of losses_3D_WSIS.py:130-151, all weights 1); bit i of
 * `paired` adds term i to term i+1 first (the reference's `offset_norm_loss + offset_dir_loss`).  d_terms is a HOST array
 * of n <= 8 device scalars. */
int64_t wsis_sp_ce_loss_workspace_bytes(int64_t S);
int wsis_sp_ce_loss_fwd(const float* d_scores, const int64_t* d_labels, int64_t S, int32_t C, int64_t ignore_label,
                        float* d_out3, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream);
int wsis_sp_ce_loss_bwd(const float* d_scores, const int64_t* d_labels, int64_t S, int32_t C, int64_t ignore_label,
                        const float* d_out3, const float* d_grad_loss, float* d_dscores, void* stream);
int wsis_loss_sum(const float* const* d_terms, int32_t n, uint32_t paired, float* d_out, void* stream);

/* Superpoint regression terms of the same loss (losses_3D_WSIS.py:79-96 offset L1 + cosine, :113-127 occupancy and
 * instance-size L1) over the rows whose two labels both differ from ignore_label: d_out5 = {offset_norm, offset_dir,
 * occupancy, instance_size, n_valid}.  bwd: gradients of the three predictions from four upstream scalars. */
int wsis_sp_regression_loss_fwd(const float* d_pred_off, const float* d_gt_off, const float* d_pred_occ,
                                const float* d_gt_occ, const float* d_pred_size, const float* d_gt_size,
                                const int64_t* d_sem_label, const int64_t* d_ins_label, int64_t S,
                                int64_t ignore_label, float* d_out5, void* stream);
int wsis_sp_regression_loss_bwd(const float* d_pred_off, const float* d_gt_off, const float* d_pred_occ,
                                const float* d_gt_occ, const float* d_pred_size, const float* d_gt_size,
                                const int64_t* d_sem_label, const int64_t* d_ins_label, int64_t S,
                                int64_t ignore_label, const float* d_out5, const float* d_g_norm,
                                const float* d_g_dir, const float* d_g_occ, const float* d_g_size, float* d_doff,
                                float* d_docc, float* d_dsize, void* stream);

/* Discriminative loss of one scene's superpoint embeddings (losses_3D_WSIS.py:157-230: pull delta_v, push on the L1
 * distance of the instance means with delta_d, regularisation), instances in n_slots <= 64 slots (slot = instance id;
 * the bound is known on the host), S <= 1536 rows, D = 7.  Rows count when both labels differ from ignore_label.
 * d_saved: wsis_disc_loss_saved_floats() floats, feeds bwd.  One workgroup, deterministic. */
int32_t wsis_disc_loss_saved_floats(void);
int wsis_disc_loss_fwd(const float* d_x, const int64_t* d_ins_label, const int64_t* d_sem_label, int64_t S,
                       int32_t D, int32_t n_slots, int64_t ignore_label, float delta_v, float delta_d, float p_var,
                       float p_dist, float p_reg, float* d_out1, float* d_saved, void* stream);
int wsis_disc_loss_bwd(const float* d_x, const int64_t* d_ins_label, const int64_t* d_sem_label, int64_t S,
                       int32_t D, int32_t n_slots, int64_t ignore_label, float delta_v, float delta_d, float p_var,
                       float p_dist, float p_reg, const float* d_saved, const float* d_grad_loss, float* d_dx,
                       void* stream);

/* ---- optimizer step (train_scannetv2.py:251; AdamW of config/ScanNet_v2_3D_WSIS.yaml:58-61) in one launch.
 * d_segments: device array of {float* p; const float* g; float* m; float* v; int64_t n; float step_size;
 * float inv_sqrt_bc2; float g_clamp; float pad;} (wsis_adamw_segment_bytes() bytes each; n == 0 skips the parameter;
 * g_clamp > 0 clamps the gradient to [-g_clamp, g_clamp] first and writes it back: train_scannetv2.py:247-249; step_size =
 * lr / (1 - beta1^t), inv_sqrt_bc2 = 1 / sqrt(1 - beta2^t) with t = updates of that parameter so far, this one
 * included); d_blocks int32 [n_blocks,2] = (segment, chunk of wsis_adamw_chunk() elements) for every chunk of every
 * segment.  Update rule of torch.optim.AdamW (decoupled weight decay), fp32. */
int32_t wsis_adamw_segment_bytes(void);
int32_t wsis_adamw_chunk(void);
int wsis_adamw_step(const void* d_segments, const int32_t* d_blocks, int64_t n_blocks, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void* stream);

/* ---- op-list executor: one C call issues a recorded forward or backward pass of the sparse UNet ------------
 * Replaces the Python-dispatched module walk of sparse_unet3d.py:103-350 (ResidualBlock.forward / UBlock.forward
 * and their autograd backward): the host records the pass as wsis_op records (plain device pointers + sizes) and
 * wsis_run_ops launches every kernel on `stream`.  Each op is exactly one of the single-op entry points above
 * (same kernels and summation orders -> bit-identical to calling them one by one):
 *   CONV         in: X, nbr, order, W, bias, residual          out: Y                      (wsis_spconv_fwd)
 *   BN_RELU      in: x, gamma, beta, running_mean, running_var out: y, mean, var           (wsis_bn_stats+apply)
 *   CAT          in: a [M,Cin], b [M,Cout]                     out: [M,Cin+Cout]           (torch.cat dim 1)
 *   SPLIT        in: [M,Cin+Cout]                              out: a [M,Cin], b [M,Cout]  (its backward)
 *   ADD          in: src [M_in*Cin]                            out: dst += src             (gradient fan-in)
 *   CONV_BWD     in: X, W, dY, nbr_f, order_f, nbr_b, order_b  out: dX (may be NULL), dW (may be NULL)
 *                M_in = rows of X / dX, M_out = rows of dY     (weight_transpose + wsis_spconv_fwd + wsis_spconv_dw)
 *   BN_RELU_BWD  in: x, dy, mean, var, gamma, beta, addend     out: dx, dgamma, dbeta      (wsis_bn_bwd)
 * BN ops use M_in rows and Cin channels.  d_ws from wsis_run_ops_workspace_bytes (max over the ops); d_sync: a
 * zero-filled block of wsis_sync_bytes() bytes (may be NULL: multi-launch forms).
 * Streams: everything the caller's later work depends on is ordered on `stream` when the call returns.  Work that the
 * chain of ops does not wait for runs on a library-owned side stream of `stream`, forked and joined with events inside
 * the call: the weight gradients of CONV_BWD ops (WSIS_DW_STREAM=0: on `stream`; forked in FRONT of their op's dIn launch
 * below WSIS_DW_EARLY_ROWS active voxels per batch -- they need X and dY, not dX; the weight gradient of a pass's last op
 * goes on `stream` itself, WSIS_DW_TAIL_MAIN) and the pass's weight transposes (WSIS_WT_SIDE; joined in front of the
 * first op that reads one).  Same kernels on either stream: the results do not depend on these switches. */
enum {
  WSIS_OP_CONV = 1, WSIS_OP_BN_RELU = 2, WSIS_OP_CAT = 3, WSIS_OP_SPLIT = 4, WSIS_OP_ADD = 5, WSIS_OP_CONV_BWD = 6,
  WSIS_OP_BN_RELU_BWD = 7
};
/* WSIS_OPF_STATS on CONV: out[1] = BatchNorm partials of the output (wsis_spconv_fwd_t d_stats).
 * WSIS_OPF_STATS on BN_RELU (training): the statistics come from partials instead of a pass over x: in[5] = partials
 * of channels [0, K), in[6] = partials of channels [K, Cin) (NULL when K == Cin; the input is a concatenation of two
 * producers' outputs), ceil(M_in / 32) rows each. */
/* WSIS_OPF_BN_IN on CONV: the input is BatchNorm(+ReLU, WSIS_OPF_RELU)(in[0]) applied on the fly (wsis_spconv_fwd_f):
 * in[6] = mean, in[7] = var, in[8] = gamma, in[9] = beta, eps = op.eps -- the BN_RELU op in front of it is then not in the
 * list at all (or only computes statistics: BN_RELU with out[0] == NULL).  On CONV_BWD: in[0] is that raw tensor and
 * in[7..10] = mean, var, gamma, beta: the weight gradient runs as wsis_spconv_dw_bn.
 * WSIS_OPF_STAT_FIN on CONV (with WSIS_OPF_STATS): the statistics of the output are finished inside the launch for
 * n = 1 or 2 BatchNorm layers: target 0 = out[2] mean, out[3] var, in[10] running_mean, in[11] running_var, op.momentum;
 * target 1 (when out[4] != NULL) = out[4], out[5], out[6], out[7], op.momentum2. */
enum {
  WSIS_OPF_RELU = 1, WSIS_OPF_TRAINING = 2, WSIS_OPF_UPDATE_RUNNING = 4, WSIS_OPF_FLIP = 8, WSIS_OPF_STATS = 16,
  WSIS_OPF_BN_IN = 32, WSIS_OPF_STAT_FIN = 64
};
typedef struct wsis_op {
  int32_t kind, flags;
  int64_t M_in, M_out;
  int32_t K, Cin, Cout, reserved;
  float eps, momentum;
  float momentum2, reserved2;
  const void* in[12];
  void* out[8];
} wsis_op;
int64_t wsis_run_ops_workspace_bytes(const wsis_op* ops, int32_t n);
int wsis_run_ops(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream);
/* The same with a milestone: once op `mark_op` (0-based) has been issued, `waiter_stream` is made to wait for
 * everything issued so far on `stream` and on the library's weight-gradient side stream.  The data-parallel step uses
 * it to start the RCCL all-reduce of the finished first part of the flat gradient buffer while the rest of the backward
 * pass still runs (the reference's DDP buckets, train_scannetv2.py:738).  mark_op < 0: no milestone. */
int wsis_run_ops_marked(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream,
                        int32_t mark_op, void* waiter_stream);
/* Creates the library's weight-gradient side stream of `stream` and binds both to their hardware queues (first command).
 * Call before creating an RCCL communicator in the same process (wsis_parallel.warm_streams does): streams take their
 * hardware queue in the order of their first use, and two streams of a step must not end up sharing one. */
int wsis_warm_streams(void* stream);
/* Launch-plan hint for the weight-gradient products (wsis_spconv_dw, wsis_spconv_dw_bn, the backward ops of wsis_run_ops):
 * `rows` = active voxels of the finest level of the batch being trained (train_scannetv2.py:191: the rows of the
 * SparseConvTensor the UNet receives).  From 250,000 rows (two ScanNet scenes per step) the launches take twice the
 * waves at every level: with several scenes per step the GPU is busy throughout and the gradients should get done
 * fast; with one scene they should stay out of the way of the dIn products.  0 (initial) = no hint.
 * Performance only -- results stay within the same tolerance, but the slab partition and with it the order of additions
 * of a weight gradient follow the plan: give the same hint to runs whose bits are compared.
 * PROCESS-GLOBAL and sticky: one word shared by every device, stream and the library's weight-gradient worker thread;
 * it keeps its value until the next call (the Python layer sets it at the top of every forward pass).  A caller of
 * wsis_spconv_dw / wsis_spconv_dw_bn that never calls this runs with the hint of whatever batch the process saw last. */
int wsis_hint_batch_rows(int64_t rows);
/* A pass issued in PARTS -- the host does something between two parts (the statistics exchange of a SyncBatchNorm layer:
 * train_scannetv2.py:734-736 converts every BatchNorm when num_gpus > 1; model/unet_native.py).  Parts with last == 0
 * leave the weight-gradient side stream un-joined; the part with last != 0 (n may be 0) joins everything forked since.
 * Every part needs its OWN workspace (wsis_run_ops_workspace_bytes of that part), alive until the last part returns. */
int wsis_run_ops_part(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream,
                      int32_t last);

#if defined(WSIS_EXPERIMENTAL) && WSIS_EXPERIMENTAL
/* A run of consecutive ops whose output tensors have at most WSIS_DEEP_ROWS (8192) rows -- the deep UNet levels of a
 * scene, sparse_unet3d.py:321-350 -- is issued as ONE resident launch (csrc/deep.hip: 256 workgroups walk the ops as
 * phases with grid barriers between them; same kernels' code, same order of additions, results identical to the
 * launch-by-launch form).  WSIS_DEEP=0 switches it off.  Counters of this process, for tests: */
int64_t wsis_deep_launches(void);
int64_t wsis_deep_phases(void);
#endif /* WSIS_EXPERIMENTAL */
/* 1: this library was built with EXPERIMENTAL=1 (the guarded entry points above exist), 0: the default build */
int32_t wsis_experimental(void);
#ifdef __cplusplus
}
#endif
#endif /* WSIS_HIP_H_ */
