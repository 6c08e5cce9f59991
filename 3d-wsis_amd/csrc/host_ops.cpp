// libwsis_host.so -- host-only operators of the 3D-WSIS hot path (no HIP runtime; safe to call
// from forked DataLoader workers, SURVEY.md 8b "Threading / streams").
//
//   voxelization_idx  [UPSTREAM PointGroup lib/pointgroup_ops voxelize_idx]
//       call sites in the reference: modules/datasets/scannetv2_dataset.py:449,528,
//       test_scannetv2.py:389
//   bfs_cluster       [UPSTREAM PointGroup bfs_cluster] (no call site in the reference; named by
//       BASELINE.json north_star)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <cmath>
#include <vector>

#include "../../include/wsis_hip.h"

namespace {
thread_local std::string g_err;
int fail(int code, const char* msg) {
  g_err = msg;
  return code;
}

inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
inline uint64_t hash4(const int64_t* c) {
  uint64_t h = mix64((uint64_t)c[0] + 0x9e3779b97f4a7c15ULL);
  h = mix64(h ^ (uint64_t)c[1]);
  h = mix64(h ^ (uint64_t)c[2]);
  h = mix64(h ^ (uint64_t)c[3]);
  return h;
}
}  // namespace

extern "C" {

int wsis_host_version(void) { return WSIS_ABI_VERSION; }
const char* wsis_host_last_error(void) { return g_err.c_str(); }

int wsis_host_voxelize_idx_map(const int64_t* h_coords, int64_t N, int32_t* h_p2v, int64_t* M,
                               int32_t* max_active) {
  if (N < 0 || (N > 0 && (!h_coords || !h_p2v)) || !M || !max_active)
    return fail(WSIS_ERR_ARG, "voxelize_idx_map: null pointer or negative N");
  if (N > 0x3fffffff) return fail(WSIS_ERR_ARG, "voxelize_idx_map: N too large for int32 maps");
  uint64_t cap = 16;
  while (cap < (uint64_t)N * 2) cap <<= 1;
  const uint64_t mask = cap - 1;
  // slot -> first point of the voxel (-1 empty); voxel id of that point is p2v[first]
  std::vector<int32_t> slot(cap, -1);
  std::vector<int32_t> count;
  count.reserve((size_t)N / 2 + 1);
  int32_t nvox = 0;
  for (int64_t p = 0; p < N; ++p) {
    const int64_t* c = h_coords + 4 * p;
    uint64_t h = hash4(c) & mask;
    for (;;) {
      int32_t f = slot[h];
      if (f < 0) {
        slot[h] = (int32_t)p;
        h_p2v[p] = nvox++;
        count.push_back(1);
        break;
      }
      const int64_t* d = h_coords + 4 * (int64_t)f;
      if (d[0] == c[0] && d[1] == c[1] && d[2] == c[2] && d[3] == c[3]) {
        int32_t v = h_p2v[f];
        h_p2v[p] = v;
        ++count[v];
        break;
      }
      h = (h + 1) & mask;
    }
  }
  int32_t ma = 0;
  for (int32_t v = 0; v < nvox; ++v)
    if (count[v] > ma) ma = count[v];
  *M = nvox;
  *max_active = ma;
  return WSIS_OK;
}

int wsis_host_voxelize_idx_fill(const int64_t* h_coords, int64_t N, const int32_t* h_p2v, int64_t M,
                                int32_t max_active, int64_t* h_voxel_locs, int32_t* h_v2p) {
  if (N < 0 || M < 0 || max_active < 0) return fail(WSIS_ERR_ARG, "voxelize_idx_fill: negative size");
  if (N > 0 && (!h_coords || !h_p2v || !h_voxel_locs || !h_v2p))
    return fail(WSIS_ERR_ARG, "voxelize_idx_fill: null pointer");
  const int64_t stride = 1 + (int64_t)max_active;
  if (M > 0) std::memset(h_v2p, 0, sizeof(int32_t) * (size_t)(M * stride));
  for (int64_t p = 0; p < N; ++p) {
    int32_t v = h_p2v[p];
    if (v < 0 || v >= M) return fail(WSIS_ERR_ARG, "voxelize_idx_fill: p2v out of range");
    int32_t* row = h_v2p + (int64_t)v * stride;
    int32_t n = row[0];
    if (n >= max_active) return fail(WSIS_ERR_OVERFLOW, "voxelize_idx_fill: max_active too small");
    if (n == 0) std::memcpy(h_voxel_locs + 4 * (int64_t)v, h_coords + 4 * p, 4 * sizeof(int64_t));
    row[1 + n] = (int32_t)p;
    row[0] = n + 1;
  }
  return WSIS_OK;
}

int wsis_host_bfs_cluster_count(const int32_t* h_semantic, const int32_t* h_ball_idx,
                                const int32_t* h_start_len, int64_t N, int32_t threshold,
                                int32_t* h_point_cluster, int32_t* h_order, int64_t* n_clusters,
                                int64_t* n_points) {
  if (N < 0 || !n_clusters || !n_points) return fail(WSIS_ERR_ARG, "bfs_cluster_count: bad args");
  if (N > 0 && (!h_semantic || !h_start_len || !h_point_cluster || !h_order))
    return fail(WSIS_ERR_ARG, "bfs_cluster_count: null pointer");
  std::vector<uint8_t> visited((size_t)N, 0);
  std::vector<int32_t> queue;
  queue.reserve(1024);
  int64_t nc = 0, np = 0;
  for (int64_t i = 0; i < N; ++i) {
    h_point_cluster[i] = -1;
    h_order[i] = -1;
  }
  for (int64_t i = 0; i < N; ++i) {
    if (visited[i]) continue;
    queue.clear();
    queue.push_back((int32_t)i);
    visited[i] = 1;
    const int32_t label = h_semantic[i];
    size_t head = 0;
    while (head < queue.size()) {
      int32_t cur = queue[head++];
      int32_t start = h_start_len[2 * (int64_t)cur], len = h_start_len[2 * (int64_t)cur + 1];
      for (int32_t j = 0; j < len; ++j) {
        int32_t nb = h_ball_idx[(int64_t)start + j];
        if (nb < 0 || nb >= N) return fail(WSIS_ERR_ARG, "bfs_cluster_count: neighbour out of range");
        if (visited[nb] || h_semantic[nb] != label) continue;
        visited[nb] = 1;
        queue.push_back(nb);
      }
    }
    if ((int64_t)queue.size() >= (int64_t)threshold) {
      for (size_t r = 0; r < queue.size(); ++r) {
        h_point_cluster[queue[r]] = (int32_t)nc;
        h_order[queue[r]] = (int32_t)r;
      }
      ++nc;
      np += (int64_t)queue.size();
    }
  }
  *n_clusters = nc;
  *n_points = np;
  return WSIS_OK;
}

int wsis_host_bfs_cluster_fill(const int32_t* h_point_cluster, const int32_t* h_order, int64_t N,
                               int64_t n_clusters, int64_t n_points, int32_t* h_cluster_idxs,
                               int32_t* h_cluster_offsets) {
  if (N < 0 || n_clusters < 0 || n_points < 0 || !h_cluster_offsets)
    return fail(WSIS_ERR_ARG, "bfs_cluster_fill: bad args");
  std::vector<int64_t> size((size_t)n_clusters, 0);
  for (int64_t p = 0; p < N; ++p) {
    int32_t c = h_point_cluster[p];
    if (c >= 0) {
      if (c >= n_clusters) return fail(WSIS_ERR_ARG, "bfs_cluster_fill: cluster id out of range");
      ++size[c];
    }
  }
  int64_t acc = 0;
  for (int64_t c = 0; c < n_clusters; ++c) {
    h_cluster_offsets[c] = (int32_t)acc;
    acc += size[c];
  }
  h_cluster_offsets[n_clusters] = (int32_t)acc;
  if (acc != n_points) return fail(WSIS_ERR_ARG, "bfs_cluster_fill: n_points mismatch");
  for (int64_t p = 0; p < N; ++p) {
    int32_t c = h_point_cluster[p];
    if (c < 0) continue;
    int64_t r = (int64_t)h_cluster_offsets[c] + h_order[p];
    h_cluster_idxs[2 * r] = c;
    h_cluster_idxs[2 * r + 1] = (int32_t)p;
  }
  return WSIS_OK;
}

// Superpoint-graph BFS of the test-time instance grouping (test_scannetv2.py:312-340, called from :372-381):
// seeds ascending over the superpoints of a valid class; a neighbour joins the seed's group iff it has the
// seed's class, is unvisited and its predicted instance centre lies closer than 0.25 * ins_size[seed] to the
// CURRENT superpoint's centre (fp32 norm, fp32 threshold, as numpy evaluates it).
// h_adj_off int32 [S+1] / h_adj int32 [nnz]: neighbours of every superpoint (igraph neighbors(mode='all')).
// Output: h_group int32 [S] = group id in seed order or -1; *n_groups.
int wsis_host_graph_bfs(const int32_t* h_label, const uint8_t* h_class_valid, int32_t n_class,
                        const float* h_centre, const float* h_ins_size, const int32_t* h_adj_off,
                        const int32_t* h_adj, int64_t S, int32_t* h_group, int64_t* n_groups) {
  if (S < 0 || n_class < 1 || !n_groups || (S > 0 && (!h_label || !h_class_valid || !h_centre || !h_ins_size ||
                                                      !h_adj_off || !h_group)))
    return fail(WSIS_ERR_ARG, "graph_bfs: bad args");
  for (int64_t s = 0; s < S; ++s) h_group[s] = -1;
  std::vector<int32_t> queue;
  int64_t ng = 0;
  for (int64_t seed = 0; seed < S; ++seed) {
    const int32_t lab = h_label[seed];
    if (lab < 0 || lab >= n_class) return fail(WSIS_ERR_ARG, "graph_bfs: class label out of range");
    if (!h_class_valid[lab] || h_group[seed] >= 0) continue;
    const float thr = 0.25f * h_ins_size[seed];
    queue.clear();
    queue.push_back((int32_t)seed);
    h_group[seed] = (int32_t)ng;
    for (size_t head = 0; head < queue.size(); ++head) {
      const int32_t cur = queue[head];
      const float* cc = h_centre + 3 * (int64_t)cur;
      for (int32_t e = h_adj_off[cur]; e < h_adj_off[cur + 1]; ++e) {
        const int32_t nb = h_adj[e];
        if (nb < 0 || nb >= S) return fail(WSIS_ERR_ARG, "graph_bfs: neighbour id out of range");
        if (h_label[nb] != lab || h_group[nb] >= 0) continue;
        const float* cn = h_centre + 3 * (int64_t)nb;
        const float dx = cc[0] - cn[0], dy = cc[1] - cn[1], dz = cc[2] - cn[2];
        const float d = std::sqrt((dx * dx + dy * dy) + dz * dz);
        if (d < thr) {
          h_group[nb] = (int32_t)ng;
          queue.push_back(nb);
        }
      }
    }
    ++ng;
  }
  *n_groups = ng;
  return WSIS_OK;
}

}  // extern "C"
