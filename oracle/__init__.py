"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy / torch-CPU) of the algorithms of the 3D-WSIS hot path, each function citing the
reference file:line (or the [UPSTREAM] semantics of SURVEY.md App. A) it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product
(``3d-wsis_amd/``) never does and has no CPU fallback.

Pinning status (DESIGN.md "Oracle"):
  * losses            -- pinned against the imported reference ``modules/model/losses_3D_WSIS.py``
                         (tests/golden/loss_golden.npz, generator tests/golden/make_golden.py).
  * label propagation -- pinned against the imported reference ``ScanNetV2Inst_spg.weak_label_propagation``
                         (tests/golden/propagation_golden.npz, same generator).
  * spconv / pointgroup_ops / torch_scatter -- the sources are NOT in /root/reference (un-vendored
    dependencies: llijiang/spconv v1.0 fork, dvlab-research/PointGroup lib/pointgroup_ops master,
    torch_scatter 2.0.x; no commit pins) => "parity unpinned" at that boundary; the
    restatements are instead checked against independent constructions (dense F.conv3d /
    conv_transpose3d in fp64, scatter_reduce, np.unique, cKDTree, csgraph) in tests/test_oracle_*.py.
"""
