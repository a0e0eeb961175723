"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of DetZero's per-frame detection hot path (and the refiner's secondary kernel set).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product package ``detzero_amd`` never does (tests/test_boundary.py greps for it).

Pinning status (see DESIGN.md "Oracle"):
  * rotated BEV IoU / overlap  - pinned against the reference's own C++ (iou3d_cpu.cpp compiled
    from /root/reference into oracle/_ref, see oracle/ref_build/) and committed fixtures.
  * MeanVFE, DynamicMeanVFE index math, BaseBEVBackbone, CenterHead convs, heat-map decode -
    pinned against outputs of the reference's Python modules imported in the build container
    (tests/golden/gen_golden.py -> tests/golden/*.npz).
  * hard voxelizer, sparse 3-D convolution semantics (spconv, un-vendored pip dependency) -
    PARITY UNPINNED: restated from spconv's documented behaviour, cross-checked against dense
    torch.nn.functional.conv3d on densified inputs.
"""
