// ORACLE (test infrastructure only).  Thin C-ABI wrapper that COMPILES THE REFERENCE'S OWN SOURCE
// where it lies (utils/detzero_utils/ops/iou3d_nms/src/iou3d_cpu.cpp under /root/reference, found
// through the include path set by oracle/refbuild.py) so the CPU restatement in oracle/c/oracle.c can
// be pinned against it.  Nothing from the reference is copied into this repository.
// The reference file marks some host functions `__device__` and includes <cuda.h>; both are
// neutralised here (empty stub headers in oracle/ref_build/stubs).
#define __device__
#include "iou3d_cpu.cpp"

extern "C" void ref_boxes_iou_bev_cpu(const float *a, int na, const float *b, int nb, float *out) {
    at::Tensor ta = torch::from_blob(const_cast<float *>(a), {na, 7}, torch::kFloat32);
    at::Tensor tb = torch::from_blob(const_cast<float *>(b), {nb, 7}, torch::kFloat32);
    at::Tensor to = torch::from_blob(out, {na, nb}, torch::kFloat32);
    boxes_iou_bev_cpu(ta, tb, to);
}
