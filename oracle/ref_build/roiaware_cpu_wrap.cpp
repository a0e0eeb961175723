// ORACLE (test infrastructure only).  Thin C-ABI wrapper that COMPILES THE REFERENCE'S OWN SOURCE where it lies
// (utils/detzero_utils/ops/roiaware_pool3d/src/roiaware_pool3d.cpp under /root/reference, found through the include path set
// by oracle/refbuild.py) so that the points-in-box test restated in oracle/c/oracle.c can be pinned against the reference's
// own host implementation, points_in_boxes_cpu (roiaware_pool3d.cpp:248-295).  Nothing from the reference is copied into
// this repository.  The file also declares the CUDA launchers of its GPU entry points; they are given empty bodies here
// (never called) so that the shared object links without the .cu file.
#define TORCH_EXTENSION_NAME roiaware_pool3d_ref_unused
#include "roiaware_pool3d.cpp"

void roiaware_pool3d_launcher(int, int, int, int, int, int, int, const float *, const float *, const float *, int *, int *, float *, int) {}
void roiaware_pool3d_backward_launcher(int, int, int, int, int, int, const int *, const int *, const float *, float *, int) {}
void points_in_boxes_launcher(int, int, int, const float *, const float *, int *) {}
void points_in_boxes_v2_launcher(int, int, int, const float *, const float *, int *) {}
void points_in_multi_boxes_launcher(int, int, int, const float *, const float *, int *, int) {}
void points_in_boxes_num_launcher(int, int, int, const float *, const float *, int *) {}
void points_in_boxes_2dlauncher(int, int, int, const float *, const float *, int *) {}
void points_in_boxes_2dlauncher_v2(int, int, int, const float *, const float *, int *) {}

extern "C" void ref_points_in_boxes_cpu(const float *boxes, int nb, const float *pts, int np, int *out) {
    at::Tensor tb = torch::from_blob(const_cast<float *>(boxes), {nb, 7}, torch::kFloat32);
    at::Tensor tp = torch::from_blob(const_cast<float *>(pts), {np, 3}, torch::kFloat32);
    at::Tensor to = torch::from_blob(out, {nb, np}, torch::kInt32);
    points_in_boxes_cpu(tb, tp, to);
}
