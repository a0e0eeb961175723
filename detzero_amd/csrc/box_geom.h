// Rotated-rectangle geometry shared by the kernels that need BEV overlaps (head_post.hip: IoU matrices, NMS;
// wbf.hip: box fusion).  Formula sequence of utils/detzero_utils/ops/iou3d_nms/src/iou3d_nms_kernel.cu:15-232,328-335
// in fp32; include it AFTER `#pragma clang fp contract(off)` so that discrete decisions (IoU > thr) agree with the CPU
// restatement wherever libm and ocml agree on sin/cos/atan2.
#pragma once
#include "common.h"

namespace dz {

// ------------------------------------------------------------------------------------------
// rotated rectangle overlap
// ------------------------------------------------------------------------------------------
struct P2 { float x, y; };
constexpr float GEO_EPS = 1e-8f;

__device__ __forceinline__ float cr2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cr3(P2 p1, P2 p2, P2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
__device__ __forceinline__ bool bbox_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
__device__ __forceinline__ bool corner_in_box(const float *box, P2 p) {
    const float MARGIN = 1e-2f;
    const float cx = box[0], cy = box[1];
    const float ac = cosf(-box[6]), as = sinf(-box[6]);
    const float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    const float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}
__device__ __forceinline__ bool edge_cross(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {
    if (!bbox_cross(p0, p1, q0, q1)) return false;
    const float s1 = cr3(q0, p1, p0);
    const float s2 = cr3(p1, q1, p0);
    const float s3 = cr3(p0, q1, q0);
    const float s4 = cr3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cr3(q1, p1, p0);
    if (fabsf(s5 - s1) > GEO_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}
__device__ __forceinline__ void spin(P2 c, float ac, float as, P2 &p) {
    const float nx = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    const float ny = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    p.x = nx; p.y = ny;
}

// Intersection polygon vertices live in LDS (column `tid` of a [16][nthreads] array): a dynamically
// indexed private array would be spilled to scratch memory, which we avoid in every kernel so that
// whole frames replay from hipGraphs without a scratch segment.
__device__ float rect_overlap(const float *A, const float *B, P2 *cp, float *ang, int ld) {
    const float adx = A[3] / 2, bdx = B[3] / 2, ady = A[4] / 2, bdy = B[4] / 2;
    const P2 ca{A[0], A[1]}, cb{B[0], B[1]};
    P2 a0{A[0] - adx, A[1] - ady}, a1{A[0] + adx, A[1] - ady}, a2{A[0] + adx, A[1] + ady}, a3{A[0] - adx, A[1] + ady};
    P2 b0{B[0] - bdx, B[1] - bdy}, b1{B[0] + bdx, B[1] - bdy}, b2{B[0] + bdx, B[1] + bdy}, b3{B[0] - bdx, B[1] + bdy};
    const float acs = cosf(A[6]), asn = sinf(A[6]), bcs = cosf(B[6]), bsn = sinf(B[6]);
    spin(ca, acs, asn, a0); spin(ca, acs, asn, a1); spin(ca, acs, asn, a2); spin(ca, acs, asn, a3);
    spin(cb, bcs, bsn, b0); spin(cb, bcs, bsn, b1); spin(cb, bcs, bsn, b2); spin(cb, bcs, bsn, b3);
    const P2 ac[5] = {a0, a1, a2, a3, a0};     // constant indices only after full unrolling -> registers
    const P2 bc[5] = {b0, b1, b2, b3, b0};

    P2 pc{0.f, 0.f};
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            P2 t;
            if (edge_cross(ac[i + 1], ac[i], bc[j + 1], bc[j], t)) {
                cp[cnt * ld] = t; pc.x = pc.x + t.x; pc.y = pc.y + t.y; ++cnt;
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (corner_in_box(A, bc[k])) { pc.x = pc.x + bc[k].x; pc.y = pc.y + bc[k].y; cp[cnt * ld] = bc[k]; ++cnt; }
        if (corner_in_box(B, ac[k])) { pc.x = pc.x + ac[k].x; pc.y = pc.y + ac[k].y; cp[cnt * ld] = ac[k]; ++cnt; }
    }
    pc.x /= cnt; pc.y /= cnt;   // cnt == 0 gives NaN exactly as the reference; the loops below do not run
    // bubble sort by polar angle around the centroid (angles cached; comparisons identical)
    for (int i = 0; i < cnt; ++i) ang[i * ld] = atan2f(cp[i * ld].y - pc.y, cp[i * ld].x - pc.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float x0 = ang[i * ld], x1 = ang[(i + 1) * ld];
            if (x0 > x1) {
                const P2 t = cp[i * ld]; cp[i * ld] = cp[(i + 1) * ld]; cp[(i + 1) * ld] = t;
                ang[i * ld] = x1; ang[(i + 1) * ld] = x0;
            }
        }
    float area = 0.f;
    const P2 c0 = cp[0];
    for (int k = 0; k < cnt - 1; ++k) {
        const P2 ck = cp[k * ld], cn = cp[(k + 1) * ld];
        const P2 u{ck.x - c0.x, ck.y - c0.y};
        const P2 v{cn.x - c0.x, cn.y - c0.y};
        area += cr2(u, v);
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float rect_iou(const float *A, const float *B, P2 *cp, float *ang, int ld) {
    const float sa = A[3] * A[4], sb = B[3] * B[4];
    const float so = rect_overlap(A, B, cp, ang, ld);
    return so / fmaxf(sa + sb - so, GEO_EPS);
}

}  // namespace dz
