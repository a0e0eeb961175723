/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (detzero_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-C restatement (single thread, fp32, no FMA contraction: built with -ffp-contract=off)
 * of the sequential algorithms on DetZero's per-frame detection path.  Each function cites the
 * reference file:line it restates.  Paths are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Hard voxelization.  Restates spconv v2 `Point2VoxelCPU3d.point_to_voxel` as it is called from
 * detection/detzero_det/datasets/processor/data_processor.py:69-83 (spconv itself is a pip
 * dependency that is NOT in the reference tree: semantics per SURVEY.md Appendix C, "parity
 * unpinned" for this function).
 *   - points visited in input order; c_j = floor((p_j - lo_j) / vs_j) in fp32
 *   - a point outside the grid on any axis is skipped
 *   - voxel ids are handed out in first-appearance order; once max_voxels exist, a point that
 *     would open a new voxel is skipped
 *   - a point is stored only while its voxel holds < max_points points
 * Outputs: voxels (max_voxels, max_points, C) zero padded, coords (max_voxels,3) int32 (z,y,x),
 *          num_points (max_voxels).  Returns the number of voxels.
 * `grid_lut` is a caller-provided int32 scratch of gx*gy*gz entries, all -1 on entry; restored
 * to -1 on exit (the reference keeps a dense coor_to_voxelidx grid the same way).
 * ------------------------------------------------------------------------------------------ */
int orc_voxelize_hard(const float *points, int n, int c, const float *range6, const float *vsize3,
                      const int *grid_xyz, int max_points, int max_voxels, float *voxels,
                      int *coords_zyx, int *num_points, int *grid_lut)
{
    const int gx = grid_xyz[0], gy = grid_xyz[1], gz = grid_xyz[2];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * c;
        int cc[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            float d = p[j] - range6[j];
            float q = d / vsize3[j];
            int v = (int)floorf(q);
            if (v < 0 || v >= grid_xyz[j]) { failed = 1; break; }
            cc[j] = v;
        }
        if (failed) continue;
        size_t lin = ((size_t)cc[2] * gy + cc[1]) * gx + cc[0];
        int vid = grid_lut[lin];
        if (vid == -1) {
            if (m >= max_voxels) continue;
            vid = m++;
            grid_lut[lin] = vid;
            coords_zyx[vid * 3 + 0] = cc[2];
            coords_zyx[vid * 3 + 1] = cc[1];
            coords_zyx[vid * 3 + 2] = cc[0];
            num_points[vid] = 0;
        }
        int k = num_points[vid];
        if (k < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + k) * c, p, sizeof(float) * c);
            num_points[vid] = k + 1;
        }
    }
    for (int v = 0; v < m; ++v) {
        size_t lin = ((size_t)coords_zyx[v * 3] * gy + coords_zyx[v * 3 + 1]) * gx + coords_zyx[v * 3 + 2];
        grid_lut[lin] = -1;
    }
    (void)gz;
    return m;
}

/* ------------------------------------------------------------------------------------------
 * Rotated BEV overlap / IoU / NMS.  Restates the geometry of
 * utils/detzero_utils/ops/iou3d_nms/src/iou3d_nms_kernel.cu:15-232 (Point, cross,
 * check_rect_cross, check_in_box2d, intersection, rotate_around_center, point_cmp,
 * box_overlap), :328-335 (iou_bev), :386-430 (nms_kernel bitmask) and the host sweep of
 * iou3d_nms.cpp:114-160.  All arithmetic fp32 as in the device code (cos/sin/atan2 are the
 * float overloads there: cosf/sinf/atan2f).
 * ------------------------------------------------------------------------------------------ */
typedef struct { float x, y; } pt2;
static const float ORC_EPS = 1e-8f;

static inline float cross2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }
static inline float cross3(pt2 p1, pt2 p2, pt2 p0)
{
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static inline int rect_cross(pt2 p1, pt2 p2, pt2 q1, pt2 q2)
{
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
static inline int in_box2d(const float *box, pt2 p)
{
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}
static inline int seg_intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 *ans)
{
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > ORC_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}
static inline void rot_center(pt2 c, float ac, float as, pt2 *p)
{
    float nx = (p->x - c.x) * ac + (p->y - c.y) * (-as) + c.x;
    float ny = (p->x - c.x) * as + (p->y - c.y) * ac + c.y;
    p->x = nx; p->y = ny;
}
static inline int pt_cmp(pt2 a, pt2 b, pt2 c)
{
    return atan2f(a.y - c.y, a.x - c.x) > atan2f(b.y - c.y, b.x - c.x);
}

float orc_box_overlap(const float *A, const float *B)
{
    float a_ang = A[6], b_ang = B[6];
    float adx = A[3] / 2, bdx = B[3] / 2, ady = A[4] / 2, bdy = B[4] / 2;
    float ax1 = A[0] - adx, ay1 = A[1] - ady, ax2 = A[0] + adx, ay2 = A[1] + ady;
    float bx1 = B[0] - bdx, by1 = B[1] - bdy, bx2 = B[0] + bdx, by2 = B[1] + bdy;
    pt2 ca = {A[0], A[1]}, cb = {B[0], B[1]};
    pt2 ac[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0, 0}};
    pt2 bc[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0, 0}};
    float acs = cosf(a_ang), asn = sinf(a_ang), bcs = cosf(b_ang), bsn = sinf(b_ang);
    for (int k = 0; k < 4; ++k) { rot_center(ca, acs, asn, &ac[k]); rot_center(cb, bcs, bsn, &bc[k]); }
    ac[4] = ac[0]; bc[4] = bc[0];

    pt2 cp[16]; pt2 pc = {0, 0}; int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            if (seg_intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], &cp[cnt])) {
                pc.x = pc.x + cp[cnt].x; pc.y = pc.y + cp[cnt].y; cnt++;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(A, bc[k])) { pc.x = pc.x + bc[k].x; pc.y = pc.y + bc[k].y; cp[cnt++] = bc[k]; }
        if (in_box2d(B, ac[k])) { pc.x = pc.x + ac[k].x; pc.y = pc.y + ac[k].y; cp[cnt++] = ac[k]; }
    }
    pc.x /= cnt; pc.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (pt_cmp(cp[i], cp[i + 1], pc)) { pt2 t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t; }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        pt2 u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
        pt2 v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

float orc_iou_bev(const float *A, const float *B)
{
    float sa = A[3] * A[4], sb = B[3] * B[4];
    float so = orc_box_overlap(A, B);
    return so / fmaxf(sa + sb - so, ORC_EPS);
}

void orc_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out)
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_box_overlap(a + i * 7, b + j * 7);
}
void orc_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out)
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_iou_bev(a + i * 7, b + j * 7);
}

/* boxes already sorted by score (descending).  keep: (n) int64 indices.  returns #kept.
 * Same result as bitmask construction + sweep of iou3d_nms.cpp:145-156: box i (not yet removed)
 * is kept and removes every j>i with iou_bev(i,j) > thr. */
int orc_nms(const float *boxes, int n, float thr, long long *keep)
{
    unsigned char *removed = (unsigned char *)calloc((size_t)n + 1, 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!removed[j] && orc_iou_bev(boxes + i * 7, boxes + j * 7) > thr) removed[j] = 1;
    }
    free(removed);
    return nk;
}

/* ------------------------------------------------------------------------------------------
 * points_in_boxes_gpu_v2: restates check_pt_in_box3d / points_in_boxes_v2_kernel of
 * utils/detzero_utils/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,352-374.
 * mask: (T, M) int32, 1 where point m lies in box t (caller zero-fills).
 * ------------------------------------------------------------------------------------------ */
void orc_points_in_boxes_margin(const float *boxes, int t, const float *pts, int m, float MARGIN, int *mask)
{
    for (int i = 0; i < m; ++i) {
        float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        for (int k = 0; k < t; ++k) {
            const float *b = boxes + k * 7;
            /* the reference writes `dz / 2.0` and `dx / 2.0 + MARGIN` with double literals, so
             * those comparisons are evaluated in double (roiaware_pool3d_kernel.cu:32-34) */
            if ((double)fabsf(z - b[2]) > (double)b[5] / 2.0) continue;
            float cosa = cosf(-b[6]), sina = sinf(-b[6]);
            float sx = x - b[0], sy = y - b[1];
            float lx = sx * cosa + sy * (-sina);
            float ly = sx * sina + sy * cosa;
            if (((double)fabsf(lx) < (double)b[3] / 2.0 + (double)MARGIN) &
                ((double)fabsf(ly) < (double)b[4] / 2.0 + (double)MARGIN))
                mask[(size_t)k * m + i] = 1;
        }
    }
}

/* the GPU kernel's margin (roiaware_pool3d_kernel.cu:26); the reference's host twin check_pt_in_box3d_cpu
 * (roiaware_pool3d.cpp:255-267) is the same test with MARGIN = 1e-2 - that one can be compiled here and pins the formula
 * (oracle/refbuild.py: points_in_boxes_cpu_reference == orc_points_in_boxes_margin(..., 1e-2f)) */
void orc_points_in_boxes_v2(const float *boxes, int t, const float *pts, int m, int *mask)
{
    orc_points_in_boxes_margin(boxes, t, pts, m, 1e-5f, mask);
}
