// hope_dev.h -- device-side helpers shared by the step kernel and the Reeds-Shepp kernel.
// gfx950 only.  Everything is float64 with contraction OFF (build flag -ffp-contract=off): the
// reference env is float64 numpy/GEOS and its collision / mask / status bits must be reproduced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hope_math.h"

namespace hope {

constexpr int NBEAM = 120;   // configs.py:96
constexpr int NACT = 42;     // configs.py:108-115
constexpr int NITER = 10;    // action_mask.py n_iter
constexpr int UPS = 10;      // action_mask.py up_sample_rate
constexpr int NL = NBEAM * UPS;
constexpr int WAVE = 64;

// configs.py:13-38
constexpr double WHEEL_BASE = 2.8;
constexpr double FRONT_HANG = 0.96;
constexpr double REAR_HANG = 0.93;
constexpr double WIDTH = 1.94;
constexpr double CAR_XF = FRONT_HANG + WHEEL_BASE;   // VehicleBox front x (configs.py:20-24)
constexpr double CAR_XR = -REAR_HANG;
constexpr double CAR_YH = WIDTH / 2;
constexpr double SPEED_LO = -2.5, SPEED_HI = 2.5;
constexpr double STEER_LO = -0.75, STEER_HI = 0.75;
constexpr int NUM_STEP = 10;
constexpr int MINI_ITER = 20;                        // vehicle.py:66
constexpr double STEP_LENGTH = 5e-2;
constexpr double LIDAR_RANGE = 10.0;
constexpr int TOLERANT_TIME = 200;
constexpr double RS_MAX_DIST = 10.0;
constexpr double PI = 3.141592653589793;

// per-scene constant record (float64 words)
enum SceneWord {
    SC_START = 0,     // x, y, heading
    SC_DEST = 3,      // x, y, heading
    SC_BBOX = 6,      // xmin, xmax, ymin, ymax
    SC_DBOX = 10,     // dest box, 4 x (x, y)
    SC_DAREA = 18,    // |dest box|
    SC_DNORM = 19,    // max(|dest - start|, 10)
    SC_DCEN = 20,     // dest box centre x, y and cos / sin of the dest heading
    SC_WORDS = 24
};
// per-scene episode state (float64 words): x, y, heading, accum_arrive_reward
constexpr int ST_WORDS = 4;

__device__ __forceinline__ double clipd(double v, double lo, double hi) {   // np.clip
    double m = v < lo ? lo : v;
    return m > hi ? hi : m;
}

// vehicle corner k of VehicleBox (CCW from rear-right)
__device__ __forceinline__ double car_x(int k) { return (k == 1 || k == 2) ? CAR_XF : CAR_XR; }
__device__ __forceinline__ double car_y(int k) { return (k >= 2) ? CAR_YH : -CAR_YH; }

// ---------------------------------------------------------------------------------------------
// Exact-sign orientation (GEOS Orientation::index semantics: fast filter, then exact expansion).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
    s = a + b;
    double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
__device__ __forceinline__ void two_prod(double a, double b, double& p, double& e) {
    p = a * b;
    e = __builtin_fma(a, b, -p);
}

// Exact sign of ax*by - ax*cy - cx*by - ay*bx + ay*cx + cy*bx (the determinant GEOS's Orientation::index takes the sign of), by an
// expansion sum with zero elimination.  Executed by ONE active lane, practically never (an orientation inside its rounding error
// bound: exactly collinear / touching input); its two work arrays t[12], ex[16] are dynamically indexed, so they live in LDS
// (`w`, 28 doubles) -- as private arrays they would be the kernel's only scratch memory, and out of line (the first version) every
// call site of the four-deep unrolled collision tests carried a call sequence: 100+ of them, 40 % of the kernel's code size.
__device__ __forceinline__ int orient_exact_lds(double ax, double ay, double bx, double by, double cx, double cy, double* w) {
    double* t = w;
    double* ex = w + 12;
    double p_, e_;
    two_prod(ax, by, p_, e_); t[0] = p_; t[1] = e_;
    two_prod(-ax, cy, p_, e_); t[2] = p_; t[3] = e_;
    two_prod(-cx, by, p_, e_); t[4] = p_; t[5] = e_;
    two_prod(-ay, bx, p_, e_); t[6] = p_; t[7] = e_;
    two_prod(ay, cx, p_, e_); t[8] = p_; t[9] = e_;
    two_prod(cy, bx, p_, e_); t[10] = p_; t[11] = e_;
    int m = 0;
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        double q = t[i];
        int mm = 0;
#pragma unroll 1
        for (int j = 0; j < m; j++) {
            double s, e;
            two_sum(q, ex[j], s, e);
            if (e != 0) ex[mm++] = e;
            q = s;
        }
        ex[mm++] = q;
        m = mm;
    }
#pragma unroll 1
    for (int j = m - 1; j >= 0; j--) {
        if (ex[j] > 0) return 1;
        if (ex[j] < 0) return -1;
    }
    return 0;
}

// Orientation::index, fast filter only: the sign of the determinant when its floating-point value is outside the error bound,
// ORIENT_UNDECIDED otherwise (the caller then takes the robust path).
constexpr int ORIENT_UNDECIDED = 2;
__device__ __forceinline__ int orient_filter(double ax, double ay, double bx, double by, double cx, double cy) {
    const double detleft = (ax - cx) * (by - cy);
    const double detright = (ay - cy) * (bx - cx);
    const double det = detleft - detright;
    const int sg = det > 0 ? 1 : (det < 0 ? -1 : 0);
    double detsum;
    if (detleft > 0.0) {
        if (detright <= 0.0) return sg;
        detsum = detleft + detright;
    } else if (detleft < 0.0) {
        if (detright >= 0.0) return sg;
        detsum = -detleft - detright;
    } else return sg;
    const double errbound = 1e-15 * detsum;
    if (det >= errbound || -det >= errbound) return sg;
    return ORIENT_UNDECIDED;
}

// the exact sign: filter, then the expansion (one active lane, LDS work area w[28])
__device__ __forceinline__ int orient_robust_lds(double ax, double ay, double bx, double by, double cx, double cy, double* w) {
    const int f = orient_filter(ax, ay, bx, by, cx, cy);
    return f != ORIENT_UNDECIDED ? f : orient_exact_lds(ax, ay, bx, by, cx, cy, w);
}

// RobustLineIntersector::computeIntersect(p1,p2,q1,q2) != NO_INTERSECTION with the orientation FILTER only:
// 0 = the segments share no point, 1 = they do, 2 = undecided (an orientation the filter could not sign and the others leave open).
__device__ __forceinline__ int segments_intersect_fast(double p1x, double p1y, double p2x, double p2y, double q1x,
                                                       double q1y, double q2x, double q2y) {
    double minq = fmin(q1x, q2x), maxq = fmax(q1x, q2x), minp = fmin(p1x, p2x), maxp = fmax(p1x, p2x);
    if (minp > maxq || maxp < minq) return 0;
    minq = fmin(q1y, q2y); maxq = fmax(q1y, q2y); minp = fmin(p1y, p2y); maxp = fmax(p1y, p2y);
    if (minp > maxq || maxp < minq) return 0;
    const int Pq1 = orient_filter(p1x, p1y, p2x, p2y, q1x, q1y);
    const int Pq2 = orient_filter(p1x, p1y, p2x, p2y, q2x, q2y);
    if ((Pq1 == 1 && Pq2 == 1) || (Pq1 == -1 && Pq2 == -1)) return 0;
    const int Qp1 = orient_filter(q1x, q1y, q2x, q2y, p1x, p1y);
    const int Qp2 = orient_filter(q1x, q1y, q2x, q2y, p2x, p2y);
    if ((Qp1 == 1 && Qp2 == 1) || (Qp1 == -1 && Qp2 == -1)) return 0;
    if (Pq1 == ORIENT_UNDECIDED || Pq2 == ORIENT_UNDECIDED || Qp1 == ORIENT_UNDECIDED || Qp2 == ORIENT_UNDECIDED) return 2;
    return 1;
}

// hull corners for a pose; matrix [cos,-sin,sin,cos,x,y] applied to VehicleBox (vehicle.py:32-36)
struct Box {
    double x[4], y[4];
};
__device__ __forceinline__ Box make_box(double px, double py, double ct, double st) {
    Box b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        b.x[k] = ct * car_x(k) + (-st) * car_y(k) + px;
        b.y[k] = st * car_x(k) + ct * car_y(k) + py;
    }
    return b;
}

// The robust form of "any hull edge of the pose shares a point with the obstacle edge (x1,y1)-(x2,y2)" (_detect_collision,
// car_parking_base.py:153-158): exact orientation signs.  ONE active lane; xl = 40 doubles of LDS (12 coordinates + the expansion's
// work area).  Rolled loops on purpose: one copy of the expansion code per call site.
constexpr int ROBUST_LDS_WORDS = 40;
__device__ __forceinline__ bool hull_edge_intersect_robust(double px, double py, double ct, double st, double x1, double y1,
                                                           double x2, double y2, double* xl) {
    {
        const Box b = make_box(px, py, ct, st);
#pragma unroll
        for (int k = 0; k < 4; k++) { xl[2 * k] = b.x[k]; xl[2 * k + 1] = b.y[k]; }
        xl[8] = x1; xl[9] = y1; xl[10] = x2; xl[11] = y2;
    }
    bool hit = false;
#pragma unroll 1
    for (int k = 0; k < 4 && !hit; k++) {
        const int k2 = (k + 1) & 3;
        const double p1x = xl[2 * k], p1y = xl[2 * k + 1], p2x = xl[2 * k2], p2y = xl[2 * k2 + 1];
        double minq = fmin(x1, x2), maxq = fmax(x1, x2), minp = fmin(p1x, p2x), maxp = fmax(p1x, p2x);
        if (minp > maxq || maxp < minq) continue;
        minq = fmin(y1, y2); maxq = fmax(y1, y2); minp = fmin(p1y, p2y); maxp = fmax(p1y, p2y);
        if (minp > maxq || maxp < minq) continue;
        // the four orientations of computeIntersect: (p1,p2,q1) (p1,p2,q2) (q1,q2,p1) (q1,q2,p2); points p1 p2 q1 q2 = 0..3
        int sgn = 0;                                       // two bits per orientation: sign + 1
#pragma unroll 1
        for (int j = 0; j < 4; j++) {
            const int ia = j < 2 ? 0 : 2, ic = j < 2 ? 2 + j : j - 2;
            double P[3][2];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int idx = q == 0 ? ia : (q == 1 ? ia + 1 : ic);             // point index 0..3
                const int w = idx == 0 ? 2 * k : (idx == 1 ? 2 * k2 : (idx == 2 ? 8 : 10));
                P[q][0] = xl[w]; P[q][1] = xl[w + 1];
            }
            sgn |= (orient_robust_lds(P[0][0], P[0][1], P[1][0], P[1][1], P[2][0], P[2][1], xl + 12) + 1) << (2 * j);
        }
        const int Pq1 = (sgn & 3) - 1, Pq2 = ((sgn >> 2) & 3) - 1, Qp1 = ((sgn >> 4) & 3) - 1, Qp2 = ((sgn >> 6) & 3) - 1;
        if ((Pq1 > 0 && Pq2 > 0) || (Pq1 < 0 && Pq2 < 0)) continue;
        if ((Qp1 > 0 && Qp2 > 0) || (Qp1 < 0 && Qp2 < 0)) continue;
        hit = true;
    }
    return hit;
}

// XCD-aware block -> scene map.  Workgroup b runs on XCD b % 8 (8 XCDs, each with its own L2 and 32 CUs).  With
// scene = b any per-scene cost pattern whose period divides 8 (e.g. scene classes interleaved with period 4, as a
// round-robin level mix produces) would pile the expensive scenes onto 2 of the 8 XCDs (measured: 2.6x slower).
// The map keeps the FOUR list entries 4 L .. 4 L + 3 (one 128-byte line of `state`, half a line of `post`, when the list is
// contiguous) on ONE XCD, so that such a line is fetched into one L2 instead of up to four (k_env_step fetch -9 %, step time
// unchanged), and spreads the lines: inside an aligned group of 256 blocks, block b = 256 G + 32 h + 8 r + x (x = its XCD) takes
// entry 256 G + 4 (8 h + (x ^ h)) + r.  A bijection of the group; every XCD gets every residue of the line index mod 8 once per
// group and every entry residue mod 4 equally often, so cost patterns of period 3, 4, 8, 16, 32 and sorted lists stay spread.
// The last partial group falls back to the round-2 map (XOR of bits 3..5 into bits 0..2 inside aligned groups of 64), the
// tail (n not a multiple of 64) is left as is.  HOPE_XCD_XOR: the round-2 map everywhere.
__device__ __forceinline__ int scene_of_block(int b, int n) {
#ifndef HOPE_XCD_XOR
    if ((b | 255) < n) {
        const int x = b & 7, r = (b >> 3) & 3, h = (b >> 5) & 7;
        return (b & ~255) + 4 * (8 * h + (x ^ h)) + r;
    }
#endif
    if ((b | 63) >= n) return b;
    return b ^ ((b >> 3) & 7);
}

// lane l's value of a double (l must be wave-uniform): two v_readlane, no LDS round trip
__device__ __forceinline__ double readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {        // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// obstacle box (xmin, xmax, ymin, ymax) in float32, rounded outwards: every cull against it keeps a superset
__device__ __forceinline__ float4 obstacle_box(const double* v) {
    return make_float4(__double2float_rd(fmin(fmin(v[0], v[2]), fmin(v[4], v[6]))),
                       __double2float_ru(fmax(fmax(v[0], v[2]), fmax(v[4], v[6]))),
                       __double2float_rd(fmin(fmin(v[1], v[3]), fmin(v[5], v[7]))),
                       __double2float_ru(fmax(fmax(v[1], v[3]), fmax(v[5], v[7]))));
}

// The float32 view of one obstacle that the Reeds-Shepp validation kernel's filter works on (k_rs_validate_f), made ONCE per
// map -- wherever a scene's tile is written -- instead of once per search (it was a quarter of that kernel's cycles): the four
// vertices relative to the scene's frame origin (ox, oy) = (map box xmin, ymin) (integers: floor / ceil of parking_map_*.py's
// map box; the subtraction is done in float64, one rounding to float32, <= 8e-6 m for |coordinates| <= 128 m), their box, and one
// flag per edge j (vertex j -> j + 1): the edge is ROBUST for the reference's tolerance-free box test (car_parking_base.py:518-526)
// -- both extents of its coordinate box >= FETA_EDGE, or it lies exactly on the world line y = 0 / x = 0 (then the candidate
// coordinate is an exact zero whatever the rounding; the back wall of every generated lot, parking_map_normal.py:70-78).
constexpr float FETA_EDGE = 1e-2f;
constexpr int OBST_F_CONVEX = 0x10, OBST_F_CCW = 0x20;   // per-obstacle shape flags next to the four edge flags (bits 0..3)
__device__ __forceinline__ void obstacle_f32(const double* v /*[8]*/, double ox, double oy, float4* fv /*[2]*/, float4* fbox, uint8_t* eflag) {
    float fx[4], fy[4];
    int fl = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        fx[j] = (float)(v[2 * j] - ox); fy[j] = (float)(v[2 * j + 1] - oy);
        const int j2 = (j + 1) & 3;
        const double x1 = v[2 * j], y1 = v[2 * j + 1], x2 = v[2 * j2], y2 = v[2 * j2 + 1];
        const bool wide_x = fabs(x2 - x1) >= (double)FETA_EDGE, wide_y = fabs(y2 - y1) >= (double)FETA_EDGE;
        const bool zero_line = (y1 == 0.0 && y2 == 0.0 && wide_x) || (x1 == 0.0 && x2 == 0.0 && wide_y);
        fl |= ((wide_x && wide_y) || zero_line) ? (1 << j) : 0;
    }
    {   // shape flags for the lidar's back-face cull (k_env_step): bit 4 = a strictly convex quadrilateral without slivers (every
        // interior angle between 25 and 155 degrees, every edge >= 5 cm), bit 5 = counter-clockwise
        double cr[4];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int j1 = (j + 1) & 3, j2 = (j + 2) & 3;
            const double ax = v[2 * j1] - v[2 * j], ay = v[2 * j1 + 1] - v[2 * j + 1];
            const double bx = v[2 * j2] - v[2 * j1], by = v[2 * j2 + 1] - v[2 * j1 + 1];
            cr[j] = ax * by - ay * bx;                                    // turn at vertex j + 1
            const double la = ax * ax + ay * ay, lb = bx * bx + by * by;
            ok = ok && la >= 0.0025 && lb >= 0.0025 && cr[j] * cr[j] >= 0.1786 * la * lb;      // sin^2(25 deg) = 0.1786
        }
        const bool ccw = cr[0] > 0;
        ok = ok && ((cr[0] > 0) == (cr[1] > 0)) && ((cr[1] > 0) == (cr[2] > 0)) && ((cr[2] > 0) == (cr[3] > 0));
        fl |= ok ? OBST_F_CONVEX : 0;
        fl |= ccw ? OBST_F_CCW : 0;
    }
    fv[0] = make_float4(fx[0], fy[0], fx[1], fy[1]);
    fv[1] = make_float4(fx[2], fy[2], fx[3], fy[3]);
    *fbox = make_float4(fminf(fminf(fx[0], fx[1]), fminf(fx[2], fx[3])), fmaxf(fmaxf(fx[0], fx[1]), fmaxf(fx[2], fx[3])),
                        fminf(fminf(fy[0], fy[1]), fminf(fy[2], fy[3])), fmaxf(fmaxf(fy[0], fy[1]), fmaxf(fy[2], fy[3])));
    *eflag = (uint8_t)fl;
}
// row stride of the per-obstacle flag bytes
__host__ __device__ inline int eflag_stride(int max_obst) { return (max_obst + 3) & ~3; }

// wave-uniform LDS synchronisation for a one-wave workgroup
__device__ __forceinline__ void wsync() { __syncthreads(); }

}  // namespace hope
