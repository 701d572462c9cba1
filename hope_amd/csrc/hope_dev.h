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

// exact sign of ax*by - ax*cy - cx*by - ay*bx + ay*cx + cy*bx; rarely executed -> keep out of line
__device__ __noinline__ int orient_exact(double ax, double ay, double bx, double by, double cx, double cy) {
    double t[12];
    two_prod(ax, by, t[0], t[1]);
    two_prod(-ax, cy, t[2], t[3]);
    two_prod(-cx, by, t[4], t[5]);
    two_prod(-ay, bx, t[6], t[7]);
    two_prod(ay, cx, t[8], t[9]);
    two_prod(cy, bx, t[10], t[11]);
    double ex[16];
    int m = 0;
    for (int i = 0; i < 12; i++) {
        double q = t[i];
        int mm = 0;
        for (int j = 0; j < m; j++) {
            double s, e;
            two_sum(q, ex[j], s, e);
            if (e != 0) ex[mm++] = e;
            q = s;
        }
        ex[mm++] = q;
        m = mm;
    }
    for (int j = m - 1; j >= 0; j--) {
        if (ex[j] > 0) return 1;
        if (ex[j] < 0) return -1;
    }
    return 0;
}

__device__ __forceinline__ int orient(double ax, double ay, double bx, double by, double cx, double cy) {
    double detleft = (ax - cx) * (by - cy);
    double detright = (ay - cy) * (bx - cx);
    double det = detleft - detright;
    double detsum = 0;
    bool ok = false;
    if (detleft > 0.0) {
        if (detright <= 0.0) ok = true; else detsum = detleft + detright;
    } else if (detleft < 0.0) {
        if (detright >= 0.0) ok = true; else detsum = -detleft - detright;
    } else ok = true;
    if (!ok) {
        double errbound = 1e-15 * detsum;
        if (det >= errbound || -det >= errbound) ok = true;
    }
    if (ok) return det > 0 ? 1 : (det < 0 ? -1 : 0);
    return orient_exact(ax, ay, bx, by, cx, cy);
}

// RobustLineIntersector::computeIntersect(p1,p2,q1,q2) != NO_INTERSECTION
__device__ __forceinline__ bool segments_intersect(double p1x, double p1y, double p2x, double p2y, double q1x,
                                                   double q1y, double q2x, double q2y) {
    double minq = fmin(q1x, q2x), maxq = fmax(q1x, q2x), minp = fmin(p1x, p2x), maxp = fmax(p1x, p2x);
    if (minp > maxq || maxp < minq) return false;
    minq = fmin(q1y, q2y); maxq = fmax(q1y, q2y); minp = fmin(p1y, p2y); maxp = fmax(p1y, p2y);
    if (minp > maxq || maxp < minq) return false;
    int Pq1 = orient(p1x, p1y, p2x, p2y, q1x, q1y);
    int Pq2 = orient(p1x, p1y, p2x, p2y, q2x, q2y);
    if ((Pq1 > 0 && Pq2 > 0) || (Pq1 < 0 && Pq2 < 0)) return false;
    int Qp1 = orient(q1x, q1y, q2x, q2y, p1x, p1y);
    int Qp2 = orient(q1x, q1y, q2x, q2y, p2x, p2y);
    if ((Qp1 > 0 && Qp2 > 0) || (Qp1 < 0 && Qp2 < 0)) return false;
    return true;
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

// XCD-aware block -> scene map.  Workgroup b runs on XCD b % 8 (8 XCDs, each with its own L2 and 32 CUs).  With
// scene = b any per-scene cost pattern whose period divides 8 (e.g. scene classes interleaved with period 4, as a
// round-robin level mix produces) would pile the expensive scenes onto 2 of the 8 XCDs (measured: 2.6x slower).
// XOR-ing bits 3..5 into bits 0..2 is a bijection inside every aligned group of 64 blocks and gives each XCD every
// residue mod 8 equally often; sorted batches stay spread too.  The tail group (n not a multiple of 64) is left as is.
__device__ __forceinline__ int scene_of_block(int b, int n) {
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

// wave-uniform LDS synchronisation for a one-wave workgroup
__device__ __forceinline__ void wsync() { __syncthreads(); }

}  // namespace hope
