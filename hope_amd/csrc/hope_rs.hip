// hope_rs.hip -- Reeds-Shepp feasibility search: ONE WAVEFRONT PER ELIGIBLE SCENE.
//
// Replaces CarParking.find_rs_path (src/env/car_parking_base.py:413-450) with
//   rsCurve.calc_all_paths / generate_path / set_path / generate_local_course / interpolate
//     (src/env/reeds_shepp.py:35-557) and CarParking.is_traj_valid (car_parking_base.py:452-534),
// for the scenes the step kernel queued (gate :293-294: t > 1, CONTINUE, |pos - dest| < 10).
//
// Two kernels per obstacle-tile class (details at each kernel):
//   k_rs_words     FOUR LANES PER SCENE (16 searches per wave): the 12 solver families of generate_path run one after
//                  the other, lane q of a quad evaluating reflection q, so the solver bodies never diverge; set_path's
//                  order-dependent de-dup (:57-76) only compares words of equal type sequence, which are known
//                  statically (quad broadcasts); lane 0 of the quad replays heapdict's array heap (push order = path
//                  order, non-strict sift-up, strict sift-down) for the pop order find_rs_path sees, ties included,
//                  and stops where the search's stop rule (:443) would; output = one record per search (RS_REC_*).
//   k_rs_validate  ONE WAVE PER SEARCH: the record arrives in one coalesced load; for each popped word the samples of
//                  generate_local_course are produced 64 at a time (`pd += d` chain in lock-step, sequential rounding
//                  kept), each lane interpolates ITS sample, builds the 4 hull edges and tests them against the
//                  obstacle tile in LDS.  A (hull edge, obstacle edge) pair can only be a hit if the two edge bounding
//                  boxes overlap (the reference requires the intersection point inside both), so pairs failing that
//                  exact pre-test skip the two float64 divisions.  The reference's obstacle cull (:482-494) and its
//                  T x 4 x E matrix are result-neutral and are not materialised.
// Deviation (documented in DESIGN.md): generate_local_course's "pop trailing samples whose local x is
// exactly 0.0" (:501-505) is not replayed beyond the unused array tail (a measure-zero event).
#include <stdlib.h>

#include "hope_dev.h"
#include "hope_internal.h"

namespace hope {

namespace {

constexpr double MAXC = RS_MAXC;                 // math.hm_tan(VALID_STEER[-1]) / WHEEL_BASE  (car_parking_base.py:422)
constexpr double RS_STEP = 0.1;                  // step_size passed by find_rs_path (:424)
constexpr double MAX_LENGTH = 1000.0;            // reeds_shepp.py:6
constexpr int NCAND = 46;

enum { TS = 0, TL = 1, TR = 2 };

__device__ __forceinline__ double py_mod(double v, double w) {   // Python float %
    double m = hm_fmod(v, w);
    if (m != 0) { if ((w < 0) != (m < 0)) m += w; } else m = copysign(0.0, w);
    return m;
}
__device__ __forceinline__ double rs_M(double theta) {           // reeds_shepp.py:581-592
    double phi = py_mod(theta, 2.0 * PI);
    if (phi < -PI) phi += 2.0 * PI;
    if (phi > PI) phi -= 2.0 * PI;
    return phi;
}
__device__ __forceinline__ void rs_R(double x, double y, double& r, double& th) { r = hm_hypot(x, y); th = hm_atan2(y, x); }
__device__ __forceinline__ double pi_2_pi(double t) {            // :561-568
    while (t > PI) t -= 2.0 * PI;
    while (t < -PI) t += 2.0 * PI;
    return t;
}

__device__ bool rs_SLS(double x, double y, double phi, double& t, double& u, double& v) {   // :133-149
    phi = rs_M(phi);
    if (y > 0.0 && 0.0 < phi && phi < PI * 0.99) {
        double xd = -y / hm_tan(phi) + x;
        t = xd - hm_tan(phi / 2.0);
        u = phi;
        v = sqrt((x - xd) * (x - xd) + y * y) - hm_tan(phi / 2.0);
        return true;
    } else if (y < 0.0 && 0.0 < phi && phi < PI * 0.99) {
        double xd = -y / hm_tan(phi) + x;
        t = xd - hm_tan(phi / 2.0);
        u = phi;
        v = -sqrt((x - xd) * (x - xd) + y * y) - hm_tan(phi / 2.0);
        return true;
    }
    return false;
}
__device__ __forceinline__ bool rs_LSL(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) {   // :79-87
    double uu, tt;
    rs_R(x - sphi, y - 1.0 + cphi, uu, tt);
    if (tt >= 0.0) {
        double vv = rs_M(phi - tt);
        if (vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ __forceinline__ bool rs_LSR(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) {   // :90-103
    double u1, t1;
    rs_R(x + sphi, y - 1.0 - cphi, u1, t1);
    u1 = u1 * u1;
    if (u1 >= 4.0) {
        double uu = sqrt(u1 - 4.0);
        double theta = hm_atan2(2.0, uu);
        double tt = rs_M(t1 + theta);
        double vv = rs_M(tt - phi);
        if (tt >= 0.0 && vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ __forceinline__ bool rs_LRL(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) {   // :106-117
    double u1, t1;
    rs_R(x - sphi, y - 1.0 + cphi, u1, t1);
    if (u1 <= 4.0) {
        double uu = -2.0 * hm_asin(0.25 * u1);
        double tt = rs_M(t1 + 0.5 * uu + PI);
        double vv = rs_M(phi - tt + uu);
        if (tt >= 0.0 && uu <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ __forceinline__ void calc_tauOmega(double u, double v, double xi, double eta, double phi, double& tau, double& omega) {
    double delta = rs_M(u - v);                                                             // :228-243
    double su, cu, sd, cd;
    hm_sincos(u, &su, &cu);
    hm_sincos(delta, &sd, &cd);
    double A = su - sd;
    double B = cu - cd - 1.0;
    double t1 = hm_atan2(eta * A - xi * B, xi * A + eta * B);
    double t2 = 2.0 * (cd - cu - cu) + 3.0;          // cos(v) = cos(u): v = +-u and hm_cos is exactly even
    if (t2 < 0) tau = rs_M(t1 + PI); else tau = rs_M(t1);
    omega = rs_M(tau - u + v - phi);
}
__device__ __forceinline__ bool rs_LRLRn(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) { // :246-257
    double xi = x + sphi, eta = y - 1.0 - cphi;
    double rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (rho <= 1.0) {
        double uu = hm_acos(rho), tt, vv;
        calc_tauOmega(uu, -uu, xi, eta, phi, tt, vv);
        if (tt >= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ __forceinline__ bool rs_LRLRp(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) { // :260-272
    double xi = x + sphi, eta = y - 1.0 - cphi;
    double rho = (20.0 - xi * xi - eta * eta) / 16.0;
    if (0.0 <= rho && rho <= 1.0) {
        double uu = -hm_acos(rho);
        if (uu >= -0.5 * PI) {
            double tt, vv;
            calc_tauOmega(uu, uu, xi, eta, phi, tt, vv);
            if (tt >= 0.0 && vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
        }
    }
    return false;
}
__device__ __forceinline__ bool rs_LRSR(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) {  // :311-323
    double xi = x + sphi, eta = y - 1.0 - cphi, rho, theta;
    rs_R(-eta, xi, rho, theta);
    if (rho >= 2.0) {
        double tt = theta, uu = 2.0 - rho, vv = rs_M(tt + 0.5 * PI - phi);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ __forceinline__ bool rs_LRSL(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) {  // :326-339
    double xi = x - sphi, eta = y - 1.0 + cphi, rho, theta;
    rs_R(xi, eta, rho, theta);
    if (rho >= 2.0) {
        double r = sqrt(rho * rho - 4.0);
        double uu = 2.0 - r;
        double tt = rs_M(theta + hm_atan2(r, -2.0));
        double vv = rs_M(phi - 0.5 * PI - tt);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ __forceinline__ bool rs_LRSLR(double x, double y, double phi, double sphi, double cphi, double& t, double& u, double& v) { // :414-429
    double xi = x + sphi, eta = y - 1.0 - cphi, rho, theta;
    rs_R(xi, eta, rho, theta);
    if (rho >= 2.0) {
        double uu = 4.0 - sqrt(rho * rho - 4.0);
        if (uu <= 0.0) {
            double tt = rs_M(hm_atan2((4.0 - uu) * xi - 2.0 * eta, -2.0 * xi + (uu - 4.0) * eta));
            double vv = rs_M(tt - phi);
            if (tt >= 0.0 && vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
        }
    }
    return false;
}

// candidate c (0..45) of generate_path in call order: group g, variant q
//   q=0: f(x, y, phi) ; q=1: f(-x, y, -phi), lengths negated ; q=2: f(x, -y, -phi), L<->R ; q=3: both
__device__ __forceinline__ void cand_decode(int c, int& g, int& q) {
    if (c < 2) { g = 0; q = c * 2; }              // SCS (:120-130): SLS(x,y,phi), SLS(x,-y,-phi)
    else { g = 1 + (c - 2) / 4; q = (c - 2) & 3; }
}
// groups: 0 SLS | 1 LSL | 2 LSR | 3 LRL | 4 LRL backwards | 5 LRLRn | 6 LRLRp | 7 LRSL | 8 LRSR |
//         9 LRSL backwards | 10 LRSR backwards | 11 LRSLR
__device__ __forceinline__ int pack_types(int a, int b, int c, int d, int e, int n) {
    return a | (b << 2) | (c << 4) | (d << 6) | (e << 8) | (n << 12);
}
__device__ __forceinline__ int mirror_type(int t) { return t == TL ? TR : (t == TR ? TL : t); }
__device__ __forceinline__ int type_of(int code, int i) { return (code >> (2 * i)) & 3; }

// interpolate (:510-537) of arc/line parameter l from origin (ox, oy, oyaw) in the local frame
__device__ __forceinline__ void interpolate(double l, int m, double ox, double oy, double oyaw, double c_noy,
                                            double s_noy, double c_oy, double s_oy, double& px, double& py,
                                            double& pyaw) {
    if (m == TS) {
        px = ox + l / MAXC * c_oy;
        py = oy + l / MAXC * s_oy;
        pyaw = oyaw;
    } else {
        double sl, cl;
        hm_sincos(l, &sl, &cl);
        double ldx = sl / MAXC;
        double ldy = (m == TL) ? (1.0 - cl) / MAXC : (1.0 - cl) / (-MAXC);
        double gdx = c_noy * ldx + s_noy * ldy;          // hm_cos(-oyaw)*ldx + hm_sin(-oyaw)*ldy
        double gdy = -s_noy * ldx + c_noy * ldy;
        px = ox + gdx;
        py = oy + gdy;
        pyaw = (m == TL) ? oyaw + l : oyaw - l;
    }
}

// wave-wide float min / max without LDS round trips: DPP butterflies inside each row of 16 lanes, then the four row
// results through readlane
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_min_f(float v) {
    v = fminf(v, dpp_f<0xB1>(v));        // quad_perm [1,0,3,2]
    v = fminf(v, dpp_f<0x4E>(v));        // quad_perm [2,3,0,1]
    v = fminf(v, dpp_f<0x141>(v));       // row_half_mirror
    v = fminf(v, dpp_f<0x140>(v));       // row_mirror
    const int i = __float_as_int(v);
    return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fminf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    const int i = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}

__device__ __forceinline__ double wave_min_d(double v) {
    v = fmin(v, __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true)));
    v = fmin(v, __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4E, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4E, 0xf, 0xf, true)));
    v = fmin(v, __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x141, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), 0x141, 0xf, 0xf, true)));
    v = fmin(v, __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x140, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), 0x140, 0xf, 0xf, true)));
    return fmin(fmin(readlane_d(v, 0), readlane_d(v, 16)), fmin(readlane_d(v, 32), readlane_d(v, 48)));
}

// Cycle accounting of the validation kernel (instantiated only when HOPE_RS_TIMING is set; s_memtime per section):
// [0] prologue [1] word setup [2] sample generator [3] interpolate + transform [4] pose_hits: hull, union box, obstacle cull
// [5] pose_hits: candidate loop [6] whole wave [7] screen pass (round 5) [8] waves [9] words tested [10] generator rounds [11] passes [12] passes with
// candidates [13] candidate obstacles visited
__device__ unsigned long long g_rs_prof[64 * 16];       // 64 shards (block index mod 64), summed by the host
// per-search log of the instrumented build (tools/rs_tail.py): cycles, words tested, samples tested, found
constexpr int RS_LOG_CAP = 1 << 21;
__device__ int g_rs_log_n;
__device__ int4 g_rs_log[RS_LOG_CAP];
#define RS_T0() unsigned long long t0_ = TIMING ? __builtin_readcyclecounter() : 0
#define RS_T(i) do { if (TIMING) { const unsigned long long t1_ = __builtin_readcyclecounter(); tsec[i] += t1_ - t0_; t0_ = t1_; } } while (0)

// is_traj_valid for the (up to 64) poses held one per lane: returns true on a lane whose pose is out of the
// map box or whose hull meets an obstacle edge (line-line intersection inside both edge boxes, no tolerance).
// Obstacles are culled per call: the union box of the active lanes' hulls is wave-reduced, one lane per
// obstacle compares its precomputed box (obb, in LDS) with it, and only the survivors (cand[], usually 0-3)
// are visited.  A pair whose boxes do not overlap cannot pass the reference's box tests, so this is exact.
template <bool TIMING, bool FULL = false>
__device__ __forceinline__ bool pose_hits(bool active, double wx, double wy, double wyaw, const double* tile,
                                          const float4* obb, int* cand, int n_obst, double xmin, double xmax,
                                          double ymin, double ymax, int lane, unsigned long long* tsec) {
    RS_T0();
    bool bad = false;
    double vx[4], vy[4];
    double hminx = INFINITY, hmaxx = -INFINITY, hminy = INFINITY, hmaxy = -INFINITY;
    if (active) {
        if (wx < xmin || wx > xmax || wy < ymin || wy > ymax) bad = true;      // :462-464
        double st, ct;
        hm_sincos(wyaw, &st, &ct);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            vx[k] = ct * car_x(k) - st * car_y(k) + wx;                         // :468-471
            vy[k] = st * car_x(k) + ct * car_y(k) + wy;
        }
        hminx = fmin(fmin(vx[0], vx[1]), fmin(vx[2], vx[3]));
        hmaxx = fmax(fmax(vx[0], vx[1]), fmax(vx[2], vx[3]));
        hminy = fmin(fmin(vy[0], vy[1]), fmin(vy[2], vy[3]));
        hmaxy = fmax(fmax(vy[0], vy[1]), fmax(vy[2], vy[3]));
    }
    if (!FULL && __any(bad)) return bad;
    // union box of the pass: float, rounded outwards (only used to cull whole obstacles, so a superset is exact)
    float ulox = active ? __double2float_rd(hminx) : INFINITY, uhix = active ? __double2float_ru(hmaxx) : -INFINITY;
    float uloy = active ? __double2float_rd(hminy) : INFINITY, uhiy = active ? __double2float_ru(hmaxy) : -INFINITY;
    ulox = wave_min_f(ulox); uhix = wave_max_f(uhix); uloy = wave_min_f(uloy); uhiy = wave_max_f(uhiy);
    int nc = 0;
    for (int base = 0; base < n_obst; base += WAVE) {
        int o = base + lane;
        bool near = false;
        if (o < n_obst) {
            const float4 bb = obb[o];                         // (xmin, xmax, ymin, ymax) rounded outwards: a superset
            near = !(bb.x > uhix || bb.y < ulox || bb.z > uhiy || bb.w < uloy);
        }
        unsigned long long m = __ballot(near);
        if (near) cand[nc + __popcll(m & ((1ull << lane) - 1))] = o;
        nc += __popcll(m);
    }
    RS_T(4);
    if (nc == 0) return bad;                              // (bad: only with FULL, which does not leave early)
    if (TIMING) { tsec[12] += 1; tsec[13] += nc; }
    wsync();
    for (int ci = 0; ci < nc; ci++) {
        const int r = cand[ci];
        const double* o = tile + 8 * r;
        const float4 bb = obb[r];
        bool near = active && !((double)bb.x > hmaxx || (double)bb.y < hminx || (double)bb.z > hmaxy || (double)bb.w < hminy);
        if (!__any(near)) continue;
        if (near) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                double x1 = o[2 * j], y1 = o[2 * j + 1], x2 = o[2 * ((j + 1) & 3)], y2 = o[2 * ((j + 1) & 3) + 1];
                double exmin = fmin(x1, x2), exmax = fmax(x1, x2), eymin = fmin(y1, y2), eymax = fmax(y1, y2);
                if (exmin > hmaxx || exmax < hminx || eymin > hmaxy || eymax < hminy) continue;
                double d = y2 - y1, e = x1 - x2, f = y1 * x2 - x1 * y2;                           // :504-506
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int k2 = (k + 1) & 3;
                    double ax1 = vx[k], ay1 = vy[k], ax2 = vx[k2], ay2 = vy[k2];
                    double vminx = fmin(ax1, ax2), vmaxx = fmax(ax1, ax2), vminy = fmin(ay1, ay2), vmaxy = fmax(ay1, ay2);
                    // exact necessary condition for the 8 box tests of :518-526
                    if (vminx > exmax || vmaxx < exmin || vminy > eymax || vmaxy < eymin) continue;
                    double a = ay2 - ay1, b = ax1 - ax2, c = ay1 * ax2 - ax1 * ay2;               // :477-479
                    double det = a * e - b * d;
                    if (det == 0) continue;
                    double raw_x = (b * f - c * e) / det;
                    double raw_y = (c * d - a * f) / det;
                    bool cx = !(raw_x > exmax) && !(raw_x < exmin) && !(raw_x > vmaxx) && !(raw_x < vminx);
                    bool cy = !(raw_y > eymax) && !(raw_y < eymin) && !(raw_y > vmaxy) && !(raw_y < vminy);
                    if (cx && cy) bad = true;
                }
                if (!FULL && __any(bad)) break;               // (the lanes that are near this obstacle)
            }
        }
        // one colliding sample condemns the path: the remaining candidates need not be looked at (the first-segment cache
        // may then learn a farther colliding sample than the nearest one -- still a true collision, only less pruning)
        if (!FULL && __any(bad)) break;
    }
    wsync();
    RS_T(5);
    return bad;
}

// ================================================================================================
// Kernel A: generate_path + set_path + heapdict order  ->  word list + pop order per queued scene
// ================================================================================================
// FOUR LANES PER SCENE (16 scenes per wave).  generate_path tries 12 solver families, each in the four
// reflections q = 0..3 of (:152-185 etc.; SCS only q = 0, 2): lane q of a quad evaluates reflection q, and all lanes
// of the wave run the SAME family at the same time -- no divergence between solver bodies, which a
// lane-per-candidate layout pays for twelve times over.  set_path's order-dependent de-dup (:57-76) only ever
// compares words of equal type sequence, and those are known statically: (g, q) with (g, q ^ 1), LRL with the
// "backwards" LRL of the next family, LRLRn with LRLRp -- a handful of quad broadcasts per family instead of a
// replay over all candidates.  heapdict's array heap (push order = path order, non-strict sift-up, strict sift-down)
// is replayed per scene by lane 0 of the quad as the words are kept; the pop order goes to rs_order.
constexpr int RSA_SCENES = WAVE / 4;

template <int J>
__device__ __forceinline__ double quad_get(double v) {
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), J * 0x55, 0xf, 0xf, true);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), J * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ int quad_get(int v) { return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xf, 0xf, true); }

// sum([x - y for x, y in zip(old.lengths, new.lengths)]) <= 0.01 (:63-66), left to right over n segments
__device__ __forceinline__ bool rs_dup(int n, double o0, double o1, double o2, double o3, double o4, double c0, double c1,
                                       double c2, double c3, double c4) {
    double s = 0;
    s = s + (o0 - c0); s = s + (o1 - c1); s = s + (o2 - c2);
    if (n > 3) s = s + (o3 - c3);
    if (n > 4) s = s + (o4 - c4);
    return s <= 0.01;
}

// blockIdx.y = family group: the twelve families are independent except for set_path's two cross-family de-dups
// (g = 4 against 3, g = 6 against 5), so four waves share a group of 16 searches -- a third of the serial length each and
// four times the waves (one wave per 16 searches left the GPU with 2.5 long waves per SIMD: 66 us per launch, hardly
// overlapped with anything).  The heap replay (push order = slot order) runs afterwards in k_rs_segs.
constexpr int RSA_GROUPS = 4;
__device__ __constant__ const int rsa_group_begin[RSA_GROUPS + 1] = {0, 3, 7, 9, 12};

__device__ __forceinline__ void set_wave_prio(int pr) {        // (s_setprio takes an immediate)
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr >= 3) __builtin_amdgcn_s_setprio(3);
}

__global__ __launch_bounds__(64) void k_rs_words(RsParams p) {
    set_wave_prio(p.prio_front);
    const int lane = threadIdx.x, q = lane & 3, ls = lane >> 2;
    const int qi = blockIdx.x * RSA_SCENES + ls;
    // the search's inputs are requested BEFORE the queue length is known (one memory round trip less): k_rs_compact wrote them by
    // queue position, rows at or beyond the count are stale or zero (hope_env_create clears them) and lanes without work store nothing
    const double* in = p.rs_in + (size_t)(qi < p.max_queue ? qi : p.max_queue - 1) * RS_IN_WORDS;
    // generate_path's goal in the start frame (:540-557), normalised by k_rs_compact (one lane per search; RS_IN_WORDS)
    const double X = in[7], Y = in[8], PHI = in[9], sPHI = in[10], cPHI = in[11];
    const double XB = in[12], YB = in[13];                    // "backwards" (:206-207, :376-377)
    const int count = *p.rs_count;
    if ((int)blockIdx.x * RSA_SCENES >= count) return;
    const bool live = qi < count;
    const int slot = p.slot_base + p.slot_dir * (live ? qi : 0);

    double* rec = p.rs_rec + (size_t)slot * RS_REC_DOUBLES;
    RsWord* words = (RsWord*)(rec + RS_REC_WORDS);            // stored by candidate slot 4 g + q (= path order)
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0;                    // previous family: my reflection's lengths, kept flag
    bool pk = false;
    const double hp = 0.5 * PI;
    const int g_begin = rsa_group_begin[blockIdx.y], g_end = rsa_group_begin[blockIdx.y + 1];

    for (int g = g_begin; g < g_end; g++) {
        const bool back = (g == 4 || g == 9 || g == 10);
        const double bx = back ? XB : X, by = back ? YB : Y;
        const double sx = (q & 1) ? -bx : bx;
        const double sy = (q & 2) ? -by : by;
        const double sp = (q == 1 || q == 2) ? -PHI : PHI;
        const double ssp = (q == 1 || q == 2) ? -sPHI : sPHI, csp = cPHI;
        double t = 0, u = 0, v = 0;
        bool ok = false;
        switch (g) {                                          // wave-uniform
            case 0: ok = (q == 0 || q == 2) && rs_SLS(sx, sy, sp, t, u, v); break;   // SCS (:120-130): q = 0, 2 only
            case 1: ok = rs_LSL(sx, sy, sp, ssp, csp, t, u, v); break;
            case 2: ok = rs_LSR(sx, sy, sp, ssp, csp, t, u, v); break;
            case 3: case 4: ok = rs_LRL(sx, sy, sp, ssp, csp, t, u, v); break;
            case 5: ok = rs_LRLRn(sx, sy, sp, ssp, csp, t, u, v); break;
            case 6: ok = rs_LRLRp(sx, sy, sp, ssp, csp, t, u, v); break;
            case 7: case 9: ok = rs_LRSL(sx, sy, sp, ssp, csp, t, u, v); break;
            case 8: case 10: ok = rs_LRSR(sx, sy, sp, ssp, csp, t, u, v); break;
            default: ok = rs_LRSLR(sx, sy, sp, ssp, csp, t, u, v); break;
        }
        ok = ok && live;
        int t0 = TL, t1 = TS, t2_ = TL, t3 = 0, t4 = 0, n = 3;
        double l0 = t, l1 = u, l2 = v, l3 = 0, l4 = 0;
        switch (g) {
            case 0: t0 = TS; t1 = TL; t2_ = TS; break;
            case 1: t0 = TL; t1 = TS; t2_ = TL; break;
            case 2: t0 = TL; t1 = TS; t2_ = TR; break;
            case 3: t0 = TL; t1 = TR; t2_ = TL; break;
            case 4: t0 = TL; t1 = TR; t2_ = TL; l0 = v; l2 = t; break;
            case 5: n = 4; t0 = TL; t1 = TR; t2_ = TL; t3 = TR; l2 = -u; l3 = v; break;
            case 6: n = 4; t0 = TL; t1 = TR; t2_ = TL; t3 = TR; l2 = u; l3 = v; break;
            case 7: n = 4; t0 = TL; t1 = TR; t2_ = TS; t3 = TL; l1 = -hp; l2 = u; l3 = v; break;
            case 8: n = 4; t0 = TL; t1 = TR; t2_ = TS; t3 = TR; l1 = -hp; l2 = u; l3 = v; break;
            case 9: n = 4; t0 = TL; t1 = TS; t2_ = TR; t3 = TL; l0 = v; l1 = u; l2 = -hp; l3 = t; break;
            case 10: n = 4; t0 = TR; t1 = TS; t2_ = TR; t3 = TL; l0 = v; l1 = u; l2 = -hp; l3 = t; break;
            default: n = 5; t0 = TL; t1 = TR; t2_ = TS; t3 = TL; t4 = TR; l1 = -hp; l2 = u; l3 = -hp; l4 = v; break;
        }
        if (q & 1) { l0 = -l0; l1 = -l1; l2 = -l2; l3 = -l3; l4 = -l4; }
        if (q & 2) { t0 = mirror_type(t0); t1 = mirror_type(t1); t2_ = mirror_type(t2_); t3 = mirror_type(t3); t4 = mirror_type(t4); }
        if (n < 4) { t3 = 0; l3 = 0; }
        if (n < 5) { t4 = 0; l4 = 0; }
        const int code = pack_types(t0, t1, t2_, t3, t4, n);
        double L = 0;                                         // :68-71
        L = L + fabs(l0); L = L + fabs(l1); L = L + fabs(l2);
        if (n > 3) L = L + fabs(l3);
        if (n > 4) L = L + fabs(l4);

        // ---- set_path: earlier KEPT words with my type sequence ------------------------------------------------
        // previous family (LRL before backwards-LRL, LRLRn before LRLRp): its reflections (q & 2) and (q & 2) + 1
        bool dup = false;
        if (g == 4 || g == 6) {                               // wave-uniform
            const double a0 = (q & 2) ? quad_get<2>(p0) : quad_get<0>(p0), a1 = (q & 2) ? quad_get<2>(p1) : quad_get<0>(p1);
            const double a2 = (q & 2) ? quad_get<2>(p2) : quad_get<0>(p2), a3 = (q & 2) ? quad_get<2>(p3) : quad_get<0>(p3);
            const int ak = (q & 2) ? quad_get<2>((int)pk) : quad_get<0>((int)pk);
            const double b0 = (q & 2) ? quad_get<3>(p0) : quad_get<1>(p0), b1 = (q & 2) ? quad_get<3>(p1) : quad_get<1>(p1);
            const double b2 = (q & 2) ? quad_get<3>(p2) : quad_get<1>(p2), b3 = (q & 2) ? quad_get<3>(p3) : quad_get<1>(p3);
            const int bk = (q & 2) ? quad_get<3>((int)pk) : quad_get<1>((int)pk);
            dup = (ak && rs_dup(n, a0, a1, a2, a3, 0, l0, l1, l2, l3, l4)) || (bk && rs_dup(n, b0, b1, b2, b3, 0, l0, l1, l2, l3, l4));
        }
        // even reflections first (no predecessor inside the family), then the odd ones against (g, q - 1)
        bool kept = ok && !dup && !(L >= MAX_LENGTH) && !(q & 1);
        {
            const double e0 = (q & 2) ? quad_get<2>(l0) : quad_get<0>(l0), e1 = (q & 2) ? quad_get<2>(l1) : quad_get<0>(l1);
            const double e2 = (q & 2) ? quad_get<2>(l2) : quad_get<0>(l2), e3 = (q & 2) ? quad_get<2>(l3) : quad_get<0>(l3);
            const double e4 = (q & 2) ? quad_get<2>(l4) : quad_get<0>(l4);
            const int ek = (q & 2) ? quad_get<2>((int)kept) : quad_get<0>((int)kept);
            if (q & 1) kept = ok && !dup && !(ek && rs_dup(n, e0, e1, e2, e3, e4, l0, l1, l2, l3, l4)) && !(L >= MAX_LENGTH);
        }
        const double Lm = L / MAXC;                            // path.L / maxc (calc_all_paths :52)
        // ---- costQueue[path] = path.L in path order (:432-433): the key of my candidate slot; k_rs_segs replays the heap ----
        if (live) rec[RS_REC_KEYS + 4 * g + q] = kept ? Lm : -1.0;
        if (kept) {
            RsWord* w = words + 4 * g + q;
            w->len[0] = l0; w->len[1] = l1; w->len[2] = l2; w->len[3] = l3; w->len[4] = l4;
            w->Lm = Lm; w->code = code; w->n = n;
        }
        p0 = l0; p1 = l1; p2 = l2; p3 = l3; pk = kept;
    }
}

// ================================================================================================
// Kernel A2: segment origins of the words the search will test -- ONE LANE PER WORD (eight searches per wave)
// ================================================================================================
// generate_local_course (:452-507) needs, per segment, its origin (ox, oy, oyaw) = the end point of the previous segment
// (interpolate(ind, l, ...) :497-498; hm_cos(-x) = cos x and hm_sin(-x) = -sin x give :522-523).  This is scalar arithmetic
// per word (two sincos and two divisions per segment); inside the validation kernel, one wave per search, it cost the
// whole wave ~400 instructions per tested word -- a quarter of that kernel's cycles.
__global__ __launch_bounds__(64) void k_rs_segs(RsParams p) {
    set_wave_prio(p.prio_front);
    __shared__ double keyl[8][RS_WORDS_PER_SCENE];            // candidate keys in path order
    __shared__ double prl[8][RS_WORDS_PER_SCENE];             // heap priorities per search
    __shared__ unsigned char hidl[8][RS_WORDS_PER_SCENE];     // heap ids (candidate slot)
    __shared__ unsigned char ordl[8][RS_WORDS_PER_SCENE];     // pop order
    __shared__ int ntl[8];
    __shared__ double hdrl[8][8];                            // per search: start pose x, y, heading, map box xmin, xmax, ymin, ymax
    const int lane = threadIdx.x, ls = lane >> 3, k0 = lane & 7;
    // the queue entry is requested BEFORE the queue length is known (see k_rs_words): a memory round trip less
    const int qi = blockIdx.x * 8 + ls;
    const int qs = qi < p.max_queue ? qi : p.max_queue - 1;
    const int entry = p.rs_list[qs];
    const int scene = rs_list_scene(entry);
    double* rec = p.rs_rec + (size_t)(p.slot_base + p.slot_dir * qs) * RS_REC_DOUBLES;
    const int count = *p.rs_count;
    if ((int)blockIdx.x * 8 >= count) return;
    const bool live = qi < count;
    double keyv[RS_WORDS_PER_SCENE / 8];
#pragma unroll
    for (int j = 0; j < RS_WORDS_PER_SCENE / 8; j++) keyv[j] = live ? rec[RS_REC_KEYS + 8 * j + k0] : -1.0;
    // the header's inputs (k_rs_compact's row of this queue position), requested now: they arrive while lane 0 replays the heap
    const double* in = p.rs_in + (size_t)qs * RS_IN_WORDS;
    double hdr = 0.0;                                       // lanes k0 = 0 .. 6 of a search: pose x, y, heading, map box
    int hdr_n = 0;
    if (k0 < 3) hdr = in[k0];
    else if (k0 < 7) hdr = in[k0];
    else hdr_n = rs_list_n_obst(entry);      // the obstacle count k_rs_compact read
    // the kept candidates of the search as a bit mask (bit c = candidate slot c): eight lanes x six keys, one ballot per stride
    unsigned long long keptm = 0;
#pragma unroll
    for (int j = 0; j < RS_WORDS_PER_SCENE / 8; j++) {
        const double key = live ? keyv[j] : -1.0;
        keyl[ls][8 * j + k0] = key;
        keptm |= ((__ballot(key >= 0.0) >> (8 * ls)) & 0xffull) << (8 * j);
    }
    __syncthreads();
    // ---- costQueue = heapdict(); costQueue[path] = path.L in path order (:432-433), then popitem until the stop rule ----
    // heapdict 1.0.1's array heap: push = append + sift-up that swaps unless parent < child; popitem = move the last entry
    // to the root + strict sift-down.  Replayed by the first of the search's eight lanes.
    if (live && k0 == 0) {
        double* pr = prl[ls];
        unsigned char* hid = hidl[ls];
        int hn = 0;
        for (unsigned long long m = keptm; m; m &= m - 1) {    // the candidates set_path kept, in path order
            const int c = __ffsll((long long)m) - 1;
            const double pv = keyl[ls][c];
            int i = hn++;
            pr[i] = pv; hid[i] = (unsigned char)c;
            while (i) {                                       // _decrease_key: swap unless parent < child
                const int parent = (i - 1) >> 1;
                if (pr[parent] < pr[i]) break;
                const double tp = pr[i]; pr[i] = pr[parent]; pr[parent] = tp;
                const unsigned char ti = hid[i]; hid[i] = hid[parent]; hid[parent] = ti;
                i = parent;
            }
        }
        const int n_kept = hn;
        int no = 0;
        double lmin = 0;
        while (hn > 0) {                                      // popitem
            const double lm = pr[0];
            if (no == 0) lmin = lm;
            if (lm > 1.6 * lmin && no + 1 > 2) break;         // stop rule (:443): this word and every later one are never tested
            ordl[ls][no++] = hid[0];
            if (hn == 1) { hn = 0; break; }
            hn--;
            pr[0] = pr[hn]; hid[0] = hid[hn];
            int i = 0;
            for (;;) {                                        // _min_heapify
                const int l = (i << 1) + 1, r = (i + 1) << 1;
                int low = (l < hn && pr[l] < pr[i]) ? l : i;
                if (r < hn && pr[r] < pr[low]) low = r;
                if (low == i) break;
                const double tp = pr[i]; pr[i] = pr[low]; pr[low] = tp;
                const unsigned char ti = hid[i]; hid[i] = hid[low]; hid[low] = ti;
                i = low;
            }
        }
        ntl[ls] = no;
        ((int2*)rec)[1] = make_int2(n_kept, no);
    }
    if (live) {
        // header: everything k_rs_validate needs before it can stage the obstacle tile -- [0] scene, n_obst [1] candidates kept,
        // words to test [2..4] start pose [5..8] map box; one field per lane of the search's eight
        if (k0 < 7) rec[2 + k0] = hdr;
        else ((int2*)rec)[0] = make_int2(scene, hdr_n);
        hdrl[ls][k0] = hdr;
    }
    __syncthreads();
    if (!live) return;
    const int n_test = ntl[ls];
    const unsigned char* order = ordl[ls];
    const double q0w = hdrl[ls][2];
    double sq0, cq0;                                      // world heading of the start pose: rotation local course -> world
    hm_sincos(q0w, &sq0, &cq0);
    // start position in the frame of the scene's float32 obstacle view (origin = map box xmin, ymin: integers)
    const double fq0x = hdrl[ls][0] - hdrl[ls][3], fq0y = hdrl[ls][1] - hdrl[ls][5];
    for (int k = k0; k < n_test; k += 8) {
        const double* W = rec + RS_REC_WORDS + 8 * (int)order[k];
        double len[5];
#pragma unroll
        for (int i = 0; i < 5; i++) len[i] = W[i];
        const double w6 = W[6];
        const int code = __double2loint(w6), nseg = __double2hiint(w6);
        double* tb = rec + RS_REC_SEGS + RS_SEG_TABLE * k;
        double ox = 0, oy = 0, hy = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i < nseg) {
                const int m = type_of(code, i);
                const double l = len[i];
                double s_oy, c_oy;
                hm_sincos(hy, &s_oy, &c_oy);
                double* sp_ = tb + RS_SEGW * i;
                sp_[0] = ox; sp_[1] = oy; sp_[2] = hy; sp_[3] = c_oy; sp_[4] = s_oy; sp_[5] = (double)m; sp_[6] = l;
                if (i == 0) sp_[7] = w6;
                {   // float32 filter row: origin rotated into the world axes, in the frame of the scene's float32 obstacle view, and
                    // the world heading at the origin by angle addition (only ever used with error margins)
                    const float fox = (float)(fq0x + (cq0 * ox - sq0 * oy)), foy = (float)(fq0y + (sq0 * ox + cq0 * oy));
                    const float fc = (float)(c_oy * cq0 - s_oy * sq0), fs = (float)(s_oy * cq0 + c_oy * sq0);
                    tb[RS_SEG_F32 + 2 * i] = __hiloint2double(__float_as_int(foy), __float_as_int(fox));
                    tb[RS_SEG_F32 + 2 * i + 1] = __hiloint2double(__float_as_int(fs), __float_as_int(fc));
                }
                if (m == TS) {                                   // interpolate(l) = next origin (:512-513)
                    ox = ox + l / MAXC * c_oy;
                    oy = oy + l / MAXC * s_oy;
                } else {
                    double sl, cl;
                    hm_sincos(l, &sl, &cl);
                    const double ldx = sl / MAXC;
                    const double ldy = (m == TL) ? (1.0 - cl) / MAXC : (1.0 - cl) / (-MAXC);
                    ox = ox + (c_oy * ldx + (-s_oy) * ldy);
                    oy = oy + (s_oy * ldx + c_oy * ldy);
                }
                hy = (m == TL) ? hy + l : ((m == TR) ? hy - l : hy);
            }
        }
        if (k == 0) {                                    // calc_all_paths' rotation by -q0 yaw (:47-49), once per search:
            tb[RS_SEGW + 7] = cq0;                       // hm_cos(-q0w): spare words of table 0 (hm_sincos is exactly even / odd)
            tb[2 * RS_SEGW + 7] = -sq0;                  // hm_sin(-q0w)
        }
    }
}

// ================================================================================================
// Kernel B: find_rs_path's main loop (:436-450) over the ordered words
// ================================================================================================
// LDS: tile 8*M doubles | obstacle boxes M float4 | scratch (doubles): the current word's segment table 5 x 8, sample queue
//      pd[256], bad1[6+2] | ints: cand[M] | bytes: seg[256]
constexpr int RSB_SEG = 0, RSB_SEGW = RS_SEGW, RSB_QPD = RS_SEG_TABLE, RSB_QCAP = 256, RSB_BAD = RSB_QPD + RSB_QCAP;
constexpr int RSB_WORDS = RSB_BAD + 8;

template <int OCC, bool TIMING>
__global__ __launch_bounds__(64, OCC) void k_rs_validate(RsParams p, int obs_f64) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    unsigned long long tsec[16] = {};
    const unsigned long long tstart_ = TIMING ? __builtin_readcyclecounter() : 0;
    RS_T0();
    const int count = *p.rs_count;
    if ((int)blockIdx.x >= count) return;
    if (obs_f64 & 0x400) return;                          // profiling switch: words kernel only
    // the queue is sorted by scene: spread neighbouring (similar) scenes over the XCDs like the step kernel does
    const int qidx = scene_of_block(blockIdx.x, count);
    const int slot = p.slot_base + p.slot_dir * qidx;
    double* tile = lds;
    float4* obb = (float4*)(lds + 8 * p.tile_cap);
    double* scr = lds + 10 * p.tile_cap;
    double* segp = scr + RSB_SEG;
    double* qpd = scr + RSB_QPD;
    int* cand = (int*)(scr + RSB_WORDS);
    unsigned char* qseg = (unsigned char*)(cand + ((p.tile_cap + 3) & ~3));

    // ---- the search record: one coalesced load brings the header; the words come as segment tables (k_rs_segs) ----
    const double* rec = p.rs_rec + (size_t)slot * RS_REC_DOUBLES;
    const double r0 = lane < RS_REC_HDR ? rec[lane] : 0.0;
    const double* tables = rec + RS_REC_SEGS;
    double tb = lane < RS_SEG_TABLE ? tables[lane] : 0.0;          // segment table of the first word: in flight during the staging
    const int n_paths = __builtin_amdgcn_readlane(__double2hiint(r0), 1);      // words the stop rule lets the search test
    if (n_paths == 0) return;
    const int scene = __builtin_amdgcn_readlane(__double2loint(r0), 0);
    const int n_obst = __builtin_amdgcn_readlane(__double2hiint(r0), 0);
    {
        const double2* src = (const double2*)(p.verts + (size_t)scene * p.max_obst * 8);
        double2* dst = (double2*)tile;
        for (int v = lane; v < 4 * n_obst; v += WAVE) dst[v] = src[v];
        const float4* ob = p.obb + (size_t)scene * p.max_obst;   // obstacle boxes (xmin, xmax, ymin, ymax), rounded outwards
        for (int o = lane; o < n_obst; o += WAVE) obb[o] = ob[o];
    }
    const double q0x = readlane_d(r0, 2), q0y = readlane_d(r0, 3), q0w = readlane_d(r0, 4);
    const double xmin = readlane_d(r0, 5), xmax = readlane_d(r0, 6), ymin = readlane_d(r0, 7), ymax = readlane_d(r0, 8);
    const double step = RS_STEP * MAXC;                   // step_size * maxc (:44)
    wsync();

    int found = -1;
    // Every path starts at the same pose, and the samples of its FIRST segment depend only on that segment's type and
    // direction: pd = d, 2d, ... from the origin (0, 0, 0).  So a first-segment sample found in collision condemns every
    // later word that starts with the same type and direction and is long enough to contain it (|l0| >= |pd|, the
    // generator's own inclusion test) -- those words are skipped without being sampled.  bad1[type * 2 + forward].
    double* bad1 = scr + RSB_BAD;                         // LDS, wave-uniform
    if (lane < 6) bad1[lane] = INFINITY;
    wsync();
    RS_T(0);
    const double c_q = readlane_d(tb, RS_SEGW + 7), s_q = readlane_d(tb, 2 * RS_SEGW + 7);   // cos / sin(-q0 yaw) (k_rs_segs)
    for (int idx = 1; idx <= n_paths; idx++) {            // the stop rule (:443) is already applied: n_paths ends there
        if ((obs_f64 & 0x200) && idx > 1) break;          // profiling switch: first path only
        const double tcur = tb;
        if (idx < n_paths && lane < RS_SEG_TABLE) tb = tables[RS_SEG_TABLE * idx + lane];      // prefetch the next word's
        const double len0 = readlane_d(tcur, 6), w6 = readlane_d(tcur, 7);
        const int code = __double2loint(w6), nseg = __double2hiint(w6);
        const int cls1 = type_of(code, 0) * 2 + (len0 > 0.0 ? 1 : 0);
        if (!(obs_f64 & 0x800) && fabs(len0) >= bad1[cls1]) continue;    // contains a sample already known to collide

        // generate_local_course (:452-507).  Samples are queued as (pd, segment) and collision-tested 64 at a
        // time, so short segments share a pass.  Sample 0 (the start pose, local (0,0,0)) = segment 0 at pd = 0.
        // The segment origins come ready-made from k_rs_segs.
        bool invalid = false;
        if (lane < RS_SEG_TABLE) segp[lane] = tcur;
        if (lane == 0) { qpd[0] = 0.0; qseg[0] = 0; }
        int nq = 1;
        wsync();
        if (TIMING) tsec[9] += 1;
        RS_T(1);
        // Resumable sample generator + ONE window test site.  Samples are appended to the queue until it cannot
        // take another chunk (or the path ends); then the window is tested coarse-to-fine: a path is invalid as
        // soon as ANY of its samples is bad, whatever the order they are looked at, and a path that crosses an
        // obstacle has long runs of bad samples -- so a strided subset that fills one wave goes first and only clean
        // windows pay for the remaining samples.  Exact: all of them are real samples
        // of generate_local_course.
        // window size 128: the generator runs while fewer than 64 samples are queued, so every test is one whole wave
        // (64 samples in path order) as early as possible -- invalid paths mostly hit something within their first
        // metres (measured 128 / 130 / 160 / 194 / 258: 0.645 / 0.659 / 0.689 / 0.759 / 0.778 ms).  At least 128: the
        // generator must always be able to bring the queue to a whole wave.
        const int win = (obs_f64 >> 12) ? min(max(obs_f64 >> 12, 128), RSB_QCAP - WAVE - 1) : 128;
        int i = 0;
        bool seg_open = false, finished = false;
        double pd = 0, ll = 0.0, lprev = 0.0, d = 0, l = 0;
        while (!invalid && (!finished || nq > 0)) {
            while (!finished && nq + WAVE + 1 <= win) {
                if (TIMING) tsec[10] += 1;
                if (!seg_open) {
                    l = segp[RSB_SEGW * i + 6];               // len[i]
                    d = l > 0.0 ? step : -step;
                    if (i >= 1 && (lprev * l) > 0) pd = -d - ll; else pd = d - ll;
                    lprev = l;
                    seg_open = true;
                }
                // `pd += d` chain (sequential rounding kept): every lane walks it, lane j keeps the value after
                // j additions; the walk stops at the first block of 8 whose last value already left the segment
                // (lane j = 8 b + r takes the value at the start of block b from the common walk and adds d another r times
                // itself: the same additions in the same order, without a select per step)
                double mine = pd, t = pd, last = pd;
                int ncap = 0;
                for (int j0 = 0; j0 < WAVE; j0 += 8) {
                    if ((lane >> 3) == (j0 >> 3)) mine = t;
#pragma unroll
                    for (int jj = 0; jj < 7; jj++) t = t + d;
                    last = t;
                    t = t + d;
                    ncap = j0 + 8;
                    if (__any(fabs(last) > fabs(l))) break;
                }
#pragma unroll
                for (int k = 0; k < 7; k++)
                    if (k < (lane & 7)) mine = mine + d;
                const bool in_seg = lane < ncap && fabs(mine) <= fabs(l);
                const unsigned long long valid = (ncap == WAVE) ? ~0ull : ((1ull << ncap) - 1);
                const unsigned long long fail = ~__ballot(in_seg) & valid;
                const int count = fail ? (__ffsll((long long)fail) - 1) : ncap;
                if (lane < count) { qpd[nq + lane] = mine; qseg[nq + lane] = (unsigned char)i; }
                nq += count;
                if (fail) {                                   // first value outside the segment: segment done
                    pd = __shfl(mine, count);
                    ll = l - pd - d;                          // "calc remain length" (:494)
                    seg_open = false;
                    if (i == nseg - 1) {                      // the final end point is the only segment end kept
                        if (lane == 0) { qpd[nq] = l; qseg[nq] = (unsigned char)i; }
                        nq += 1;
                        finished = true;
                    }
                    i++;
                } else pd = t;                                // all 64 inside: keep walking
            }
            wsync();
            RS_T(2);
            // only whole waves of samples are tested while the path goes on; the remainder (< 64) stays queued and is
            // topped up by the next segments, so every pass but the path's last runs with all 64 lanes busy
            const int n_all = nq;
            const int n = finished ? n_all : (n_all & ~(WAVE - 1));
            // coarse pass: every `stride`-th sample with stride = ceil(n / 64), i.e. as dense as one full-wave pass
            // allows (a fixed stride of 8 left 3/4 of the lanes idle on a 130-sample window)
            const int stride = (n + WAVE - 1) / WAVE;
            const int n_coarse = (n + stride - 1) / stride;
            const int n_rest = n - n_coarse;
            for (int rnd = 0; rnd == 0 || (rnd - 1) * WAVE < n_rest; rnd++) {
                int idx;
                if (rnd == 0) idx = (lane < n_coarse) ? stride * lane : -1;
                else { const int r = (rnd - 1) * WAVE + lane; idx = r < n_rest ? r + r / (stride - 1) + 1 : -1; }
                const bool active = idx >= 0;
                double px = 0, py = 0, pyaw = 0;
                if (active) {
                    const double spd = qpd[idx];
                    const double* sp_ = segp + RSB_SEGW * (int)qseg[idx];
                    const int m = (int)sp_[5];
                    interpolate(spd, m, sp_[0], sp_[1], sp_[2], sp_[3], -sp_[4], sp_[3], sp_[4], px, py, pyaw);
                }
                const double wx = c_q * px + s_q * py + q0x;      // calc_all_paths :47-49
                const double wy = -s_q * px + c_q * py + q0y;
                const double wyaw = pi_2_pi(pyaw + q0w);
                if (TIMING) tsec[11] += 1;
                RS_T(3);
                const bool hit = !(obs_f64 & 0x100) && pose_hits<TIMING>(active, wx, wy, wyaw, tile, obb, cand, n_obst, xmin, xmax, ymin, ymax, lane, tsec);
                if (TIMING) t0_ = __builtin_readcyclecounter();
                if (__any(hit)) {
                    // remember the nearest colliding sample of the FIRST segment (re-read from the queue: rare path)
                    const double mine1 = (hit && active && qseg[idx] == 0) ? fabs(qpd[idx]) : INFINITY;
                    const double v = fmin(bad1[cls1], wave_min_d(mine1));
                    wsync();
                    if (lane < 6 && (lane == cls1 || v == 0.0)) bad1[lane] = fmin(bad1[lane], v);   // the start pose (pd = 0) is in every path
                    invalid = true;
                    break;
                }
            }
            {   // carry the untested tail to the front of the queue
                const int rem = n_all - n;
                double cd = 0;
                unsigned char cs = 0;
                if (lane < rem) { cd = qpd[n + lane]; cs = qseg[n + lane]; }
                wsync();
                if (lane < rem) { qpd[lane] = cd; qseg[lane] = cs; }
                nq = rem;
            }
            wsync();
        }
        if (!invalid) { found = idx - 1; break; }
    }
    if (TIMING) {
        tsec[6] = __builtin_readcyclecounter() - tstart_;
        tsec[8] = 1;
        if (lane == 0) {
            for (int i = 0; i < 16; i++) if (tsec[i]) atomicAdd(&g_rs_prof[(blockIdx.x & 63) * 16 + i], tsec[i]);
            const int li = atomicAdd(&g_rs_log_n, 1);
            if (li < RS_LOG_CAP) g_rs_log[li] = make_int4((int)tsec[6], (int)tsec[9] | (n_paths << 8) | ((p.tile_cap > 32) << 16), (int)tsec[11], found >= 0);
        }
    }
    if (found < 0) return;

    // ---- output: PATH.ctypes / PATH.lengths (metres) of the first collision-free path ----------------
    // `found` is the last table copied to segp: lengths at [8 i + 6], type code and segment count at [7]
    const double wlen = lane < 5 ? segp[RSB_SEGW * lane + 6] : 0.0;
    const int code = __double2loint(segp[7]), nseg = __double2hiint(segp[7]);
    if (lane < 5) {
        double lm = lane < nseg ? wlen / MAXC : 0.0;              // path.lengths = [l / maxc ...] (:51)
        if (p.rs_lengths) {
            if (obs_f64 & 1) ((double*)p.rs_lengths)[5 * (size_t)scene + lane] = lm;
            else ((float*)p.rs_lengths)[5 * (size_t)scene + lane] = (float)lm;
        }
        p.rs_word[8 * (size_t)scene + lane] = lane < nseg ? (int8_t)type_of(code, lane) : (int8_t)HOPE_RS_NONE;
    }
    if (lane == 5) p.rs_word[8 * (size_t)scene + 5] = (int8_t)nseg;
    if (lane == 6) p.rs_word[8 * (size_t)scene + 6] = 1;
}


// ================================================================================================
// Kernel B': the same search with a float32 FILTER in front of the float64 arithmetic
// ================================================================================================
// k_rs_validate above evaluates every sample's pose, hull and edge tests in float64 (two hm_sincos per sample, ~250 float64
// instructions per candidate obstacle).  Almost every pass either contains samples that are DEEP inside an obstacle or is
// clear by centimetres, so this kernel decides those in float32 with explicit error margins and runs the reference
// arithmetic (exact_pass: the float64 code of the kernel above, unchanged) only for samples it cannot decide:
//   * frame: world axes, origin at the search's start position (coordinates stay below ~100 m: float32 ulp <= 8e-6 m);
//     obstacle vertices are converted once per search, the sample poses come from k_rs_segs' float32 segment rows;
//   * per sample and candidate obstacle the four vertices are taken into the CAR frame, where the hull is the axis-aligned
//     rectangle |u| <= HL, |w| <= HW, and every obstacle edge is classified against it with margin FEPS:
//       clear    separated from the rectangle inflated by FEPS on one of the three SAT axes of (segment, rectangle), or
//                both end points inside the rectangle deflated by FEPS (edge inside the hull: boundaries do not meet);
//       hit      meets the deflated rectangle with an end point outside the inflated one (it crosses the hull boundary),
//                AND the crossing is robust for the reference's tolerance-free test (:518-526, the intersection of the two
//                LINES must lie inside both edges' coordinate boxes): no hull corner within FEPS of the edge's line, the
//                edge not within FKAPPA (sine) of parallel to a hull side its line crosses, neither edge axis-parallel in
//                the world frame (a degenerate coordinate box is met only by bit-equality) -- then the exact intersection
//                point lies >= 1e-7 m inside both boxes, eight orders above the rounding of raw_x / raw_y;
//       else     undecided.
//   * float32 error budget (FEPS = 1e-3 m is ~8x the bound): pd -> float 5e-7 (x 3 m/unit), native sin / cos 4e-6 (x 3 m,
//     and x 3.9 m lever arm on the heading), ~10 roundings of <= 100 m quantities at 6e-6 each: <= 1.2e-4 m in total.
// A pass with a `hit` sample condemns the word at once; a pass with undecided samples and no hit re-evaluates just those
// samples in float64; a pass whose samples are all clear is clear.  Identical results to k_rs_validate by construction
// (the tests run both kernels on the same queues: tests/test_gpu_parity.py); HOPE_RS_EXACT=1 selects the kernel above.
constexpr float FEPS = 1e-3f, FKAPPA = 2e-2f, FETA_HULL = 2e-3f;      // (FETA_EDGE: hope_dev.h, obstacle_f32)
constexpr float F_INV_MAXC = (float)(1.0 / MAXC);
constexpr float F_HL = (float)(0.5 * (CAR_XF - CAR_XR)), F_HW = (float)CAR_YH, F_MID = (float)(0.5 * (CAR_XF + CAR_XR));
constexpr int RS_TAB = 256;
constexpr int RSB_TABF = RSB_BAD + 8, RSB_WORDS_F = RSB_TABF + RS_TAB;   // k_rs_validate_f: + the first-segment sample table
constexpr int RSF_OCC = 4;                 // waves per SIMD the register allocation of k_rs_validate_f aims at
__device__ double g_rs_steps[RS_TAB];     // first-segment samples: T[k] = step added k + 1 times in sequence (every word's
                                          // first segment starts with pd = d and walks pd += d: the same chain for all)
__device__ unsigned long long g_rs_fstat[16];
__device__ double g_rs_fdump[64 * 16];    // self-check: details of the first 64 contradicted verdicts   // [0] passes [1] passes decided by a float32 hit [2] passes that ran exact_pass
                                               // [3] undecided samples [4] float32 "hit" that exact says clear (must stay 0)
                                               // [5] float32 "clear" that exact says hit (must stay 0) [6] samples checked
                                               // [8..12] passes with an undecided edge because: not a certain crossing / axis-parallel
                                               // obstacle edge / axis-parallel hull / hull corner near the line / shallow angle
                                               // [7] words the screen condemned that the full walk found VALID (self-check build; must stay 0)
                                               // [13] words condemned by the screen [14] searches whose every word the screen condemned

// the float64 evaluation of the queued samples `idx` (one per lane, -1: none): interpolate (:510-537), calc_all_paths'
// rotation (:47-49) and is_traj_valid's tests -- exactly the per-pass body of k_rs_validate.  Rare: out of line, reads the
// obstacle tile and boxes from global memory.
template <bool FULL>
__device__ __noinline__ bool exact_pass(int idx, const double* segp, const double* qpd, const unsigned char* qseg, double c_q,
                                        double s_q, double q0x, double q0y, double q0w, const double* verts_g,
                                        const float4* obb_g, int* cand, int n_obst, double xmin, double xmax, double ymin,
                                        double ymax, int lane) {
    const bool active = idx >= 0;
    double px = 0, py = 0, pyaw = 0;
    if (active) {
        const double spd = qpd[idx];
        const double* sp_ = segp + RSB_SEGW * (int)qseg[idx];
        const int m = (int)sp_[5];
        interpolate(spd, m, sp_[0], sp_[1], sp_[2], sp_[3], -sp_[4], sp_[3], sp_[4], px, py, pyaw);
    }
    const double wx = c_q * px + s_q * py + q0x;
    const double wy = -s_q * px + c_q * py + q0y;
    const double wyaw = pi_2_pi(pyaw + q0w);
    unsigned long long tsec[16];
    return pose_hits<false, FULL>(active, wx, wy, wyaw, verts_g, obb_g, cand, n_obst, xmin, xmax, ymin, ymax, lane, tsec);
}

// The reference arithmetic for the samples the float32 filter left undecided, and only for the edges it left open (ua / ub:
// obstacle | edge mask << 8): every other edge of these samples is certainly clear.  A sample with open edges on more than two
// obstacles (uover: rare) takes every edge of every candidate obstacle of the pass (cand[0 .. nc)) instead -- the exact test of an
// edge the filter found clear says clear, so looking at more edges cannot change the verdict.  Same expressions as pose_hits.
// INLINE with rolled loops and the hull corners recomputed where they are used (round 4; it was an out-of-line function with
// corner arrays: its callee-saved registers went to scratch memory on every call -- 13 x the kernel's output bytes in HBM writes --
// and a kernel with a call cannot be free of scratch at all).
__device__ __forceinline__ void hull_corner(int k, double ct, double st, double wx, double wy, double& cx, double& cy) {
    cx = ct * car_x(k) - st * car_y(k) + wx;                                     // :468-471
    cy = st * car_x(k) + ct * car_y(k) + wy;
}
__device__ __forceinline__ bool exact_edges(bool unc, int sidx, int ua, int ub, bool uover, const int* cand, int nc, const double* segp,
                                            const double* qpd, const unsigned char* qseg, const double* tile64, const double* rec) {
    bool bad = false;
    if (unc) {
        // start pose, map box and the rotation of calc_all_paths: from the search record (global memory, the same address on every
        // lane) -- kept in registers across the whole kernel they were 18 SGPRs of a register file that was already spilling
        const double q0x = rec[2], q0y = rec[3], q0w = rec[4], xmin = rec[5], xmax = rec[6], ymin = rec[7], ymax = rec[8];
        const double c_q = rec[RS_REC_SEGS + RS_SEGW + 7], s_q = rec[RS_REC_SEGS + 2 * RS_SEGW + 7];     // cos / sin(-q0 yaw) (k_rs_segs)
        double px = 0, py = 0, pyaw = 0;
        {
            const double spd = qpd[sidx];
            const double* sp_ = segp + RSB_SEGW * (int)qseg[sidx];
            interpolate(spd, (int)sp_[5], sp_[0], sp_[1], sp_[2], sp_[3], -sp_[4], sp_[3], sp_[4], px, py, pyaw);
        }
        const double wx = c_q * px + s_q * py + q0x;
        const double wy = -s_q * px + c_q * py + q0y;
        const double wyaw = pi_2_pi(pyaw + q0w);
        bad = wx < xmin || wx > xmax || wy < ymin || wy > ymax;             // :462-464
        double st_, ct_;
        hm_sincos(wyaw, &st_, &ct_);
        double hminx, hmaxx, hminy, hmaxy;
        {
            double cx0, cy0, cx1, cy1, cx2, cy2, cx3, cy3;
            hull_corner(0, ct_, st_, wx, wy, cx0, cy0); hull_corner(1, ct_, st_, wx, wy, cx1, cy1);
            hull_corner(2, ct_, st_, wx, wy, cx2, cy2); hull_corner(3, ct_, st_, wx, wy, cx3, cy3);
            hminx = fmin(fmin(cx0, cx1), fmin(cx2, cx3)); hmaxx = fmax(fmax(cx0, cx1), fmax(cx2, cx3));
            hminy = fmin(fmin(cy0, cy1), fmin(cy2, cy3)); hmaxy = fmax(fmax(cy0, cy1), fmax(cy2, cy3));
        }
        const int n_slot = uover ? nc : 2;
#pragma unroll 1
        for (int slot = 0; slot < n_slot; slot++) {
            const int rec_ = uover ? (cand[slot] | 0xF00) : (slot ? ub : ua);
            if (rec_ < 0) continue;
            const double* o = tile64 + 8 * (rec_ & 0xff);
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                if (!((rec_ >> (8 + j)) & 1)) continue;
                const double x1 = o[2 * j], y1 = o[2 * j + 1], x2 = o[2 * ((j + 1) & 3)], y2 = o[2 * ((j + 1) & 3) + 1];
                const double exmin = fmin(x1, x2), exmax = fmax(x1, x2), eymin = fmin(y1, y2), eymax = fmax(y1, y2);
                if (exmin > hmaxx || exmax < hminx || eymin > hmaxy || eymax < hminy) continue;
                const double d_ = y2 - y1, e_ = x1 - x2, f_ = y1 * x2 - x1 * y2;                  // :504-506
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    double ax1, ay1, ax2, ay2;
                    hull_corner(k, ct_, st_, wx, wy, ax1, ay1);
                    hull_corner((k + 1) & 3, ct_, st_, wx, wy, ax2, ay2);
                    const double vminx = fmin(ax1, ax2), vmaxx = fmax(ax1, ax2), vminy = fmin(ay1, ay2), vmaxy = fmax(ay1, ay2);
                    if (vminx > exmax || vmaxx < exmin || vminy > eymax || vmaxy < eymin) continue;
                    const double a_ = ay2 - ay1, b_ = ax1 - ax2, c_ = ay1 * ax2 - ax1 * ay2;      // :477-479
                    const double det = a_ * e_ - b_ * d_;
                    if (det == 0) continue;
                    const double raw_x = (b_ * f_ - c_ * e_) / det;
                    const double raw_y = (c_ * d_ - a_ * f_) / det;
                    const bool cx_ = !(raw_x > exmax) && !(raw_x < exmin) && !(raw_x > vmaxx) && !(raw_x < vminx);
                    const bool cy_ = !(raw_y > eymax) && !(raw_y < eymin) && !(raw_y > vmaxy) && !(raw_y < vminy);
                    if (cx_ && cy_) bad = true;
                }
            }
        }
    }
    return bad;
}

// a wave-uniform double moved to scalar registers (the generator's per-segment state: five doubles that the compiler, which cannot
// see that LDS reads of a uniform address are uniform, kept in vector registers for the whole kernel)
__device__ __forceinline__ double uni_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// LDS ordering inside the ONE wave of a workgroup without draining the vector-memory queue (a __syncthreads() waits for vmcnt(0),
// which would end the overlap of the tile's global_load_lds with the first word's set-up): LDS operations of a wave complete in
// order, so the wave only has to have ISSUED its writes before the reads; the clobber keeps the compiler from moving them
__device__ __forceinline__ void lsync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- the SCREEN pass of k_rs_validate_f (round 5) ---------------------------------------------------------------------------
// 92 % of the searches end with every tested word invalid, and an invalid word's colliding samples come in long runs (the hull
// is 4.7 m long, the samples 0.1 m apart): of the invalid words of the bench workload 98 % have a colliding sample among every
// 8th of their first 128 (tools/rs_screen_study.py, CPU oracle).  So before the words are walked one by one, up to FOUR words are
// looked at in ONE pass: 64 / nw lanes per word, lane k of a word takes the sample nearest to arc position (k + 1) stride step
// (stride = 128 / lanes per word) and runs the float32 filter's CERTAIN-HIT test on it.  A word with a certain hit is invalid
// whatever else its samples do (is_traj_valid: any colliding sample, car_parking_base.py:452-534) and is skipped by the main loop;
// a word without one is walked as before, from its first sample -- the screen never accepts anything.
//
// The samples are produced in CLOSED FORM, lane-parallel, instead of by the sequential `pd += d` chain.  generate_local_course
// (reeds_shepp.py:452-507) walks segment i at pd = pd0_i + j d_i while |pd| <= |l_i|; with u = pd sign(l_i) (position along the
// segment's own direction), s = step and r_i = (first u beyond the segment) - |l_i| the code's `ll` bookkeeping gives
//     u0_0 = s,   u0_i = +r_(i-1) if l_(i-1) l_i > 0 (same direction), -r_(i-1) otherwise,   r_i = u0_i + n_i s - |l_i|,
// n_i = 0 if |u0_i| > |l_i|, else floor((|l_i| - u0_i) / s) + 1.  A closed form differs from the chain by ~1e-14 (64 roundings),
// nine orders below the float32 filter's error budget -- EXCEPT at a tie, where a sample within rounding of a segment end is in
// or out (n_i off by one).  Then r_i flips between ~0 and s, i.e. the next segment's lattice {u0 + j s} keeps its points and only
// gains or loses its FIRST one (u ~ 0: the same pose as the previous segment's end; or, across a direction change, the
// reference's off-path sample at u = -s).  The screen therefore only uses lattice points with SCREEN_EPS < u < |l_i| - SCREEN_EPS:
// every one of them is a sample of the reference whichever way the ties fall.
constexpr double SCREEN_EPS = 1e-7;
constexpr int SCREEN_SPAN = 128;         // samples of a word the screen spreads its lanes over

// one obstacle against one sample pose in the float32 frame: a CERTAIN hit on one of the four edges (the same classification,
// margins and robustness conditions as the main pass of k_rs_validate_f below: phase 1 "not clear", phase 2 "certain crossing")
__device__ __forceinline__ bool screen_obstacle_hit(const float4 v01, const float4 v23, float cx, float cy, float hc, float hs,
                                                    int efl, bool hull_ok) {
    constexpr float FEPS_ = FEPS, FKAPPA_ = FKAPPA, HL = F_HL, HW = F_HW;
    float u[4], w[4];
    {
        const float dx0 = v01.x - cx, dy0 = v01.y - cy, dx1 = v01.z - cx, dy1 = v01.w - cy;
        const float dx2 = v23.x - cx, dy2 = v23.y - cy, dx3 = v23.z - cx, dy3 = v23.w - cy;
        u[0] = hc * dx0 + hs * dy0; w[0] = hc * dy0 - hs * dx0;
        u[1] = hc * dx1 + hs * dy1; w[1] = hc * dy1 - hs * dx1;
        u[2] = hc * dx2 + hs * dy2; w[2] = hc * dy2 - hs * dx2;
        u[3] = hc * dx3 + hs * dy3; w[3] = hc * dy3 - hs * dx3;
    }
    const float umin = fminf(fminf(u[0], u[1]), fminf(u[2], u[3])), umax = fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3]));
    const float wmin = fminf(fminf(w[0], w[1]), fminf(w[2], w[3])), wmax = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
    if (umin > HL + FEPS_ || umax < -HL - FEPS_ || wmin > HW + FEPS_ || wmax < -HW - FEPS_) return false;
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = fmaxf(fabsf(u[k]) - HL, fabsf(w[k]) - HW);
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int k2 = (k + 1) & 3;
        const float nu = w[k2] - w[k], nw = u[k] - u[k2];
        const float dist = fabsf(__builtin_fmaf(nu, u[k], nw * w[k]));
        const float dn = dist - __builtin_fmaf(fabsf(nu), HL, fabsf(nw) * HW);
        const float eu0 = fminf(u[k], u[k2]), eu1 = fmaxf(u[k], u[k2]);
        const float ew0 = fminf(w[k], w[k2]), ew1 = fmaxf(w[k], w[k2]);
        const float esep = fmaxf(fmaxf(eu0 - HL, -HL - eu1), fmaxf(ew0 - HW, -HW - ew1));
        const float n1 = fabsf(nu) + fabsf(nw), mg = FEPS_ * n1;
        const bool clear = dn > mg || esep > FEPS_ || fmaxf(g[k], g[k2]) < -FEPS_;
        const bool cross = dn < -mg && esep < -FEPS_ && fminf(fabsf(g[k]), fabsf(g[k2])) > FEPS_ && fmaxf(g[k], g[k2]) > FEPS_;
        const float A = fabsf(nu) * HL, B = fabsf(nw) * HW;
        const bool corners_ok = fminf(fabsf(dn), fabsf(dist - fabsf(A - B))) > mg;
        const bool cross_u = fabsf(dist - A) <= B + mg, cross_w = fabsf(dist - B) <= A + mg;
        const bool angle_ok = (!cross_u || fabsf(nw) >= FKAPPA_ * n1) && (!cross_w || fabsf(nu) >= FKAPPA_ * n1);
        hit = hit || (!clear && cross && corners_ok && angle_ok && hull_ok && ((efl >> k) & 1));
    }
    return hit;
}

// the screen pass itself (shared by k_rs_screen and the one-kernel form of k_rs_validate_f): the obstacle view (fv / fbox / eflag) and
// the first four segment tables (qpd[0 .. 200)) are in LDS; returns the mask of condemned words (bit k: the k-th popped word)
template <bool TIMING>
__device__ __forceinline__ unsigned long long screen_words(int lane, int n_paths, int n_obst, const double* tables, double* qpd, int* pq, int* cand,
                                                           const float2* fv, const float4* fbox, const unsigned char* eflag, float fxmax,
                                                           float fymax, unsigned long long* tsec) {
    const float fxmin = 0.0f, fymin = 0.0f;
    const double step = RS_STEP * MAXC;
    unsigned long long condemned = 0;
    RS_T0();
    const double inv_step = 1.0 / step;
    for (int w0 = 0; w0 < n_paths; w0 += 4) {
        const int nw = min(4, n_paths - w0);
        // the chunk's segment tables: into the sample queue's LDS words (not in use yet)
        if (w0 > 0)                                        // (the first chunk's were requested in the prologue)
            for (int t = lane; t < RS_SEG_TABLE * nw; t += WAVE) qpd[t] = tables[RS_SEG_TABLE * w0 + t];
        lsync();
        RS_T(14);
        const int G = nw == 1 ? 64 : (nw == 2 ? 32 : (nw == 3 ? 21 : 16));         // lanes per word
        const int stride = SCREEN_SPAN / G;                                           // 2, 4, 6, 8 samples between two lanes
        const int wl_ = nw == 1 ? 0 : (nw == 2 ? lane >> 5 : (nw == 3 ? (lane >= 42 ? 2 : (lane >= 21 ? 1 : 0)) : lane >> 4));
        const int kk = lane - wl_ * G;
        const double* T = qpd + RS_SEG_TABLE * wl_;
        const double w7s = T[7];
        const int codew = __double2loint(w7s), nsegw = __double2hiint(w7s);
        // closed-form sample: the lattice point of the segment that holds arc position a, at or behind it
        const double a = (double)((kk + 1) * stride) * step;
        double c = 0.0, u0 = step, r = 0.0, lprev = 0.0, pdv = 0.0;
        int si = -1;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i < nsegw) {
                const double l = T[RS_SEGW * i + 6], al = fabs(l);
                if (i >= 1) u0 = (lprev * l > 0) ? r : -r;
                if (si < 0 && a < c + al) {
                    double j = ceil((a - c - u0) * inv_step);
                    j = j < 0.0 ? 0.0 : j;
                    const double u = u0 + j * step;
                    // (a segment shorter than the remainder a cusp carries into it, |u0| > |l|, has NO lattice samples at all --
                    // reeds_shepp.py:488, n_i = 0 below --: u = u0 + j step inside it would be a pose the reference never visits)
                    si = (fabs(u0) <= al && u > SCREEN_EPS && u < al - SCREEN_EPS) ? i : 5;   // 5: no usable sample for this lane
                    pdv = l > 0.0 ? u : -u;
                }
                const double n = fabs(u0) > al ? 0.0 : floor((al - u0) * inv_step) + 1.0;
                r = u0 + n * step - al;
                c += al;
                lprev = l;
            }
        }
        const bool active = lane < G * nw && si >= 0 && si < 5;
        float X = 0, Y = 0, hc = 1, hs = 0;
        if (active) {                                      // the pose, as in the main pass below
            const float lf = (float)pdv;
            const float4 row = ((const float4*)(T + RS_SEG_F32))[si];
            const int m = type_of(codew, si);
            const float rev = lf * 0.15915494309189535f;
            const float sl = m == TS ? 0.0f : __builtin_amdgcn_sinf(rev), cl = m == TS ? 1.0f : __builtin_amdgcn_cosf(rev);
            const float sgn = m == TR ? -1.0f : 1.0f;
            const float ldx = m == TS ? lf * F_INV_MAXC : sl * F_INV_MAXC;
            const float ldy = sgn * (1.0f - cl) * F_INV_MAXC;
            X = row.x + (row.z * ldx - row.w * ldy);
            Y = row.y + (row.w * ldx + row.z * ldy);
            const float ss = sgn * sl;
            hc = row.z * cl - row.w * ss;
            hs = row.w * cl + row.z * ss;
        }
        const bool oob = active && (X < fxmin - FEPS || X > fxmax + FEPS || Y < fymin - FEPS || Y > fymax + FEPS);
        const float cx = X + hc * F_MID, cy = Y + hs * F_MID;
        const float ex = fabsf(hc) * F_HL + fabsf(hs) * F_HW + FEPS, ey = fabsf(hs) * F_HL + fabsf(hc) * F_HW + FEPS;
        const float lox = cx - ex, hix = cx + ex, loy = cy - ey, hiy = cy + ey;
        const float ulox = wave_min_f(active ? lox : INFINITY), uhix = wave_max_f(active ? hix : -INFINITY);
        const float uloy = wave_min_f(active ? loy : INFINITY), uhiy = wave_max_f(active ? hiy : -INFINITY);
        int nc = 0;
        for (int base = 0; base < n_obst; base += WAVE) {
            const int o = base + lane;
            bool near = false;
            if (o < n_obst) {
                const float4 bb = fbox[o];
                near = !(bb.x > uhix || bb.y < ulox || bb.z > uhiy || bb.w < uloy);
            }
            const unsigned long long mm = __ballot(near);
            if (near) cand[nc + __popcll(mm & ((1ull << lane) - 1))] = o;
            nc += __popcll(mm);
        }
        RS_T(15);
        const int all_w = (1 << nw) - 1;
        int dead = 0;                                      // bit w: word w0 + w has a certain hit
        for (int w = 0; w < nw; w++) dead |= __ballot(oob && wl_ == w) ? (1 << w) : 0;
        // The obstacles near the union of the samples' hulls are few for any ONE sample but many for the wave (four words leave
        // the start pose in different directions): a loop over the candidates with one sample per lane kept 60 of 64 lanes idle
        // per obstacle (28 k cycles per search in the first version).  So the (sample, obstacle) pairs whose boxes meet are
        // queued and classified 64 pairs at a time, every lane busy; the sample's pose travels by ds_bpermute.
        auto classify = [&](int count) {                   // pairs pq[0 .. count), count <= 64
            const bool has = lane < count;
            const int pr = has ? pq[lane] : 0;
            const int sl_ = pr & 63, ro = pr >> 6;
            const float scx = __shfl(cx, sl_), scy = __shfl(cy, sl_), shc = __shfl(hc, sl_), shs = __shfl(hs, sl_);
            const int sw = __shfl(wl_, sl_);
            bool h = false;
            if (has && !((dead >> sw) & 1)) {
                const float4 v01 = ((const float4*)fv)[2 * ro], v23 = ((const float4*)fv)[2 * ro + 1];
                h = screen_obstacle_hit(v01, v23, scx, scy, shc, shs, (int)eflag[ro], fminf(fabsf(shc), fabsf(shs)) >= FETA_HULL);
            }
            for (int w = 0; w < nw; w++) dead |= __ballot(h && sw == w) ? (1 << w) : 0;
        };
        if (nc > 0 && dead != all_w) {
            lsync();
            int npq = 0;
            for (int ci = 0; ci < nc; ci++) {
                const int ro = cand[ci];
                const float4 bb = fbox[ro];
                const bool near = active && !oob && !((dead >> wl_) & 1) && !(bb.x > hix || bb.y < lox || bb.z > hiy || bb.w < loy);
                const unsigned long long nm = __ballot(near);
                if (!nm) continue;
                if (near) pq[npq + __popcll(nm & ((1ull << lane) - 1))] = (ro << 6) | lane;
                npq += __popcll(nm);
                if (npq >= WAVE) {
                    lsync();
                    classify(WAVE);
                    if (dead == all_w) { npq = 0; break; }     // every word of the chunk is condemned
                    const int rem = npq - WAVE;               // carry the tail to the front of the queue
                    int cv = 0;
                    if (lane < rem) cv = pq[WAVE + lane];
                    lsync();
                    if (lane < rem) pq[lane] = cv;
                    npq = rem;
                }
            }
            if (npq > 0 && dead != all_w) { lsync(); classify(npq); }
        }
        condemned |= (unsigned long long)dead << w0;
        lsync();                                           // the next chunk's tables / the first word's queue overwrite these words
        RS_T(7);
    }
    return condemned;
}

// ================================================================================================
// Kernel B0 (round 5, HOPE_RS_SPLIT=1 only): the screen pass as a kernel of its own -- 8 waves per SIMD instead of the walk's 4
// ================================================================================================
// Four searches out of five end at the screen (every tested word condemned), and what the screen needs is small: the float32
// obstacle view, four segment tables, a pair queue.  In the one-kernel form those searches occupied a 125-register wave slot of
// k_rs_validate_f for their whole life -- most of it waiting for two memory round trips.  Here they run in a kernel that fits
// 64 registers; it writes the mask of condemned words into the record's header and queues the searches that still have a word
// to walk (queue index, queue entry) for k_rs_validate_f<..., PRE = true>, which then starts only for those.
template <bool TIMING, int OCC = 8>
__global__ __launch_bounds__(64, OCC) void k_rs_screen(RsParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    unsigned long long tsec[16] = {};
    const int b_ = (int)blockIdx.x;
    const int qidx = scene_of_block(b_, p.max_queue);     // (a bijection of the launch's blocks: independent of the queue length)
    const int entry = p.rs_list[qidx];
    const int count = *p.rs_count;
    if (qidx >= count) return;
    const int slot = p.slot_base + p.slot_dir * qidx;
    const int scene = rs_list_scene(entry), n_obst = rs_list_n_obst(entry);
    // LDS: float2 V[4 cap] | float4 box[cap] | four segment tables [200] | pair queue int[128] | int cand[cap] | edge flags[cap]
    float2* fv = (float2*)lds;
    float4* fbox = (float4*)(lds + 4 * p.tile_cap);
    double* tabs = lds + 6 * p.tile_cap;
    int* pq = (int*)(tabs + 4 * RS_SEG_TABLE);
    int* cand = pq + 2 * WAVE;
    unsigned char* eflag = (unsigned char*)(cand + ((p.tile_cap + 3) & ~3));
    double* rec = p.rs_rec + (size_t)slot * RS_REC_DOUBLES;
    const double r0 = lane < RS_REC_HDR ? rec[lane] : 0.0;
    const double* tables = rec + RS_REC_SEGS;
    {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const double2* t2 = (const double2*)tables;
        __builtin_amdgcn_global_load_lds((const void*)(t2 + lane), (lds_ptr)tabs, 16, 0, 0);
        if (lane < 2 * RS_SEG_TABLE - WAVE) __builtin_amdgcn_global_load_lds((const void*)(t2 + WAVE + lane), (lds_ptr)(tabs + 2 * WAVE), 16, 0, 0);
        const float4* gfv = p.fverts + (size_t)scene * p.max_obst * 2;
        const float4* gfb = p.fbox + (size_t)scene * p.max_obst;
        const uint32_t* gfl = (const uint32_t*)(p.eflag + (size_t)scene * eflag_stride(p.max_obst));
        float4* lfv = (float4*)fv;
        for (int base = 0; base < 2 * n_obst; base += WAVE)
            if (base + lane < 2 * n_obst) __builtin_amdgcn_global_load_lds((const void*)(gfv + base + lane), (lds_ptr)(lfv + base), 16, 0, 0);
        for (int base = 0; base < n_obst; base += WAVE)
            if (base + lane < n_obst) __builtin_amdgcn_global_load_lds((const void*)(gfb + base + lane), (lds_ptr)(fbox + base), 16, 0, 0);
        for (int base = 0; 4 * base < n_obst; base += WAVE)
            if (4 * (base + lane) < n_obst) __builtin_amdgcn_global_load_lds((const void*)(gfl + base + lane), (lds_ptr)((uint32_t*)eflag + base), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // header, tables and obstacle view: one round trip (and no load into LDS in flight at an exit)
    const int n_paths = __builtin_amdgcn_readlane(__double2hiint(r0), 1);
    if (n_paths == 0) return;
    const float fxmax = (float)(readlane_d(r0, 6) - readlane_d(r0, 5)), fymax = (float)(readlane_d(r0, 8) - readlane_d(r0, 7));
    lsync();
    const unsigned long long condemned = screen_words<TIMING>(lane, n_paths, n_obst, tables, tabs, pq, cand, fv, fbox, eflag, fxmax, fymax, tsec);
    if (lane == 0) {
        ((unsigned long long*)rec)[9] = condemned;
        if (condemned != ((1ull << n_paths) - 1)) p.surv_list[atomicAdd(p.surv_count, 1)] = make_int2(qidx, entry);
    }
}

template <int OCC, bool TIMING, bool STATS, bool PRE = false>
__global__ __launch_bounds__(64, OCC) void k_rs_validate_f(RsParams p, int obs_f64) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    // Raised wave priority: this kernel ends the longer launch chain -- the step's critical path -- and shares the CUs with the
    // observation half of k_env_step, which has slack.  Measured at 65 536 scenes: 0.672 -> 0.656 ms per step (priority 1, 2 and
    // 3 alike); the same on k_rs_words costs 9 %, on k_rs_segs 1 %.  (flag bit 0x10000 = default priority, HOPE_RS_PRIO=0)
    if (!(obs_f64 & 0x10000)) __builtin_amdgcn_s_setprio(1);
    unsigned long long tsec[16] = {};
    const unsigned long long tstart_ = TIMING ? __builtin_readcyclecounter() : 0;
    RS_T0();
    const int b_ = (int)blockIdx.x;
    // Two memory round trips in front of the first sample instead of three (round 5): the block -> queue entry map does not depend on
    // the queue length (a bijection of the launch's max_queue blocks; the length only decides who leaves), and the queue entry itself
    // carries the scene AND its obstacle count (k_rs_compact), so that the record's header, the segment tables and the obstacle view
    // are all requested at once -- the header used to be a round trip of its own in front of the tile.
    int qidx, entry;
    if (PRE) {                                             // two-kernel form: the searches k_rs_screen left a word of
        const int2 sv = p.surv_list[b_];                   // (requested before the queue length is known: stale entries are harmless)
        if (b_ >= *p.surv_count) return;
        qidx = sv.x; entry = sv.y;
    } else {
#ifdef HOPE_RS_3TRIPS                                      // (A/B build: the round-4 order -- queue length, then header, then tile)
        const int count = *p.rs_count;
        if (b_ >= count) return;
        qidx = scene_of_block(b_, count);
        entry = p.rs_list[qidx];
#else
        qidx = scene_of_block(b_, p.max_queue);
        entry = p.rs_list[qidx];                           // (entries beyond the queue length are stale or zero: valid scenes, and we leave)
        const int count = *p.rs_count;
        if (qidx >= count) return;
#endif
    }
    const int slot = p.slot_base + p.slot_dir * qidx;
#ifndef HOPE_RS_HDR
    const int scene = rs_list_scene(entry);
    const int n_obst = rs_list_n_obst(entry);
#endif
    // LDS: float2 V[4 cap] | float4 box[cap] | float64 tile 8 cap + world boxes float4[cap] (filled when the first pass needs the
    //      float64 arithmetic) | scratch doubles: segment table 50, sample queue pd[256], bad1[8] | int cand[cap] | bytes: qseg[256],
    //      edge flags[cap]
    float2* fv = (float2*)lds;
    float4* fbox = (float4*)(lds + 4 * p.tile_cap);
    double* scr = lds + 6 * p.tile_cap;            // (the float64 re-evaluation reads its few obstacle edges from global memory)
    double* segp = scr + RSB_SEG;
    double* qpd = scr + RSB_QPD;
    int* cand = (int*)(scr + RSB_WORDS_F);
    double* tabl = scr + RSB_TABF;                 // first-segment samples T[0 .. 256) (a copy per wave: four registers per lane were
                                                   // the difference between 128 VGPRs and spilling)
    unsigned char* qseg = (unsigned char*)(cand + ((p.tile_cap + 3) & ~3));
    unsigned char* eflag = qseg + RSB_QCAP;

    const double* rec = p.rs_rec + (size_t)slot * RS_REC_DOUBLES;
    const double r0 = lane < RS_REC_HDR ? rec[lane] : 0.0;
    const double* tables = rec + RS_REC_SEGS;
#ifdef HOPE_RS_3TRIPS
    if (__builtin_amdgcn_readlane(__double2hiint(r0), 1) == 0) return;        // (waits for the header, as round 4 did)
#endif
#ifdef HOPE_RS_HDR                                         // (bisecting build: scene / obstacle count from the record's header)
    const int scene = __builtin_amdgcn_readlane(__double2loint(r0), 0);
    const int n_obst = __builtin_amdgcn_readlane(__double2hiint(r0), 0);
    (void)entry;
#endif
    // the segment tables of the first FOUR words lie back to back behind the header (200 doubles; a record has room for 48 tables, so
    // the request never leaves it): all of them now, for the screen pass ...
    double tb = lane < RS_SEG_TABLE ? tables[lane] : 0.0;
    if (!PRE) {   // ... straight into the sample queue's LDS words (global_load_lds, 16 bytes per lane: no registers, nobody waits here)
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const double2* t2 = (const double2*)tables;
        __builtin_amdgcn_global_load_lds((const void*)(t2 + lane), (lds_ptr)qpd, 16, 0, 0);
        if (lane < 2 * RS_SEG_TABLE - WAVE) __builtin_amdgcn_global_load_lds((const void*)(t2 + WAVE + lane), (lds_ptr)(qpd + 2 * WAVE), 16, 0, 0);
    }
    // The float32 view of the obstacles -- vertices in that frame, their boxes, the edges' robustness flags -- was made when the
    // scene got its map (obstacle_f32, hope_dev.h).  It goes from global memory STRAIGHT INTO LDS (global_load_lds: no staging
    // registers, and the wave does not wait): the first word's setup and its first samples are generated while the tile is in
    // flight; `tile_wait()` below is the one place that waits for it.  (Rounds 2-3 converted the float64 tile here, once per SEARCH,
    // with the wave stalled on the loads: 26 % of the kernel's cycles.)
    {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const float4* gfv = p.fverts + (size_t)scene * p.max_obst * 2;
        const float4* gfb = p.fbox + (size_t)scene * p.max_obst;
        const uint32_t* gfl = (const uint32_t*)(p.eflag + (size_t)scene * eflag_stride(p.max_obst));
        float4* lfv = (float4*)fv;
        for (int base = 0; base < 2 * n_obst; base += WAVE)
            if (base + lane < 2 * n_obst) __builtin_amdgcn_global_load_lds((const void*)(gfv + base + lane), (lds_ptr)(lfv + base), 16, 0, 0);
        for (int base = 0; base < n_obst; base += WAVE)
            if (base + lane < n_obst) __builtin_amdgcn_global_load_lds((const void*)(gfb + base + lane), (lds_ptr)(fbox + base), 16, 0, 0);
        for (int base = 0; 4 * base < n_obst; base += WAVE)
            if (4 * (base + lane) < n_obst) __builtin_amdgcn_global_load_lds((const void*)(gfl + base + lane), (lds_ptr)((uint32_t*)eflag + base), 4, 0, 0);
    }
    // (everything that READS the header comes behind the requests above: a use of r0 makes the wave wait for it)
    const int n_paths = __builtin_amdgcn_readlane(__double2hiint(r0), 1);      // words the stop rule lets the search test
    if (n_paths == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a wave must not end with loads into its LDS in flight)
        return;
    }
    const double* verts_g = p.verts + (size_t)scene * p.max_obst * 8;
    // map box in the frame of the scene's float32 obstacle view (origin = its lower left corner): the rear axle must stay inside (:462-464)
    const float fxmin = 0.0f, fxmax = (float)(readlane_d(r0, 6) - readlane_d(r0, 5)), fymin = 0.0f, fymax = (float)(readlane_d(r0, 8) - readlane_d(r0, 7));
    const double step = RS_STEP * MAXC;
    bool tile_pending = true;
    int found = -1;
    double* bad1 = scr + RSB_BAD;
    if (lane < 6) bad1[lane] = INFINITY;
    lsync();
    RS_T(0);
    const bool paranoid = (obs_f64 & 0x2000) != 0;        // self-check: float64 for every sample, disagreements counted
    unsigned long long st_pass = 0, st_hit = 0, st_exact = 0, st_unc = 0, st_bad_hit = 0, st_bad_clear = 0, st_samples = 0;
    unsigned long long st_why[5] = {};
    // ---- screen pass (see screen_obstacle_hit above): up to four words per pass, certain float32 hits only ----
    unsigned long long condemned = 0;                      // bit k: the k-th popped word has a certainly colliding sample
    if (PRE) condemned = (unsigned long long)__double_as_longlong(readlane_d(r0, 9));      // k_rs_screen's verdict (nothing of ours in LDS yet: no wait)
    else if (!(obs_f64 & 0x20000)) {                       // (HOPE_RS_DEBUG=0x20000: no screen -- A/B and the parity tests)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the obstacle view's global_load_lds have landed
        tile_pending = false;
        condemned = screen_words<TIMING>(lane, n_paths, n_obst, tables, qpd, (int*)tabl /* pair queue: the sample table is loaded behind the screen */,
                                         cand, fv, fbox, eflag, fxmax, fymax, tsec);
        if (TIMING) t0_ = __builtin_readcyclecounter();
    }
    else {                                                 // (no screen: the tables' loads must have landed before the queue's words are used)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile_pending = false;
    }
    const bool verify_screen = STATS && paranoid;          // self-check build: condemned words are walked anyway and must come out invalid
    const bool all_condemned = condemned == ((1ull << n_paths) - 1);
    unsigned long long st_scr_words = __popcll(condemned), st_scr_dead = all_condemned ? 1 : 0, st_scr_bad = 0;
    if (!all_condemned || verify_screen) {
        // first-segment samples into LDS, for the words that are walked (it was part of the prologue; four searches out of five
        // now end at the screen and never need it)
        const double T0 = g_rs_steps[lane], T1 = g_rs_steps[WAVE + lane], T2 = g_rs_steps[2 * WAVE + lane], T3 = g_rs_steps[3 * WAVE + lane];
        tabl[lane] = T0; tabl[WAVE + lane] = T1; tabl[2 * WAVE + lane] = T2; tabl[3 * WAVE + lane] = T3;
        lsync();
    }
    for (int idx = 1; idx <= n_paths; idx++) {
        if (all_condemned && !verify_screen) break;        // nothing left to walk: no path
        const double tcur = tb;
        if (idx < n_paths && lane < RS_SEG_TABLE) tb = tables[RS_SEG_TABLE * idx + lane];      // prefetch the next word's
        const double len0 = readlane_d(tcur, 6), w6 = readlane_d(tcur, 7);
        const int code = __double2loint(w6), nseg = __double2hiint(w6);
        const int cls1 = type_of(code, 0) * 2 + (len0 > 0.0 ? 1 : 0);
        if (fabs(len0) >= bad1[cls1]) continue;           // contains a sample already known to collide
        const bool screened = (condemned >> (idx - 1)) & 1;
        if (screened && !verify_screen) continue;         // the screen found a certainly colliding sample of this word
        bool invalid = false;
        if (lane < RS_SEG_TABLE) segp[lane] = tcur;
        if (TIMING) tsec[9] += 1;
        {   // (a zero made here: as a loop invariant the allocator kept it in SCRATCH and reloaded it for every word)
            double zero;
            asm volatile("v_mov_b64 %0, 0" : "=v"(zero));
            if (lane == 0) { qpd[0] = zero; qseg[0] = (unsigned char)__double2loint(zero); }
        }
        int nq = 1;
        lsync();
        RS_T(1);
        const int win = 128;
        int i = 0, t_off = 0;                              // t_off: first-segment samples taken from the table so far
        bool seg_open = false, finished = false;
        double pd = 0, ll = 0.0, lprev = 0.0, d = 0, l = 0;
        while (!invalid && (!finished || nq > 0)) {
            while (!finished && nq + WAVE + 1 <= win) {
                if (TIMING) tsec[10] += 1;
                if (!seg_open) {
                    l = uni_d(segp[RSB_SEGW * i + 6]);
                    d = uni_d(l > 0.0 ? step : -step);
                    pd = uni_d((i >= 1 && (lprev * l) > 0) ? -d - ll : d - ll);
                    lprev = l;
                    seg_open = true;
                }
                double mine, t = pd;
                int ncap;
                if (i == 0 && t_off + WAVE <= RS_TAB) {
                    // first segment: pd = d, d + d, ... is the same chain for every word of every search -> table
                    const double v = tabl[t_off + lane];
                    mine = d > 0.0 ? v : -v;
                    ncap = WAVE;
                    t_off += WAVE;
                    if (t_off < RS_TAB) {                         // the chain's next value: the first entry of the next chunk
                        const double nx = tabl[t_off];
                        t = d > 0.0 ? nx : -nx;
                    } else t = readlane_d(mine, WAVE - 1) + d;
                } else {
                    // `pd += d` chain (sequential rounding kept), see k_rs_validate
                    mine = pd;
                    double last = pd;
                    ncap = 0;
                    for (int j0 = 0; j0 < WAVE; j0 += 8) {
                        if ((lane >> 3) == (j0 >> 3)) mine = t;
#pragma unroll
                        for (int jj = 0; jj < 7; jj++) t = t + d;
                        last = t;
                        t = t + d;
                        ncap = j0 + 8;
                        if (__any(fabs(last) > fabs(l))) break;
                    }
#pragma unroll
                    for (int k = 0; k < 7; k++)
                        if (k < (lane & 7)) mine = mine + d;
                }
                const bool in_seg = lane < ncap && fabs(mine) <= fabs(l);
                const unsigned long long valid = (ncap == WAVE) ? ~0ull : ((1ull << ncap) - 1);
                const unsigned long long fail = ~__ballot(in_seg) & valid;
                const int cnt = fail ? (__ffsll((long long)fail) - 1) : ncap;
                if (lane < cnt) { qpd[nq + lane] = mine; qseg[nq + lane] = (unsigned char)i; }
                nq += cnt;
                if (fail) {
                    pd = readlane_d(mine, cnt);
                    ll = uni_d(l - pd - d);
                    seg_open = false;
                    if (i == nseg - 1) {
                        if (lane == 0) { qpd[nq] = l; qseg[nq] = (unsigned char)i; }
                        nq += 1;
                        finished = true;
                    }
                    i++;
                } else pd = uni_d(t);
            }
            lsync();
            RS_T(2);
            const int n_all = nq;
            const int n = finished ? n_all : (n_all & ~(WAVE - 1));
            const int stride = (n + WAVE - 1) >> 6;                  // 1 .. 3 (n <= 192): no integer divisions below
            const int n_coarse = stride == 1 ? n : (stride == 2 ? (n + 1) >> 1 : ((n + 2) * 171) >> 9);      // ceil(n / stride)
            const int n_rest = n - n_coarse;
            for (int rnd = 0; rnd == 0 || (rnd - 1) * WAVE < n_rest; rnd++) {
                int sidx;
                if (rnd == 0) sidx = (lane < n_coarse) ? stride * lane : -1;
                else { const int r = (rnd - 1) * WAVE + lane; sidx = r < n_rest ? r + (stride == 2 ? r : r >> 1) + 1 : -1; }
                const bool active = sidx >= 0;
                if (TIMING) tsec[11] += 1;
                // ---- float32 pose in the start-position frame ----
                float X = 0, Y = 0, hc = 1, hs = 0;
                int sg0 = 0;
                if (active) {
                    const float lf = (float)qpd[sidx];
                    sg0 = (int)qseg[sidx];
                    const float4 row = ((const float4*)(segp + RS_SEG_F32))[sg0];        // Ox, Oy, cos, sin of the world heading
                    const int m = type_of(code, sg0);
                    const float rev = lf * 0.15915494309189535f;
                    const float sl = m == TS ? 0.0f : __builtin_amdgcn_sinf(rev), cl = m == TS ? 1.0f : __builtin_amdgcn_cosf(rev);
                    const float sgn = m == TR ? -1.0f : 1.0f;
                    const float ldx = m == TS ? lf * F_INV_MAXC : sl * F_INV_MAXC;
                    const float ldy = sgn * (1.0f - cl) * F_INV_MAXC;
                    X = row.x + (row.z * ldx - row.w * ldy);
                    Y = row.y + (row.w * ldx + row.z * ldy);
                    const float ss = sgn * sl;                                               // heading + l (L), - l (R)
                    hc = row.z * cl - row.w * ss;
                    hs = row.w * cl + row.z * ss;
                }
                RS_T(3);
                // out of the map box: decided / undecided with margin
                int why_cnt = 0, hit_rec = -1;
                int ua = -1, ub = -1;                              // undecided edges of this lane: obstacle | edge mask << 8, two obstacles
                bool uover = false;                                // ... more than two: the generic float64 pass
                bool hit = active && (X < fxmin - FEPS || X > fxmax + FEPS || Y < fymin - FEPS || Y > fymax + FEPS);
                bool unc = active && !hit && (X < fxmin + FEPS || X > fxmax - FEPS || Y < fymin + FEPS || Y > fymax - FEPS);
                // hull box (inflated by FEPS): centre + |cos|, |sin| extents
                const float cx = X + hc * F_MID, cy = Y + hs * F_MID;
                const float ex = fabsf(hc) * F_HL + fabsf(hs) * F_HW + FEPS, ey = fabsf(hs) * F_HL + fabsf(hc) * F_HW + FEPS;
                const float lox = cx - ex, hix = cx + ex, loy = cy - ey, hiy = cy + ey;
                const float ulox = wave_min_f(active ? lox : INFINITY), uhix = wave_max_f(active ? hix : -INFINITY);
                const float uloy = wave_min_f(active ? loy : INFINITY), uhiy = wave_max_f(active ? hiy : -INFINITY);
                if (tile_pending) {                                  // first look at the obstacles: the tile's loads must have landed
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    tile_pending = false;
                }
                int nc = 0;
                for (int base = 0; base < n_obst; base += WAVE) {
                    const int o = base + lane;
                    bool near = false;
                    if (o < n_obst) {
                        const float4 bb = fbox[o];
                        near = !(bb.x > uhix || bb.y < ulox || bb.z > uhiy || bb.w < uloy);
                    }
                    const unsigned long long mm = __ballot(near);
                    if (near) cand[nc + __popcll(mm & ((1ull << lane) - 1))] = o;
                    nc += __popcll(mm);
                }
                RS_T(4);
                if (nc > 0 && !__any(hit)) {
                    if (TIMING) { tsec[12] += 1; tsec[13] += nc; }
                    lsync();
                    const bool hull_ok = fminf(fabsf(hc), fabsf(hs)) >= FETA_HULL;      // no hull edge axis-parallel in the world
                    for (int ci = 0; ci < nc; ci++) {
                        const int r = cand[ci];
                        const float4 bb = fbox[r];
                        const bool near = active && !(bb.x > hix || bb.y < lox || bb.z > hiy || bb.w < loy);
                        if (!__any(near)) continue;
                        // the four vertices in the car frame (origin = hull centre)
                        const float4 v01 = ((const float4*)fv)[2 * r], v23 = ((const float4*)fv)[2 * r + 1];
                        float u[4], w[4];
                        {
                            const float dx0 = v01.x - cx, dy0 = v01.y - cy, dx1 = v01.z - cx, dy1 = v01.w - cy;
                            const float dx2 = v23.x - cx, dy2 = v23.y - cy, dx3 = v23.z - cx, dy3 = v23.w - cy;
                            u[0] = hc * dx0 + hs * dy0; w[0] = hc * dy0 - hs * dx0;
                            u[1] = hc * dx1 + hs * dy1; w[1] = hc * dy1 - hs * dx1;
                            u[2] = hc * dx2 + hs * dy2; w[2] = hc * dy2 - hs * dx2;
                            u[3] = hc * dx3 + hs * dy3; w[3] = hc * dy3 - hs * dx3;
                        }
                        const float umin = fminf(fminf(u[0], u[1]), fminf(u[2], u[3])), umax = fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3]));
                        const float wmin = fminf(fminf(w[0], w[1]), fminf(w[2], w[3])), wmax = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
                        // whole obstacle beyond one side of the (inflated) rectangle: clear
                        const bool maybe = near && !(umin > F_HL + FEPS || umax < -F_HL - FEPS || wmin > F_HW + FEPS || wmax < -F_HW - FEPS);
                        if (!__any(maybe)) continue;
                        // phase 1, branch-free: which edges are NOT certainly clear of the hull
                        const int efl = (int)eflag[r];
                        float g[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) g[k] = fmaxf(fabsf(u[k]) - F_HL, fabsf(w[k]) - F_HW);    // > 0 outside
                        float nu[4], nw[4], dn[4], esep[4];        // edge normal, |C| - (A + B), separation on the rectangle's axes
                        float dist[4];
                        int open = 0;                              // bit k: edge k undecided so far
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int k2 = (k + 1) & 3;
                            nu[k] = w[k2] - w[k]; nw[k] = u[k] - u[k2];                       // normal of the edge (not normalised)
                            dist[k] = fabsf(__builtin_fmaf(nu[k], u[k], nw[k] * w[k]));       // |n . P|: distance of the edge's LINE from the centre
                            dn[k] = dist[k] - __builtin_fmaf(fabsf(nu[k]), F_HL, fabsf(nw[k]) * F_HW);   // minus the rectangle's radius along n
                            const float eu0 = fminf(u[k], u[k2]), eu1 = fmaxf(u[k], u[k2]);
                            const float ew0 = fminf(w[k], w[k2]), ew1 = fmaxf(w[k], w[k2]);
                            esep[k] = fmaxf(fmaxf(eu0 - F_HL, -F_HL - eu1), fmaxf(ew0 - F_HW, -F_HW - ew1));
                            const float mg = FEPS * (fabsf(nu[k]) + fabsf(nw[k]));
                            // separated on the edge's normal or on a rectangle axis (inflated), or wholly inside (deflated)
                            const bool clear = dn[k] > mg || esep[k] > FEPS || fmaxf(g[k], g[k2]) < -FEPS;
                            open |= (maybe && !clear) ? (1 << k) : 0;
                        }
                        // phase 2, only for edges some lane still has open: a certain AND robust crossing, or undecided
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (!__any((open >> k) & 1)) continue;
                            const int k2 = (k + 1) & 3;
                            const float n1 = fabsf(nu[k]) + fabsf(nw[k]), mg = FEPS * n1;
                            // meets the deflated rectangle, both end points off the boundary, one of them outside
                            const bool cross = dn[k] < -mg && esep[k] < -FEPS && fminf(fabsf(g[k]), fabsf(g[k2])) > FEPS && fmaxf(g[k], g[k2]) > FEPS;
                            // Robustness for the tolerance-free reference test.  With A = |nu| HL, B = |nw| HW the edge's line n . p = C is at
                            // | |C| - (A + B) | and | |C| - |A - B| | (times 1 / |n|) from the nearest hull corners, it crosses a side u = +-HL iff
                            // | |C| - A | <= B and a side w = +-HW iff | |C| - B | <= A (margin added: "may cross").
                            const float A = fabsf(nu[k]) * F_HL, B = fabsf(nw[k]) * F_HW;
                            const bool corners_ok = fminf(fabsf(dn[k]), fabsf(dist[k] - fabsf(A - B))) > mg;
                            const bool cross_u = fabsf(dist[k] - A) <= B + mg, cross_w = fabsf(dist[k] - B) <= A + mg;
                            const bool angle_ok = (!cross_u || fabsf(nw[k]) >= FKAPPA * n1) && (!cross_w || fabsf(nu[k]) >= FKAPPA * n1);
                            const bool certain = cross && corners_ok && angle_ok && hull_ok && ((efl >> k) & 1);
                            const bool mine_open = (open >> k) & 1;
                            if (mine_open && certain) { hit = true; if (STATS && hit_rec < 0) hit_rec = r * 4 + k; }
                            if (STATS && mine_open && !certain) why_cnt |= 1 << (!cross ? 0 : !((efl >> k) & 1) ? 1 : !hull_ok ? 2 : !corners_ok ? 3 : 4);
                            // (the record of the undecided edges only matters while no sample of the pass is a certain hit)
                            if (!__any(hit) && mine_open && !certain) {
                                unc = true;
                                if (ua < 0) ua = r | (0x100 << k);
                                else if ((ua & 0xff) == r) ua |= 0x100 << k;
                                else if (ub < 0) ub = r | (0x100 << k);
                                else if ((ub & 0xff) == r) ub |= 0x100 << k;
                                else uover = true;
                            }
                        }
                        if (__any(hit)) break;                   // one certain collision condemns the word
                    }
                    lsync();
                }
                RS_T(5);
                // ---- decision of the pass ----
                bool bad = hit;
                const bool any_hit = __any(hit);
                const bool need64 = (STATS && paranoid) || (!any_hit && __any(unc));
                (void)need64;
                if (STATS) {
                    st_pass += 1;
                    if (any_hit) st_hit += 1;
                    if (!any_hit && __any(unc)) {
                        const unsigned long long wm = __ballot(why_cnt != 0);
                        int wor = 0;
                        for (int b = 0; b < 5; b++) if (__any((why_cnt >> b) & 1)) wor |= 1 << b;
                        for (int b = 0; b < 5; b++) st_why[b] += (wor >> b) & 1;
                        (void)wm;
                    }
                }
                if (STATS && paranoid) {                         // float64 for every sample; compare with the float32 verdicts
                    const double q0x = rec[2], q0y = rec[3], q0w = rec[4], xmin = rec[5], xmax = rec[6], ymin = rec[7], ymax = rec[8];
                    const double c_q = rec[RS_REC_SEGS + RS_SEGW + 7], s_q = rec[RS_REC_SEGS + 2 * RS_SEGW + 7];
                    const bool ex = exact_pass<true>(active ? sidx : -1, segp, qpd, qseg, c_q, s_q, q0x, q0y, q0w, verts_g, p.obb + (size_t)scene * p.max_obst, cand, n_obst, xmin, xmax, ymin, ymax, lane);
                    st_samples += __popcll(__ballot(active));
                    st_bad_hit += __popcll(__ballot(active && hit && !ex));
                    if (active && ((hit && !ex) || (!any_hit && !hit && !unc && ex))) {
                        const int di = (int)atomicAdd(&g_rs_fstat[15], 1ull);
                        if (di < 64) {
                            double px = 0, py = 0, pyaw = 0;
                            const double spd = qpd[sidx];
                            const double* sp_ = segp + RSB_SEGW * (int)qseg[sidx];
                            interpolate(spd, (int)sp_[5], sp_[0], sp_[1], sp_[2], sp_[3], -sp_[4], sp_[3], sp_[4], px, py, pyaw);
                            const double wyaw = pi_2_pi(pyaw + q0w);
                            double* dd = g_rs_fdump + 16 * di;
                            dd[0] = scene; dd[1] = idx; dd[2] = spd; dd[3] = (double)qseg[sidx]; dd[4] = X; dd[5] = Y; dd[6] = hc; dd[7] = hs;
                            dd[8] = c_q * px + s_q * py; dd[9] = -s_q * px + c_q * py; dd[10] = hm_cos(wyaw); dd[11] = hm_sin(wyaw);
                            dd[12] = hit ? 1.0 : 0.0; dd[13] = (double)why_cnt; dd[14] = sp_[5]; dd[15] = (double)hit_rec;
                        }
                    }
                    // (a pass stops at its first certain hit: samples it did not finish looking at have no verdict)
                    if (!any_hit) st_bad_clear += __popcll(__ballot(active && !hit && !unc && ex));
                    bad = ex;
                } else if (!any_hit && __any(unc)) {
                    if (STATS) { st_exact += 1; st_unc += __popcll(__ballot(unc)); }
                    bad = exact_edges(unc, sidx, ua, ub, uover, cand, nc, segp, qpd, qseg, verts_g, rec);
                }
                if (TIMING) t0_ = __builtin_readcyclecounter();
                if (__any(bad)) {
                    const double mine1 = (bad && active && qseg[sidx] == 0) ? fabs(qpd[sidx]) : INFINITY;
                    const double v = fmin(bad1[cls1], wave_min_d(mine1));
                    lsync();
                    if (lane < 6 && (lane == cls1 || v == 0.0)) bad1[lane] = fmin(bad1[lane], v);
                    invalid = true;
                    break;
                }
            }
            {   // carry the untested tail to the front of the queue
                const int rem = n_all - n;
                double cd = 0;
                unsigned char cs = 0;
                if (lane < rem) { cd = qpd[n + lane]; cs = qseg[n + lane]; }
                lsync();
                if (lane < rem) { qpd[lane] = cd; qseg[lane] = cs; }
                nq = rem;
            }
            lsync();
        }
        if (STATS && screened && !invalid) { st_scr_bad += 1; continue; }   // (must never happen; the product build never gets here)
        if (!invalid) { found = idx - 1; break; }
    }
    if (STATS && lane == 0) {                              // statistics of the filter (tools/rs_filter_stats.py, the soak test)
        atomicAdd(&g_rs_fstat[7], st_scr_bad); atomicAdd(&g_rs_fstat[13], st_scr_words); atomicAdd(&g_rs_fstat[14], st_scr_dead);
        for (int b = 0; b < 5; b++) atomicAdd(&g_rs_fstat[8 + b], st_why[b]);
        atomicAdd(&g_rs_fstat[0], st_pass); atomicAdd(&g_rs_fstat[1], st_hit); atomicAdd(&g_rs_fstat[2], st_exact);
        atomicAdd(&g_rs_fstat[3], st_unc); atomicAdd(&g_rs_fstat[4], st_bad_hit); atomicAdd(&g_rs_fstat[5], st_bad_clear);
        atomicAdd(&g_rs_fstat[6], st_samples);
    }
    if (TIMING) {
        tsec[6] = __builtin_readcyclecounter() - tstart_;
        tsec[8] = 1;
        if (lane == 0) {
            for (int i = 0; i < 16; i++) if (tsec[i]) atomicAdd(&g_rs_prof[(blockIdx.x & 63) * 16 + i], tsec[i]);
            const int li = atomicAdd(&g_rs_log_n, 1);
            if (li < RS_LOG_CAP) g_rs_log[li] = make_int4((int)tsec[6], (int)tsec[9] | (n_paths << 8) | ((p.tile_cap > 32) << 16), (int)tsec[11], found >= 0);
        }
    }
    if (found < 0) return;
    const double wlen = lane < 5 ? segp[RSB_SEGW * lane + 6] : 0.0;
    const int code = __double2loint(segp[7]), nseg = __double2hiint(segp[7]);
    if (lane < 5) {
        double lm = lane < nseg ? wlen / MAXC : 0.0;
        if (p.rs_lengths) {
            if (obs_f64 & 1) ((double*)p.rs_lengths)[5 * (size_t)scene + lane] = lm;
            else ((float*)p.rs_lengths)[5 * (size_t)scene + lane] = (float)lm;
        }
        p.rs_word[8 * (size_t)scene + lane] = lane < nseg ? (int8_t)type_of(code, lane) : (int8_t)HOPE_RS_NONE;
    }
    if (lane == 5) p.rs_word[8 * (size_t)scene + 5] = (int8_t)nseg;
    if (lane == 6) p.rs_word[8 * (size_t)scene + 6] = 1;
}

}  // namespace

static size_t rs_lds_bytes_exact(int max_obst) {
    return (size_t)(10 * max_obst + RSB_WORDS) * 8 + (size_t)((max_obst + 3) & ~3) * 4 + RSB_QCAP;
}
static size_t rs_lds_bytes_filter(int max_obst) {
    return (size_t)(6 * max_obst + RSB_WORDS_F) * 8 + (size_t)((max_obst + 3) & ~3) * 4 + RSB_QCAP + (size_t)((max_obst + 3) & ~3);
}
size_t rs_lds_bytes(int max_obst) { return std::max(rs_lds_bytes_exact(max_obst), rs_lds_bytes_filter(max_obst)); }
size_t rs_screen_lds_bytes(int max_obst) {                // k_rs_screen: view 48 B / obstacle | four tables | pair queue | cand | edge flags
    return (size_t)(6 * max_obst + 4 * RS_SEG_TABLE) * 8 + 2 * WAVE * 4 + (size_t)((max_obst + 3) & ~3) * 4 + (size_t)((max_obst + 3) & ~3);
}
size_t rs_rec_bytes_per_scene() { return sizeof(double) * RS_REC_DOUBLES; }

hipError_t rs_prof_read(unsigned long long* out, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    static unsigned long long buf[64 * 16];
    e = hipMemcpyFromSymbol(buf, HIP_SYMBOL(g_rs_prof), sizeof(buf));
    for (int i = 0; i < 16; i++) { out[i] = 0; for (int sh = 0; sh < 64; sh++) out[i] += buf[sh * 16 + i]; }
    if (e == hipSuccess && reset) {
        for (auto& v : buf) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_rs_prof), buf, sizeof(buf));
    }
    return e;
}

hipError_t rs_fdump_read(double* out /*[64][16]*/) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_fdump), sizeof(double) * 64 * 16);
}

hipError_t rs_log_read(int* out /*[cap][4]*/, int cap, int* n, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(n, HIP_SYMBOL(g_rs_log_n), sizeof(int));
    if (e != hipSuccess) return e;
    const int m = *n < cap ? (*n < RS_LOG_CAP ? *n : RS_LOG_CAP) : cap;
    if (m > 0) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_log), sizeof(int4) * (size_t)m);
    if (e == hipSuccess && reset) { const int z = 0; e = hipMemcpyToSymbol(HIP_SYMBOL(g_rs_log_n), &z, sizeof(int)); }
    return e;
}

hipError_t rs_fstat_read(unsigned long long* out /*[16]*/, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_fstat), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_rs_fstat), z, sizeof(z));
    }
    return e;
}

// first-segment sample table of k_rs_validate_f: step added k + 1 times, sequentially, in float64 (the reference's `pd += d`)
hipError_t rs_init_tables() {
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
    double tab[RS_TAB];
    const volatile double step = RS_STEP * MAXC;          // (volatile: every addition rounds to float64, whatever the host compiler does)
    volatile double t = step;
    for (int k = 0; k < RS_TAB; k++) { tab[k] = t; t = t + step; }
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_rs_steps), tab, sizeof(tab));
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
    return e;
}

hipError_t launch_rs_search(const RsParams& p, hipStream_t stream, LaunchTimer* timer, hipEvent_t after_segs) {
    if (p.max_queue <= 0) return hipSuccess;
    // (read per call: the tests switch kernels inside one process)
    static const bool timing = getenv("HOPE_RS_TIMING") != nullptr;      // cycle accounting build (tools/rs_timing.py)
    const bool exact = getenv("HOPE_RS_EXACT") != nullptr;               // the all-float64 validation kernel (the float32 filter's reference)
    const int occ = getenv("HOPE_RS_OCC") ? atoi(getenv("HOPE_RS_OCC")) : 0;   // exact kernel: 3 (168 VGPRs, default) or 4 (128, spills); filter: 4 (default), 5, 6
    const int dbg = getenv("HOPE_RS_DEBUG") ? (int)strtol(getenv("HOPE_RS_DEBUG"), nullptr, 0) : 0;   // profiling / self-check switches
    const bool stats = (dbg & 0x6000) != 0;                              // float32-filter statistics / self-check build
    size_t lds = exact ? rs_lds_bytes_exact(p.tile_cap) : rs_lds_bytes_filter(p.tile_cap);
    if (!exact) {
        // Waves of this kernel per CU, enforced through the LDS request (HOPE_RS_WPC1 / HOPE_RS_WPC0: large- / small-tile launch).
        // The large-tile launch shares the GPU with the observation half of k_env_step: measured at 65 536 scenes, steady state,
        // 0.692 ms per step with 4 .. 8 waves per CU, 0.705 with 10, 0.73 with the 13 its 9.5 KB of LDS would allow -- the step
        // kernel's waves are the better use of the wave slots.  The small-tile launch must not be limited (8 per CU: 0.92 ms).
        const char* w = getenv(p.tile_cap > 32 ? "HOPE_RS_WPC1" : "HOPE_RS_WPC0");
        const int wpc = w ? atoi(w) : (p.tile_cap > 32 ? 6 : 0);       // (round 5, with the screen pass: 6 -> 0.4930 ms, 8 -> 0.4958, 12 -> 0.500, unlimited 0.503)
        if (wpc > 0) lds = std::max(lds, (size_t)((158 * 1024 / wpc) & ~255));
    }
    if (lds > 48 * 1024) {                                  // (the two-kernel form's walk kernel; k_rs_screen stays far below)
        hipError_t e = hipFuncSetAttribute((const void*)k_rs_validate_f<RSF_OCC, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const void* vk = exact ? (timing ? (const void*)k_rs_validate<3, true> : occ != 4 ? (const void*)k_rs_validate<3, false> : (const void*)k_rs_validate<4, false>)
                           : (timing ? (const void*)k_rs_validate_f<RSF_OCC, true, false> : stats ? (const void*)k_rs_validate_f<RSF_OCC, false, true>
                              : occ == 5 ? (const void*)k_rs_validate_f<5, false, false>
                              : occ == 3 ? (const void*)k_rs_validate_f<3, false, false> : (const void*)k_rs_validate_f<RSF_OCC, false, false>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(vk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (!exact) { hipError_t e = rs_init_tables(); if (e != hipSuccess) return e; }
    // grid = number of scenes in this tile class (upper bound of the queue length, which lives on the device)
    const int stop_after = getenv("HOPE_RS_STOP") ? atoi(getenv("HOPE_RS_STOP")) : 0;   // (debugging: 1 = no RS kernel, 2 = words only, 3 = words + segs; results invalid)
    if (stop_after == 1) return hipSuccess;
    if (timer) timer->begin(HOPE_K_RS_WORDS, stream);
    hipLaunchKernelGGL(k_rs_words, dim3((p.max_queue + RSA_SCENES - 1) / RSA_SCENES, RSA_GROUPS), dim3(WAVE), 0, stream, p);
    if (timer) timer->end(stream);
    if (stop_after == 2) return hipGetLastError();
    if (timer) timer->begin(HOPE_K_RS_SEGS, stream);
    hipLaunchKernelGGL(k_rs_segs, dim3((p.max_queue + 7) / 8), dim3(WAVE), 0, stream, p);
    if (timer) timer->end(stream);
    if (after_segs) { hipError_t e = hipEventRecord(after_segs, stream); if (e != hipSuccess) return e; }   // (pipelined steps: the next motion launch waits here)
    if (stop_after == 3) return hipGetLastError();
    // two-kernel validation (HOPE_RS_SPLIT=1; measured and NOT adopted): k_rs_screen at 8 waves per SIMD, then the walk only for the
    // searches it left a word of.  65 536 scenes, steady state, same box: 0.531 ms per step against 0.514 for the one-kernel form
    // (profiles/r05_ab_two_kernel_validation.txt) -- the second launch over the whole queue, the survivors' second prologue and the
    // extra link in the search chain cost more than the freed register slots give back.  Kept as a build that the tests compare.
    const bool split = !exact && !timing && !stats && !(dbg & 0x20000) && p.surv_count && p.surv_list &&
                       getenv("HOPE_RS_SPLIT") && atoi(getenv("HOPE_RS_SPLIT")) == 1;
    if (split) {
        if (timer) timer->begin(HOPE_K_RS_SCREEN, stream);
        static const int socc = getenv("HOPE_RS_SCREEN_OCC") ? atoi(getenv("HOPE_RS_SCREEN_OCC")) : 8;      // (A/B: 7 = 72 registers, no spills)
        if (socc == 7) hipLaunchKernelGGL((k_rs_screen<false, 7>), dim3(p.max_queue), dim3(WAVE), rs_screen_lds_bytes(p.tile_cap), stream, p);
        else if (socc == 6) hipLaunchKernelGGL((k_rs_screen<false, 6>), dim3(p.max_queue), dim3(WAVE), rs_screen_lds_bytes(p.tile_cap), stream, p);
        else hipLaunchKernelGGL((k_rs_screen<false, 8>), dim3(p.max_queue), dim3(WAVE), rs_screen_lds_bytes(p.tile_cap), stream, p);
        if (timer) timer->end(stream);
    }
    if (timer) timer->begin(HOPE_K_RS_VALIDATE, stream);
    static const bool no_prio = getenv("HOPE_RS_PRIO") && atoi(getenv("HOPE_RS_PRIO")) == 0;
    // only for the launch of the class with more scenes, i.e. the longer chain (both launches: 0.657 ms / steady 0.688; only the
    // longer chain's: 0.657 / 0.678; only the shorter chain's: 0.672 / 0.696); below 32 768 scenes neutral to -1.6 %: off
    const bool longer_chain = 2 * (long long)p.max_queue >= p.n;
    const int flags = (p.obs_f64 ? 1 : 0) | dbg | ((no_prio || p.n < 32768 || !longer_chain) ? 0x10000 : 0);
    if (exact) {
        if (timing) hipLaunchKernelGGL((k_rs_validate<3, true>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
        else if (occ != 4) hipLaunchKernelGGL((k_rs_validate<3, false>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
        else hipLaunchKernelGGL((k_rs_validate<4, false>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    } else if (split) hipLaunchKernelGGL((k_rs_validate_f<RSF_OCC, false, false, true>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    else if (timing) hipLaunchKernelGGL((k_rs_validate_f<RSF_OCC, true, false>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    else if (stats) hipLaunchKernelGGL((k_rs_validate_f<RSF_OCC, false, true>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    else if (occ == 5) hipLaunchKernelGGL((k_rs_validate_f<5, false, false>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    else if (occ == 3) hipLaunchKernelGGL((k_rs_validate_f<3, false, false>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    else hipLaunchKernelGGL((k_rs_validate_f<RSF_OCC, false, false>), dim3(p.max_queue), dim3(WAVE), lds, stream, p, flags);
    if (timer) timer->end(stream);
    return hipGetLastError();
}

}  // namespace hope
