// hope_rs.hip -- Reeds-Shepp feasibility search: ONE WAVEFRONT PER ELIGIBLE SCENE.
//
// Replaces CarParking.find_rs_path (src/env/car_parking_base.py:413-450) with
//   rsCurve.calc_all_paths / generate_path / set_path / generate_local_course / interpolate
//     (src/env/reeds_shepp.py:35-557) and CarParking.is_traj_valid (car_parking_base.py:452-534),
// for the scenes the step kernel queued (gate :293-294: t > 1, CONTINUE, |pos - dest| < 10).
//
// Mapping onto the wave:
//   1. the 46 word-solver calls of generate_path run one per lane (lanes 0..45);
//   2. set_path's ORDER-DEPENDENT de-dup (signed length-sum test, :63-66) is replayed sequentially over
//      the candidates with a cross-lane ballot; L = sum |len|, L >= 1000 dropped (:68-71);
//   3. lane 0 replays heapdict's array heap (push order = path order, non-strict sift-up, strict
//      sift-down) to obtain the pop order find_rs_path sees, ties included;
//   4. for each popped path (stop rule :443) the samples of generate_local_course are produced 64 at a
//      time: the `pd += d` chain is run in lock-step by all lanes (sequential rounding kept), each
//      lane interpolates ITS sample, builds the 4 hull edges and tests them against the obstacle tile
//      in LDS.  A (hull edge, obstacle edge) pair can only be a hit if the two edge bounding boxes
//      overlap (the reference requires the intersection point inside both), so pairs failing that
//      exact pre-test skip the two float64 divisions.  The reference's obstacle cull (:482-494) and
//      its T x 4 x E matrix are result-neutral and are not materialised.
// Deviation (documented in DESIGN.md): generate_local_course's "pop trailing samples whose local x is
// exactly 0.0" (:501-505) is not replayed beyond the unused array tail (a measure-zero event).
#include <stdlib.h>

#include "hope_dev.h"
#include "hope_internal.h"

namespace hope {

namespace {

constexpr double MAXC = 0.3327130214085973;      // math.hm_tan(VALID_STEER[-1]) / WHEEL_BASE  (car_parking_base.py:422)
constexpr double RS_STEP = 0.1;                  // step_size passed by find_rs_path (:424)
constexpr double MAX_LENGTH = 1000.0;            // reeds_shepp.py:6
constexpr int NCAND = 46;

enum { TS = 0, TL = 1, TR = 2 };

// lane l's value of a wave-uniformly indexed double (l must be wave-uniform)
__device__ __forceinline__ double readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

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
__device__ bool rs_LSL(double x, double y, double phi, double& t, double& u, double& v) {   // :79-87
    double uu, tt;
    rs_R(x - hm_sin(phi), y - 1.0 + hm_cos(phi), uu, tt);
    if (tt >= 0.0) {
        double vv = rs_M(phi - tt);
        if (vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ bool rs_LSR(double x, double y, double phi, double& t, double& u, double& v) {   // :90-103
    double u1, t1;
    rs_R(x + hm_sin(phi), y - 1.0 - hm_cos(phi), u1, t1);
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
__device__ bool rs_LRL(double x, double y, double phi, double& t, double& u, double& v) {   // :106-117
    double u1, t1;
    rs_R(x - hm_sin(phi), y - 1.0 + hm_cos(phi), u1, t1);
    if (u1 <= 4.0) {
        double uu = -2.0 * hm_asin(0.25 * u1);
        double tt = rs_M(t1 + 0.5 * uu + PI);
        double vv = rs_M(phi - tt + uu);
        if (tt >= 0.0 && uu <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ void calc_tauOmega(double u, double v, double xi, double eta, double phi, double& tau, double& omega) {
    double delta = rs_M(u - v);                                                             // :228-243
    double A = hm_sin(u) - hm_sin(delta);
    double B = hm_cos(u) - hm_cos(delta) - 1.0;
    double t1 = hm_atan2(eta * A - xi * B, xi * A + eta * B);
    double t2 = 2.0 * (hm_cos(delta) - hm_cos(v) - hm_cos(u)) + 3.0;
    if (t2 < 0) tau = rs_M(t1 + PI); else tau = rs_M(t1);
    omega = rs_M(tau - u + v - phi);
}
__device__ bool rs_LRLRn(double x, double y, double phi, double& t, double& u, double& v) { // :246-257
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi);
    double rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (rho <= 1.0) {
        double uu = hm_acos(rho), tt, vv;
        calc_tauOmega(uu, -uu, xi, eta, phi, tt, vv);
        if (tt >= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ bool rs_LRLRp(double x, double y, double phi, double& t, double& u, double& v) { // :260-272
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi);
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
__device__ bool rs_LRSR(double x, double y, double phi, double& t, double& u, double& v) {  // :311-323
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi), rho, theta;
    rs_R(-eta, xi, rho, theta);
    if (rho >= 2.0) {
        double tt = theta, uu = 2.0 - rho, vv = rs_M(tt + 0.5 * PI - phi);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
__device__ bool rs_LRSL(double x, double y, double phi, double& t, double& u, double& v) {  // :326-339
    double xi = x - hm_sin(phi), eta = y - 1.0 + hm_cos(phi), rho, theta;
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
__device__ bool rs_LRSLR(double x, double y, double phi, double& t, double& u, double& v) { // :414-429
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi), rho, theta;
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
        double ldx = hm_sin(l) / MAXC;
        double ldy = (m == TL) ? (1.0 - hm_cos(l)) / MAXC : (1.0 - hm_cos(l)) / (-MAXC);
        double gdx = c_noy * ldx + s_noy * ldy;          // hm_cos(-oyaw)*ldx + hm_sin(-oyaw)*ldy
        double gdy = -s_noy * ldx + c_noy * ldy;
        px = ox + gdx;
        py = oy + gdy;
        pyaw = (m == TL) ? oyaw + l : oyaw - l;
    }
}

// is_traj_valid for the (up to 64) poses held one per lane: returns true on a lane whose pose is out of the
// map box or whose hull meets an obstacle edge (line-line intersection inside both edge boxes, no tolerance).
// Obstacles are culled per call: the union box of the active lanes' hulls is wave-reduced, one lane per
// obstacle compares its precomputed box (obb, in LDS) with it, and only the survivors (cand[], usually 0-3)
// are visited.  A pair whose boxes do not overlap cannot pass the reference's box tests, so this is exact.
__device__ __forceinline__ bool pose_hits(bool active, double wx, double wy, double wyaw, const double* tile,
                                          const double* obb, int* cand, int n_obst, double xmin, double xmax,
                                          double ymin, double ymax, int lane) {
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
    if (__any(bad)) return bad;
    // union box of the pass: float, rounded outwards (only used to cull whole obstacles, so a superset is exact)
    float ulox = active ? __double2float_rd(hminx) : INFINITY, uhix = active ? __double2float_ru(hmaxx) : -INFINITY;
    float uloy = active ? __double2float_rd(hminy) : INFINITY, uhiy = active ? __double2float_ru(hmaxy) : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ulox = fminf(ulox, __shfl_xor(ulox, off));
        uhix = fmaxf(uhix, __shfl_xor(uhix, off));
        uloy = fminf(uloy, __shfl_xor(uloy, off));
        uhiy = fmaxf(uhiy, __shfl_xor(uhiy, off));
    }
    const double uminx = ulox, umaxx = uhix, uminy = uloy, umaxy = uhiy;
    int nc = 0;
    for (int base = 0; base < n_obst; base += WAVE) {
        int o = base + lane;
        bool near = false;
        if (o < n_obst) {
            const double* bb = obb + 4 * o;
            near = !(bb[0] > umaxx || bb[1] < uminx || bb[2] > umaxy || bb[3] < uminy);
        }
        unsigned long long m = __ballot(near);
        if (near) cand[nc + __popcll(m & ((1ull << lane) - 1))] = o;
        nc += __popcll(m);
    }
    if (nc == 0) return false;
    wsync();
    for (int ci = 0; ci < nc; ci++) {
        const int r = cand[ci];
        const double* o = tile + 8 * r;
        const double* bb = obb + 4 * r;
        bool near = active && !(bb[0] > hmaxx || bb[1] < hminx || bb[2] > hmaxy || bb[3] < hminy);
        if (!__any(near)) continue;
        if (near) {
#pragma unroll
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
            }
        }
    }
    wsync();
    return bad;
}

// ================================================================================================
// Kernel A: generate_path + set_path + heapdict order  ->  ordered word list per queued scene
// ================================================================================================
// The 46 solver calls run one per lane.  Instead of twelve divergent solver bodies, the lanes share the
// expensive steps: one hm_sincos(phi'), one (hypot, atan2) of the solver's polar argument, one asin/acos,
// one second atan2, then short per-family tails -- the arithmetic of each solver is unchanged.
constexpr int RSA_LM = 0, RSA_PR = 64, RSA_WORDS = 128;     // LDS doubles, then ints hid[64], order[64]

__global__ __launch_bounds__(64) void k_rs_words(RsParams p) {
    __shared__ double scr[RSA_WORDS];
    __shared__ int hid[64];
    __shared__ int order[64];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= *p.rs_count) return;
    const int scene = p.rs_list[blockIdx.x];
    const int slot = p.slot_base + p.slot_dir * (int)blockIdx.x;
    const double* sc = p.scene_c + (size_t)scene * SC_WORDS;
    const double* st = p.state + (size_t)scene * ST_WORDS;
    const double q0x = st[0], q0y = st[1], q0w = st[2];
    const double gx = sc[SC_DEST], gy = sc[SC_DEST + 1], gw = sc[SC_DEST + 2];

    // ---- generate_path (:540-557): normalise the goal into the start frame ------------------------
    double X, Y, PHI;
    {
        double dx = gx - q0x, dy = gy - q0y;
        PHI = gw - q0w;
        double c = hm_cos(q0w), s = hm_sin(q0w);
        X = (c * dx + s * dy) * MAXC;
        Y = (-s * dx + c * dy) * MAXC;
    }
    double l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0;
    int code = 0, wn = 0;
    bool ok = false;
    if (lane < NCAND) {
        int g, q;
        cand_decode(lane, g, q);
        double bx = X, by = Y;
        if (g == 4 || g == 9 || g == 10) {               // "backwards" (:206-207, :376-377)
            bx = X * hm_cos(PHI) + Y * hm_sin(PHI);
            by = X * hm_sin(PHI) - Y * hm_cos(PHI);
        }
        const double sx = (q & 1) ? -bx : bx;
        const double sy = (q & 2) ? -by : by;
        const double sp = (q == 1 || q == 2) ? -PHI : PHI;
        double t = 0, u = 0, v = 0;
        if (g == 0) {
            ok = rs_SLS(sx, sy, sp, t, u, v);
        } else {
            double s_, c_;
            hm_sincos(sp, &s_, &c_);
            const bool plus = (g == 2 || g == 5 || g == 6 || g == 8 || g == 10 || g == 11);
            const double xi = plus ? sx + s_ : sx - s_;
            const double eta = plus ? sy - 1.0 - c_ : sy - 1.0 + c_;
            const bool isLRLR = (g == 5 || g == 6);
            const bool isLRSR = (g == 8 || g == 10);
            double r = 0, th = 0;
            if (!isLRLR) {                                // R(.,.) (:571-578); LRSR uses R(-eta, xi) (:314)
                double ra = isLRSR ? -eta : xi, rb = isLRSR ? xi : eta;
                r = hm_hypot(ra, rb);
                th = hm_atan2(rb, ra);
            }
            bool alive = true, needC = false;
            double cy = 0, cx = 1, vv = 0, t2 = 0;
            if (g == 1) {                                 // LSL :79-87
                t = th; u = r; alive = t >= 0.0;
            } else if (g == 2) {                          // LSR :90-103
                double u1 = r * r;
                alive = u1 >= 4.0;
                if (alive) { u = sqrt(u1 - 4.0); cy = 2.0; cx = u; needC = true; }
            } else if (g == 3 || g == 4) {                // LRL :106-117
                alive = r <= 4.0;
                if (alive) u = -2.0 * hm_asin(0.25 * r);
            } else if (isLRSR) {                          // LRSR :311-323
                alive = r >= 2.0;
                if (alive) { t = th; u = 2.0 - r; v = rs_M(t + 0.5 * PI - sp); }
            } else if (g == 7 || g == 9) {                // LRSL :326-339
                alive = r >= 2.0;
                if (alive) { double rr = sqrt(r * r - 4.0); u = 2.0 - rr; cy = rr; cx = -2.0; needC = true; }
            } else if (g == 11) {                         // LRSLR :414-429
                alive = r >= 2.0;
                if (alive) {
                    u = 4.0 - sqrt(r * r - 4.0);
                    alive = u <= 0.0;
                    if (alive) { cy = (4.0 - u) * xi - 2.0 * eta; cx = -2.0 * xi + (u - 4.0) * eta; needC = true; }
                }
            } else {                                      // LRLRn :246-257 / LRLRp :260-272 + calc_tauOmega :228-243
                if (g == 5) {
                    double rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
                    alive = rho <= 1.0;
                    if (alive) { u = hm_acos(rho); vv = -u; }
                } else {
                    double rho = (20.0 - xi * xi - eta * eta) / 16.0;
                    alive = 0.0 <= rho && rho <= 1.0;
                    if (alive) { u = -hm_acos(rho); alive = u >= -0.5 * PI; vv = u; }
                }
                if (alive) {
                    double delta = rs_M(u - vv);
                    double su, cu, sd, cd;
                    hm_sincos(u, &su, &cu);
                    hm_sincos(delta, &sd, &cd);
                    double A = su - sd;
                    double B = cu - cd - 1.0;
                    cy = eta * A - xi * B; cx = xi * A + eta * B; needC = true;
                    t2 = 2.0 * (cd - hm_cos(vv) - cu) + 3.0;
                }
            }
            double th2 = 0;
            if (alive && needC) th2 = hm_atan2(cy, cx);
            if (alive) {
                if (g == 1) { v = rs_M(sp - t); ok = v >= 0.0; }
                else if (g == 2) { t = rs_M(th + th2); v = rs_M(t - sp); ok = t >= 0.0 && v >= 0.0; }
                else if (g == 3 || g == 4) { t = rs_M(th + 0.5 * u + PI); v = rs_M(sp - t + u); ok = t >= 0.0 && u <= 0.0; }
                else if (isLRSR) { ok = t >= 0.0 && u <= 0.0 && v <= 0.0; }
                else if (g == 7 || g == 9) { t = rs_M(th + th2); v = rs_M(sp - 0.5 * PI - t); ok = t >= 0.0 && u <= 0.0 && v <= 0.0; }
                else if (g == 11) { t = rs_M(th2); v = rs_M(t - sp); ok = t >= 0.0 && v >= 0.0; }
                else {
                    t = t2 < 0 ? rs_M(th2 + PI) : rs_M(th2);
                    v = rs_M(t - u + vv - sp);
                    ok = (g == 5) ? (t >= 0.0 && v <= 0.0) : (t >= 0.0 && v >= 0.0);
                }
            }
        }
        const double hp = 0.5 * PI;
        int t0 = TL, t1 = TS, t2_ = TL, t3 = 0, t4 = 0, n = 3;
        l0 = t; l1 = u; l2 = v;
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
        wn = n;
        code = pack_types(t0, t1, t2_, t3, t4, n);
    }

    // ---- set_path (:57-76) -------------------------------------------------------------------------------
    // kept[c] = ok[c] and no EARLIER KEPT path with the same word has sum(old - new) <= 0.01 (:63-66) and L < 1000 (:70).
    // Only candidates that share their word with another candidate can interact, so every lane first computes its own
    // L (parallel) and whether it has such a twin; the order-dependent replay then runs over the twins only.
    double myL = 0;
    myL = myL + fabs(l0); myL = myL + fabs(l1); myL = myL + fabs(l2);
    if (wn > 3) myL = myL + fabs(l3);
    if (wn > 4) myL = myL + fabs(l4);
    const unsigned long long okmask = __ballot(ok);
    bool twin = false;
    for (unsigned long long mm = okmask; mm; mm &= mm - 1) {
        const int c = __builtin_ctzll(mm);
        twin = twin || (c != lane && code == __builtin_amdgcn_readlane(code, c));
    }
    const unsigned long long twins = __ballot(ok && twin);
    unsigned long long kept = __ballot(ok && !twin && !(myL >= MAX_LENGTH));
    for (unsigned long long mm = twins; mm; mm &= mm - 1) {            // call order
        const int c = __builtin_ctzll(mm);
        const int code_c = __builtin_amdgcn_readlane(code, c);
        const int n_c = __builtin_amdgcn_readlane(wn, c);
        const double c0 = readlane_d(l0, c), c1 = readlane_d(l1, c), c2 = readlane_d(l2, c), c3 = readlane_d(l3, c), c4 = readlane_d(l4, c);
        bool dup = false;
        if (((kept >> lane) & 1) && code == code_c) {
            double s = 0;                                  // sum([x - y ...]) left to right (:65)
            s = s + (l0 - c0); s = s + (l1 - c1); s = s + (l2 - c2);
            if (n_c > 3) s = s + (l3 - c3);
            if (n_c > 4) s = s + (l4 - c4);
            dup = s <= 0.01;
        }
        if (__any(dup)) continue;
        if (readlane_d(myL, c) >= MAX_LENGTH) continue;    // :70
        kept |= 1ull << c;
    }
    const int n_paths = __popcll(kept);
    if (lane == 0) p.rs_nwords[slot] = n_paths;
    if (n_paths == 0) return;                              // find_rs_path :427-428

    // ---- path.L / maxc (calc_all_paths :52) and heapdict pop order ----------------------------------
    const double myLm = myL / MAXC;
    const bool mine = (kept >> lane) & 1;
    // heapdict pops distinct priorities in ascending order whatever the heap looked like, so the pop rank is a count;
    // only equal priorities (twin words) depend on the heap's history and need the replay
    int rank = 0;
    bool tie = false;
    for (unsigned long long mm = kept; mm; mm &= mm - 1) {
        const int c = __builtin_ctzll(mm);
        const double Lc = readlane_d(myLm, c);
        rank += Lc < myLm;
        tie = tie || (c != lane && Lc == myLm);
    }
    if (__any(mine && tie)) {
        if (mine) scr[RSA_LM + lane] = myLm;
        __syncthreads();
        if (lane == 0) {
            double* pr = scr + RSA_PR;
            int hn = 0;
            for (int c = 0; c < NCAND; c++) {                 // costQueue[path] = path.L in path order (:432-433)
                if (!((kept >> c) & 1)) continue;
                int i = hn++;
                pr[i] = scr[RSA_LM + c]; hid[i] = c;
                while (i) {                                   // _decrease_key: swap unless parent < child
                    int parent = (i - 1) >> 1;
                    if (pr[parent] < pr[i]) break;
                    double tp = pr[i]; pr[i] = pr[parent]; pr[parent] = tp;
                    int ti = hid[i]; hid[i] = hid[parent]; hid[parent] = ti;
                    i = parent;
                }
            }
            int no = 0;
            while (hn > 0) {                                  // popitem
                order[hid[0]] = no++;
                if (hn == 1) { hn = 0; break; }
                hn--;
                pr[0] = pr[hn]; hid[0] = hid[hn];
                int i = 0;
                for (;;) {                                    // _min_heapify
                    int l = (i << 1) + 1, r = (i + 1) << 1, low;
                    if (l < hn && pr[l] < pr[i]) low = l; else low = i;
                    if (r < hn && pr[r] < pr[low]) low = r;
                    if (low == i) break;
                    double tp = pr[i]; pr[i] = pr[low]; pr[low] = tp;
                    int ti = hid[i]; hid[i] = hid[low]; hid[low] = ti;
                    i = low;
                }
            }
        }
        __syncthreads();
        rank = order[lane];
    }
    if ((kept >> lane) & 1) {                             // lane c writes its word at its pop rank
        RsWord* w = p.rs_words + (size_t)slot * RS_WORDS_PER_SCENE + rank;
        w->len[0] = l0; w->len[1] = l1; w->len[2] = l2; w->len[3] = l3; w->len[4] = l4;
        w->Lm = myLm; w->code = code; w->n = wn;
    }
}

// ================================================================================================
// Kernel B: find_rs_path's main loop (:436-450) over the ordered words
// ================================================================================================
// LDS (doubles): tile 8*M | obstacle boxes 4*M | segment params 5 x 8 | sample queue pd[512] | ints: cand[M] | bytes: seg[512]
constexpr int RSB_SEG = 0, RSB_SEGW = 10, RSB_QPD = 50, RSB_WORDS = 562;
constexpr int RSB_QCAP = 512;

__global__ __launch_bounds__(64, 3) void k_rs_validate(RsParams p, int obs_f64) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= *p.rs_count) return;
    if (obs_f64 & 0x400) return;                          // profiling switch: words kernel only
    const int slot = p.slot_base + p.slot_dir * (int)blockIdx.x;
    const int n_paths = p.rs_nwords[slot];
    if (n_paths == 0) return;
    const int scene = p.rs_list[blockIdx.x];
    const int n_obst = p.n_obst[scene];
    double* tile = lds;
    double* obb = lds + 8 * p.tile_cap;
    double* scr = lds + 12 * p.tile_cap;
    double* segp = scr + RSB_SEG;
    double* qpd = scr + RSB_QPD;
    int* cand = (int*)(scr + RSB_WORDS);
    unsigned char* qseg = (unsigned char*)(cand + ((p.tile_cap + 3) & ~3));

    {
        const double2* src = (const double2*)(p.verts + (size_t)scene * p.max_obst * 8);
        double2* dst = (double2*)tile;
        for (int v = lane; v < 4 * n_obst; v += WAVE) dst[v] = src[v];
        for (int o = lane; o < n_obst; o += WAVE) {           // obstacle boxes (xmin, xmax, ymin, ymax)
            const double* v = p.verts + ((size_t)scene * p.max_obst + o) * 8;
            obb[4 * o] = fmin(fmin(v[0], v[2]), fmin(v[4], v[6]));
            obb[4 * o + 1] = fmax(fmax(v[0], v[2]), fmax(v[4], v[6]));
            obb[4 * o + 2] = fmin(fmin(v[1], v[3]), fmin(v[5], v[7]));
            obb[4 * o + 3] = fmax(fmax(v[1], v[3]), fmax(v[5], v[7]));
        }
    }
    const double* sc = p.scene_c + (size_t)scene * SC_WORDS;
    const double* st = p.state + (size_t)scene * ST_WORDS;
    const double q0x = st[0], q0y = st[1], q0w = st[2];
    const double xmin = sc[SC_BBOX], xmax = sc[SC_BBOX + 1], ymin = sc[SC_BBOX + 2], ymax = sc[SC_BBOX + 3];
    const double c_q = hm_cos(-q0w), s_q = hm_sin(-q0w);
    const double step = RS_STEP * MAXC;                   // step_size * maxc (:44)
    wsync();

    const RsWord* words = p.rs_words + (size_t)slot * RS_WORDS_PER_SCENE;
    double min_path_len = -1;
    int found = -1;
    for (int idx = 1; idx <= n_paths; idx++) {
        const RsWord* W = words + (idx - 1);
        const double Lm = W->Lm;
        if (min_path_len < 0) min_path_len = Lm;
        if (Lm > 1.6 * min_path_len && idx > 2) break;    // :443
        if ((obs_f64 & 0x200) && idx > 1) break;          // profiling switch: first path only
        const int code = W->code, nseg = W->n;
        double len[5];
#pragma unroll
        for (int i = 0; i < 5; i++) len[i] = W->len[i];

        // generate_local_course (:452-507).  Samples are queued as (pd, segment) and collision-tested 64 at a
        // time, so short segments share a pass.  Sample 0 (the start pose, local (0,0,0)) = segment 0 at pd = 0.
        //
        // Segment origins first.  The origin headings are plain sums (oyaw_{i+1} = oyaw_i +- l_i), so every
        // sine/cosine the path needs -- of the five origin headings and of the five segment lengths (for the
        // segment end points, interpolate(ind, l, ...) :497-498) -- comes from ONE lane-parallel sincos
        // (lanes 0-4: headings, lanes 5-9: lengths); hm_cos(-x) = cos x and hm_sin(-x) = -sin x give :522-523.
        bool invalid = false;
        {
            double hy[5];
            hy[0] = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = type_of(code, i);
                hy[i + 1] = (m == TL) ? hy[i] + len[i] : ((m == TR) ? hy[i] - len[i] : hy[i]);
            }
            double arg = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (lane == i) arg = hy[i];
                if (lane == 5 + i) arg = len[i];
            }
            double sv, cv;
            hm_sincos(arg, &sv, &cv);
            double ox = 0, oy = 0;
            wsync();
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (i < nseg) {
                    const int m = type_of(code, i);
                    const double c_oy = __shfl(cv, i), s_oy = __shfl(sv, i);
                    const double sl = __shfl(sv, 5 + i), cl = __shfl(cv, 5 + i);
                    if (lane == 0) {
                        double* sp_ = segp + RSB_SEGW * i;
                        sp_[0] = ox; sp_[1] = oy; sp_[2] = hy[i]; sp_[3] = c_oy; sp_[4] = s_oy; sp_[5] = c_oy; sp_[6] = -s_oy;
                        sp_[7] = (double)m; sp_[8] = len[i];
                    }
                    const double l = len[i];
                    if (m == TS) {                                   // interpolate(l) = next origin (:512-513)
                        ox = ox + l / MAXC * c_oy;
                        oy = oy + l / MAXC * s_oy;
                    } else {
                        const double ldx = sl / MAXC;
                        const double ldy = (m == TL) ? (1.0 - cl) / MAXC : (1.0 - cl) / (-MAXC);
                        ox = ox + (c_oy * ldx + (-s_oy) * ldy);
                        oy = oy + (s_oy * ldx + c_oy * ldy);
                    }
                }
            }
            if (lane == 0) { qpd[0] = 0.0; qseg[0] = 0; }
        }
        int nq = 1;
        wsync();
        // Resumable sample generator + ONE window test site.  Samples are appended to the queue until it cannot
        // take another chunk (or the path ends); then the window is tested coarse-to-fine: a path is invalid as
        // soon as ANY of its samples is bad, whatever the order they are looked at, and a path that crosses an
        // obstacle has long runs of bad samples -- so a strided subset that fills one wave goes first and only clean
        // windows pay for the remaining samples.  Exact: all of them are real samples
        // of generate_local_course.
        // window size: generate <= 130 samples, test them, continue (measured best of 66/130/258/512: larger windows
        // lose the early exit on invalid paths, smaller ones pay more partially filled passes)
        const int win = (obs_f64 >> 12) ? (obs_f64 >> 12) : 130;
        int i = 0;
        bool seg_open = false, finished = false;
        double pd = 0, ll = 0.0, lprev = 0.0, d = 0, l = 0;
        while (!invalid && (!finished || nq > 0)) {
            while (!finished && nq + WAVE + 1 <= win) {
                if (!seg_open) {
                    l = segp[RSB_SEGW * i + 8];               // len[i]
                    d = l > 0.0 ? step : -step;
                    if (i >= 1 && (lprev * l) > 0) pd = -d - ll; else pd = d - ll;
                    lprev = l;
                    seg_open = true;
                }
                // `pd += d` chain (sequential rounding kept): every lane walks it, lane j keeps the value after
                // j additions; the walk stops at the first block of 8 whose last value already left the segment
                double mine = pd, t = pd, last = pd;
                int ncap = 0;
                for (int j0 = 0; j0 < WAVE; j0 += 8) {
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        if (lane == j0 + jj) mine = t;
                        last = t;
                        t = t + d;
                    }
                    ncap = j0 + 8;
                    if (__any(fabs(last) > fabs(l))) break;
                }
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
            const int n = nq;
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
                    const int m = (int)sp_[7];
                    interpolate(spd, m, sp_[0], sp_[1], sp_[2], sp_[5], sp_[6], sp_[3], sp_[4], px, py, pyaw);
                }
                const double wx = c_q * px + s_q * py + q0x;      // calc_all_paths :47-49
                const double wy = -s_q * px + c_q * py + q0y;
                const double wyaw = pi_2_pi(pyaw + q0w);
                if (!(obs_f64 & 0x100) && __any(pose_hits(active, wx, wy, wyaw, tile, obb, cand, n_obst, xmin, xmax, ymin, ymax, lane))) {
                    invalid = true;
                    break;
                }
            }
            nq = 0;
            wsync();
        }
        if (!invalid) { found = idx - 1; break; }
    }
    if (found < 0) return;

    // ---- output: PATH.ctypes / PATH.lengths (metres) of the first collision-free path ----------------
    const RsWord* W = words + found;
    const int code = W->code, nseg = W->n;
    if (lane < 5) {
        double lm = lane < nseg ? W->len[lane] / MAXC : 0.0;      // path.lengths = [l / maxc ...] (:51)
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

size_t rs_lds_bytes(int max_obst) {
    return (size_t)(12 * max_obst + RSB_WORDS) * 8 + (size_t)((max_obst + 3) & ~3) * 4 + RSB_QCAP;
}
size_t rs_words_bytes_per_scene() { return sizeof(RsWord) * RS_WORDS_PER_SCENE; }

hipError_t launch_rs_search(const RsParams& p, hipStream_t stream, LaunchTimer* timer) {
    if (p.max_queue <= 0) return hipSuccess;
    size_t lds = rs_lds_bytes(p.tile_cap);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_rs_validate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // grid = number of scenes in this tile class (upper bound of the queue length, which lives on the device)
    if (timer) timer->begin(HOPE_K_RS_WORDS, stream);
    hipLaunchKernelGGL(k_rs_words, dim3(p.max_queue), dim3(WAVE), 0, stream, p);
    if (timer) timer->end(stream);
    static const int dbg = getenv("HOPE_RS_DEBUG") ? atoi(getenv("HOPE_RS_DEBUG")) : 0;   // profiling switches
    if (timer) timer->begin(HOPE_K_RS_VALIDATE, stream);
    hipLaunchKernelGGL(k_rs_validate, dim3(p.max_queue), dim3(WAVE), lds, stream, p, (p.obs_f64 ? 1 : 0) | dbg);
    if (timer) timer->end(stream);
    return hipGetLastError();
}

}  // namespace hope
