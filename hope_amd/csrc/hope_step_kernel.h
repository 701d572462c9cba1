// hope_step_kernel.h -- the fused env-step kernel: ONE WAVEFRONT (= one 64-thread workgroup) PER SCENE.
//
// Replaces, for every scene in the batch (reference file:line):
//   CarParkingWrapper.step       env_wrapper.py:73-81   action_rescale, reward_shaping, done
//   CarParking.step              car_parking_base.py:235-299 (sub-step loop, retreat, status, reward)
//   KSModel.step                 vehicle.py:69-96       (20 Euler micro-steps x 10 sub-steps)
//   _check_arrived/_detect_collision/_check_status   car_parking_base.py:153-184
//   LidarSimlator.get_observation lidar_simulator.py:31-135
//   ActionMask.get_steps/post_process action_mask.py:166-196
//   _get_targt_repr/_get_reward   car_parking_base.py:186-227,372-381
//
// Data movement per scene-step: the scene's obstacle tile (64 B/obstacle) is read from HBM once into
// LDS and reused by the collision sub-steps, the status check and (transformed in place to the ego
// frame) the lidar; observations are written once.  The mask table (4 MB, shared by all scenes) stays
// L2/MALL resident.  No MFMA: this is geometry, not a contraction.
#pragma once
#include "hope_dev.h"
#include "hope_env.h"

namespace hope {

struct StepParams {
    int n, max_obst;
    uint32_t stages;
    int has_action;
    const double* verts;      // [n][max_obst][4][2]
    const int32_t* n_obst;    // [n]
    const double* scene_c;    // [n][SC_WORDS]
    double* state;            // [n][ST_WORDS]
    int32_t* tstep;           // [n]
    const void* actions;      // [n][2]
    const uint8_t* active;    // [n] or null
    const double* tab;        // prefix-max mask table [NL][NITER][NACT]
    const double* pmax;       // [NL] max over (a,k) of tab
    const double* hull_base;  // [NBEAM]
    const double* beam_ab;    // [NBEAM][2]
    hope_step_out out;
    int32_t* rs_count;        // [1] number of scenes queued for the Reeds-Shepp kernel
    int32_t* rs_list;         // [n]
};

// LDS per wave (doubles): tile 8*max_obst | tx[200] ty[200] | hb[10] cb[10] sb[10] | sh[64] | x[121]+pad | dest box[8] | keep ints
constexpr int LDS_TX = 0, LDS_TY = 200, LDS_HB = 400, LDS_CB = 410, LDS_SB = 420, LDS_SH = 432, LDS_X = 496,
              LDS_DBOX = 624, LDS_KEEP = 632, LDS_SCRATCH_WORDS = 632;
__host__ __device__ inline size_t step_lds_bytes(int max_obst) {
    return (size_t)(8 * max_obst + LDS_SCRATCH_WORDS) * 8 + (size_t)((max_obst + 3) & ~3) * 4;
}

// GEOS Area::ofRingSigned over an open vertex list (ring closed implicitly); lane-0 code, LDS arrays
__device__ __forceinline__ double ring_area_signed_lds(const double* px, const double* py, int n) {
    if (n < 3) return 0.0;
    double sum = 0.0, x0 = px[0];
    for (int i = 1; i < n; i++) {
        double x = px[i] - x0;
        int ip = (i + 1 == n) ? 0 : i + 1;
        sum += x * (py[i - 1] - py[ip]);
    }
    return sum / 2.0;
}

// Polygon(A).intersection(Polygon(B)).area for convex CCW quads: Sutherland-Hodgman + shoelace.
// Runs on ONE lane with LDS scratch sh[64] (two 8-vertex ping-pong buffers of x and y).
__device__ __noinline__ double quad_intersection_area_lane0(const Box& A, const double* B /*8 words x,y*/,
                                                            double* sh) {
    double* ax = sh;      double* ay = sh + 16;
    double* bx = sh + 32; double* by = sh + 48;
    int n = 4;
#pragma unroll
    for (int i = 0; i < 4; i++) { ax[i] = A.x[i]; ay[i] = A.y[i]; }
    for (int e = 0; e < 4 && n > 0; e++) {
        double c1x = B[2 * e], c1y = B[2 * e + 1];
        double c2x = B[2 * ((e + 1) & 3)], c2y = B[2 * ((e + 1) & 3) + 1];
        double ex = c2x - c1x, ey = c2y - c1y;
        int m = 0;
        for (int i = 0; i < n; i++) {
            int i2 = (i + 1 == n) ? 0 : i + 1;
            double sx = ax[i], sy = ay[i], tx = ax[i2], ty = ay[i2];
            double ds = ex * (sy - c1y) - ey * (sx - c1x);
            double dt = ex * (ty - c1y) - ey * (tx - c1x);
            bool sin_ = ds >= 0, tin = dt >= 0;
            if (sin_) { bx[m] = sx; by[m] = sy; m++; }
            if (sin_ != tin) {
                double r = ds / (ds - dt);
                bx[m] = sx + r * (tx - sx);
                by[m] = sy + r * (ty - sy);
                m++;
            }
        }
        n = m;
        double* t;
        t = ax; ax = bx; bx = t;
        t = ay; ay = by; by = t;
    }
    if (n < 3) return 0.0;
    return fabs(ring_area_signed_lds(ax, ay, n));
}

// |hull ∩ dest| with an exact quick reject: both boxes lie inside discs of radius rho about their
// centres; disjoint discs -> GEOS returns an empty intersection, area 0.0.
__device__ __forceinline__ double overlap_area(const Box& box, const double* dbox_lds, double* sh, int lane) {
    double cx = 0.5 * (box.x[0] + box.x[2]), cy = 0.5 * (box.y[0] + box.y[2]);
    double dx = 0.5 * (dbox_lds[0] + dbox_lds[4]) - cx, dy = 0.5 * (dbox_lds[1] + dbox_lds[5]) - cy;
    const double reach = 5.2;   // 2 * half-diagonal (2.5378) + slack
    if (dx * dx + dy * dy > reach * reach) return 0.0;
    double area = 0.0;
    if (lane == 0) area = quad_intersection_area_lane0(box, dbox_lds, sh);
    return __shfl(area, 0);
}

// _detect_collision (car_parking_base.py:153-158): any hull edge x any obstacle edge share a point.
__device__ __forceinline__ bool detect_collision(const Box& b, const double* tile, int n_slots, int lane) {
    double hminx = fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3]));
    double hmaxx = fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3]));
    double hminy = fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3]));
    double hmaxy = fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3]));
    for (int base = 0; base < n_slots; base += WAVE) {
        int e = base + lane;
        bool hit = false;
        if (e < n_slots) {
            int e2 = (e & ~3) | ((e + 1) & 3);
            double x1 = tile[2 * e], y1 = tile[2 * e + 1], x2 = tile[2 * e2], y2 = tile[2 * e2 + 1];
            // envelope of the obstacle edge vs envelope of the hull: necessary for any segment pair
            if (!(fmin(x1, x2) > hmaxx || fmax(x1, x2) < hminx || fmin(y1, y2) > hmaxy || fmax(y1, y2) < hminy)) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int k2 = (k + 1) & 3;
                    hit = hit || segments_intersect(b.x[k], b.y[k], b.x[k2], b.y[k2], x1, y1, x2, y2);
                }
            }
        }
        if (__any(hit)) return true;
    }
    return false;
}

// GEOS Distance::pointToSegment from the origin
__device__ __forceinline__ double origin_seg_dist(double ax, double ay, double bx, double by) {
    if (ax == bx && ay == by) return sqrt(ax * ax + ay * ay);
    double len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);
    double r = ((0.0 - ax) * (bx - ax) + (0.0 - ay) * (by - ay)) / len2;
    if (r <= 0.0) return sqrt(ax * ax + ay * ay);
    if (r >= 1.0) return sqrt(bx * bx + by * by);
    double s = ((ay - 0.0) * (bx - ax) - (ax - 0.0) * (by - ay)) / len2;
    return fabs(s) * sqrt(len2);
}

// one (beam, edge) pair of _fast_calc_lidar_obs (lidar_simulator.py:98-133); returns range or +inf
__device__ __forceinline__ double beam_edge(int i, double a, double b, double x1, double y1, double x2, double y2,
                                            double d, double e, double f) {
    double det = a * e - b * d;
    if (det == 0) return INFINITY;
    double raw_x = (b * f - 0.0 * e) / det;
    double raw_y = (0.0 * d - a * f) / det;
    const double tz = 1e-8;
    bool ok = true;
    if (i < NBEAM / 4 || i >= NBEAM / 4 * 3) ok = ok && !(raw_x < -tz); else ok = ok && !(raw_x > tz);
    if (i < NBEAM / 2) ok = ok && !(raw_y < -tz); else ok = ok && !(raw_y > tz);
    ok = ok && !(raw_x > fmax(x1, x2)) && !(raw_x < fmin(x1, x2));
    ok = ok && !(raw_y > fmax(y1, y2)) && !(raw_y < fmin(y1, y2));
    return ok ? sqrt(raw_x * raw_x + raw_y * raw_y) : INFINITY;
}

template <typename OT, typename AT>
__global__ __launch_bounds__(64) void k_env_step(StepParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int scene = blockIdx.x;
    if (scene >= p.n) return;
    if (p.active && !p.active[scene]) return;

    double* tile = lds;
    double* scr = lds + 8 * p.max_obst;
    int* keep = (int*)(scr + LDS_KEEP);

    // ---- stage the scene: constants (192 B), state (32 B), obstacle tile (64 B x n_obst) ---------
    const double* sc = p.scene_c + (size_t)scene * SC_WORDS;
    const int n_obst = p.n_obst[scene];
    const int n_slots = 4 * n_obst;
    {
        const double2* src = (const double2*)(p.verts + (size_t)scene * p.max_obst * 8);
        double2* dst = (double2*)tile;
        for (int v = lane; v < n_slots; v += WAVE) dst[v] = src[v];   // 16 B/lane, coalesced
    }
    double* dbox = scr + LDS_DBOX;
    if (lane < 8) dbox[lane] = sc[SC_DBOX + lane];
    const double destx = sc[SC_DEST], desty = sc[SC_DEST + 1], desth = sc[SC_DEST + 2];
    const double dest_area = sc[SC_DAREA];
    double* st = p.state + (size_t)scene * ST_WORDS;
    double x = st[0], y = st[1], h = st[2], accum = st[3];
    const double prev_x = x, prev_y = y, prev_h = h;      // prev_state (car_parking_base.py:255)
    int t = p.tstep[scene];
    wsync();

    bool arrive = false;
    bool known_free = false;     // final pose already passed _detect_collision in the sub-step loop
    bool have_ua = false;        // overlap area of the final pose already computed
    double ua = 0.0, ua_prev = 0.0;
    double ct = 0, sn = 0;       // cos/sin of the final heading
    bool have_cs = false;

    if ((p.stages & HOPE_STAGE_MOTION) && p.has_action) {
        // ---- action_rescale (env_wrapper.py:37-50) + KSModel clip (vehicle.py:85-86) -------------
        const AT* act = (const AT*)p.actions;
        double a0 = (double)act[2 * (size_t)scene], a1 = (double)act[2 * (size_t)scene + 1];
        double steer = a0, speed = a1;
        if (!(p.stages & HOPE_ACTION_PHYSICAL)) {
            steer = clipd(a0, -1, 1) * (STEER_HI - STEER_LO) / 2 + (STEER_HI + STEER_LO) / 2;
            speed = clipd(a1, -1, 1) * (SPEED_HI - SPEED_LO) / 2 + (SPEED_HI + SPEED_LO) / 2;
        }
        speed = clipd(speed, SPEED_LO, SPEED_HI);
        steer = clipd(steer, STEER_LO, STEER_HI);
        const double dh = speed * tan(steer) / WHEEL_BASE * STEP_LENGTH / MINI_ITER;

        // ---- heading chain: h_{m+1} = h_m + dh (sequential rounding, vehicle.py:92-93).  Every lane
        // runs the 200-add chain; lane l captures h_m for m = l, l+64, l+128, l+192. -----------------
        double hj = h, hm0 = 0, hm1 = 0, hm2 = 0, hm3 = 0;
        for (int m = 0; m < NUM_STEP * MINI_ITER; m++) {
            int r = m >> 6, i = m & 63;
            if (lane == i) {
                if (r == 0) hm0 = hj; else if (r == 1) hm1 = hj; else if (r == 2) hm2 = hj; else hm3 = hj;
            }
            hj = hj + dh;
            if ((m + 1) % MINI_ITER == 0 && lane == 0) scr[LDS_HB + (m + 1) / MINI_ITER - 1] = hj;
        }
        wsync();
        // lanes 8..17 of round 3 evaluate the ten sub-step boundary headings instead (m = 200..209 unused)
        if (lane >= 8 && lane < 18) hm3 = scr[LDS_HB + lane - 8];
        // ---- per-micro-step displacement terms speed*cos(h)*step_len/mini_iter (vehicle.py:90-91) ----
        {
            double s_, c_;
            sincos(hm0, &s_, &c_);
            scr[LDS_TX + lane] = speed * c_ * STEP_LENGTH / MINI_ITER;
            scr[LDS_TY + lane] = speed * s_ * STEP_LENGTH / MINI_ITER;
            sincos(hm1, &s_, &c_);
            scr[LDS_TX + 64 + lane] = speed * c_ * STEP_LENGTH / MINI_ITER;
            scr[LDS_TY + 64 + lane] = speed * s_ * STEP_LENGTH / MINI_ITER;
            sincos(hm2, &s_, &c_);
            scr[LDS_TX + 128 + lane] = speed * c_ * STEP_LENGTH / MINI_ITER;
            scr[LDS_TY + 128 + lane] = speed * s_ * STEP_LENGTH / MINI_ITER;
            if (lane < 18) {
                sincos(hm3, &s_, &c_);
                if (lane < 8) {
                    scr[LDS_TX + 192 + lane] = speed * c_ * STEP_LENGTH / MINI_ITER;
                    scr[LDS_TY + 192 + lane] = speed * s_ * STEP_LENGTH / MINI_ITER;
                } else {
                    scr[LDS_CB + lane - 8] = c_;
                    scr[LDS_SB + lane - 8] = s_;
                }
            }
        }
        wsync();

        // ---- sub-step loop (car_parking_base.py:259-271) ------------------------------------------
        bool cur_free = false;      // pose at loop entry not checked in this step
        for (int k = 0; k < NUM_STEP; k++) {
            const double px = x, py = y, ph = h;          // prev_info
            const bool prev_free = cur_free;
            ua_prev = ua;
            const bool prev_have_ua = have_ua;
            for (int j = 0; j < MINI_ITER; j++) {         // x += ...; y += ... in micro-step order
                x += scr[LDS_TX + k * MINI_ITER + j];
                y += scr[LDS_TY + k * MINI_ITER + j];
            }
            h = scr[LDS_HB + k];
            ct = scr[LDS_CB + k];
            sn = scr[LDS_SB + k];
            have_cs = true;
            Box box = make_box(x, y, ct, sn);
            ua = overlap_area(box, dbox, scr + LDS_SH, lane);
            have_ua = true;
            if (ua / dest_area > 0.95) { arrive = true; break; }        // _check_arrived :164-170
            if (detect_collision(box, tile, n_slots, lane)) {            // retreat :264-271
                x = px; y = py; h = ph;
                known_free = prev_free;
                ua = ua_prev;
                have_ua = prev_have_ua;
                have_cs = false;
                break;
            }
            cur_free = true;
            known_free = true;
        }
    }
    t += 1;                                                             // :277

    if (!have_cs) sincos(h, &sn, &ct);
    Box box = make_box(x, y, ct, sn);

    // ---- status (:279-282, _check_status :175-184) -------------------------------------------------
    int status = HOPE_STATUS_CONTINUE;
    if (p.stages & (HOPE_STAGE_REWARD | HOPE_STAGE_RS)) {
        if (arrive) status = HOPE_STATUS_ARRIVED;
        else {
            const double xmin = sc[SC_BBOX], xmax = sc[SC_BBOX + 1], ymin = sc[SC_BBOX + 2], ymax = sc[SC_BBOX + 3];
            bool coll = known_free ? false : detect_collision(box, tile, n_slots, lane);
            if (coll) status = HOPE_STATUS_COLLIDED;
            else if (x > xmax || x < xmin || y > ymax || y < ymin) status = HOPE_STATUS_OUTBOUND;
            else {
                if (!have_ua) { ua = overlap_area(box, dbox, scr + LDS_SH, lane); have_ua = true; }
                if (ua / dest_area > 0.95) status = HOPE_STATUS_ARRIVED;
                else if (t > TOLERANT_TIME) status = HOPE_STATUS_OUTTIME;
            }
        }
    }

    // ---- reward (_get_reward :186-227, reward_shaping env_wrapper.py:10-35) ------------------------
    double ri0 = 0, ri2 = 0, ri3 = 0, ri4 = 0, reward = 0;
    if (p.stages & HOPE_STAGE_REWARD) {
        if (status == HOPE_STATUS_CONTINUE) {
            if (!have_ua) { ua = overlap_area(box, dbox, scr + LDS_SH, lane); have_ua = true; }
            ri0 = -tanh((double)t / (10 * TOLERANT_TIME));
            double ddx = x - destx, ddy = y - desty;
            double dist_diff = sqrt(ddx * ddx + ddy * ddy);
            double pdx = prev_x - destx, pdy = prev_y - desty;
            double prev_dist_diff = sqrt(pdx * pdx + pdy * pdy);
            double ad = acos(cos(h - desth));
            ad = ad < PI / 2 ? ad : PI - ad;
            double pad = acos(cos(prev_h - desth));
            pad = pad < PI / 2 ? pad : PI - pad;
            const double dnorm = sc[SC_DNORM];
            ri2 = prev_dist_diff / dnorm - dist_diff / dnorm;
            ri3 = pad / PI - ad / PI;
            double bur = ua / (2 * dest_area - ua);
            if (bur < accum) bur = 0;
            else { double pa = accum; accum = bur; bur -= pa; }
            ri4 = bur;
            double rw = 0;
            rw += 1 * ri0; rw += 0 * 0.0; rw += 5 * ri2; rw += 0 * ri3; rw += 10 * ri4;
            reward = rw;
        } else if (status == HOPE_STATUS_OUTBOUND) reward = -50;
        else if (status == HOPE_STATUS_OUTTIME) reward = -1;
        else if (status == HOPE_STATUS_ARRIVED) reward = 50;
        else if (status == HOPE_STATUS_COLLIDED) reward = -50;
        reward *= 0.1;
    }

    // ---- write state + scalar outputs ---------------------------------------------------------------
    if (lane == 0) {
        st[0] = x; st[1] = y; st[2] = h; st[3] = accum;
        p.tstep[scene] = t;
        if (p.out.pose) { p.out.pose[3 * (size_t)scene] = x; p.out.pose[3 * (size_t)scene + 1] = y; p.out.pose[3 * (size_t)scene + 2] = h; }
        if (p.stages & HOPE_STAGE_REWARD) {
            if (p.out.status) p.out.status[scene] = status;
            if (p.out.done) p.out.done[scene] = status != HOPE_STATUS_CONTINUE;
            if (p.out.reward) ((OT*)p.out.reward)[scene] = (OT)reward;
            if (p.out.reward_info) {
                OT* ri = (OT*)p.out.reward_info + 5 * (size_t)scene;
                ri[0] = (OT)ri0; ri[1] = (OT)0; ri[2] = (OT)ri2; ri[3] = (OT)ri3; ri[4] = (OT)ri4;
            }
        }
        if (p.out.rs_word) {   // cleared here; the Reeds-Shepp kernel fills it for eligible scenes
            int8_t* w = p.out.rs_word + 8 * (size_t)scene;
            w[0] = w[1] = w[2] = w[3] = w[4] = HOPE_RS_NONE; w[5] = 0; w[6] = 0; w[7] = 0;
        }
        if ((p.stages & HOPE_STAGE_RS) && t > 1 && status == HOPE_STATUS_CONTINUE) {       // gate :293-294
            double ddx = x - destx, ddy = y - desty;
            if (sqrt(ddx * ddx + ddy * ddy) < RS_MAX_DIST) {
                int slot = atomicAdd(p.rs_count, 1);
                p.rs_list[slot] = scene;
            }
        }
    }
    if (p.out.rs_lengths && lane < 5) ((OT*)p.out.rs_lengths)[5 * (size_t)scene + lane] = (OT)0;

    if (!(p.stages & HOPE_STAGE_OBS)) return;

    // ---- target representation (_get_targt_repr :372-381; 5th entry is cos again) ---------------
    if (p.out.target && lane == 0) {
        double rdx = destx - x, rdy = desty - y;
        double rel_distance = sqrt(rdx * rdx + rdy * rdy);
        double rel_angle = atan2(rdy, rdx) - h;
        double rel_dest_heading = desth - h;
        OT* tg = (OT*)p.out.target + 5 * (size_t)scene;
        tg[0] = (OT)rel_distance;
        tg[1] = (OT)cos(rel_angle);
        tg[2] = (OT)sin(rel_angle);
        tg[3] = (OT)cos(rel_dest_heading);
        tg[4] = (OT)cos(rel_dest_heading);
    }

    // ---- lidar (lidar_simulator.py:31-135) -----------------------------------------------------------
    // world -> ego in place: affine [a, b, -b, a, x_off, y_off] (:58-64)
    {
        const double a = ct, b = sn;
        const double x_off = -x * a - y * b;
        const double y_off = x * b - y * a;
        for (int v = lane; v < n_slots; v += WAVE) {
            double px = tile[2 * v], py = tile[2 * v + 1];
            tile[2 * v] = a * px + b * py + x_off;
            tile[2 * v + 1] = (-b) * px + a * py + y_off;
        }
    }
    wsync();
    // ring kept iff distance(ring, origin) < lidar_range (:69); 4 consecutive lanes = one ring
    for (int base = 0; base < n_slots; base += WAVE) {
        int e = base + lane;
        double dd = INFINITY;
        if (e < n_slots) {
            int e2 = (e & ~3) | ((e + 1) & 3);
            dd = origin_seg_dist(tile[2 * e], tile[2 * e + 1], tile[2 * e2], tile[2 * e2 + 1]);
        }
        dd = fmin(dd, __shfl_xor(dd, 1));
        dd = fmin(dd, __shfl_xor(dd, 2));
        if (e < n_slots && (e & 3) == 0) keep[e >> 2] = dd < LIDAR_RANGE;
    }
    wsync();
    // beams: lane l owns beams l and l+64
    const int i0 = lane, i1 = lane + 64;
    const bool has1 = i1 < NBEAM;
    const double a0 = p.beam_ab[2 * i0], b0 = p.beam_ab[2 * i0 + 1];
    const double a1 = has1 ? p.beam_ab[2 * i1] : 0.0, b1 = has1 ? p.beam_ab[2 * i1 + 1] : 0.0;
    double best0 = INFINITY, best1 = INFINITY;
    for (int r = 0; r < n_obst; r++) {
        if (!keep[r]) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int e = 4 * r + j, e2 = 4 * r + ((j + 1) & 3);
            double x1 = tile[2 * e], y1 = tile[2 * e + 1], x2 = tile[2 * e2], y2 = tile[2 * e2 + 1];
            double d = y2 - y1, ee = x1 - x2, f = y1 * x2 - x1 * y2;
            best0 = fmin(best0, beam_edge(i0, a0, b0, x1, y1, x2, y2, d, ee, f));
            if (has1) best1 = fmin(best1, beam_edge(i1, a1, b1, x1, y1, x2, y2, d, ee, f));
        }
    }
    const double base0 = p.hull_base[i0], base1 = has1 ? p.hull_base[i1] : 0.0;
    const double lid0 = clipd(best0, 0, LIDAR_RANGE) - base0;      // get_observation :46
    const double lid1 = clipd(best1, 0, LIDAR_RANGE) - base1;
    if (p.out.lidar) {
        OT* lo = (OT*)p.out.lidar + (size_t)NBEAM * scene;
        lo[i0] = (OT)lid0;
        if (has1) lo[i1] = (OT)lid1;
    }
    if (!p.out.action_mask) return;

    // ---- action mask (action_mask.py:166-196) ----------------------------------------------------------
    double* xs = scr + LDS_X;                                     // lidar_obs = clip(raw,0,10) + base (:170)
    xs[i0] = clipd(lid0, 0, 10) + base0;
    if (has1) xs[i1] = clipd(lid1, 0, 10) + base1;
    wsync();
    if (lane == 0) xs[NBEAM] = xs[0];                             // circular (:158)
    wsync();
    int mstep = NITER;                                            // lane a: min over beams of first-exceed index
    for (int r = 0; r < (NL + WAVE - 1) / WAVE; r++) {
        int l = r * WAVE + lane;
        double dl = 0;
        bool act = false;
        if (l < NL) {
            int i = l / UPS, j = l % UPS;
            double w2 = (double)j / UPS, w1 = 1 - w2;
            dl = xs[i] * w1 + xs[i + 1] * w2;                     // _linear_interpolate (:161-162)
            act = dl < p.pmax[l];
        }
        unsigned long long m = __ballot(act);
        while (m) {
            int bpos = __ffsll((long long)m) - 1;
            m &= m - 1;
            int ll = r * WAVE + bpos;
            double d_ll = __shfl(dl, bpos);
            if (lane < NACT) {
                const double* row = p.tab + (size_t)ll * NITER * NACT + lane;
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < NITER; k++) cnt += (row[k * NACT] <= d_ll) ? 1 : 0;
                mstep = min(mstep, cnt);
            }
        }
    }
    // post_process (:186-196): ends of each direction half -1, min filter (5, reflect), clip, /10
    int v = mstep;
    if (lane == 0 || lane == NACT / 2 - 1 || lane == NACT / 2 || lane == NACT - 1) v -= 1;
    const int half = NACT / 2;
    const int hb_ = lane < half ? 0 : half;
    const int li = lane - hb_;
    int mn = v;
#pragma unroll
    for (int off = -2; off <= 2; off++) {
        int j = li + off;
        if (j < 0) j = -j - 1;
        if (j >= half) j = 2 * half - 1 - j;
        int src = hb_ + j;
        int o = __shfl(v, src < NACT ? src : 0);
        mn = min(mn, o);
    }
    mn = max(0, min(NITER, mn));
    double mo = (double)mn / NITER;
    unsigned long long nz = __ballot(lane < NACT && mn > 0);
    if (nz == 0) mo = clipd(mo, 0.01, 1);                          // all-zero -> 0.01 (:182-183)
    if (lane < NACT) ((OT*)p.out.action_mask)[(size_t)NACT * scene + lane] = (OT)mo;
}

}  // namespace hope
