// hope_motion_pair.h -- the motion launch of the SMALL-TILE class (scenes of <= 32 obstacles) with TWO SCENES PER WAVEFRONT: lanes
// 0..31 step list entry 2 b, lanes 32..63 entry 2 b + 1.  Same results, bit for bit, as k_env_step<.., PART 1> (hope_step_kernel.h),
// which stays the large-tile class's motion launch, the reset observation's, and this kernel's reference (stage bit 0x8000 selects it:
// tests/test_gpu_parity.py).  Replaces, per scene (reference file:line):
//   CarParking.step's sub-step loop and retreat   car_parking_base.py:255-277 (the ten poses come from k_kinematics)
//   _check_arrived / _detect_collision / _check_status   car_parking_base.py:153-184
//   the fused episode turnover (CarParking.reset :128-138 on a new map from the device-resident pool + its action-less step)
//
// Why (round 6).  A wave of the motion launch executes ~165 vector instructions for its scene and spends the rest of its ~5 us in four
// dependent memory round trips (list entry -> scene header / sub-step poses / obstacle boxes -> near obstacles' vertices -> outputs) at
// 128 registers, i.e. four waves per SIMD, which it shares with the previous step's validation kernel: the launch is bound by wave
// slots x latency.  Two scenes share every round trip of a wave -- half the waves for the same scenes.
//   * near-obstacle scan: one lane per obstacle slot of its half (32 = the tile capacity), ballots split into their 32-bit halves;
//   * sub-step loop: lane = (sub-step, edge slot) INSIDE a half, S = 4 .. 32 edge slots per sub-step and G = 32 / S sub-steps per pass
//     (more than eight near obstacles: passes of 32 edge slots per sub-step); the two halves run their passes in lockstep, each with
//     its own counters;
//   * arrival: the exact slab bound comes from k_kinematics (bit k of the record's mask; bit 10 = the pose the step starts from), the
//     polygon clip runs on lane 0 of each half -- both halves' clips at once when both need one;
//   * the robust path (an orientation inside its error bound) keeps its one-lane-at-a-time loop over the whole wave and ONE work area;
//   * episode turnover (~0.5 % of the scene-steps): by the WHOLE wave in the one-scene kernel's layout, one half after the other, with the
//     one-scene kernel's own helpers (hope_step_kernel.h) -- the new lot has at most 32 obstacles (pool class 0) and fits the half's tile.
// LDS per wave 6.7 KB.  All synchronisation is LDS-only (ssync).
#pragma once
#include "hope_obs_pair.h"

namespace hope {

// per-half LDS block (doubles): tile[8 * OP_CAP] | kin record [56] (h cos sin x y of the ten poses, arrival bits, hull box) | dest box [8] |
// sh[64] clip scratch | near list [OP_CAP] i32
constexpr int MP_TILE = 0, MP_KIN = 8 * OP_CAP, MP_DBOX = MP_KIN + KIN_WORDS, MP_SH = MP_DBOX + 8, MP_LIST = MP_SH + 64,
              MP_START = MP_LIST + OP_CAP / 2,      // x, y, heading the step starts from, |dest box|
              MP_HALF_W = MP_START + 4;
constexpr size_t MP_LDS_BYTES = (size_t)(2 * MP_HALF_W + ROBUST_LDS_WORDS) * 8;
static_assert(KIN_WORDS <= 2 * OP_HALF, "two loads per lane fetch a scene's kinematics record");

// (hull of pose q) x (obstacle edge e): the orientation filter's verdict for this lane's pair -- hit (certain), und (undecided)
__device__ __forceinline__ void mp_edge_test(double qx, double qy, double qc, double qs, double ex1, double ey1, double ex2, double ey2,
                                             bool& hit, bool& und) {
    const Box b = make_box(qx, qy, qc, qs);
    const double hminx = fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3])), hmaxx = fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3]));
    const double hminy = fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3])), hmaxy = fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3]));
    if (!(fmin(ex1, ex2) > hmaxx || fmax(ex1, ex2) < hminx || fmin(ey1, ey2) > hmaxy || fmax(ey1, ey2) < hminy)) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int c2 = (c + 1) & 3;
            const int r = segments_intersect_fast(b.x[c], b.y[c], b.x[c2], b.y[c2], ex1, ey1, ex2, ey2);
            hit = hit || r == 1;
            und = und || r == 2;
        }
    }
}

// |hull(pose) ∩ dest box| for the halves whose `want` is set (half-uniform): exact quick reject, then Sutherland-Hodgman on lane 0 of
// the half (overlap_area of the one-scene kernel, same expressions); the value is broadcast inside the half
__device__ __forceinline__ double mp_overlap_area(bool want, double px, double py, double ct, double st, const double* dbox, double* sh,
                                                  int hl, int hw) {
    double area = 0.0;
    if (want) {
        const Box box = make_box(px, py, ct, st);
        const double cx = 0.5 * (box.x[0] + box.x[2]), cy = 0.5 * (box.y[0] + box.y[2]);
        const double dx = 0.5 * (dbox[0] + dbox[4]) - cx, dy = 0.5 * (dbox[1] + dbox[5]) - cy;
        const double reach = 5.2;
        if (!(dx * dx + dy * dy > reach * reach) && hl == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) { sh[i] = box.x[i]; sh[16 + i] = box.y[i]; }
            area = quad_intersection_area_lane0(dbox, sh);
        }
    }
    return __shfl(area, hw << 5);
}

// Fused episode turnover of ONE scene by the WHOLE wave (HOPE_AUTO_RESET; the one-scene kernel's code, hope_step_kernel.h:1089-1170):
// with HOPE_AUTO_REDRAW a new lot from the device-resident pool (map.reset, car_parking_base.py:134) copied into the scene's slots, then
// CarParking.reset's state and the status of its action-less step.  tb = the LDS block of the scene's half (tile, dest box, lists: the
// finished step's contents are dead).  Leaves x, y, heading, cos, sin, accum_arrive_reward of the new episode in tb[MP_KIN .. + 5].
// Out of line on purpose: ~0.5 % of the scene-steps take it, and inlined its temporaries set the kernel's register count.
__device__ __noinline__ void mp_turnover(const double* verts, const float4* obb, const double* scene_c, const int32_t* n_obst_g,
                                         const StepCold* cp, int max_obst, uint32_t stages, int tscene, double* tb, double* xl, int lane) {
    double* ttile = tb + MP_TILE;
    double* tdbox = tb + MP_DBOX;
    int* tlist = (int*)(tb + MP_LIST);
    const double* tsc = scene_c + (size_t)tscene * SC_WORDS;
    int nob = min(n_obst_g[tscene], OP_CAP);
    bool redrawn = false;
    ssync();
    if (stages & HOPE_AUTO_REDRAW) {
        const int cls = cp->slot_cls[tscene] ? 1 : 0;
        const int cnt = cp->pool_cls_n[cls];
        if (cnt > 0 && (cp->pool_verts || cp->dlp.n_cases > 0)) {
            const uint32_t ep = cp->episode[tscene];
            const uint64_t key = mix64(cp->redraw_seed ^ mix64(((uint64_t)tscene << 32) | ep));
            const int j = cp->pool_cls[cls][(int)(key % (uint64_t)cnt)];
            if (j >= 0 && cp->pool_nobst[j] <= OP_CAP) {            // a complete lot of the pool (this class's lists hold nothing else)
                nob = cp->pool_nobst[j];
                const double* pv = cp->pool_verts + (size_t)j * max_obst * 8;
                const double2* psrc = (const double2*)pv;
                double2* gdst = (double2*)(const_cast<double*>(verts) + (size_t)tscene * max_obst * 8);
                float4* gobb = const_cast<float4*>(obb) + (size_t)tscene * max_obst;
                double* gsc = const_cast<double*>(scene_c) + (size_t)tscene * SC_WORDS;
                double2* ldst = (double2*)ttile;
                for (int v = lane; v < 4 * nob; v += WAVE) { const double2 q2 = psrc[v]; gdst[v] = q2; ldst[v] = q2; }
                const double* pc = cp->pool_c + (size_t)j * SC_WORDS;
                {
                    const double fox = pc[SC_BBOX], foy = pc[SC_BBOX + 2];
                    float4* gfv = cp->fverts + (size_t)tscene * max_obst * 2;
                    float4* gfb = cp->fbox + (size_t)tscene * max_obst;
                    uint8_t* gfl = cp->eflag + (size_t)tscene * eflag_stride(max_obst);
                    for (int o = lane; o < nob; o += WAVE) {
                        gobb[o] = obstacle_box(pv + (size_t)o * 8);
                        obstacle_f32(pv + (size_t)o * 8, fox, foy, gfv + 2 * o, gfb + o, gfl + o);
                    }
                }
                if (lane < SC_WORDS) gsc[lane] = pc[lane];
                tsc = pc;                                           // the new scene's constants, straight from the pool
                if (lane == 0) {
                    const_cast<int32_t*>(n_obst_g)[tscene] = nob; cp->cur_pool[tscene] = j; cp->episode[tscene] = ep + 1;
                    if (cp->layer_valid) cp->layer_valid[tscene] = 0;
                }
                if (lane < 8) tdbox[lane] = tsc[SC_DBOX + lane];
                redrawn = true;
            } else if (lane == 0 && cp->pool_overflow) atomicAdd(cp->pool_overflow, 1);   // (cannot happen: class-0 lists hold lots of <= 32 obstacles)
        }
    }
    const double tarea = tsc[SC_DAREA];
    const double nx = tsc[SC_START], ny = tsc[SC_START + 1], nh = tsc[SC_START + 2];
    double nsn, nct;
    hm_sincos(nh, &nsn, &nct);
    double nacc = 0.0;
    // the action-less step's status decides whether _get_reward runs (it only touches accum_arrive_reward)
    ssync();
    const double2* tsrc = (const double2*)(verts + (size_t)tscene * max_obst * 8);
    const int n_near0 = redrawn ? build_near_list(ttile, nob, nx, ny, 3.9, tlist, lane)
                                : stage_near(obb + (size_t)tscene * max_obst, tsrc, nob, nx - 3.9, nx + 3.9, ny - 3.9, ny + 3.9, ttile, tlist, lane);
    ssync();
    const double bx0 = tsc[SC_BBOX], bx1 = tsc[SC_BBOX + 1], by0 = tsc[SC_BBOX + 2], by1 = tsc[SC_BBOX + 3];
    const bool cont = !detect_collision(nx, ny, nct, nsn, ttile, tlist, n_near0, xl, lane) && !(nx > bx1 || nx < bx0 || ny > by1 || ny < by0);
    if (cont) {
        const double ua0 = overlap_area(nx, ny, nct, nsn, tdbox, tb + MP_SH, lane);
        if (!(ua0 / tarea > 0.95)) {                                // not ARRIVED (and t = 1 is not OUTTIME): CONTINUE
            const double bur = ua0 / (2 * tarea - ua0);
            if (!(bur < nacc)) nacc = bur;                          // :221-226 with accum = 0
        }
    }
    ssync();
    if (lane == 0) { double* o = tb + MP_KIN; o[0] = nx; o[1] = ny; o[2] = nh; o[3] = nct; o[4] = nsn; o[5] = nacc; }
    ssync();
}

#ifndef HOPE_MP_PRIO
#define HOPE_MP_PRIO 0        // wave priority of k_motion_pair (2 / 3 measured: see DESIGN 9a)
#endif
#ifndef HOPE_MP_OCC
#define HOPE_MP_OCC 4         // waves per SIMD k_motion_pair is compiled for
#endif
__global__ __launch_bounds__(64, HOPE_MP_OCC) void k_motion_pair(StepParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
#if HOPE_MP_PRIO
    __builtin_amdgcn_s_setprio(HOPE_MP_PRIO);               // (A/B: the launch shares its SIMDs with the validation kernel, which runs at priority 1)
#endif
    const int lane = threadIdx.x, hw = lane >> 5, hl = lane & (OP_HALF - 1);
    const int n_pairs = (p.n_list + 1) >> 1;
    if ((int)blockIdx.x >= n_pairs) return;
    if (p.rs_count_zero && blockIdx.x == 0 && lane == 0) p.rs_count_zero[0] = 0;   // this class's queue length, for the k_rs_compact that follows
    const int li = 2 * scene_of_block(blockIdx.x, n_pairs) + hw;
    bool live = li < p.n_list;
    const int scene = p.scene_list[live ? li : li - 1];
    {
        const bool act = !(p.active && !p.active[scene]);
        if (live && p.active_out && hl == 0) p.active_out[scene] = act;
        live = live && act;
    }
    if (!__any(live)) return;

    double* hb = lds + hw * MP_HALF_W;
    double* tile = hb + MP_TILE;
    double* kin = hb + MP_KIN;                      // [0..9] h, [10..19] cos, [20..29] sin, [30..39] x, [40..49] y, [50] bits, [52..55] hull box
    double* dbox = hb + MP_DBOX;
    double* sh = hb + MP_SH;
    int* klist = (int*)(hb + MP_LIST);
    double* xl = lds + 2 * MP_HALF_W;               // the robust path's work area: one lane of the wave at a time

    // ---- the scene's first loads, all requested at once: obstacle count, sub-step poses, this lane's obstacle box, constants, state
    const int n_obst = live ? min(p.n_obst[scene], OP_CAP) : 0;
    const double* kr = p.kin + (size_t)scene * KIN_WORDS;
    const double kv0 = kr[hl], kv1 = hl < KIN_WORDS - OP_HALF ? kr[OP_HALF + hl] : 0.0;
    // (the obstacle boxes of the first OP_EAGER slots with the first loads -- generated lots have at most 13 obstacles --, the others only
    // for a scene that has them: 256 instead of 512 bytes per scene and launch)
    const float4* obb_s = p.obb + (size_t)scene * p.max_obst;
    float4 bb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (hl < OP_EAGER) bb = obb_s[hl];
    const double* sc = p.scene_c + (size_t)scene * SC_WORDS;
    const double dbv = hl < 8 ? sc[SC_DBOX + hl] : 0.0;
    double* st = p.state + (size_t)scene * ST_WORDS;
    // lane hl < 3 of the half: word hl of the state (x, y, heading); lane 3: |dest box| -- parked in LDS, the registers are needed below
    const double stv = hl < 3 ? st[hl] : sc[SC_DAREA];
    int t = p.tstep[scene];
    const double2* src = (const double2*)(p.verts + (size_t)scene * p.max_obst * 8);
    double* pr = p.post + (size_t)scene * POST_WORDS;
    if (live && hl < 3) pr[hl] = stv;                               // prev_state (car_parking_base.py:255), k_post's hand-over
    double* start = hb + MP_START;
    if (hl < 4) start[hl] = stv;

    if (hl >= OP_EAGER && hl < n_obst) bb = obb_s[hl];            // (behind every other first load: this one waits for the count)
    kin[hl] = kv0;
    if (hl < KIN_WORDS - OP_HALF) kin[OP_HALF + hl] = kv1;
    if (hl < 8) dbox[hl] = dbv;
    ssync();
    const int apmask = live ? __double2loint(kin[50]) : 0;          // arrival_possible of the ten poses (bits 0..9) and of the start pose (bit 10)
    // ---- near obstacles: those whose box (float32, rounded outwards: a superset) meets the box around every hull of this step
    int n_near;
    {
        const double bx0 = kin[52], bx1 = kin[53], by0 = kin[54], by1 = kin[55];
        const bool near = hl < n_obst && !((double)bb.x > bx1 || (double)bb.y < bx0 || (double)bb.z > by1 || (double)bb.w < by0);
        const unsigned long long m = __ballot(near);
        const unsigned hm = hw ? (unsigned)(m >> 32) : (unsigned)m;
        n_near = __popc(hm);
        if (near) klist[__popc(hm & ((1u << hl) - 1))] = hl;
    }
    ssync();
    for (int i = hl; i < 4 * n_near; i += OP_HALF) ((double2*)tile)[i] = src[4 * klist[i >> 2] + (i & 3)];   // tile slot k = the k-th near obstacle
    ssync();

    // ---- sub-step loop (car_parking_base.py:259-271): "arrived? -> stop; collided? -> retreat and stop", in sub-step order
    const int E = 4 * n_near;                                       // edge slots of this half's scene
    const int S = E <= 4 ? 4 : (E <= 8 ? 8 : (E <= 16 ? 16 : 32));  // edge slots per sub-step in one pass, G = 32 / S sub-steps per pass
    const int lgS = E <= 4 ? 2 : (E <= 8 ? 3 : (E <= 16 ? 4 : 5));
    const int G = OP_HALF >> lgS;
    const int g = hl >> lgS, e0 = hl & (S - 1);
    const int nch = E <= OP_HALF ? 1 : (E + OP_HALF - 1) / OP_HALF; // passes over the edges per sub-step (more than 8 near obstacles)
    const int nch_max = max(__builtin_amdgcn_readlane(nch, 0), __builtin_amdgcn_readlane(nch, OP_HALF));
    // (a half with no near obstacle and no sub-step pose that can reach the dest box has nothing to walk: the step ends at the tenth pose)
    int ev_k = NUM_STEP, k0 = (E == 0 && (apmask & ((1 << NUM_STEP) - 1)) == 0) ? NUM_STEP : 0;
    bool ev_arrive = false;
    double ua = 0.0;
    const double dest_area = start[3];
    while (__any(live && k0 < NUM_STEP && ev_k == NUM_STEP)) {
        const bool on = live && k0 < NUM_STEP && ev_k == NUM_STEP;  // this half still walks
        const int k = k0 + g;
        const bool kv = on && k < NUM_STEP;
        const int kk = k < NUM_STEP ? k : NUM_STEP - 1;
        const double qx = kin[30 + kk], qy = kin[40 + kk], qc = kin[10 + kk], qs = kin[20 + kk];
        unsigned hacc = 0;                                          // lanes of this half with a hit, over the edge passes
        for (int ch = 0; ch < nch_max; ch++) {
            const int e = ch * OP_HALF + e0;
            const bool has_edge = kv && ch < nch && e < E;
            bool hit = false, und = false;
            double ex1 = 0, ey1 = 0, ex2 = 0, ey2 = 0;
            if (has_edge) {
                const double* v = tile + 8 * (e >> 2);
                const int j = e & 3, j2 = (e + 1) & 3;
                ex1 = v[2 * j]; ey1 = v[2 * j + 1]; ex2 = v[2 * j2]; ey2 = v[2 * j2 + 1];
                mp_edge_test(qx, qy, qc, qs, ex1, ey1, ex2, ey2, hit, und);
            }
            {   // pairs the orientation filter left open (practically never): the robust path, one lane at a time
                unsigned long long um = __ballot(und && !hit);
                while (um) {
                    const int l = __ffsll((long long)um) - 1;
                    um &= um - 1;
                    if (lane == l) hit = hull_edge_intersect_robust(qx, qy, qc, qs, ex1, ey1, ex2, ey2, xl);
                }
            }
            const unsigned long long hm = __ballot(hit);
            hacc |= hw ? (unsigned)(hm >> 32) : (unsigned)hm;
        }
        // walk this pass's sub-steps in order
        const int gmax = max(__builtin_amdgcn_readlane(on ? G : 0, 0), __builtin_amdgcn_readlane(on ? G : 0, OP_HALF));
        const unsigned gm0 = S == 32 ? 0xFFFFFFFFu : ((1u << S) - 1u);
        // (only when this pass holds an event candidate: a certain hit, or a pose the slab bound lets arrive)
        const bool cand = on && (hacc != 0 || (((unsigned)apmask >> k0) & ((1u << G) - 1u)) != 0);
        if (__any(cand))
        for (int gg = 0; gg < gmax; gg++) {
            const int kq = k0 + gg;
            const bool walk = on && gg < G && kq < NUM_STEP && ev_k == NUM_STEP;
            const int kqq = kq < NUM_STEP ? kq : NUM_STEP - 1;
            const bool want_a = walk && ((apmask >> kqq) & 1);      // _check_arrived :164-170 (the slab bound says it is possible)
            if (__any(want_a)) {
                const double a_ = mp_overlap_area(want_a, kin[30 + kqq], kin[40 + kqq], kin[10 + kqq], kin[20 + kqq], dbox, sh, hl, hw);
                if (want_a) {
                    ua = a_;
                    if (ua / dest_area > 0.95) { ev_k = kq; ev_arrive = true; }
                }
            }
            if (walk && ev_k == NUM_STEP && (hacc & (gm0 << (gg << lgS)))) ev_k = kq;     // _detect_collision :264
        }
        if (on) k0 += G;
    }
    // final pose of the motion: the arrival pose, the pose BEFORE the colliding sub-step (retreat :264-271), or the tenth pose
    const bool arrive = ev_arrive;
    const int kf = ev_arrive ? ev_k : (ev_k == NUM_STEP ? NUM_STEP - 1 : ev_k - 1);   // -1: the pose the step started from
    const bool moved = kf >= 0;
    double x, y, h, ct = 0.0, sn = 0.0;
    if (kf >= 0) { x = kin[30 + kf]; y = kin[40 + kf]; h = kin[kf]; ct = kin[10 + kf]; sn = kin[20 + kf]; }
    else { x = start[0]; y = start[1]; h = start[2]; }
    const bool known_free = kf >= 0 && !ev_arrive;                  // passed its own collision test in the loop
    bool have_ua = ev_arrive;
    t += 1;                                                         // :277
    if (__any(live && kf < 0)) {                                    // blocked at the first sub-step: the start pose's heading
        double s_, c_;
        hm_sincos(h, &s_, &c_);
        if (kf < 0) { sn = s_; ct = c_; }
    }

    // ---- status (:279-282, _check_status :175-184)
    int status = HOPE_STATUS_CONTINUE;
    if (p.stages & (HOPE_STAGE_REWARD | HOPE_STAGE_RS)) {
        bool coll = false;
        const bool need_c = live && !arrive && !known_free;
        if (__any(need_c)) {                                        // the start pose against the near obstacles, 32 edge slots per pass
            unsigned hacc = 0;
            for (int ch = 0; ch < nch_max; ch++) {
                const int e = ch * OP_HALF + hl;
                const bool has_edge = need_c && e < E;
                bool hit = false, und = false;
                double ex1 = 0, ey1 = 0, ex2 = 0, ey2 = 0;
                if (has_edge) {
                    const double* v = tile + 8 * (e >> 2);
                    const int j = e & 3, j2 = (e + 1) & 3;
                    ex1 = v[2 * j]; ey1 = v[2 * j + 1]; ex2 = v[2 * j2]; ey2 = v[2 * j2 + 1];
                    mp_edge_test(x, y, ct, sn, ex1, ey1, ex2, ey2, hit, und);
                }
                unsigned long long um = __ballot(und && !hit);
                while (um) {
                    const int l = __ffsll((long long)um) - 1;
                    um &= um - 1;
                    if (lane == l) hit = hull_edge_intersect_robust(x, y, ct, sn, ex1, ey1, ex2, ey2, xl);
                }
                const unsigned long long hm = __ballot(hit);
                hacc |= hw ? (unsigned)(hm >> 32) : (unsigned)hm;
            }
            coll = need_c && hacc != 0;
        }
        const bool inside = (apmask >> (16 + (kf >= 0 ? kf : NUM_STEP))) & 1;       // the pose's OUTBOUND test, by k_kinematics
        // arrival of the final pose: already known (the loop stopped on it), or possible by the slab bound -> the clip
        const bool want_a = live && !arrive && !coll && inside && !have_ua && ((apmask >> (kf >= 0 ? kf : NUM_STEP)) & 1);
        if (__any(want_a)) {
            const double a_ = mp_overlap_area(want_a, x, y, ct, sn, dbox, sh, hl, hw);
            if (want_a) { ua = a_; have_ua = true; }
        }
        if (arrive) status = HOPE_STATUS_ARRIVED;
        else if (coll) status = HOPE_STATUS_COLLIDED;
        else if (!inside) status = HOPE_STATUS_OUTBOUND;
        else if (have_ua && ua / dest_area > 0.95) status = HOPE_STATUS_ARRIVED;
        else if (t > TOLERANT_TIME) status = HOPE_STATUS_OUTTIME;
    }
    // ---- fused episode turnover (HOPE_AUTO_RESET) wanted?  The hand-over to k_post describes the FINISHED step: written now
    const bool turnover = live && (p.stages & HOPE_AUTO_RESET) && (p.stages & HOPE_STAGE_REWARD) && status != HOPE_STATUS_CONTINUE;
    if (live && hl == 0) {
        const bool need_ua = (p.stages & HOPE_STAGE_REWARD) && status == HOPE_STATUS_CONTINUE && !have_ua;   // k_post clips (lane per scene)
        pr[3] = x; pr[4] = y; pr[5] = h; pr[6] = ua;
        const int fl = ((p.stages & HOPE_STAGE_REWARD) ? POST_F_REWARD : 0) | (turnover ? POST_F_TURNOVER : 0) | (need_ua ? POST_F_NEED_UA : 0);
        pr[7] = __hiloint2double(fl, status | (t << 8));
    }
    double accum = 0.0;                                             // written only on a turnover (else the state word stays)

    // ---- the turnover: by the whole wave, one half's scene after the other (the one-scene kernel's code)
    const unsigned long long tmask = __ballot(turnover);
    if (tmask) {
        for (int hsel = 0; hsel < 2; hsel++) {
            if (!((tmask >> (OP_HALF * hsel)) & 1)) continue;
            double* tb = lds + hsel * MP_HALF_W;
            mp_turnover(p.verts, p.obb, p.scene_c, p.n_obst, p.cold, p.max_obst, p.stages, __builtin_amdgcn_readlane(scene, OP_HALF * hsel), tb, xl, lane);
            if (hw == hsel) { x = tb[MP_KIN]; y = tb[MP_KIN + 1]; h = tb[MP_KIN + 2]; ct = tb[MP_KIN + 3]; sn = tb[MP_KIN + 4]; accum = tb[MP_KIN + 5]; t = 1; }
        }
    }

    // ---- write state + scalar outputs
    if (live && hl == 0) {
        st[0] = x; st[1] = y; st[2] = h;
        if (turnover) st[3] = accum;
        p.cs[2 * (size_t)scene] = ct; p.cs[2 * (size_t)scene + 1] = sn;
        p.tstep[scene] = t;
        if (p.hflags & STEP_HF_TRAJ) {
            // vehicle.trajectory: of the sub-step states only the last kept one stays (car_parking_base.py:259-276, vehicle.py:144,158);
            // a step blocked at its first sub-step adds nothing; reset leaves [start]
            const StepCold* cp = p.cold;
            double* tr = cp->traj + (size_t)scene * 60;
            int32_t* tlen = cp->traj_len;
            const int tl = turnover ? 0 : tlen[scene];
            if (turnover) cp->traj_valid[scene] = 0;
            if (turnover || moved) {
                double* e = tr + 3 * (tl % 20);
                e[0] = x; e[1] = y; e[2] = h;
                tlen[scene] = tl + 1;
            }
        }
    }
}

}  // namespace hope
