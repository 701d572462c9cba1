// hope_obs_pair.h -- the observation launch of the SMALL-TILE class (scenes of <= 32 obstacles: three quarters of the headline
// mix) with TWO SCENES PER WAVEFRONT: lanes 0..31 work on list entry 2 b, lanes 32..63 on entry 2 b + 1.
//
// Same outputs, bit for bit, as k_env_step<OT, AT, false, 2> (hope_step_kernel.h), which stays the large-tile class's
// observation launch and this kernel's reference (tests/test_gpu_parity.py: pair form vs single form, vs the oracle):
//   LidarSimlator.get_observation   lidar_simulator.py:31-135
//   ActionMask.get_steps / post_process   action_mask.py:166-196
//
// Why (round 6).  A generated lot has ~7 obstacles = 28 edges: the per-edge phases of a one-scene wave (ego transform, ring keep,
// beam spans) ran with 28 of 64 lanes, the mask stage with 42.  More important than the idle lanes: the step is bound by
// wave-slot x latency, not by issue slots (each kernel alone already fills the machine, concurrent launches add up to the sum of
// their solo durations, and a wave of this launch spends most of its life in ~12 dependent memory round trips -- halving the
// waves per CU of the observation launch does not change the step time, profiles/r06_ab_obs_concurrency.txt).  Two scenes share
// every round trip of the wave: half the waves for the same scenes at about the same wave lifetime.
//   * per-edge phases: each half-wave walks its own scene's edges (ballots split into their 32-bit halves);
//   * the (beam, edge) pair queue and its drain are SHARED: an entry carries the half, a drain chunk holds pairs of both scenes;
//   * outputs: 120 beams per half = four beams per lane;
//   * mask: lane hl < 21 of a half owns the forward action hl (register f) and the backward action 21 + hl (register b) -- the
//     two direction halves post_process filters separately (action_mask.py:186-196); a probe step visits one active coarse beam
//     of EACH scene, so the mask's dependent table probes are shared as well.
// LDS per wave 7.3 KB (two 32-obstacle tiles, two beam-minimum arrays, one queue of 16-bit entries); all synchronisation is
// LDS-only (ssync: no vmcnt(0) drain), so the hull-range / table-maximum loads requested early stay in flight across the phases.
#pragma once
#include "hope_step_kernel.h"

namespace hope {

constexpr int OP_HALF = 32;                       // lanes per scene
constexpr int OP_CAP = SMALL_TILE;                // obstacle slots per scene tile
static_assert(OP_CAP == OP_HALF, "one lane per obstacle slot in the near-obstacle scan");
#ifndef HOPE_PAIR_NARROW
#define HOPE_PAIR_NARROW 6
#endif
static_assert(HOPE_PAIR_NARROW * 64 <= 512, "narrow edges of one pass must fit the pair queue");
constexpr int OP_EAGER = 16;                     // obstacle-box slots requested before the obstacle count is known
constexpr int OP_LQ = 512;                        // (beam, edge) pair queue entries, shared by the two scenes
// per-half LDS block (doubles): tile[8 * OP_CAP] | best[128] (u64; later xs[121]) | klist[OP_CAP] i32 | cfl[OP_CAP] u8
constexpr int OP_TILE_W = 8 * OP_CAP, OP_BEST_W = 128, OP_KLIST_W = OP_CAP / 2, OP_CFL_W = OP_CAP / 8;
constexpr int OP_HALF_W = OP_TILE_W + OP_BEST_W + OP_KLIST_W + OP_CFL_W;
constexpr size_t OP_LDS_BYTES = (size_t)2 * OP_HALF_W * 8 + (size_t)OP_LQ * 2;
static_assert(NACT == 42 && NBEAM == 120 && NBEAM <= 4 * OP_HALF, "lane layout of the pair kernel");

template <typename OT>
__global__ __launch_bounds__(64) void k_obs_pair(StepParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, hw = lane >> 5, hl = lane & (OP_HALF - 1);
    const int n_pairs = (p.n_list + 1) >> 1;
    if ((int)blockIdx.x >= n_pairs) return;
    const int li = 2 * scene_of_block(blockIdx.x, n_pairs) + hw;
    bool live = li < p.n_list;
    const int scene = p.scene_list[live ? li : li - 1];
    if (p.active) live = live && p.active[scene] != 0;
    if (!__any(live)) return;

    double* hbase = lds + hw * OP_HALF_W;
    double* tile = hbase;
    unsigned long long* best = (unsigned long long*)(hbase + OP_TILE_W);
    int* klist = (int*)(hbase + OP_TILE_W + OP_BEST_W);
    uint8_t* cfl = (uint8_t*)(klist + OP_CAP);
    uint16_t* queue = (uint16_t*)(lds + 2 * OP_HALF_W);

    // ---- the scene's first loads, all requested at once (one round trip): obstacle count, pose, cos / sin, this lane's obstacle box + flags
    const int n_obst = live ? min(p.n_obst[scene], OP_CAP) : 0;
    const double* st = p.state + (size_t)scene * ST_WORDS;
    const double x = st[0], y = st[1];
    const double ct = p.cs[2 * (size_t)scene], sn = p.cs[2 * (size_t)scene + 1];          // hm_sincos(h) as the motion launch computed it
    // (obstacle boxes: the first OP_EAGER slots with the first loads, the others only for a scene that has them -- generated lots have
    // at most 13 obstacles: 256 instead of 512 bytes per scene and launch)
    const float4* obb_s = p.obb + (size_t)scene * p.max_obst;
    float4 bb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (hl < OP_EAGER) bb = obb_s[hl];
    const uint8_t gfl = (p.eflag + (size_t)scene * eflag_stride(p.max_obst))[hl];
    const double2* src = (const double2*)(p.verts + (size_t)scene * p.max_obst * 8);

    // ---- obstacles whose box comes within lidar_range of the sensor (a superset of the rings :69 keeps), compacted: tile slot k = the
    // k-th such obstacle of the scene (stage_near of the one-scene kernel, one lane per obstacle slot)
    if (hl >= OP_EAGER && hl < n_obst) bb = obb_s[hl];            // (behind every other first load: this one waits for the count)
    const double lr = LIDAR_RANGE + 1e-6;
    const bool near = hl < n_obst && !((double)bb.x > x + lr || (double)bb.y < x - lr || (double)bb.z > y + lr || (double)bb.w < y - lr);
    int n_l;
    {
        const unsigned long long m = __ballot(near);
        const unsigned hm = hw ? (unsigned)(m >> 32) : (unsigned)m;
        n_l = __popc(hm);
        if (near) {
            const int pos = __popc(hm & ((1u << hl) - 1));
            klist[pos] = hl;
            cfl[pos] = gfl;
        }
    }
    ssync();
    for (int i = hl; i < 4 * n_l; i += OP_HALF) ((double2*)tile)[i] = src[4 * klist[i >> 2] + (i & 3)];
    // requested now, used after the drain: this lane's four beams' hull ranges and table maxima
    double base[4], pm[4];                                        // (pm: the table's maximum at the beam, or with HOPE_MASK_LUT its bins per metre)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bi = hl + OP_HALF * r;
        base[r] = bi < NBEAM ? p.hull_base[bi] : 0.0;
        pm[r] = bi < NBEAM ? (HOPE_MASK_LUT ? p.mask_bsc[bi] : p.pmax[UPS * bi]) : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) best[hl + OP_HALF * r] = 0x7ff0000000000000ull;              // +inf
    ssync();
    // ---- world -> ego in place: affine [a, b, -b, a, x_off, y_off] (lidar_simulator.py:58-64)
    {
        const double a = ct, b = sn;
        const double x_off = -x * a - y * b;
        const double y_off = x * b - y * a;
        for (int i = hl; i < 4 * n_l; i += OP_HALF) {
            const double px = tile[2 * i], py = tile[2 * i + 1];
            tile[2 * i] = a * px + b * py + x_off;
            tile[2 * i + 1] = (-b) * px + a * py + y_off;
        }
    }
    ssync();
    // ---- ring kept iff distance(ring, origin) < lidar_range (:69); 4 consecutive lanes = one ring.  Decided from squared quantities
    // with 1e-9 of margin, GEOS's point-to-segment arithmetic only inside the margin (hope_step_kernel.h, the same expressions)
    const int nl_max = max(__builtin_amdgcn_readlane(n_l, 0), __builtin_amdgcn_readlane(n_l, OP_HALF));
    int n_k = 0;
    for (int base_i = 0; base_i < 4 * nl_max; base_i += OP_HALF) {
        const int i = base_i + hl;
        const bool in = i < 4 * n_l;
        const int o = in ? i >> 2 : 0;
        double dd = INFINITY;
        bool amb;
        {
            bool keep_c = false, drop_c = false;
            if (in) {
                const int e = 4 * o + (i & 3), e2 = 4 * o + ((i + 1) & 3);
                const double ax = tile[2 * e], ay = tile[2 * e + 1], bx = tile[2 * e2], by = tile[2 * e2 + 1];
                const double ddx = bx - ax, ddy = by - ay;
                const double len2 = ddx * ddx + ddy * ddy;
                const double dot = (0.0 - ax) * ddx + (0.0 - ay) * ddy;
                const double qa = ax * ax + ay * ay, qb = bx * bx + by * by;
                const double num = ay * ddx - ax * ddy;
                const double R2 = LIDAR_RANGE * LIDAR_RANGE, ETA = 1e-9;
                const double lo = R2 * (1.0 - ETA), hi = R2 * (1.0 + ETA);
                const double n2 = num * num;
                const bool line_far = n2 > hi * len2, line_near = n2 < lo * len2;
                const bool foot_in = dot > ETA * len2 && dot < (1.0 - ETA) * len2;
                const bool foot_a = dot < -ETA * len2, foot_b = dot > (1.0 + ETA) * len2;
                keep_c = qa < lo || qb < lo || (foot_in && line_near);
                drop_c = !keep_c && (len2 > 0.0 ? (line_far || (foot_a && qa > hi) || (foot_b && qb > hi)) : qa > hi);
            }
            const unsigned long long kb = __ballot(keep_c), db = __ballot(drop_c);
            const int sh = lane & ~3;
            const bool ring_keep = ((kb >> sh) & 0xF) != 0, ring_drop = ((db >> sh) & 0xF) == 0xF;
            if (ring_keep) dd = 0.0;
            amb = in && !ring_keep && !ring_drop;
        }
        if (__any(amb)) {
            if (amb) {
                const int e = 4 * o + (i & 3), e2 = 4 * o + ((i + 1) & 3);
                dd = origin_seg_dist(tile[2 * e], tile[2 * e + 1], tile[2 * e2], tile[2 * e2 + 1]);
            }
            dd = fmin(dd, dpp_d<0xB1>(dd));                  // quad_perm [1,0,3,2]
            dd = fmin(dd, dpp_d<0x4E>(dd));                  // quad_perm [2,3,0,1]
        }
        const bool kq = in && (i & 3) == 0 && dd < LIDAR_RANGE;
        const unsigned long long km = __ballot(kq);
        const unsigned hkm = hw ? (unsigned)(km >> 32) : (unsigned)km;
        if (kq) klist[n_k + __popc(hkm & ((1u << hl) - 1))] = o;          // (slot o >= its rank: in-place compaction of the identity list)
        n_k += __popc(hkm);
    }
    ssync();
    const int n_kslots = 4 * n_k;
    const int nk_max = max(__builtin_amdgcn_readlane(n_kslots, 0), __builtin_amdgcn_readlane(n_kslots, OP_HALF));

    // ---- beams.  Pass 1 (one lane per kept edge): the beams inside the angle the edge subtends (float32, 2e-3 rad of margin: a
    // superset) are appended as (half, edge, beam) entries to the shared queue; pass 2 (drain): 64 pairs at a time through the
    // reference arithmetic (lidar_simulator.py:98-133) and an LDS atomic-min per (scene, beam) over the squared ranges.
    int qn = 0;
    auto drain = [&]() {
        ssync();
        for (int q0 = 0; q0 < qn; q0 += WAVE) {
            const int q = q0 + lane;
            if (q < qn) {
                const int pr = queue[q];
                const int bi = pr & 127, e = (pr >> 7) & 127, hh = pr >> 14;
                const int e2 = (e & ~3) | ((e + 1) & 3);
                const double* tl = lds + hh * OP_HALF_W;
                const double x1 = tl[2 * e], y1 = tl[2 * e + 1], x2 = tl[2 * e2], y2 = tl[2 * e2 + 1];
                const double d = y2 - y1, ee = x1 - x2, f = y1 * x2 - x1 * y2;
                const double ba = p.beam_ab[2 * bi], bb_ = p.beam_ab[2 * bi + 1];
                double r = INFINITY;
                if (beam_may_hit(ba, bb_, x1, y1, x2, y2)) r = beam_edge(bi, ba, bb_, x1, y1, x2, y2, d, ee, f);
                if (r < INFINITY) atomicMin((unsigned long long*)(tl + OP_TILE_W) + bi, (unsigned long long)__double_as_longlong(r));
            }
        }
        ssync();
        qn = 0;
    };
    {
        const float PITCH = 6.283185307179586f / NBEAM, MARGIN = 2e-3f;
        for (int base_i = 0; base_i < nk_max && !(p.stages & 0x1000); base_i += OP_HALF) {   // 0x1000: profiling switch
            const int i = base_i + hl;
            const bool in = i < n_kslots;
            const int e = in ? 4 * klist[i >> 2] + (i & 3) : 0;
            int lo = 0, cnt = 0;
            bool front = false, back = false, risky = true;
            if (in) {
                const int e2 = (e & ~3) | ((e + 1) & 3);
                const double dx1 = tile[2 * e], dy1 = tile[2 * e + 1], dx2 = tile[2 * e2], dy2 = tile[2 * e2 + 1];
                const float x1 = (float)dx1, y1 = (float)dy1;
                const float x2 = (float)dx2, y2 = (float)dy2;
                const float t1 = span_angle(y1, x1);
                const float t2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t1), 0x39, 0xf, 0xf, true));   // quad_perm [1,2,3,0]: the next vertex's
                float dth = t2 - t1;
                if (dth > 3.14159265f) dth -= 6.28318531f;
                if (dth <= -3.14159265f) dth += 6.28318531f;
                const float span = fabsf(dth);
                const bool wild = span > 3.13f || !(span == span) || fminf(x1 * x1 + y1 * y1, x2 * x2 + y2 * y2) < 0.01f;
                if (wild) { lo = 0; cnt = NBEAM; }
                else {
                    float ts = dth >= 0 ? t1 : t2;
                    if (ts < 0) ts += 6.28318531f;
                    const int ilo = (int)ceilf((ts - MARGIN) * (1.0f / PITCH));
                    const int ihi = (int)floorf((ts + span + MARGIN) * (1.0f / PITCH));
                    cnt = ihi - ilo + 1;
                    if (cnt < 0) cnt = 0;
                    if (cnt > NBEAM) cnt = NBEAM;
                    lo = ilo < 0 ? ilo + NBEAM : ilo >= NBEAM ? ilo - NBEAM : ilo;
                    if ((unsigned)lo >= (unsigned)NBEAM) { lo = 0; cnt = NBEAM; }
                }
                // back-face cull (hope_step_kernel.h: the conditions under which a convex ring's back edges can never be a beam's minimum)
                const int fl = (int)cfl[e >> 2];
                const double cr = dx1 * dy2 - dx2 * dy1;
                const double sc2 = (dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2);
                const bool decisive = cr * cr > 1e-12 * sc2;
                const bool left = cr > 0;
                front = decisive && (left != ((fl & OBST_F_CCW) != 0));
                back = decisive && !front;
                const float q1 = t1 * (1.0f / PITCH);
                const bool near_beam = fabsf(q1 - rintf(q1)) * PITCH <= 2.0e-4f + 1e-5f;
                const bool thin = front && (fabs(dx2 - dx1) < 1e-4 || fabs(dy2 - dy1) < 1e-4);
                risky = wild || !decisive || near_beam || thin || !(fl & OBST_F_CONVEX);
            }
            {
                const unsigned long long rb = __ballot(risky), fb = __ballot(front), bb2 = __ballot(back);
                const int sh = lane & ~3;
                const bool ring_ok = ((rb >> sh) & 0xF) == 0 && ((fb >> sh) & 0xF) != 0 && ((bb2 >> sh) & 0xF) != 0;
                if (ring_ok && back && !(p.stages & 0x4000)) cnt = 0;          // (0x4000: A/B switch, no cull)
            }
            constexpr int NARROW = HOPE_PAIR_NARROW;          // edges of at most this many beams are appended lane-parallel (A/B: 4 / 8 not faster)
            const int tag = (hw << 14) | (e << 7);
            const int cs = (cnt > 0 && cnt <= NARROW) ? cnt : 0;
            const int incl = wave_incl_scan_i(cs, lane);
            const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
            if (total > 0) {
                if (qn + total > OP_LQ) drain();
                const int off = qn + incl - cs;
#pragma unroll
                for (int k = 0; k < NARROW; k++) {
                    if (k < cs) {
                        int bi = lo + k;
                        if (bi >= NBEAM) bi -= NBEAM;
                        queue[off + k] = (uint16_t)(tag | bi);
                    }
                }
                qn += total;
            }
            unsigned long long todo = __ballot(cnt > NARROW);
            while (todo) {
                const int el = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int lo_e = __builtin_amdgcn_readlane(lo, el), cnt_e = __builtin_amdgcn_readlane(cnt, el);
                if (qn + cnt_e > OP_LQ) drain();
                const int tag_e = __builtin_amdgcn_readlane(tag, el);
                if (lane < cnt_e) {
                    int bi = lo_e + lane;
                    if (bi >= NBEAM) bi -= NBEAM;
                    queue[qn + lane] = (uint16_t)(tag_e | bi);
                }
                if (cnt_e > WAVE && lane + WAVE < cnt_e) {        // an edge that spans more than 64 beams
                    int bi = lo_e + lane + WAVE;
                    if (bi >= NBEAM) bi -= NBEAM;
                    queue[qn + lane + WAVE] = (uint16_t)(tag_e | bi);
                }
                qn += cnt_e;
            }
        }
    }
    if (qn > 0) drain();
    else ssync();

    // ---- outputs: beams hl + 32 r of this half's scene (get_observation :46)
    double lid[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bi = hl + OP_HALF * r;
        const double bq = sqrt(__longlong_as_double((long long)best[bi < NBEAM ? bi : 0]));     // min of the roots = root of the min
        lid[r] = clipd(bq, 0, LIDAR_RANGE) - base[r];
    }
    if (p.lidar && live) {
        OT* lo_ = (OT*)p.lidar + (size_t)NBEAM * scene;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int bi = hl + OP_HALF * r;
            if (bi < NBEAM) lo_[bi] = (OT)lid[r];
        }
    }
    if (!p.action_mask || (p.stages & 0x2000)) return;            // 0x2000: internal profiling switch

    // ---- action mask (action_mask.py:166-196; the coarse-beam decision of hope_step_kernel.h) ---------------------------------
    ssync();                                                      // every lane has read its best[] words: xs[] takes their place
    double* xs = (double*)best;                                   // lidar_obs = clip(raw,0,10) + base (:170); [NBEAM + 1]
    double xv_own[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bi = hl + OP_HALF * r;
        xv_own[r] = clipd(lid[r], 0, 10) + base[r];
        if (bi < NBEAM) xs[bi] = xv_own[r];
    }
    if (hl == 0) xs[NBEAM] = xv_own[0];                           // circular (:158)
    constexpr int HALF_ACT = NACT / 2;                            // 21 actions per direction
    const bool alane = hl < HALF_ACT;
    int ms[2] = {NITER, NITER};                                   // step counts of action hl (forward) and HALF_ACT + hl (backward)
    bool tie = false;
#if HOPE_MASK_LUT
    // ---- round 6: the counts of a coarse beam come from the count-interval table (hope_env_upload_tables), one 2-byte load per (beam,
    // lane) for both directions and all rows k, every beam's load independent of every other's -- the row probes above were a chain of
    // dependent L2 round trips (probe at the current minimum, walk down, next group) and ~390 vector instructions per group of four
    // beams.  cnt_lo <= count <= cnt_hi per (beam, action); the minimum over the beams is known when min cnt_lo == min cnt_hi, else
    // the few (beam, action) pairs that can still lower it are decided on the float64 entries cnt_lo .. cnt_hi - 1 themselves, where
    // an entry within 1e-9 of the scan raises `tie` (-> the exact 1200-beam evaluation) exactly as the row probes did.
    uint32_t* alist = (uint32_t*)tile;                            // this half's active coarse beams: beam << 16 | table row (the tile is dead)
    int n_act = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bi = hl + OP_HALF * r;
        const double q = (xv_own[r] - (base[r] - 1e-6)) * pm[r];  // bin of the scan value; beyond the last bin: every entry <= x - 1e-9
        const bool c = live && bi < NBEAM && q < (double)MASK_LUT_NB;
        const unsigned long long m = __ballot(c);
        const unsigned hm = hw ? (unsigned)(m >> 32) : (unsigned)m;
        if (c) alist[n_act + __popc(hm & ((1u << hl) - 1))] = (uint32_t)(bi << 16) | (uint32_t)(bi * MASK_LUT_NB + (int)q);
        n_act += __popc(hm);
    }
    ssync();
    const int na_max = max(__builtin_amdgcn_readlane(n_act, 0), __builtin_amdgcn_readlane(n_act, OP_HALF));
    if (na_max > 0) {
        constexpr uint32_t NONE = (uint32_t)(NBEAM * MASK_LUT_NB);        // the table's last row: [10, 10] for every action
        const char* lutb = (const char*)p.mask_lut + 2 * hl;
        unsigned mlo[2] = {NITER, NITER}, mhi[2] = {NITER, NITER};
        constexpr int PG = 8;                                     // beams whose loads are in flight together
        // the first PG beams (all of them for nine waves out of ten): their table words stay in registers for the second visit --
        // one round trip for the first visit, none for the second visit's words
        unsigned pk[PG / 2];                                      // two 16-bit words per register
        {
            unsigned v0[PG];
#pragma unroll
            for (int g = 0; g < PG; g++) {
                const uint32_t e = g < n_act ? alist[g] & 0xFFFFu : NONE;
                v0[g] = *(const uint16_t*)(lutb + e * (MASK_LUT_ROW * 2u));
            }
#pragma unroll
            for (int g = 0; g < PG; g++) {
                mlo[0] = min(mlo[0], v0[g] & 15u); mhi[0] = min(mhi[0], (v0[g] >> 4) & 15u);
                mlo[1] = min(mlo[1], (v0[g] >> 8) & 15u); mhi[1] = min(mhi[1], v0[g] >> 12);
            }
#pragma unroll
            for (int g = 0; g < PG / 2; g++) pk[g] = v0[2 * g] | v0[2 * g + 1] << 16;
        }
        for (int k0 = PG; k0 < na_max; k0 += PG) {
            unsigned v[PG];
#pragma unroll
            for (int g = 0; g < PG; g++) {
                const uint32_t e = k0 + g < n_act ? alist[k0 + g] & 0xFFFFu : NONE;
                v[g] = *(const uint16_t*)(lutb + e * (MASK_LUT_ROW * 2u));
            }
#pragma unroll
            for (int g = 0; g < PG; g++) {
                mlo[0] = min(mlo[0], v[g] & 15u); mhi[0] = min(mhi[0], (v[g] >> 4) & 15u);
                mlo[1] = min(mlo[1], (v[g] >> 8) & 15u); mhi[1] = min(mhi[1], v[g] >> 12);
            }
        }
        if (__any(mlo[0] < mhi[0] || mlo[1] < mhi[1])) {
            // second visit: the (beam, action) pairs with cnt_lo < cnt_hi and cnt_lo below the minimum of the upper bounds
            const char* tabb = (const char*)p.tab;
            for (int k = 0; k < na_max; k++) {
                const bool has = k < n_act;
                const uint32_t ae = has ? alist[k] : NONE;
                unsigned v;
                switch (k >> 1) {                                 // (wave-uniform)
                    case 0: v = pk[0]; break; case 1: v = pk[1]; break; case 2: v = pk[2]; break; case 3: v = pk[3]; break;
                    default: v = (unsigned)*(const uint16_t*)(lutb + (ae & 0xFFFFu) * (MASK_LUT_ROW * 2u)) << ((k & 1) << 4); break;
                }
                v = (v >> ((k & 1) << 4)) & 0xFFFFu;
                const unsigned lo0 = v & 15u, hi0 = (v >> 4) & 15u, lo1 = (v >> 8) & 15u, hi1 = v >> 12;
                const bool n0 = lo0 < hi0 && lo0 < mhi[0], n1 = lo1 < hi1 && lo1 < mhi[1];
                if (!__any(n0 || n1)) continue;
                const int ib = (int)(ae >> 16);
                const double xv = xs[has ? ib : 0];
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    if (d == 0 ? n0 : n1) {
                        unsigned c = d == 0 ? lo0 : lo1;
                        const unsigned lim = min(d == 0 ? hi0 : hi1, mhi[d]);
                        const unsigned rowo = (unsigned)(UPS * ib) * (NITER * NACT * 8u) + (unsigned)(hl + d * HALF_ACT) * 8u;
                        while (c < lim) {                         // (one entry almost always: a bin holds one table value)
                            const double t = *(const double*)(tabb + (rowo + c * (NACT * 8u)));
                            if (t > xv) break;
                            if (t > xv - 1e-9) tie = true;
                            c++;
                        }
                        mhi[d] = min(mhi[d], c);
                    }
                }
            }
        }
        ms[0] = (int)mhi[0]; ms[1] = (int)mhi[1];
    }
#else
    uint8_t* alist = (uint8_t*)tile;                              // this half's active coarse beams (the tile is dead)
    int n_act = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int bi = hl + OP_HALF * r;
        const bool c = live && bi < NBEAM && xv_own[r] - 1e-9 < pm[r];
        const unsigned long long m = __ballot(c);
        const unsigned hm = hw ? (unsigned)(m >> 32) : (unsigned)m;
        if (c) alist[n_act + __popc(hm & ((1u << hl) - 1))] = (uint8_t)bi;
        n_act += __popc(hm);
    }
    ssync();
    const int na_max = max(__builtin_amdgcn_readlane(n_act, 0), __builtin_amdgcn_readlane(n_act, OP_HALF));
    constexpr int MG = HOPE_MASK_MG;
    for (int k0 = 0; k0 < na_max; k0 += MG) {
        int ib[MG];
        bool has[MG];
#pragma unroll
        for (int g = 0; g < MG; g++) {
            has[g] = k0 + g < n_act;
            ib[g] = has[g] ? (int)alist[k0 + g] : 0;
        }
        if (alane) {
            double v[MG][2];
            const int mp0 = ms[0], mp1 = ms[1];
            // (table offsets as 32-bit BYTE offsets off the scalar table pointer: the 4 MB table needs 22 bits, and a 64-bit multiply-add per
            // lane and load was a fifth of this stage's vector instructions)
            const char* tabb = (const char*)p.tab;
#pragma unroll
            for (int g = 0; g < MG; g++) {
                const unsigned rowo = (unsigned)(UPS * ib[g]) * (NITER * NACT * 8u) + (unsigned)hl * 8u;
                v[g][0] = (has[g] && mp0 > 0) ? *(const double*)(tabb + (rowo + (unsigned)(mp0 - 1) * (NACT * 8u))) : 0.0;
                v[g][1] = (has[g] && mp1 > 0) ? *(const double*)(tabb + (rowo + (unsigned)(mp1 - 1) * (NACT * 8u) + HALF_ACT * 8u)) : 0.0;
            }
#pragma unroll
            for (int g = 0; g < MG; g++) {
                if (!has[g]) continue;
                const double xv = xs[ib[g]];
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    if (ms[d] > 0) {
                        const int mprobe = d == 0 ? mp0 : mp1;
                        double bv = v[g][d];                      // boundary value: largest examined entry <= x
                        if (bv > xv) {
                            const unsigned rowo = (unsigned)(UPS * ib[g]) * (NITER * NACT * 8u) + (unsigned)(hl + d * HALF_ACT) * 8u;
                            // walk down to the first entry <= x, four rows per trip; row ms - 1 is already known to exceed when nothing
                            // lowered ms since the probe
                            int c = ms[d] == mprobe ? ms[d] - 1 : ms[d];
                            bv = -INFINITY;
                            // (round 6: no per-row compare-and-branch -- the table is monotone in k, so the rows that exceed x form a
                            // prefix of the four loaded ones and their NUMBER is the step down; the boundary value is the next row)
                            while (c > 0) {
                                const unsigned o0 = rowo + (unsigned)(c - 1) * (NACT * 8u);
                                const double b0 = *(const double*)(tabb + o0);
                                const double b1 = c > 1 ? *(const double*)(tabb + (o0 - NACT * 8u)) : -INFINITY;
                                const double b2 = c > 2 ? *(const double*)(tabb + (o0 - 2 * NACT * 8u)) : -INFINITY;
                                const double b3 = c > 3 ? *(const double*)(tabb + (o0 - 3 * NACT * 8u)) : -INFINITY;
                                const int nf = (b0 > xv ? 1 : 0) + (b1 > xv ? 1 : 0) + (b2 > xv ? 1 : 0) + (b3 > xv ? 1 : 0);
                                bv = nf == 0 ? b0 : (nf == 1 ? b1 : (nf == 2 ? b2 : (nf == 3 ? b3 : -INFINITY)));
                                c -= nf;
                                if (nf < 4) break;
                            }
                            if (c == 0) bv = -INFINITY;
                            ms[d] = c;
                        }
                        if (bv > xv - 1e-9) tie = true;
                    }
                }
            }
        }
    }
#endif
    {
        // a table entry within 1e-9 of the scan (6e-7 of the scene-steps): the exact 1200-beam evaluation for that scene, by the
        // whole wave in the one-scene kernel's layout (lane = action), from scratch -- the coarse beams are among the 1200
        const unsigned long long tm = __ballot(tie);
        if (tm) {
            int* mx = (int*)(tile + 64);                          // [NACT] step counts of the scene being re-evaluated (this half's tile words)
            ssync();
            for (int hsel = 0; hsel < 2; hsel++) {
                if (!((tm >> (OP_HALF * hsel)) & 0xFFFFFFFFull)) continue;
                const double* xh = lds + hsel * OP_HALF_W + OP_TILE_W;
                int mstep = NITER;
                for (int r = 0; r < (NL + WAVE - 1) / WAVE; r++) {
                    const int l = r * WAVE + lane;
                    double dl = 0;
                    bool act = false;
                    if (l < NL) {
                        const int i = l / UPS, j = l % UPS;
                        const double w2 = (double)j / UPS, w1 = 1 - w2;      // (j % 10) / 10 of _linear_interpolate
                        dl = xh[i] * w1 + xh[i + 1] * w2;                     // _linear_interpolate (:161-162)
                        act = dl < p.pmax[l];
                    }
                    unsigned long long m = __ballot(act);
                    while (m) {
                        const int bpos = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const double d_ll = __shfl(dl, bpos);
                        if (lane < NACT) {
                            const double* row = p.tab + (size_t)(r * WAVE + bpos) * NITER * NACT + lane;
                            int c = mstep;
                            while (c > 0 && row[(c - 1) * NACT] > d_ll) c--;
                            mstep = c;
                        }
                    }
                }
                int* mxh = (int*)(lds + hsel * OP_HALF_W + 64);
                if (lane < NACT) mxh[lane] = mstep;
            }
            ssync();
            if ((tm >> (OP_HALF * hw)) & 0xFFFFFFFFull) {
                if (alane) { ms[0] = mx[hl]; ms[1] = mx[HALF_ACT + hl]; }
            }
        }
    }
    // post_process (:186-196) per direction half: both ends -1, min filter (5, reflect), clip, /10
    double mo[2];
    bool nzl = false;
    if (HOPE_MASK_LUT && na_max == 0) {
        // no scene of the wave has an active beam (two thirds of the scenes): every count is NITER, the filter's result is known --
        // NITER - 1 within two actions of either end of a direction half (the ends' decrement, spread by the 5-wide minimum), NITER elsewhere
        const int mn = (hl <= 2 || hl >= HALF_ACT - 3) ? NITER - 1 : NITER;
        mo[0] = mo[1] = HOPE_MASK_FRAC_TABLE ? MASK_STEP_FRACTION[mn] : mask_fraction(mn);
        nzl = true;
    } else
#pragma unroll
    for (int d = 0; d < 2; d++) {
        int v = ms[d];
        if (hl == 0 || hl == HALF_ACT - 1) v -= 1;
        int mn = v;
#pragma unroll
        for (int off = -2; off <= 2; off++) {
            int j = hl + off;
            if (j < 0) j = -j - 1;
            if (j >= HALF_ACT) j = 2 * HALF_ACT - 1 - j;
            j = j < 0 ? 0 : (j >= HALF_ACT ? HALF_ACT - 1 : j);      // (lanes beyond the 21 actions: any lane of the half)
            const int o = __shfl(v, (hw << 5) + j);
            mn = min(mn, o);
        }
        mn = max(0, min(NITER, mn));
        mo[d] = HOPE_MASK_FRAC_TABLE ? MASK_STEP_FRACTION[mn] : mask_fraction(mn);
        nzl = nzl || mn > 0;
    }
    {
        const unsigned long long nz = __ballot(alane && nzl);
        const unsigned hnz = hw ? (unsigned)(nz >> 32) : (unsigned)nz;
        if (hnz == 0) { mo[0] = clipd(mo[0], 0.01, 1); mo[1] = clipd(mo[1], 0.01, 1); }   // all-zero -> 0.01 (:182-183)
    }
    if (alane && live) {
        OT* mk = (OT*)p.action_mask + (size_t)NACT * scene;
        mk[hl] = (OT)mo[0];
        mk[HALF_ACT + hl] = (OT)mo[1];
    }
}

}  // namespace hope
