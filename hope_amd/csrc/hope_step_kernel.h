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
#ifndef HOPE_PART0_OCC
#define HOPE_PART0_OCC 4     // waves per SIMD the one-launch form of the step kernel is compiled for.  3 = 137 VGPRs, no scratch, no VGPR spills -- and slower
                            // from 4 096 scenes on (16 384: 0.235 vs 0.222 ms, profiles/r05_ab_part0_occupancy.txt): 4 with its 4 spilled VGPRs (12 B) stays
#endif
#ifndef HOPE_KIN_PRIO
#define HOPE_KIN_PRIO 3       // wave priority (s_setprio) of k_kinematics: it heads the step's critical chain (kinematics -> motion -> observation of
                            // the larger class) and shares its SIMDs with the previous step's observation / validation waves: 0.4967 -> 0.4937 ms
                            // (profiles/r05_ab_wave_priority.txt; priorities for the motion / observation launches on top of it: nothing)
#endif
#ifndef HOPE_MASK_MG
#define HOPE_MASK_MG 4      // action-mask rows probed together (A/B builds: -DHOPE_MASK_MG=8 measured 1.3 % slower)
#endif
#include "hope_env.h"
#ifndef HOPE_STEP_SYNC_FULL
#define HOPE_STEP_SYNC_FULL 0   // 1: every LDS synchronisation point of the step kernel is a __syncthreads() (rounds 1-5)
#endif
#ifndef HOPE_MASK_LUT
#define HOPE_MASK_LUT 1         // 1: the mask stage of k_obs_pair reads the count-interval table (round 6); 0: the row probes of rounds 2-5
#endif
#ifndef HOPE_MASK_FRAC_TABLE
#define HOPE_MASK_FRAC_TABLE 0  // 1: the mask's k / n_iter from a table in constant memory (round 5: a dependent load at the kernel's very end)
#endif

namespace hope {

// LDS ordering inside the ONE wave of a step-kernel workgroup.  A __syncthreads() is a workgroup-scope fence: the compiler puts
// s_waitcnt vmcnt(0) in front of it, so EVERY synchronisation point also waited for the wave's outstanding global loads and stores --
// the table / constant loads requested early "to be in flight" during the LDS phases, and (gfx9 counts stores in vmcnt too) the
// acknowledgement of stores just issued, e.g. the lidar row in front of the mask stage: hidden memory round trips in a kernel whose
// waves spend most of their life waiting.  LDS instructions of one wave execute in order, so LDS write -> read across lanes only
// needs the compiler not to reorder them (and the returned data: lgkmcnt).  No lane of these kernels reads GLOBAL memory another lane
// of the same launch wrote.
__device__ __forceinline__ void ssync() {
    if (HOPE_STEP_SYNC_FULL) __syncthreads();
    else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
}

// k / 10 as IEEE division rounds it, without a table lookup or a division (the argument of div_by_20 below, checked for all eleven
// k in tests/test_math.py): q0 = RN(k R) with R = RN(0.1), the remainder k - 10 q0 exactly by fma, the correction by a second fma
__device__ __forceinline__ double mask_fraction(int k) {
    const double x = (double)k, R = 0.1;
    const double q0 = x * R;
    const double r = fma(-q0, 10.0, x);
    return fma(r, R, q0);
}

// Dragon-Lake-Parking cases resident on the device (hope_env_set_dlp_cases): what ParkingMapDLP.reset draws from
// (src/env/parking_map_dlp.py:38-86).  A pool list entry j <= -2 names case -2 - j.
struct DlpCases {
    int n_cases;
    const double* dest;       // [n_cases][3]
    const int32_t* cand_off;  // [n_cases + 1] start candidates of case c: cand[cand_off[c] .. cand_off[c + 1])
    const double* cand;       // [][3]
    const int32_t* case_set;  // [n_cases] obstacle set of the case
    const int32_t* set_off;   // [n_sets + 1]
    const double* set_verts;  // [][4][2] obstacle rings of all sets (a triangle repeats its last vertex)
};

// the per-scene constant record from (start, dest, map box): what hope_env_set_scenes computes per scene
__device__ __forceinline__ void fill_scene_consts(double* c, const double* start, const double* dest, const double* bbox) {
    for (int i = 0; i < 3; i++) { c[SC_START + i] = start[i]; c[SC_DEST + i] = dest[i]; }
    for (int i = 0; i < 4; i++) c[SC_BBOX + i] = bbox[i];
    double sn, ct;
    hm_sincos(dest[2], &sn, &ct);
    Box b = make_box(dest[0], dest[1], ct, sn);                   // dest.create_box()
    for (int v = 0; v < 4; v++) { c[SC_DBOX + 2 * v] = b.x[v]; c[SC_DBOX + 2 * v + 1] = b.y[v]; }
    double sum = 0.0, x0 = b.x[0];                                // Polygon(dest_box).area: GEOS Area::ofRingSigned
    for (int i = 1; i < 4; i++) sum += (b.x[i] - x0) * (b.y[i - 1] - b.y[(i + 1) & 3]);
    c[SC_DAREA] = fabs(sum / 2.0);
    double dx = dest[0] - start[0], dy = dest[1] - start[1];
    c[SC_DNORM] = fmax(sqrt(dx * dx + dy * dy), 10.0);            // car_parking_base.py:211
    c[SC_DCEN] = 0.5 * (b.x[0] + b.x[2]);
    c[SC_DCEN + 1] = 0.5 * (b.y[0] + b.y[2]);
    c[SC_DCEN + 2] = ct;
    c[SC_DCEN + 3] = sn;
}

// uniform in [0, 1) number i of the counter-based stream `key`
__device__ __forceinline__ double draw_uniform(uint64_t key, int i) {
    return (double)(mix64(key + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
}

// ParkingMapDLP.reset (parking_map_dlp.py:38-86) for case `cs`, by one wave: a start candidate drawn uniformly + N(0, 0.05^2) m /
// N(0, 0.02^2) rad jitter (:60-65), the map box floor / ceil(min / max(start, dest) -/+ 20) (:70-73), the case's obstacles
// culled by that box in their order (filter_obstacles :88-101), then independent 50 % flips of dest and start about their box
// centres (:80-83, _flip_box_orientation :117-123).  Writes the scene's tile (and its LDS copy when ltile != null), obstacle
// boxes, the float32 view of the obstacles (obstacle_f32) and the constant record `c24` (24 doubles, LDS or global, written by
// lane 0); returns the obstacle count.
__device__ __forceinline__ int draw_dlp_case(const DlpCases& D, int cs, uint64_t key, int max_obst, double* gverts, float4* gobb,
                                             float4* gfv, float4* gfbox, uint8_t* geflag,
                                             double* c24, double* ltile, int32_t* overflow, int lane) {
    const int c0 = D.cand_off[cs], n_c = D.cand_off[cs + 1] - c0;
    int ci = (int)(draw_uniform(key, 0) * n_c);
    ci = ci < n_c ? ci : n_c - 1;
    double st[3], de[3];
    for (int i = 0; i < 3; i++) { st[i] = D.cand[3 * (size_t)(c0 + ci) + i]; de[i] = D.dest[3 * (size_t)cs + i]; }
    {   // three standard normals (Box-Muller on two pairs of uniforms)
        const double u1 = fmax(draw_uniform(key, 1), 1e-300), u2 = draw_uniform(key, 2);
        const double u3 = fmax(draw_uniform(key, 3), 1e-300), u4 = draw_uniform(key, 4);
        double s2, c2, s4, c4;
        hm_sincos(2.0 * PI * u2, &s2, &c2);
        hm_sincos(2.0 * PI * u4, &s4, &c4);
        const double r1 = sqrt(-2.0 * log(u1)), r3 = sqrt(-2.0 * log(u3));
        st[0] += 0.05 * (r1 * c2); st[1] += 0.05 * (r1 * s2); st[2] += 0.02 * (r3 * c4);
        (void)s4;
    }
    double bb[4] = {floor(fmin(st[0], de[0]) - 20.0), ceil(fmax(st[0], de[0]) + 20.0),
                    floor(fmin(st[1], de[1]) - 20.0), ceil(fmax(st[1], de[1]) + 20.0)};
    // cull, keeping the order of the set
    const int so = D.case_set[cs], a = D.set_off[so], n_set = D.set_off[so + 1] - a;
    int cnt = 0;
    for (int base = 0; base < n_set; base += WAVE) {
        const int o = base + lane;
        bool keep = false;
        double v[8];
        if (o < n_set) {
            const double2* src = (const double2*)(D.set_verts + 8 * (size_t)(a + o));
#pragma unroll
            for (int k = 0; k < 4; k++) { const double2 q = src[k]; v[2 * k] = q.x; v[2 * k + 1] = q.y; }
            const double mnx = fmin(fmin(v[0], v[2]), fmin(v[4], v[6])), mxx = fmax(fmax(v[0], v[2]), fmax(v[4], v[6]));
            const double mny = fmin(fmin(v[1], v[3]), fmin(v[5], v[7])), mxy = fmax(fmax(v[1], v[3]), fmax(v[5], v[7]));
            keep = !(mxx <= bb[0] || mnx >= bb[1] || mxy <= bb[2] || mny >= bb[3]);
        }
        const unsigned long long m = __ballot(keep);
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1));
        if (keep && pos < max_obst) {
            double2* dst = (double2*)(gverts + 8 * (size_t)pos);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = make_double2(v[2 * k], v[2 * k + 1]);
            if (ltile) {
#pragma unroll
                for (int k = 0; k < 8; k++) ltile[8 * pos + k] = v[k];
            }
            gobb[pos] = obstacle_box(v);
            obstacle_f32(v, bb[0], bb[2], gfv + 2 * pos, gfbox + pos, geflag + pos);     // frame origin = (map box xmin, ymin)
        }
        cnt += __popcll(m);
    }
    if (cnt > max_obst) { if (lane == 0 && overflow) atomicAdd(overflow, 1); cnt = max_obst; }
    if (lane == 0) {
        // flips about the box centre: the centre is pose + R(yaw) (mid, 0), the flipped pose its mirror image, heading + pi
        const double mid = 0.5 * (CAR_XF + CAR_XR);
        if (draw_uniform(key, 5) > 0.5) { double s_, c_; hm_sincos(de[2], &s_, &c_); de[0] += 2.0 * mid * c_; de[1] += 2.0 * mid * s_; de[2] += PI; }
        if (draw_uniform(key, 6) > 0.5) { double s_, c_; hm_sincos(st[2], &s_, &c_); st[0] += 2.0 * mid * c_; st[1] += 2.0 * mid * s_; st[2] += PI; }
        fill_scene_consts(c24, st, de, bb);
    }
    return cnt;
}

// What the step kernel touches rarely (episode turnover, image bookkeeping): read through ONE pointer at the point of use.
// As by-value kernel arguments these ~25 pointers were all loaded at kernel entry and stayed live in SGPRs to their (rare) use:
// 106 SGPRs, 68 of them spilled to VGPR lanes, in a kernel that sits on the 128-VGPR occupancy step.  The record lives in
// device memory (a small ring in the handle, re-uploaded stream-ordered when its content changes: a pool commit, a new seed).
struct StepCold {
    double* traj;             // HOPE_F_IMAGE: [n][20][3] ring of vehicle.trajectory (entry e in slot e % 20), else null
    int32_t* traj_len;        // HOPE_F_IMAGE: [n] len(vehicle.trajectory)
    int32_t* traj_valid;      // HOPE_F_IMAGE: [n] span-table watermark of the image kernels (0 after a reset)
    int32_t* layer_valid;     // HOPE_F_IMAGE: [n] the static image layer matches the scene's map (0 after a new map)
    // HOPE_AUTO_REDRAW: the device-resident scene pool (hope_env_set_pool), or null pointers
    const double* pool_verts; // [pool_n][max_obst][8]
    const double* pool_c;     // [pool_n][SC_WORDS]
    const int32_t* pool_nobst;
    const int32_t* pool_cls[2];
    int pool_cls_n[2];
    int32_t* cur_pool;        // [n]
    uint32_t* episode;        // [n]
    unsigned long long redraw_seed;
    DlpCases dlp;             // HOPE_AUTO_REDRAW: Dragon-Lake-Parking cases drawn on the device (hope_env_set_dlp_cases), or n_cases = 0
    int32_t* pool_overflow;   // [1] draws whose culled obstacle set exceeded max_obst (truncated): must stay 0
    const uint8_t* slot_cls;  // [n] draw class of every scene slot (0: lots of <= 32 obstacles, 1: larger)
    float4* fverts;           // [n][max_obst][2] float32 view of the obstacles (obstacle_f32), rewritten with every new map
    float4* fbox;             // [n][max_obst]
    uint8_t* eflag;           // [n][eflag_stride(max_obst)]
};

constexpr uint32_t STEP_HF_TRAJ = 1;   // StepParams::hflags: the handle keeps vehicle.trajectory (HOPE_F_IMAGE)

struct StepParams {
    int n, max_obst;          // max_obst = HBM tile stride (obstacle slots per scene)
    int tile_cap;             // LDS tile capacity of THIS launch (obstacles)
    const int32_t* scene_list; // scenes of this launch's tile class (dense launch: grid = n_list)
    int n_list;
    uint32_t stages;
    int has_action;
    uint32_t hflags;          // STEP_HF_*
    const double* verts;      // [n][max_obst][4][2]
    const float4* obb;        // [n][max_obst] obstacle boxes (xmin, xmax, ymin, ymax), float32 rounded outwards
    const uint8_t* eflag;     // [n][eflag_stride(max_obst)] per-obstacle flags (obstacle_f32): bits 4, 5 = shape flags of the lidar's back-face cull
    const int32_t* n_obst;    // [n]
    const double* scene_c;    // [n][SC_WORDS]
    double* state;            // [n][ST_WORDS]
    double* cs;               // [n][2] cos / sin of the heading the motion launch left in state (read by the observation launch)
    int32_t* tstep;           // [n]
    const void* actions;      // [n][2] (AT) this step's actions: read by the wave's own kinematics (FKIN), else by k_kinematics
    double* kin;              // [n][50] sub-step poses from k_kinematics: h[10] cos[10] sin[10] x[10] y[10]
    const uint8_t* active;    // [n] or null (the caller's mask: only read by launches the caller's stream is ordered after)
    uint8_t* active_out;      // [n] or null: the motion launch snapshots the mask here for the Reeds-Shepp chain (k_rs_compact), which
                              //   with HOPE_DEFER_RS runs after the caller's stream has been released
    const double* tab;        // prefix-max mask table [NL][NITER][NACT]
    const double* pmax;       // [NL] max over (a,k) of tab
    const uint16_t* mask_lut; // [MASK_LUT_ROWS][MASK_LUT_ROW] count intervals of the coarse beams by scan-value bin (hope_env_upload_tables)
    const double* mask_bsc;   // [NBEAM] bins per metre
    const double* hull_base;  // [NBEAM]
    const double* beam_ab;    // [NBEAM][2]
    void* lidar;              // hope_step_out.lidar / .action_mask (the other outputs are written by k_post / the Reeds-Shepp kernels)
    void* action_mask;
    const StepCold* cold;     // device memory
    double* post;             // [n][POST_WORDS] per-scene hand-over to k_post (reward / target arithmetic, lane = scene)
    int32_t* rs_count_zero;   // this tile class's RS queue counter, cleared here for the k_rs_compact that follows; or null
};

// LDS per wave (doubles): tile 8*tile_cap | region A [320] | hb[10] cb[10] sb[10] px[10] py[10] | dest box[8] |
//   w2[10] | ints: near / keep list.  Region A is reused by the phases in turn:
//     arrival   : sh[64]  Sutherland-Hodgman scratch
//     lidar     : best[128] u64 + queue[384] i32
//     mask      : x[121]
constexpr int LDS_TX = 0, LDS_SH = 0, LDS_X = 0, LDS_HB = 320, LDS_CB = 330, LDS_SB = 340, LDS_PX = 350,
              LDS_PY = 360, LDS_DBOX = 370, LDS_W2 = 378, LDS_KEEP = 388, LDS_SCRATCH_WORDS = 388,
              LDS_ROBUST = 96;   // [ROBUST_LDS_WORDS] in region A, behind sh[64] and the turnover's constant record [64..88)
constexpr int KIN_WORDS = 56;   // h[10] cos[10] sin[10] x[10] y[10], [50] = int2(bits 0..9 arrival-possible of the ten poses, bit 10 of the start pose, bits 16..26 inside-the-map-box of the eleven; 0), [51] pad,
                                // [52..55] = box (xmin, xmax, ymin, ymax) around the hulls of the start pose and the ten poses
// k_env_step -> k_post record: previous pose, final pose of the finished step, overlap area, (status | t << 8 | flags << 24)
constexpr int POST_WORDS = 8;
constexpr int POST_F_REWARD = 1, POST_F_TURNOVER = 2, POST_F_NEED_UA = 4;   // NEED_UA: k_post computes the overlap area of the final pose
__host__ __device__ inline size_t step_lds_bytes(int tile_cap) {
    return (size_t)(8 * tile_cap + LDS_SCRATCH_WORDS) * 8 + (size_t)((tile_cap + 3) & ~3) * 4 + (size_t)((tile_cap + 3) & ~3);   // + near / keep list + shape flags
}
// Count-interval table of the action mask (round 6): per coarse beam MASK_LUT_NB scan-value bins, per bin one row of 32 uint16 --
// word hl < 21 = (cnt_lo, cnt_hi) of the forward action hl in bits 0-3 / 4-7 and of the backward action 21 + hl in bits 8-11 / 12-15;
// the other words and the last row (the "no beam" entry of a probe group) read [10, 10].
constexpr int MASK_LUT_NB = 128, MASK_LUT_ROW = 32, MASK_LUT_ROWS = NBEAM * MASK_LUT_NB + 1;
constexpr size_t MASK_LUT_BYTES = (size_t)MASK_LUT_ROWS * MASK_LUT_ROW * sizeof(uint16_t);
constexpr int SMALL_TILE = 32;   // scenes with <= 32 obstacles run in a launch with a 2 KB tile (higher occupancy)

// GEOS Area::ofRingSigned over an open vertex list (ring closed implicitly); lane-0 code, LDS arrays
__device__ __forceinline__ double ring_area_signed_lds(const double* px, const double* py, int n) {
    if (n < 3) return 0.0;
    double sum = 0.0, x0 = px[0];
#pragma unroll 1
    for (int i = 1; i < n; i++) {
        double x = px[i] - x0;
        int ip = (i + 1 == n) ? 0 : i + 1;
        sum += x * (py[i - 1] - py[ip]);
    }
    return sum / 2.0;
}

// Polygon(A).intersection(Polygon(B)).area for convex CCW quads: Sutherland-Hodgman + shoelace.
// Runs on ONE lane with LDS scratch sh[64] (two 8-vertex ping-pong buffers of x and y); quad A is already in
// sh[0..3] (x) and sh[16..19] (y).  Every argument is an LDS pointer (a Box passed by reference to an out-of-line function had
// to live in scratch memory: those stores were most of the round-1 kernel's HBM writes).  Inlined with ROLLED loops since
// round 4: a call in the middle of the step kernel made the register allocator spill what was live across it, and the kernel
// is to run without any scratch memory.
__device__ __forceinline__ double quad_intersection_area_lane0(const double* B /*8 words x,y*/, double* sh) {
    double* ax = sh;      double* ay = sh + 16;
    double* bx = sh + 32; double* by = sh + 48;
    int n = 4;
#pragma unroll 1
    for (int e = 0; e < 4 && n > 0; e++) {
        double c1x = B[2 * e], c1y = B[2 * e + 1];
        double c2x = B[2 * ((e + 1) & 3)], c2y = B[2 * ((e + 1) & 3) + 1];
        double ex = c2x - c1x, ey = c2y - c1y;
        int m = 0;
#pragma unroll 1
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

// The same clip with one LANE per polygon pair (k_post): the ping-pong buffers are lane-private LDS columns, word i of
// buffer b of column l at sh[(8 b + i) * CLIP_COLS + l] (b = 0..3: ax, ay, bx, by; a convex quad clipped by a convex quad has
// at most 8 vertices).  Same expressions in the same order as quad_intersection_area_lane0.  CLIP_COLS lanes of a wave clip at a
// time (k_post walks the lanes that need it in groups of 16 -- in steady state a third of the scenes are within reach of their slot): 4 KB of LDS per block -- with a column per LANE (16 KB per 64-thread
// block, rounds 2-3) the blocks of this tiny kernel waited for LDS behind the observation launches (5 KB per wave, 30 per CU)
// and took 130 us.
constexpr int CLIP_COLS = 16;
#define WAVE_CLIP CLIP_COLS
__device__ __noinline__ double quad_intersection_area_private(const double* B /*8 words x,y, global*/, double* sh /* + column */) {
    double* ax = sh;                    double* ay = sh + 8 * WAVE_CLIP;
    double* bx = sh + 16 * WAVE_CLIP;   double* by = sh + 24 * WAVE_CLIP;
    int n = 4;
    for (int e = 0; e < 4 && n > 0; e++) {
        const double c1x = B[2 * e], c1y = B[2 * e + 1];
        const double c2x = B[2 * ((e + 1) & 3)], c2y = B[2 * ((e + 1) & 3) + 1];
        const double ex = c2x - c1x, ey = c2y - c1y;
        int m = 0;
        for (int i = 0; i < n; i++) {
            const int i2 = (i + 1 == n) ? 0 : i + 1;
            const double sx = ax[i * WAVE_CLIP], sy = ay[i * WAVE_CLIP], tx = ax[i2 * WAVE_CLIP], ty = ay[i2 * WAVE_CLIP];
            const double ds = ex * (sy - c1y) - ey * (sx - c1x);
            const double dt = ex * (ty - c1y) - ey * (tx - c1x);
            const bool sin_ = ds >= 0, tin = dt >= 0;
            if (sin_) { bx[m * WAVE_CLIP] = sx; by[m * WAVE_CLIP] = sy; m++; }
            if (sin_ != tin) {
                const double r = ds / (ds - dt);
                bx[m * WAVE_CLIP] = sx + r * (tx - sx);
                by[m * WAVE_CLIP] = sy + r * (ty - sy);
                m++;
            }
        }
        n = m;
        double* t;
        t = ax; ax = bx; bx = t;
        t = ay; ay = by; by = t;
    }
    if (n < 3) return 0.0;
    double sum = 0.0;                                  // GEOS Area::ofRingSigned (ring_area_signed_lds)
    const double x0 = ax[0];
    for (int i = 1; i < n; i++) {
        const double x = ax[i * WAVE_CLIP] - x0;
        const int ip = (i + 1 == n) ? 0 : i + 1;
        sum += x * (ay[(i - 1) * WAVE_CLIP] - ay[ip * WAVE_CLIP]);
    }
    return fabs(sum / 2.0);
}

// |hull ∩ dest| with an exact quick reject: both boxes lie inside discs of radius rho about their
// centres; disjoint discs -> GEOS returns an empty intersection, area 0.0.
__device__ __forceinline__ double overlap_area(double px, double py, double ct, double st, const double* dbox_lds, double* sh, int lane) {
    const Box box = make_box(px, py, ct, st);
    double cx = 0.5 * (box.x[0] + box.x[2]), cy = 0.5 * (box.y[0] + box.y[2]);
    double dx = 0.5 * (dbox_lds[0] + dbox_lds[4]) - cx, dy = 0.5 * (dbox_lds[1] + dbox_lds[5]) - cy;
    const double reach = 5.2;   // 2 * half-diagonal (2.5378) + slack
    if (dx * dx + dy * dy > reach * reach) return 0.0;
    double area = 0.0;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) { sh[i] = box.x[i]; sh[16 + i] = box.y[i]; }
        area = quad_intersection_area_lane0(dbox_lds, sh);
    }
    return __shfl(area, 0);
}

// Exact necessary condition for |hull ∩ dest| / |dest| > 0.95 (car_parking_base.py:164-170) that avoids the
// polygon clip: hull ∩ dest lies inside the hull AND inside the slab spanned by dest's projection on each hull
// axis, so area <= width x (overlap length along the long axis) and <= length x (overlap along the short axis).
// If either overlap is below 94 % of the hull extent the ratio is < 0.95 whatever the rounding.
__device__ __forceinline__ bool arrival_possible(double x, double y, double ct, double sn, double dcx, double dcy,
                                                 double cd, double sd) {
    const double mid = 0.5 * (CAR_XF + CAR_XR), hl = 0.5 * (CAR_XF - CAR_XR), hw = CAR_YH;
    double dx = dcx - (x + ct * mid), dy = dcy - (y + sn * mid);
    double cphi = fabs(ct * cd + sn * sd), sphi = fabs(ct * sd - sn * cd);
    double p = dx * ct + dy * sn, q = dy * ct - dx * sn;
    double hB = hl * cphi + hw * sphi, wB = hl * sphi + hw * cphi;
    double lov = fmin(hl, p + hB) - fmax(-hl, p - hB);
    double wov = fmin(hw, q + wB) - fmax(-hw, q - wB);
    return lov >= 0.94 * 2 * hl && wov >= 0.94 * 2 * hw;
}

// Obstacles that can touch the hull during this step: the hull stays inside the disc of radius `reach`
// about the step's start position (rear axle travels <= 1.25 m, hull radius about the axle 3.883 m).
// Compacts their indices into list[]; exact (bounding boxes only, never drops a candidate).
__device__ __forceinline__ int build_near_list(const double* tile, int n_obst, double x0, double y0, double reach,
                                               int* list, int lane) {
    int cnt = 0;
    for (int base = 0; base < n_obst; base += WAVE) {
        int o = base + lane;
        bool near = false;
        if (o < n_obst) {
            const double* v = tile + 8 * o;
            double mnx = fmin(fmin(v[0], v[2]), fmin(v[4], v[6])), mxx = fmax(fmax(v[0], v[2]), fmax(v[4], v[6]));
            double mny = fmin(fmin(v[1], v[3]), fmin(v[5], v[7])), mxy = fmax(fmax(v[1], v[3]), fmax(v[5], v[7]));
            near = !(mnx > x0 + reach || mxx < x0 - reach || mny > y0 + reach || mxy < y0 - reach);
        }
        unsigned long long m = __ballot(near);
        if (near) list[cnt + __popcll(m & ((1ull << lane) - 1))] = o;
        cnt += __popcll(m);
    }
    return cnt;
}

// the same with a box [bx0, bx1] x [by0, by1] instead of the disc's box
__device__ __forceinline__ int build_near_list_box(const double* tile, int n_obst, double bx0, double bx1, double by0, double by1,
                                                   int* list, int lane) {
    int cnt = 0;
    for (int base = 0; base < n_obst; base += WAVE) {
        int o = base + lane;
        bool near = false;
        if (o < n_obst) {
            const double* v = tile + 8 * o;
            double mnx = fmin(fmin(v[0], v[2]), fmin(v[4], v[6])), mxx = fmax(fmax(v[0], v[2]), fmax(v[4], v[6]));
            double mny = fmin(fmin(v[1], v[3]), fmin(v[5], v[7])), mxy = fmax(fmax(v[1], v[3]), fmax(v[5], v[7]));
            near = !(mnx > bx1 || mxx < bx0 || mny > by1 || mxy < by0);
        }
        unsigned long long m = __ballot(near);
        if (near) list[cnt + __popcll(m & ((1ull << lane) - 1))] = o;
        cnt += __popcll(m);
    }
    return cnt;
}

// The two-launch form of the step kernel never stages the whole tile: the obstacles whose precomputed box (float32, rounded
// outwards: a superset) meets [bx0, bx1] x [by0, by1] are found from 16 bytes per obstacle, and only THEIR vertices are
// copied to LDS, compacted: tile slot i = i-th such obstacle, list[i] = i.  Returns their number.
__device__ __forceinline__ int stage_near(const float4* obb, const double2* src, int n_obst, double bx0, double bx1,
                                          double by0, double by1, double* tile, int* list, int lane,
                                          const uint8_t* gflags = nullptr, uint8_t* lflags = nullptr, const float4* pre = nullptr) {
    // pre: the box of obstacle `lane` (the first chunk), requested by the caller together with the scene's first loads
    int cnt = 0;
    for (int base = 0; base < n_obst; base += WAVE) {
        const int o = base + lane;
        bool near = false;
        if (o < n_obst) {
            const float4 bb = (pre && base == 0) ? *pre : obb[o];
            near = !((double)bb.x > bx1 || (double)bb.y < bx0 || (double)bb.z > by1 || (double)bb.w < by0);
        }
        const unsigned long long m = __ballot(near);
        if (near) list[cnt + __popcll(m & ((1ull << lane) - 1))] = o;
        cnt += __popcll(m);
    }
    ssync();
    double2* dst = (double2*)tile;
    for (int i = lane; i < 4 * cnt; i += WAVE) {
        dst[i] = src[4 * list[i >> 2] + (i & 3)];
        if (gflags && (i & 3) == 0) lflags[i >> 2] = gflags[list[i >> 2]];       // the staged obstacles' shape flags (lidar cull)
    }
    ssync();
    for (int i = lane; i < cnt; i += WAVE) list[i] = i;
    return cnt;
}

// _detect_collision (car_parking_base.py:153-158): any hull edge x any obstacle edge share a point.
// Edges are taken from the obstacles in list[0..n_list).  The hull is that of pose (px, py, cos, sin).  Fast path: the
// orientation filter (segments_intersect_fast); a lane whose pair it leaves undecided -- and only if no lane has a certain hit --
// takes the robust path, one lane at a time, with the LDS work area xl[ROBUST_LDS_WORDS].
__device__ __forceinline__ bool detect_collision(double px, double py, double ct, double st, const double* tile, const int* list,
                                                 int n_list, double* xl, int lane, unsigned* n_undecided = nullptr) {
    const int n_slots = 4 * n_list;
    const Box b = make_box(px, py, ct, st);
    double hminx = fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3]));
    double hmaxx = fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3]));
    double hminy = fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3]));
    double hmaxy = fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3]));
#pragma unroll 1
    for (int base = 0; base < n_slots; base += WAVE) {
        int e = base + lane;
        bool hit = false, und = false;
        double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        if (e < n_slots) {
            const double* v = tile + 8 * list[e >> 2];
            int j = e & 3, j2 = (e + 1) & 3;
            x1 = v[2 * j]; y1 = v[2 * j + 1]; x2 = v[2 * j2]; y2 = v[2 * j2 + 1];
            // envelope of the obstacle edge vs envelope of the hull: necessary for any segment pair
            if (!(fmin(x1, x2) > hmaxx || fmax(x1, x2) < hminx || fmin(y1, y2) > hmaxy || fmax(y1, y2) < hminy)) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int k2 = (k + 1) & 3;
                    const int r = segments_intersect_fast(b.x[k], b.y[k], b.x[k2], b.y[k2], x1, y1, x2, y2);
                    hit = hit || r == 1;
                    und = und || r == 2;
                }
            }
        }
        if (__any(hit)) return true;
        unsigned long long um = __ballot(und);
        if (um) {
            if (n_undecided) *n_undecided += __popcll(um);
            bool hit2 = false;
            while (um) {
                const int l = __ffsll((long long)um) - 1;
                um &= um - 1;
                if (lane == l) hit2 = hull_edge_intersect_robust(px, py, ct, st, x1, y1, x2, y2, xl);
            }
            if (__any(hit2)) return true;
        }
    }
    return false;
}

constexpr float SPAN_ANGLE_C[6] = {0.9999772310256958f, -0.33262282609939575f, 0.19354036450386047f,
                                   -0.11642643809318542f, 0.05264730006456375f, -0.01171911507844925f};
// Direction of (x, y) for the lidar's beam-span FILTER (not a reference quantity): atan2 to within 4e-6 rad -- a degree-11 odd
// minimax polynomial of min/max (1.75e-6 in float32 over [0, 1], tests/test_lidar_span_angle.py) behind the hardware reciprocal,
// ~17 vector instructions where the library's correctly-rounded-to-an-ulp atan2f takes ~40.  NaN for (0, 0).
__device__ __forceinline__ float span_angle(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float q = __builtin_fmaf(s, SPAN_ANGLE_C[5], SPAN_ANGLE_C[4]);
    q = __builtin_fmaf(q, s, SPAN_ANGLE_C[3]);
    q = __builtin_fmaf(q, s, SPAN_ANGLE_C[2]);
    q = __builtin_fmaf(q, s, SPAN_ANGLE_C[1]);
    q = __builtin_fmaf(q, s, SPAN_ANGLE_C[0]);
    float r = q * t;
    if (ay > ax) r = 1.57079633f - r;
    if (x < 0.0f) r = 3.14159265f - r;
    return copysignf(r, y);
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

// Exact-safe necessary conditions for beam (a, b) = (sin th, -cos th) to register a hit on edge (x1,y1)-(x2,y2):
//  * the reference accepts a hit only if the intersection of the beam LINE with the edge LINE lies inside the
//    edge's box, i.e. on the segment, so the end points lie on opposite sides of (or on) the beam line; both
//    clearly on one side (signed distance beyond 1e-9 m, ~1e4 x the rounding of raw_x / raw_y) -> no hit;
//  * a point of the segment that is on the beam line has a forward coordinate t between the end points'; if both
//    are < -1e-6 the point is behind the sensor and raw_x or raw_y violates the +-1e-8 quadrant test (:120-124).
__device__ __forceinline__ bool beam_may_hit(double a, double b, double x1, double y1, double x2, double y2) {
    double s1 = a * x1 + b * y1, s2 = a * x2 + b * y2;
    const double eps = 1e-9, back = -1e-6;
    if ((s1 > eps && s2 > eps) || (s1 < -eps && s2 < -eps)) return false;
    double t1 = a * y1 - b * x1, t2 = a * y2 - b * x2;
    if (t1 < back && t2 < back) return false;
    return true;
}

// one (beam, edge) pair of _fast_calc_lidar_obs (lidar_simulator.py:98-133); returns the SQUARE of the range (raw_x^2 + raw_y^2 as :131
// forms it) or +inf: the square root is monotone, so the beam's minimum is taken over the squares and rooted once per beam
__device__ __forceinline__ double beam_edge(int i, double a, double b, double x1, double y1, double x2, double y2,
                                            double d, double e, double f) {
    double det = a * e - b * d;
    if (det == 0) return INFINITY;
    // (b f - 0 e) / det and (0 d - a f) / det of :110-111 without the zero products: for finite coordinates they change at most the
    // SIGN of a zero numerator, and nothing below looks at the sign of a zero
    double raw_x = (b * f) / det;
    double raw_y = (-(a * f)) / det;
    const double tz = 1e-8;
    // the quadrant tests of :120-124 with the beam's signs as factors (a product with +-1 is exact, the compare flips with it; a NaN
    // fails neither form): !(raw_x < -tz) for the beams looking towards +x, !(raw_x > tz) = !(-raw_x < -tz) for the others
    const double sx = (i < NBEAM / 4 || i >= NBEAM / 4 * 3) ? 1.0 : -1.0, sy = (i < NBEAM / 2) ? 1.0 : -1.0;
    bool ok = !(sx * raw_x < -tz) && !(sy * raw_y < -tz);
    ok = ok && !(raw_x > fmax(x1, x2)) && !(raw_x < fmin(x1, x2));
    ok = ok && !(raw_y > fmax(y1, y2)) && !(raw_y < fmin(y1, y2));
    return ok ? raw_x * raw_x + raw_y * raw_y : INFINITY;
}

// ------------------------------------------------------------------------------------------------------------
// k_kinematics: FOUR LANES PER SCENE (16 scenes per wave).  action_rescale (env_wrapper.py:37-50), KSModel clip
// (vehicle.py:85-86) and the 10 x 20 explicit-Euler micro-steps (vehicle.py:88-93) form a strictly sequential
// float64 recurrence per scene: h += dh and x += v cos(h) dt/20 accumulate rounding step by step.  The recurrence
// is kept exactly, but its expensive part is spread over a quad: lane q walks the SAME heading chain and stops at
// micro-steps m = 4r + q (4 sequential additions per round, so every lane sees the reference's partial sums), all
// four evaluate sincos + the displacement terms of "their" micro-step at once, and the x / y sums then consume
// the four terms in micro-step order through quad broadcasts.  The ten sub-step poses do not depend on collisions
// (only where the step stops does), so k_env_step just reads them: kin[scene][50] = h[10] cos[10] sin[10] x[10]
// y[10], written through LDS with coalesced stores.
// ------------------------------------------------------------------------------------------------------------
constexpr int KIN_SCENES_PER_BLOCK = WAVE / 4;

// k / NITER for the action mask's step counts 0 .. NITER (the host compiler's IEEE quotients)
__constant__ const double MASK_STEP_FRACTION[NITER + 1] = {0.0 / NITER, 1.0 / NITER, 2.0 / NITER, 3.0 / NITER, 4.0 / NITER, 5.0 / NITER,
                                                           6.0 / NITER, 7.0 / NITER, 8.0 / NITER, 9.0 / NITER, 10.0 / NITER};
static_assert(NITER == 10, "MASK_STEP_FRACTION lists NITER + 1 quotients");

// inclusive prefix sum over the wave: Hillis-Steele inside each row of 16 (DPP row_shr), row totals by readlane
__device__ __forceinline__ int wave_incl_scan_i(int x, int lane) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47);
    return x + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}

// DPP move of a double (two 32-bit moves), e.g. CTRL = quad_perm
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// value of lane J of the caller's quad (DPP quad_perm [J,J,J,J])
template <int J>
__device__ __forceinline__ double quad_bcast(double v) {
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), J * 0x55, 0xf, 0xf, true);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), J * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// x / 20.0 exactly as IEEE division rounds it, in three instructions instead of the ~11 of the general division (two per round of
// the micro-step loop).  R = RN(1/20) = RN(0.8) 2^-4 has relative error 2^-54 (0.8 = 0x3FE999999999999A, rounded up by 0.4 ulp).
// q0 = RN(x R) lies within one ulp of x / 20 (0.4 ulp from R + 0.5 ulp rounding for quotients in [1, 1.6) 2^k; 0.5 + 0.5 in
// [0.8, 1) 2^k), so the remainder r = x - 20 q0 is representable and the fma delivers it exactly; x / 20 = q0 + r / 20, and
// q0 + r R differs from it by at most 2^-54 ulp.  That cannot move it across a rounding boundary: with x = X 2^e (X a 53-bit
// integer) the quotient is 4 X / 5 or 8 X / 5 ulps, whose distance to any midpoint m + 1/2 is an odd integer over 10: >= 0.1 ulp.
// So RN(q0 + r R) = RN(x / 20) for every x without under- / overflow (the terms here are ~1e-3 .. 1e-1).  Pose parity (tolerance
// 0.0, tests/test_gpu_parity.py) compares the sums of these quotients with the oracle's plain divisions.
static_assert(MINI_ITER == 20, "div_by_20");
__device__ __forceinline__ double div_by_20(double x) {
    const double R = 0.05;
    const double q0 = x * R;
    const double r = fma(-q0, 20.0, x);
    return copysign(fma(r, R, q0), x);                   // (the quotient has x's sign; the fma chain turns -0 into +0)
}

// action_rescale (env_wrapper.py:37-50) + KSModel's clip (vehicle.py:85-86): the speed and the heading increment per micro-step
template <typename AT>
__device__ __forceinline__ void kin_controls(const void* actions, int scene, uint32_t stages, double& speed, double& dh) {
    const AT* act = (const AT*)actions;
    const double a0 = (double)act[2 * (size_t)scene], a1 = (double)act[2 * (size_t)scene + 1];
    double steer = a0;
    speed = a1;
    if (!(stages & HOPE_ACTION_PHYSICAL)) {
        if (sizeof(AT) == 4 && (stages & HOPE_ACTION_RESCALE_F32)) {          // float32 Box arithmetic, term by term
            const float f0 = fminf(fmaxf((float)a0, -1.0f), 1.0f), f1 = fminf(fmaxf((float)a1, -1.0f), 1.0f);
            steer = (double)(f0 * ((float)STEER_HI - (float)STEER_LO) / 2.0f + ((float)STEER_HI + (float)STEER_LO) / 2.0f);
            speed = (double)(f1 * ((float)SPEED_HI - (float)SPEED_LO) / 2.0f + ((float)SPEED_HI + (float)SPEED_LO) / 2.0f);
        } else {
            steer = clipd(a0, -1, 1) * (STEER_HI - STEER_LO) / 2 + (STEER_HI + STEER_LO) / 2;
            speed = clipd(a1, -1, 1) * (SPEED_HI - SPEED_LO) / 2 + (SPEED_HI + SPEED_LO) / 2;
        }
    }
    speed = clipd(speed, SPEED_LO, SPEED_HI);
    steer = clipd(steer, STEER_LO, STEER_HI);
    dh = speed * hm_tan(steer) / WHEEL_BASE * STEP_LENGTH / MINI_ITER;
}

template <typename AT>
__global__ __launch_bounds__(64) void k_kinematics(int n, const int32_t* scene_list, const double* state, const void* actions,
                                                  const uint8_t* active, uint32_t stages, const double* scene_c, double* kin) {
    // n entries of scene_list (a tile class's dense list: the kinematics head each class's launch chain), or scenes 0..n-1
    if (HOPE_KIN_PRIO) __builtin_amdgcn_s_setprio(HOPE_KIN_PRIO);
    __shared__ double buf[KIN_SCENES_PER_BLOCK * KIN_WORDS];
    __shared__ double terms[MINI_ITER][2 * KIN_SCENES_PER_BLOCK];   // one round's displacement terms: [micro-step][2 scene + (x | y)]
    __shared__ int sid[KIN_SCENES_PER_BLOCK];
#ifdef HOPE_KIN_PAD                                      // A/B probe: extra LDS per block (does the kernel's LDS footprint matter in the pipelined step?)
    __shared__ double kin_pad[HOPE_KIN_PAD];
    if (n < 0) kin_pad[threadIdx.x] = 0.0, kin[0] = kin_pad[(threadIdx.x * 7) % HOPE_KIN_PAD];
#endif
    const int lane = threadIdx.x, q = lane & 3, ls = lane >> 2;
    const int idx = blockIdx.x * KIN_SCENES_PER_BLOCK + ls;
    const int scene = idx < n ? (scene_list ? scene_list[idx] : idx) : -1;
    const bool live = scene >= 0 && !(active && !active[scene]);
    if (__all(!live)) return;
    if (q == 0) sid[ls] = live ? scene : -1;
    const int sc_ = live ? scene : 0;
    const double* st = state + (size_t)sc_ * ST_WORDS;
    double h = st[2];
    // the position sums live one per lane: lane 2 s + c (< 32) owns x (c = 0) or y (c = 1) of the block's scene s
    double acc = 0.0;
    if (lane < 2 * KIN_SCENES_PER_BLOCK) {
        const int idx2 = blockIdx.x * KIN_SCENES_PER_BLOCK + (lane >> 1);
        const int sc2 = idx2 < n ? (scene_list ? scene_list[idx2] : idx2) : 0;
        acc = state[(size_t)sc2 * ST_WORDS + (lane & 1)];
    }
    double speed, dh;
    kin_controls<AT>(actions, sc_, stages, speed, dh);
    double* out = buf + ls * KIN_WORDS;
    for (int i = 0; i < q; i++) h = h + dh;                  // lane q starts at micro-step q
    // Round r = sub-step r: its 20 micro-steps 20 r + 4 j + q, j = 0 .. 4, FIVE per lane.  The five headings of a lane (four steps
    // of the sequential chain apart) give five independent sincos evaluations per round for the scheduler to interleave (it was
    // ONE dependent sincos chain per round of four micro-steps, 50 rounds).  Same values, same order of every sum.  Measured:
    // 0.0759 -> 0.0731 ms per step for the two launches, the step itself unchanged -- with three waves per SIMD the class-0 launch
    // is bound by its ~5 500 vector instructions per wave (200 sincos per scene), not by their latency.
    // The sums x += ..., y += ... (vehicle.py:90-91, strictly in micro-step order) are NOT formed in every lane of the quad from
    // quad broadcasts (8 additions + 16 DPP moves per micro-step group and wave, a quarter of the kernel's vector instructions; the
    // step is VALU-issue bound): the round's 20 x 2 terms per scene go through LDS and 32 lanes -- one per (scene, coordinate) --
    // add them in order, 20 additions per round and wave.
    constexpr int KJ = MINI_ITER / 4;
    static_assert(MINI_ITER % 4 == 0, "micro-steps per lane and round");
#pragma unroll 1
    for (int r = 0; r < NUM_STEP; r++) {
        double hj[KJ], sj[KJ], cj[KJ];
        hj[0] = h;
#pragma unroll
        for (int j = 1; j < KJ; j++) { double t = hj[j - 1]; t = t + dh; t = t + dh; t = t + dh; t = t + dh; hj[j] = t; }
#pragma unroll
        for (int j = 0; j < KJ; j++) hm_sincos(hj[j], &sj[j], &cj[j]);
        if (q == 0 && r > 0) {                               // lane 0 sits on a sub-step boundary: h_{20 r}
            const int k = r - 1;
            out[k] = hj[0]; out[10 + k] = cj[0]; out[20 + k] = sj[0];
        }
#pragma unroll
        for (int j = 0; j < KJ; j++) {
            terms[4 * j + q][2 * ls] = div_by_20(speed * cj[j] * STEP_LENGTH);      // ... / MINI_ITER, correctly rounded (below)
            terms[4 * j + q][2 * ls + 1] = div_by_20(speed * sj[j] * STEP_LENGTH);
        }
        ssync();
        if (lane < 2 * KIN_SCENES_PER_BLOCK) {
#pragma unroll
            for (int m = 0; m < MINI_ITER; m++) acc += terms[m][lane];            // x += ..., y += ... in micro-step order
            buf[(lane >> 1) * KIN_WORDS + 30 + 10 * (lane & 1) + r] = acc;        // the position after sub-step r + 1
        }
        ssync();
        h = hj[KJ - 1];
        h = h + dh; h = h + dh; h = h + dh; h = h + dh;      // four steps of the sequential chain
    }
    if (q == 0) {                                            // after micro-step 199: the 10th sub-step pose
        double s_, c_;
        hm_sincos(h, &s_, &c_);
        out[9] = h; out[19] = c_; out[29] = s_;
    }
    __syncthreads();
    {   // the slab bound of the arrival test (arrival_possible) for the ten poses: lane q takes poses q, q + 4, q + 8.  In
        // k_env_step (one wave per scene) this scalar test cost the whole wave ~40 instructions per pass of its sub-step loop.
        const double* sc = scene_c + (size_t)sc_ * SC_WORDS;
        const double dcx = sc[SC_DCEN], dcy = sc[SC_DCEN + 1], dcd = sc[SC_DCEN + 2], dsd = sc[SC_DCEN + 3];
        double s0, c0;
        hm_sincos(st[2], &s0, &c0);
        int bits = 0;
        for (int k = q; k < NUM_STEP; k += 4)
            if (arrival_possible(out[30 + k], out[40 + k], out[10 + k], out[20 + k], dcx, dcy, dcd, dsd)) bits |= 1 << k;
        // bit NUM_STEP: the pose the step starts from (where a step blocked at its first sub-step stays; read by k_motion_pair)
        if (q == 3 && arrival_possible(st[0], st[1], c0, s0, dcx, dcy, dcd, dsd)) bits |= 1 << NUM_STEP;
        {   // bits 16 + k: the rear axle of pose k lies inside the map box (the OUTBOUND test of _check_status, car_parking_base.py:178-179,
            // same comparisons on the same doubles); bit 16 + NUM_STEP: the start pose.  Read by k_motion_pair.
            const double xmin = sc[SC_BBOX], xmax = sc[SC_BBOX + 1], ymin = sc[SC_BBOX + 2], ymax = sc[SC_BBOX + 3];
            for (int k = q; k < NUM_STEP; k += 4)
                if (!(out[30 + k] > xmax || out[30 + k] < xmin || out[40 + k] > ymax || out[40 + k] < ymin)) bits |= 1 << (16 + k);
            if (q == 3 && !(st[0] > xmax || st[0] < xmin || st[1] > ymax || st[1] < ymin)) bits |= 1 << (16 + NUM_STEP);
        }
        bits |= __builtin_amdgcn_mov_dpp(bits, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
        bits |= __builtin_amdgcn_mov_dpp(bits, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
        if (q == 0) { out[50] = __hiloint2double(0, bits); out[51] = 0.0; }
        // box around every hull of this step (start pose + ten poses): k_env_step's near list keeps the obstacles whose box
        // meets it -- far fewer than a disc about the start pose, so more sub-steps fit in one pass of its collision loop
        Box b = make_box(st[0], st[1], c0, s0);
        double bx0 = fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3])), bx1 = fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3]));
        double by0 = fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3])), by1 = fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3]));
        for (int k = q; k < NUM_STEP; k += 4) {
            b = make_box(out[30 + k], out[40 + k], out[10 + k], out[20 + k]);
            bx0 = fmin(bx0, fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3]))); bx1 = fmax(bx1, fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3])));
            by0 = fmin(by0, fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3]))); by1 = fmax(by1, fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3])));
        }
        bx0 = fmin(bx0, dpp_d<0xB1>(bx0)); bx0 = fmin(bx0, dpp_d<0x4E>(bx0));
        bx1 = fmax(bx1, dpp_d<0xB1>(bx1)); bx1 = fmax(bx1, dpp_d<0x4E>(bx1));
        by0 = fmin(by0, dpp_d<0xB1>(by0)); by0 = fmin(by0, dpp_d<0x4E>(by0));
        by1 = fmax(by1, dpp_d<0xB1>(by1)); by1 = fmax(by1, dpp_d<0x4E>(by1));
        if (q == 0) { out[52] = bx0; out[53] = bx1; out[54] = by0; out[55] = by1; }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < KIN_SCENES_PER_BLOCK * KIN_WORDS; w += WAVE) {     // one 400-byte row per scene
        const int s = w / KIN_WORDS;
        const int sc = sid[s];
        if (sc >= 0) kin[(size_t)sc * KIN_WORDS + (w - s * KIN_WORDS)] = buf[w];
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same kinematics by the scene's OWN wave: the one-launch form of the step kernel at small batches (k_env_step<.., PART 0,
// FKIN>), where the step is a chain of launch latencies and a separate kinematics launch costs the critical stream its ~16 us
// plus the gap to the next launch.  Same values as k_kinematics, bit for bit: lane L walks the SAME sequential heading chain to the
// micro-steps m = L + 64 i (i = 0 .. 3, m <= 200: 255 dependent additions, ~1 us, instead of 200 in each of 4 lanes), evaluates
// sincos and the displacement terms of its (at most four) micro-steps, and lanes 0 / 1 add the x / y terms in micro-step order from
// LDS (two phases of 128 and 72 micro-steps in region A).  The sub-step poses land where the motion part reads them
// (scr[LDS_HB ..]: h cos sin x y of poses 0 .. 9); returns arrival_possible's bits and the box around the step's hulls.
// ------------------------------------------------------------------------------------------------------------
template <typename AT>
__device__ __forceinline__ void wave_kinematics(const void* actions, int scene, uint32_t stages, double x0, double y0, double h0,
                                                const double* sc, double* scr, int lane, int& apmask, double (&kb)[4]) {
    double speed, dh;
    kin_controls<AT>(actions, scene, stages, speed, dh);
    constexpr int NM = NUM_STEP * MINI_ITER;             // 200 micro-steps; heading h_m is the one micro-step m starts from
    constexpr int NJ = (NM + WAVE) / WAVE;               // micro-steps per lane (the last one also covers m = NM, the final heading)
    static_assert(NJ == 4 && NM % WAVE < WAVE, "lane L owns micro-steps L + 64 i");
    double hm[NJ], sm[NJ], cm[NJ];
    {
        double hh = h0;
        for (int i = 0; i < lane; i++) hh = hh + dh;      // h_L
        hm[0] = hh;
#pragma unroll
        for (int j = 1; j < NJ; j++) {
#pragma unroll 8
            for (int i = 0; i < WAVE; i++) hh = hh + dh;
            hm[j] = hh;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) hm_sincos(hm[j], &sm[j], &cm[j]);
#pragma unroll
    for (int j = 0; j < NJ; j++) {                        // the sub-step boundaries m = 20 (k + 1): pose k's heading, cos, sin
        const int m = lane + WAVE * j;
        if (m >= MINI_ITER && m <= NM && m % MINI_ITER == 0) {
            const int k = m / MINI_ITER - 1;
            scr[LDS_HB + k] = hm[j]; scr[LDS_CB + k] = cm[j]; scr[LDS_SB + k] = sm[j];
        }
    }
    double acc = lane == 0 ? x0 : y0;                     // lane 0: x, lane 1: y
#pragma unroll
    for (int ph = 0; ph < 2; ph++) {                      // micro-steps [0, 128) and [128, 200): 2 x 128 doubles of region A
        ssync();
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int j = 2 * ph + jj, m = lane + WAVE * j;
            if (m < NM) {
                scr[2 * (m - 2 * WAVE * ph)] = div_by_20(speed * cm[j] * STEP_LENGTH);       // ... / MINI_ITER, correctly rounded
                scr[2 * (m - 2 * WAVE * ph) + 1] = div_by_20(speed * sm[j] * STEP_LENGTH);
            }
        }
        ssync();
        if (lane < 2) {
            const int m_end = ph == 0 ? 2 * WAVE : NM;
#pragma unroll 4
            for (int m = 2 * WAVE * ph; m < m_end; m++) {
                acc += scr[2 * (m - 2 * WAVE * ph) + lane];                                   // x += ..., y += ... (vehicle.py:90-91)
                if ((m + 1) % MINI_ITER == 0) scr[LDS_PX + NUM_STEP * lane + (m + 1) / MINI_ITER - 1] = acc;
            }
        }
    }
    ssync();
    // arrival_possible of the ten poses, the box around the hulls of the start pose and the ten poses: lane k = pose k, lane 10 = start
    const double c_0 = readlane_d(cm[0], 0), s_0 = readlane_d(sm[0], 0);                      // heading h_0 = the start heading
    const bool pose = lane < NUM_STEP;
    const int kk = pose ? lane : 0;
    const double px = pose ? scr[LDS_PX + kk] : x0, py = pose ? scr[LDS_PY + kk] : y0;
    const double pc = pose ? scr[LDS_CB + kk] : c_0, ps = pose ? scr[LDS_SB + kk] : s_0;
    const double dcx = sc[SC_DCEN], dcy = sc[SC_DCEN + 1], dcd = sc[SC_DCEN + 2], dsd = sc[SC_DCEN + 3];
    apmask = (int)(__ballot(pose && arrival_possible(px, py, pc, ps, dcx, dcy, dcd, dsd)) & ((1ull << NUM_STEP) - 1));
    const Box b = make_box(px, py, pc, ps);
    const bool in = lane <= NUM_STEP;
    double bx0 = in ? fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3])) : INFINITY, bx1 = in ? fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3])) : -INFINITY;
    double by0 = in ? fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3])) : INFINITY, by1 = in ? fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3])) : -INFINITY;
    // lanes 0 .. 15: quad, then the row's halves and the mirrored row (DPP)
    bx0 = fmin(bx0, dpp_d<0xB1>(bx0)); bx0 = fmin(bx0, dpp_d<0x4E>(bx0)); bx0 = fmin(bx0, dpp_d<0x141>(bx0)); bx0 = fmin(bx0, dpp_d<0x140>(bx0));
    bx1 = fmax(bx1, dpp_d<0xB1>(bx1)); bx1 = fmax(bx1, dpp_d<0x4E>(bx1)); bx1 = fmax(bx1, dpp_d<0x141>(bx1)); bx1 = fmax(bx1, dpp_d<0x140>(bx1));
    by0 = fmin(by0, dpp_d<0xB1>(by0)); by0 = fmin(by0, dpp_d<0x4E>(by0)); by0 = fmin(by0, dpp_d<0x141>(by0)); by0 = fmin(by0, dpp_d<0x140>(by0));
    by1 = fmax(by1, dpp_d<0xB1>(by1)); by1 = fmax(by1, dpp_d<0x4E>(by1)); by1 = fmax(by1, dpp_d<0x141>(by1)); by1 = fmax(by1, dpp_d<0x140>(by1));
    kb[0] = readlane_d(bx0, 0); kb[1] = readlane_d(bx1, 0); kb[2] = readlane_d(by0, 0); kb[3] = readlane_d(by1, 0);
    ssync();
}

// Cycle accounting of the step kernel (the <float, float, true> instantiation, launched when HOPE_STEP_TIMING is set;
// tools/step_timing.py): [0] staging + near list [1] sub-step loop [2] status, reward, turnover, outputs, target
// [3] lidar: ego transform + ring keep [4] lidar: per-edge beam ranges [5] lidar: enqueue [6] lidar: exact pairs (drain)
// [7] action mask [8] whole wave [9] waves
__device__ unsigned long long g_step_prof[64 * 16];
// Tie census (the same instrumented instantiation; tools/tie_census.py): how close the workload comes to the decisions whose
// arithmetic the reference delegates to GEOS -- [0] arrival-ratio evaluations [1] min |overlap / dest area - 0.95| (double bits)
// [2] lidar ring-keep evaluations [3] min |ring distance - 10 m| [4] (hull edge, obstacle edge) pairs an orientation filter
// left undecided (the exact path decided them) [6] action-mask table compares [7] min NON-ZERO |table entry - scan value| [8] scene-steps
// whose mask took the exact 1200-beam evaluation (a table entry within 1e-9 of the scan) [9] scene-steps [10] compares with entry == scan
// bit for bit (the structural tie: a touching obstacle clips the scan to the hull range the table was built from)
__device__ unsigned long long g_census[64 * 16];
#define CEN_ARR(ua_) do { if (TIMING) { cen_arr_n += 1; cen_arr = fmin(cen_arr, fabs((ua_) / dest_area - 0.95)); } } while (0)
#define ST_T0() unsigned long long t0_ = TIMING ? __builtin_readcyclecounter() : 0
#define ST_T(i) do { if (TIMING) { const unsigned long long t1_ = __builtin_readcyclecounter(); tsec[i] += t1_ - t0_; t0_ = t1_; } } while (0)
#define ST_FLUSH() do { if (TIMING) { tsec[8] = __builtin_readcyclecounter() - tstart_; tsec[9] = 1; \
        if (lane == 0) for (int i_ = 0; i_ < 16; i_++) if (tsec[i_]) atomicAdd(&g_step_prof[(blockIdx.x & 63) * 16 + i_], tsec[i_]); \
        double cr_ = cen_ring, cm_ = cen_mask; unsigned nr_ = cen_ring_n, nm_ = cen_mask_n, ne_ = cen_eq; \
        for (int o_ = 32; o_ > 0; o_ >>= 1) { cr_ = fmin(cr_, __shfl_xor(cr_, o_)); cm_ = fmin(cm_, __shfl_xor(cm_, o_)); nr_ += __shfl_xor(nr_, o_); nm_ += __shfl_xor(nm_, o_); ne_ += __shfl_xor(ne_, o_); } \
        if (lane == 0) { unsigned long long* c_ = g_census + (blockIdx.x & 63) * 16; \
            atomicAdd(&c_[0], (unsigned long long)cen_arr_n); atomicMin(&c_[1], (unsigned long long)__double_as_longlong(cen_arr)); \
            atomicAdd(&c_[2], (unsigned long long)nr_); atomicMin(&c_[3], (unsigned long long)__double_as_longlong(cr_)); \
            atomicAdd(&c_[4], (unsigned long long)cen_und); atomicAdd(&c_[6], (unsigned long long)nm_); \
            atomicMin(&c_[7], (unsigned long long)__double_as_longlong(cm_)); atomicAdd(&c_[8], (unsigned long long)cen_tie); atomicAdd(&c_[9], 1ull); atomicAdd(&c_[10], (unsigned long long)ne_); } } } while (0)

// PART: 0 = the whole scene-step; 1 = motion, status, turnover and state only (everything the Reeds-Shepp chain and k_post
// wait for); 2 = the observation only (lidar + action mask) of the pose PART 1 left in `state`.  With HOPE_F_OVERLAP the
// library launches 1 and 2 separately so that the observation runs NEXT TO the Reeds-Shepp kernels of the same tile class
// (k_rs_compact / k_rs_words / k_rs_segs are short on parallelism and left the GPU half empty when they ran alone).
template <typename OT, typename AT, bool TIMING = false, int PART = 0, bool FKIN = false>
__global__ __launch_bounds__(64, (PART == 0 && !TIMING) ? HOPE_PART0_OCC : 4) void k_env_step(StepParams p) {
    static_assert(!FKIN || PART == 0, "the wave's own kinematics: one-launch form only");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= p.n_list) return;
    unsigned long long tsec[16] = {};
    const unsigned long long tstart_ = TIMING ? __builtin_readcyclecounter() : 0;
    double cen_arr = INFINITY, cen_ring = INFINITY, cen_mask = INFINITY;      // tie census (TIMING builds only; dead code otherwise)
    unsigned cen_arr_n = 0, cen_ring_n = 0, cen_mask_n = 0, cen_und = 0, cen_tie = 0, cen_eq = 0;
    ST_T0();
    if (PART != 2 && p.rs_count_zero && blockIdx.x == 0 && threadIdx.x == 0) p.rs_count_zero[0] = 0;   // this class's queue length
    const int scene = p.scene_list[scene_of_block(blockIdx.x, p.n_list)];
    {
        const bool act = !(p.active && !p.active[scene]);
        if (PART != 2 && p.active_out && lane == 0) p.active_out[scene] = act;
        if (!act) return;
    }

    int n_obst = p.n_obst[scene];
    // the sub-step poses of this step (k_kinematics), requested together with everything else the scene needs
    const bool moving = PART != 2 && (p.stages & HOPE_STAGE_MOTION) && p.has_action;
    const double kinv = (moving && !FKIN && lane < KIN_WORDS) ? p.kin[(size_t)scene * KIN_WORDS + lane] : 0.0;

    double* tile = lds;
    double* scr = lds + 8 * p.tile_cap;
    int* keep = (int*)(scr + LDS_KEEP);
    uint8_t* cfl = (uint8_t*)(keep + ((p.tile_cap + 3) & ~3));     // shape flags of the staged obstacles (lidar)

    // ---- stage the scene: constants (192 B), state (32 B), obstacle tile (64 B x n_obst) ---------
    const double* sc = p.scene_c + (size_t)scene * SC_WORDS;
    const int n_slots = 4 * n_obst;
    const double2* src = (const double2*)(p.verts + (size_t)scene * p.max_obst * 8);
    const float4* obb_s = p.obb + (size_t)scene * p.max_obst;
    // (observation launch: this lane's two beam directions too -- four loads that waited a round trip of their own in front of the lidar)
    double bm0, bm1, bm2, bm3;
    if (PART == 2) {
        bm0 = p.beam_ab[2 * lane]; bm1 = p.beam_ab[2 * lane + 1];
        const bool h1 = lane + 64 < NBEAM;
        bm2 = h1 ? p.beam_ab[2 * (lane + 64)] : 0.0; bm3 = h1 ? p.beam_ab[2 * (lane + 64) + 1] : 0.0;
    }
    if (PART == 0) {
        double2* dst = (double2*)tile;
        for (int v = lane; v < n_slots; v += WAVE) dst[v] = src[v];   // 16 B/lane, coalesced
        // (the obstacles' shape flags for the lidar's back-face cull, with the tile: not a dependent load in the middle of the lidar)
        const uint32_t* gf = (const uint32_t*)(p.eflag + (size_t)scene * eflag_stride(p.max_obst));
        for (int i = lane; 4 * i < n_obst; i += WAVE) ((uint32_t*)cfl)[i] = gf[i];
    }
    // the first chunk of obstacle boxes of the near-obstacle scan (stage_near), requested NOW with the scene's other first loads: only
    // the launch's tile capacity of them (the small-tile class: 32 slots = 512 B; slots beyond n_obst belong to the scene, unused)
    float4 obb_pre = make_float4(0, 0, 0, 0);
    if (PART != 0 && lane < min(p.tile_cap, WAVE)) obb_pre = obb_s[lane];
    double* dbox = scr + LDS_DBOX;
    double* xl = scr + LDS_ROBUST;                       // work area of the robust collision path (one lane at a time)
    if (PART != 2 && lane < 8) dbox[lane] = sc[SC_DBOX + lane];
    double dest_area = PART != 2 ? sc[SC_DAREA] : 0.0;
    double* st = p.state + (size_t)scene * ST_WORDS;
    double x = st[0], y = st[1], h = st[2], accum = st[3];
    const double prev_x = x, prev_y = y, prev_h = h;      // prev_state (car_parking_base.py:255)
    int t = p.tstep[scene];
    double ct = 0, sn = 0;       // cos/sin of the final heading
    if (PART == 2) { ct = p.cs[2 * (size_t)scene]; sn = p.cs[2 * (size_t)scene + 1]; }   // hm_sincos(h) as the motion launch computed it
    ssync();
    // the sub-step poses: from k_kinematics' record, or (small batches, FKIN) by this wave itself while the tile's loads are in flight
    int apmask_k = 0;
    double kb[4] = {0.0, 0.0, 0.0, 0.0};                 // box around the step's hulls
    if (FKIN && moving) wave_kinematics<AT>(p.actions, scene, p.stages, x, y, h, sc, scr, lane, apmask_k, kb);
    else if (moving) { kb[0] = readlane_d(kinv, 52); kb[1] = readlane_d(kinv, 53); kb[2] = readlane_d(kinv, 54); kb[3] = readlane_d(kinv, 55); }

    if (PART != 2) {
    bool arrive = false, moved = false;
    bool known_free = false;     // final pose already passed _detect_collision in the sub-step loop
    bool have_ua = false;        // overlap area of the final pose already computed
    double ua = 0.0;
    bool have_cs = false;
    const double dcx = sc[SC_DCEN], dcy = sc[SC_DCEN + 1], dcd = sc[SC_DCEN + 2], dsd = sc[SC_DCEN + 3];
    int* nlist = keep;           // near-obstacle list (motion/status); the lidar reuses the words as keep flags
    int apmask = 0, kf_pose = -1;    // kf_pose: the final pose is sub-step pose kf_pose of this step (-1: the pose the step started from)
    // moving: the box around every hull of this step (k_kinematics); else the hull's disc about the rear axle (3.883 m + slack)
    int n_near;
    if (PART == 0) {
        if (moving)
            n_near = build_near_list_box(tile, n_obst, kb[0], kb[1], kb[2], kb[3], nlist, lane);
        else n_near = build_near_list(tile, n_obst, x, y, 3.9, nlist, lane);
    } else if (moving)
        n_near = stage_near(obb_s, src, n_obst, kb[0], kb[1], kb[2], kb[3], tile, nlist, lane, nullptr, nullptr, &obb_pre);
    else n_near = stage_near(obb_s, src, n_obst, x - 3.9, x + 3.9, y - 3.9, y + 3.9, tile, nlist, lane, nullptr, nullptr, &obb_pre);
    ssync();
    ST_T(0);

    if (moving) {
        // the ten sub-step poses (x, y, heading, cos, sin) were produced by k_kinematics (one THREAD per scene):
        // they do not depend on the collision outcome, only where we stop does
        if (!FKIN && lane < 50) scr[LDS_HB + lane] = kinv;
        apmask = FKIN ? apmask_k : __builtin_amdgcn_readlane(__double2loint(kinv), 50);        // arrival_possible of the ten poses (k_kinematics)
        ssync();

        // ---- sub-step loop (car_parking_base.py:259-271) ------------------------------------------
        // The ten poses are known, so several sub-steps can be examined per pass when the near list is short:
        // S = edge slots per sub-step (power of two >= 4 n_near), G = 64 / S sub-steps per pass, lane = (g, e).
        // Two ballots (arrival possible / collision) are then walked in sub-step order, which reproduces
        // "arrived? -> stop; collided? -> retreat and stop" exactly.
        const int S = n_near <= 1 ? 4 : (n_near <= 2 ? 8 : (n_near <= 4 ? 16 : (n_near <= 8 ? 32 : 64)));
        int ev_k = NUM_STEP;              // first sub-step with an event (NUM_STEP: none)
        bool ev_arrive = false;
        if (n_near == 0 && (apmask & ((1 << NUM_STEP) - 1)) == 0) {
            // no obstacle near the step's hulls and no pose the slab bound lets arrive: the step ends at the tenth pose (round 6)
        } else if (S < WAVE) {
            const int G = WAVE / S;
            const int g = lane / S, e = lane % S;
            const bool has_edge = e < 4 * n_near;
            double ex1 = 0, ey1 = 0, ex2 = 0, ey2 = 0;
            if (has_edge) {
                const double* v = tile + 8 * nlist[e >> 2];
                const int j = e & 3, j2 = (e + 1) & 3;
                ex1 = v[2 * j]; ey1 = v[2 * j + 1]; ex2 = v[2 * j2]; ey2 = v[2 * j2 + 1];
            }
            const unsigned long long gmask0 = (1ull << S) - 1;
            for (int k0 = 0; k0 < NUM_STEP && ev_k == NUM_STEP; k0 += G) {
                const int k = k0 + g;
                const bool kv = k < NUM_STEP;
                const int kk = kv ? k : NUM_STEP - 1;
                const double qx = scr[LDS_PX + kk], qy = scr[LDS_PY + kk], qc = scr[LDS_CB + kk], qs = scr[LDS_SB + kk];
                const bool ap = kv && e == 0 && ((apmask >> kk) & 1);
                bool hit = false, und = false;
                if (kv && has_edge) {
                    const Box b = make_box(qx, qy, qc, qs);
                    const double hminx = fmin(fmin(b.x[0], b.x[1]), fmin(b.x[2], b.x[3]));
                    const double hmaxx = fmax(fmax(b.x[0], b.x[1]), fmax(b.x[2], b.x[3]));
                    const double hminy = fmin(fmin(b.y[0], b.y[1]), fmin(b.y[2], b.y[3]));
                    const double hmaxy = fmax(fmax(b.y[0], b.y[1]), fmax(b.y[2], b.y[3]));
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
                {   // pairs the orientation filter left open (practically never): the robust path, one lane at a time
                    unsigned long long um = __ballot(und && !hit);
                    if (TIMING) cen_und += __popcll(um);
                    while (um) {
                        const int l = __ffsll((long long)um) - 1;
                        um &= um - 1;
                        if (lane == l) hit = hull_edge_intersect_robust(qx, qy, qc, qs, ex1, ey1, ex2, ey2, xl);
                    }
                }
                const unsigned long long hm = __ballot(hit), am = __ballot(ap);
                if (hm | am) {
                    for (int gg = 0; gg < G && k0 + gg < NUM_STEP; gg++) {
                        const unsigned long long gm = gmask0 << (gg * S);
                        const int kq = k0 + gg;
                        if (am & gm) {                                                   // _check_arrived :164-170
                            ua = overlap_area(scr[LDS_PX + kq], scr[LDS_PY + kq], scr[LDS_CB + kq], scr[LDS_SB + kq], dbox, scr + LDS_SH, lane);
                            CEN_ARR(ua);
                            if (ua / dest_area > 0.95) { ev_k = kq; ev_arrive = true; break; }
                        }
                        if (hm & gm) { ev_k = kq; break; }                               // _detect_collision :264
                    }
                }
            }
        } else {
            for (int k = 0; k < NUM_STEP; k++) {
                const double qx = scr[LDS_PX + k], qy = scr[LDS_PY + k], qc = scr[LDS_CB + k], qs = scr[LDS_SB + k];
                if ((apmask >> k) & 1) {                                                  // _check_arrived :164-170
                    ua = overlap_area(qx, qy, qc, qs, dbox, scr + LDS_SH, lane);
                    CEN_ARR(ua);
                    if (ua / dest_area > 0.95) { ev_k = k; ev_arrive = true; break; }
                }
                if (detect_collision(qx, qy, qc, qs, tile, nlist, n_near, xl, lane, TIMING ? &cen_und : nullptr)) { ev_k = k; break; }  // _detect_collision :264
            }
        }
        // final pose of the motion: the arrival pose, the pose BEFORE the colliding sub-step (retreat :264-271), or
        // the tenth pose
        arrive = ev_arrive;
        const int kf = ev_arrive ? ev_k : (ev_k == NUM_STEP ? NUM_STEP - 1 : ev_k - 1);
        moved = kf >= 0;
        kf_pose = kf;
        if (kf >= 0) {
            x = scr[LDS_PX + kf]; y = scr[LDS_PY + kf]; h = scr[LDS_HB + kf];
            ct = scr[LDS_CB + kf]; sn = scr[LDS_SB + kf];
            have_cs = true;
            known_free = !ev_arrive;            // passed its own collision test (or is the arrival pose: not needed)
        }
        have_ua = ev_arrive;
    }
    t += 1;                                                             // :277

    if (!have_cs) hm_sincos(h, &sn, &ct);

    ST_T(1);
    // ---- status (:279-282, _check_status :175-184) -------------------------------------------------
    int status = HOPE_STATUS_CONTINUE;
    if (p.stages & (HOPE_STAGE_REWARD | HOPE_STAGE_RS)) {
        if (arrive) status = HOPE_STATUS_ARRIVED;
        else {
            const double xmin = sc[SC_BBOX], xmax = sc[SC_BBOX + 1], ymin = sc[SC_BBOX + 2], ymax = sc[SC_BBOX + 3];
            bool coll = known_free ? false : detect_collision(x, y, ct, sn, tile, nlist, n_near, xl, lane, TIMING ? &cen_und : nullptr);
            if (coll) status = HOPE_STATUS_COLLIDED;
            else if (x > xmax || x < xmin || y > ymax || y < ymin) status = HOPE_STATUS_OUTBOUND;
            else {
                bool arrived = false;
                if (have_ua) arrived = ua / dest_area > 0.95;
                else if (kf_pose >= 0 ? ((apmask >> kf_pose) & 1) != 0 : arrival_possible(x, y, ct, sn, dcx, dcy, dcd, dsd)) {
                    ua = overlap_area(x, y, ct, sn, dbox, scr + LDS_SH, lane);
                    CEN_ARR(ua);
                    have_ua = true;
                    arrived = ua / dest_area > 0.95;
                }
                if (arrived) status = HOPE_STATUS_ARRIVED;
                else if (t > TOLERANT_TIME) status = HOPE_STATUS_OUTTIME;
            }
        }
    }

    // ---- reward (_get_reward :186-227, reward_shaping env_wrapper.py:10-35) ------------------------
    // Nothing of it stays here unless the arrival test already needed the overlap area: the polygon clip of the final
    // pose against the dest box and the arithmetic of the reward and of the target representation (tanh, acos(cos .),
    // atan2, sincos, square roots, divisions: hundreds of wave instructions that are the same on every lane) run in
    // k_post with one LANE per scene.
    const bool need_ua = (p.stages & HOPE_STAGE_REWARD) && status == HOPE_STATUS_CONTINUE && !have_ua;   // k_post clips (lane per scene)
    const double fin_x = x, fin_y = y, fin_h = h, fin_ua = ua;
    const int fin_t = t;

    // ---- fused episode turnover (HOPE_AUTO_RESET): CarParking.reset on the same map + its action-less step -------
    const bool turnover = (p.stages & HOPE_AUTO_RESET) && (p.stages & HOPE_STAGE_REWARD) && status != HOPE_STATUS_CONTINUE;
    if (turnover) {
        // HOPE_AUTO_REDRAW: a NEW map first (map.reset, car_parking_base.py:134): a pool entry of this scene's tile class,
        // picked exactly as hope_env_redraw picks it, copied into the scene's slots by this wave; the rest of the turnover
        // (and of this kernel, in the one-launch form) then works on the new scene, whose whole tile is staged in LDS
        bool redrawn = false;
        if (p.stages & HOPE_AUTO_REDRAW) {
            const StepCold* cp = p.cold;                               // (loaded here, by the few waves that turn over)
            const int cls = cp->slot_cls[scene] ? 1 : 0;               // the slot's class, not the current map's size
            const int cnt = cp->pool_cls_n[cls];
            if (cnt > 0 && (cp->pool_verts || cp->dlp.n_cases > 0)) {
                const uint32_t ep = cp->episode[scene];
                const uint64_t key = mix64(cp->redraw_seed ^ mix64(((uint64_t)scene << 32) | ep));
                const int j = cp->pool_cls[cls][(int)(key % (uint64_t)cnt)];
                double2* gdst = (double2*)(const_cast<double*>(p.verts) + (size_t)scene * p.max_obst * 8);
                float4* gobb = const_cast<float4*>(p.obb) + (size_t)scene * p.max_obst;
                double* gsc = const_cast<double*>(p.scene_c) + (size_t)scene * SC_WORDS;
                int nob;
                ssync();
                if (j >= 0) {                                           // a complete scene of the pool
                    nob = cp->pool_nobst[j];
                    const double* pv = cp->pool_verts + (size_t)j * p.max_obst * 8;
                    const double2* psrc = (const double2*)pv;
                    double2* ldst = (double2*)tile;
                    for (int v = lane; v < 4 * nob; v += WAVE) { const double2 q2 = psrc[v]; gdst[v] = q2; ldst[v] = q2; }
                    const double* pc = cp->pool_c + (size_t)j * SC_WORDS;
                    {
                        const double fox = pc[SC_BBOX], foy = pc[SC_BBOX + 2];
                        float4* gfv = cp->fverts + (size_t)scene * p.max_obst * 2;
                        float4* gfb = cp->fbox + (size_t)scene * p.max_obst;
                        uint8_t* gfl = cp->eflag + (size_t)scene * eflag_stride(p.max_obst);
                        for (int o = lane; o < nob; o += WAVE) {
                            gobb[o] = obstacle_box(pv + (size_t)o * 8);
                            obstacle_f32(pv + (size_t)o * 8, fox, foy, gfv + 2 * o, gfb + o, gfl + o);
                        }
                    }
                    if (lane < SC_WORDS) gsc[lane] = pc[lane];
                    sc = pc;                                            // the new scene's constants, straight from the pool
                } else {                                                // a Dragon-Lake-Parking case: drawn here (ParkingMapDLP.reset)
                    double* c24 = scr + LDS_SH + 64;                    // (region A is free between the sub-step loop and the lidar)
                    nob = draw_dlp_case(cp->dlp, -2 - j, mix64(key ^ 0xD1B54A32D192ED03ull), p.max_obst, (double*)gdst, gobb,
                                        cp->fverts + (size_t)scene * p.max_obst * 2, cp->fbox + (size_t)scene * p.max_obst,
                                        cp->eflag + (size_t)scene * eflag_stride(p.max_obst), c24, tile, cp->pool_overflow, lane);
                    ssync();
                    if (lane < SC_WORDS) gsc[lane] = c24[lane];
                    sc = c24;
                }
                if (lane == 0) {
                    const_cast<int32_t*>(p.n_obst)[scene] = nob; cp->cur_pool[scene] = j; cp->episode[scene] = ep + 1;
                    if (cp->layer_valid) cp->layer_valid[scene] = 0;
                }
                if (lane < 8) dbox[lane] = sc[SC_DBOX + lane];
                dest_area = sc[SC_DAREA];
                n_obst = nob;
                redrawn = true;
                // (one-launch form: the staged shape flags belong to the old map; without them the new episode's first lidar scan
                // simply runs without the back-face cull)
                if (PART == 0) for (int i_ = lane; 4 * i_ < nob; i_ += WAVE) ((uint32_t*)cfl)[i_] = 0;
                ssync();
            }
        }
        x = sc[SC_START]; y = sc[SC_START + 1]; h = sc[SC_START + 2];
        accum = 0.0;
        t = 1;                                                          // reset: t = 0, then step() -> t = 1
        hm_sincos(h, &sn, &ct);
        // the action-less step's status decides whether _get_reward runs (it only touches accum_arrive_reward)
        ssync();
        const int n_near0 = (PART == 0 || redrawn) ? build_near_list(tile, n_obst, x, y, 3.9, nlist, lane)
                                                   : stage_near(obb_s, src, n_obst, x - 3.9, x + 3.9, y - 3.9, y + 3.9, tile, nlist, lane);
        ssync();
        const double xmin = sc[SC_BBOX], xmax = sc[SC_BBOX + 1], ymin = sc[SC_BBOX + 2], ymax = sc[SC_BBOX + 3];
        bool cont = !detect_collision(x, y, ct, sn, tile, nlist, n_near0, xl, lane, TIMING ? &cen_und : nullptr) && !(x > xmax || x < xmin || y > ymax || y < ymin);
        if (cont) {
            const double ua0 = overlap_area(x, y, ct, sn, dbox, scr + LDS_SH, lane);
            CEN_ARR(ua0);
            if (!(ua0 / dest_area > 0.95)) {                            // not ARRIVED (and t = 1 is not OUTTIME): CONTINUE
                const double bur = ua0 / (2 * dest_area - ua0);
                if (!(bur < accum)) accum = bur;                        // :221-226 with accum = 0
            }
        }
    }

    // ---- write state + scalar outputs ---------------------------------------------------------------
    if (lane == 0) {
        st[0] = x; st[1] = y; st[2] = h; st[3] = accum;
        if (PART == 1) { p.cs[2 * (size_t)scene] = ct; p.cs[2 * (size_t)scene + 1] = sn; }
        p.tstep[scene] = t;
        if (p.hflags & STEP_HF_TRAJ) {
            // vehicle.trajectory: of the sub-step states only the last kept one stays (car_parking_base.py:259-276,
            // vehicle.py:144,158); a step blocked at its first sub-step adds nothing; reset leaves [start]
            const StepCold* cp = p.cold;
            double* tr = cp->traj + (size_t)scene * 60;
            int32_t* tlen = cp->traj_len;
            int tl = turnover ? 0 : tlen[scene];
            if (turnover) cp->traj_valid[scene] = 0;
            if (turnover || moved) {
                double* e = tr + 3 * (tl % 20);
                e[0] = x; e[1] = y; e[2] = h;
                tlen[scene] = tl + 1;
            }
        }
        {   // hand-over to k_post, which also writes the per-scene scalar outputs (pose, status, done, RS gate flag)
            double* pr = p.post + (size_t)scene * POST_WORDS;
            pr[0] = prev_x; pr[1] = prev_y; pr[2] = prev_h; pr[3] = fin_x; pr[4] = fin_y; pr[5] = fin_h; pr[6] = fin_ua;
            const int fl = ((p.stages & HOPE_STAGE_REWARD) ? POST_F_REWARD : 0) | (turnover ? POST_F_TURNOVER : 0) | (need_ua ? POST_F_NEED_UA : 0);
            pr[7] = __hiloint2double(fl, status | (fin_t << 8));
        }
    }

    }                            // PART != 2
    if (PART == 1 || !(p.stages & HOPE_STAGE_OBS)) { ST_T(2); ST_FLUSH(); return; }

    ST_T(2);
    // ---- lidar (lidar_simulator.py:31-135) -----------------------------------------------------------
    // world -> ego in place: affine [a, b, -b, a, x_off, y_off] (:58-64)
    // Only obstacles whose box comes within lidar_range of the sensor can pass the ring test of :69 (the ring lies inside
    // its box; 1e-6 m of slack against the rounding of the ego transform), so only those are transformed and measured:
    // a lot of 100 obstacles has a dozen within 10 m.  llist: their indices, then (compacted in place) the kept rings'.
    int* llist = keep;
    ssync();                                                  // the near list (same words) is dead
    const double lr = LIDAR_RANGE + 1e-6;
    const uint8_t* eflag_s = p.eflag + (size_t)scene * eflag_stride(p.max_obst);
    const int n_l = PART == 2 ? stage_near(obb_s, src, n_obst, x - lr, x + lr, y - lr, y + lr, tile, llist, lane, eflag_s, cfl, &obb_pre)
                              : build_near_list(tile, n_obst, x, y, lr, llist, lane);
    ssync();
    {
        const double a = ct, b = sn;
        const double x_off = -x * a - y * b;
        const double y_off = x * b - y * a;
        for (int i = lane; i < 4 * n_l; i += WAVE) {
            const int v = 4 * llist[i >> 2] + (i & 3);
            double px = tile[2 * v], py = tile[2 * v + 1];
            tile[2 * v] = a * px + b * py + x_off;
            tile[2 * v + 1] = (-b) * px + a * py + y_off;
        }
    }
    ssync();
    // ring kept iff distance(ring, origin) < lidar_range (:69); 4 consecutive lanes = one ring
    int n_k = 0;
    for (int base = 0; base < 4 * n_l; base += WAVE) {
        const int i = base + lane;
        const bool in = i < 4 * n_l;
        const int o = in ? llist[i >> 2] : 0;
        // Almost every ring is decided WITHOUT GEOS's point-to-segment arithmetic (two divisions and three square roots per edge, ~90
        // vector instructions per 64 edges; the step is VALU-issue bound) from squared quantities with 1e-9 of relative margin (the
        // distance's own rounding is ~1e-15): an edge is certainly nearer than the range when one of its end points is (the
        // segment's distance is at most the vertex's), certainly farther when even its LINE is, or when the foot of the perpendicular
        // lies clearly beyond an end point that is farther; a foot clearly inside the segment makes the line's distance the
        // segment's.  A ring is kept when one edge is certainly nearer, dropped when all four are certainly farther; anything else
        // -- a value within the margin of the range or of a case boundary -- takes the exact path below (the tie census: a handful of
        // rings in 1.6e9 come that near; the exact path is what the instrumented build always runs).
        double dd = INFINITY;
        bool amb = in;
        if (!TIMING) {
            bool keep_c = false, drop_c = false;
            if (in) {
                const int e = 4 * o + (i & 3), e2 = 4 * o + ((i + 1) & 3);
                const double ax = tile[2 * e], ay = tile[2 * e + 1], bx = tile[2 * e2], by = tile[2 * e2 + 1];
                const double ddx = bx - ax, ddy = by - ay;
                const double len2 = ddx * ddx + ddy * ddy;
                const double dot = (0.0 - ax) * ddx + (0.0 - ay) * ddy;                       // r = dot / len2 (Distance::pointToSegment)
                const double qa = ax * ax + ay * ay, qb = bx * bx + by * by;
                const double num = ay * ddx - ax * ddy;                                       // s = num / len2, distance |s| sqrt(len2)
                const double R2 = LIDAR_RANGE * LIDAR_RANGE, ETA = 1e-9;
                const double lo = R2 * (1.0 - ETA), hi = R2 * (1.0 + ETA);
                const double n2 = num * num;
                const bool line_far = n2 > hi * len2, line_near = n2 < lo * len2;
                const bool foot_in = dot > ETA * len2 && dot < (1.0 - ETA) * len2;             // 0 < r < 1, clearly
                const bool foot_a = dot < -ETA * len2, foot_b = dot > (1.0 + ETA) * len2;      // r < 0 / r > 1, clearly
                keep_c = qa < lo || qb < lo || (foot_in && line_near);
                // (a triangle repeats its last vertex: that edge is the point itself)
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
        if (TIMING && in && (i & 3) == 0) { cen_ring_n += 1; cen_ring = fmin(cen_ring, fabs(dd - LIDAR_RANGE)); }
        const bool kq = in && (i & 3) == 0 && dd < LIDAR_RANGE;
        const unsigned long long km = __ballot(kq);
        if (kq) llist[n_k + __popcll(km & ((1ull << lane) - 1))] = o;   // in place: writes stay below the next chunk's reads
        n_k += __popcll(km);
    }
    ssync();
    const int n_kslots = 4 * n_k;
    ST_T(3);
    // beams: lane l owns beams l and l+64.  Two passes (SIMT pays for the union of lanes, and every edge is
    // crossed by SOME beam, so the two float64 divisions of a pair must not sit in the lane-per-beam loop):
    //  pass 1 (cheap): per kept edge each lane tests its beams with exact-safe necessary conditions and
    //          appends the surviving (beam, edge) pairs to an LDS queue;
    //  pass 2 (dense): the queue is drained 64 pairs at a time through the full reference arithmetic and an
    //          LDS atomic-min per beam (non-negative doubles order like their bit patterns).
    const int i0 = lane, i1 = lane + 64;
    const bool has1 = i1 < NBEAM;
    const double a0 = PART == 2 ? bm0 : p.beam_ab[2 * i0], b0 = PART == 2 ? bm1 : p.beam_ab[2 * i0 + 1];
    const double a1 = PART == 2 ? bm2 : (has1 ? p.beam_ab[2 * i1] : 0.0), b1 = PART == 2 ? bm3 : (has1 ? p.beam_ab[2 * i1 + 1] : 0.0);
    unsigned long long* best = (unsigned long long*)(scr + LDS_TX);          // [128] (tx/ty are dead by now)
    int* queue = (int*)(scr + LDS_TX + 128);                                 // [LQ]
    constexpr int LQ = 384;
    best[lane] = 0x7ff0000000000000ull;                                      // +inf
    best[lane + 64] = 0x7ff0000000000000ull;
    ssync();
    int qn = 0;
    auto drain = [&]() {
        const unsigned long long td_ = TIMING ? __builtin_readcyclecounter() : 0;
        ssync();
        for (int q0 = 0; q0 < qn; q0 += WAVE) {
            int q = q0 + lane;
            if (q < qn) {
                int pr = queue[q];
                int bi = pr & 127, e = pr >> 7;
                int e2 = (e & ~3) | ((e + 1) & 3);
                double x1 = tile[2 * e], y1 = tile[2 * e + 1], x2 = tile[2 * e2], y2 = tile[2 * e2 + 1];
                double d = y2 - y1, ee = x1 - x2, f = y1 * x2 - x1 * y2;
                const double ba = p.beam_ab[2 * bi], bb = p.beam_ab[2 * bi + 1];
                double r = INFINITY;
                if (beam_may_hit(ba, bb, x1, y1, x2, y2)) r = beam_edge(bi, ba, bb, x1, y1, x2, y2, d, ee, f);
                if (r < INFINITY) atomicMin(&best[bi], (unsigned long long)__double_as_longlong(r));
            }
        }
        ssync();
        qn = 0;
        if (TIMING) { const unsigned long long dt_ = __builtin_readcyclecounter() - td_; tsec[6] += dt_; t0_ += dt_; }
    };
    // pass 1, one lane per edge slot: the beams that can see an edge are those inside the angle it subtends at
    // the sensor (a segment not through the origin subtends < pi).  The range is taken from float32 atan2 with
    // a 2e-3 rad margin on both sides (beam pitch 5.2e-2 rad), so it is a superset of the reference's hits;
    // pass 2 applies the exact tests.  An edge whose span is ill-defined (passes within ~1e-2 rad of the origin
    // direction flip) is paired with all 120 beams.
    {
        const float PITCH = 6.283185307179586f / NBEAM, MARGIN = 2e-3f;
        for (int base = 0; base < n_kslots && !(p.stages & 0x1000); base += WAVE) {   // 0x1000: profiling switch
            const int i = base + lane;
            const int e = i < n_kslots ? 4 * llist[i >> 2] + (i & 3) : 0;
            int lo = 0, cnt = 0;
            bool front = false, back = false, risky = true;     // back-face cull (below): this edge's facing; "the ring cannot be culled"
            if (i < n_kslots) {
                const int e2 = (e & ~3) | ((e + 1) & 3);
                const double dx1 = tile[2 * e], dy1 = tile[2 * e + 1], dx2 = tile[2 * e2], dy2 = tile[2 * e2 + 1];
                const float x1 = (float)dx1, y1 = (float)dy1;
                const float x2 = (float)dx2, y2 = (float)dy2;
                // (t2 is the NEXT vertex's t1: the same atan2f on the same floats, computed by the next lane of the ring's quad -- a
                // quad rotation instead of a second ~30-instruction evaluation; the step is VALU-issue bound)
                const float t1 = span_angle(y1, x1);
                const float t2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t1), 0x39, 0xf, 0xf, true));   // quad_perm [1,2,3,0]
                float dth = t2 - t1;
                if (dth > 3.14159265f) dth -= 6.28318531f;
                if (dth <= -3.14159265f) dth += 6.28318531f;
                const float span = fabsf(dth);
                // all beams: span ill-defined, or an end point so close to the sensor (< 0.1 m: far inside the hull) that the
                // float32 rounding of its coordinates could move its direction by more than the margin
                const bool wild = span > 3.13f || !(span == span) || fminf(x1 * x1 + y1 * y1, x2 * x2 + y2 * y2) < 0.01f;
                if (wild) { lo = 0; cnt = NBEAM; }
                else {
                    float ts = dth >= 0 ? t1 : t2;                  // start of the arc, counter-clockwise
                    if (ts < 0) ts += 6.28318531f;
                    // beams i with theta_i = i PITCH inside [ts - MARGIN, ts + span + MARGIN]; none when the edge is seen
                    // between two beams (float32 rounding of the quotients: ~1e-5 of a pitch, far inside the margin)
                    // (products with the rounded 1 / PITCH instead of quotients: one more ulp, 1e-5 of a pitch again)
                    const int ilo = (int)ceilf((ts - MARGIN) * (1.0f / PITCH));
                    const int ihi = (int)floorf((ts + span + MARGIN) * (1.0f / PITCH));
                    cnt = ihi - ilo + 1;
                    if (cnt < 0) cnt = 0;
                    if (cnt > NBEAM) cnt = NBEAM;
                    lo = ilo < 0 ? ilo + NBEAM : ilo >= NBEAM ? ilo - NBEAM : ilo;     // ts in [0, 2 pi]: ilo in [-1, NBEAM]
                    if ((unsigned)lo >= (unsigned)NBEAM) { lo = 0; cnt = NBEAM; }       // (cannot happen; all beams if it does)
                }
                // ---- back-face cull.  A beam that crosses a BACK edge of a convex ring seen from outside has entered the ring
                // through a front edge first, nearer to the sensor: if that nearer hit is certain to pass the reference's tests
                // (lidar_simulator.py:116-129), the back edge can never be the beam's minimum (:131) and its pairs are dropped.
                // Certain means: the ring is a convex quadrilateral without slivers (shape flag, made with the tile), the sensor is
                // clearly outside (every edge decisively front or back, both kinds present), NO vertex of the ring lies within
                // 2e-4 rad of a beam direction (so the entry point is >= 2e-5 m inside a front edge -- its coordinate-box tests hold
                // with 1000 x the rounding error -- and the ring is >= 1e-5 m thick along the beam), and no front edge is
                // within 1e-4 m of axis-parallel (a degenerate coordinate box is met by bit-equality only).  Anything else: the
                // ring keeps all its pairs, as before.
                const int fl = (int)cfl[e >> 2];
                const double cr = dx1 * dy2 - dx2 * dy1;            // > 0: the sensor is on the left of the directed edge
                const double sc2 = (dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2);
                const bool decisive = cr * cr > 1e-12 * sc2;       // |sin(angle subtended)| > 1e-6
                const bool left = cr > 0;
                front = decisive && (left != ((fl & OBST_F_CCW) != 0));          // CCW ring: the inside is on the left
                back = decisive && !front;
                const float q1 = t1 * (1.0f / PITCH);
                const bool near_beam = fabsf(q1 - rintf(q1)) * PITCH <= 2.0e-4f + 1e-5f;     // this edge's first vertex vs the beam directions
                const bool thin = front && (fabs(dx2 - dx1) < 1e-4 || fabs(dy2 - dy1) < 1e-4);
                risky = wild || !decisive || near_beam || thin || !(fl & OBST_F_CONVEX);
            }
            {   // ring = 4 consecutive lanes: cull the back edges iff nothing about the ring is risky and it has both kinds of edges
                const unsigned long long rb = __ballot(risky), fb = __ballot(front), bb = __ballot(back);
                const int sh = lane & ~3;
                const bool ring_ok = ((rb >> sh) & 0xF) == 0 && ((fb >> sh) & 0xF) != 0 && ((bb >> sh) & 0xF) != 0;
                if (ring_ok && back && !(p.stages & 0x4000)) cnt = 0;          // (0x4000: profiling / A-B switch, no cull)
            }
            // Append the pairs to the queue EDGE BY EDGE: for every edge with beams (a scalar walk over the ballot) its range
            // [lo, lo + cnt) is written by the lanes 0 .. cnt-1 in one store -- no per-beam ballots / prefix counts at all
            // (a lane-per-edge loop over the beams cost ~14 VALU instructions per beam of the widest edge; a four-beams-per-
            // round variant with bit-plane prefix sums ~8).
            // Edges that subtend at most NARROW beams are appended all at once, one lane per edge: queue offsets from a
            // wave prefix sum, then NARROW predicated stores (6 x 64 = the queue's capacity).
            ST_T(4);
            constexpr int NARROW = 6;
            const int cs = (cnt > 0 && cnt <= NARROW) ? cnt : 0;
            const int incl = wave_incl_scan_i(cs, lane);
            const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
            if (total > 0) {
                if (qn + total > LQ) drain();
                const int off = qn + incl - cs, tag = e << 7;
#pragma unroll
                for (int k = 0; k < NARROW; k++) {
                    if (k < cs) {
                        int bi = lo + k;
                        if (bi >= NBEAM) bi -= NBEAM;
                        queue[off + k] = tag | bi;
                    }
                }
                qn += total;
            }
            unsigned long long todo = __ballot(cnt > NARROW);
            while (todo) {
                const int el = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int lo_e = __builtin_amdgcn_readlane(lo, el), cnt_e = __builtin_amdgcn_readlane(cnt, el);
                if (qn + cnt_e > LQ) drain();
                const int tag = __builtin_amdgcn_readlane(e, el) << 7;
                if (lane < cnt_e) {
                    int bi = lo_e + lane;
                    if (bi >= NBEAM) bi -= NBEAM;
                    queue[qn + lane] = tag | bi;
                }
                if (cnt_e > WAVE && lane + WAVE < cnt_e) {        // an edge that spans more than 64 beams
                    int bi = lo_e + lane + WAVE;
                    if (bi >= NBEAM) bi -= NBEAM;
                    queue[qn + lane + WAVE] = tag | bi;
                }
                qn += cnt_e;
            }
            ST_T(5);
        }
    }
    // for the mask stage, in flight during the drain: the table's maximum at the lane's beams, or (HOPE_MASK_LUT) their bins per metre
    constexpr bool MLUT = HOPE_MASK_LUT && !TIMING;             // (the instrumented build keeps the row probes: its census counts their compares)
    const double pm0 = MLUT ? p.mask_bsc[i0] : p.pmax[UPS * i0], pm1 = has1 ? (MLUT ? p.mask_bsc[i1] : p.pmax[UPS * i1]) : 0.0;
    if (qn > 0) drain();
    const double best0 = sqrt(__longlong_as_double((long long)best[i0]));                     // min of the roots = root of the min
    const double best1 = has1 ? sqrt(__longlong_as_double((long long)best[i1])) : INFINITY;
    const double base0 = p.hull_base[i0], base1 = has1 ? p.hull_base[i1] : 0.0;
    const double lid0 = clipd(best0, 0, LIDAR_RANGE) - base0;      // get_observation :46
    const double lid1 = clipd(best1, 0, LIDAR_RANGE) - base1;
    if (p.lidar) {
        OT* lo = (OT*)p.lidar + (size_t)NBEAM * scene;
        lo[i0] = (OT)lid0;
        if (has1) lo[i1] = (OT)lid1;
    }
    ST_T(5);
    if (!p.action_mask || (p.stages & 0x2000)) { ST_FLUSH(); return; }    // 0x2000: internal profiling switch

    // ---- action mask (action_mask.py:166-196) ----------------------------------------------------------
    ssync();                                                      // region A: best[]/queue[] are dead from here
    double* xs = scr + LDS_X;                                     // lidar_obs = clip(raw,0,10) + base (:170)
    xs[i0] = clipd(lid0, 0, 10) + base0;
    if (has1) xs[i1] = clipd(lid1, 0, 10) + base1;
    ssync();
    if (lane == 0) xs[NBEAM] = xs[0];                             // circular (:158)
    ssync();
    // step_len[a] = min over the 1200 upsampled beams l of cnt(l,a) = #{k : tab[l][k][a] <= d_l} (tab is prefix-maxed
    // over k, so the count IS the first-exceed index of :176-177).
    //
    // Coarse decision.  Fine beam l = 10i + j carries d_l = x_i*w1 + x_{i+1}*w2 and the fine table is the SAME
    // interpolation of the coarse table (:142,:161-162), hence for every l of interval i
    //     tab[l][k][a] <= w1*tab[10i][k][a] + w2*tab[10i+10][k][a]   (up to ~1e-15 rounding),
    // and at the coarse beams themselves (w1 = 1, w2 = 0: exact) cnt(10i,a) = #{k : tab[10i][k][a] <= x_i}.
    // With  m    = min_i #{k : tab[10i][k][a] <= x_i}          (exact counts at the 120 coarse beams: upper bound)
    //       mlow = min_i #{k : tab[10i][k][a] <= x_i - 1e-9}    (=> every fine beam has cnt >= mlow: lower bound)
    // the answer lies in [mlow, m].  They can differ only if some table entry is within 1e-9 of x_i (e.g. the
    // structural tie of the straight arcs when an obstacle touches the hull side); then, and only then, the
    // full 1200-beam evaluation below runs.  120 row probes instead of up to 1200 x 10 rows.
    // `tie` replaces a second set of probes for mlow: the table is monotone in k, so if the largest entry that is
    // <= x_i (the one at the lane's count boundary, which the probe / walk has already loaded) is also <= x_i - 1e-9,
    // every lower entry is too and this row cannot push the lower bound below the exact count.  Any boundary value
    // inside (x_i - 1e-9, x_i] raises `tie` -> exact evaluation.
    int mstep = NITER;
    bool tie = false, mask_trivial = false;
    if (MLUT) {
        // ---- round 6: the count-interval table (hope_env_upload_tables; hope_obs_pair.h has the argument).  Lane = action: its byte of
        // the beam's row holds cnt_lo | cnt_hi << 4 for all ten rows k; one independent 2-byte load per active beam instead of the
        // dependent row probes; the float64 entries only where cnt_lo < cnt_hi can still lower the minimum.
        const double q0 = (xs[i0] - (base0 - 1e-6)) * pm0, q1 = has1 ? (xs[i1] - (base1 - 1e-6)) * pm1 : (double)MASK_LUT_NB;
        const bool c0 = q0 < (double)MASK_LUT_NB, c1 = has1 && q1 < (double)MASK_LUT_NB;
        const int row0 = i0 * MASK_LUT_NB + (c0 ? (int)q0 : 0), row1 = i1 * MASK_LUT_NB + (c1 ? (int)q1 : 0);
        const unsigned long long am[2] = {__ballot(c0), __ballot(c1)};
        mask_trivial = !(am[0] | am[1]);                     // no active beam (two thirds of the scene-steps): every count is NITER
        if (am[0] | am[1]) {
            constexpr int NONE = NBEAM * MASK_LUT_NB;                     // the table's last row: [10, 10]
            const int hl_ = lane < NACT / 2 ? lane : (lane < NACT ? lane - NACT / 2 : 31), sh_ = (lane >= NACT / 2 && lane < NACT) ? 8 : 0;
            const char* lutb = (const char*)p.mask_lut + 2 * hl_;
            unsigned mlo = NITER, mhi = NITER;
            constexpr int PG = 8;                                         // beams whose loads are in flight together
            unsigned long long m0 = am[0], m1 = am[1];
            auto pop_row = [&]() -> int {                                // the next active beam's table row (beams 0..63, then 64..119)
                int row = NONE;
                if (m0) { const int l_ = __ffsll((long long)m0) - 1; m0 &= m0 - 1; row = __builtin_amdgcn_readlane(row0, l_); }
                else if (m1) { const int l_ = __ffsll((long long)m1) - 1; m1 &= m1 - 1; row = __builtin_amdgcn_readlane(row1, l_); }
                return row;
            };
            // the first PG beams (all of them for nine waves out of ten): their table words stay in registers, two per register, for
            // the second visit
            unsigned pk[PG / 2];
            {
                unsigned v0[PG];
#pragma unroll
                for (int g = 0; g < PG; g++) v0[g] = (unsigned)*(const uint16_t*)(lutb + (unsigned)pop_row() * (MASK_LUT_ROW * 2u)) >> sh_ & 0xFFu;
#pragma unroll
                for (int g = 0; g < PG; g++) { mlo = min(mlo, v0[g] & 15u); mhi = min(mhi, v0[g] >> 4); }
#pragma unroll
                for (int g = 0; g < PG / 2; g++) pk[g] = v0[2 * g] | v0[2 * g + 1] << 16;
            }
            while (m0 | m1) {
                unsigned v[PG];
#pragma unroll
                for (int g = 0; g < PG; g++) v[g] = (unsigned)*(const uint16_t*)(lutb + (unsigned)pop_row() * (MASK_LUT_ROW * 2u)) >> sh_ & 0xFFu;
#pragma unroll
                for (int g = 0; g < PG; g++) { mlo = min(mlo, v[g] & 15u); mhi = min(mhi, v[g] >> 4); }
            }
            if (__any(mlo < mhi)) {
                const char* tabb = (const char*)p.tab;
                int k = 0;                                               // position of the beam in the order of the first visit
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    unsigned long long m = am[half];
                    while (m) {
                        const int l_ = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int ib = 64 * half + l_;
                        unsigned w;
                        switch (k >> 1) {                                // (wave-uniform)
                            case 0: w = pk[0]; break; case 1: w = pk[1]; break; case 2: w = pk[2]; break; case 3: w = pk[3]; break;
                            default: {
                                const int row = __builtin_amdgcn_readlane(half ? row1 : row0, l_);
                                w = ((unsigned)*(const uint16_t*)(lutb + (unsigned)row * (MASK_LUT_ROW * 2u)) >> sh_ & 0xFFu) << ((k & 1) << 4);
                            }
                        }
                        w = (w >> ((k & 1) << 4)) & 0xFFu;
                        k++;
                        const unsigned lo = w & 15u, hi = w >> 4;
                        const bool need = lo < hi && lo < mhi;
                        if (!__any(need)) continue;
                        const double xv = xs[ib];
                        if (need) {
                            unsigned c = lo;
                            const unsigned lim = min(hi, mhi);
                            const unsigned rowo = (unsigned)(UPS * ib) * (NITER * NACT * 8u) + (unsigned)(lane < NACT ? lane : 0) * 8u;
                            while (c < lim) {
                                const double t = *(const double*)(tabb + (rowo + c * (NACT * 8u)));
                                if (t > xv) break;
                                if (t > xv - 1e-9) tie = true;
                                c++;
                            }
                            mhi = min(mhi, c);
                        }
                    }
                }
            }
            mstep = (int)mhi;
        }
    } else
    {
        // lane-parallel activity test (one lane per coarse beam), then only the active rows are visited,
        // four at a time so that their probes are in flight together
        const bool c0 = xs[i0] - 1e-9 < pm0;
        const bool c1 = has1 && xs[i1] - 1e-9 < pm1;
        unsigned long long am[2] = {__ballot(c0), __ballot(c1)};
#pragma unroll
        for (int half = 0; half < 2; half++) {
            unsigned long long m = am[half];
            while (m) {
                constexpr int MG = HOPE_MASK_MG;                              // rows probed together: their loads are in flight at once
                int ib[MG];
#pragma unroll
                for (int g = 0; g < MG; g++) {
                    if (m) { ib[g] = 64 * half + __ffsll((long long)m) - 1; m &= m - 1; }
                    else ib[g] = ib[0];                           // duplicate: min() is idempotent
                }
                if (lane < NACT) {
                    double v[MG];
                    const int mprobe = mstep;
#pragma unroll
                    for (int g = 0; g < MG; g++)
                        v[g] = mstep > 0 ? p.tab[(size_t)(UPS * ib[g]) * NITER * NACT + lane + (mstep - 1) * NACT] : 0.0;
#pragma unroll
                    for (int g = 0; g < MG; g++) {
                        if (mstep > 0) {
                            const double xv = xs[ib[g]];
                            double bv = v[g];                     // boundary value: largest examined entry <= x
                            if (TIMING) { cen_mask_n += 1; const double df_ = fabs(bv - xv); if (df_ > 0) cen_mask = fmin(cen_mask, df_); else cen_eq += 1; }
                            if (bv > xv) {
                                const double* row = p.tab + (size_t)(UPS * ib[g]) * NITER * NACT + lane;
                                // walk down to the first entry <= x: four rows per trip, loaded together (a dependent
                                // load per row made this stage a chain of L2 latencies); row mstep - 1 is already known
                                // to exceed when nothing lowered mstep since the probe
                                int c = mstep == mprobe ? mstep - 1 : mstep;
                                bv = -INFINITY;
                                while (c > 0) {
                                    const double b0 = row[(c - 1) * NACT];
                                    const double b1 = c > 1 ? row[(c - 2) * NACT] : -INFINITY;
                                    const double b2 = c > 2 ? row[(c - 3) * NACT] : -INFINITY;
                                    const double b3 = c > 3 ? row[(c - 4) * NACT] : -INFINITY;
                                    bv = b0; if (!(bv > xv)) break;
                                    if (--c == 0) break;
                                    bv = b1; if (!(bv > xv)) break;
                                    if (--c == 0) break;
                                    bv = b2; if (!(bv > xv)) break;
                                    if (--c == 0) break;
                                    bv = b3; if (!(bv > xv)) break;
                                    --c;
                                }
                                if (c == 0) bv = -INFINITY;
                                mstep = c;
                                if (TIMING && c > 0) { cen_mask_n += 1; const double df_ = fabs(bv - xv); if (df_ > 0) cen_mask = fmin(cen_mask, df_); else cen_eq += 1; }
                            }
                            if (bv > xv - 1e-9) tie = true;
                        }
                    }
                }
            }
        }
    }
    if (TIMING && __any(tie)) cen_tie = 1;
    if (__any(tie)) {
        // ---- exact fall-back over all 1200 beams (rare) ---------------------------------------------------
        for (int r = 0; r < (NL + WAVE - 1) / WAVE; r++) {
            int l = r * WAVE + lane;
            double dl = 0;
            bool act = false;
            if (l < NL) {
                int i = l / UPS, j = l % UPS;
                double w2 = (double)j / UPS, w1 = 1 - w2;      // (j % 10) / 10 of _linear_interpolate
                dl = xs[i] * w1 + xs[i + 1] * w2;                 // _linear_interpolate (:161-162)
                act = dl < p.pmax[l];
            }
            unsigned long long m = __ballot(act);
            while (m) {
                int bpos = __ffsll((long long)m) - 1;
                m &= m - 1;
                double d_ll = __shfl(dl, bpos);
                if (lane < NACT) {
                    const double* row = p.tab + (size_t)(r * WAVE + bpos) * NITER * NACT + lane;
                    int c = mstep;
                    while (c > 0 && row[(c - 1) * NACT] > d_ll) c--;
                    mstep = c;
                }
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
    if (mask_trivial) mn = (li <= 2 || li >= half - 3) ? NITER - 1 : NITER;   // the filter's result for all-NITER counts: the ends' decrement, spread by the 5-wide minimum
    else
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
    double mo = HOPE_MASK_FRAC_TABLE ? MASK_STEP_FRACTION[mn] : mask_fraction(mn);   // mn / n_iter (a float64 division per scene otherwise)
    unsigned long long nz = __ballot(lane < NACT && mn > 0);
    if (nz == 0) mo = clipd(mo, 0.01, 1);                          // all-zero -> 0.01 (:182-183)
    if (lane < NACT) ((OT*)p.action_mask)[(size_t)NACT * scene + lane] = (OT)mo;
    ST_T(7);
    ST_FLUSH();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_post: reward (_get_reward car_parking_base.py:186-227, reward_shaping env_wrapper.py:10-35) and target representation
// (_get_targt_repr :372-381; the 5th entry is cos again) -- ONE LANE PER SCENE.  Everything here is scalar arithmetic per
// scene; inside k_env_step (one wave per scene) it cost the whole wave ~500 instructions.  Same expressions, same order.
// ---------------------------------------------------------------------------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(64) void k_post(int n_list, const int32_t* scene_list, const uint8_t* active, uint32_t stages,
                                            const double* scene_c, double* state, const double* post, uint8_t* rs_flag,
                                            hope_step_out out) {
    __shared__ double clip_lds[32 * CLIP_COLS];               // polygon buffers of the clip: CLIP_COLS lanes at a time
    const int idx = blockIdx.x * WAVE + threadIdx.x;
    const int scene = idx < n_list ? scene_list[idx] : 0;
    const bool live = idx < n_list && !(active && !active[scene]);
    if (!__any(live)) return;
    const double* sc = scene_c + (size_t)scene * SC_WORDS;
    const double* pr = post + (size_t)scene * POST_WORDS;
    double* st = state + (size_t)scene * ST_WORDS;
    const double destx = sc[SC_DEST], desty = sc[SC_DEST + 1], desth = sc[SC_DEST + 2];
    const int packed = __double2loint(pr[7]), fl = __double2hiint(pr[7]);
    const int status = packed & 0xff, t = packed >> 8;
    // ---- per-scene scalar outputs ------------------------------------------------------------------------
    if (live) {
        if (out.pose) { out.pose[3 * (size_t)scene] = st[0]; out.pose[3 * (size_t)scene + 1] = st[1]; out.pose[3 * (size_t)scene + 2] = st[2]; }
        if (stages & HOPE_STAGE_REWARD) {
            if (out.status) out.status[scene] = status;
            if (out.done) out.done[scene] = status != HOPE_STATUS_CONTINUE;
        }
        if (!(stages & HOPE_STAGE_RS)) {                     // (with the Reeds-Shepp stage k_rs_compact clears them, ahead of the search)
            if (out.rs_word) {
                const unsigned long long none = (unsigned char)HOPE_RS_NONE;
                *(unsigned long long*)(out.rs_word + 8 * (size_t)scene) = none | none << 8 | none << 16 | none << 24 | none << 32;
            }
            if (out.rs_lengths) {
#pragma unroll
                for (int i = 0; i < 5; i++) ((OT*)out.rs_lengths)[5 * (size_t)scene + i] = (OT)0;
            }
        }
    }
    // ---- overlap area of the final pose with the dest box, for the scenes whose step did not need it already (most CONTINUE
    // scenes): exact quick reject first (as overlap_area: disjoint discs -> empty intersection), then the polygon clip for the few
    // lanes within reach of their slot, CLIP_COLS of them at a time (wave-level loop: every lane of the wave passes here)
    double ua_ = live ? pr[6] : 0.0;
    {
        const bool want = live && (fl & POST_F_REWARD) && (stages & HOPE_STAGE_REWARD) && status == HOPE_STATUS_CONTINUE && (fl & POST_F_NEED_UA);
        bool clip = false;
        Box box;
        if (want) {
            double sn, ct;
            hm_sincos(pr[5], &sn, &ct);
            box = make_box(pr[3], pr[4], ct, sn);
            const double* dbox = sc + SC_DBOX;
            const double cx = 0.5 * (box.x[0] + box.x[2]), cy = 0.5 * (box.y[0] + box.y[2]);
            const double dx = 0.5 * (dbox[0] + dbox[4]) - cx, dy = 0.5 * (dbox[1] + dbox[5]) - cy;
            const double reach = 5.2;
            ua_ = 0.0;
            clip = !(dx * dx + dy * dy > reach * reach);
        }
        unsigned long long cm = __ballot(clip);
        while (cm) {
            // the next CLIP_COLS lanes that clip: column = rank of the lane among them
            const int rank = __popcll(cm & ((1ull << (threadIdx.x & 63)) - 1));
            const bool mine = clip && ((cm >> (threadIdx.x & 63)) & 1) && rank < CLIP_COLS;
            if (mine) {
                double* sh = clip_lds + rank;
#pragma unroll
                for (int i = 0; i < 4; i++) { sh[i * CLIP_COLS] = box.x[i]; sh[(8 + i) * CLIP_COLS] = box.y[i]; }
                ua_ = quad_intersection_area_private(sc + SC_DBOX, sh);
            }
            // drop the lanes just served
            unsigned long long served = 0;
            {
                unsigned long long t_ = cm;
                for (int k = 0; k < CLIP_COLS && t_; k++) { served |= t_ & (~t_ + 1); t_ &= t_ - 1; }
            }
            cm &= ~served;
            __syncthreads();
        }
    }
    if (!live) return;
    if ((fl & POST_F_REWARD) && (stages & HOPE_STAGE_REWARD)) {
        double ri0 = 0, ri2 = 0, ri3 = 0, ri4 = 0, reward = 0;
        if (status == HOPE_STATUS_CONTINUE) {
            const double dest_area = sc[SC_DAREA], dnorm = sc[SC_DNORM];
            double ua;
            ua = ua_;
            ri0 = -hm_tanh((double)t / (10 * TOLERANT_TIME));
            double dq[2], aq[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {                      // k = 0: current pose, 1: previous pose
                const double ddx = pr[k ? 0 : 3] - destx, ddy = pr[k ? 1 : 4] - desty;
                dq[k] = sqrt(ddx * ddx + ddy * ddy) / dnorm;                     // dist / max(|dest - start|, 10)
                double fold = hm_acos(hm_cos(pr[k ? 2 : 5] - desth));
                fold = fold < PI / 2 ? fold : PI - fold;
                aq[k] = fold / PI;
            }
            ri2 = dq[1] - dq[0];
            ri3 = aq[1] - aq[0];
            double accum = st[3];
            double bur = ua / (2 * dest_area - ua);
            if (bur < accum) bur = 0;
            else { double pa = accum; accum = bur; bur -= pa; }
            ri4 = bur;
            st[3] = accum;                                     // (a scene with this status was not turned over)
            double rw = 0;
            rw += 1 * ri0; rw += 0 * 0.0; rw += 5 * ri2; rw += 0 * ri3; rw += 10 * ri4;
            reward = rw;
        } else if (status == HOPE_STATUS_OUTBOUND) reward = -50;
        else if (status == HOPE_STATUS_OUTTIME) reward = -1;
        else if (status == HOPE_STATUS_ARRIVED) reward = 50;
        else if (status == HOPE_STATUS_COLLIDED) reward = -50;
        reward *= 0.1;
        if (out.reward) ((OT*)out.reward)[scene] = (OT)reward;
        if (out.reward_info) {
            OT* ri = (OT*)out.reward_info + 5 * (size_t)scene;
            ri[0] = (OT)ri0; ri[1] = (OT)0; ri[2] = (OT)ri2; ri[3] = (OT)ri3; ri[4] = (OT)ri4;
        }
    }
    if ((stages & HOPE_STAGE_OBS) && out.target) {            // the pose the observation belongs to (after a turnover: the start)
        const double x = st[0], y = st[1], h = st[2];
        const double rdx = destx - x, rdy = desty - y;
        const double rel_distance = sqrt(rdx * rdx + rdy * rdy);
        const double rel_angle = hm_atan2(rdy, rdx) - h;
        const double rel_dest_heading = desth - h;
        double sv, cv, sd, cd;
        hm_sincos(rel_angle, &sv, &cv);
        hm_sincos(rel_dest_heading, &sd, &cd);
        OT* tg = (OT*)out.target + 5 * (size_t)scene;
        tg[0] = (OT)rel_distance; tg[1] = (OT)cv; tg[2] = (OT)sv; tg[3] = (OT)cd; tg[4] = (OT)cd;
    }
}

}  // namespace hope
