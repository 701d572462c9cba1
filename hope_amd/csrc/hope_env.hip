// hope_env.hip -- host side of the C ABI (include/hope_env.h) + the fused step kernel instantiation.
// gfx950 only; no CPU fallback: every entry point fails with HOPE_ENODEV/HOPE_EHIP when no device works.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "hope_dev.h"
#include "hope_env.h"
#include "hope_internal.h"
#include "hope_step_kernel.h"
#include "hope_obs_pair.h"
#include "hope_motion_pair.h"

using namespace hope;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(HOPE_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

struct hope_env {
    int n = 0, max_obst = 0, device = 0;
    uint32_t flags = 0;
    uint32_t profile_mask = ~0u;   // HOPE_F_PROFILE: kernels (bit = HOPE_K_*) whose launches are bracketed by events
    bool have_tables = false, have_scenes = false;
    char arch[64] = {0};
    // device memory
    double* verts = nullptr;
    int32_t* n_obst = nullptr;
    double* scene_c = nullptr;
    double* state = nullptr;
    double* cs = nullptr;         // [n][2] cos / sin of state's heading (motion launch -> observation launch)
    int32_t* tstep = nullptr;
    double* tab = nullptr;
    double* pmax = nullptr;
    uint16_t* mask_lut = nullptr;   // count-interval table of the action mask's coarse beams (MASK_LUT_*, hope_step_kernel.h)
    double* mask_bsc = nullptr;     // [NBEAM] bins per metre of every coarse beam's table
    double* hull_base = nullptr;
    double* beam_ab = nullptr;
    int32_t* rs_count = nullptr;
    int32_t* rs_list = nullptr;
    double* rs_in = nullptr;      // [2 n][RS_IN_WORDS] search inputs by queue position (k_rs_compact -> k_rs_words / k_rs_segs)
    uint8_t* rs_flag = nullptr;
    double* kin = nullptr;
    double* post = nullptr;        // [n][POST_WORDS] k_env_step -> k_post hand-over
    double* traj = nullptr;        // HOPE_F_IMAGE: [n][20][3] ring of vehicle.trajectory
    int32_t* traj_len = nullptr;   // HOPE_F_IMAGE: [n] len(vehicle.trajectory)
    int32_t* traj_valid = nullptr; // HOPE_F_IMAGE: [n] entries below this index have span tables in bev_scratch
    int32_t* layer_valid = nullptr; // HOPE_F_IMAGE: [n] the static image layer (obstacles, start outline, dest) matches the scene's map
    uint8_t* bev_layer = nullptr;   // HOPE_F_IMAGE: [n][64 KiB] the static layer, 2 bits per pixel, tiled
    uint8_t* bev_dyn = nullptr;     // HOPE_F_IMAGE: [n][256 KiB] the trajectory layer, one byte per pixel, tiled
    int32_t* bev_list = nullptr;    // HOPE_F_IMAGE: [1 + n] count + scenes whose layer must be rebuilt this step
    int32_t* bev_legacy = nullptr;  // HOPE_F_IMAGE: [1 + n] count + scenes for the per-tile raster launch of k_bev_image
    int* bev_scratch = nullptr;    // HOPE_F_IMAGE: [n][BEV_SCENE_INTS]
    // per tile class (0: n_obst <= SMALL_TILE, 1: larger) dense scene lists; classes are static between set_scenes calls
    int32_t* cls_list[2] = {nullptr, nullptr};
    int cls_count[2] = {0, 0};
    std::vector<int32_t> n_obst_host;
    std::vector<uint8_t> slot_cls_host;          // draw / launch class of every scene slot (0: <= 32 obstacles, 1: larger lots)
    uint8_t* slot_cls = nullptr;                 // the same on the device
    double* rs_rec = nullptr;
    int32_t* rs_surv_count = nullptr;            // [MAX_CHAINS] two-kernel validation: searches k_rs_screen left a word of, per chain
    int2* rs_surv = nullptr;                     // [2 n] (queue index, queue entry), laid out like rs_list
    uint8_t* active_snap = nullptr; // [n] the caller's `active` mask as the motion launch saw it (read by k_rs_compact, HOPE_DEFER_RS)
    // StepCold (rarely used kernel parameters) in device memory: a ring of immutable versions, uploaded stream-ordered on change
    static constexpr int COLD_RING = 8;
    StepCold* cold_dev = nullptr;   // [COLD_RING]
    StepCold* cold_host = nullptr;  // [COLD_RING] pinned
    hipEvent_t cold_ev[COLD_RING] = {};
    int cold_idx = -1;
    StepCold cold_last;
    uint64_t pool_generation = 0;   // bumped by every change of what a draw can return (pool commit / upload, Dragon-Lake cases)
    uint64_t redraw_seed = 0;       // HOPE_AUTO_REDRAW
    float4* obb = nullptr;          // [n][max_obst] obstacle boxes (xmin, xmax, ymin, ymax), float32 rounded outwards
    float4* fverts = nullptr;       // [n][max_obst][2] float32 view of the obstacles relative to (map box xmin, ymin): obstacle_f32
    float4* fbox = nullptr;         // [n][max_obst]
    uint8_t* eflag = nullptr;       // [n][eflag_stride(max_obst)]
    // staging for set_scenes
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // HOPE_F_PROFILE: event pairs recorded on the launch stream, drained by hope_env_kernel_ms
    struct EvPair { hipEvent_t a, b; int kind; uint64_t seq; };   // seq: the hope_env_step / reset_obs call the launch belongs to
    uint64_t step_seq = 0;
    double union_ms[HOPE_N_KERNELS] = {};       // per kernel: time during which >= 1 launch of a call was running
    int64_t union_calls[HOPE_N_KERNELS] = {};
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> free_events;
    double ms[HOPE_N_KERNELS] = {};
    int64_t launches[HOPE_N_KERNELS] = {};
    // scene pool (hope_env_set_pool): complete scenes resident in HBM; hope_env_redraw copies one into a finished slot
    int pool_n = 0;
    double* pool_verts = nullptr;   // [pool_n][max_obst][8]
    double* pool_c = nullptr;       // [pool_n][SC_WORDS]
    int32_t* pool_nobst = nullptr;  // [pool_n]
    // double-buffered storage behind the pointers above: a new pool is uploaded (asynchronously, from pinned staging) into the
    // set the step kernels are NOT reading and swapped in by stream order (hope_env_pool_staging / hope_env_commit_pool)
    struct PoolSet { double* verts = nullptr; double* c = nullptr; int32_t* nobst = nullptr; int32_t* list[2] = {nullptr, nullptr}; int cap = 0, list_cap = 0; };
    PoolSet pset[2];
    int pactive = -1;
    struct PoolStage { double* start = nullptr; double* dest = nullptr; double* bbox = nullptr; double* verts = nullptr; int32_t* nobst = nullptr;
                       int32_t* list = nullptr; int cap = 0; bool busy = false; };
    PoolStage pstage;                            // pinned host memory
    double* pstage_dev = nullptr;                // device staging of start | dest | bbox
    int pstage_dev_cap = 0;
    hipStream_t pool_stream = nullptr;
    hipEvent_t ev_pool_ready = nullptr, ev_pool_copied = nullptr, ev_last_step = nullptr;
    bool pool_wait_pending = false;
    // hope_env_commit_pool_relaxed: the uploaded set takes over once ev_pool_ready has passed (apply_pending_pool)
    struct PendingPool { bool on = false; int set = 0, n_pool = 0, cls_n[2] = {0, 0}; uint64_t content = 0; std::vector<int32_t> nobst_host; } pend;
    int32_t* pool_cls[2] = {nullptr, nullptr};   // pool entries of each tile class
    int pool_cls_n[2] = {0, 0};
    std::vector<int32_t> pool_nobst_host;        // n_obst of the pool's complete scenes (class lists are rebuilt from it)
    // Dragon-Lake-Parking cases drawn on the device (hope_env_set_dlp_cases)
    DlpCases dlp = {};
    void* dlp_mem[6] = {};
    int32_t* pool_overflow = nullptr;            // [1] device counter: draws truncated to max_obst obstacles
    int32_t* cur_pool = nullptr;    // [n] pool entry a scene currently holds (-1: uploaded by set_scenes; -1 - c... see hope_env.h)
    uint32_t* episode = nullptr;    // [n] redraw counter (part of the draw's hash)
    // HOPE_F_OVERLAP: the two tile classes' launches go to two streams (fork / join with events)
    static constexpr int MAX_CHAINS = 8;                    // launch chains in flight: tile classes x HOPE_CHAINS sub-lists
    int sub_chains = 1;
    bool sub_chains_auto = true;                            // HOPE_CHAINS not given: hope_env_step picks (single-class batches, see there)
    hipStream_t side[MAX_CHAINS] = {};                     // [0] unused: chain 0 runs on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join[MAX_CHAINS] = {}, ev_step[2] = {}, ev_segs[2] = {}, ev_post[2] = {};
    hipEvent_t ev_collect = nullptr;                        // round 6: ONE join point for the caller's stream (the library stream of the other chain's observation collects the rest)
    bool last_via_steps = false;                            // the last step recorded ev_step[0 / 1] behind both motion launches (the pool's only readers)
    hipEvent_t ev_bev[2] = {};                              // image: fork / join of the static-layer rebuild next to k_bev_prep
    int rs_parity = 0;                                      // which of the two queue counters of a chain this step uses (pipelined steps)
    int queue_of_role[MAX_CHAINS] = {-1, -1, -1, -1, -1, -1, -1, -1};   // measured hardware-queue class of role r's stream ([0]: the NULL stream)
    int n_queues = 0;
    double queue_check_ms = 0.0;
    // HOPE_DEFER_RS: the chains of the last step have not been joined into the caller's stream (events ev_join[1], ev_join[RS_SIDE])
    static constexpr int RS_SIDE = 5;                       // the stream of the first chain when it may not run on the caller's
    bool rs_pending = false;
};


// Live handles: hope_env_destroy on a pointer that is not (or no longer) a handle, and the hot entry points on a destroyed one, fail
// with HOPE_EINVAL instead of touching freed memory (a ~50 ns set lookup per call next to ~15 launches).
#include <mutex>
#include <unordered_set>
static std::mutex g_live_m;
static std::unordered_set<const void*> g_live;
static bool is_live(const void* h) {
    std::lock_guard<std::mutex> lk(g_live_m);
    return h && g_live.count(h) != 0;
}

// every entry point runs on the handle's device and puts the caller's current device back (a process that drives
// several GPUs keeps PyTorch's notion of the current device)
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

static hipEvent_t get_event(hope_env* h) {
    if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
struct EventTimer : LaunchTimer {
    hope_env* h;
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;
    bool failed = false, skip = false;
    explicit EventTimer(hope_env* h_) : h(h_) {}
    void begin(int k, hipStream_t s) override {
        kind = k;
        skip = !((h->profile_mask >> k) & 1u);
        if (skip) return;
        a = get_event(h); b = get_event(h);
        if (!a || !b || hipEventRecord(a, s) != hipSuccess) failed = true;
    }
    void end(hipStream_t s) override {
        if (skip) return;
        if (failed || hipEventRecord(b, s) != hipSuccess) { failed = true; return; }
        h->pending.push_back({a, b, kind, h->step_seq});
    }
};

static int drain_events(hope_env* h) {
    // per launch: duration; per (call, kernel): the union of its launches' intervals (with HOPE_F_OVERLAP the launches of the
    // two tile classes run concurrently on two streams; event times are compared through hipEventElapsedTime)
    for (size_t i = 0; i < h->pending.size(); i++) {
        auto& p = h->pending[i];
        float ms = 0;
        hipError_t e = hipEventSynchronize(p.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e != hipSuccess) return fail(HOPE_EHIP, std::string("event timing: ") + hipGetErrorString(e));
        h->ms[p.kind] += ms;
        h->launches[p.kind] += 1;
    }
    for (size_t i = 0; i < h->pending.size(); i++) {
        auto& p = h->pending[i];
        if (!p.a) continue;                                   // already merged into an earlier launch of its call
        // intervals of this (call, kernel) relative to p.a: merge (at most a handful per call)
        std::vector<std::pair<float, float>> iv;
        for (size_t j = i; j < h->pending.size(); j++) {
            auto& q = h->pending[j];
            if (!q.a || q.kind != p.kind || q.seq != p.seq) continue;
            float t0 = 0, t1 = 0;
            hipError_t e = hipSuccess;
            if (j != i) { hipEventSynchronize(q.b); e = hipEventElapsedTime(&t0, p.a, q.a); }
            if (e == hipSuccess) e = hipEventElapsedTime(&t1, p.a, q.b);
            if (e != hipSuccess) { t0 = 0; t1 = 0; }          // (a later event recorded "before" the reference: negative times are fine)
            iv.push_back({t0, t1});
            if (j != i) { h->free_events.push_back(q.a); h->free_events.push_back(q.b); q.a = nullptr; }
        }
        std::sort(iv.begin(), iv.end());
        double total = 0, lo = iv[0].first, hi = iv[0].second;
        for (size_t k = 1; k < iv.size(); k++) {
            if (iv[k].first <= hi) hi = std::max<double>(hi, iv[k].second);
            else { total += hi - lo; lo = iv[k].first; hi = iv[k].second; }
        }
        total += hi - lo;
        h->union_ms[p.kind] += total;
        h->union_calls[p.kind] += 1;
        h->free_events.push_back(p.a);
        h->free_events.push_back(p.b);
        p.a = nullptr;
    }
    h->pending.clear();
    return HOPE_OK;
}

// ------------------------------------------------------------------------------------------------
// scene upload kernels
// ------------------------------------------------------------------------------------------------
namespace {

// one thread per uploaded scene: constants, derived dest box, episode state reset
__global__ void k_set_scene_consts(int n, const int32_t* ids, const double* start, const double* dest,
                                   const double* bbox, const int32_t* nob, double* scene_c, double* state,
                                   int32_t* tstep, int32_t* n_obst, double* traj, int32_t* traj_len, int32_t* traj_valid,
                                   int32_t* layer_valid) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int s = ids ? ids[k] : k;
    double* c = scene_c + (size_t)s * SC_WORDS;
    fill_scene_consts(c, start + 3 * k, dest + 3 * k, bbox + 4 * k);
    if (!state) return;                                           // (pool entries: constants only)
    double* st = state + (size_t)s * ST_WORDS;
    st[0] = start[3 * k]; st[1] = start[3 * k + 1]; st[2] = start[3 * k + 2]; st[3] = 0.0;
    tstep[s] = 0;
    n_obst[s] = nob[k];
    if (traj) {                                                   // vehicle.reset: trajectory = [start]  vehicle.py:132-133
        double* tr = traj + (size_t)s * BEV_TRAJ_LEN * 3;
        tr[0] = st[0]; tr[1] = st[1]; tr[2] = st[2];
        traj_len[s] = 1;
        traj_valid[s] = 0;
        layer_valid[s] = 0;                                           // a new map: the static image layer is rebuilt
    }
}

// hope_env_create's hardware-queue measurement: one wave that idles for `ticks` of the 100 MHz real-time counter
// and reports when it ran (stamps[0] = first, stamps[1] = last reading of the counter, which all queues share): two spins on streams of
// DIFFERENT hardware queues overlap in time, two on the same queue cannot -- decided on the GPU's own clock, not the host's
__global__ void k_spin(long long ticks, long long* stamps) {
    const long long t0 = (long long)wall_clock64();
    long long t1 = t0;
    while (t1 - t0 < ticks) { __builtin_amdgcn_s_sleep(16); t1 = (long long)wall_clock64(); }
    if (stamps && threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; }
}

__global__ void k_debug_math(int fn, int n, const double* a, const double* b, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r;
    switch (fn) {
        case 0: r = hm_sin(x); break;
        case 1: r = hm_cos(x); break;
        case 2: r = hm_tan(x); break;
        case 3: r = hm_atan2(x, y); break;
        case 4: r = hm_asin(x); break;
        case 5: r = hm_acos(x); break;
        case 6: r = hm_hypot(x, y); break;
        case 7: r = hm_fmod(x, y); break;
        case 8: r = hm_tanh(x); break;
        case 9: r = hm_exp(x); break;
        case 10: r = sqrt(x); break;
        case 12: r = div_by_20(x); break;                  // k_kinematics' division by MINI_ITER (hope_step_kernel.h)
        case 13: r = mask_fraction((int)x); break;         // the action mask's k / n_iter (hope_step_kernel.h)
        default: r = x / y; break;
    }
    out[i] = r;
}

// Calibration of the memory-side traffic counters (FETCH_SIZE / WRITE_SIZE) on THIS library's access widths: a grid-stride sweep
// over a buffer far larger than the 256 MiB Infinity Cache, with a known byte count (tools/pmc_calib.py).  MODE: reads 0 = 16 B per
// lane (the obstacle tile copies; the guide's calibrated case), 1 = 8 B per lane (kin / post / table rows), 2 = 4 B per lane, 3 = one
// 8-byte word per 64-byte line (one lane per scene record); writes 4 = 16 B, 5 = 8 B, 6 = 4 B per lane (float32 observations),
// 7 = one 8-byte word per 64-byte line.
template <int MODE>
__global__ __launch_bounds__(256) void k_traffic_calib(size_t bytes, const char* src, char* dst, double* sink) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    if (MODE == 0) { const double2* q = (const double2*)src; for (size_t i = tid; i < bytes / 16; i += nth) { const double2 v = q[i]; acc += v.x + v.y; } }
    if (MODE == 1) { const double* q = (const double*)src; for (size_t i = tid; i < bytes / 8; i += nth) acc += q[i]; }
    if (MODE == 2) { const float* q = (const float*)src; for (size_t i = tid; i < bytes / 4; i += nth) acc += (double)q[i]; }
    if (MODE == 3) { const double* q = (const double*)src; for (size_t i = tid; i < bytes / 64; i += nth) acc += q[8 * i]; }
    if (MODE == 4) { double2* q = (double2*)dst; for (size_t i = tid; i < bytes / 16; i += nth) q[i] = make_double2((double)i, 1.0); }
    if (MODE == 5) { double* q = (double*)dst; for (size_t i = tid; i < bytes / 8; i += nth) q[i] = (double)i; }
    if (MODE == 6) { float* q = (float*)dst; for (size_t i = tid; i < bytes / 4; i += nth) q[i] = (float)i; }
    if (MODE == 7) { double* q = (double*)dst; for (size_t i = tid; i < bytes / 64; i += nth) q[8 * i] = (double)i; }
    if (MODE < 4 && acc == 1.2345e301) sink[0] = acc;       // (keeps the loads)
}

// episode restart: pose = start, t = 0, accum = 0 for masked scenes
__global__ void k_restart(int n, const uint8_t* mask, const double* scene_c, double* state, int32_t* tstep, double* traj,
                          int32_t* traj_len, int32_t* traj_valid) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || !mask[s]) return;
    const double* c = scene_c + (size_t)s * SC_WORDS;
    double* st = state + (size_t)s * ST_WORDS;
    st[0] = c[SC_START]; st[1] = c[SC_START + 1]; st[2] = c[SC_START + 2]; st[3] = 0.0;
    tstep[s] = 0;
    if (traj) {
        double* tr = traj + (size_t)s * BEV_TRAJ_LEN * 3;
        tr[0] = st[0]; tr[1] = st[1]; tr[2] = st[2];
        traj_len[s] = 1;
        traj_valid[s] = 0;
    }
}

// New map at episode turnover (map.reset + vehicle.reset, car_parking_base.py:127-137), without the host: a finished scene
// (mask) takes a pool entry of ITS tile class -- the dense per-class launch lists stay valid -- picked by a counter-based
// hash of (seed, scene, episodes drawn so far), copies its obstacle tile and constants and restarts.  One wave per scene.
__global__ __launch_bounds__(64) void k_redraw(int max_obst, const uint8_t* mask, uint64_t seed, const int32_t* pl0, int n0,
                                               const int32_t* pl1, int n1, const double* pverts, const double* pc,
                                               const int32_t* pnob, double* verts, double* scene_c, int32_t* n_obst,
                                               double* state, int32_t* tstep, double* traj, int32_t* traj_len,
                                               int32_t* traj_valid, int32_t* cur_pool, uint32_t* episode, float4* obb, DlpCases dlp,
                                               int32_t* overflow, const uint8_t* slot_cls, int32_t* layer_valid, float4* fverts,
                                               float4* fbox, uint8_t* eflag) {
    __shared__ double c24[SC_WORDS];
    const int s = blockIdx.x, lane = threadIdx.x;
    if (!mask[s]) return;
    const int cls = slot_cls[s] ? 1 : 0;
    const int32_t* pl = cls ? pl1 : pl0;
    const int cnt = cls ? n1 : n0;
    const uint32_t ep = episode[s];
    __syncthreads();
    double* c = scene_c + (size_t)s * SC_WORDS;
    if (cnt > 0) {
        const uint64_t key = mix64(seed ^ mix64(((uint64_t)s << 32) | ep));
        const int j = pl[(int)(key % (uint64_t)cnt)];
        int nob;
        if (j >= 0) {
            nob = pnob[j];
            const double2* src = (const double2*)(pverts + (size_t)j * max_obst * 8);
            double2* dst = (double2*)(verts + (size_t)s * max_obst * 8);
            for (int v = lane; v < 4 * nob; v += WAVE) dst[v] = src[v];
            const double fox = pc[(size_t)j * SC_WORDS + SC_BBOX], foy = pc[(size_t)j * SC_WORDS + SC_BBOX + 2];
            for (int o = lane; o < nob; o += WAVE) {
                const double* pv = pverts + ((size_t)j * max_obst + o) * 8;
                obb[(size_t)s * max_obst + o] = obstacle_box(pv);
                obstacle_f32(pv, fox, foy, fverts + ((size_t)s * max_obst + o) * 2, fbox + (size_t)s * max_obst + o,
                             eflag + (size_t)s * eflag_stride(max_obst) + o);
            }
            if (lane < SC_WORDS) c24[lane] = pc[(size_t)j * SC_WORDS + lane];
        } else                                                // a Dragon-Lake-Parking case, drawn exactly as the fused turnover draws it
            nob = draw_dlp_case(dlp, -2 - j, mix64(key ^ 0xD1B54A32D192ED03ull), max_obst, verts + (size_t)s * max_obst * 8,
                                obb + (size_t)s * max_obst, fverts + (size_t)s * max_obst * 2, fbox + (size_t)s * max_obst,
                                eflag + (size_t)s * eflag_stride(max_obst), c24, nullptr, overflow, lane);
        __syncthreads();
        if (lane < SC_WORDS) c[lane] = c24[lane];
        if (lane == 0) { n_obst[s] = nob; cur_pool[s] = j; if (layer_valid) layer_valid[s] = 0; }
    } else if (lane < SC_WORDS) c24[lane] = c[lane];
    __syncthreads();
    if (lane == 0) {
        episode[s] = ep + 1;
        double* st = state + (size_t)s * ST_WORDS;
        st[0] = c24[SC_START]; st[1] = c24[SC_START + 1]; st[2] = c24[SC_START + 2]; st[3] = 0.0;
        tstep[s] = 0;
        if (traj) {
            double* tr = traj + (size_t)s * BEV_TRAJ_LEN * 3;
            tr[0] = st[0]; tr[1] = st[1]; tr[2] = st[2];
            traj_len[s] = 1;
            traj_valid[s] = 0;
        }
    }
}

// Reeds-Shepp work queues: the flagged scenes of each tile class (blockIdx.y), compacted.  Each block scans its share
// of the class's scene list (thread = a few consecutive entries, block-wide exclusive scan) and reserves one
// contiguous range of the queue with a single atomicAdd.
// Small workgroups on purpose: the kernel runs while other streams keep every CU busy with one-wave workgroups, and a
// 1024-thread workgroup then waits for 16 free wave slots on ONE CU (measured: up to 0.4 ms of queueing).
#ifndef HOPE_COMPACT_THREADS
#define HOPE_COMPACT_THREADS 256
#endif
constexpr int COMPACT_THREADS = HOPE_COMPACT_THREADS;
// The gate itself (car_parking_base.py:293-294: t > 1, status CONTINUE, closer than RS_MAX_DIST to the dest -- of the
// finished step, from k_env_step's hand-over record) is evaluated here, and the Reeds-Shepp outputs of every scene of the
// class are cleared here, so that the chain motion -> compact -> words -> segs -> validate does not wait for k_post.
// One thread per scene: the queue is sorted by scene inside each block's 256 scenes; the blocks append in arrival order.
template <typename OT>
__global__ __launch_bounds__(COMPACT_THREADS) void k_rs_compact(const int32_t* list, int n, uint8_t* flag, const uint8_t* active,
                                                                 int32_t* out, int32_t* rs_count, const double* post,
                                                                 const double* scene_c, int8_t* rs_word, void* rs_lengths, const int32_t* n_obst,
                                                                 int32_t* surv_count, double* rs_in) {
    __shared__ int wsum[COMPACT_THREADS / WAVE];
    __shared__ int base;
    if (surv_count && blockIdx.x == 0 && threadIdx.x == 0) *surv_count = 0;    // k_rs_screen's queue of this chain (same stream, behind the last walk)
    const int i = blockIdx.x * COMPACT_THREADS + threadIdx.x;
    int s = -1;
    bool gate = false;
    if (i < n) {
        s = list[i];
        if (!active || active[s]) {
            const double* pr = post + (size_t)s * POST_WORDS;
            const double* sc = scene_c + (size_t)s * SC_WORDS;
            const int packed = __double2loint(pr[7]);
            const int status = packed & 0xff, t = packed >> 8;
            const double ddx = pr[3] - sc[SC_DEST], ddy = pr[4] - sc[SC_DEST + 1];
            gate = t > 1 && status == HOPE_STATUS_CONTINUE && sqrt(ddx * ddx + ddy * ddy) < RS_MAX_DIST;
            flag[s] = gate;
            // cleared here; k_rs_validate fills them for the scenes whose search finds a path: {NONE x 5, 0, 0, 0}
            const unsigned long long none = (unsigned char)HOPE_RS_NONE;
            *(unsigned long long*)(rs_word + 8 * (size_t)s) = none | none << 8 | none << 16 | none << 24 | none << 32;
            if (rs_lengths) {
#pragma unroll
                for (int k = 0; k < 5; k++) ((OT*)rs_lengths)[5 * (size_t)s + k] = (OT)0;
            }
        }
    }
    // exclusive scan over the block: inside the wave by ballot, across the 4 waves through LDS
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long m = __ballot(gate);
    const int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < COMPACT_THREADS / WAVE; w++) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
    if (threadIdx.x == 0) base = total ? atomicAdd(rs_count, total) : 0;
    __syncthreads();
    if (gate) {
        const int qpos = base + woff + before;
        out[qpos] = rs_list_pack(s, n_obst[s]);                                     // (hope_internal.h)
        // the search's inputs by queue position (RS_IN_WORDS): the pose is the finished step's final pose (CONTINUE: also what
        // `state` holds), so nothing behind this kernel on the search stream reads `state` / `post` / the scene constants
        const double* pr = post + (size_t)s * POST_WORDS;
        const double* sc = scene_c + (size_t)s * SC_WORDS;
        double* in = rs_in + (size_t)qpos * RS_IN_WORDS;
        const double q0x = pr[3], q0y = pr[4], q0w = pr[5];
        in[0] = q0x; in[1] = q0y; in[2] = q0w;
        in[3] = sc[SC_BBOX]; in[4] = sc[SC_BBOX + 1]; in[5] = sc[SC_BBOX + 2]; in[6] = sc[SC_BBOX + 3];
        // generate_path (reeds_shepp.py:540-557): the goal in the start frame, scaled by the maximum curvature
        const double dx = sc[SC_DEST] - q0x, dy = sc[SC_DEST + 1] - q0y;
        const double PHI = sc[SC_DEST + 2] - q0w;
        const double c = hm_cos(q0w), sn = hm_sin(q0w);
        const double X = (c * dx + sn * dy) * RS_MAXC;
        const double Y = (-sn * dx + c * dy) * RS_MAXC;
        double sPHI, cPHI;                                    // hm_sincos is exactly odd / even: serves -PHI too
        hm_sincos(PHI, &sPHI, &cPHI);
        in[7] = X; in[8] = Y; in[9] = PHI; in[10] = sPHI; in[11] = cPHI;
        in[12] = X * cPHI + Y * sPHI;                         // "backwards" (:206-207, :376-377)
        in[13] = X * sPHI - Y * cPHI;
    }
}

// one block per uploaded scene: copy its obstacle tile
__global__ void k_set_scene_tiles(const int32_t* ids, const int32_t* nob, const double* verts_in, const double* bbox_in, double* verts,
                                  float4* obb, float4* fverts, float4* fbox, uint8_t* eflag, int max_obst) {
    int k = blockIdx.x;
    int s = ids ? ids[k] : k;
    const double2* src = (const double2*)(verts_in + (size_t)k * max_obst * 8);
    double2* dst = (double2*)(verts + (size_t)s * max_obst * 8);
    int nv = 4 * nob[k];
    for (int v = threadIdx.x; v < nv; v += blockDim.x) dst[v] = src[v];
    if (obb) {
        const double fox = bbox_in[4 * (size_t)k], foy = bbox_in[4 * (size_t)k + 2];          // frame origin of the float32 view
        for (int o = threadIdx.x; o < nob[k]; o += blockDim.x) {
            const double* pv = verts_in + ((size_t)k * max_obst + o) * 8;
            obb[(size_t)s * max_obst + o] = obstacle_box(pv);
            obstacle_f32(pv, fox, foy, fverts + ((size_t)s * max_obst + o) * 2, fbox + (size_t)s * max_obst + o,
                         eflag + (size_t)s * eflag_stride(max_obst) + o);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
template <int PART, bool FKIN = false>
static void launch_env_step(bool of64, bool af64, dim3 grid, dim3 block, size_t lds, hipStream_t sc, const StepParams& p) {
    if (of64 && af64) hipLaunchKernelGGL((k_env_step<double, double, false, PART, FKIN>), grid, block, lds, sc, p);
    else if (of64) hipLaunchKernelGGL((k_env_step<double, float, false, PART, FKIN>), grid, block, lds, sc, p);
    else if (af64) hipLaunchKernelGGL((k_env_step<float, double, false, PART, FKIN>), grid, block, lds, sc, p);
    else hipLaunchKernelGGL((k_env_step<float, float, false, PART, FKIN>), grid, block, lds, sc, p);
}

extern "C" {

// HOPE_DEFER_RS bookkeeping: `s` waits for the chains the last step left unjoined / the host does
static int join_rs(hope_env_t* h, hipStream_t s) {
    if (!h->rs_pending) return HOPE_OK;
    HIPCHK(hipStreamWaitEvent(s, h->ev_join[1], 0));
    HIPCHK(hipStreamWaitEvent(s, h->ev_join[hope_env::RS_SIDE], 0));
    h->rs_pending = false;
    return HOPE_OK;
}
static int settle_rs(hope_env_t* h) {
    if (!h || !h->rs_pending) return HOPE_OK;
    DeviceGuard guard(h->device);
    HIPCHK(hipEventSynchronize(h->ev_join[1]));
    HIPCHK(hipEventSynchronize(h->ev_join[hope_env::RS_SIDE]));
    h->rs_pending = false;
    return HOPE_OK;
}

const char* hope_last_error(void) { return g_err.c_str(); }
int hope_abi_version(void) { return HOPE_ABI_VERSION; }

static int destroy_impl(hope_env_t* h);
int hope_env_create(hope_env_t** out, int n_scenes, int max_obstacles, int device_id, uint32_t flags) {
    if (!out || n_scenes <= 0 || max_obstacles <= 0) return fail(HOPE_EINVAL, "hope_env_create: bad argument");
    if (flags & 0x20)
        return fail(HOPE_EINVAL, "hope_env_create: flag 0x20 (hipGraph replay of the step, ABI <= 6) was removed in ABI 7: it was slower than plain launches at every batch size");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(HOPE_ENODEV, std::string("hope_env_create: no HIP device (") + hipGetErrorString(e) +
                                     "); libhope_env has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(HOPE_EINVAL, "hope_env_create: device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    size_t lds = step_lds_bytes(max_obstacles);
    size_t lds_rs = rs_lds_bytes(max_obstacles);
    if (max_obstacles > HOPE_MAX_OBSTACLES)
        return fail(HOPE_EINVAL, "hope_env_create: max_obstacles " + std::to_string(max_obstacles) + " exceeds HOPE_MAX_OBSTACLES (" +
                                 std::to_string(HOPE_MAX_OBSTACLES) + "): the Reeds-Shepp queue entry carries the obstacle count in 8 bits");
    if (lds > 160 * 1024 || lds_rs > 160 * 1024)
        return fail(HOPE_EINVAL, "hope_env_create: max_obstacles " + std::to_string(max_obstacles) + " does not fit the 160 KiB LDS tile of the step / search kernels");
    if (n_scenes >= RS_LIST_MAX_SCENES)
        return fail(HOPE_EINVAL, "hope_env_create: more than 2^24 - 1 scenes per handle");
    hope_env* h = new (std::nothrow) hope_env();
    if (!h) return fail(HOPE_ENOMEM, "hope_env_create: host allocation failed");
    h->n = n_scenes; h->max_obst = max_obstacles; h->device = device_id; h->flags = flags;
    h->n_obst_host.assign(n_scenes, 0);
    h->slot_cls_host.assign(n_scenes, 0);
    snprintf(h->arch, sizeof(h->arch), "%s", prop.gcnArchName);
    size_t N = (size_t)n_scenes;
#define ALLOC(ptr, bytes)                                                                          \
    do {                                                                                           \
        hipError_t e2 = hipMalloc((void**)&(ptr), (bytes));                                        \
        if (e2 != hipSuccess) {                                                                    \
            destroy_impl(h);                                                                       \
            return fail(HOPE_ENOMEM, std::string("hipMalloc " #ptr ": ") + hipGetErrorString(e2)); \
        }                                                                                          \
    } while (0)
    ALLOC(h->verts, N * max_obstacles * 8 * sizeof(double));
    ALLOC(h->obb, N * max_obstacles * sizeof(float4));
    ALLOC(h->fverts, N * max_obstacles * 2 * sizeof(float4));
    ALLOC(h->fbox, N * max_obstacles * sizeof(float4));
    ALLOC(h->eflag, N * (size_t)eflag_stride(max_obstacles));
    ALLOC(h->n_obst, N * sizeof(int32_t));
    ALLOC(h->scene_c, N * SC_WORDS * sizeof(double));
    ALLOC(h->state, N * ST_WORDS * sizeof(double));
    ALLOC(h->cs, N * 2 * sizeof(double));
    ALLOC(h->tstep, N * sizeof(int32_t));
    ALLOC(h->tab, (size_t)NL * NITER * NACT * sizeof(double));
    ALLOC(h->pmax, NL * sizeof(double));
    ALLOC(h->mask_lut, MASK_LUT_BYTES);
    ALLOC(h->mask_bsc, NBEAM * sizeof(double));
    ALLOC(h->hull_base, NBEAM * sizeof(double));
    ALLOC(h->beam_ab, 2 * NBEAM * sizeof(double));
    ALLOC(h->rs_count, 2 * hope_env::MAX_CHAINS * sizeof(int32_t));
    ALLOC(h->rs_list, 2 * N * sizeof(int32_t));
    ALLOC(h->rs_in, 2 * N * RS_IN_WORDS * sizeof(double));
    ALLOC(h->rs_flag, N);
    ALLOC(h->kin, N * KIN_WORDS * sizeof(double));
    ALLOC(h->post, N * POST_WORDS * sizeof(double));
    ALLOC(h->cls_list[0], N * sizeof(int32_t));
    ALLOC(h->cls_list[1], N * sizeof(int32_t));
    ALLOC(h->rs_rec, N * rs_rec_bytes_per_scene());
    ALLOC(h->rs_surv_count, hope_env::MAX_CHAINS * sizeof(int32_t));
    ALLOC(h->rs_surv, 2 * N * sizeof(int2));
    ALLOC(h->cur_pool, N * sizeof(int32_t));
    ALLOC(h->episode, N * sizeof(uint32_t));
    ALLOC(h->pool_overflow, sizeof(int32_t));
    ALLOC(h->slot_cls, N);
    ALLOC(h->active_snap, N);
    ALLOC(h->cold_dev, hope_env::COLD_RING * sizeof(StepCold));
    if (flags & HOPE_F_IMAGE) {
        ALLOC(h->traj, N * BEV_TRAJ_LEN * 3 * sizeof(double));
        ALLOC(h->traj_len, N * sizeof(int32_t));
        ALLOC(h->traj_valid, N * sizeof(int32_t));
        ALLOC(h->layer_valid, N * sizeof(int32_t));
        ALLOC(h->bev_layer, N * (size_t)BEV_LAYER_ROWS * BEV_LAYER_STRIDE);
        ALLOC(h->bev_dyn, N * BEV_DYN_BYTES);
        ALLOC(h->bev_list, (N + 1) * sizeof(int32_t));
        ALLOC(h->bev_legacy, (N + 1) * sizeof(int32_t));
        ALLOC(h->bev_scratch, N * BEV_SCENE_INTS * sizeof(int));
    }
#undef ALLOC
    if (h->traj) {
        HIPCHK(hipMemset(h->traj, 0, N * BEV_TRAJ_LEN * 3 * sizeof(double)));
        HIPCHK(hipMemset(h->traj_len, 0, N * sizeof(int32_t)));
        HIPCHK(hipMemset(h->traj_valid, 0, N * sizeof(int32_t)));
        HIPCHK(hipMemset(h->layer_valid, 0, N * sizeof(int32_t)));
        HIPCHK(hipMemset(h->bev_dyn, 0, N * BEV_DYN_BYTES));
        HIPCHK(hipMemset(h->bev_list, 0, (N + 1) * sizeof(int32_t)));
        HIPCHK(hipMemset(h->bev_legacy, 0, (N + 1) * sizeof(int32_t)));
        HIPCHK(hipMemset(h->bev_scratch, 0, N * BEV_SCENE_INTS * sizeof(int)));
    }
    HIPCHK(hipMemset(h->n_obst, 0, N * sizeof(int32_t)));
    HIPCHK(hipMemset(h->scene_c, 0, N * SC_WORDS * sizeof(double)));
    HIPCHK(hipMemset(h->state, 0, N * ST_WORDS * sizeof(double)));
    HIPCHK(hipMemset(h->tstep, 0, N * sizeof(int32_t)));
    HIPCHK(hipMemset(h->rs_count, 0, 2 * hope_env::MAX_CHAINS * sizeof(int32_t)));
    HIPCHK(hipMemset(h->rs_flag, 0, N));
    HIPCHK(hipMemset(h->rs_surv_count, 0, hope_env::MAX_CHAINS * sizeof(int32_t)));
    HIPCHK(hipMemset(h->rs_surv, 0, 2 * N * sizeof(int2)));
    HIPCHK(hipMemset(h->rs_in, 0, 2 * N * RS_IN_WORDS * sizeof(double)));
    HIPCHK(hipMemset(h->rs_list, 0, 2 * N * sizeof(int32_t)));      // (k_rs_words / k_rs_segs read queue entries before they know the queue length)
    HIPCHK(hipMemset(h->cur_pool, 0xFF, N * sizeof(int32_t)));
    HIPCHK(hipMemset(h->episode, 0, N * sizeof(uint32_t)));
    HIPCHK(hipMemset(h->pool_overflow, 0, sizeof(int32_t)));
    HIPCHK(hipMemset(h->slot_cls, 0, N));
    HIPCHK(hipMemset(h->active_snap, 1, N));
    HIPCHK(hipHostMalloc((void**)&h->cold_host, hope_env::COLD_RING * sizeof(StepCold)));
    for (int i = 0; i < hope_env::COLD_RING; i++) HIPCHK(hipEventCreateWithFlags(&h->cold_ev[i], hipEventDisableTiming));
    memset(&h->cold_last, 0, sizeof(h->cold_last));
    if (getenv("HOPE_DEBUG_PTRS")) {
        fprintf(stderr, "hope_env %p: verts %p obb %p scene_c %p state %p kin %p post %p rs_rec %p rs_list %p cls0 %p cls1 %p", (void*)h, (void*)h->verts, (void*)h->obb, (void*)h->scene_c,
                (void*)h->state, (void*)h->kin, (void*)h->post, (void*)h->rs_rec, (void*)h->rs_list, (void*)h->cls_list[0], (void*)h->cls_list[1]);
        if (h->traj) fprintf(stderr, " traj %p layer %p..%p bev_list %p scratch %p..%p", (void*)h->traj, (void*)h->bev_layer, (void*)(h->bev_layer + N * (size_t)BEV_LAYER_ROWS * BEV_LAYER_STRIDE),
                             (void*)h->bev_list, (void*)h->bev_scratch, (void*)(h->bev_scratch + N * BEV_SCENE_INTS));
        fprintf(stderr, "\n");
    }
    HIPCHK(rs_init_tables());        // (a synchronous symbol copy: here, never inside a captured step)
    if (getenv("HOPE_OBS_WPC0") || getenv("HOPE_OBS_WPC1")) {
        for (const void* f : {(const void*)k_env_step<float, float, false, 2>, (const void*)k_env_step<float, double, false, 2>,
                              (const void*)k_env_step<double, float, false, 2>, (const void*)k_env_step<double, double, false, 2>})
            HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    } else if (lds > 48 * 1024) {
        for (const void* f : {(const void*)k_env_step<float, float>, (const void*)k_env_step<float, float, true>,
                              (const void*)k_env_step<float, double>, (const void*)k_env_step<double, float>,
                              (const void*)k_env_step<double, double>,
                              (const void*)k_env_step<float, float, false, 0, true>, (const void*)k_env_step<float, double, false, 0, true>,
                              (const void*)k_env_step<double, float, false, 0, true>, (const void*)k_env_step<double, double, false, 0, true>,
                              (const void*)k_env_step<float, float, false, 1>, (const void*)k_env_step<float, double, false, 1>,
                              (const void*)k_env_step<double, float, false, 1>, (const void*)k_env_step<double, double, false, 1>,
                              (const void*)k_env_step<float, float, false, 2>, (const void*)k_env_step<float, double, false, 2>,
                              (const void*)k_env_step<double, float, false, 2>, (const void*)k_env_step<double, double, false, 2>})
            HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (flags & HOPE_F_OVERLAP) {
        const char* ch = getenv("HOPE_CHAINS");              // sub-lists per tile class, each its own chain / stream
        h->sub_chains = ch ? std::max(1, std::min(hope_env::MAX_CHAINS / 2, atoi(ch))) : 1;
        h->sub_chains_auto = ch == nullptr;
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_collect, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_bev[i], hipEventDisableTiming));
        for (int i = 0; i < 2; i++) { HIPCHK(hipEventCreateWithFlags(&h->ev_step[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_segs[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_post[i], hipEventDisableTiming)); }
        // HOPE_PRIO=1 (experiment, rejected): highest stream priority for the launch chains (the critical path), lowest for
        // the observation / image streams [2], [3], [4].  Measured 0.80 -> 1.08 ms per step at 65 536 scenes: the second
        // chain's kernels then wait hundreds of microseconds between launches behind the first chain's.
        int prio_lo = 0, prio_hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        // HOPE_PRIO (experiment): stream priorities by ROLE -- 1: chains highest, observation / image lowest; 2: chains default,
        // observation lowest; 3: chains highest, observation default; 4: chains (the search streams of pipelined steps) lowest
        static const int prio_mode = getenv("HOPE_PRIO") ? atoi(getenv("HOPE_PRIO")) : 0;
        int perm[hope_env::MAX_CHAINS] = {0, 1, 6, 3, 2, 5, 4, 7};
        // Which library stream plays which role decides which roles share a HARDWARE queue (the runtime spreads streams over a few
        // queues in creation order, and launches of streams that share one serialise): HOPE_SIDE_PERM="a,b,c,d,e,f,g" gives role i
        // (1: chain of the small-tile class, 2: image, 3 / 4: observation half of the large- / small-tile class, 5: chain of the
        // large-tile class with HOPE_DEFER_RS) the a-th ... created stream.  Default: the two observation launches on streams that do
        // not share a hardware queue (3 and 7; with 3 and 4 the larger class's observation started only when the smaller class's
        // was done, 180 us after its motion launch): 0.695 -> 0.675 ms (profiles/r04_stream_roles.txt).  Third session, final
        // (pipelined) structure, profiles/r04_stream_roles_pipelined.txt: the deferred forms use roles 1, 3, 5 (+ the caller's stream)
        // and their assignment is in the best class; role 4 matters to the JOINED form, which wants it on the 2nd stream (65 536 scenes
        // 0.672 -> 0.660 ms, 16 384: 0.343 -> 0.305), and to the second sub-chain of a single-class batch, which wants the 7th stream
        // (on the 2nd: 0.68 -> 0.89 ms) and therefore takes role 7 (hope_env_step).
        // A handle with the image (HOPE_F_IMAGE) has six streams at work in a step (the image on the caller's stream, two env, two search
        // streams, the layer rebuild): other roles have to share.  Measured over 40 random assignments under the final launch structure
        // (profiles/r04_stream_roles_pipelined.txt): 65 536 scenes 1.92 -> 1.85 ms, 8 192 scenes 0.527 -> 0.445 ms with the image; the
        // same assignment costs a step WITHOUT the image 15 % (0.60 -> 0.69 ms), hence by the handle's flag, not for everyone.
        if (flags & HOPE_F_IMAGE) { static const int pi[hope_env::MAX_CHAINS] = {0, 6, 2, 3, 1, 4, 5, 7}; for (int i = 0; i < hope_env::MAX_CHAINS; i++) perm[i] = pi[i]; }
        {
            const char* pe = getenv("HOPE_SIDE_PERM");
            if (pe) {
                int k = 1, tmp[hope_env::MAX_CHAINS] = {0, 1, 2, 3, 4, 5, 6, 7};
                for (const char* q = pe; *q && k < hope_env::MAX_CHAINS; k++) {
                    tmp[k] = atoi(q);
                    while (*q && *q != ',') q++;
                    if (*q == ',') q++;
                }
                bool used[hope_env::MAX_CHAINS] = {};
                bool ok = true;
                for (int i = 1; i < hope_env::MAX_CHAINS; i++) { if (tmp[i] < 1 || tmp[i] >= hope_env::MAX_CHAINS || used[tmp[i]]) ok = false; else used[tmp[i]] = true; }
                if (ok) for (int i = 1; i < hope_env::MAX_CHAINS; i++) perm[i] = tmp[i];
            }
        }
        hipStream_t created[hope_env::MAX_CHAINS] = {};
        for (int c = 1; c < hope_env::MAX_CHAINS; c++) {          // streams in creation order; the role of the c-th decides its priority
            int role = c;
            for (int r = 1; r < hope_env::MAX_CHAINS; r++) if (perm[r] == c) role = r;
            const bool is_obs = role >= 2 && role <= 4, is_chain = role == 1 || role == 5;
            int prio = 0;
            if (prio_mode == 1) prio = is_obs ? prio_lo : prio_hi;
            else if (prio_mode == 2) prio = is_obs ? prio_lo : 0;
            else if (prio_mode == 3) prio = is_chain ? prio_hi : 0;
            else if (prio_mode == 4) prio = is_chain ? prio_lo : 0;      // round 6: the search streams lowest, everything else default
            HIPCHK(hipStreamCreateWithPriority(&created[c], hipStreamNonBlocking, prio));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join[c], hipEventDisableTiming));
        }
        // ---- which created streams share a hardware queue: measure, then give the roles that are busy together different queues ----
        // cls[c]: queue class of the c-th created stream (c = 0: the NULL stream, where PyTorch's default current stream lives).
        // A stream is compared with one representative of every class seen so far: <= 4 classes x 7 streams pairs.
        int cls[hope_env::MAX_CHAINS];
        for (int c = 0; c < hope_env::MAX_CHAINS; c++) cls[c] = -1;
        static const bool no_check = getenv("HOPE_QUEUE_CHECK") && atoi(getenv("HOPE_QUEUE_CHECK")) == 0;
        if (!no_check) {
            const auto t_begin = std::chrono::steady_clock::now();
            const long long ticks = 10000;                    // 100 us at 100 MHz
            // Two spin kernels, one per stream: they OVERLAP on the GPU's clock iff the streams sit on different hardware queues.  An overlap
            // is proof of two queues; its absence may be a late second launch (a loaded host), so a pair that did not overlap is tried up to
            // three times before it counts as one queue (the host's wall clock with fixed thresholds misread pairs on a busy box).
            long long* stamps = nullptr;                      // [2][2], host-visible
            if (hipHostMalloc((void**)&stamps, 4 * sizeof(long long), hipHostMallocDefault) != hipSuccess) stamps = nullptr;
            bool ok = stamps != nullptr;
            auto pair_overlaps = [&](hipStream_t a, hipStream_t b) -> int {   // 1 / 0, -1: the measurement is not to be trusted
                for (int rep_ = 0; rep_ < 3; rep_++) {
                    hipStreamSynchronize(a); hipStreamSynchronize(b);
                    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, stamps);
                    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, stamps + 2);
                    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
                    const long long lo = std::max(stamps[0], stamps[2]), hi = std::min(stamps[1], stamps[3]);
                    if (hi - lo > ticks / 4) return 1;        // ran side by side for > 25 us
                    if (std::max(stamps[1], stamps[3]) - std::min(stamps[0], stamps[2]) > 20 * ticks) return -1;   // > 2 ms: the GPU is shared
                }
                return 0;
            };
            created[0] = nullptr;
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, created[1], 100LL, (long long*)nullptr);      // (first launch of the kernel: module load, not measured)
            hipStreamSynchronize(created[1]);
            int rep[hope_env::MAX_CHAINS], n_cls = 0;
            for (int c = 0; c < hope_env::MAX_CHAINS && ok; c++) {
                int found = -1;
                for (int k = 0; k < n_cls && found < 0 && ok; k++) {
                    const int ov = pair_overlaps(created[rep[k]], created[c]);
                    if (ov < 0) ok = false;                   // something else is using the GPU: do not trust any of it
                    else if (ov == 0) found = k;              // serialised: the same hardware queue
                }
                if (found < 0) { rep[n_cls] = c; found = n_cls++; }
                cls[c] = found;
            }
            if (stamps) hipHostFree(stamps);
            HIPCHK(hipGetLastError());
            h->queue_check_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            if (!ok) { for (int c = 0; c < hope_env::MAX_CHAINS; c++) cls[c] = -1; n_cls = 0; }
            h->n_queues = n_cls;
            // Assignment.  Roles busy at the same time: deferred step {caller, 5, 1, 3}; joined step {caller, 1, 3, 4}; two sub-chains of one
            // class {5, 1, 3, 7}.  Greedy in that order of importance: every role takes the free stream whose class collides with the fewest
            // roles it must not share with; ties keep the hand-found table (so a box that shows nothing to gain gets exactly round 4's
            // assignment).  Not for image handles (six busy streams on four queues: the measured table stays), not when the user gave one.
            if (n_cls >= 2 && !getenv("HOPE_SIDE_PERM") && !(flags & HOPE_F_IMAGE)) {
                static const int order[7] = {5, 1, 3, 4, 7, 2, 6};
                static const int conflicts[hope_env::MAX_CHAINS][4] = {{-1, -1, -1, -1}, {0, 5, -1, -1}, {-1, -1, -1, -1}, {0, 5, 1, -1},
                                                                       {0, 1, 3, -1}, {0, -1, -1, -1}, {-1, -1, -1, -1}, {5, 1, 3, -1}};
                int newperm[hope_env::MAX_CHAINS] = {0, 0, 0, 0, 0, 0, 0, 0};
                bool taken[hope_env::MAX_CHAINS] = {};
                for (int oi = 0; oi < 7; oi++) {
                    const int r = order[oi];
                    int best = -1, best_cost = 1 << 30;
                    for (int c = 1; c < hope_env::MAX_CHAINS; c++) {
                        if (taken[c]) continue;
                        int cost = 0;
                        for (int k = 0; k < 4; k++) {
                            const int o = conflicts[r][k];
                            if (o < 0) continue;
                            const int oc = o == 0 ? cls[0] : (newperm[o] ? cls[newperm[o]] : -2);
                            if (oc == cls[c]) cost += 4;
                        }
                        if (c != perm[r]) cost += 1;          // prefer the table's stream among equals
                        if (cost < best_cost) { best_cost = cost; best = c; }
                    }
                    newperm[r] = best; taken[best] = true;
                }
                for (int r = 1; r < hope_env::MAX_CHAINS; r++) perm[r] = newperm[r];
            }
        }
        if (getenv("HOPE_DEBUG")) {
            fprintf(stderr, "hope_env_create: stream of role 1..7 =");
            for (int r = 1; r < hope_env::MAX_CHAINS; r++) fprintf(stderr, " %d(q%d)", perm[r], cls[perm[r]]);
            fprintf(stderr, "; caller q%d; queue check %.2f ms\n", cls[0], h->queue_check_ms);
        }
        for (int r = 1; r < hope_env::MAX_CHAINS; r++) h->side[r] = created[perm[r]];
        h->queue_of_role[0] = cls[0];
        for (int r = 1; r < hope_env::MAX_CHAINS; r++) h->queue_of_role[r] = cls[perm[r]];
    }
    HIPCHK(hipDeviceSynchronize());
    { std::lock_guard<std::mutex> lk(g_live_m); g_live.insert(h); }
    *out = h;
    return HOPE_OK;
}

int hope_env_destroy(hope_env_t* h) {
    if (!h) return HOPE_OK;
    {
        std::lock_guard<std::mutex> lk(g_live_m);
        if (g_live.erase(h) == 0) return fail(HOPE_EINVAL, "hope_env_destroy: not a live handle (destroyed twice?)");
    }
    return destroy_impl(h);
}
static int destroy_impl(hope_env_t* h) {                   // (also the clean-up of a hope_env_create that failed half-way: not registered yet)
    settle_rs(h);
    DeviceGuard guard(h->device);
    drain_events(h);
    hipDeviceSynchronize();
    for (hipEvent_t e : {h->ev_collect, h->ev_bev[0], h->ev_bev[1], h->ev_fork, h->ev_step[0], h->ev_step[1], h->ev_segs[0], h->ev_segs[1], h->ev_post[0], h->ev_post[1]}) if (e) hipEventDestroy(e);
    for (int i = 0; i < hope_env::MAX_CHAINS; i++) {
        if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]);
        if (h->side[i]) hipStreamDestroy(h->side[i]);
    }
    if (h->pool_stream) hipStreamDestroy(h->pool_stream);
    for (hipEvent_t e : h->cold_ev) if (e) hipEventDestroy(e);
    if (h->cold_host) hipHostFree(h->cold_host);
    for (hipEvent_t e : {h->ev_pool_ready, h->ev_pool_copied, h->ev_last_step}) if (e) hipEventDestroy(e);
    for (void* q : {(void*)h->pstage.start, (void*)h->pstage.dest, (void*)h->pstage.bbox, (void*)h->pstage.verts, (void*)h->pstage.nobst, (void*)h->pstage.list}) if (q) hipHostFree(q);
    for (hipEvent_t e : h->free_events) hipEventDestroy(e);
    void* ptrs[] = {h->obb, h->fverts, h->fbox, h->eflag, h->verts, h->n_obst, h->scene_c, h->state, h->cs, h->tstep, h->tab, h->pmax, h->mask_lut, h->mask_bsc,
                    h->hull_base, h->beam_ab, h->rs_count, h->rs_surv_count, h->rs_surv, h->rs_list, h->rs_in, h->rs_flag, h->kin, h->post, h->cls_list[0], h->cls_list[1], h->rs_rec, h->cur_pool, h->episode, h->pset[0].verts, h->pset[0].c, h->pset[0].nobst, h->pset[0].list[0], h->pset[0].list[1], h->pset[1].verts, h->pset[1].c, h->pset[1].nobst, h->pset[1].list[0], h->pset[1].list[1], h->pstage_dev, h->pool_overflow, h->slot_cls, h->active_snap, h->cold_dev, h->dlp_mem[0], h->dlp_mem[1], h->dlp_mem[2], h->dlp_mem[3], h->dlp_mem[4], h->dlp_mem[5], h->stage, h->traj, h->traj_len, h->traj_valid, h->layer_valid, h->bev_layer, h->bev_dyn, h->bev_list, h->bev_legacy, h->bev_scratch};
    for (void* q : ptrs)
        if (q) hipFree(q);
    delete h;
    return HOPE_OK;
}

int hope_debug_step_prof(uint64_t* out, int reset) {
    static unsigned long long buf[64 * 16];
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(buf, HIP_SYMBOL(g_step_prof), sizeof(buf));
    for (int i = 0; i < 16; i++) { out[i] = 0; for (int sh = 0; sh < 64; sh++) out[i] += buf[sh * 16 + i]; }
    if (e == hipSuccess && reset) {
        for (auto& v : buf) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_step_prof), buf, sizeof(buf));
    }
    return e == hipSuccess ? HOPE_OK : fail(HOPE_EHIP, std::string("hope_debug_step_prof: ") + hipGetErrorString(e));
}

int hope_debug_census(uint64_t* out, int reset) {
    // [1] [3] [7] are minima (bit patterns of non-negative doubles), the rest are counts; 64 shards
    static unsigned long long buf[64 * 16];
    if (!out) return fail(HOPE_EINVAL, "hope_debug_census: null argument");
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(buf, HIP_SYMBOL(g_census), sizeof(buf));
    for (int i = 0; i < 16; i++) {
        const bool is_min = i == 1 || i == 3 || i == 7;
        out[i] = is_min ? ~0ull : 0;
        for (int sh = 0; sh < 64; sh++) out[i] = is_min ? std::min<uint64_t>(out[i], buf[sh * 16 + i]) : out[i] + buf[sh * 16 + i];
    }
    if (e == hipSuccess && reset) {
        for (int sh = 0; sh < 64; sh++) for (int i = 0; i < 16; i++) buf[sh * 16 + i] = (i == 1 || i == 3 || i == 7) ? 0x7ff0000000000000ull : 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_census), buf, sizeof(buf));
    }
    return e == hipSuccess ? HOPE_OK : fail(HOPE_EHIP, std::string("hope_debug_census: ") + hipGetErrorString(e));
}

int hope_debug_rs_prof(uint64_t* out, int reset) {
    hipError_t e = rs_prof_read((unsigned long long*)out, reset);
    return e == hipSuccess ? HOPE_OK : fail(HOPE_EHIP, std::string("hope_debug_rs_prof: ") + hipGetErrorString(e));
}

int hope_debug_rs_filter_stats(uint64_t* out, int reset) {
    if (!out) return fail(HOPE_EINVAL, "hope_debug_rs_filter_stats: null argument");
    hipError_t e = rs_fstat_read((unsigned long long*)out, reset);
    return e == hipSuccess ? HOPE_OK : fail(HOPE_EHIP, std::string("hope_debug_rs_filter_stats: ") + hipGetErrorString(e));
}

int hope_debug_rs_filter_dump(double* out) {
    if (!out) return fail(HOPE_EINVAL, "hope_debug_rs_filter_dump: null argument");
    hipError_t e = rs_fdump_read(out);
    return e == hipSuccess ? HOPE_OK : fail(HOPE_EHIP, std::string("hope_debug_rs_filter_dump: ") + hipGetErrorString(e));
}

int hope_debug_rs_log(int32_t* out, int cap, int32_t* n, int reset) {
    if (!out || !n || cap < 0) return fail(HOPE_EINVAL, "hope_debug_rs_log: bad argument");
    hipError_t e = rs_log_read(out, cap, n, reset);
    return e == hipSuccess ? HOPE_OK : fail(HOPE_EHIP, std::string("hope_debug_rs_log: ") + hipGetErrorString(e));
}

int hope_env_queue_check(hope_env_t* h, int32_t* queue_of_role, int32_t* n_queues, double* ms) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_queue_check: null handle");
    if (queue_of_role) for (int r = 0; r < hope_env::MAX_CHAINS; r++) queue_of_role[r] = h->queue_of_role[r];
    if (n_queues) *n_queues = h->n_queues;
    if (ms) *ms = h->queue_check_ms;
    return HOPE_OK;
}

int hope_env_num_scenes(const hope_env_t* h) { return h ? h->n : HOPE_EINVAL; }
int hope_env_max_obstacles(const hope_env_t* h) { return h ? h->max_obst : HOPE_EINVAL; }
const char* hope_env_device_arch(const hope_env_t* h) { return h ? h->arch : ""; }

// Count-interval table of the action mask's coarse beams (round 6; read by k_obs_pair's mask stage).  The scan value x of coarse beam i
// falls into bin b = floor((x - lo_i) scale_i), lo_i = hull_base_i - 1e-6 (x >= hull_base_i always); the MASK_LUT_NB bins reach to the
// table's maximum at that beam + 4e-9.  Per (i, b, action): cnt_lo <= #{k : tab <= x - 1e-9} and #{k : tab <= x} <= cnt_hi for EVERY x
// whose computed bin is b -- the bin's edges are widened by 1e-6 of a bin (the device evaluates b in float64: its rounding is ~1e-13 of
// a bin) and by 2e-9 m (the tie band of the coarse decision is 1e-9 m).  cnt_lo == cnt_hi: the count is known and no table entry lies
// within 1e-9 of x; otherwise the kernel looks at the float64 entries cnt_lo .. cnt_hi - 1 themselves.  tab: [NL][NITER][NACT] prefix-maxed.
static void build_mask_lut(const std::vector<double>& tab, const std::vector<double>& pmax, const double* hull_base,
                           std::vector<uint16_t>& lut, std::vector<double>& bsc) {
    lut.assign((size_t)MASK_LUT_ROWS * MASK_LUT_ROW, (uint16_t)0xAAAA);       // padding lanes and the neutral last row: [10, 10]
    bsc.assign(NBEAM, 0.0);
    for (int i = 0; i < NBEAM; i++) {
        const int l = UPS * i;
        const double lo = hull_base[i] - 1e-6;
        const double scale = (double)MASK_LUT_NB / (pmax[l] + 4e-9 - lo);
        bsc[i] = scale;
        for (int b = 0; b < MASK_LUT_NB; b++) {
            const double e_lo = lo + ((double)b - 1e-6) / scale - 2e-9, e_hi = lo + ((double)b + 1.0 + 1e-6) / scale + 2e-9;
            uint16_t* row = lut.data() + ((size_t)i * MASK_LUT_NB + b) * MASK_LUT_ROW;
            for (int hl = 0; hl < NACT / 2; hl++) {
                int c[2][2];
                for (int d = 0; d < 2; d++) {
                    const int a = hl + d * (NACT / 2);
                    int cl = 0, ch = 0;
                    for (int k = 0; k < NITER; k++) {
                        const double t = tab[((size_t)l * NITER + k) * NACT + a];
                        cl += t <= e_lo; ch += t <= e_hi;
                    }
                    c[d][0] = cl; c[d][1] = ch;
                }
                row[hl] = (uint16_t)(c[0][0] | c[0][1] << 4 | c[1][0] << 8 | c[1][1] << 12);
            }
        }
    }
}

// device layout of the mask table: prefix-max over k (first exceedance of a sequence == first exceedance of its running max: exact),
// transposed to [l][k][a] so that lane = action reads coalesced rows; pmax[l] = the maximum over (a, k)
static void build_mask_tab(const double* dist_star, std::vector<double>& tab, std::vector<double>& pmax) {
    tab.assign((size_t)NL * NITER * NACT, 0.0);
    pmax.assign(NL, 0.0);
    for (int l = 0; l < NL; l++) {
        double pm = -INFINITY;
        for (int a = 0; a < NACT; a++) {
            double run = -INFINITY;
            for (int k = 0; k < NITER; k++) {
                double v = dist_star[((size_t)l * NACT + a) * NITER + k];
                run = v > run ? v : run;
                tab[((size_t)l * NITER + k) * NACT + a] = run;
            }
            pm = run > pm ? run : pm;
        }
        pmax[l] = pm;
    }
}

int hope_debug_mask_lut(const double* dist_star, const double* hull_base, uint16_t* lut_out, double* scale_out) {
    if (!dist_star || !hull_base || !lut_out || !scale_out) return fail(HOPE_EINVAL, "hope_debug_mask_lut: null argument");
    std::vector<double> tab, pmax, bsc;
    std::vector<uint16_t> lut;
    build_mask_tab(dist_star, tab, pmax);
    build_mask_lut(tab, pmax, hull_base, lut, bsc);
    memcpy(lut_out, lut.data(), MASK_LUT_BYTES);
    memcpy(scale_out, bsc.data(), NBEAM * sizeof(double));
    return HOPE_OK;
}

int hope_env_upload_tables(hope_env_t* h, const double* dist_star, const double* hull_base, const double* beam_ab) {
    if (!h || !dist_star || !hull_base || !beam_ab) return fail(HOPE_EINVAL, "hope_env_upload_tables: null argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    std::vector<double> tab, pmax;
    build_mask_tab(dist_star, tab, pmax);
    HIPCHK(hipMemcpy(h->tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->pmax, pmax.data(), pmax.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<uint16_t> lut;
        std::vector<double> bsc;
        build_mask_lut(tab, pmax, hull_base, lut, bsc);
        HIPCHK(hipMemcpy(h->mask_lut, lut.data(), MASK_LUT_BYTES, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->mask_bsc, bsc.data(), NBEAM * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(h->hull_base, hull_base, NBEAM * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->beam_ab, beam_ab, 2 * NBEAM * sizeof(double), hipMemcpyHostToDevice));
    h->have_tables = true;
    return HOPE_OK;
}

static int upload_scenes(hope_env_t* h, const int32_t* ids, int n, const double* start, const double* dest, const double* bbox,
                         const double* verts, const int32_t* n_obst, double* d_scene_c, double* d_state, int32_t* d_t,
                         int32_t* d_nobst, double* d_verts, float4* d_obb, double* d_traj, int32_t* d_traj_len, int32_t* d_traj_valid,
                         int32_t* d_layer_valid);
static int apply_pending_pool(hope_env_t* h, bool wait);

// The dense per-class scene lists of the launch chains and the per-slot class byte (host mirror; reset-time only).  A slot's class
// decides which launch chain steps it (LDS tile of 32 or max_obstacles obstacles) AND which pool entries it draws at episode
// turnover: by default the size class of the map hope_env_set_scenes gave it, or what hope_env_set_draw_class says.
static int rebuild_class_lists(hope_env_t* h) {
    std::vector<int32_t> l0, l1;
    l0.reserve(h->n); l1.reserve(h->n);
    const bool two = h->max_obst > SMALL_TILE;
    for (int i = 0; i < h->n; i++) (two && h->slot_cls_host[i] ? l1 : l0).push_back(i);
    // Two launch chains run concurrently (HOPE_F_OVERLAP): the step ends with the longer one running alone.  A scene with
    // few obstacles may run in the large-tile launches too (it only gets more LDS than it needs), so the small-tile class
    // hands scenes over until the large-tile chain holds `frac` of all scenes (HOPE_CLS1_FRAC, default below).  (The handed-over
    // slots keep their class byte: they go on drawing small lots.)
    if (two && (h->flags & HOPE_F_OVERLAP)) {
        const char* fr = getenv("HOPE_CLS1_FRAC");
        // measured (mixed scene set, 25 % large-tile scenes by themselves): 0.42 gives +1 / +4 / +4 % at 4 096 / 8 192 / 16 384
        // scenes, 0 at 32 768, -3 % at 65 536 (the moved scenes lose occupancy there and nothing is left to hide)
        const double frac = fr ? atof(fr) : (h->n < 32768 ? 0.42 : 0.0);
        const size_t want1 = (size_t)(frac * h->n);
        if (l1.size() < want1) {
            const size_t move = std::min(l0.size(), want1 - l1.size());
            l1.insert(l1.end(), l0.end() - move, l0.end());
            l0.resize(l0.size() - move);
            std::sort(l1.begin(), l1.end());
        }
    }
    h->cls_count[0] = (int)l0.size(); h->cls_count[1] = (int)l1.size();
    if (!l0.empty()) HIPCHK(hipMemcpy(h->cls_list[0], l0.data(), l0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!l1.empty()) HIPCHK(hipMemcpy(h->cls_list[1], l1.data(), l1.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->slot_cls, h->slot_cls_host.data(), (size_t)h->n, hipMemcpyHostToDevice));
    return HOPE_OK;
}

int hope_env_set_scenes(hope_env_t* h, const int32_t* scene_ids, int n, const double* start, const double* dest,
                        const double* bbox, const double* verts, const int32_t* n_obst) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || n < 0 || (n > 0 && (!scene_ids || !start || !dest || !bbox || !n_obst)))
        return fail(HOPE_EINVAL, "hope_env_set_scenes: null argument");
    if (n == 0) return HOPE_OK;
    for (int k = 0; k < n; k++) {
        if (scene_ids[k] < 0 || scene_ids[k] >= h->n) return fail(HOPE_EINVAL, "hope_env_set_scenes: scene id out of range");
        if (n_obst[k] < 0 || n_obst[k] > h->max_obst) return fail(HOPE_EINVAL, "hope_env_set_scenes: n_obst exceeds max_obstacles");
        if (n_obst[k] > 0 && !verts) return fail(HOPE_EINVAL, "hope_env_set_scenes: verts is null");
    }
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    {
        int rc = upload_scenes(h, scene_ids, n, start, dest, bbox, verts, n_obst, h->scene_c, h->state, h->tstep, h->n_obst,
                               h->verts, h->obb, h->traj, h->traj_len, h->traj_valid, h->layer_valid);
        if (rc != HOPE_OK) return rc;
    }
    for (int k = 0; k < n; k++) {
        h->n_obst_host[scene_ids[k]] = n_obst[k];
        h->slot_cls_host[scene_ids[k]] = (h->max_obst > SMALL_TILE && n_obst[k] > SMALL_TILE) ? 1 : 0;
    }
    { int rc = rebuild_class_lists(h); if (rc != HOPE_OK) return rc; }
    HIPCHK(hipDeviceSynchronize());
    h->have_scenes = true;
    return HOPE_OK;
}

// The rarely used parameters of the step kernel (StepCold) live in device memory: when their content differs from the version
// uploaded last, the next ring slot is filled and copied on the caller's stream -- ahead of every launch of the step, whose chains
// fork from that stream.  A slot is immutable while launches may read it: it comes round again after COLD_RING changes, and its
// event (the copy that last filled it) is waited for before the pinned source is rewritten.
static int sync_cold(hope_env_t* h, hipStream_t s) {
    StepCold c;
    memset(&c, 0, sizeof(c));
    c.traj = h->traj; c.traj_len = h->traj_len; c.traj_valid = h->traj_valid; c.layer_valid = h->layer_valid;
    c.pool_verts = h->pool_verts; c.pool_c = h->pool_c; c.pool_nobst = h->pool_nobst;
    c.pool_cls[0] = h->pool_cls[0]; c.pool_cls[1] = h->pool_cls[1]; c.pool_cls_n[0] = h->pool_cls_n[0]; c.pool_cls_n[1] = h->pool_cls_n[1];
    c.cur_pool = h->cur_pool; c.episode = h->episode; c.redraw_seed = h->redraw_seed;
    c.dlp = h->dlp; c.pool_overflow = h->pool_overflow; c.slot_cls = h->slot_cls;
    c.fverts = h->fverts; c.fbox = h->fbox; c.eflag = h->eflag;
    if (h->cold_idx >= 0 && memcmp(&c, &h->cold_last, sizeof(c)) == 0) return HOPE_OK;
    const int slot = (h->cold_idx + 1) % hope_env::COLD_RING;
    HIPCHK(hipEventSynchronize(h->cold_ev[slot]));
    memcpy(&h->cold_host[slot], &c, sizeof(c));
    HIPCHK(hipMemcpyAsync(h->cold_dev + slot, h->cold_host + slot, sizeof(c), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(h->cold_ev[slot], s));
    h->cold_idx = slot;
    memcpy(&h->cold_last, &c, sizeof(c));
    return HOPE_OK;
}

// Enqueues the launches of one step on `s`.  With a side stream `s2` (HOPE_F_OVERLAP) the two tile classes run
// concurrently: fork -> { k_kinematics, k_env_step, k_rs_compact, k_rs_words, k_rs_validate of class 1 | of class 0 } -> join -> image.  Every class kernel is latency-bound per wave, so at
// <= 16 k scenes per GPU (BASELINE config 4: 8 192) one class alone cannot fill the 1024 SIMDs.
static int enqueue_step(hope_env_t* h, const void* actions, const uint8_t* active, uint32_t stages, const hope_step_out* out,
                        hipStream_t s, bool overlap, int has_action, LaunchTimer* tm) {
    StepParams p;
    memset(&p, 0, sizeof(p));
    static const uint32_t dbg_stages = getenv("HOPE_DEBUG_STAGES") ? (uint32_t)strtol(getenv("HOPE_DEBUG_STAGES"), nullptr, 0) : 0;   // profiling switches 0x1000 / 0x2000 (results invalid)
    p.n = h->n; p.max_obst = h->max_obst; p.stages = stages | dbg_stages; p.has_action = has_action;
    p.hflags = h->traj ? STEP_HF_TRAJ : 0;
    p.verts = h->verts; p.obb = h->obb; p.eflag = h->eflag; p.n_obst = h->n_obst; p.scene_c = h->scene_c; p.state = h->state; p.cs = h->cs; p.tstep = h->tstep;
    p.active = active; p.active_out = active ? h->active_snap : nullptr; p.kin = h->kin; p.actions = actions; p.post = h->post;
    p.tab = h->tab; p.pmax = h->pmax; p.mask_lut = h->mask_lut; p.mask_bsc = h->mask_bsc; p.hull_base = h->hull_base; p.beam_ab = h->beam_ab;
    p.lidar = out->lidar; p.action_mask = out->action_mask;
    p.cold = h->cold_dev + h->cold_idx;                     // (sync_cold ran on the caller's stream before any launch of this step)
    const uint8_t* active_rs = active ? h->active_snap : nullptr;   // what the Reeds-Shepp chain reads instead of the caller's mask
    const bool of64 = h->flags & HOPE_F_OBS_F64, af64 = h->flags & HOPE_F_ACTION_F64;
    dim3 block(WAVE);

    // One chain of launches per tile class (scenes with few obstacles get a small LDS tile and therefore more resident
    // waves): k_env_step -> k_rs_compact -> k_rs_words -> k_rs_validate.  The chains share nothing but read-only data
    // (own scene list, own queue counter, record slots filled from opposite ends), so with a side stream they run
    // concurrently from the fork after the kinematics to the join before the image.
    static const bool step_timing = getenv("HOPE_STEP_TIMING") != nullptr;      // cycle accounting build (tools/step_timing.py)
    const int n_cls = h->max_obst > SMALL_TILE ? 2 : 1;
    const bool want_rs = (stages & HOPE_STAGE_RS) && out->rs_word;
    // chains: (class, sub-list) pairs with work, the large-tile class first (its chains are the longer ones)
    struct Chain { int c, a, b, st; };                      // st: 0 = the caller's stream, k = side[k]
    Chain chains[hope_env::MAX_CHAINS];
    int n_chain = 0, n_streams = 1;
    int subs = overlap ? h->sub_chains : 1;
    // A batch whose scenes all sit in ONE tile class has one chain: no second stream, and none of the two-chain forms below (observation
    // half on its own stream, pipelined steps).  Cut that class's list into two sub-chains where that was measured to pay for
    // deferred steps (profiles/r04_single_class_chains.txt; same bits: the chains share nothing): the small-tile class from 32 768
    // scenes (65 536 generated lots 0.790 -> 0.680 ms; below that the single chain in the one-launch form is the faster one), the
    // large-tile class from 4 096 to 32 768 scenes (4 096 Dragon-Lake scenes 0.190 -> 0.155 ms, 16 384: 0.278 -> 0.232; 65 536: 0.698 vs 0.717, not there).
    // HOPE_CHAINS=n overrides; HOPE_AUTO_CHAINS=lo0:hi0:lo1:hi1 moves the two ranges (tests force the form at small sizes).
    bool auto_subs = false;
    if (overlap && h->sub_chains_auto && want_rs && (stages & HOPE_STAGE_OBS) && (stages & HOPE_DEFER_RS) && !(stages & HOPE_STAGE_IMG) &&
        (h->cls_count[0] == 0) != (h->cls_count[1] == 0)) {
        int range[4] = {32768, 1 << 30, 4096, 32768};
        if (const char* e = getenv("HOPE_AUTO_CHAINS")) sscanf(e, "%d:%d:%d:%d", range, range + 1, range + 2, range + 3);   // (read per call)
        const int c1 = h->cls_count[0] == 0 ? 1 : 0, cnt = h->cls_count[c1];
        if (cnt >= range[2 * c1] && cnt <= range[2 * c1 + 1]) { subs = 2; auto_subs = true; }
    }
    static const int balance = getenv("HOPE_BALANCE") ? atoi(getenv("HOPE_BALANCE")) : 100;
    // experiment: 1 = the chain of the class with more scenes is enqueued first; 2 = and the other chain starts only when the
    // first chain's motion launch is done (its kinematics / motion kernels then do not share the GPU with the critical ones)
    const int order_mode = getenv("HOPE_ORDER") ? atoi(getenv("HOPE_ORDER")) : 0;
    if (overlap && subs == 1 && n_cls == 2 && balance < 100 && h->cls_count[0] > 0 && h->cls_count[1] > 0) {
        // two streams, three chains: the large-tile class, then the tail of the small-tile class behind it on the side stream,
        // the head of the small-tile class on the caller's stream -- so that both streams finish together
        const int cut = (int)((long long)h->cls_count[0] * balance / 100);
        chains[n_chain++] = {1, 0, h->cls_count[1], 1};
        if (cut > 0) chains[n_chain++] = {0, 0, cut, 0};
        if (cut < h->cls_count[0]) chains[n_chain++] = {0, cut, h->cls_count[0], 1};
        n_streams = 2;
    } else {
        // the class with more scenes first: its chain is the step's critical path (HOPE_ORDER=0: the large-tile class first)
        const bool big_first = n_cls == 2 && order_mode > 0 && h->cls_count[0] > h->cls_count[1];
        for (int cc = n_cls - 1; cc >= 0; cc--) {
            const int c = big_first ? (n_cls - 1 - cc) : cc;
            for (int j = 0; j < subs; j++) {
                const int a = (int)((long long)h->cls_count[c] * j / subs), b = (int)((long long)h->cls_count[c] * (j + 1) / subs);
                if (b > a) { chains[n_chain] = {c, a, b, overlap ? n_chain : 0}; n_chain++; }
            }
        }
        n_streams = overlap ? n_chain : 1;
    }
    const bool fork = overlap && n_streams > 1;
    // the observation half of the step kernel on its own stream, next to the Reeds-Shepp kernels of the same class
    static const bool no_split = getenv("HOPE_NO_SPLIT") != nullptr;
    // (measured: +4.5 % at 65 536 scenes, +3.7 % at 131 072, 0 at 16 384, -3 % at 8 192 and below: two more launches per class)
    const char* split_min = getenv("HOPE_SPLIT_MIN");      // (read per call: the tests force the split at small sizes)
    // (with pipelined steps -- HOPE_DEFER_RS, below -- the one-launch form stays the faster one up to 32 768 scenes: 16 384 scenes
    // 0.249 vs 0.266 ms)
    static const bool pipe_env0 = !(getenv("HOPE_PIPE") && atoi(getenv("HOPE_PIPE")) == 0);
    const bool split = fork && n_chain == 2 && want_rs && (stages & HOPE_STAGE_OBS) && !step_timing && !no_split &&
                       h->n >= (split_min ? atoi(split_min) : ((stages & HOPE_DEFER_RS) && pipe_env0) ? 32768 : 16384);
    // HOPE_DEFER_RS: both chains on library streams, the caller's stream joins the observation half only (hope_env.h)
    // (measured: 32 768 scenes 0.433 -> 0.426 ms, 65 536 0.681 -> 0.669; 16 384 0.322 -> 0.361: below 32 768 the joined form)
    // Round 4: with pipelined steps (below) the bit pays at every batch size, in the one-launch form of the step kernel too (pipe1:
    // env stream = k_kinematics -> k_env_step, search stream = k_post -> k_rs_compact -> ... -> k_rs_validate_f): a small batch is
    // launch-latency bound, seven dependent launches per class, and overlapping the search of step k with the env launches of step
    // k + 1 hides most of that chain (8 192 scenes 0.266 -> see RESULTS.md).  HOPE_DEFER_MIN restores a threshold.
    static const bool pipe_env = !(getenv("HOPE_PIPE") && atoi(getenv("HOPE_PIPE")) == 0);
    const char* defer_min = getenv("HOPE_DEFER_MIN");
    const bool defer_ok = fork && n_chain == 2 && want_rs && (stages & HOPE_STAGE_OBS) && !step_timing && (stages & HOPE_DEFER_RS) &&
                          h->n >= (defer_min ? atoi(defer_min) : pipe_env ? 0 : 32768);
    const bool pipe1 = defer_ok && !split && pipe_env;      // pipelined steps with the one-launch step kernel
    const bool defer = (split && defer_ok) || pipe1;
    if (!defer) { int rcj = join_rs(h, s); if (rcj != HOPE_OK) return rcj; }    // (a deferred step's launches follow the unjoined ones on the same streams)
    // PIPELINED steps (round 4; with HOPE_DEFER_RS, HOPE_PIPE=0 switches it off): each tile class runs on TWO library streams --
    //   env stream  : k_kinematics -> k_env_step<motion> -> k_env_step<observation> -> k_post     (what the caller's stream joins)
    //   search stream: [motion done] k_rs_compact -> k_rs_words -> k_rs_segs -> k_rs_validate      (hope_env_wait_rs / never)
    // so that step k + 1's kinematics and motion launch do not queue behind step k's validation kernel, the longest launch of the
    // step: they only wait for step k's k_rs_segs (the last reader of `state` / `post` on the search stream) and -- through the
    // caller's stream -- for its observation.  The search of step k may then read obstacle tiles that step k + 1's episode
    // turnover is rewriting: only for scenes whose episode ended in step k + 1, whose search result nobody can read any more
    // (k_rs_compact of step k + 1 clears it, after the validation kernel, on the same stream), and all reads stay inside the
    // scene's own tile slots.  The queue counter alternates between two words per chain (the motion launch of step k + 1 zeroes
    // the one step k + 1 uses while step k's validation blocks are still reading theirs).
    const bool pipe = defer && pipe_env && n_chain == 2;
    if (want_rs) h->rs_parity ^= 1;
    if (fork) {
        HIPCHK(hipEventRecord(h->ev_fork, s));
        for (int i = 1; i < n_streams; i++) HIPCHK(hipStreamWaitEvent(h->side[i], h->ev_fork, 0));
        if (defer) HIPCHK(hipStreamWaitEvent(h->side[hope_env::RS_SIDE], h->ev_fork, 0));
    }
    bool joined_on_caller[2] = {false, false}, post_on_rs[2] = {false, false};
    hipStream_t obs_stream[2] = {nullptr, nullptr};
    // every chain's motion launch is followed by its ev_step record: those two events cover the pool's readers (the pool upload waits
    // for them instead of a per-step record on the caller's stream)
    h->last_via_steps = false;
    for (int i = 0; i < n_chain; i++) {
        const Chain& ch = chains[i];
        const int c = ch.c;
        hipStream_t sc = (fork && ch.st > 0) ? h->side[ch.st] : defer ? h->side[hope_env::RS_SIDE] : s;
        int32_t* counter = h->rs_count + i + hope_env::MAX_CHAINS * h->rs_parity;
        // two-launch form: everything the search does not wait for goes to its own stream.  (Experiment knobs: HOPE_OBS_SIDE0 / 1 =
        // which library stream carries the observation half of chain 0 / 1, HOPE_OBS_WPC0 / 1 = its waves per CU, enforced
        // through the LDS request.)
        static const int obs_side[2] = {getenv("HOPE_OBS_SIDE0") ? atoi(getenv("HOPE_OBS_SIDE0")) : 3, getenv("HOPE_OBS_SIDE1") ? atoi(getenv("HOPE_OBS_SIDE1")) : 4};
        static const int obs_wpc[2] = {getenv("HOPE_OBS_WPC0") ? atoi(getenv("HOPE_OBS_WPC0")) : 0, getenv("HOPE_OBS_WPC1") ? atoi(getenv("HOPE_OBS_WPC1")) : 0};
        hipStream_t so = (split || pipe1) ? h->side[(auto_subs && (i & 1)) ? 7 : std::max(1, std::min(hope_env::MAX_CHAINS - 1, obs_side[i & 1]))] : sc;
        // pipelined: the env stream of the class with MORE scenes is the caller's stream itself -- its kernels are the step's critical
        // cycle (kinematics -> motion -> observation -> the caller's next actions -> kinematics ...), and every hop between a library
        // stream and the caller's costs that cycle 20-30 us (two hops per step: 0.648 -> 0.60 ms).  Not with the image, which
        // runs on the caller's stream next to the observation launches.
        static const bool pipe_on_caller = !(getenv("HOPE_PIPE_CALLER") && atoi(getenv("HOPE_PIPE_CALLER")) == 0);
        // (two sub-chains of one class: neither -- both would land on the caller's stream and serialise: 0.80 vs 0.68 ms)
        const bool on_caller = pipe && pipe_on_caller && !auto_subs && !(stages & HOPE_STAGE_IMG) && h->cls_count[c] >= h->cls_count[1 - c];
        if (on_caller) so = s;
        // ... and k_post, whose outputs the caller's stream joins too, runs at the head of the search stream instead of behind the
        // observation launch (the search stream has slack, the env stream is the critical one)
        // (two-launch form only: in the one-launch form of small batches k_post stays behind the step kernel on the env stream --
        // 8 192 scenes 0.205 vs 0.188 ms -- the search chain is the longer one there; HOPE_POST_SEARCH=0 / 1 forces either)
        static const int post_on_search = getenv("HOPE_POST_SEARCH") ? atoi(getenv("HOPE_POST_SEARCH")) : -1;
        const bool post_rs = pipe && want_rs && (post_on_search < 0 ? split : post_on_search != 0);
        if (i < 2) { joined_on_caller[i] = on_caller; post_on_rs[i] = post_rs; obs_stream[i] = so; }
        hipStream_t sk = pipe ? so : sc;                        // the stream of the kinematics and the motion launch
        if (pipe && !on_caller) HIPCHK(hipStreamWaitEvent(sk, h->ev_fork, 0));
        p.tile_cap = (c == 0 && n_cls == 2) ? SMALL_TILE : h->max_obst;
        p.scene_list = h->cls_list[c] + ch.a;
        p.n_list = ch.b - ch.a;
        p.rs_count_zero = want_rs ? counter : nullptr;
        if (order_mode >= 2 && fork && n_chain == 2 && i == 1 && (split || (stages & HOPE_STAGE_IMG)))
            HIPCHK(hipStreamWaitEvent(sc, h->ev_step[0], 0));   // staggered: behind the first chain's motion launch
        // one-launch form (small batches): the step kernel's waves compute their scene's sub-step poses themselves -- the step is a
        // chain of launch latencies there, and the kinematics launch cost the critical stream ~16 us plus a launch gap
        // Only where the GPU is far from full: the wave-per-scene form spends ~2x the vector instructions of k_kinematics' four
        // lanes per scene (measured, steady ms per step, fused / separate: 4 096 scenes 0.152 / 0.156, 8 192 0.186 / 0.176,
        // 16 384 0.249 / 0.222).  HOPE_FUSE_KIN = largest batch that takes it (0: never).
        static const int fuse_kin_max = getenv("HOPE_FUSE_KIN") ? atoi(getenv("HOPE_FUSE_KIN")) : 4096;
        const bool fuse_kin = h->n <= fuse_kin_max && !split && !step_timing;
        if ((stages & HOPE_STAGE_MOTION) && has_action && !fuse_kin) {       // this class's sub-step poses head its chain
            dim3 kg((p.n_list + KIN_SCENES_PER_BLOCK - 1) / KIN_SCENES_PER_BLOCK);
            if (tm) tm->begin(HOPE_K_KINEMATICS, sk);
            if (af64) hipLaunchKernelGGL((k_kinematics<double>), kg, block, 0, sk, p.n_list, p.scene_list, h->state, actions, active, stages, h->scene_c, h->kin);
            else hipLaunchKernelGGL((k_kinematics<float>), kg, block, 0, sk, p.n_list, p.scene_list, h->state, actions, active, stages, h->scene_c, h->kin);
            if (tm) tm->end(sk);
        }
        const dim3 grid(p.n_list);
        size_t lds = step_lds_bytes(p.tile_cap);
        // pipelined: the last step's k_rs_compact / k_rs_words / k_rs_segs (search stream) read `post` and `state`, which the motion
        // launch rewrites
        if (pipe) HIPCHK(hipStreamWaitEvent(sk, h->ev_segs[i], 0));
        if (tm) tm->begin(HOPE_K_STEP, sk);
        // small-tile class, moving step: two scenes per wave (hope_motion_pair.h; HOPE_MOTION_PAIR=0 or stage bit 0x8000: the one-scene kernel)
        static const bool motion_pair = !(getenv("HOPE_MOTION_PAIR") && atoi(getenv("HOPE_MOTION_PAIR")) == 0);
        if (split && motion_pair && c == 0 && n_cls == 2 && p.tile_cap == SMALL_TILE && (stages & HOPE_STAGE_MOTION) && has_action && !(stages & 0x8000))
            hipLaunchKernelGGL(k_motion_pair, dim3((p.n_list + 1) / 2), block, MP_LDS_BYTES, sk, p);
        else if (split) launch_env_step<1>(of64, af64, grid, block, lds, sk, p);
        else if (step_timing && !of64 && !af64) hipLaunchKernelGGL((k_env_step<float, float, true>), grid, block, lds, sk, p);
        else if (fuse_kin) launch_env_step<0, true>(of64, af64, grid, block, lds, sk, p);
        else launch_env_step<0>(of64, af64, grid, block, lds, sk, p);
        if (tm) tm->end(sk);
        if (fork && n_chain == 2 && (split || pipe || (stages & HOPE_STAGE_IMG))) { HIPCHK(hipEventRecord(h->ev_step[i], sk)); h->last_via_steps = true; }   // poses final
        if (split && !pipe) HIPCHK(hipStreamWaitEvent(so, h->ev_step[i], 0));
        // k_post BEHIND the observation half on that stream: nothing waits for its outputs before the join, the observation is the
        // long launch (0.675 -> 0.669 ms; HOPE_POST_LAST=0: the round-3 order)
        static const bool post_last = !(getenv("HOPE_POST_LAST") && atoi(getenv("HOPE_POST_LAST")) == 0);
        auto launch_post = [&]() {                              // scalar outputs, reward / target arithmetic: one lane per scene
            dim3 pg((p.n_list + WAVE - 1) / WAVE);
            if (tm) tm->begin(HOPE_K_POST, so);
            if (of64) hipLaunchKernelGGL((k_post<double>), pg, block, 0, so, p.n_list, p.scene_list, active, stages, h->scene_c, h->state, h->post, h->rs_flag, *out);
            else hipLaunchKernelGGL((k_post<float>), pg, block, 0, so, p.n_list, p.scene_list, active, stages, h->scene_c, h->state, h->post, h->rs_flag, *out);
            if (tm) tm->end(so);
        };
        if (!(split && post_last) && !post_rs) launch_post();
        if (split) {
            if (tm) tm->begin(HOPE_K_STEP, so);
            const size_t lds_obs = obs_wpc[i & 1] > 0 ? std::max(lds, (size_t)((158 * 1024 / obs_wpc[i & 1]) & ~255)) : lds;
            // small-tile class: two scenes per wave (hope_obs_pair.h; HOPE_OBS_PAIR=0: the one-scene kernel, its reference)
            static const bool obs_pair = !(getenv("HOPE_OBS_PAIR") && atoi(getenv("HOPE_OBS_PAIR")) == 0);
            if (obs_pair && p.tile_cap == SMALL_TILE && !(stages & 0x8000)) {          // (0x8000: A/B switch of the tests, one-scene kernel)
                const dim3 pgrid((p.n_list + 1) / 2);
                const size_t lds_pair = obs_wpc[i & 1] > 0 ? std::max(OP_LDS_BYTES, (size_t)((158 * 1024 / obs_wpc[i & 1]) & ~255)) : OP_LDS_BYTES;
                if (of64) hipLaunchKernelGGL((k_obs_pair<double>), pgrid, block, lds_pair, so, p);
                else hipLaunchKernelGGL((k_obs_pair<float>), pgrid, block, lds_pair, so, p);
            } else
            launch_env_step<2>(of64, af64, grid, block, lds_obs, so, p);
            if (tm) tm->end(so);
            if (post_last && !post_rs) launch_post();
            if (!on_caller) HIPCHK(hipEventRecord(h->ev_join[3 + i], so));   // (the event index stays 3 + i whatever stream carries the launch)
        } else if (pipe1 && !on_caller) HIPCHK(hipEventRecord(h->ev_join[3 + i], so));     // one-launch form: the observation is part of it
        if (!want_rs) continue;
        if (pipe) HIPCHK(hipStreamWaitEvent(sc, h->ev_step[i], 0));  // the search stream starts behind this step's motion launch
        if (post_rs) {
            hipStream_t keep = so;
            so = sc;                                                  // (launch_post launches on `so`)
            launch_post();
            so = keep;
            HIPCHK(hipEventRecord(h->ev_post[i], sc));
        }
        int32_t* qlist = h->rs_list + (size_t)c * h->n + ch.a;       // this chain's part of the class's queue storage
        double* rs_in = h->rs_in + ((size_t)c * h->n + ch.a) * RS_IN_WORDS;      // ... and of the search inputs, same index
        if (tm) tm->begin(HOPE_K_RS_COMPACT, sc);
        {
            const dim3 cg((p.n_list + COMPACT_THREADS - 1) / COMPACT_THREADS);
            if (of64) hipLaunchKernelGGL(k_rs_compact<double>, cg, dim3(COMPACT_THREADS), 0, sc, p.scene_list, p.n_list, h->rs_flag, active_rs, qlist,
                                         counter, (const double*)h->post, (const double*)h->scene_c, out->rs_word, out->rs_lengths, (const int32_t*)h->n_obst, h->rs_surv_count + i, rs_in);
            else hipLaunchKernelGGL(k_rs_compact<float>, cg, dim3(COMPACT_THREADS), 0, sc, p.scene_list, p.n_list, h->rs_flag, active_rs, qlist,
                                    counter, (const double*)h->post, (const double*)h->scene_c, out->rs_word, out->rs_lengths, (const int32_t*)h->n_obst, h->rs_surv_count + i, rs_in);
        }
        if (tm) tm->end(sc);
        // pipelined steps: the next step's motion launch rewrites `state` / `post` (and, on an episode turnover, the scene constants),
        // which k_post and k_rs_compact are the last to read -- k_rs_words / k_rs_segs work from k_rs_compact's rows (RS_IN_WORDS).
        // (Rounds 4-5a waited for k_rs_segs: at small batches the step's critical cycle was motion -> compact -> words -> segs ->
        // motion; HOPE_PIPE_AFTER=segs restores that for A/B runs.)
        static const bool after_segs = getenv("HOPE_PIPE_AFTER") && !strcmp(getenv("HOPE_PIPE_AFTER"), "segs");
        if (pipe && !after_segs) HIPCHK(hipEventRecord(h->ev_segs[i], sc));
        RsParams r;
        r.n = h->n; r.max_obst = h->max_obst; r.obs_f64 = of64;
        static const int prio_front = getenv("HOPE_PRIO_FRONT") ? atoi(getenv("HOPE_PRIO_FRONT")) : 0;
        r.prio_front = prio_front;
        r.tile_cap = p.tile_cap;
        r.max_queue = p.n_list;
        r.slot_base = (c == 0) ? ch.a : h->n - 1 - ch.a;    // the two classes fill the record storage from both ends
        r.slot_dir = (c == 0) ? 1 : -1;
        r.verts = h->verts; r.obb = h->obb; r.n_obst = h->n_obst; r.scene_c = h->scene_c; r.state = h->state;
        r.fverts = h->fverts; r.fbox = h->fbox; r.eflag = h->eflag;
        r.rs_count = counter; r.rs_list = qlist; r.rs_in = rs_in;
        r.rs_rec = h->rs_rec;
        r.surv_count = h->rs_surv_count + i;
        r.surv_list = h->rs_surv + (size_t)c * h->n + ch.a;
        r.rs_word = out->rs_word; r.rs_lengths = out->rs_lengths;
        HIPCHK(launch_rs_search(r, sc, tm, (pipe && after_segs) ? h->ev_segs[i] : nullptr));
    }
    HIPCHK(hipGetLastError());
    if (stages & HOPE_STAGE_IMG) {
        BevParams b;
        b.n = h->n; b.max_obst = h->max_obst; b.verts = h->verts; b.n_obst = h->n_obst; b.scene_c = h->scene_c;
        b.state = h->state; b.traj = h->traj; b.traj_len = h->traj_len; b.traj_valid = h->traj_valid; b.scratch = h->bev_scratch; b.img = out->img;
        b.layer = h->bev_layer; b.layer_valid = h->layer_valid; b.rebuild = h->bev_list; b.dyn = h->bev_dyn;
        b.legacy_list = h->bev_legacy;
        // with auto-reset every scene shows its NEW episode's first observation, like lidar / action_mask / target
        b.active = active;
        b.debug = (stages >> 12) & 0xF;
        if (const char* e_ = getenv("HOPE_BEV_DEBUG")) b.debug |= atoi(e_);   // (profiling) further switches: 16 no block cache, 64 no image stores
        if (getenv("HOPE_BEV_LEGACY")) b.debug |= 32;          // (tests) the per-tile raster of the moving boxes instead of the trajectory layer
        if (fork && n_chain == 2) {
            // the image depends on the step kernels only (pose, trajectory ring), not on the Reeds-Shepp search: render it on
            // a third stream while the two chains run k_rs_words / k_rs_validate
            // (with HOPE_DEFER_RS both chains run on library streams and the caller's stream is idle during the step: the image,
            // which the caller waits for anyway, goes there -- the third library stream shares a hardware queue with a chain)
            hipStream_t si = defer ? s : h->side[2];
            HIPCHK(hipStreamWaitEvent(si, h->ev_step[0], 0));
            HIPCHK(hipStreamWaitEvent(si, h->ev_step[1], 0));
            // (pipelined steps: the image is on the caller's stream and the third library stream is free for the layer rebuild)
            static const bool bev_side = !(getenv("HOPE_BEV_SIDE") && atoi(getenv("HOPE_BEV_SIDE")) == 0);
            if (defer && bev_side && h->side[2] && h->ev_bev[0]) HIPCHK(launch_bev_image(b, si, tm, h->side[2], h->ev_bev[0], h->ev_bev[1]));
            else HIPCHK(launch_bev_image(b, si, tm));
            HIPCHK(hipEventRecord(h->ev_join[2], si));
        } else {
            if (fork) for (int i = 1; i < n_streams; i++) { HIPCHK(hipEventRecord(h->ev_join[i], h->side[i])); HIPCHK(hipStreamWaitEvent(s, h->ev_join[i], 0)); }
            if (split) for (int i = 0; i < 2; i++) HIPCHK(hipStreamWaitEvent(s, h->ev_join[3 + i], 0));
            HIPCHK(launch_bev_image(b, s, tm));
            return HOPE_OK;
        }
    }
    if (fork) {
        if (defer) {                                            // the chains stay unjoined: hope_env_wait_rs / the next step
            HIPCHK(hipEventRecord(h->ev_join[1], h->side[1]));
            HIPCHK(hipEventRecord(h->ev_join[hope_env::RS_SIDE], h->side[hope_env::RS_SIDE]));
            h->rs_pending = true;
        } else
            for (int i = 1; i < n_streams; i++) { HIPCHK(hipEventRecord(h->ev_join[i], h->side[i])); HIPCHK(hipStreamWaitEvent(s, h->ev_join[i], 0)); }
        if ((stages & HOPE_STAGE_IMG) && n_chain == 2) HIPCHK(hipStreamWaitEvent(s, h->ev_join[2], 0));
        if (split || pipe1) {
            // Round 6: every event the caller's stream waits for or records is a packet in ITS queue, in front of the next step's
            // kinematics -- the head of the step's critical cycle: three waits + two records made a 50 us hole behind the observation
            // launch (profiles/r05_step_timeline_pipelined.txt).  With one chain on the caller's stream the OTHER chain's observation
            // stream -- it has slack -- waits for the two k_post launches and records ONE event for the caller (HOPE_JOIN_COLLECT=0: the
            // round-5 joins).
            static const bool collect_on = !(getenv("HOPE_JOIN_COLLECT") && atoi(getenv("HOPE_JOIN_COLLECT")) == 0);
            if (collect_on && h->ev_collect && n_chain == 2 && joined_on_caller[0] != joined_on_caller[1]) {
                const int o = joined_on_caller[0] ? 1 : 0;
                for (int i = 0; i < 2; i++) if (post_on_rs[i]) HIPCHK(hipStreamWaitEvent(obs_stream[o], h->ev_post[i], 0));
                HIPCHK(hipEventRecord(h->ev_collect, obs_stream[o]));
                HIPCHK(hipStreamWaitEvent(s, h->ev_collect, 0));
            } else for (int i = 0; i < 2; i++) {
                if (!joined_on_caller[i]) HIPCHK(hipStreamWaitEvent(s, h->ev_join[3 + i], 0));
                if (post_on_rs[i]) HIPCHK(hipStreamWaitEvent(s, h->ev_post[i], 0));
            }
        }
    }
    return HOPE_OK;
}

// A scene draws its next map from the pool entries of ITS tile class (the dense per-class launch lists are fixed between
// hope_env_set_scenes calls).  A resident class without entries would silently keep its maps: refuse instead.
static int check_pool_classes(hope_env_t* h, const char* who) {
    bool resident[2] = {false, false};
    for (int i = 0; i < h->n; i++) resident[h->slot_cls_host[i] ? 1 : 0] = true;
    for (int c = 0; c < 2; c++)
        if (resident[c] && h->pool_cls_n[c] == 0)
            return fail(HOPE_ESTATE, std::string(who) + ": scene slots of the " + (c ? "large" : "small (<= 32 obstacles)") +
                                     " size class are resident but the pool holds no map of that size class (a slot draws from its own class: "
                                     "hope_env_set_pool / hope_env_set_dlp_cases / hope_env_set_draw_class)");
    return HOPE_OK;
}

static int launch_step(hope_env_t* h, const void* actions, const uint8_t* active, uint32_t stages,
                       const hope_step_out* out, void* stream, int has_action) {
    if (!h || !out) return fail(HOPE_EINVAL, "hope_env_step: null argument");
    if (!is_live(h)) return fail(HOPE_EINVAL, "hope_env_step: not a live handle (destroyed?)");
    if (!h->have_tables) return fail(HOPE_ESTATE, "hope_env_step: hope_env_upload_tables has not been called");
    if (!h->have_scenes) return fail(HOPE_ESTATE, "hope_env_step: hope_env_set_scenes has not been called");
    if (has_action && !actions) return fail(HOPE_EINVAL, "hope_env_step: actions is null");
    if (stages & HOPE_STAGE_RS) stages |= HOPE_STAGE_REWARD;       // the RS gate needs the status
    if (stages & HOPE_AUTO_REDRAW) {
        if (!(stages & HOPE_AUTO_RESET)) return fail(HOPE_EINVAL, "hope_env_step: HOPE_AUTO_REDRAW needs HOPE_AUTO_RESET");
        if (h->pool_n <= 0 && h->dlp.n_cases <= 0) return fail(HOPE_ESTATE, "hope_env_step: HOPE_AUTO_REDRAW without a scene pool (hope_env_set_pool / hope_env_set_dlp_cases)");
        int rc0 = check_pool_classes(h, "hope_env_step");
        if (rc0 != HOPE_OK) return rc0;
    }
    if (stages & HOPE_STAGE_IMG) {
        if (!h->traj) return fail(HOPE_ESTATE, "hope_env_step: HOPE_STAGE_IMG needs a handle created with HOPE_F_IMAGE");
        if (!out->img) return fail(HOPE_EINVAL, "hope_env_step: HOPE_STAGE_IMG without out->img");
    }
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    const bool prof = h->flags & HOPE_F_PROFILE;
    h->step_seq++;
    { int rcp = apply_pending_pool(h, false); if (rcp != HOPE_OK) return rcp; }    // a relaxed commit whose upload has finished takes over here
    if (h->pool_wait_pending) {                             // a pool committed since the last launch: its upload orders before this step
        HIPCHK(hipStreamWaitEvent(s, h->ev_pool_ready, 0));
        h->pool_wait_pending = false;
    }
    { int rcc = sync_cold(h, s); if (rcc != HOPE_OK) return rcc; }
    struct LastStep {                                       // (the next pool upload must not overwrite a set this step still reads)
        hope_env_t* h; hipStream_t s;
        ~LastStep() { if (h->pactive >= 0 && h->ev_last_step && !h->last_via_steps) hipEventRecord(h->ev_last_step, s); }
    } last_step{h, s};
    if (prof && h->pending.size() > 4096) { int rc = drain_events(h); if (rc) return rc; }
    EventTimer timer(h);
    LaunchTimer* tm = prof ? &timer : nullptr;
    int rc = enqueue_step(h, actions, active, stages, out, s, (h->flags & HOPE_F_OVERLAP) != 0, has_action, tm);
    if (rc != HOPE_OK) return rc;
    if (timer.failed) return fail(HOPE_EHIP, "hipEventRecord failed");
    return HOPE_OK;
}


// Shared by set_scenes and set_pool: stage the host arrays and write constants / tiles for entries ids[0..n)
static int upload_scenes(hope_env_t* h, const int32_t* ids, int n, const double* start, const double* dest, const double* bbox,
                         const double* verts, const int32_t* n_obst, double* d_scene_c, double* d_state, int32_t* d_t,
                         int32_t* d_nobst, double* d_verts, float4* d_obb, double* d_traj, int32_t* d_traj_len, int32_t* d_traj_valid,
                         int32_t* d_layer_valid) {
    size_t tile = (size_t)h->max_obst * 8 * sizeof(double);
    size_t o_ids = 0, o_nob = o_ids + sizeof(int32_t) * n, o_start = (o_nob + sizeof(int32_t) * n + 15) & ~(size_t)15;
    size_t o_dest = o_start + 24 * (size_t)n, o_bbox = o_dest + 24 * (size_t)n, o_verts = (o_bbox + 32 * (size_t)n + 15) & ~(size_t)15;
    size_t need = o_verts + tile * n;
    if (need > h->stage_bytes) {
        if (h->stage) hipFree(h->stage);
        h->stage = nullptr; h->stage_bytes = 0;
        hipError_t e2 = hipMalloc(&h->stage, need);
        if (e2 != hipSuccess) return fail(HOPE_ENOMEM, std::string("hipMalloc stage: ") + hipGetErrorString(e2));
        h->stage_bytes = need;
    }
    char* sp = (char*)h->stage;
    HIPCHK(hipMemcpy(sp + o_ids, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp + o_nob, n_obst, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp + o_start, start, 24 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp + o_dest, dest, 24 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp + o_bbox, bbox, 32 * (size_t)n, hipMemcpyHostToDevice));
    if (verts) HIPCHK(hipMemcpy(sp + o_verts, verts, tile * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_scene_consts, dim3((n + 127) / 128), dim3(128), 0, 0, n, (const int32_t*)(sp + o_ids),
                       (const double*)(sp + o_start), (const double*)(sp + o_dest), (const double*)(sp + o_bbox),
                       (const int32_t*)(sp + o_nob), d_scene_c, d_state, d_t, d_nobst, d_traj, d_traj_len, d_traj_valid, d_layer_valid);
    if (verts)
        hipLaunchKernelGGL(k_set_scene_tiles, dim3(n), dim3(128), 0, 0, (const int32_t*)(sp + o_ids),
                           (const int32_t*)(sp + o_nob), (const double*)(sp + o_verts), (const double*)(sp + o_bbox), d_verts, d_obb,
                           h->fverts, h->fbox, h->eflag, h->max_obst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return HOPE_OK;
}

// ---- scene pool: pinned staging -> asynchronous upload into the set the kernels are not reading -> swap by stream order ----
// pool identity (hope_env_pool_generation): a hash chain over what was uploaded, so that equal histories give equal values across
// processes and different pools differ (a counter said "1" for any first pool)
static uint64_t hmix64(uint64_t z) {                        // splitmix64 finaliser (host twin of hope_dev.h's mix64)
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t fold64(uint64_t a, uint64_t b) { return hmix64(a ^ hmix64(b + 0x9E3779B97F4A7C15ull)); }
// (one multiply per 8 bytes: hope_env_commit_pool hashes ~0.8 MB on the thread that enqueues the steps -- ~0.1 ms per commit; the
// splitmix chain above cost 0.7 ms, more than a 65 536-scene step)
static uint64_t hash_bytes(uint64_t hsh, const void* p, size_t bytes) {
    const unsigned char* q = (const unsigned char*)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, q + i, 8); hsh = (hsh ^ w) * 0x100000001B3ull; hsh ^= hsh >> 29; }
    uint64_t w = 0;
    if (i < bytes) { memcpy(&w, q + i, bytes - i); hsh = (hsh ^ w) * 0x100000001B3ull; }
    return fold64(hsh, bytes);
}
static void bump_pool_generation(hope_env_t* h, uint64_t content) {
    h->pool_generation = fold64(h->pool_generation, content);
    if (h->pool_generation == 0) h->pool_generation = 1;   // 0 is reserved: "no pool" / "skip the check"
}

static int pool_init_streams(hope_env_t* h) {
    if (h->pool_stream) return HOPE_OK;
    HIPCHK(hipStreamCreateWithFlags(&h->pool_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_pool_ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_pool_copied, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_last_step, hipEventDisableTiming));
    return HOPE_OK;
}

// draw lists of the two size classes into pinned staging: complete pool scenes by their obstacle count, and -- in the large class
// -- the device-drawn Dragon-Lake cases as -2 - case (-1 is "the map hope_env_set_scenes uploaded")
static void build_pool_lists(hope_env_t* h, const int32_t* n_obst, int n_pool, std::vector<int32_t>& l0, std::vector<int32_t>& l1) {
    const bool two = h->max_obst > SMALL_TILE;
    for (int k = 0; k < n_pool; k++) (two && n_obst[k] > SMALL_TILE ? l1 : l0).push_back(k);
    for (int c = 0; c < h->dlp.n_cases; c++) (two ? l1 : l0).push_back(-2 - c);
}

static int pool_set_reserve(hope_env_t* h, hope_env::PoolSet& ps, int n_pool, int n_list) {
    if (n_pool > ps.cap) {
        for (void* q : {(void*)ps.verts, (void*)ps.c, (void*)ps.nobst}) if (q) hipFree(q);
        ps.verts = ps.c = nullptr; ps.nobst = nullptr; ps.cap = 0;
        const size_t P = (size_t)n_pool;
        hipError_t e_ = hipMalloc((void**)&ps.verts, P * h->max_obst * 8 * sizeof(double));
        if (e_ == hipSuccess) e_ = hipMalloc((void**)&ps.c, P * SC_WORDS * sizeof(double));
        if (e_ == hipSuccess) e_ = hipMalloc((void**)&ps.nobst, P * sizeof(int32_t));
        if (e_ != hipSuccess) return fail(HOPE_ENOMEM, std::string("hipMalloc pool: ") + hipGetErrorString(e_));
        ps.cap = n_pool;
    }
    if (n_list > ps.list_cap) {
        for (int c = 0; c < 2; c++) { if (ps.list[c]) hipFree(ps.list[c]); ps.list[c] = nullptr; }
        for (int c = 0; c < 2; c++) {
            hipError_t e_ = hipMalloc((void**)&ps.list[c], (size_t)n_list * sizeof(int32_t));
            if (e_ != hipSuccess) return fail(HOPE_ENOMEM, std::string("hipMalloc pool lists: ") + hipGetErrorString(e_));
        }
        ps.list_cap = n_list;
    }
    return HOPE_OK;
}

int hope_env_pool_staging(hope_env_t* h, int n_pool, double** start, double** dest, double** bbox, double** verts, int32_t** n_obst) {
    if (!h || n_pool <= 0 || !start || !dest || !bbox || !verts || !n_obst) return fail(HOPE_EINVAL, "hope_env_pool_staging: bad argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    int rc = pool_init_streams(h);
    if (rc != HOPE_OK) return rc;
    if (h->pstage.busy) { HIPCHK(hipEventSynchronize(h->ev_pool_copied)); h->pstage.busy = false; }   // the previous upload still reads it
    if (n_pool > h->pstage.cap) {
        for (void* q : {(void*)h->pstage.start, (void*)h->pstage.dest, (void*)h->pstage.bbox, (void*)h->pstage.verts, (void*)h->pstage.nobst, (void*)h->pstage.list})
            if (q) hipHostFree(q);
        h->pstage = hope_env::PoolStage{};
        const size_t P = (size_t)n_pool;
        hipError_t e_ = hipHostMalloc((void**)&h->pstage.start, P * 24);
        if (e_ == hipSuccess) e_ = hipHostMalloc((void**)&h->pstage.dest, P * 24);
        if (e_ == hipSuccess) e_ = hipHostMalloc((void**)&h->pstage.bbox, P * 32);
        if (e_ == hipSuccess) e_ = hipHostMalloc((void**)&h->pstage.verts, P * h->max_obst * 8 * sizeof(double));
        if (e_ == hipSuccess) e_ = hipHostMalloc((void**)&h->pstage.nobst, P * sizeof(int32_t));
        if (e_ == hipSuccess) e_ = hipHostMalloc((void**)&h->pstage.list, (P + 4096) * sizeof(int32_t));
        if (e_ != hipSuccess) return fail(HOPE_ENOMEM, std::string("hipHostMalloc pool staging: ") + hipGetErrorString(e_));
        h->pstage.cap = n_pool;
    }
    *start = h->pstage.start; *dest = h->pstage.dest; *bbox = h->pstage.bbox; *verts = h->pstage.verts; *n_obst = h->pstage.nobst;
    return HOPE_OK;
}

int hope_env_pool_staging_ready(hope_env_t* h) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_pool_staging_ready: null handle");
    if (!h->pstage.busy) return 1;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    const hipError_t e = hipEventQuery(h->ev_pool_copied);
    if (e == hipSuccess) { h->pstage.busy = false; return 1; }
    if (e == hipErrorNotReady) return 0;
    return fail(HOPE_EHIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
}

int hope_env_pool_generation(hope_env_t* h, uint64_t* generation) {
    if (!h || !generation) return fail(HOPE_EINVAL, "hope_env_pool_generation: null argument");
    { DeviceGuard guard(h->device); int rcp = apply_pending_pool(h, true); if (rcp != HOPE_OK) return rcp; }
    *generation = h->pool_generation;
    return HOPE_OK;
}

// the swap of hope_env_commit_pool_relaxed, once its upload is complete (wait: block until it is)
static int apply_pending_pool(hope_env_t* h, bool wait) {
    if (!h->pend.on) return HOPE_OK;
    if (wait) HIPCHK(hipEventSynchronize(h->ev_pool_ready));
    else {
        const hipError_t e = hipEventQuery(h->ev_pool_ready);
        if (e == hipErrorNotReady) return HOPE_OK;
        if (e != hipSuccess) return fail(HOPE_EHIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    }
    hope_env::PoolSet& ps = h->pset[h->pend.set];
    h->pool_verts = ps.verts; h->pool_c = ps.c; h->pool_nobst = ps.nobst;
    h->pool_cls[0] = ps.list[0]; h->pool_cls[1] = ps.list[1];
    h->pool_cls_n[0] = h->pend.cls_n[0]; h->pool_cls_n[1] = h->pend.cls_n[1];
    h->pool_n = h->pend.n_pool;
    h->pool_nobst_host.swap(h->pend.nobst_host);
    h->pactive = h->pend.set;
    h->pend.on = false;
    bump_pool_generation(h, h->pend.content);
    return HOPE_OK;
}

static int commit_pool_impl(hope_env_t* h, int n_pool, bool relaxed);
int hope_env_commit_pool(hope_env_t* h, int n_pool, void* stream) { (void)stream; return commit_pool_impl(h, n_pool, false); }
int hope_env_commit_pool_relaxed(hope_env_t* h, int n_pool) { return commit_pool_impl(h, n_pool, true); }

static int commit_pool_impl(hope_env_t* h, int n_pool, bool relaxed) {
    // (no join of an unjoined Reeds-Shepp chain here: only the motion launches read the pool, and ev_last_step covers them)
    if (!h || n_pool <= 0) return fail(HOPE_EINVAL, "hope_env_commit_pool: bad argument");
    if (n_pool > h->pstage.cap) return fail(HOPE_ESTATE, "hope_env_commit_pool: more entries than hope_env_pool_staging provided");
    for (int k = 0; k < n_pool; k++)
        if (h->pstage.nobst[k] < 0 || h->pstage.nobst[k] > h->max_obst) return fail(HOPE_EINVAL, "hope_env_commit_pool: n_obst exceeds max_obstacles");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    std::vector<int32_t> l0, l1;
    build_pool_lists(h, h->pstage.nobst, n_pool, l0, l1);
    if ((int)(l0.size() + l1.size()) > h->pstage.cap + 4096) return fail(HOPE_EINVAL, "hope_env_commit_pool: too many Dragon-Lake cases for the list staging");
    { int rcp = apply_pending_pool(h, true); if (rcp != HOPE_OK) return rcp; }     // (a relaxed swap still in flight: it comes first)
    const int t = h->pactive < 0 ? 0 : 1 - h->pactive;
    hope_env::PoolSet& ps = h->pset[t];
    // two commits without a hope_env_pool_staging call in between: the previous upload may still be reading the pinned arrays
    if (h->pstage.busy) { HIPCHK(hipEventSynchronize(h->ev_pool_copied)); h->pstage.busy = false; }
    int rc = pool_set_reserve(h, ps, n_pool, (int)std::max(l0.size(), l1.size()) + 1);
    if (rc != HOPE_OK) return rc;
    if (3 * 32 * n_pool / 8 > h->pstage_dev_cap) {           // start | dest | bbox: 24 + 24 + 32 bytes per entry
        if (h->pstage_dev) hipFree(h->pstage_dev);
        h->pstage_dev = nullptr; h->pstage_dev_cap = 0;
        hipError_t e_ = hipMalloc((void**)&h->pstage_dev, (size_t)n_pool * 80);
        if (e_ != hipSuccess) return fail(HOPE_ENOMEM, std::string("hipMalloc pool staging: ") + hipGetErrorString(e_));
        h->pstage_dev_cap = 12 * n_pool;
    }
    hipStream_t us = h->pool_stream;
    // the target set was the active one until the previous swap: steps enqueued before that swap may still be reading it
    if (h->pactive >= 0) {
        HIPCHK(hipStreamWaitEvent(us, h->ev_last_step, 0));
        if (h->last_via_steps) for (int i = 0; i < 2; i++) HIPCHK(hipStreamWaitEvent(us, h->ev_step[i], 0));   // (the motion launches of the last step)
    }
    const size_t P = (size_t)n_pool;
    char* dv = (char*)h->pstage_dev;
    {   // only the used prefix of every entry's obstacle tile travels (a 2-D copy: one row per entry): generated lots hold ~7 obstacles of
        // the 128 slots, and the full 67 MB of an 8 192-entry pool took ~1 ms on the hardware queue the upload stream shares with one of
        // the step's streams -- the step enqueued right after a commit took 1.52 ms instead of 0.49, now 0.65 (tools/commit_cost.py; the
        // excess scaled with the pool size; moving k_set_scene_consts off the upload stream did not change it).  The tail of a row
        // keeps the previous generation's bytes; nothing reads beyond n_obst.
        int maxn = 1;
        for (int k = 0; k < n_pool; k++) maxn = std::max(maxn, (int)h->pstage.nobst[k]);
        const size_t pitch = (size_t)h->max_obst * 8 * sizeof(double), width = (size_t)maxn * 8 * sizeof(double);
        if (2 * width <= pitch) HIPCHK(hipMemcpy2DAsync(ps.verts, pitch, h->pstage.verts, pitch, width, P, hipMemcpyHostToDevice, us));
        else HIPCHK(hipMemcpyAsync(ps.verts, h->pstage.verts, P * pitch, hipMemcpyHostToDevice, us));
    }
    HIPCHK(hipMemcpyAsync(ps.nobst, h->pstage.nobst, P * sizeof(int32_t), hipMemcpyHostToDevice, us));
    HIPCHK(hipMemcpyAsync(dv, h->pstage.start, P * 24, hipMemcpyHostToDevice, us));
    HIPCHK(hipMemcpyAsync(dv + P * 24, h->pstage.dest, P * 24, hipMemcpyHostToDevice, us));
    HIPCHK(hipMemcpyAsync(dv + P * 48, h->pstage.bbox, P * 32, hipMemcpyHostToDevice, us));
    memcpy(h->pstage.list, l0.data(), l0.size() * sizeof(int32_t));
    memcpy(h->pstage.list + l0.size(), l1.data(), l1.size() * sizeof(int32_t));
    if (!l0.empty()) HIPCHK(hipMemcpyAsync(ps.list[0], h->pstage.list, l0.size() * sizeof(int32_t), hipMemcpyHostToDevice, us));
    if (!l1.empty()) HIPCHK(hipMemcpyAsync(ps.list[1], h->pstage.list + l0.size(), l1.size() * sizeof(int32_t), hipMemcpyHostToDevice, us));
    HIPCHK(hipEventRecord(h->ev_pool_copied, us));          // the pinned staging may be refilled once this has passed
    h->pstage.busy = true;
    hipLaunchKernelGGL(k_set_scene_consts, dim3((n_pool + 127) / 128), dim3(128), 0, us, n_pool, (const int32_t*)nullptr,
                       (const double*)dv, (const double*)(dv + P * 24), (const double*)(dv + P * 48), (const int32_t*)nullptr, ps.c,
                       (double*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev_pool_ready, us));
    uint64_t content;
    {   // identity of the new pool: every scalar of every entry and two vertex words of each
        uint64_t c = hash_bytes(0x706f6f6cull, h->pstage.start, P * 24);
        c = hash_bytes(c, h->pstage.dest, P * 24);
        c = hash_bytes(c, h->pstage.bbox, P * 32);
        c = hash_bytes(c, h->pstage.nobst, P * sizeof(int32_t));
        const size_t tile_words = (size_t)h->max_obst * 8;
        for (size_t k = 0; k < P; k++) {                     // two vertex words of every entry (the whole tile array is 67 MB per 8 192 lots)
            const size_t nw = (size_t)h->pstage.nobst[k] * 8;
            if (nw == 0) continue;
            uint64_t b0, b1;
            memcpy(&b0, h->pstage.verts + k * tile_words + (k * 7) % nw, 8);
            memcpy(&b1, h->pstage.verts + k * tile_words + nw - 1, 8);
            c = (c ^ b0) * 0x100000001B3ull; c = (c ^ b1) * 0x100000001B3ull; c ^= c >> 29;
        }
        content = c;
    }
    if (relaxed && h->pactive >= 0) {
        // the swap waits for the upload, not the steps for the swap: apply_pending_pool (launch_step / hope_env_redraw / the next commit)
        h->pend.on = true; h->pend.set = t; h->pend.n_pool = n_pool;
        h->pend.cls_n[0] = (int)l0.size(); h->pend.cls_n[1] = (int)l1.size(); h->pend.content = content;
        h->pend.nobst_host.assign(h->pstage.nobst, h->pstage.nobst + n_pool);
        return HOPE_OK;
    }
    // swap: every launch enqueued from now on reads the new set, after waiting (on its own stream) for the upload
    h->pool_verts = ps.verts; h->pool_c = ps.c; h->pool_nobst = ps.nobst;
    h->pool_cls[0] = ps.list[0]; h->pool_cls[1] = ps.list[1];
    h->pool_cls_n[0] = (int)l0.size(); h->pool_cls_n[1] = (int)l1.size();
    h->pool_n = n_pool;
    h->pool_nobst_host.assign(h->pstage.nobst, h->pstage.nobst + n_pool);
    h->pactive = t;
    h->pool_wait_pending = true;
    bump_pool_generation(h, content);
    return HOPE_OK;
}

int hope_env_set_pool(hope_env_t* h, int n_pool, const double* start, const double* dest, const double* bbox,
                      const double* verts, const int32_t* n_obst) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || n_pool <= 0 || !start || !dest || !bbox || !verts || !n_obst) return fail(HOPE_EINVAL, "hope_env_set_pool: bad argument");
    for (int k = 0; k < n_pool; k++)
        if (n_obst[k] < 0 || n_obst[k] > h->max_obst) return fail(HOPE_EINVAL, "hope_env_set_pool: n_obst exceeds max_obstacles");
    double *ps, *pd, *pb, *pv;
    int32_t* pn;
    int rc = hope_env_pool_staging(h, n_pool, &ps, &pd, &pb, &pv, &pn);
    if (rc != HOPE_OK) return rc;
    const size_t P = (size_t)n_pool;
    memcpy(ps, start, P * 24); memcpy(pd, dest, P * 24); memcpy(pb, bbox, P * 32);
    memcpy(pv, verts, P * h->max_obst * 8 * sizeof(double)); memcpy(pn, n_obst, P * sizeof(int32_t));
    rc = hope_env_commit_pool(h, n_pool, nullptr);
    if (rc != HOPE_OK) return rc;
    DeviceGuard guard(h->device);
    HIPCHK(hipStreamSynchronize(h->pool_stream));           // the synchronous form: the caller's arrays are free on return anyway
    return HOPE_OK;
}

// the class lists of the ACTIVE set after the Dragon-Lake cases changed (rare; host-synchronous)
static int refresh_pool_lists_sync(hope_env_t* h) {
    int rc = pool_init_streams(h);
    if (rc != HOPE_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    std::vector<int32_t> l0, l1;
    build_pool_lists(h, h->pool_nobst_host.data(), h->pool_n, l0, l1);
    if (h->pactive < 0) h->pactive = 0;
    hope_env::PoolSet& ps = h->pset[h->pactive];
    rc = pool_set_reserve(h, ps, std::max(h->pool_n, ps.cap), (int)std::max(l0.size(), l1.size()) + 1);
    if (rc != HOPE_OK) return rc;
    if (!l0.empty()) HIPCHK(hipMemcpy(ps.list[0], l0.data(), l0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!l1.empty()) HIPCHK(hipMemcpy(ps.list[1], l1.data(), l1.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->pool_cls[0] = ps.list[0]; h->pool_cls[1] = ps.list[1];
    h->pool_cls_n[0] = (int)l0.size(); h->pool_cls_n[1] = (int)l1.size();
    return HOPE_OK;
}

int hope_env_set_dlp_cases(hope_env_t* h, int n_cases, const double* dest, const int32_t* cand_off, const double* cand,
                           const int32_t* case_set, int n_sets, const int32_t* set_off, const double* set_verts) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || n_cases < 0 || (n_cases > 0 && (!dest || !cand_off || !cand || !case_set || n_sets <= 0 || !set_off || !set_verts)))
        return fail(HOPE_EINVAL, "hope_env_set_dlp_cases: bad argument");
    for (int c = 0; c < n_cases; c++) {
        if (cand_off[c + 1] <= cand_off[c]) return fail(HOPE_EINVAL, "hope_env_set_dlp_cases: a case without start candidates");
        if (case_set[c] < 0 || case_set[c] >= n_sets) return fail(HOPE_EINVAL, "hope_env_set_dlp_cases: case_set out of range");
    }
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    // a relaxed commit still pending takes over FIRST: its class lists were built with the old cases, and refresh_pool_lists_sync
    // below rebuilds the lists of the set that is active when it runs
    { int rcp = apply_pending_pool(h, true); if (rcp != HOPE_OK) return rcp; }
    HIPCHK(hipDeviceSynchronize());
    for (void*& q : h->dlp_mem) { if (q) hipFree(q); q = nullptr; }
    h->dlp = DlpCases{};
    if (n_cases > 0) {
        const size_t nc = (size_t)cand_off[n_cases], nv = (size_t)set_off[n_sets];
        const void* src[6] = {dest, cand_off, cand, case_set, set_off, set_verts};
        const size_t bytes[6] = {sizeof(double) * 3 * n_cases, sizeof(int32_t) * (n_cases + 1), sizeof(double) * 3 * nc,
                                 sizeof(int32_t) * n_cases, sizeof(int32_t) * (n_sets + 1), sizeof(double) * 8 * nv};
        for (int i = 0; i < 6; i++) {
            hipError_t e_ = hipMalloc(&h->dlp_mem[i], bytes[i]);
            if (e_ != hipSuccess) return fail(HOPE_ENOMEM, std::string("hipMalloc dlp cases: ") + hipGetErrorString(e_));
            HIPCHK(hipMemcpy(h->dlp_mem[i], src[i], bytes[i], hipMemcpyHostToDevice));
        }
        h->dlp.n_cases = n_cases;
        h->dlp.dest = (const double*)h->dlp_mem[0]; h->dlp.cand_off = (const int32_t*)h->dlp_mem[1];
        h->dlp.cand = (const double*)h->dlp_mem[2]; h->dlp.case_set = (const int32_t*)h->dlp_mem[3];
        h->dlp.set_off = (const int32_t*)h->dlp_mem[4]; h->dlp.set_verts = (const double*)h->dlp_mem[5];
        uint64_t c = 0x646c70ull;
        for (int i = 0; i < 6; i++) c = hash_bytes(c, src[i], bytes[i]);
        bump_pool_generation(h, c);
    } else bump_pool_generation(h, 0x6e6f646c70ull);        // the cases were removed
    return refresh_pool_lists_sync(h);
}

int hope_env_set_draw_class(hope_env_t* h, const int32_t* scene_ids, int n, const uint8_t* cls) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || n < 0 || (n > 0 && (!scene_ids || !cls))) return fail(HOPE_EINVAL, "hope_env_set_draw_class: null argument");
    if (!h->have_scenes) return fail(HOPE_ESTATE, "hope_env_set_draw_class: hope_env_set_scenes has not been called");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    // the obstacle counts as they are NOW: device-side draws (HOPE_AUTO_REDRAW, hope_env_redraw) change them behind the host copy
    HIPCHK(hipMemcpy(h->n_obst_host.data(), h->n_obst, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) {
        if (scene_ids[k] < 0 || scene_ids[k] >= h->n) return fail(HOPE_EINVAL, "hope_env_set_draw_class: scene id out of range");
        if (!cls[k] && h->n_obst_host[scene_ids[k]] > SMALL_TILE)
            return fail(HOPE_EINVAL, "hope_env_set_draw_class: a scene with more than 32 obstacles cannot join the small class");
    }
    for (int k = 0; k < n; k++) h->slot_cls_host[scene_ids[k]] = (cls[k] && h->max_obst > SMALL_TILE) ? 1 : 0;
    int rc = rebuild_class_lists(h);
    if (rc != HOPE_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    return HOPE_OK;
}

int hope_env_pool_overflow(hope_env_t* h, int32_t* count) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || !count) return fail(HOPE_EINVAL, "hope_env_pool_overflow: null argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(count, h->pool_overflow, sizeof(int32_t), hipMemcpyDeviceToHost));
    return HOPE_OK;
}

// the maps the scenes hold NOW (after device-side draws they exist on the device only): host-synchronous
int hope_env_download_scenes(hope_env_t* h, const int32_t* scene_ids, int n, double* start, double* dest, double* bbox, double* verts,
                             int32_t* n_obst) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || n < 0 || (n > 0 && !scene_ids)) return fail(HOPE_EINVAL, "hope_env_download_scenes: bad argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    const size_t tile = (size_t)h->max_obst * 8;
    std::vector<double> c(SC_WORDS);
    for (int k = 0; k < n; k++) {
        const int s_ = scene_ids[k];
        if (s_ < 0 || s_ >= h->n) return fail(HOPE_EINVAL, "hope_env_download_scenes: scene id out of range");
        HIPCHK(hipMemcpy(c.data(), h->scene_c + (size_t)s_ * SC_WORDS, SC_WORDS * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; i++) { if (start) start[3 * (size_t)k + i] = c[SC_START + i]; if (dest) dest[3 * (size_t)k + i] = c[SC_DEST + i]; }
        for (int i = 0; i < 4; i++) if (bbox) bbox[4 * (size_t)k + i] = c[SC_BBOX + i];
        if (verts) HIPCHK(hipMemcpy(verts + (size_t)k * tile, h->verts + (size_t)s_ * tile, tile * sizeof(double), hipMemcpyDeviceToHost));
        if (n_obst) HIPCHK(hipMemcpy(n_obst + k, h->n_obst + s_, sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return HOPE_OK;
}

int hope_env_set_redraw_seed(hope_env_t* h, uint64_t seed) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_set_redraw_seed: null handle");
    h->redraw_seed = seed;
    return HOPE_OK;
}

int hope_env_redraw(hope_env_t* h, const uint8_t* mask, uint64_t seed, void* stream) {
    if (h) { DeviceGuard g0(h->device); int rcs = join_rs(h, (hipStream_t)stream); if (rcs != HOPE_OK) return rcs; }
    if (!h || !mask) return fail(HOPE_EINVAL, "hope_env_redraw: null argument");
    if (!h->have_scenes) return fail(HOPE_ESTATE, "hope_env_redraw: hope_env_set_scenes has not been called");
    if (h->pool_n <= 0 && h->dlp.n_cases <= 0) return fail(HOPE_ESTATE, "hope_env_redraw: no scene pool (hope_env_set_pool / hope_env_set_dlp_cases)");
    int rc0 = check_pool_classes(h, "hope_env_redraw");
    if (rc0 != HOPE_OK) return rc0;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    { int rcp = apply_pending_pool(h, false); if (rcp != HOPE_OK) return rcp; }
    if (h->pool_wait_pending) { HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->ev_pool_ready, 0)); h->pool_wait_pending = false; }
    hipLaunchKernelGGL(k_redraw, dim3(h->n), dim3(WAVE), 0, (hipStream_t)stream, h->max_obst, mask, seed, h->pool_cls[0],
                       h->pool_cls_n[0], h->pool_cls[1], h->pool_cls_n[1], h->pool_verts, h->pool_c, h->pool_nobst, h->verts,
                       h->scene_c, h->n_obst, h->state, h->tstep, h->traj, h->traj_len, h->traj_valid, h->cur_pool, h->episode, h->obb,
                       h->dlp, h->pool_overflow, h->slot_cls, h->layer_valid, h->fverts, h->fbox, h->eflag);
    HIPCHK(hipGetLastError());
    if (h->pactive >= 0 && h->ev_last_step) HIPCHK(hipEventRecord(h->ev_last_step, (hipStream_t)stream));
    return HOPE_OK;
}

// Snapshot / restore of the map each scene holds after device-side draws.  A draw is a pure function of (seed, scene, episode
// counter) and of the pool / case lists, so a map is restored by repeating its draw: the counters go back by one and k_redraw runs
// for the scenes that held a drawn map.  Pose / t / accumulator are restored separately (hope_env_upload_state, afterwards).
int hope_env_download_pool_state(hope_env_t* h, int32_t* pool_index, uint32_t* episode) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h) return fail(HOPE_EINVAL, "hope_env_download_pool_state: null handle");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    { int rcp = apply_pending_pool(h, true); if (rcp != HOPE_OK) return rcp; }     // the snapshot belongs to the pool the next step draws from
    HIPCHK(hipDeviceSynchronize());
    if (pool_index) HIPCHK(hipMemcpy(pool_index, h->cur_pool, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (episode) HIPCHK(hipMemcpy(episode, h->episode, (size_t)h->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HOPE_OK;
}

int hope_env_restore_maps(hope_env_t* h, const uint8_t* drawn /* host [N]: the scene held a drawn map */, const uint32_t* episode, uint64_t seed,
                          uint64_t pool_generation) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || !drawn || !episode) return fail(HOPE_EINVAL, "hope_env_restore_maps: null argument");
    { DeviceGuard guard0(h->device); int rcp = apply_pending_pool(h, true); if (rcp != HOPE_OK) return rcp; }   // (compare generations after a pending swap)
    if (h->pool_n <= 0 && h->dlp.n_cases <= 0) return fail(HOPE_ESTATE, "hope_env_restore_maps: no scene pool");
    if (pool_generation != 0 && pool_generation != h->pool_generation)
        return fail(HOPE_ESTATE, "hope_env_restore_maps: the scene pool has been replaced since the snapshot (pool generation " +
                                 std::to_string(h->pool_generation) + " now, " + std::to_string(pool_generation) + " in the snapshot): repeating the draws "
                                 "would give other maps -- snapshot refreshed runs with hope_env_download_scenes / hope_env_set_scenes");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint32_t> ep(episode, episode + h->n);
    for (int i = 0; i < h->n; i++) if (drawn[i]) { if (ep[i] == 0) return fail(HOPE_EINVAL, "hope_env_restore_maps: a drawn scene with episode counter 0"); ep[i] -= 1; }
    HIPCHK(hipMemcpy(h->episode, ep.data(), (size_t)h->n * sizeof(uint32_t), hipMemcpyHostToDevice));
    uint8_t* dmask = nullptr;
    HIPCHK(hipMalloc((void**)&dmask, (size_t)h->n));
    hipError_t e = hipMemcpy(dmask, drawn, (size_t)h->n, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? hope_env_redraw(h, dmask, seed, nullptr) : fail(HOPE_EHIP, "hipMemcpy mask");
    hipDeviceSynchronize();
    hipFree(dmask);
    return rc;
}

int hope_env_download_pool_index(hope_env_t* h, int32_t* out) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || !out) return fail(HOPE_EINVAL, "hope_env_download_pool_index: null argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, h->cur_pool, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return HOPE_OK;
}

int hope_env_restart(hope_env_t* h, const uint8_t* mask, void* stream) {
    if (h) { DeviceGuard g0(h->device); int rcs = join_rs(h, (hipStream_t)stream); if (rcs != HOPE_OK) return rcs; }
    if (!h || !mask) return fail(HOPE_EINVAL, "hope_env_restart: null argument");
    if (!h->have_scenes) return fail(HOPE_ESTATE, "hope_env_restart: hope_env_set_scenes has not been called");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(k_restart, dim3((h->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->n, mask, h->scene_c,
                       h->state, h->tstep, h->traj, h->traj_len, h->traj_valid);
    HIPCHK(hipGetLastError());
    return HOPE_OK;
}

int hope_env_kernel_ms(hope_env_t* h, double* ms, int64_t* launches, int reset) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_kernel_ms: null handle");
    if (!(h->flags & HOPE_F_PROFILE)) return fail(HOPE_ESTATE, "hope_env_kernel_ms: handle was not created with HOPE_F_PROFILE");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    int rc = drain_events(h);
    if (rc) return rc;
    for (int k = 0; k < HOPE_N_KERNELS; k++) {
        if (ms) ms[k] = h->ms[k];
        if (launches) launches[k] = h->launches[k];
        if (reset) { h->ms[k] = 0; h->launches[k] = 0; }
    }
    return HOPE_OK;
}

int hope_env_kernel_union_ms(hope_env_t* h, double* ms, int64_t* calls, int reset) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_kernel_union_ms: null handle");
    if (!(h->flags & HOPE_F_PROFILE)) return fail(HOPE_ESTATE, "hope_env_kernel_union_ms: handle was not created with HOPE_F_PROFILE");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    int rc = drain_events(h);
    if (rc) return rc;
    for (int k = 0; k < HOPE_N_KERNELS; k++) {
        if (ms) ms[k] = h->union_ms[k];
        if (calls) calls[k] = h->union_calls[k];
        if (reset) { h->union_ms[k] = 0; h->union_calls[k] = 0; }
    }
    return HOPE_OK;
}

int hope_env_profile_kernels(hope_env_t* h, uint32_t kernel_mask) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_profile_kernels: null handle");
    if (!(h->flags & HOPE_F_PROFILE)) return fail(HOPE_ESTATE, "hope_env_profile_kernels: handle was not created with HOPE_F_PROFILE");
    h->profile_mask = kernel_mask;
    return HOPE_OK;
}

int hope_env_step(hope_env_t* h, const void* actions, const uint8_t* active, uint32_t stages,
                  const hope_step_out* out, void* stream) {
    return launch_step(h, actions, active, stages, out, stream, 1);
}

int hope_env_wait_rs(hope_env_t* h, void* stream) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_wait_rs: null handle");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    return join_rs(h, (hipStream_t)stream);
}

int hope_env_last_step(hope_env_t* h, uint64_t* step) {
    if (!h || !step) return fail(HOPE_EINVAL, "hope_env_last_step: null argument");
    *step = h->step_seq;
    return HOPE_OK;
}

int hope_env_wait_rs_step(hope_env_t* h, uint64_t step, void* stream) {
    if (!h) return fail(HOPE_EINVAL, "hope_env_wait_rs_step: null handle");
    if (step == 0 || step > h->step_seq) return fail(HOPE_EINVAL, "hope_env_wait_rs_step: no such step (hope_env_last_step)");
    if (step != h->step_seq)
        return fail(HOPE_ESTATE, "hope_env_wait_rs_step: step " + std::to_string(step) + " is not the last one (" + std::to_string(h->step_seq) +
                                 "): a later hope_env_step / hope_env_reset_obs has replaced its Reeds-Shepp outputs -- wait before enqueuing the next step");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    return join_rs(h, (hipStream_t)stream);
}

int hope_env_download_n_obst(hope_env_t* h, int32_t* n_obst) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h || !n_obst) return fail(HOPE_EINVAL, "hope_env_download_n_obst: null argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(n_obst, h->n_obst, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return HOPE_OK;
}

int hope_env_reset_obs(hope_env_t* h, const uint8_t* active, uint32_t stages, const hope_step_out* out, void* stream) {
    return launch_step(h, nullptr, active, stages & ~HOPE_STAGE_MOTION, out, stream, 0);
}

int hope_debug_math(int fn, int n, const double* a, const double* b, double* out, void* stream) {
    if (n < 0 || !a || !out) return fail(HOPE_EINVAL, "hope_debug_math: bad argument");
    if (n == 0) return HOPE_OK;
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, fn, n, a, b, out);
    HIPCHK(hipGetLastError());
    return HOPE_OK;
}

int hope_debug_traffic(int mode, size_t bytes, void* buf, void* stream) {
    if (mode < 0 || mode > 7 || !buf || bytes < 4096) return fail(HOPE_EINVAL, "hope_debug_traffic: bad argument");
    const dim3 g(256 * 32), b(256);
    hipStream_t s = (hipStream_t)stream;
    char* q = (char*)buf;
    double* sink = (double*)buf;
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_traffic_calib<0>), g, b, 0, s, bytes, q, q, sink); break;
        case 1: hipLaunchKernelGGL((k_traffic_calib<1>), g, b, 0, s, bytes, q, q, sink); break;
        case 2: hipLaunchKernelGGL((k_traffic_calib<2>), g, b, 0, s, bytes, q, q, sink); break;
        case 3: hipLaunchKernelGGL((k_traffic_calib<3>), g, b, 0, s, bytes, q, q, sink); break;
        case 4: hipLaunchKernelGGL((k_traffic_calib<4>), g, b, 0, s, bytes, q, q, sink); break;
        case 5: hipLaunchKernelGGL((k_traffic_calib<5>), g, b, 0, s, bytes, q, q, sink); break;
        case 6: hipLaunchKernelGGL((k_traffic_calib<6>), g, b, 0, s, bytes, q, q, sink); break;
        default: hipLaunchKernelGGL((k_traffic_calib<7>), g, b, 0, s, bytes, q, q, sink); break;
    }
    HIPCHK(hipGetLastError());
    return HOPE_OK;
}

int hope_env_download_state(hope_env_t* h, double* pose, int32_t* t, double* accum) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h) return fail(HOPE_EINVAL, "hope_env_download_state: null handle");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> st((size_t)h->n * ST_WORDS);
    HIPCHK(hipMemcpy(st.data(), h->state, st.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->n; i++) {
        if (pose) { pose[3 * i] = st[4 * (size_t)i]; pose[3 * i + 1] = st[4 * (size_t)i + 1]; pose[3 * i + 2] = st[4 * (size_t)i + 2]; }
        if (accum) accum[i] = st[4 * (size_t)i + 3];
    }
    if (t) HIPCHK(hipMemcpy(t, h->tstep, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return HOPE_OK;
}

int hope_env_upload_state(hope_env_t* h, const double* pose, const int32_t* t, const double* accum) {
    { int rcs = settle_rs(h); if (rcs != HOPE_OK) return rcs; }
    if (!h) return fail(HOPE_EINVAL, "hope_env_upload_state: null handle");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(HOPE_EHIP, "hipSetDevice failed");
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> st((size_t)h->n * ST_WORDS);
    HIPCHK(hipMemcpy(st.data(), h->state, st.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->n; i++) {
        if (pose) { st[4 * (size_t)i] = pose[3 * i]; st[4 * (size_t)i + 1] = pose[3 * i + 1]; st[4 * (size_t)i + 2] = pose[3 * i + 2]; }
        if (accum) st[4 * (size_t)i + 3] = accum[i];
    }
    HIPCHK(hipMemcpy(h->state, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
    if (t) HIPCHK(hipMemcpy(h->tstep, t, (size_t)h->n * sizeof(int32_t), hipMemcpyHostToDevice));
    return HOPE_OK;
}

}  // extern "C"
