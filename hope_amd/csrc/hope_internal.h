// hope_internal.h -- declarations shared between the translation units of libhope_env.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hope_env.h"

namespace hope {

// One Reeds-Shepp word as the validation kernel reads it (64 B)
struct RsWord {
    double len[5];   // normalised (curvature-1) signed lengths
    double Lm;       // path.L / maxc  [m]
    int code;        // packed segment types + count
    int n;
    double pad;
};
constexpr int RS_WORDS_PER_SCENE = 48;   // >= the 46 candidate words of generate_path
// A Reeds-Shepp queue entry (rs_list, written by k_rs_compact): scene << 8 | obstacle count -- the validation kernel requests the
// scene's obstacle view with the record's header instead of behind it (n_obst <= HOPE_MAX_OBSTACLES = 255, enforced by
// hope_env_create -- the LDS formulas alone would allow more --; < 2^24 scenes per handle).  The scene sits in the HIGH bits and comes out by a shift on purpose: with `entry & 0xFFFFFF`
// the AMDGPU backend (ROCm 7.2) selected a 24-bit multiply for the `scene * SC_WORDS` address, dropped the mask as redundant for
// it, and then widened the multiply back to v_mad_u64_u32 on the UNMASKED register -- k_rs_words faulted on the first packed entry.
constexpr int RS_LIST_MAX_SCENES = 1 << 24;
__host__ __device__ inline int rs_list_pack(int scene, int n_obst) { return (int)(((unsigned)scene << 8) | (unsigned)n_obst); }
__host__ __device__ inline int rs_list_scene(int entry) { return (int)((unsigned)entry >> 8); }
__host__ __device__ inline int rs_list_n_obst(int entry) { return (int)((unsigned)entry & 0xFFu); }
// Per-search record (float64 words):
//   [0] int2 (scene, n_obst)   [1] int2 (kept words, words the stop rule :443 lets find_rs_path test)
//   [2..4] pose x, y, heading  [5..8] map box xmin, xmax, ymin, ymax   [9] (two-kernel validation) mask of the words k_rs_screen condemned
//   [10..15] the pop order: candidate slot (4 * family + reflection) of the k-th popped word, one byte each  -- k_rs_segs
//   [16 + c]  key of candidate slot c: path.L / maxc when set_path kept the word, -1 otherwise               -- k_rs_words
//   [64 + 8 c ..] the word of candidate slot c (RsWord)                                                       -- k_rs_words
//   [448 + 50 k ..] segment table of the k-th popped word (k < words to test), written by k_rs_segs: 5 segments x 8 doubles
//       (origin x, y, heading, cos, sin of the heading, type, length, [seg 0 only] int2 (type code, segment count));
//       table 0 also carries cos / sin(-pose heading) in the spare words [15], [23];
//       then 5 x 2 doubles = 5 x 4 floats for the float32 filter of k_rs_validate: per segment the origin in the frame
//       "world minus (map box xmin, ymin)" [m] (the frame of the scene's float32 obstacle view) and cos / sin of the WORLD heading at the origin
// Per queued search, written by k_rs_compact next to the queue entry (same index): everything k_rs_words / k_rs_segs need from the
// step's state -- [0..2] pose x, y, heading (the finished step's final pose), [3..6] map box xmin, xmax, ymin, ymax, [7..13] the goal
// normalised into the start frame (generate_path, reeds_shepp.py:540-557): X, Y, PHI, sin PHI, cos PHI and the "backwards" pair XB, YB
// (:206-207), computed ONCE per search by k_rs_compact's lane instead of by each of k_rs_words' four family-group waves -- so that the
// search chain behind k_rs_compact reads nothing the NEXT step's motion launch rewrites (`state`, `post`, and on an episode turnover the
// scene constants): pipelined steps only wait for k_rs_compact, and the front kernels start without the queue entry -> scene round trip.
constexpr int RS_IN_WORDS = 14;
constexpr double RS_MAXC = 0.3327130214085973;      // math.hm_tan(VALID_STEER[-1]) / WHEEL_BASE  (car_parking_base.py:422)
constexpr int RS_REC_HDR = 16;
constexpr int RS_REC_ORDER = 10;
constexpr int RS_REC_KEYS = RS_REC_HDR;
constexpr int RS_REC_WORDS = RS_REC_KEYS + RS_WORDS_PER_SCENE;
constexpr int RS_REC_SEGS = RS_REC_WORDS + 8 * RS_WORDS_PER_SCENE;
constexpr int RS_SEGW = 8, RS_SEG_F32 = 5 * RS_SEGW, RS_SEG_TABLE = RS_SEG_F32 + 5 * 2;
constexpr int RS_REC_DOUBLES = RS_REC_SEGS + RS_SEG_TABLE * RS_WORDS_PER_SCENE;

struct RsParams {
    int n, max_obst;
    int tile_cap;             // LDS tile capacity of this launch's tile class
    int max_queue;            // upper bound of *rs_count (= scenes in the class): grid size
    int slot_base, slot_dir;  // word storage slot of queue entry q = slot_base + slot_dir * q (classes fill from both ends)
    int obs_f64;
    int prio_front;           // wave priority (s_setprio) of k_rs_words / k_rs_segs: the latency-bound front of the launch chain
    const float4* obb;        // [n][max_obst] obstacle boxes (xmin, xmax, ymin, ymax), float32 rounded outwards
    const float4* fverts;     // [n][max_obst][2] float32 view of the obstacles in the scene's frame (origin = map box xmin, ymin): hope_dev.h obstacle_f32
    const float4* fbox;       // [n][max_obst] their boxes in that frame
    const uint8_t* eflag;     // [n][eflag_stride(max_obst)] per edge: robust for the reference's tolerance-free box test
    const double* verts;      // [n][max_obst][4][2] world frame
    const int32_t* n_obst;    // [n]
    const double* scene_c;    // [n][SC_WORDS]
    const double* state;      // [n][ST_WORDS]
    const double* rs_in;      // [max_queue][RS_IN_WORDS] search inputs by queue position (k_rs_compact)
    const int32_t* rs_count;  // [1]
    const int32_t* rs_list;   // [n]
    double* rs_rec;           // [n][RS_REC_DOUBLES] one record per queued scene (slot), see RS_REC_*
    int8_t* rs_word;          // [n][8]
    void* rs_lengths;         // real [n][5]
    // two-kernel validation (round 5): k_rs_screen condemns words at 8 waves per SIMD and queues the searches that still have a word
    // to walk (queue index, queue entry) for k_rs_validate_f; the counter is cleared by k_rs_compact.  Null: the one-kernel form
    int32_t* surv_count;      // [1]
    int2* surv_list;          // [max_queue]
};

// bird's-eye image observation (hope_bev.hip)
constexpr int BEV_IMG = HOPE_IMG_SIZE;
constexpr int BEV_TRAJ_LEN = HOPE_TRAJ_RENDER_LEN;
constexpr int BEV_LAYER_ROWS = 512, BEV_LAYER_STRIDE = 128;   // 500 x 500 world pixels, 4 per byte, tiled in 32 x 16 pixel blocks of 128 bytes: 16 x 32 blocks = 64 KiB per scene
constexpr int BEV_DYN_DIM = 256;               // trajectory layer: a TORUS of 256 x 256 world pixels (pixel (x, y) lives at (x mod 256, y mod 256)), one byte each,
constexpr size_t BEV_DYN_BYTES = (size_t)BEV_DYN_DIM * BEV_DYN_DIM;   // tiled in 16 x 8 pixel blocks of 128 bytes (16 x 32 blocks): 64 KiB per scene (round 3: the whole 500 x 500 surface, 256 KiB)
constexpr int BEV_SCENE_INTS = 16 + (3 + BEV_TRAJ_LEN) * 16 + (2 + BEV_TRAJ_LEN) * 64;   // k_bev_prep's per-scene scratch
struct BevParams {
    int n, max_obst;
    const double* verts;      // [n][max_obst][4][2] world frame
    const int32_t* n_obst;    // [n]
    const double* scene_c;    // [n][SC_WORDS]
    const double* state;      // [n][ST_WORDS]
    const double* traj;       // [n][BEV_TRAJ_LEN][3] ring of vehicle.trajectory: entry e lives in slot e % BEV_TRAJ_LEN
    const int32_t* traj_len;  // [n] len(vehicle.trajectory)
    int32_t* traj_valid;      // [n] trajectory entries below this index already have their span table
    uint8_t* layer;           // [n][BEV_LAYER_ROWS][BEV_LAYER_STRIDE] static layer (obstacles, start outline, dest): 2 bits per pixel
    uint8_t* dyn;             // [n][BEV_DYN_BYTES] trajectory layer: per pixel the code of the NEWEST trajectory box that covers it (0: none)
    int32_t* layer_valid;     // [n] the layer matches the scene's map
    int32_t* legacy_list;     // [1 + n] count + the scenes k_bev_prep queued for the per-tile raster launch of k_bev_image
    int32_t* rebuild;         // [1 + n] count, then the scenes k_bev_prep found with a stale layer (k_bev_static rebuilds them)
    int* scratch;             // [n][BEV_SCENE_INTS] map + box headers + span tables (k_bev_prep -> k_bev_image)
    const uint8_t* active;    // [n] or null
    uint8_t* img;             // [n][3][64][64]
    int debug;                // internal profiling switches (stages bits 0x1000.. >> 12)
};

// optional per-launch profiling hook: begin(kind) / end() bracket ONE kernel launch
struct LaunchTimer {
    virtual void begin(int kind, hipStream_t s) = 0;
    virtual void end(hipStream_t s) = 0;
};
// launches the Reeds-Shepp feasibility kernels over the scenes queued in rs_list (hope_rs.hip)
hipError_t launch_rs_search(const RsParams& p, hipStream_t stream, LaunchTimer* timer, hipEvent_t after_segs = nullptr);   // one tile class; after_segs: recorded behind k_rs_segs
hipError_t rs_prof_read(unsigned long long* out /*[16]*/, int reset);   // HOPE_RS_TIMING cycle accounting
hipError_t rs_init_tables();                                              // k_rs_validate_f's sample table, once per device (hope_env_create)
hipError_t rs_fstat_read(unsigned long long* out /*[16]*/, int reset);    // float32-filter statistics (HOPE_RS_DEBUG & 0x4000)
hipError_t rs_fdump_read(double* out /*[64][16]*/);
hipError_t rs_log_read(int* out /*[cap][4]*/, int cap, int* n, int reset);   // HOPE_RS_TIMING per-search log
size_t rs_lds_bytes(int max_obst);
size_t rs_screen_lds_bytes(int max_obst);
size_t rs_rec_bytes_per_scene();
hipError_t launch_bev_image(const BevParams& p, hipStream_t stream, LaunchTimer* timer, hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr,
                            hipEvent_t ev_join = nullptr);   // side: stream for the static-layer rebuild (next to k_bev_prep), or null
size_t bev_lds_bytes(bool legacy);

}  // namespace hope
