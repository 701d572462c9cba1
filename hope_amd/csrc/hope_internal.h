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
};
constexpr int RS_WORDS_PER_SCENE = 48;   // >= the 46 candidate words of generate_path

struct RsParams {
    int n, max_obst;
    int tile_cap;             // LDS tile capacity of this launch's tile class
    int max_queue;            // upper bound of *rs_count (= scenes in the class): grid size
    int slot_base, slot_dir;  // word storage slot of queue entry q = slot_base + slot_dir * q (classes fill from both ends)
    int obs_f64;
    const double* verts;      // [n][max_obst][4][2] world frame
    const int32_t* n_obst;    // [n]
    const double* scene_c;    // [n][SC_WORDS]
    const double* state;      // [n][ST_WORDS]
    const int32_t* rs_count;  // [1]
    const int32_t* rs_list;   // [n]
    RsWord* rs_words;         // [n][RS_WORDS_PER_SCENE] kept words of the queued scenes, indexed by candidate call index
    uint8_t* rs_order;        // [n][RS_WORDS_PER_SCENE] heapdict pop order: call index of the k-th popped word
    int32_t* rs_nwords;       // [n]
    int8_t* rs_word;          // [n][8]
    void* rs_lengths;         // real [n][5]
};

// bird's-eye image observation (hope_bev.hip)
constexpr int BEV_IMG = HOPE_IMG_SIZE;
constexpr int BEV_TRAJ_LEN = HOPE_TRAJ_RENDER_LEN;
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
hipError_t launch_rs_search(const RsParams& p, hipStream_t stream, LaunchTimer* timer);   // one tile class
size_t rs_lds_bytes(int max_obst);
size_t rs_words_bytes_per_scene();
hipError_t launch_bev_image(const BevParams& p, hipStream_t stream, LaunchTimer* timer);
size_t bev_lds_bytes();

}  // namespace hope
