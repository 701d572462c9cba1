// hope_scenegen.cpp -- host-side generator of Normal / Complex / Extrem parking lots, multi-threaded.
//
// The reference draws a new case for every episode with Python rejection samplers: generate_bay_parking_case /
// generate_parallel_parking_case (src/env/parking_map_normal.py:40-246, 248-457) behind ParkingMapNormal.reset (:474-494),
// ~300 cases/s per core.  A batch of 65 536 scenes that turns over ~1 % of its episodes per step needs ~10^6 new maps per second,
// so the maps come from here: the same recipe (the distributions are pinned to the reference's samplers by two-sample KS tests,
// tests/test_scenes_distribution.py) in C++ on all host cores, written straight into the packed arrays hope_env_set_scenes /
// hope_env_set_pool take.  Scene i of a call depends only on (seed, i): the result does not depend on the thread count.
// The reference delegates `distance` / `intersects` to shapely; here they are the boundary-only ring predicates of
// hope_amd/scenes.py (rings_intersect / rings_distance), restated.
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include <pthread.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <new>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "hope_env.h"

namespace {

// src/configs.py:13-17, 43-70; parking_map_normal.py:20-22
constexpr double WHEEL_BASE = 2.8, FRONT_HANG = 0.96, REAR_HANG = 0.93, WIDTH = 1.94;
constexpr double LENGTH = WHEEL_BASE + FRONT_HANG + REAR_HANG;
constexpr double GAP = 0.1;                        // MIN_DIST_TO_OBST
constexpr double P_WALL = 0.5, P_EXTRA = 0.7;
constexpr int N_EXTRA = 3;
constexpr double PI = 3.14159265358979323846;

struct Level {
    double min_lot_len, max_lot_len, min_lot_wid, max_lot_wid, para_wall, bay_wall;
    int n_obst;
};
// index 0 Normal, 1 Complex, 2 Extrem (no bay lots at the Extrem level: configs.py:57-70)
const Level LEVELS[3] = {
    {LENGTH * 1.25, LENGTH * 1.25 + 0.5, WIDTH + 0.85, WIDTH + 1.2, 4.5, 7.0, 3},
    {LENGTH + 0.9, LENGTH * 1.25, WIDTH + 0.4, WIDTH + 0.85, 4.0, 6.0, 5},
    {LENGTH + 0.6, LENGTH + 0.9, 0.0, 0.0, 3.5, 0.0, 8},
};

struct Rng {                                       // splitmix64 stream + Box-Muller
    uint64_t s;
    bool have = false;
    double spare = 0;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }          // [0, 1)
    double normal() {
        if (have) { have = false; return spare; }
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        const double r = sqrt(-2.0 * log(u1)), a = 2.0 * PI * u2;
        spare = r * sin(a);
        have = true;
        return r * cos(a);
    }
};

struct P2 { double x, y; };
struct Quad { P2 p[4]; };

Quad create_box(double x, double y, double yaw) {   // State.create_box (vehicle.py:32-36)
    const double c = cos(yaw), s = sin(yaw);
    const double bx[4] = {-REAR_HANG, FRONT_HANG + WHEEL_BASE, FRONT_HANG + WHEEL_BASE, -REAR_HANG};
    const double by[4] = {-WIDTH / 2, -WIDTH / 2, WIDTH / 2, WIDTH / 2};
    Quad q;
    for (int k = 0; k < 4; k++) q.p[k] = {c * bx[k] + (-s) * by[k] + x, s * bx[k] + c * by[k] + y};
    return q;
}
inline double cross(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
inline int sgn(double v) { return (v > 0) - (v < 0); }

// LinearRing.intersects(LinearRing): any pair of boundary segments shares a point
bool rings_intersect(const Quad& a, const Quad& b) {
    for (int i = 0; i < 4; i++) {
        const P2 p1 = a.p[i], p2 = a.p[(i + 1) & 3];
        for (int j = 0; j < 4; j++) {
            const P2 q1 = b.p[j], q2 = b.p[(j + 1) & 3];
            if (!(std::min(p1.x, p2.x) <= std::max(q1.x, q2.x) && std::min(p1.y, p2.y) <= std::max(q1.y, q2.y) &&
                  std::min(q1.x, q2.x) <= std::max(p1.x, p2.x) && std::min(q1.y, q2.y) <= std::max(p1.y, p2.y)))
                continue;
            const double d1 = cross(p2.x - p1.x, p2.y - p1.y, q1.x - p1.x, q1.y - p1.y);
            const double d2 = cross(p2.x - p1.x, p2.y - p1.y, q2.x - p1.x, q2.y - p1.y);
            const double d3 = cross(q2.x - q1.x, q2.y - q1.y, p1.x - q1.x, p1.y - q1.y);
            const double d4 = cross(q2.x - q1.x, q2.y - q1.y, p2.x - q1.x, p2.y - q1.y);
            if (sgn(d1) * sgn(d2) <= 0 && sgn(d3) * sgn(d4) <= 0) return true;
        }
    }
    return false;
}
double pt_seg(P2 p, P2 a, P2 b) {
    const double abx = b.x - a.x, aby = b.y - a.y, den = abx * abx + aby * aby;
    double t = den > 0 ? ((p.x - a.x) * abx + (p.y - a.y) * aby) / den : 0.0;
    t = std::min(1.0, std::max(0.0, t));
    const double cx = a.x + t * abx, cy = a.y + t * aby;
    return sqrt((p.x - cx) * (p.x - cx) + (p.y - cy) * (p.y - cy));
}
// LinearRing.distance(LinearRing): 0 when they meet, else the closest vertex-to-edge gap
double rings_distance(const Quad& a, const Quad& b) {
    if (rings_intersect(a, b)) return 0.0;
    double d = 1e300;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            d = std::min(d, pt_seg(a.p[i], b.p[j], b.p[(j + 1) & 3]));
            d = std::min(d, pt_seg(b.p[i], a.p[j], a.p[(j + 1) & 3]));
        }
    return d;
}
inline double clipn(Rng& r, double mean, double sd, double lo, double hi) { return std::min(hi, std::max(lo, r.normal() * sd + mean)); }
inline double uni(Rng& r, double lo, double hi) { return r.uni() * (hi - lo) + lo; }
inline P2 polar(Rng& r, P2 o, double a0, double a1, double r0, double r1) {
    const double ang = clipn(r, (a0 + a1) / 2, (a1 - a0) / 4, a0, a1);
    const double rad = clipn(r, (r0 + r1) / 2, (r1 - r0) / 4, r0, r1);
    return {o.x + cos(ang) * rad, o.y + sin(ang) * rad};
}

struct Case {
    double start[3], dest[3];
    std::vector<Quad> rings;
};

// one rejection-sampling attempt of generate_bay_parking_case (bay) / generate_parallel_parking_case; false: rejected
bool one_case(int level, bool bay, Rng& rng, Case& out) {
    const Level& L = LEVELS[level];
    const double half = bay ? 15.0 : 18.0;
    double space_hi, space_lo, wall, yaw0, pitch, yaw_lo, yaw_hi;
    int low_a, low_b, n_extra;
    if (bay) {
        space_hi = L.max_lot_wid - WIDTH; space_lo = L.min_lot_wid - WIDTH;
        wall = L.bay_wall; yaw0 = PI / 2; pitch = WIDTH;
        yaw_lo = PI * 5 / 12; yaw_hi = PI * 7 / 12;
        low_a = 0; low_b = 3;                       // rear-right, rear-left corners touch the back wall
        n_extra = N_EXTRA;
    } else {
        space_hi = L.max_lot_len - LENGTH; space_lo = L.min_lot_len - LENGTH;
        wall = L.para_wall; yaw0 = 0.0; pitch = LENGTH;
        yaw_lo = -PI / 12; yaw_hi = PI / 12;
        low_a = 0; low_b = 1;                       // rear-right, front-right
        n_extra = N_EXTRA - 1;
    }
    const Quad back = {{{half, 0.0}, {half, -1.0}, {-half, -1.0}, {-half, 0.0}}};
    auto slot_pose = [&](double x, double* pose) {
        const double yaw = clipn(rng, yaw0, PI / 36, yaw_lo, yaw_hi);
        const Quad b = create_box(x, 0.0, yaw);
        const double y_min = -std::min(b.p[low_a].y, b.p[low_b].y) + GAP;
        const double y = clipn(rng, y_min + 0.4, 0.2, y_min, y_min + 0.8);
        pose[0] = x; pose[1] = y; pose[2] = yaw;
    };
    double dest[3];
    slot_pose(0.0, dest);
    const Quad dest_ring = create_box(dest[0], dest[1], dest[2]);
    const P2 rb = dest_ring.p[0], rf = dest_ring.p[1], lf = dest_ring.p[2], lb = dest_ring.p[3];
    bool ok = true;
    std::vector<Quad> extras;
    // obstacle next to the slot on side `sign` (-1 left, +1 right): a wall-like quad or a parked car followed by
    // further parked cars (each kept with probability .7)
    auto side = [&](int sign, P2 near_a, P2 near_b, double d_lo, double d_hi) -> Quad {
        if (rng.uni() < P_WALL) {
            const double a0 = sign < 0 ? PI * 11 / 12 : -PI / 12, a1 = sign < 0 ? PI * 13 / 12 : PI / 12;
            const P2 pa = polar(rng, near_a, a0, a1, d_lo, d_hi);
            const P2 pb = polar(rng, near_b, a0, a1, d_lo, d_hi);
            if (sign < 0) return Quad{{pa, pb, {-half, 0.0}, {-half, pa.y}}};
            return Quad{{{half, pa.y}, {half, 0.0}, pb, pa}};
        }
        double x = sign * (pitch + uni(rng, d_lo, d_hi));
        double pose[3];
        slot_pose(x, pose);
        const Quad first = create_box(pose[0], pose[1], pose[2]);
        for (int k = 0; k < n_extra; k++) {
            x += sign * (pitch + GAP + uni(rng, d_lo, d_hi));
            const double y = pose[1] + clipn(rng, 0, 0.05, -0.1, 0.1);
            pose[0] = x; pose[1] = y; pose[2] = clipn(rng, yaw0, PI / 36, yaw_lo, yaw_hi);
            const Quad ring = create_box(pose[0], pose[1], pose[2]);
            if (rng.uni() < P_EXTRA) extras.push_back(ring);
        }
        return first;
    };
    const Quad left = bay ? side(-1, lf, lb, space_hi / 5 * 1, space_hi / 5 * 4) : side(-1, lb, rb, space_lo / 5 * 1, space_hi / 5 * 4);
    const double gap_l = rings_distance(dest_ring, left);
    const double d_lo = std::max(space_lo - gap_l, 0.0) + GAP, d_hi = std::max(space_hi - gap_l, 0.0) + GAP;
    const Quad right = bay ? side(+1, rf, rb, d_lo, d_hi) : side(+1, lf, rf, d_lo, d_hi);
    const double gap_r = rings_distance(dest_ring, right);
    if (gap_r + gap_l < space_lo || gap_r + gap_l > space_hi || gap_l < GAP || gap_r < GAP) ok = false;
    std::vector<Quad> rings;
    rings.push_back(back); rings.push_back(left); rings.push_back(right);
    for (auto& e : extras) rings.push_back(e);
    for (auto& r : rings) if (rings_intersect(r, dest_ring)) ok = false;
    double top = -1e300;
    for (auto& r : rings) for (int k = 0; k < 4; k++) top = std::max(top, r.p[k].y);
    top += GAP;
    std::vector<Quad> far;
    if (rng.uni() < 0.2) {                              // only a thin wall across the aisle
        const double y0 = wall + top + GAP;
        far.push_back(Quad{{{-half, y0}, {half, y0}, {half, y0 + 0.1}, {-half, y0 + 0.1}}});
    } else {
        const Quad zone = {{{-half, wall + top}, {half, wall + top}, {half, wall + top + 8}, {-half, wall + top + 8}}};
        for (int k = 0; k < L.n_obst; k++) {
            const double px = uni(rng, -half + 2, half - 2), py = uni(rng, wall + top + 2, wall + top + 6), pyaw = rng.uni() * PI * 2;
            Quad ring = create_box(px, py, pyaw);
            for (int v = 0; v < 4; v++) { ring.p[v].x += 0.5 * rng.uni(); ring.p[v].y += 0.5 * rng.uni(); }
            bool hit = rings_intersect(ring, zone);
            for (size_t o = 0; o < far.size() && !hit; o++) hit = rings_intersect(ring, far[o]);
            if (!hit) far.push_back(ring);
        }
    }
    for (auto& f : far) rings.push_back(f);
    double sx, sy, syaw;
    for (;;) {                                          // start pose in the aisle, clear of everything
        sx = uni(rng, -half / 2, half / 2);
        sy = uni(rng, top + 1, wall + top - 1);
        syaw = clipn(rng, 0, PI / 6, -PI / 2, PI / 2);
        if (rng.uni() < 0.5) syaw += PI;
        const Quad sbox = create_box(sx, sy, syaw);
        bool hit = rings_intersect(dest_ring, sbox);
        for (size_t o = 0; o < rings.size() && !hit; o++) hit = rings_intersect(rings[o], sbox);
        if (!hit) break;
    }
    if (!bay && cos(syaw) < 0) {                        // parallel: face the slot the way the car arrives
        const Quad b = create_box(dest[0], dest[1], dest[2]);   // _flip_box_orientation (parking_map_dlp.py:117-123)
        const double cx = 0.25 * (b.p[0].x + b.p[1].x + b.p[2].x + b.p[3].x), cy = 0.25 * (b.p[0].y + b.p[1].y + b.p[2].y + b.p[3].y);
        dest[0] = 2 * cx - dest[0]; dest[1] = 2 * cy - dest[1]; dest[2] += PI;
    }
    if (!ok) return false;
    out.start[0] = sx; out.start[1] = sy; out.start[2] = syaw;
    memcpy(out.dest, dest, sizeof(dest));
    out.rings.swap(rings);
    return true;
}

int level_index(int level) { return level < 0 || level > 2 ? -1 : level; }

// Default fan-out: HOPE_HOST_THREADS if set, else the CPUs this process may run on (sched_getaffinity) -- NOT every hardware
// thread of the node: with one process per GPU each rank pins itself to its share of the cores (hope_amd.dist.pin_rank_to_cores),
// and 8 ranks x hardware_concurrency() threads would oversubscribe the host eightfold.
int default_threads() {
    if (const char* e = getenv("HOPE_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    // ... and not more than twice the container's CPU-time quota (cgroup v2 cpu.max "quota period"): a box that shows 256 CPUs but
    // grants 16 CPUs' worth of time throttles a wider team (round 6: the generator peaked at 32 threads, the oracle's OpenMP loop too)
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        double per = 0.0;
        if (fscanf(f, "%31s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0.0) {
            const int cap = (int)(2.0 * atof(q) / per + 0.5);
            if (cap >= 1 && cap < n) n = cap;
        }
        fclose(f);
    }
    return n;
}


// Persistent worker pool (round 5).  hope_scenegen_generate used to start fresh std::threads on every call: a pool refill is
// three calls of ~2 700 lots, i.e. ~1 ms of generation behind ~3 ms of thread start-up for the 86 threads a 256-CPU host gave
// it -- 0.9 M lots/s on 256 hardware threads against 0.25 M/s on ONE, and worse on a rank pinned to 32 CPUs.  The workers are
// now created once (lazily, up to the largest fan-out asked for), sleep on a condition variable between calls and are shared
// by every caller of the process (calls are serialised; a fork()ed child starts with an empty pool).
class WorkerPool {
    std::mutex run_m_;                     // one job at a time
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    const std::function<void()>* job_ = nullptr;
    uint64_t generation_ = 0;
    int want_ = 0, claimed_ = 0, pending_ = 0;

    std::vector<int> cpus_;                // the CPUs of the process's affinity mask when the current job was posted

    void loop(int index) {
        uint64_t seen = 0;
        int my_cpu = -1;
        // lowest scheduling priority: a worker is pinned, and the thread that enqueues the env steps may sit on that very CPU (the job
        // is posted by a background fill thread, whose CPU says nothing about the step loop's) -- measured: 20-step windows of the
        // bench 0.52 .. 0.60 ms per step while a fill ran.  On an idle CPU a nice-19 thread runs at full speed.
        setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 19);
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            if (claimed_ >= want_) continue;                // more workers than this job wants
            claimed_++;
            const std::function<void()>* job = job_;
            // Worker i sits on the i-th CPU FROM THE END of the mask, the caller's current CPU left out.  Without this a woken worker
            // is queued on the CPU that woke it -- the caller's, which is busy with the job -- and waits for the periodic load
            // balancer: measured here, a 27 ms job on 2 .. 8 sleeping workers finished in 27 ms.  Re-pinned when the mask changed
            // (hope_amd.dist.pin_rank_to_cores runs after the first pool use in a multi-rank process).
            // ... counted from the END of the mask: the first CPUs are where interrupt handling and the runtime's own threads tend to
            // live (on one of two boxes a refill pinned to CPUs 0 .. 7 cost the 20-step bench window 9 %, on the other nothing)
            const int cpu = cpus_.empty() ? -1 : cpus_[cpus_.size() - 1 - (size_t)index % cpus_.size()];
            lk.unlock();
            if (cpu >= 0 && cpu != my_cpu) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpu, &one);
                if (sched_setaffinity(0, sizeof(one), &one) == 0) my_cpu = cpu;
            }
            (*job)();
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }

public:
    // runs fn on n_threads threads (the caller is one of them) and returns when all of them are done
    void run(int n_threads, const std::function<void()>& fn) {
        std::lock_guard<std::mutex> run_lk(run_m_);
        const int helpers = std::max(0, n_threads - 1);
        {
            std::unique_lock<std::mutex> lk(m_);
            cpus_.clear();
            cpu_set_t set;
            CPU_ZERO(&set);
            const int here = sched_getcpu();                 // not the CPU the caller runs on: a pinned worker there would hold up the
            if (sched_getaffinity(0, sizeof(set), &set) == 0) //   thread that posted the job (often the one that enqueues the env steps)
                for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &set) && c != here) cpus_.push_back(c);
            while ((int)workers_.size() < helpers) {
                const int index = (int)workers_.size();
                workers_.emplace_back([this, index] { loop(index); });
                workers_.back().detach();                   // (process-lifetime threads: nothing to join at exit)
            }
            job_ = &fn; want_ = helpers; claimed_ = 0; pending_ = helpers;
            generation_++;
        }
        if (helpers > 0) cv_work_.notify_all();
        fn();
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
    void forget_workers() { new (this) WorkerPool(); }      // fork(): the child has none of the parent's threads
};
WorkerPool* g_pool = nullptr;
std::once_flag g_pool_once;
WorkerPool& pool() {
    std::call_once(g_pool_once, [] {
        g_pool = new WorkerPool();
        pthread_atfork(nullptr, nullptr, [] { if (g_pool) g_pool->forget_workers(); });
    });
    return *g_pool;
}

}  // namespace

extern "C" {

// what hope_scenegen_generate uses when n_threads <= 0 (reported by bench.py next to the rank count)
int hope_scenegen_default_threads(void) { return default_threads(); }

// n scenes of `level` (0 Normal, 1 Complex, 2 Extrem) as ParkingMapNormal.reset draws them (parking_map_normal.py:474-494: bay
// with probability 1/2 for Normal / Complex, parallel otherwise; bbox = floor / ceil of min / max(start, dest) -/+ 10 m).
// bay_mode: -1 as the reference, 0 parallel only, 1 bay only (the test hook of the distribution tests).  Outputs (host): start
// [n][3], dest [n][3], bbox [n][4], verts [n][max_obstacles][4][2] (unused slots untouched), n_obst [n], case_id [n] (0 bay,
// 1 parallel; may be null).  Scene i depends only on (seed, first_index + i).  n_threads <= 0: HOPE_HOST_THREADS, else the CPUs in
// this process's affinity mask (a rank's share of the node once hope_amd.dist.pin_rank_to_cores ran).
// Returns 0, HOPE_EINVAL, or -100 - i if scene i had more than max_obstacles obstacles (cannot happen for max_obstacles >= 18).
int hope_scenegen_generate(int level, int bay_mode, int n, uint64_t seed, int64_t first_index, int max_obstacles, double* start,
                           double* dest, double* bbox, double* verts, int32_t* n_obst, int32_t* case_id, int n_threads) {
    if (level_index(level) < 0 || n < 0 || max_obstacles <= 0 || !start || !dest || !bbox || !verts || !n_obst) return HOPE_EINVAL;
    if (n == 0) return HOPE_OK;
    // Fan-out: at most the CPUs of the affinity mask, at most 64 unless asked for (n_threads > 0 / HOPE_HOST_THREADS), and at least
    // 64 lots per thread -- measured on the 256-CPU host of an MI355X box (profiles/r05_host_generator_threads.txt, a pool refill =
    // 3 calls of 2 730 lots): 1 thread 0.61 M lots/s, 8: 4.3 M, 32: 9.6 M, 64: 8.0 M, 128: 5.4 M, 256: 0.3 M (waking and pinning 255
    // sleepers costs more than the 3 us a lot takes).
    int nt = n_threads > 0 ? n_threads : std::min(default_threads(), getenv("HOPE_HOST_THREADS") ? (1 << 20) : 64);
    nt = std::max(1, std::min(nt, n / 64));
    // chunks of 8 .. 32 lots handed out from one counter: ~8 chunks per thread, so that the rejection samplers' uneven cost per lot
    // evens out
    const int chunk = std::max(8, std::min(32, n / std::max(1, 8 * nt)));
    nt = std::max(1, std::min(nt, (n + chunk - 1) / chunk));
    std::atomic<int> next{0}, err{0};
    const std::function<void()> work = [&]() {
        Case c;
        for (;;) {
            const int a = next.fetch_add(chunk);
            if (a >= n) break;
            for (int i = a; i < std::min(n, a + chunk); i++) {
                Rng rng(seed * 0x9E3779B97F4A7C15ull + (uint64_t)(first_index + i) * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull);
                rng.next();
                const bool bay = level != 2 && (bay_mode == 1 || (bay_mode < 0 && rng.uni() > 0.5));
                while (!one_case(level, bay, rng, c)) {}
                if ((int)c.rings.size() > max_obstacles) { err.store(-100 - i); continue; }
                memcpy(start + 3 * (size_t)i, c.start, 24);
                memcpy(dest + 3 * (size_t)i, c.dest, 24);
                double* bb = bbox + 4 * (size_t)i;
                bb[0] = floor(std::min(c.start[0], c.dest[0]) - 10); bb[1] = ceil(std::max(c.start[0], c.dest[0]) + 10);
                bb[2] = floor(std::min(c.start[1], c.dest[1]) - 10); bb[3] = ceil(std::max(c.start[1], c.dest[1]) + 10);
                double* v = verts + (size_t)i * max_obstacles * 8;
                for (size_t o = 0; o < c.rings.size(); o++)
                    for (int k = 0; k < 4; k++) { v[8 * o + 2 * k] = c.rings[o].p[k].x; v[8 * o + 2 * k + 1] = c.rings[o].p[k].y; }
                n_obst[i] = (int32_t)c.rings.size();
                if (case_id) case_id[i] = bay ? 0 : 1;
            }
        }
    };
    if (nt == 1) work();
    else pool().run(nt, work);
    return err.load();
}

}  // extern "C"
