// hope_bev.hip -- bird's-eye image observation obs['img'], gfx950 only.
//
// Reference pipeline (car_parking_base.py:301-350, observation_processor.py:11-23, env_wrapper.py:53-54):
//   pygame draws obstacles / start outline / dest / vehicle / last 20 trajectory boxes into a 500 x 500 WORLD-aligned
//   surface (12 px/m, integer vertices), pygame.transform.rotate turns it by the heading with 16.16 fixed-point
//   nearest-neighbour sampling, two integer blits centre it on the vehicle, a 256 x 256 crop is taken, white becomes
//   black, and cv2.resize(INTER_LINEAR) to 64 x 64 reads exactly the 2 x 2 centre of every 4 x 4 block.
//
// MI355X formulation: nothing outside those 2 x 2 centres is ever observed, so no 500 x 500 surface, no rotated copy
// and no crop are materialised.  One 64-lane wave renders one 16 x 16-output tile of one scene:
//   1. the composite crop -> rotate -> world map is evaluated at the tile's four corner samples: the world pixels it
//      can touch form a window of at most 90 x 90 px, kept as one byte per pixel (palette id) in LDS (8 KB);
//   2. polygons are converted to integer pixels lane-parallel (lane = obstacle / car box), culled against the window
//      with a ballot, and the survivors are rasterised one after the other in the reference's draw order with
//      pygame's exact scan-line rule (lane = row; floor / ceil on alternate intersections; horizontal border pass)
//      or its Bresenham walk in closed form (lane = step) for the start outline;
//   3. the 1024 samples of the tile gather their palette id through the fixed-point map, colours are summed in a
//      packed 3 x 10-bit word, rounded like OpenCV ((s + 2) >> 2) and stored as uint8 CHW.
// All tiles of a scene run on the same XCD (block -> (scene, tile) map below) so the obstacle tile is fetched into
// one L2 only.  Integer work throughout, except the pose -> pixel conversion and the rotation setup (float64, shared
// hope_math.h), so the result is bit-identical to oracle/hope_oracle_img.c.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "hope_dev.h"
#include "hope_internal.h"

namespace hope {

namespace {

constexpr int WIN = 500;                       // WIN_W = WIN_H  configs.py:92-93
constexpr int CROP = 256;                      // OBS_W = OBS_H  configs.py:88-89
constexpr int CROP_OFF = (WIN - CROP) / 2;     // subsurface origin :343-344
constexpr int TILE_OUT = 16;                   // outputs per tile side
constexpr int TILES = BEV_IMG / TILE_OUT;      // 4 x 4 tiles per scene
constexpr int FB_DIM = 90;                     // window side: 61 crop px * sqrt(2) + rounding
constexpr int FB_STRIDE = 92;                  // bytes per window row: 23 dwords (odd -> lane-per-row is conflict-free)
constexpr int FB_BYTES = FB_DIM * FB_STRIDE;
constexpr double RENDER_K = 12.0;              // K  configs.py:103
constexpr int SRC_MAX = (WIN << 16) - 1;

struct Window { int x0, y0, x1, y1; };         // inclusive world-pixel bounds, inside [0, 499]^2

// the crop -> world-surface map of _get_img_observation, wave-uniform
struct Mapping {
    int turns;                                 // >= 0: rotate90 path with this many quarter turns; -1: general
    int cy, xd, yd, isin, icos, ax, ay;        // transform.c rotate()
    int x0, y0;                                // capture.get_rect(center=(250, 250)) placement
    int ox, oy;                                // observation.blit(rotate, (int(-dx), int(-dy)))
};

__device__ __forceinline__ int to_px(double X, double Y, double k0, double k1, double off) {
    return (int)(k0 * X + k1 * Y + off);       // shapely affine_transform (a*x + b*y + xoff), then C truncation
}

// source position of crop pixel (x, y) WITHOUT the range tests (used for the window bounds)
__device__ __forceinline__ void map_raw(const Mapping& m, int x, int y, int& sx, int& sy, int& dx, int& dy) {
    const int rx = CROP_OFF + x - m.ox, ry = CROP_OFF + y - m.oy;
    if (m.turns >= 0) {
        switch (m.turns) {                     // transform.c rotate90
            case 0: sx = rx; sy = ry; break;
            case 1: sx = WIN - 1 - ry; sy = rx; break;
            case 2: sx = WIN - 1 - rx; sy = WIN - 1 - ry; break;
            default: sx = ry; sy = WIN - 1 - rx; break;
        }
        dx = sx << 16; dy = sy << 16;
        return;
    }
    const int X = rx - m.x0, Y = ry - m.y0;
    dx = (m.ax + m.isin * (m.cy - Y)) + m.xd + m.icos * X;
    dy = (m.ay - m.icos * (m.cy - Y)) + m.yd + m.isin * X;
    sx = dx >> 16; sy = dy >> 16;
}

// 0: world pixel (sx, sy); 1: outside `rotate` -> observation.fill(BG_COLOR); 2: outside the source -> rotate()'s bgcolor
__device__ __forceinline__ int map_sample(const Mapping& m, int x, int y, int& sx, int& sy) {
    const int rx = CROP_OFF + x - m.ox, ry = CROP_OFF + y - m.oy;
    if (rx < 0 || rx >= WIN || ry < 0 || ry >= WIN) return 1;
    int dx, dy;
    map_raw(m, x, y, sx, sy, dx, dy);
    if (m.turns < 0 && (dx < 0 || dy < 0 || dx > SRC_MAX || dy > SRC_MAX)) return 2;
    return 0;
}

// ---- pygame draw.c on the LDS window ------------------------------------------------------------------------------
__device__ __forceinline__ void hline(uint8_t* fb, const Window& w, int id, int xa, int y, int xb, int lane) {
    if (y < w.y0 || y > w.y1) return;          // the window lies inside the surface: this is also drawhorzlineclip's test
    if (xb < xa) { const int t = xa; xa = xb; xb = t; }
    xa = max(xa, w.x0); xb = min(xb, w.x1);
    uint8_t* row = fb + (y - w.y0) * FB_STRIDE - w.x0;
    for (int x = xa + lane; x <= xb; x += WAVE) row[x] = (uint8_t)id;
}

// draw_fillpoly: n points (closing point included), wave-uniform
__device__ __forceinline__ void fill_poly(uint8_t* fb, const Window& w, const int (&px)[5], const int (&py)[5], int n,
                                          int id, int lane) {
    int miny = py[0], maxy = py[0], minx = px[0], maxx = px[0];
#pragma unroll
    for (int i = 1; i < 5; i++)
        if (i < n) { miny = min(miny, py[i]); maxy = max(maxy, py[i]); minx = min(minx, px[i]); maxx = max(maxx, px[i]); }
    if (miny == maxy) { hline(fb, w, id, minx, miny, maxx, lane); return; }
    const int ylo = max(miny, w.y0), yhi = min(maxy, w.y1);
    for (int yb = ylo; yb <= yhi; yb += WAVE) {
        const int y = yb + lane;
        int cnt = 0, a0 = INT_MAX, a1 = INT_MAX, a2 = INT_MAX, a3 = INT_MAX;
        if (y <= yhi) {
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (i >= n) continue;
                const int ip = i ? i - 1 : n - 1;
                int y1 = py[ip], y2 = py[i], x1 = px[ip], x2 = px[i];
                if (y1 == y2) continue;                                  // horizontal edges: border pass below
                if (y1 > y2) { int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
                if ((y >= y1 && y < y2) || (y == maxy && y2 == maxy)) {
                    float f = (float)((y - y1) * (x2 - x1)) / (float)(y2 - y1);
                    f = (cnt & 1) ? ceilf(f) : floorf(f);                // alternate floor / ceil in discovery order
                    const int xv = (int)f + x1;
                    if (cnt == 0) a0 = xv; else if (cnt == 1) a1 = xv; else if (cnt == 2) a2 = xv; else a3 = xv;
                    cnt++;
                }
            }
        }
        // qsort of at most four values (missing ones are INT_MAX and stay at the end)
        int t;
        if (a0 > a1) { t = a0; a0 = a1; a1 = t; }
        if (a2 > a3) { t = a2; a2 = a3; a3 = t; }
        if (a0 > a2) { t = a0; a0 = a2; a2 = t; }
        if (a1 > a3) { t = a1; a1 = a3; a3 = t; }
        if (a1 > a2) { t = a1; a1 = a2; a2 = t; }
        uint8_t* row = fb + (y - w.y0) * FB_STRIDE - w.x0;
        int xa = cnt >= 2 ? max(a0, w.x0) : 1, xb = cnt >= 2 ? min(a1, w.x1) : 0;
        while (__any(xa <= xb)) {
            if (xa <= xb) row[xa] = (uint8_t)id;
            xa++;
        }
        if (__any(cnt >= 4)) {
            xa = cnt >= 4 ? max(a2, w.x0) : 1; xb = cnt >= 4 ? min(a3, w.x1) : 0;
            while (__any(xa <= xb)) {
                if (xa <= xb) row[xa] = (uint8_t)id;
                xa++;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {                                        // horizontal border edges strictly inside in y
        if (i >= n) continue;
        const int ip = i ? i - 1 : n - 1;
        const int y = py[i];
        if (miny < y && py[ip] == y && y < maxy) hline(fb, w, id, px[i], y, px[ip], lane);
    }
}

// draw_line (Bresenham, err = (dx > dy ? dx : -dy) / 2) in closed form: step k of the major axis lands on
//   x-major: (x1 + k sx, y1 + sy ceil((k dy - dx/2) / dx))      y-major: (x1 + sx ceil((k dx - dy/2) / dy), y1 + k sy)
__device__ __forceinline__ void line(uint8_t* fb, const Window& w, int id, int x1, int y1, int x2, int y2, int lane) {
    const int dx = abs(x2 - x1), dy = abs(y2 - y1), sx = x1 < x2 ? 1 : -1, sy = y1 < y2 ? 1 : -1;
    const int steps = max(dx, dy);
    for (int k = lane; k <= steps; k += WAVE) {
        int x, y;
        if (steps == 0) { x = x1; y = y1; }
        else if (dx > dy) { x = x1 + k * sx; y = y1 + sy * (int)ceilf((float)(k * dy - dx / 2) / (float)dx); }
        else { y = y1 + k * sy; x = x1 + sx * (int)ceilf((float)(k * dx - dy / 2) / (float)dy); }
        if (x >= w.x0 && x <= w.x1 && y >= w.y0 && y <= w.y1) fb[(y - w.y0) * FB_STRIDE + x - w.x0] = (uint8_t)id;
    }
}

// packed colour (r | g << 10 | b << 20) of a palette id AFTER change_bg_color (white -> black)
__device__ __forceinline__ uint32_t palette(int id) {
    uint32_t r, g, b;
    switch (id) {
        case 0: r = 0; g = 0; b = 0; break;            // BG_COLOR (255,255,255) -> (0,0,0)  observation_processor.py:18-23
        case 1: r = 150; g = 150; b = 150; break;      // OBSTACLE_COLOR
        case 2: r = 100; g = 149; b = 237; break;      // START_COLOR
        case 3: r = 69; g = 139; b = 0; break;         // DEST_COLOR
        case 4: r = 30; g = 144; b = 255; break;       // COLOR_POOL[0]
        default: r = 10; g = 10; b = 10 + 10 * (id - 5); break;   // TRAJ_COLORS[id - 5]  configs.py:84-89
    }
    return r | (g << 10) | (b << 20);
}

__global__ __launch_bounds__(64) void k_bev_image(BevParams p) {
    extern __shared__ __align__(16) uint8_t lds_raw[];
    uint8_t* fb = lds_raw;
    uint32_t* pal = (uint32_t*)(lds_raw + FB_BYTES + 8);
    const int lane = threadIdx.x;
    // blocks b, b + 8, b + 16, ... share an XCD: give all 16 tiles of a scene to one XCD
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int scene = ((j >> 4) << 3) + xcd, tile = j & 15;
    if (scene >= p.n) return;
    if (p.active && !p.active[scene]) return;
    const int tx = tile & 3, ty = tile >> 2;

    const double* sc = p.scene_c + (size_t)scene * SC_WORDS;
    const double* st = p.state + (size_t)scene * ST_WORDS;
    const double px_ = st[0], py_ = st[1], ph = st[2];
    // coord_transform_matrix (car_parking_base.py:139-147)
    const double offx = 0.5 * (WIN - RENDER_K * (sc[SC_BBOX + 1] + sc[SC_BBOX])), offy = 0.5 * (WIN - RENDER_K * (sc[SC_BBOX + 3] + sc[SC_BBOX + 2]));

    // ---- the crop -> world map (wave-uniform) --------------------------------------------------------------------
    Mapping m;
    {
        double sh, ch;
        hm_sincos(ph, &sh, &ch);
        const Box vb = make_box(px_, py_, ch, sh);
        // LinearRing.centroid (GEOS Centroid::addLineSegments)
        double len = 0, sx = 0, sy = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int i2 = (i + 1) & 3;
            const double ex = vb.x[i] - vb.x[i2], ey = vb.y[i] - vb.y[i2];
            const double seg = sqrt(ex * ex + ey * ey);
            if (seg == 0.0) continue;
            len += seg;
            sx += seg * ((vb.x[i] + vb.x[i2]) / 2);
            sy += seg * ((vb.y[i] + vb.y[i2]) / 2);
        }
        const double ccx = sx / len, ccy = sy / len;
        const double vcx = RENDER_K * ccx + 0.0 * ccy + offx, vcy = 0.0 * ccx + RENDER_K * ccy + offy;
        const double ddx = (vcx - WIN / 2) * ch + (vcy - WIN / 2) * sh;
        const double ddy = -(vcx - WIN / 2) * sh + (vcy - WIN / 2) * ch;
        m.ox = (int)(-ddx); m.oy = (int)(-ddy);
        const float angle = (float)(ph * (180.0 / 3.141592653589793));        // np.rad2deg -> C float argument
        if (hm_fmod((double)angle, 90.0) == 0.0) {
            int t = ((int)angle / 90) % 4;
            if (t < 0) t += 4;
            m.turns = t;
            m.x0 = 0; m.y0 = 0;
            m.cy = m.xd = m.yd = m.isin = m.icos = m.ax = m.ay = 0;
        } else {
            m.turns = -1;
            const double radangle = angle * .01745329251994329;
            double sangle, cangle;
            hm_sincos(radangle, &sangle, &cangle);
            const double cx = cangle * WIN, cy = cangle * WIN, sxx = sangle * WIN, syy = sangle * WIN;
            const int nw = (int)fmax(fmax(fmax(fabs(cx + syy), fabs(cx - syy)), fabs(-cx + syy)), fabs(-cx - syy));
            const int nh = (int)fmax(fmax(fmax(fabs(sxx + cy), fabs(sxx - cy)), fabs(-sxx + cy)), fabs(-sxx - cy));
            m.cy = nh / 2;
            m.xd = (WIN - nw) * 32768;
            m.yd = (WIN - nh) * 32768;
            m.isin = (int)(sangle * 65536);
            m.icos = (int)(cangle * 65536);
            m.ax = (nw * 32768) - (int)(cangle * ((nw - 1) * 32768));
            m.ay = (nh * 32768) - (int)(sangle * ((nw - 1) * 32768));
            m.x0 = WIN / 2 - (nw >> 1); m.y0 = WIN / 2 - (nh >> 1);
        }
    }

    // ---- world window of this tile: the map is affine, so the extremes are at the corner samples --------------------
    Window w;
    bool need_bg = false;
    {
        const int cxa = 4 * TILE_OUT * tx + 1, cxb = 4 * TILE_OUT * tx + 4 * TILE_OUT - 2;
        const int cya = 4 * TILE_OUT * ty + 1, cyb = 4 * TILE_OUT * ty + 4 * TILE_OUT - 2;
        int lo_x = INT_MAX, hi_x = INT_MIN, lo_y = INT_MAX, hi_y = INT_MIN;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int sx, sy, dx, dy;
            map_raw(m, (c & 1) ? cxb : cxa, (c & 2) ? cyb : cya, sx, sy, dx, dy);
            lo_x = min(lo_x, sx); hi_x = max(hi_x, sx); lo_y = min(lo_y, sy); hi_y = max(hi_y, sy);
            need_bg = need_bg || (m.turns < 0 && (dx < 0 || dy < 0 || dx > SRC_MAX || dy > SRC_MAX));
        }
        w.x0 = max(lo_x, 0); w.x1 = min(hi_x, WIN - 1); w.y0 = max(lo_y, 0); w.y1 = min(hi_y, WIN - 1);
    }

    if (lane < 25) pal[lane] = palette(lane);

    const int n_obst = p.n_obst[scene];
    const int traj_len = p.traj_len[scene];
    const int m_traj = min(traj_len, BEV_TRAJ_LEN);
    int bg_id = 0;
    // pass 0 (rare): the rotate() background colour is the surface's top-left pixel -- rasterise a 1 x 1 window there
    for (int pass = need_bg ? 0 : 1; pass < 2; pass++) {
        const Window cw = pass == 0 ? Window{0, 0, 0, 0} : w;
        {   // surface.fill(BG_COLOR)
            uint32_t* f4 = (uint32_t*)fb;
            for (int i = lane; i < FB_BYTES / 4; i += WAVE) f4[i] = 0;
        }
        wsync();
        if (cw.x0 <= cw.x1 && cw.y0 <= cw.y1) {
            // obstacles (car_parking_base.py:303-305), lane = obstacle
            for (int base = 0; base < n_obst; base += WAVE) {
                const int o = base + lane;
                int vx[4] = {0, 0, 0, 0}, vy[4] = {0, 0, 0, 0}, nv = 4;
                bool hit = false;
                if (o < n_obst) {
                    const double* v = p.verts + ((size_t)scene * p.max_obst + o) * 8;
#pragma unroll
                    for (int k = 0; k < 4; k++) { vx[k] = to_px(v[2 * k], v[2 * k + 1], RENDER_K, 0.0, offx); vy[k] = to_px(v[2 * k], v[2 * k + 1], 0.0, RENDER_K, offy); }
                    if (v[6] == v[4] && v[7] == v[5]) nv = 3;              // triangles repeat their last vertex (include/hope_env.h)
                    const int bx0 = min(min(vx[0], vx[1]), min(vx[2], vx[3])), bx1 = max(max(vx[0], vx[1]), max(vx[2], vx[3]));
                    const int by0 = min(min(vy[0], vy[1]), min(vy[2], vy[3])), by1 = max(max(vy[0], vy[1]), max(vy[2], vy[3]));
                    hit = !(bx1 < cw.x0 || bx0 > cw.x1 || by1 < cw.y0 || by0 > cw.y1);
                }
                unsigned long long mask = __ballot(hit);
                while (mask) {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    int qx[5], qy[5];
#pragma unroll
                    for (int k = 0; k < 4; k++) { qx[k] = __builtin_amdgcn_readlane(vx[k], l); qy[k] = __builtin_amdgcn_readlane(vy[k], l); }
                    const int qn = __builtin_amdgcn_readlane(nv, l);
                    if (qn == 3) { qx[3] = qx[0]; qy[3] = qy[0]; qx[4] = qx[0]; qy[4] = qy[0]; }
                    else { qx[4] = qx[0]; qy[4] = qy[0]; }
                    fill_poly(fb, cw, qx, qy, qn + 1, 1, lane);
                }
            }
            // start outline, dest, vehicle, trajectory boxes oldest -> newest (:307-320), lane = box
            {
                int vx[4] = {0, 0, 0, 0}, vy[4] = {0, 0, 0, 0}, id = 0;
                bool hit = false;
                const int n_box = 3 + (traj_len > 1 ? m_traj : 0);
                if (lane < n_box) {
                    double bxp, byp, bh;
                    if (lane == 0) { bxp = sc[SC_START]; byp = sc[SC_START + 1]; bh = sc[SC_START + 2]; id = 2; }
                    else if (lane == 1) { bxp = sc[SC_DEST]; byp = sc[SC_DEST + 1]; bh = sc[SC_DEST + 2]; id = 3; }
                    else if (lane == 2) { bxp = px_; byp = py_; bh = ph; id = 4; }
                    else {
                        const int i = lane - 3;
                        const int e = traj_len - m_traj + i;                 // vehicle.trajectory[-(m - i)]
                        const double* tp = p.traj + ((size_t)scene * BEV_TRAJ_LEN + (e % BEV_TRAJ_LEN)) * 3;
                        bxp = tp[0]; byp = tp[1]; bh = tp[2];
                        id = 5 + (BEV_TRAJ_LEN - m_traj + i);                // TRAJ_COLORS[-(m - i)]
                    }
                    double sb, cb;
                    hm_sincos(bh, &sb, &cb);
                    const Box bb = make_box(bxp, byp, cb, sb);
#pragma unroll
                    for (int k = 0; k < 4; k++) { vx[k] = to_px(bb.x[k], bb.y[k], RENDER_K, 0.0, offx); vy[k] = to_px(bb.x[k], bb.y[k], 0.0, RENDER_K, offy); }
                    const int bx0 = min(min(vx[0], vx[1]), min(vx[2], vx[3])), bx1 = max(max(vx[0], vx[1]), max(vx[2], vx[3]));
                    const int by0 = min(min(vy[0], vy[1]), min(vy[2], vy[3])), by1 = max(max(vy[0], vy[1]), max(vy[2], vy[3]));
                    hit = !(bx1 < cw.x0 || bx0 > cw.x1 || by1 < cw.y0 || by0 > cw.y1);
                }
                unsigned long long mask = __ballot(hit);
                while (mask) {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    int qx[5], qy[5];
#pragma unroll
                    for (int k = 0; k < 4; k++) { qx[k] = __builtin_amdgcn_readlane(vx[k], l); qy[k] = __builtin_amdgcn_readlane(vy[k], l); }
                    qx[4] = qx[0]; qy[4] = qy[0];
                    const int qid = __builtin_amdgcn_readlane(id, l);
                    if (qid == 2) {                                          // width=1: lines(closed=True)
#pragma unroll
                        for (int k = 1; k < 5; k++) line(fb, cw, 2, qx[k - 1], qy[k - 1], qx[k], qy[k], lane);
                        line(fb, cw, 2, qx[4], qy[4], qx[0], qy[0], lane);
                    } else {
                        fill_poly(fb, cw, qx, qy, 5, qid, lane);
                    }
                    wsync();                                                 // painter's order between overlapping boxes
                }
            }
        }
        wsync();
        if (pass == 0) { bg_id = fb[0]; wsync(); }
    }

    // ---- gather: lane = 4 consecutive outputs of one tile row; cv2.resize reads crop pixels (4u+1|2, 4v+1|2) ---------
    const int v = TILE_OUT * ty + (lane >> 2);
    const int u0 = TILE_OUT * tx + 4 * (lane & 3);
    const uint32_t pal_bg = pal[bg_id];
    uint32_t out_r = 0, out_g = 0, out_b = 0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const int u = u0 + jj;
        uint32_t sum = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            int sx, sy;
            const int kind = map_sample(m, 4 * u + 1 + (s & 1), 4 * v + 1 + (s >> 1), sx, sy);
            uint32_t c = 0;                                                   // white -> black
            if (kind == 0) c = pal[fb[(sy - w.y0) * FB_STRIDE + sx - w.x0]];
            else if (kind == 2) c = pal_bg;
            sum += c;
        }
        const uint32_t r = ((sum & 1023) + 2) >> 2, g = (((sum >> 10) & 1023) + 2) >> 2, bl = (((sum >> 20) & 1023) + 2) >> 2;
        out_r |= r << (8 * jj); out_g |= g << (8 * jj); out_b |= bl << (8 * jj);
    }
    uint8_t* img = p.img + (size_t)scene * 3 * BEV_IMG * BEV_IMG + v * BEV_IMG + u0;
    *(uint32_t*)(img) = out_r;
    *(uint32_t*)(img + BEV_IMG * BEV_IMG) = out_g;
    *(uint32_t*)(img + 2 * BEV_IMG * BEV_IMG) = out_b;
}

}  // namespace

size_t bev_lds_bytes() { return FB_BYTES + 8 + 32 * sizeof(uint32_t); }

hipError_t launch_bev_image(const BevParams& p, hipStream_t stream, LaunchTimer* timer) {
    const int groups = (p.n + 7) / 8;
    const dim3 grid(groups * 8 * TILES * TILES), block(WAVE);
    if (timer) timer->begin(HOPE_K_IMAGE, stream);
    hipLaunchKernelGGL(k_bev_image, grid, block, bev_lds_bytes(), stream, p);
    if (timer) timer->end(stream);
    return hipGetLastError();
}

}  // namespace hope
