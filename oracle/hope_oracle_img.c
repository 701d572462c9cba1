/*
 * hope_oracle_img.c -- CPU ORACLE for the bird's-eye image observation (test infrastructure, NOT product code).
 *
 * Restates obs['img'] of the reference: CarParking._render (src/env/car_parking_base.py:301-320),
 * _get_img_observation (:322-350), Obs_Processor.process_img (src/env/observation_processor.py:11-23) and the
 * transpose of env_wrapper.py:53-54.  Same rules as hope_oracle.c: only tests/, smoke() and bench.py's cpu_baseline
 * leg may load it.
 *
 * PARITY UNPINNED.  The pixels are produced by two third-party libraries that are absent from /root/reference and
 * from this image and that requirements.txt leaves unversioned: pygame (SDL2) and opencv-python.  The functions
 * below restate their published algorithms as of pygame 2.1-2.6 / OpenCV 4.x:
 *   - pygame.draw.polygon, width=0: src_c/draw.c draw_fillpoly (integer vertices by C truncation; scan-line with the
 *     floor/ceil rule on alternate intersections; the extra pass for horizontal border edges; clip to the surface)
 *   - pygame.draw.polygon, width=1: draw.c lines(closed) -> draw_line (Bresenham, err = (dx > dy ? dx : -dy) / 2)
 *   - pygame.transform.rotate: src_c/transform.c surf_rotate + rotate() (angle parsed as C float, 16.16 fixed-point
 *     nearest-neighbour sampling, background = top-left source pixel, rotate90 for multiples of 90 degrees)
 *   - Rect(center=...) and Surface.blit / subsurface integer placement
 *   - cv2.resize(INTER_LINEAR) 256 -> 64 on uint8: source x = 4 dx + 1.5, i.e. the rounded mean of the 2 x 2 centre of
 *     every 4 x 4 block ((a + b + c + d + 2) >> 2 in OpenCV's 11-bit fixed point)
 *   - shapely: affine_transform (a*x + b*y + xoff, left to right) and LinearRing.centroid (GEOS
 *     Centroid::addLineSegments, length-weighted segment midpoints)
 * There is no reference-produced image anywhere under /root/reference to compare with, so the only pins are the
 * hand-checkable known-answer tests of tests/test_oracle_image.py.
 *
 * Palette ids in the 500 x 500 world raster: 0 background, 1 obstacle, 2 start outline, 3 dest, 4 vehicle,
 * 5 + j trajectory colour TRAJ_COLORS[j] (configs.py:80-89).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../hope_amd/csrc/hope_math.h"
#ifdef ORC_USE_LIBM
#define hm_sin sin
#define hm_cos cos
#endif

void orc_create_box(const double *pose, double *box);

#define WIN 500          /* WIN_W = WIN_H  configs.py:92-93 */
#define OBS 256          /* OBS_W = OBS_H  configs.py:88-89 */
#define OUTW 64          /* OBS_W // downsample_rate  observation_processor.py:8,13 */
#define RENDER_K 12.0    /* K  configs.py:103 */
#define TRAJ_LEN 20      /* TRAJ_RENDER_LEN  configs.py:86 */

/* ---- colours (configs.py:80-89, 26-30) ------------------------------------------------------------------- */
static void palette_rgb(int id, uint8_t *rgb) {
    static const uint8_t fixed[5][3] = {{255, 255, 255}, {150, 150, 150}, {100, 149, 237}, {69, 139, 0}, {30, 144, 255}};
    if (id < 5) { rgb[0] = fixed[id][0]; rgb[1] = fixed[id][1]; rgb[2] = fixed[id][2]; return; }
    /* np.linspace((10,10,10,255), (10,10,200,255), 20, dtype=uint8): step 190/19 = 10 exactly */
    rgb[0] = 10; rgb[1] = 10; rgb[2] = (uint8_t)(10 + 10 * (id - 5));
}

/* ---- pygame draw.c ------------------------------------------------------------------------------------------ */
static void set_at(uint8_t *surf, int x, int y, int id) {       /* set_and_check_rect: clipped to the surface */
    if (x >= 0 && x < WIN && y >= 0 && y < WIN) surf[y * WIN + x] = (uint8_t)id;
}
static void horz_line_clip(uint8_t *surf, int id, int x1, int y, int x2) {   /* drawhorzlineclip */
    if (y < 0 || y >= WIN) return;
    if (x2 < x1) { int t = x1; x1 = x2; x2 = t; }
    if (x1 < 0) x1 = 0;
    if (x2 > WIN - 1) x2 = WIN - 1;
    if (x2 < 0 || x1 >= WIN) return;
    for (int x = x1; x <= x2; x++) surf[y * WIN + x] = (uint8_t)id;
}
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static void draw_fillpoly(uint8_t *surf, const int *px, const int *py, int n, int id) {
    int miny = py[0], maxy = py[0];
    int xi[16];
    for (int i = 1; i < n; i++) { if (py[i] < miny) miny = py[i]; if (py[i] > maxy) maxy = py[i]; }
    if (miny == maxy) {                                        /* polygon one pixel high */
        int minx = px[0], maxx = px[0];
        for (int i = 1; i < n; i++) { if (px[i] < minx) minx = px[i]; if (px[i] > maxx) maxx = px[i]; }
        horz_line_clip(surf, id, minx, miny, maxx);
        return;
    }
    for (int y = miny; y <= maxy; y++) {
        int ni = 0;
        for (int i = 0; i < n; i++) {
            int ip = i ? i - 1 : n - 1;
            int y1 = py[ip], y2 = py[i], x1, x2;
            if (y1 < y2) { x1 = px[ip]; x2 = px[i]; }
            else if (y1 > y2) { y2 = py[ip]; y1 = py[i]; x2 = px[ip]; x1 = px[i]; }
            else continue;                                     /* horizontal edges: handled below */
            if ((y >= y1 && y < y2) || (y == maxy && y2 == maxy)) {
                float intersect = (float)((y - y1) * (x2 - x1)) / (float)(y2 - y1);
                if (ni % 2 == 0) intersect = (float)floor(intersect);
                else intersect = (float)ceil(intersect);
                xi[ni++] = (int)intersect + x1;
            }
        }
        qsort(xi, ni, sizeof(int), cmp_int);
        for (int i = 0; i + 1 < ni; i += 2) horz_line_clip(surf, id, xi[i], y, xi[i + 1]);
    }
    for (int i = 0; i < n; i++) {                              /* horizontal border edges strictly inside in y */
        int ip = i ? i - 1 : n - 1;
        int y = py[i];
        if (miny < y && py[ip] == y && y < maxy) horz_line_clip(surf, id, px[i], y, px[ip]);
    }
}

static void draw_line(uint8_t *surf, int x1, int y1, int x2, int y2, int id) {
    if (x1 == x2 && y1 == y2) { set_at(surf, x1, y1, id); return; }
    if (y1 == y2) { int d = x1 < x2 ? 1 : -1; for (int s = 0; s <= abs(x1 - x2); s++) set_at(surf, x1 + d * s, y1, id); return; }
    if (x1 == x2) { int d = y1 < y2 ? 1 : -1; for (int s = 0; s <= abs(y1 - y2); s++) set_at(surf, x1, y1 + d * s, id); return; }
    int dx = abs(x2 - x1), sx = x1 < x2 ? 1 : -1;
    int dy = abs(y2 - y1), sy = y1 < y2 ? 1 : -1;
    int err = (dx > dy ? dx : -dy) / 2;
    while (x1 != x2 || y1 != y2) {
        set_at(surf, x1, y1, id);
        int e2 = err;
        if (e2 > -dx) { err -= dy; x1 += sx; }
        if (e2 < dy) { err += dx; y1 += sy; }
    }
    set_at(surf, x2, y2, id);
}

/* pygame.draw.polygon(surface, colour, points, width): points are the CLOSED shapely coordinate list (first point
 * repeated at the end), converted to int by C truncation */
static void draw_polygon(uint8_t *surf, const double *pts /*[n][2] pixel coords*/, int n, int id, int width) {
    int px[8], py[8];
    for (int i = 0; i < n; i++) { px[i] = (int)pts[2 * i]; py[i] = (int)pts[2 * i + 1]; }
    if (width == 0) { draw_fillpoly(surf, px, py, n, id); return; }
    for (int i = 1; i < n; i++) draw_line(surf, px[i - 1], py[i - 1], px[i], py[i], id);       /* lines(closed=True) */
    if (n > 2) draw_line(surf, px[n - 1], py[n - 1], px[0], py[0], id);
}

/* shapely affine_transform with matrix [k,0,0,k,bx,by] on a ring of nv vertices; appends the closing point */
static int ring_to_pixels(const double *ring /*[nv][2]*/, int nv, double bx, double by, double *out) {
    for (int i = 0; i <= nv; i++) {
        const double x = ring[2 * (i % nv)], y = ring[2 * (i % nv) + 1];
        out[2 * i] = RENDER_K * x + 0.0 * y + bx;
        out[2 * i + 1] = 0.0 * x + RENDER_K * y + by;
    }
    return nv + 1;
}

/* coord_transform_matrix (car_parking_base.py:139-147) */
static void render_offsets(const double *bbox /*xmin,xmax,ymin,ymax*/, double *bx, double *by) {
    *bx = 0.5 * (WIN - RENDER_K * (bbox[1] + bbox[0]));
    *by = 0.5 * (WIN - RENDER_K * (bbox[3] + bbox[2]));
}

/* _render (:301-320).  traj: the last m <= 20 entries of vehicle.trajectory, oldest first (m = min(len, 20));
 * traj_len = len(vehicle.trajectory) */
void orc_bev_world(const double *verts, const int32_t *nvert, int n_obst, const double *start, const double *dest,
                   const double *bbox, const double *pose, const double *traj, int m, int traj_len, uint8_t *world) {
    double bx, by, box[8], pix[10];
    render_offsets(bbox, &bx, &by);
    memset(world, 0, WIN * WIN);                                                        /* surface.fill(BG_COLOR) */
    for (int o = 0; o < n_obst; o++) {
        int n = ring_to_pixels(verts + 8 * o, nvert[o], bx, by, pix);
        draw_polygon(world, pix, n, 1, 0);
    }
    orc_create_box(start, box);
    draw_polygon(world, pix, ring_to_pixels(box, 4, bx, by, pix), 2, 1);
    orc_create_box(dest, box);
    draw_polygon(world, pix, ring_to_pixels(box, 4, bx, by, pix), 3, 0);
    orc_create_box(pose, box);
    draw_polygon(world, pix, ring_to_pixels(box, 4, bx, by, pix), 4, 0);
    if (traj_len > 1) {                                                                 /* RENDER_TRAJ :315-320 */
        for (int i = 0; i < m; i++) {
            orc_create_box(traj + 3 * i, box);                                          /* trajectory[-(m - i)] */
            draw_polygon(world, pix, ring_to_pixels(box, 4, bx, by, pix), 5 + (TRAJ_LEN - m + i), 0);
        }
    }
}

/* ---- pygame transform.c ------------------------------------------------------------------------------------ */
typedef struct {
    int turns;                 /* -1: general rotation, else rotate90 with this many quarter turns */
    int nw, nh;                /* size of the rotated surface */
    int cy, xd, yd, isin, icos, ax, ay;
} rot_t;

static void rotate_setup(double heading, rot_t *r) {
    const float angle = (float)(heading * (180.0 / 3.141592653589793));   /* np.rad2deg, then parsed as C float */
    if (fmod((double)angle, 90.0) == 0.0) {
        int t = ((int)angle / 90) % 4;
        if (t < 0) t += 4;
        r->turns = t;
        r->nw = WIN; r->nh = WIN;
        return;
    }
    r->turns = -1;
    const double radangle = angle * .01745329251994329;
    const double sangle = hm_sin(radangle), cangle = hm_cos(radangle);
    const double x = WIN, y = WIN;
    const double cx = cangle * x, cy = cangle * y, sx = sangle * x, sy = sangle * y;
    r->nw = (int)fmax(fmax(fmax(fabs(cx + sy), fabs(cx - sy)), fabs(-cx + sy)), fabs(-cx - sy));
    r->nh = (int)fmax(fmax(fmax(fabs(sx + cy), fabs(sx - cy)), fabs(-sx + cy)), fabs(-sx - cy));
    r->cy = r->nh / 2;
    r->xd = (WIN - r->nw) * 32768;
    r->yd = (WIN - r->nh) * 32768;
    r->isin = (int)(sangle * 65536);
    r->icos = (int)(cangle * 65536);
    r->ax = (r->nw * 32768) - (int)(cangle * ((r->nw - 1) * 32768));
    r->ay = (r->nh * 32768) - (int)(sangle * ((r->nw - 1) * 32768));
}

/* pixel (X, Y) of pygame.transform.rotate(world, angle): palette id, or -1 for the rotate background colour */
static int rotated_pixel(const uint8_t *world, const rot_t *r, int X, int Y) {
    if (r->turns >= 0) {
        int sx, sy;
        switch (r->turns) {
            case 0: sx = X; sy = Y; break;
            case 1: sx = WIN - 1 - Y; sy = X; break;
            case 2: sx = WIN - 1 - X; sy = WIN - 1 - Y; break;
            default: sx = Y; sy = WIN - 1 - X; break;
        }
        return world[sy * WIN + sx];
    }
    const int dx = (r->ax + r->isin * (r->cy - Y)) + r->xd + r->icos * X;
    const int dy = (r->ay - r->icos * (r->cy - Y)) + r->yd + r->isin * X;
    const int maxval = (WIN << 16) - 1;
    if (dx < 0 || dy < 0 || dx > maxval || dy > maxval) return -1;
    return world[(dy >> 16) * WIN + (dx >> 16)];
}

/* LinearRing.centroid of the vehicle box (GEOS Centroid::addLineSegments) */
static void ring_centroid(const double *box /*[4][2]*/, double *cx, double *cy) {
    double len = 0, sx = 0, sy = 0;
    for (int i = 0; i < 4; i++) {
        const double *p = box + 2 * i, *q = box + 2 * ((i + 1) % 4);
        const double dx = p[0] - q[0], dy = p[1] - q[1];
        const double seg = sqrt(dx * dx + dy * dy);
        if (seg == 0.0) continue;
        len += seg;
        sx += seg * ((p[0] + q[0]) / 2);
        sy += seg * ((p[1] + q[1]) / 2);
    }
    *cx = sx / len;
    *cy = sy / len;
}

/* _get_img_observation (:322-350): raw [256][256][3] RGB */
void orc_bev_raw(const uint8_t *world, const double *bbox, const double *pose, uint8_t *raw) {
    double bx, by, box[8], ccx, ccy;
    rot_t r;
    render_offsets(bbox, &bx, &by);
    rotate_setup(pose[2], &r);
    /* rotate.blit(capture, capture.get_rect(center=old_center)): Rect centre setter x = cx - (w >> 1) */
    const int x0 = WIN / 2 - (r.nw >> 1), y0 = WIN / 2 - (r.nh >> 1);
    orc_create_box(pose, box);
    ring_centroid(box, &ccx, &ccy);
    const double vcx = RENDER_K * ccx + 0.0 * ccy + bx, vcy = 0.0 * ccx + RENDER_K * ccy + by;
    const double ca = hm_cos(pose[2]), sa = hm_sin(pose[2]);
    const double ddx = (vcx - WIN / 2) * ca + (vcy - WIN / 2) * sa;
    const double ddy = -(vcx - WIN / 2) * sa + (vcy - WIN / 2) * ca;
    const int ox = (int)(-ddx), oy = (int)(-ddy);                                       /* observation.blit(rotate, (int(-dx), int(-dy))) */
    const int off = (WIN - OBS) / 2;                                                   /* subsurface */
    const int bg = world[0];                                                           /* rotate(): bgcolor = top-left source pixel */
    for (int y = 0; y < OBS; y++)
        for (int x = 0; x < OBS; x++) {
            const int rx = off + x - ox, ry = off + y - oy;                            /* pixel of `rotate` */
            int id = 0;                                                                /* observation.fill(BG_COLOR) */
            if (rx >= 0 && rx < WIN && ry >= 0 && ry < WIN) {
                id = rotated_pixel(world, &r, rx - x0, ry - y0);
                if (id < 0) id = bg;
            }
            palette_rgb(id, raw + 3 * (y * OBS + x));
        }
}

/* Obs_Processor.process_img (observation_processor.py:11-23) without the final /255: out [64][64][3] */
void orc_bev_process(const uint8_t *raw, uint8_t *out) {
    uint8_t *img = (uint8_t *)malloc(OBS * OBS * 3);
    memcpy(img, raw, OBS * OBS * 3);
    for (int i = 0; i < OBS * OBS; i++)                                                /* change_bg_color */
        if (img[3 * i] == 255 && img[3 * i + 1] == 255 && img[3 * i + 2] == 255) img[3 * i] = img[3 * i + 1] = img[3 * i + 2] = 0;
    for (int v = 0; v < OUTW; v++)
        for (int u = 0; u < OUTW; u++)
            for (int c = 0; c < 3; c++) {
                const int a = img[3 * ((4 * v + 1) * OBS + 4 * u + 1) + c], b = img[3 * ((4 * v + 1) * OBS + 4 * u + 2) + c];
                const int d = img[3 * ((4 * v + 2) * OBS + 4 * u + 1) + c], e = img[3 * ((4 * v + 2) * OBS + 4 * u + 2) + c];
                out[3 * (v * OUTW + u) + c] = (uint8_t)((a + b + d + e + 2) >> 2);
            }
    free(img);
}

/* whole observation: obs['img'] * 255 as uint8 [3][64][64] (after the wrapper's transpose, env_wrapper.py:53-54) */
void orc_bev_image(const double *verts, const int32_t *nvert, int n_obst, const double *start, const double *dest,
                   const double *bbox, const double *pose, const double *traj, int m, int traj_len, uint8_t *out_chw) {
    uint8_t *world = (uint8_t *)malloc(WIN * WIN), *raw = (uint8_t *)malloc(OBS * OBS * 3), hwc[OUTW * OUTW * 3];
    orc_bev_world(verts, nvert, n_obst, start, dest, bbox, pose, traj, m, traj_len, world);
    orc_bev_raw(world, bbox, pose, raw);
    orc_bev_process(raw, hwc);
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < OUTW * OUTW; i++) out_chw[c * OUTW * OUTW + i] = hwc[3 * i + c];
    free(world);
    free(raw);
}
