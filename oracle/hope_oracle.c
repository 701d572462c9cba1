/*
 * hope_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, float64 restatement of the jiamiya/HOPE `src/env` step hot path, function by
 * function, each citing the reference file:line it follows.  It exists so that the HIP
 * kernels in hope_amd/csrc can be checked for parity; it is also timed by bench.py as the
 * `cpu_baseline` ("port").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under hope_amd/ links, imports or calls it.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).  Two flavours: the default
 * takes sin/cos/atan2/... from hope_amd/csrc/hope_math.h (deterministic, bit-compatible with the HIP kernels);
 * -DORC_USE_LIBM takes them from glibc (what Python's math gave the reference run).  Both are pinned by
 * tests/test_oracle_golden.py.
 *
 * Pinning status:
 *   - pinned against reference-generated golden vectors (tests/golden, .npz files): KSModel.step,
 *     _fast_calc_lidar_obs, ActionMask tables + get_steps + post_process, Reeds-Shepp
 *     calc_all_paths, is_traj_valid, find_rs_path, _get_targt_repr, _get_reward arithmetic,
 *     action_rescale, reward_shaping.
 *   - PARITY UNPINNED (arithmetic lives in shapely/GEOS, absent from /root/reference and from
 *     this image; requirements.txt leaves it unversioned; the pickle format implies
 *     shapely 1.8 / GEOS 3.x): LinearRing.intersects (collision), Polygon.intersection().area
 *     (arrival, box-union reward), LinearRing.distance(Point) (10 m lidar ring cull) and the
 *     ray/hull ranges.  These follow GEOS's published algorithms (RobustLineIntersector
 *     envelope+orientation test with an exact-sign fallback; Area::ofRingSigned;
 *     Distance::pointToSegment) and are pinned only by analytic known-answer tests.
 */
#include <math.h>
#include <stdint.h>

/* Deterministic elementary functions shared with the HIP kernels (see the header for why).  The oracle stays an
 * independent restatement of the reference ALGORITHM; only sin/cos/atan2/... come from this common header, and
 * they are validated against libm in tests/test_math.py.  Build with -DORC_USE_LIBM to use glibc instead. */
#include "../hope_amd/csrc/hope_math.h"
#ifdef ORC_USE_LIBM
#define hm_sin sin
#define hm_cos cos
#define hm_tan tan
#define hm_atan2 atan2
#define hm_asin asin
#define hm_acos acos
#define hm_hypot hypot
#define hm_fmod fmod
#define hm_tanh tanh
#endif
#include <stdlib.h>
#include <string.h>

#define NBEAM 120
#define NACT 42
#define NITER 10
#define UPS 10
#define NL (NBEAM * UPS)

/* ---- src/configs.py:13-38,95-104 ---------------------------------------------------- */
static const double WHEEL_BASE = 2.8, FRONT_HANG = 0.96, REAR_HANG = 0.93, WIDTH = 1.94;
static const double VALID_SPEED_LO = -2.5, VALID_SPEED_HI = 2.5;
static const double VALID_STEER_LO = -0.75, VALID_STEER_HI = 0.75;
#define NUM_STEP 10
static const double STEP_LENGTH = 5e-2;
static const double LIDAR_RANGE = 10.0;
static const double TOLERANT_TIME = 200;
static const double RS_MAX_DIST = 10;
static const double PI = 3.141592653589793; /* math.pi */

enum { ST_CONTINUE = 1, ST_ARRIVED = 2, ST_COLLIDED = 3, ST_OUTBOUND = 4, ST_OUTTIME = 5 };

/* configs.py:20-24  VehicleBox (closed ring, CCW) */
static void vehicle_box_local(double c[4][2]) {
    c[0][0] = -REAR_HANG;              c[0][1] = -WIDTH / 2;
    c[1][0] = FRONT_HANG + WHEEL_BASE; c[1][1] = -WIDTH / 2;
    c[2][0] = FRONT_HANG + WHEEL_BASE; c[2][1] = WIDTH / 2;
    c[3][0] = -REAR_HANG;              c[3][1] = WIDTH / 2;
}

static double clipd(double v, double lo, double hi) { /* np.clip */
    double m = v < lo ? lo : v;
    return m > hi ? hi : m;
}

/* ===================================================================================== */
/* tables                                                                                */
/* ===================================================================================== */
static int g_init = 0;
static double g_actions[NACT][2];
static double g_boxes[NACT][NITER][4][2];
static double g_hull_base[NBEAM];
static double g_beam_a[NBEAM], g_beam_b[NBEAM];
static double *g_dist_star = 0; /* [NL][NACT][NITER] */
static double g_ds_max[NL];      /* max over (a, k) of g_dist_star[l]: a beam whose scan value reaches it restricts no action */
static const double *g_ds_max_of = 0; /* the table g_ds_max was computed from */

/* configs.py:108-115: np.arange(0.75, -(0.75+0.075), -0.075) -> start + i*step, 21 values */
static void build_actions(void) {
    double start = VALID_STEER_HI;
    double stop = -(VALID_STEER_HI + VALID_STEER_HI / 10);
    double step = -VALID_STEER_HI / 10;
    int n = (int)ceil((stop - start) / step); /* numpy arange length */
    (void)n;                                  /* == 21 */
    /* numpy's arange fill: buf[0]=start, buf[1]=start+step, delta=buf[1]-buf[0], buf[i]=start+i*delta */
    double second = start + step;
    double delta = second - start;
    for (int i = 0; i < 21; i++) {
        double v = i == 0 ? start : (i == 1 ? second : start + i * delta);
        g_actions[i][0] = v;
        g_actions[i][1] = 1;
        g_actions[21 + i][0] = v;
        g_actions[21 + i][1] = -1;
    }
}

/* action_mask.py:84-112 init_vehicle_box */
static void build_vehicle_boxes(void) {
    double car[4][2];
    vehicle_box_local(car);
    for (int a = 0; a < NACT; a++) {
        double radius = 1 / (hm_tan(g_actions[a][0]) / WHEEL_BASE);
        double Ox = 0 - radius * hm_sin(0.0);
        double Oy = 0 + radius * hm_cos(0.0);
        double delta_phi = 0.5 * g_actions[a][1] / 10 / radius;
        double ptheta = 0;
        for (int k = 0; k < NITER; k++) {
            ptheta = ptheta + delta_phi;
            double px = Ox + radius * hm_sin(ptheta);
            double py = Oy - radius * hm_cos(ptheta);
            double ct = hm_cos(ptheta), st = hm_sin(ptheta);
            for (int v = 0; v < 4; v++) {
                g_boxes[a][k][v][0] = ct * car[v][0] - st * car[v][1] + px;
                g_boxes[a][k][v][1] = st * car[v][0] + ct * car[v][1] + py;
            }
        }
    }
}

/* Range from the rear axle to the hull along beam i.  Reference: LineString((0,0),(cos,sin)*10)
 * .intersection(VehicleBox).distance(ORIGIN)  (lidar_simulator.py:48-53, action_mask.py:21-29).
 * GEOS is absent: closed-form ray/rectangle hit point, then sqrt(x^2+y^2).  UNPINNED. */
static double hull_range(double ex, double ey) {
    double best = INFINITY;
    double car[4][2];
    vehicle_box_local(car);
    for (int k = 0; k < 4; k++) {
        double ax = car[k][0], ay = car[k][1], bx = car[(k + 1) & 3][0], by = car[(k + 1) & 3][1];
        double sx = bx - ax, sy = by - ay;
        double den = ex * sy - ey * sx;
        if (den == 0) continue;
        double t = (ax * sy - ay * sx) / den;
        double u = (ax * ey - ay * ex) / den;
        if (t >= 0 && t <= 1 && u >= 0 && u <= 1) {
            double hx = t * ex, hy = t * ey;
            double dd = sqrt(hx * hx + hy * hy);
            if (dd < best) best = dd;
        }
    }
    return best;
}

static void build_hull_base(void) {
    for (int l = 0; l < NBEAM; l++) {
        /* action_mask.py:25-26: np.hm_cos(l*np.pi/lidar_num*2)*lidar_range */
        double th = l * PI / NBEAM * 2;
        g_hull_base[l] = hull_range(hm_cos(th) * LIDAR_RANGE, hm_sin(th) * LIDAR_RANGE);
        /* lidar_simulator.py:86-88: theta = a*pi/lidar_num*2 ; a=sin, b=-cos */
        g_beam_a[l] = hm_sin(th);
        g_beam_b[l] = -hm_cos(th);
    }
}

/* action_mask.py:31-82 _intersect for ONE (lidar edge, vehicle edge) pair -> norm or inf */
static double mask_intersect_norm(double x1s1, double y1s1, double x2s1, double y2s1, double x1s2,
                                  double y1s2, double x2s2, double y2s2) {
    const double tol = 1e-8;
    double a = y2s1 - y1s1, b = x1s1 - x2s1, c = y1s1 * x2s1 - x1s1 * y2s1;
    double d = y2s2 - y1s2, e = x1s2 - x2s2, f = y1s2 * x2s2 - x1s2 * y2s2;
    double det = a * e - b * d;
    int parallel = (det == 0);
    if (parallel) det = 1;
    double raw_x = (b * f - c * e) / det;
    double raw_y = (c * d - a * f) / det;
    if (raw_x > fmax(x1s1, x2s1) + tol) raw_x = INFINITY;
    if (raw_x < fmin(x1s1, x2s1) - tol) raw_x = INFINITY;
    if (raw_y > fmax(y1s1, y2s1) + tol) raw_y = INFINITY;
    if (raw_y < fmin(y1s1, y2s1) - tol) raw_y = INFINITY;
    if (raw_x > fmax(x1s2, x2s2) + tol) raw_x = INFINITY;
    if (raw_x < fmin(x1s2, x2s2) - tol) raw_x = INFINITY;
    if (raw_y > fmax(y1s2, y2s2) + tol) raw_y = INFINITY;
    if (raw_y < fmin(y1s2, y2s2) - tol) raw_y = INFINITY;
    if (parallel) raw_x = INFINITY;
    return sqrt(raw_x * raw_x + raw_y * raw_y); /* np.linalg.norm(axis=-1) */
}

/* action_mask.py:114-163 precompute + _linear_interpolate */
void orc_init(void);
static void upsample_dist_star(double *coarse);
static void build_dist_star(void) {
    double *coarse = (double *)malloc(sizeof(double) * (NBEAM + 1) * NACT * NITER);
    const double max_distance = LIDAR_RANGE * 10;
    for (int l = 0; l < NBEAM; l++) {
        double ang = (double)l / NBEAM * 2 * PI; /* lidar_line_idx/lidar_num*2*np.pi */
        double ex = hm_cos(ang) * max_distance, ey = hm_sin(ang) * max_distance;
        for (int a = 0; a < NACT; a++)
            for (int k = 0; k < NITER; k++) {
                double best = -INFINITY;
                for (int v = 0; v < 4; v++) {
                    /* edge v: point0 = shifted box (next vertex), point1 = this vertex  (:131-134) */
                    const double *p0 = g_boxes[a][k][(v + 1) & 3];
                    const double *p1 = g_boxes[a][k][v];
                    double n = mask_intersect_norm(0, 0, ex, ey, p0[0], p0[1], p1[0], p1[1]);
                    if (n == INFINITY) n = 0;
                    if (n > best) best = n;
                }
                coarse[(l * NACT + a) * NITER + k] = best;
            }
    }
    upsample_dist_star(coarse);
    free(coarse);
}

static void refresh_ds_max(void) {
    for (int l = 0; l < NL; l++) {
        double m = -INFINITY;
        for (int q = 0; q < NACT * NITER; q++) m = fmax(m, g_dist_star[(size_t)l * NACT * NITER + q]);
        g_ds_max[l] = m;
    }
    g_ds_max_of = g_dist_star;
}

/* action_mask.py:145-163 _linear_interpolate on the (120,42,10) table -> (1200,42,10).
 * coarse has room for NBEAM+1 rows (row NBEAM = circular copy of row 0). */
static void upsample_dist_star(double *coarse) {
    memcpy(coarse + NBEAM * NACT * NITER, coarse, sizeof(double) * NACT * NITER); /* circular */
    if (!g_dist_star) g_dist_star = (double *)malloc(sizeof(double) * NL * NACT * NITER);
    for (int j = 0; j < NL; j++) {
        double w2 = (double)(j % UPS) / UPS;
        double w1 = 1 - w2;
        const double *x0 = coarse + (j / UPS) * NACT * NITER;
        const double *x1 = coarse + (j / UPS + 1) * NACT * NITER;
        for (int q = 0; q < NACT * NITER; q++) g_dist_star[j * NACT * NITER + q] = x0[q] * w1 + x1[q] * w2;
    }
    refresh_ds_max();
}

/* inject a coarse (120,42,10) table (e.g. the one captured from the reference) and upsample it */
void orc_set_dist_star_coarse(const double *coarse_in) {
    orc_init();
    double *coarse = (double *)malloc(sizeof(double) * (NBEAM + 1) * NACT * NITER);
    memcpy(coarse, coarse_in, sizeof(double) * NBEAM * NACT * NITER);
    upsample_dist_star(coarse);
    free(coarse);
}

void orc_init(void) {
    if (g_init) return;
    build_actions();
    build_vehicle_boxes();
    build_hull_base();
    build_dist_star();
    g_init = 1;
}

void orc_get_tables(double *actions, double *boxes, double *hull_base, double *beam_a, double *beam_b,
                    double *dist_star) {
    orc_init();
    if (actions) memcpy(actions, g_actions, sizeof(g_actions));
    if (boxes) memcpy(boxes, g_boxes, sizeof(g_boxes));
    if (hull_base) memcpy(hull_base, g_hull_base, sizeof(g_hull_base));
    if (beam_a) memcpy(beam_a, g_beam_a, sizeof(g_beam_a));
    if (beam_b) memcpy(beam_b, g_beam_b, sizeof(g_beam_b));
    if (dist_star) memcpy(dist_star, g_dist_star, sizeof(double) * NL * NACT * NITER);
}

/* Inject tables (any pointer may be NULL = keep).  The beam sin/cos table matters bit-for-bit:
 * the reference's lidar has tolerance-free bbox tests, so a hit on an exactly axis-aligned
 * edge depends on the last ulp of sin/hm_cos(theta_i), which numpy (SIMD) and libm round
 * differently on some beams.  Tests pin the oracle to the reference's captured table. */
void orc_set_tables(const double *hull_base, const double *beam_a, const double *beam_b, const double *dist_star) {
    orc_init();
    if (hull_base) memcpy(g_hull_base, hull_base, sizeof(g_hull_base));
    if (beam_a) memcpy(g_beam_a, beam_a, sizeof(g_beam_a));
    if (beam_b) memcpy(g_beam_b, beam_b, sizeof(g_beam_b));
    if (dist_star) { memcpy(g_dist_star, dist_star, sizeof(double) * NL * NACT * NITER); refresh_ds_max(); }
}

/* ===================================================================================== */
/* kinematics                                                                            */
/* ===================================================================================== */
/* vehicle.py:69-96 KSModel.step with step_time=1 (called from Vehicle.step :136-148).
 * pose = (x, y, heading) in/out; returns clipped speed/steer. */
void orc_ks_step(double *pose, const double *action, double *speed_steer) {
    double x = pose[0], y = pose[1], h = pose[2];
    double steer = action[0], speed = action[1];
    speed = clipd(speed, VALID_SPEED_LO, VALID_SPEED_HI);
    steer = clipd(steer, VALID_STEER_LO, VALID_STEER_HI);
    const int mini_iter = 20;
    for (int it = 0; it < mini_iter; it++) {
        x += speed * hm_cos(h) * STEP_LENGTH / mini_iter;
        y += speed * hm_sin(h) * STEP_LENGTH / mini_iter;
        h += speed * hm_tan(steer) / WHEEL_BASE * STEP_LENGTH / mini_iter;
    }
    pose[0] = x; pose[1] = y; pose[2] = h;
    if (speed_steer) { speed_steer[0] = speed; speed_steer[1] = steer; }
}

/* vehicle.py:32-36 State.create_box: affine [cos,-sin,sin,cos,x,y] on VehicleBox */
void orc_create_box(const double *pose, double *box /*[4][2]*/) {
    double car[4][2];
    vehicle_box_local(car);
    double ct = hm_cos(pose[2]), st = hm_sin(pose[2]);
    for (int v = 0; v < 4; v++) {
        box[2 * v] = ct * car[v][0] + (-st) * car[v][1] + pose[0];
        box[2 * v + 1] = st * car[v][0] + ct * car[v][1] + pose[1];
    }
}

/* ===================================================================================== */
/* GEOS-semantics geometry (UNPINNED, see header)                                        */
/* ===================================================================================== */
static void two_sum(double a, double b, double *s, double *e) {
    *s = a + b;
    double bb = *s - a;
    *e = (a - (*s - bb)) + (b - bb);
}
static void two_prod(double a, double b, double *p, double *e) {
    *p = a * b;
    *e = fma(a, b, -*p);
}
/* exact sign of sum of n doubles: grow a non-overlapping expansion (Shewchuk) */
static int exact_sum_sign(const double *t, int n) {
    double ex[16];
    int m = 0;
    for (int i = 0; i < n; i++) {
        double q = t[i];
        int mm = 0;
        for (int j = 0; j < m; j++) {
            double s, e;
            two_sum(q, ex[j], &s, &e);
            if (e != 0) ex[mm++] = e;
            q = s;
        }
        ex[mm++] = q;
        m = mm;
    }
    for (int j = m - 1; j >= 0; j--) { /* largest component last */
        if (ex[j] > 0) return 1;
        if (ex[j] < 0) return -1;
    }
    return 0;
}

/* Orientation::index(p1,p2,q): GEOS CGAlgorithmsDD::orientationIndex = fast filter
 * (orientationIndexFilter, DP_SAFE_EPSILON 1e-15) then extended precision.  Here the
 * fallback is the EXACT sign of  ax*by - ax*cy - cx*by - ay*bx + ay*cx + cy*bx. */
int orc_orient(double ax, double ay, double bx, double by, double cx, double cy) {
    double detleft = (ax - cx) * (by - cy);
    double detright = (ay - cy) * (bx - cx);
    double det = detleft - detright;
    double detsum;
    int ok = 0;
    if (detleft > 0.0) {
        if (detright <= 0.0) ok = 1; else detsum = detleft + detright;
    } else if (detleft < 0.0) {
        if (detright >= 0.0) ok = 1; else detsum = -detleft - detright;
    } else ok = 1;
    if (!ok) {
        double errbound = 1e-15 * detsum;
        if (det >= errbound || -det >= errbound) ok = 1;
    }
    if (ok) return det > 0 ? 1 : (det < 0 ? -1 : 0);
    double t[12];
    two_prod(ax, by, &t[0], &t[1]);
    two_prod(-ax, cy, &t[2], &t[3]);
    two_prod(-cx, by, &t[4], &t[5]);
    two_prod(-ay, bx, &t[6], &t[7]);
    two_prod(ay, cx, &t[8], &t[9]);
    two_prod(cy, bx, &t[10], &t[11]);
    return exact_sum_sign(t, 12);
}

/* RobustLineIntersector::computeIntersect(p1,p2,q1,q2) != NO_INTERSECTION */
int orc_segments_intersect(double p1x, double p1y, double p2x, double p2y, double q1x, double q1y,
                           double q2x, double q2y) {
    /* Envelope::intersects(p1,p2,q1,q2) */
    double minq = fmin(q1x, q2x), maxq = fmax(q1x, q2x), minp = fmin(p1x, p2x), maxp = fmax(p1x, p2x);
    if (minp > maxq) return 0;
    if (maxp < minq) return 0;
    minq = fmin(q1y, q2y); maxq = fmax(q1y, q2y); minp = fmin(p1y, p2y); maxp = fmax(p1y, p2y);
    if (minp > maxq) return 0;
    if (maxp < minq) return 0;
    int Pq1 = orc_orient(p1x, p1y, p2x, p2y, q1x, q1y);
    int Pq2 = orc_orient(p1x, p1y, p2x, p2y, q2x, q2y);
    if ((Pq1 > 0 && Pq2 > 0) || (Pq1 < 0 && Pq2 < 0)) return 0;
    int Qp1 = orc_orient(q1x, q1y, q2x, q2y, p1x, p1y);
    int Qp2 = orc_orient(q1x, q1y, q2x, q2y, p2x, p2y);
    if ((Qp1 > 0 && Qp2 > 0) || (Qp1 < 0 && Qp2 < 0)) return 0;
    /* collinear case: envelopes already overlap -> some endpoint lies on the other segment */
    return 1;
}

/* LinearRing.intersects(LinearRing) (car_parking_base.py:156): boundaries only, touching
 * counts, containment does not.  ring = nv (3|4) vertices, implicitly closed. */
int orc_ring_intersects(const double *box, const double *ring, int nv) {
    for (int i = 0; i < 4; i++) {
        const double *a = box + 2 * i, *b = box + 2 * ((i + 1) & 3);
        for (int j = 0; j < nv; j++) {
            const double *c = ring + 2 * j, *d = ring + 2 * ((j + 1) % nv);
            if (orc_segments_intersect(a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1])) return 1;
        }
    }
    return 0;
}

/* car_parking_base.py:153-158 _detect_collision */
int orc_detect_collision(const double *box, const double *verts, const int32_t *nvert, int n_obst) {
    for (int o = 0; o < n_obst; o++)
        if (orc_ring_intersects(box, verts + 8 * o, nvert[o])) return 1;
    return 0;
}

/* GEOS Area::ofRingSigned on an OPEN list of n vertices (ring closed implicitly) */
static double ring_area_signed(const double (*p)[2], int n) {
    if (n < 3) return 0.0;
    double sum = 0.0, x0 = p[0][0];
    for (int i = 1; i < n; i++) { /* closed ring has n+1 points; loop i=1..n-1 over them */
        double x = p[i][0] - x0;
        double y1 = p[(i + 1) % n][1];
        double y2 = p[i - 1][1];
        sum += x * (y2 - y1);
    }
    return sum / 2.0;
}

double orc_quad_area(const double *q) { return fabs(ring_area_signed((const double(*)[2])q, 4)); }

/* Polygon(A).intersection(Polygon(B)).area for two convex CCW quads: Sutherland-Hodgman
 * clip of A by the 4 half-planes of B, then the shoelace above.  (car_parking_base.py:164-170,
 * :217-219.)  GEOS overlay is absent: UNPINNED, continuous quantity. */
double orc_quad_intersection_area(const double *A, const double *B) {
    double poly[16][2], tmp[16][2];
    int n = 4;
    for (int i = 0; i < 4; i++) { poly[i][0] = A[2 * i]; poly[i][1] = A[2 * i + 1]; }
    for (int e = 0; e < 4 && n > 0; e++) {
        double c1x = B[2 * e], c1y = B[2 * e + 1];
        double c2x = B[2 * ((e + 1) & 3)], c2y = B[2 * ((e + 1) & 3) + 1];
        double ex = c2x - c1x, ey = c2y - c1y;
        int m = 0;
        for (int i = 0; i < n; i++) {
            const double *s = poly[i], *t = poly[(i + 1) % n];
            double ds = ex * (s[1] - c1y) - ey * (s[0] - c1x);
            double dt = ex * (t[1] - c1y) - ey * (t[0] - c1x);
            int sin_ = ds >= 0, tin = dt >= 0;
            if (sin_) { tmp[m][0] = s[0]; tmp[m][1] = s[1]; m++; }
            if (sin_ != tin) {
                double r = ds / (ds - dt);
                tmp[m][0] = s[0] + r * (t[0] - s[0]);
                tmp[m][1] = s[1] + r * (t[1] - s[1]);
                m++;
            }
        }
        n = m;
        memcpy(poly, tmp, sizeof(double) * 2 * n);
    }
    if (n < 3) return 0.0;
    return fabs(ring_area_signed((const double(*)[2])poly, n));
}

/* GEOS Distance::pointToSegment */
static double pt_seg_dist(double px, double py, double ax, double ay, double bx, double by) {
    if (ax == bx && ay == by) return sqrt((px - ax) * (px - ax) + (py - ay) * (py - ay));
    double len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);
    double r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2;
    if (r <= 0.0) return sqrt((px - ax) * (px - ax) + (py - ay) * (py - ay));
    if (r >= 1.0) return sqrt((px - bx) * (px - bx) + (py - by) * (py - by));
    double s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2;
    return fabs(s) * sqrt(len2);
}

double orc_pt_seg_dist(double px, double py, double ax, double ay, double bx, double by) {
    return pt_seg_dist(px, py, ax, ay, bx, by);
}

/* ===================================================================================== */
/* lidar                                                                                 */
/* ===================================================================================== */
/* lidar_simulator.py:74-135 _fast_calc_lidar_obs.  rings: ego-frame vertices [n][4][2],
 * nvert[n] in {3,4}.  Literal sequence of masked assignments. */
void orc_lidar_fast(const double *rv, const int32_t *nvert, int n_rings, double *out) {
    orc_init();
    int E = 0;
    for (int r = 0; r < n_rings; r++) E += nvert[r];
    if (E == 0) {
        for (int i = 0; i < NBEAM; i++) out[i] = 1.0 * LIDAR_RANGE;
        return;
    }
    const double tmp_inf = 100, tmp_zero = 1e-8, c = 0;
    for (int i = 0; i < NBEAM; i++) {
        double a = g_beam_a[i], b = g_beam_b[i];
        double best = INFINITY;
        for (int r = 0; r < n_rings; r++) {
            int nv = nvert[r];
            for (int j = 0; j < nv; j++) {
                double x1 = rv[8 * r + 2 * j], y1 = rv[8 * r + 2 * j + 1];
                double x2 = rv[8 * r + 2 * ((j + 1) % nv)], y2 = rv[8 * r + 2 * ((j + 1) % nv) + 1];
                double d = y2 - y1, e = x1 - x2, f = y1 * x2 - x1 * y2;
                double det = a * e - b * d;
                int parallel = (det == 0);
                if (parallel) det = 1;
                double raw_x = (b * f - c * e) / det;
                double raw_y = (c * d - a * f) / det;
                if (i < NBEAM / 4 && raw_x < -tmp_zero) raw_x = tmp_inf;
                if (i >= NBEAM / 4 * 3 && raw_x < -tmp_zero) raw_x = tmp_inf;
                if (i >= NBEAM / 4 && i < NBEAM / 4 * 3 && raw_x > tmp_zero) raw_x = tmp_inf;
                if (i < NBEAM / 2 && raw_y < -tmp_zero) raw_y = tmp_inf;
                if (i >= NBEAM / 2 && raw_y > tmp_zero) raw_y = tmp_inf;
                if (raw_x > fmax(x1, x2)) raw_x = tmp_inf;
                if (raw_x < fmin(x1, x2)) raw_x = tmp_inf;
                if (raw_y > fmax(y1, y2)) raw_y = tmp_inf;
                if (raw_y < fmin(y1, y2)) raw_y = tmp_inf;
                if (parallel) raw_x = tmp_inf;
                double dist = sqrt(raw_x * raw_x + raw_y * raw_y);
                if (dist < best) best = dist;
            }
        }
        out[i] = clipd(best, 0, LIDAR_RANGE);
    }
}

/* lidar_simulator.py:31-72 get_observation = rotate+filter (:55-72), fast calc, minus hull */
void orc_lidar_observation(const double *pose, const double *verts, const int32_t *nvert, int n_obst,
                           double *out) {
    orc_init();
    double x = pose[0], y = pose[1], theta = pose[2];
    double a = hm_cos(theta), b = hm_sin(theta);
    double x_off = -x * a - y * b;
    double y_off = x * b - y * a;
    double *rv = (double *)malloc(sizeof(double) * 8 * (n_obst > 0 ? n_obst : 1));
    int32_t *nv = (int32_t *)malloc(sizeof(int32_t) * (n_obst > 0 ? n_obst : 1));
    int kept = 0;
    for (int o = 0; o < n_obst; o++) {
        double r[4][2];
        int n = nvert[o];
        for (int v = 0; v < n; v++) {
            double px = verts[8 * o + 2 * v], py = verts[8 * o + 2 * v + 1];
            r[v][0] = a * px + b * py + x_off;       /* affine [a, b, -b, a, xoff, yoff] */
            r[v][1] = (-b) * px + a * py + y_off;
        }
        double dmin = INFINITY;
        for (int v = 0; v < n; v++) {
            double dd = pt_seg_dist(0, 0, r[v][0], r[v][1], r[(v + 1) % n][0], r[(v + 1) % n][1]);
            if (dd < dmin) dmin = dd;
        }
        if (dmin < LIDAR_RANGE) {
            for (int v = 0; v < n; v++) { rv[8 * kept + 2 * v] = r[v][0]; rv[8 * kept + 2 * v + 1] = r[v][1]; }
            nv[kept] = n;
            kept++;
        }
    }
    double raw[NBEAM];
    orc_lidar_fast(rv, nv, kept, raw);
    for (int i = 0; i < NBEAM; i++) out[i] = raw[i] - g_hull_base[i];
    free(rv);
    free(nv);
}

/* ===================================================================================== */
/* action mask                                                                           */
/* ===================================================================================== */
/* scipy.ndimage.minimum_filter1d(x, 5) default mode='reflect', origin 0, on int array */
static void min_filter5_reflect(const long *x, int n, long *out) {
    for (int i = 0; i < n; i++) {
        long m = x[i];
        for (int k = -2; k <= 2; k++) {
            int j = i + k;
            if (j < 0) j = -j - 1;
            if (j >= n) j = 2 * n - 1 - j;
            if (x[j] < m) m = x[j];
        }
        out[i] = m;
    }
}

/* action_mask.py:166-196 get_steps + post_process.  hull_base: the array ADDED to the scan
 * (self.vehicle_lidar_base); dist_star: [1200][42][10].  NULL -> the oracle's own tables. */
void orc_get_steps(const double *raw_scan, const double *hull_base, const double *dist_star, double *out) {
    orc_init();
    if (!hull_base) hull_base = g_hull_base;
    if (!dist_star) dist_star = g_dist_star;
    double lo[NBEAM + 1], dist_obs[NL];
    for (int i = 0; i < NBEAM; i++) lo[i] = clipd(raw_scan[i], 0, 10) + hull_base[i];
    lo[NBEAM] = lo[0];
    for (int j = 0; j < NL; j++) {
        double w2 = (double)(j % UPS) / UPS, w1 = 1 - w2;
        dist_obs[j] = lo[j / UPS] * w1 + lo[j / UPS + 1] * w2;
    }
    long step_len[NACT];
    /* np.min over the 1200 beams of max_step (:178): every beam contributes at most n_iter, so the minimum starts there; and a beam
     * whose scan value is >= every entry of its 420-entry block has all step_save flags set (max_step = n_iter for all 42 actions)
     * and cannot lower it -- its block is not read.  Same result, entry for entry; without the skip every scene-step streamed the
     * whole 4 MB table, and the OpenMP batch loop (bench.py's all-core CPU baseline) stopped scaling at 32 threads on the 128-core
     * host (profiles/r06_cpu_baseline_threads.txt). */
    const double *ds_max = (dist_star == g_dist_star && g_ds_max_of == g_dist_star) ? g_ds_max : 0;
    for (int a = 0; a < NACT; a++) step_len[a] = NITER;
    for (int l = 0; l < NL; l++) {
        if (ds_max && dist_obs[l] >= ds_max[l]) continue;
        for (int a = 0; a < NACT; a++) {
            const double *ds = dist_star + ((size_t)l * NACT + a) * NITER;
            int sum = 0, first0 = -1;
            for (int k = 0; k < NITER; k++) {
                int s = ds[k] <= dist_obs[l]; /* step_save */
                sum += s;
                if (!s && first0 < 0) first0 = k;
            }
            long max_step = (first0 < 0) ? 0 : first0; /* argmin: first 0; all-1 -> argmin 0 */
            if (sum == NITER) max_step = NITER;
            if (max_step < step_len[a]) step_len[a] = max_step;
        }
    }
    /* post_process */
    long fwd[NACT / 2], bwd[NACT / 2], f2[NACT / 2], b2[NACT / 2];
    for (int i = 0; i < NACT / 2; i++) { fwd[i] = step_len[i]; bwd[i] = step_len[NACT / 2 + i]; }
    fwd[0] -= 1; fwd[NACT / 2 - 1] -= 1; bwd[0] -= 1; bwd[NACT / 2 - 1] -= 1;
    min_filter5_reflect(fwd, NACT / 2, f2);
    min_filter5_reflect(bwd, NACT / 2, b2);
    double sum = 0;
    for (int i = 0; i < NACT; i++) {
        long v = i < NACT / 2 ? f2[i] : b2[i - NACT / 2];
        if (v < 0) v = 0;
        if (v > NITER) v = NITER;
        out[i] = (double)v / NITER;
        sum += out[i];
    }
    if (sum == 0)
        for (int i = 0; i < NACT; i++) out[i] = clipd(out[i], 0.01, 1);
}

/* ===================================================================================== */
/* Reeds-Shepp  (src/env/reeds_shepp.py)                                                  */
/* ===================================================================================== */
#define RS_MAXP 64
enum { C_S = 0, C_L = 1, C_R = 2 };
typedef struct {
    int n;
    int ct[5];
    double len[5];
    double L;
} rs_word;
typedef struct { rs_word w[RS_MAXP]; int n; } rs_set;

/* Python float %: result takes the sign of the divisor */
static double py_mod(double v, double w) {
    double m = hm_fmod(v, w);
    if (m != 0) { if ((w < 0) != (m < 0)) m += w; } else m = copysign(0.0, w);
    return m;
}
static double rs_M(double theta) { /* :581-592 */
    double phi = py_mod(theta, 2.0 * PI);
    if (phi < -PI) phi += 2.0 * PI;
    if (phi > PI) phi -= 2.0 * PI;
    return phi;
}
static void rs_R(double x, double y, double *r, double *th) { *r = hm_hypot(x, y); *th = hm_atan2(y, x); } /* :571 */
static double pi_2_pi(double t) { /* :561-568 */
    while (t > PI) t -= 2.0 * PI;
    while (t < -PI) t += 2.0 * PI;
    return t;
}

/* :57-76 set_path */
static void set_path(rs_set *ps, const double *lengths, const int *ct, int n) {
    for (int e = 0; e < ps->n; e++) {
        const rs_word *pe = &ps->w[e];
        if (pe->n != n) continue;
        int same = 1;
        for (int i = 0; i < n; i++) if (pe->ct[i] != ct[i]) same = 0;
        if (same) {
            double s = 0; /* builtin sum(): 0 + d0 + d1 ... */
            for (int i = 0; i < n; i++) s = s + (pe->len[i] - lengths[i]);
            if (s <= 0.01) return;
        }
    }
    double L = 0;
    for (int i = 0; i < n; i++) L = L + fabs(lengths[i]);
    if (L >= 1000.0) return; /* MAX_LENGTH */
    /* assert path.L >= 0.001 (reference raises AssertionError) -- caller never gets here at a goal pose */
    if (ps->n >= RS_MAXP) return;
    rs_word *w = &ps->w[ps->n++];
    w->n = n;
    for (int i = 0; i < n; i++) { w->ct[i] = ct[i]; w->len[i] = lengths[i]; }
    w->L = L;
}

static int rs_SLS(double x, double y, double phi, double *t, double *u, double *v) { /* :133-149 */
    phi = rs_M(phi);
    if (y > 0.0 && 0.0 < phi && phi < PI * 0.99) {
        double xd = -y / hm_tan(phi) + x;
        *t = xd - hm_tan(phi / 2.0);
        *u = phi;
        *v = sqrt((x - xd) * (x - xd) + y * y) - hm_tan(phi / 2.0);
        return 1;
    } else if (y < 0.0 && 0.0 < phi && phi < PI * 0.99) {
        double xd = -y / hm_tan(phi) + x;
        *t = xd - hm_tan(phi / 2.0);
        *u = phi;
        *v = -sqrt((x - xd) * (x - xd) + y * y) - hm_tan(phi / 2.0);
        return 1;
    }
    return 0;
}
static int rs_LSL(double x, double y, double phi, double *t, double *u, double *v) { /* :79-87 */
    double uu, tt;
    rs_R(x - hm_sin(phi), y - 1.0 + hm_cos(phi), &uu, &tt);
    if (tt >= 0.0) {
        double vv = rs_M(phi - tt);
        if (vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
static int rs_LSR(double x, double y, double phi, double *t, double *u, double *v) { /* :90-103 */
    double u1, t1;
    rs_R(x + hm_sin(phi), y - 1.0 - hm_cos(phi), &u1, &t1);
    u1 = u1 * u1;
    if (u1 >= 4.0) {
        double uu = sqrt(u1 - 4.0);
        double theta = hm_atan2(2.0, uu);
        double tt = rs_M(t1 + theta);
        double vv = rs_M(tt - phi);
        if (tt >= 0.0 && vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
static int rs_LRL(double x, double y, double phi, double *t, double *u, double *v) { /* :106-117 */
    double u1, t1;
    rs_R(x - hm_sin(phi), y - 1.0 + hm_cos(phi), &u1, &t1);
    if (u1 <= 4.0) {
        double uu = -2.0 * hm_asin(0.25 * u1);
        double tt = rs_M(t1 + 0.5 * uu + PI);
        double vv = rs_M(phi - tt + uu);
        if (tt >= 0.0 && uu <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
static void calc_tauOmega(double u, double v, double xi, double eta, double phi, double *tau, double *omega) {
    double delta = rs_M(u - v); /* :228-243 */
    double A = hm_sin(u) - hm_sin(delta);
    double B = hm_cos(u) - hm_cos(delta) - 1.0;
    double t1 = hm_atan2(eta * A - xi * B, xi * A + eta * B);
    double t2 = 2.0 * (hm_cos(delta) - hm_cos(v) - hm_cos(u)) + 3.0;
    if (t2 < 0) *tau = rs_M(t1 + PI); else *tau = rs_M(t1);
    *omega = rs_M(*tau - u + v - phi);
}
static int rs_LRLRn(double x, double y, double phi, double *t, double *u, double *v) { /* :246-257 */
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi);
    double rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (rho <= 1.0) {
        double uu = hm_acos(rho), tt, vv;
        calc_tauOmega(uu, -uu, xi, eta, phi, &tt, &vv);
        if (tt >= 0.0 && vv <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
static int rs_LRLRp(double x, double y, double phi, double *t, double *u, double *v) { /* :260-272 */
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi);
    double rho = (20.0 - xi * xi - eta * eta) / 16.0;
    if (0.0 <= rho && rho <= 1.0) {
        double uu = -hm_acos(rho);
        if (uu >= -0.5 * PI) {
            double tt, vv;
            calc_tauOmega(uu, uu, xi, eta, phi, &tt, &vv);
            if (tt >= 0.0 && vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
        }
    }
    return 0;
}
static int rs_LRSR(double x, double y, double phi, double *t, double *u, double *v) { /* :311-323 */
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi), rho, theta;
    rs_R(-eta, xi, &rho, &theta);
    if (rho >= 2.0) {
        double tt = theta, uu = 2.0 - rho, vv = rs_M(tt + 0.5 * PI - phi);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
static int rs_LRSL(double x, double y, double phi, double *t, double *u, double *v) { /* :326-339 */
    double xi = x - hm_sin(phi), eta = y - 1.0 + hm_cos(phi), rho, theta;
    rs_R(xi, eta, &rho, &theta);
    if (rho >= 2.0) {
        double r = sqrt(rho * rho - 4.0);
        double uu = 2.0 - r;
        double tt = rs_M(theta + hm_atan2(r, -2.0));
        double vv = rs_M(phi - 0.5 * PI - tt);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
static int rs_LRSLR(double x, double y, double phi, double *t, double *u, double *v) { /* :414-429 */
    double xi = x + hm_sin(phi), eta = y - 1.0 - hm_cos(phi), rho, theta;
    rs_R(xi, eta, &rho, &theta);
    if (rho >= 2.0) {
        double uu = 4.0 - sqrt(rho * rho - 4.0);
        if (uu <= 0.0) {
            double tt = rs_M(hm_atan2((4.0 - uu) * xi - 2.0 * eta, -2.0 * xi + (uu - 4.0) * eta));
            double vv = rs_M(tt - phi);
            if (tt >= 0.0 && vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
        }
    }
    return 0;
}

#define SP3(a0, a1, a2, c0, c1, c2) do { double l_[3] = {a0, a1, a2}; int c_[3] = {c0, c1, c2}; set_path(ps, l_, c_, 3); } while (0)
#define SP4(a0, a1, a2, a3, c0, c1, c2, c3) do { double l_[4] = {a0, a1, a2, a3}; int c_[4] = {c0, c1, c2, c3}; set_path(ps, l_, c_, 4); } while (0)
#define SP5(a0, a1, a2, a3, a4, c0, c1, c2, c3, c4) do { double l_[5] = {a0, a1, a2, a3, a4}; int c_[5] = {c0, c1, c2, c3, c4}; set_path(ps, l_, c_, 5); } while (0)

/* :540-557 generate_path (+ SCS :120, CSC :152, CCC :188, CCCC :275, CCSC :342, CCSCC :432) */
static void rs_generate_path(const double *q0, const double *q1, double maxc, rs_set *ps) {
    double dx = q1[0] - q0[0], dy = q1[1] - q0[1], dth = q1[2] - q0[2];
    double c = hm_cos(q0[2]), s = hm_sin(q0[2]);
    double x = (c * dx + s * dy) * maxc;
    double y = (-s * dx + c * dy) * maxc;
    double phi = dth, t, u, v;
    const int S = C_S, L = C_L, R = C_R;
    const double hp = 0.5 * PI;
    ps->n = 0;
    /* SCS */
    if (rs_SLS(x, y, phi, &t, &u, &v)) SP3(t, u, v, S, L, S);
    if (rs_SLS(x, -y, -phi, &t, &u, &v)) SP3(t, u, v, S, R, S);
    /* CSC */
    if (rs_LSL(x, y, phi, &t, &u, &v)) SP3(t, u, v, L, S, L);
    if (rs_LSL(-x, y, -phi, &t, &u, &v)) SP3(-t, -u, -v, L, S, L);
    if (rs_LSL(x, -y, -phi, &t, &u, &v)) SP3(t, u, v, R, S, R);
    if (rs_LSL(-x, -y, phi, &t, &u, &v)) SP3(-t, -u, -v, R, S, R);
    if (rs_LSR(x, y, phi, &t, &u, &v)) SP3(t, u, v, L, S, R);
    if (rs_LSR(-x, y, -phi, &t, &u, &v)) SP3(-t, -u, -v, L, S, R);
    if (rs_LSR(x, -y, -phi, &t, &u, &v)) SP3(t, u, v, R, S, L);
    if (rs_LSR(-x, -y, phi, &t, &u, &v)) SP3(-t, -u, -v, R, S, L);
    /* CCC */
    if (rs_LRL(x, y, phi, &t, &u, &v)) SP3(t, u, v, L, R, L);
    if (rs_LRL(-x, y, -phi, &t, &u, &v)) SP3(-t, -u, -v, L, R, L);
    if (rs_LRL(x, -y, -phi, &t, &u, &v)) SP3(t, u, v, R, L, R);
    if (rs_LRL(-x, -y, phi, &t, &u, &v)) SP3(-t, -u, -v, R, L, R);
    {
        double xb = x * hm_cos(phi) + y * hm_sin(phi);
        double yb = x * hm_sin(phi) - y * hm_cos(phi);
        if (rs_LRL(xb, yb, phi, &t, &u, &v)) SP3(v, u, t, L, R, L);
        if (rs_LRL(-xb, yb, -phi, &t, &u, &v)) SP3(-v, -u, -t, L, R, L);
        if (rs_LRL(xb, -yb, -phi, &t, &u, &v)) SP3(v, u, t, R, L, R);
        if (rs_LRL(-xb, -yb, phi, &t, &u, &v)) SP3(-v, -u, -t, R, L, R);
    }
    /* CCCC */
    if (rs_LRLRn(x, y, phi, &t, &u, &v)) SP4(t, u, -u, v, L, R, L, R);
    if (rs_LRLRn(-x, y, -phi, &t, &u, &v)) SP4(-t, -u, u, -v, L, R, L, R);
    if (rs_LRLRn(x, -y, -phi, &t, &u, &v)) SP4(t, u, -u, v, R, L, R, L);
    if (rs_LRLRn(-x, -y, phi, &t, &u, &v)) SP4(-t, -u, u, -v, R, L, R, L);
    if (rs_LRLRp(x, y, phi, &t, &u, &v)) SP4(t, u, u, v, L, R, L, R);
    if (rs_LRLRp(-x, y, -phi, &t, &u, &v)) SP4(-t, -u, -u, -v, L, R, L, R);
    if (rs_LRLRp(x, -y, -phi, &t, &u, &v)) SP4(t, u, u, v, R, L, R, L);
    if (rs_LRLRp(-x, -y, phi, &t, &u, &v)) SP4(-t, -u, -u, -v, R, L, R, L);
    /* CCSC */
    if (rs_LRSL(x, y, phi, &t, &u, &v)) SP4(t, -hp, u, v, L, R, S, L);
    if (rs_LRSL(-x, y, -phi, &t, &u, &v)) SP4(-t, hp, -u, -v, L, R, S, L);
    if (rs_LRSL(x, -y, -phi, &t, &u, &v)) SP4(t, -hp, u, v, R, L, S, R);
    if (rs_LRSL(-x, -y, phi, &t, &u, &v)) SP4(-t, hp, -u, -v, R, L, S, R);
    if (rs_LRSR(x, y, phi, &t, &u, &v)) SP4(t, -hp, u, v, L, R, S, R);
    if (rs_LRSR(-x, y, -phi, &t, &u, &v)) SP4(-t, hp, -u, -v, L, R, S, R);
    if (rs_LRSR(x, -y, -phi, &t, &u, &v)) SP4(t, -hp, u, v, R, L, S, L);
    if (rs_LRSR(-x, -y, phi, &t, &u, &v)) SP4(-t, hp, -u, -v, R, L, S, L);
    {
        double xb = x * hm_cos(phi) + y * hm_sin(phi);
        double yb = x * hm_sin(phi) - y * hm_cos(phi);
        if (rs_LRSL(xb, yb, phi, &t, &u, &v)) SP4(v, u, -hp, t, L, S, R, L);
        if (rs_LRSL(-xb, yb, -phi, &t, &u, &v)) SP4(-v, -u, hp, -t, L, S, R, L);
        if (rs_LRSL(xb, -yb, -phi, &t, &u, &v)) SP4(v, u, -hp, t, R, S, L, R);
        if (rs_LRSL(-xb, -yb, phi, &t, &u, &v)) SP4(-v, -u, hp, -t, R, S, L, R);
        if (rs_LRSR(xb, yb, phi, &t, &u, &v)) SP4(v, u, -hp, t, R, S, R, L);
        if (rs_LRSR(-xb, yb, -phi, &t, &u, &v)) SP4(-v, -u, hp, -t, R, S, R, L);
        if (rs_LRSR(xb, -yb, -phi, &t, &u, &v)) SP4(v, u, -hp, t, L, S, L, R);
        if (rs_LRSR(-xb, -yb, phi, &t, &u, &v)) SP4(-v, -u, hp, -t, L, S, L, R);
    }
    /* CCSCC */
    if (rs_LRSLR(x, y, phi, &t, &u, &v)) SP5(t, -hp, u, -hp, v, L, R, S, L, R);
    if (rs_LRSLR(-x, y, -phi, &t, &u, &v)) SP5(-t, hp, -u, hp, -v, L, R, S, L, R);
    if (rs_LRSLR(x, -y, -phi, &t, &u, &v)) SP5(t, -hp, u, -hp, v, R, L, S, R, L);
    if (rs_LRSLR(-x, -y, phi, &t, &u, &v)) SP5(-t, hp, -u, hp, -v, R, L, S, R, L);
}

/* :510-537 interpolate */
static void rs_interpolate(int ind, double l, int m, double maxc, double ox, double oy, double oyaw,
                           double *px, double *py, double *pyaw, int *dir) {
    if (m == C_S) {
        px[ind] = ox + l / maxc * hm_cos(oyaw);
        py[ind] = oy + l / maxc * hm_sin(oyaw);
        pyaw[ind] = oyaw;
    } else {
        double ldx = hm_sin(l) / maxc, ldy = 0;
        if (m == C_L) ldy = (1.0 - hm_cos(l)) / maxc;
        else if (m == C_R) ldy = (1.0 - hm_cos(l)) / (-maxc);
        double gdx = hm_cos(-oyaw) * ldx + hm_sin(-oyaw) * ldy;
        double gdy = -hm_sin(-oyaw) * ldx + hm_cos(-oyaw) * ldy;
        px[ind] = ox + gdx;
        py[ind] = oy + gdy;
    }
    if (m == C_L) pyaw[ind] = oyaw + l;
    else if (m == C_R) pyaw[ind] = oyaw - l;
    dir[ind] = l > 0.0 ? 1 : -1;
}

/* Per-thread scratch for the Reeds-Shepp sample arrays (stack discipline: mark at entry, release at exit).  A search makes up to
 * ~60 arrays of up to 13 k points; with malloc / free per array the OpenMP batch loop of bench.py's all-core CPU baseline spent its
 * time in the allocator's locks on a 128-core host (10.5x on 128 threads).  Chunks are kept for the thread's life. */
#define ARENA_CHUNKS 64
static __thread struct { char *chunk[ARENA_CHUNKS]; size_t cap[ARENA_CHUNKS]; int cur; size_t used; } g_arena;
typedef struct { int cur; size_t used; } arena_mark_t;
static arena_mark_t arena_mark(void) { arena_mark_t m = {g_arena.cur, g_arena.used}; return m; }
static void arena_release(arena_mark_t m) { g_arena.cur = m.cur; g_arena.used = m.used; }
static void *arena_alloc(size_t bytes, int zero) {
    bytes = (bytes + 63) & ~(size_t)63;
    for (;;) {
        int c = g_arena.cur;
        if (g_arena.chunk[c] && g_arena.used + bytes <= g_arena.cap[c]) {
            void *q = g_arena.chunk[c] + g_arena.used;
            g_arena.used += bytes;
            if (zero) memset(q, 0, bytes);
            return q;
        }
        if (g_arena.chunk[c]) {
            if (c + 1 >= ARENA_CHUNKS) abort();                 /* (64 chunks of >= 1 MB: a search needs ~6 MB) */
            g_arena.cur = ++c; g_arena.used = 0;
        }
        if (!g_arena.chunk[c] || g_arena.cap[c] < bytes) {      /* a fresh (or too small, unused) slot */
            size_t cap = bytes > ((size_t)1 << 20) ? bytes : ((size_t)1 << 20);
            free(g_arena.chunk[c]);
            g_arena.chunk[c] = (char *)malloc(cap);
            g_arena.cap[c] = cap;
            g_arena.used = 0;
        }
    }
}

/* :452-507 generate_local_course.  Returns the number of points kept; arrays from the thread's arena. */
static int rs_local_course(double L, const double *lengths, const int *mode, int nseg, double maxc,
                           double step_size, double **opx, double **opy, double **opyaw, int **odir) {
    int point_num = (int)(L / step_size) + nseg + 3;
    double *px = (double *)arena_alloc(point_num * sizeof(double), 1);
    double *py = (double *)arena_alloc(point_num * sizeof(double), 1);
    double *pyaw = (double *)arena_alloc(point_num * sizeof(double), 1);
    int *dir = (int *)arena_alloc(point_num * sizeof(int), 1);
    int ind = 1;
    dir[0] = lengths[0] > 0.0 ? 1 : -1;
    double d = lengths[0] > 0.0 ? step_size : -step_size;
    double pd = d, ll = 0.0;
    for (int i = 0; i < nseg; i++) {
        int m = mode[i];
        double l = lengths[i];
        d = l > 0.0 ? step_size : -step_size;
        double ox = px[ind], oy = py[ind], oyaw = pyaw[ind];
        ind -= 1;
        if (i >= 1 && (lengths[i - 1] * lengths[i]) > 0) pd = -d - ll; else pd = d - ll;
        while (fabs(pd) <= fabs(l)) {
            ind += 1;
            rs_interpolate(ind, pd, m, maxc, ox, oy, oyaw, px, py, pyaw, dir);
            pd += d;
        }
        ll = l - pd - d;
        ind += 1;
        rs_interpolate(ind, l, m, maxc, ox, oy, oyaw, px, py, pyaw, dir);
    }
    int n = point_num;
    while (n > 0 && px[n - 1] == 0.0) n--; /* "remove unused data" */
    *opx = px; *opy = py; *opyaw = pyaw; *odir = dir;
    return n;
}

typedef struct {
    rs_word w;   /* lengths/L already divided by maxc (metres) */
    int npts;
    double *x, *y, *yaw;
    int *dir;
} rs_path;


/* :35-54 calc_all_paths */
static int rs_calc_all_paths(const double *q0, const double *q1, double maxc, double step_size, rs_path *out) {
    rs_set ps;
    rs_generate_path(q0, q1, maxc, &ps);
    for (int i = 0; i < ps.n; i++) {
        rs_path *p = &out[i];
        p->w = ps.w[i];
        double *lx, *ly, *lyaw;
        int *ldir;
        int n = rs_local_course(p->w.L, p->w.len, p->w.ct, p->w.n, maxc, step_size * maxc, &lx, &ly, &lyaw, &ldir);
        p->npts = n;
        p->x = (double *)arena_alloc(sizeof(double) * (n > 0 ? n : 1), 0);
        p->y = (double *)arena_alloc(sizeof(double) * (n > 0 ? n : 1), 0);
        p->yaw = lyaw;
        p->dir = ldir;
        for (int k = 0; k < n; k++) {
            p->x[k] = hm_cos(-q0[2]) * lx[k] + hm_sin(-q0[2]) * ly[k] + q0[0];
            p->y[k] = -hm_sin(-q0[2]) * lx[k] + hm_cos(-q0[2]) * ly[k] + q0[1];
            p->yaw[k] = pi_2_pi(lyaw[k] + q0[2]);
        }
        for (int k = 0; k < p->w.n; k++) p->w.len[k] = p->w.len[k] / maxc;
        p->w.L = p->w.L / maxc;
    }
    return ps.n;
}

/* flat export for tests: per path (n, ctypes[5], lengths[5], L, npts, first3[9], last3[9], sums[4]) */
int orc_rs_all_paths(const double *q0, const double *q1, double maxc, double step_size, int max_paths,
                     int32_t *nseg, int32_t *ctypes, double *lengths, double *L, int32_t *npts,
                     double *first3, double *last3, double *sums) {
    rs_path P[RS_MAXP];
    const arena_mark_t am_ = arena_mark();
    int n = rs_calc_all_paths(q0, q1, maxc, step_size, P);
    for (int i = 0; i < n && i < max_paths; i++) {
        nseg[i] = P[i].w.n;
        for (int k = 0; k < 5; k++) {
            ctypes[5 * i + k] = k < P[i].w.n ? P[i].w.ct[k] : -1;
            lengths[5 * i + k] = k < P[i].w.n ? P[i].w.len[k] : 0.0;
        }
        L[i] = P[i].w.L;
        npts[i] = P[i].npts;
        int np_ = P[i].npts, m = np_ < 3 ? np_ : 3;
        for (int k = 0; k < 9; k++) { first3[9 * i + k] = 0; last3[9 * i + k] = 0; }
        for (int k = 0; k < m; k++) {
            first3[9 * i + 3 * k] = P[i].x[k]; first3[9 * i + 3 * k + 1] = P[i].y[k]; first3[9 * i + 3 * k + 2] = P[i].yaw[k];
            int s = np_ - m + k, r = 3 - m + k;
            last3[9 * i + 3 * r] = P[i].x[s]; last3[9 * i + 3 * r + 1] = P[i].y[s]; last3[9 * i + 3 * r + 2] = P[i].yaw[s];
        }
        double sx = 0, sy = 0, sw = 0, sd = 0;
        for (int k = 0; k < np_; k++) { sx += P[i].x[k]; sy += P[i].y[k]; sw += P[i].yaw[k]; sd += P[i].dir[k]; }
        sums[4 * i] = sx; sums[4 * i + 1] = sy; sums[4 * i + 2] = sw; sums[4 * i + 3] = sd;
    }
    (void)n;
    arena_release(am_);
    return n;
}

/* full sample export of one path (index pi) -- for debugging / trajectory tests */
int orc_rs_path_samples(const double *q0, const double *q1, double maxc, double step_size, int pi,
                        int cap, double *xyz) {
    rs_path P[RS_MAXP];
    const arena_mark_t am_ = arena_mark();
    int n = rs_calc_all_paths(q0, q1, maxc, step_size, P);
    int np_ = -1;
    if (pi < n) {
        np_ = P[pi].npts;
        for (int k = 0; k < np_ && k < cap; k++) { xyz[3 * k] = P[pi].x[k]; xyz[3 * k + 1] = P[pi].y[k]; xyz[3 * k + 2] = P[pi].yaw[k]; }
    }
    (void)n;
    arena_release(am_);
    return np_;
}

/* car_parking_base.py:452-534 is_traj_valid.  traj [T][3]; obstacles verts [n][4][2] + nvert;
 * bbox = xmin,xmax,ymin,ymax.  Returns 1 valid / 0 invalid. */
int orc_is_traj_valid(const double *traj, int T, const double *verts, const int32_t *nvert, int n_obst,
                      const double *bbox) {
    double car[4][2];
    vehicle_box_local(car);
    if (T <= 0) return 1;
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (int t = 0; t < T; t++) {
        mnx = fmin(mnx, traj[3 * t]); mxx = fmax(mxx, traj[3 * t]);
        mny = fmin(mny, traj[3 * t + 1]); mxy = fmax(mxy, traj[3 * t + 1]);
    }
    if (mnx < bbox[0] || mxx > bbox[1] || mny < bbox[2] || mxy > bbox[3]) return 0;
    /* hull edges per pose: (corner k -> corner k+1), car_coords1=coords[:4], car_coords2=coords[1:] */
    const arena_mark_t vm_ = arena_mark();
    double *vx1 = (double *)arena_alloc(sizeof(double) * 4 * T, 0), *vy1 = (double *)arena_alloc(sizeof(double) * 4 * T, 0);
    double *vx2 = (double *)arena_alloc(sizeof(double) * 4 * T, 0), *vy2 = (double *)arena_alloc(sizeof(double) * 4 * T, 0);
    double x_max = -INFINITY, x_min = INFINITY, y_max = -INFINITY, y_min = INFINITY;
    for (int t = 0; t < T; t++) {
        double ct = hm_cos(traj[3 * t + 2]), st = hm_sin(traj[3 * t + 2]);
        double vx = traj[3 * t], vy = traj[3 * t + 1];
        for (int k = 0; k < 4; k++) {
            int k2 = (k + 1) & 3;
            vx1[4 * t + k] = ct * car[k][0] - st * car[k][1] + vx;
            vy1[4 * t + k] = st * car[k][0] + ct * car[k][1] + vy;
            vx2[4 * t + k] = ct * car[k2][0] - st * car[k2][1] + vx;
            vy2[4 * t + k] = st * car[k2][0] + ct * car[k2][1] + vy;
            x_max = fmax(x_max, vx1[4 * t + k]); x_min = fmin(x_min, vx1[4 * t + k]);
            y_max = fmax(y_max, vy1[4 * t + k]); y_min = fmin(y_min, vy1[4 * t + k]);
        }
    }
    x_max += 5; x_min -= 5; y_max += 5; y_min -= 5;
    int collide = 0, n_edges = 0;
    for (int o = 0; o < n_obst && !collide; o++) {
        int nv = nvert[o];
        const double *r = verts + 8 * o;
        int all_gx = 1, all_lx = 1, all_gy = 1, all_ly = 1;
        for (int v = 0; v < nv; v++) {
            if (!(r[2 * v] > x_max)) all_gx = 0;
            if (!(r[2 * v] < x_min)) all_lx = 0;
            if (!(r[2 * v + 1] > y_max)) all_gy = 0;
            if (!(r[2 * v + 1] < y_min)) all_ly = 0;
        }
        if (all_gx || all_lx || all_gy || all_ly) continue;
        n_edges += nv;
        for (int j = 0; j < nv && !collide; j++) {
            double x1 = r[2 * j], y1 = r[2 * j + 1], x2 = r[2 * ((j + 1) % nv)], y2 = r[2 * ((j + 1) % nv) + 1];
            double d = y2 - y1, e = x1 - x2, f = y1 * x2 - x1 * y2;
            double exmax = fmax(x1, x2), exmin = fmin(x1, x2), eymax = fmax(y1, y2), eymin = fmin(y1, y2);
            for (int q = 0; q < 4 * T; q++) {
                double a = vy2[q] - vy1[q], b = vx1[q] - vx2[q], c = vy1[q] * vx2[q] - vx1[q] * vy2[q];
                double det = a * e - b * d;
                if (det == 0) continue;
                double raw_x = (b * f - c * e) / det;
                double raw_y = (c * d - a * f) / det;
                int cx = 1, cy = 1;
                if (raw_x > exmax) cx = 0;
                if (raw_x < exmin) cx = 0;
                if (raw_y > eymax) cy = 0;
                if (raw_y < eymin) cy = 0;
                if (raw_x > fmax(vx1[q], vx2[q])) cx = 0;
                if (raw_x < fmin(vx1[q], vx2[q])) cx = 0;
                if (raw_y > fmax(vy1[q], vy2[q])) cy = 0;
                if (raw_y < fmin(vy1[q], vy2[q])) cy = 0;
                if (cx && cy) { collide = 1; break; }
            }
        }
    }
    arena_release(vm_);
    (void)n_edges;
    return collide ? 0 : 1;
}

/* heapdict 1.0.1 (third-party, unpinned in requirements.txt) restated: array heap of
 * (priority, id); __setitem__ = append + _decrease_key (non-strict parent test),
 * popitem = move last to root + _min_heapify (strict tests). */
typedef struct { double pr[RS_MAXP]; int id[RS_MAXP]; int n; } hd_t;
static void hd_swap(hd_t *h, int i, int j) {
    double p = h->pr[i]; h->pr[i] = h->pr[j]; h->pr[j] = p;
    int k = h->id[i]; h->id[i] = h->id[j]; h->id[j] = k;
}
static void hd_push(hd_t *h, double pr, int id) {
    int i = h->n++;
    h->pr[i] = pr; h->id[i] = id;
    while (i) {
        int parent = (i - 1) >> 1;
        if (h->pr[parent] < h->pr[i]) break;
        hd_swap(h, i, parent);
        i = parent;
    }
}
static int hd_pop(hd_t *h) {
    int top = h->id[0];
    if (h->n == 1) { h->n = 0; return top; }
    h->n--;
    h->pr[0] = h->pr[h->n]; h->id[0] = h->id[h->n];
    int i = 0, n = h->n;
    for (;;) {
        int l = (i << 1) + 1, r = (i + 1) << 1, low;
        if (l < n && h->pr[l] < h->pr[i]) low = l; else low = i;
        if (r < n && h->pr[r] < h->pr[low]) low = r;
        if (low == i) break;
        hd_swap(h, i, low);
        i = low;
    }
    return top;
}

/* car_parking_base.py:413-450 find_rs_path.  Returns 1 if a collision-free path was found. */
int orc_find_rs_path(const double *pose, const double *dest, const double *verts, const int32_t *nvert,
                     int n_obst, const double *bbox, int32_t *out_nseg, int32_t *out_ct, double *out_len,
                     double *out_L, int32_t *out_ntested) {
    /* math.tan(VALID_STEER[-1]) / WHEEL_BASE (car_parking_base.py:422) as Python's math (glibc) evaluates it; the
     * value is a constant of the path, kept literal so that both math flavours and the kernels share it */
    const double radius = 0.3327130214085973;
    rs_path P[RS_MAXP];
    const arena_mark_t am_ = arena_mark();
    int n = rs_calc_all_paths(pose, dest, radius, 0.1, P);
    int found = 0, ntested = 0;
    if (n > 0) {
        hd_t h;
        h.n = 0;
        for (int i = 0; i < n; i++) hd_push(&h, P[i].w.L, i);
        double min_path_len = -1;
        int idx = 0;
        while (h.n != 0) {
            idx += 1;
            int pi = hd_pop(&h);
            if (min_path_len < 0) min_path_len = P[pi].w.L;
            if (P[pi].w.L > 1.6 * min_path_len && idx > 2) break;
            int T = P[pi].npts;
            const arena_mark_t tm_ = arena_mark();
            double *traj = (double *)arena_alloc(sizeof(double) * 3 * (T > 0 ? T : 1), 0);
            for (int k = 0; k < T; k++) { traj[3 * k] = P[pi].x[k]; traj[3 * k + 1] = P[pi].y[k]; traj[3 * k + 2] = P[pi].yaw[k]; }
            int ok = orc_is_traj_valid(traj, T, verts, nvert, n_obst, bbox);
            arena_release(tm_);
            ntested++;
            if (ok) {
                found = 1;
                *out_nseg = P[pi].w.n;
                for (int k = 0; k < 5; k++) {
                    out_ct[k] = k < P[pi].w.n ? P[pi].w.ct[k] : -1;
                    out_len[k] = k < P[pi].w.n ? P[pi].w.len[k] : 0.0;
                }
                *out_L = P[pi].w.L;
                break;
            }
        }
    }
    if (!found) {
        *out_nseg = 0;
        for (int k = 0; k < 5; k++) { out_ct[k] = -1; out_len[k] = 0.0; }
        *out_L = 0.0;
    }
    if (out_ntested) *out_ntested = ntested;
    (void)n;
    arena_release(am_);
    return found;
}

/* ===================================================================================== */
/* observation pieces, reward, wrapper                                                   */
/* ===================================================================================== */
/* car_parking_base.py:372-381 _get_targt_repr (5th entry is cos again, :380) */
void orc_target_repr(const double *ego, const double *dest, double *out) {
    double rel_distance = sqrt((dest[0] - ego[0]) * (dest[0] - ego[0]) + (dest[1] - ego[1]) * (dest[1] - ego[1]));
    double rel_angle = hm_atan2(dest[1] - ego[1], dest[0] - ego[0]) - ego[2];
    double rel_dest_heading = dest[2] - ego[2];
    out[0] = rel_distance;
    out[1] = hm_cos(rel_angle);
    out[2] = hm_sin(rel_angle);
    out[3] = hm_cos(rel_dest_heading);
    out[4] = hm_cos(rel_dest_heading);
}

static double angle_diff(double a1, double a2) { /* :203-206 */
    double d = hm_acos(hm_cos(a1 - a2));
    return d < PI / 2 ? d : PI - d;
}
static double pdist(double ax, double ay, double bx, double by) { /* Point.distance */
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}

/* car_parking_base.py:186-227 _get_reward, with the overlay areas supplied by the caller
 * (union_area = |hull ∩ dest|, dest_area = |dest|).  accum in/out. */
void orc_reward_terms(const double *prev, const double *cur, const double *dest, const double *start,
                      double t, double union_area, double dest_area, double *accum, double *out) {
    double time_cost = -hm_tanh(t / (10 * TOLERANT_TIME));
    double rs_dist_reward = 0; /* REWARD_WEIGHT['rs_dist_reward'] == 0 (:192) */
    double dist_diff = pdist(cur[0], cur[1], dest[0], dest[1]);
    double ang = angle_diff(cur[2], dest[2]);
    double prev_dist_diff = pdist(prev[0], prev[1], dest[0], dest[1]);
    double prev_ang = angle_diff(prev[2], dest[2]);
    double dist_norm_ratio = fmax(pdist(dest[0], dest[1], start[0], start[1]), 10);
    double angle_norm_ratio = PI;
    double dist_reward = prev_dist_diff / dist_norm_ratio - dist_diff / dist_norm_ratio;
    double angle_reward = prev_ang / angle_norm_ratio - ang / angle_norm_ratio;
    double box_union_reward = union_area / (2 * dest_area - union_area);
    if (box_union_reward < *accum) box_union_reward = 0;
    else {
        double prev_arrive = *accum;
        *accum = box_union_reward;
        box_union_reward -= prev_arrive;
    }
    out[0] = time_cost; out[1] = rs_dist_reward; out[2] = dist_reward; out[3] = angle_reward; out[4] = box_union_reward;
}

/* env_wrapper.py:10-35 reward_shaping */
double orc_reward_shaping(const double *info, int status) {
    static const double W[5] = {1, 0, 5, 0, 10}; /* configs.py:183-187 */
    double reward = 0;
    if (status == ST_CONTINUE) {
        for (int i = 0; i < 5; i++) reward += W[i] * info[i];
    } else if (status == ST_OUTBOUND) reward = -50;
    else if (status == ST_OUTTIME) reward = -1;
    else if (status == ST_ARRIVED) reward = 50;
    else if (status == ST_COLLIDED) reward = -50;
    reward *= 0.1; /* REWARD_RATIO */
    return reward;
}

/* env_wrapper.py:37-50 action_rescale (float64 evaluation; epsilon = 0) */
void orc_action_rescale(const double *act, double *out) {
    const double lo[2] = {VALID_STEER_LO, VALID_SPEED_LO}, hi[2] = {VALID_STEER_HI, VALID_SPEED_HI};
    for (int i = 0; i < 2; i++) {
        double a = clipd(act[i], -1, 1);
        out[i] = a * (hi[i] - lo[i]) / 2 + (hi[i] + lo[i]) / 2;
    }
}

/* ===================================================================================== */
/* whole env step  (car_parking_base.py:235-299 + env_wrapper.py:73-81)                   */
/* ===================================================================================== */
typedef struct {
    /* scene (constant over an episode) */
    int32_t n_obst;
    const double *verts;   /* [n_obst][4][2] world frame */
    const int32_t *nvert;  /* [n_obst] */
    double start[3], dest[3];
    double bbox[4];        /* xmin, xmax, ymin, ymax */
    /* episode state */
    double pose[3];
    double t;
    double accum_arrive_reward;
} orc_scene;

typedef struct {
    double lidar[NBEAM];
    double mask[NACT];
    double target[5];
    double reward_info[5];
    double reward;      /* shaped (wrapper) */
    int32_t status;
    int32_t done;
    int32_t rs_found;
    int32_t rs_nseg;
    int32_t rs_ctypes[5];
    double rs_lengths[5];
    double rs_L;
    int32_t substeps;   /* sub-steps kept (diagnostic) */
    int32_t collided_substep;
} orc_obs;

static int detect_outbound(const orc_scene *s) { /* :160-162 */
    double x = s->pose[0], y = s->pose[1];
    return x > s->bbox[1] || x < s->bbox[0] || y > s->bbox[3] || y < s->bbox[2];
}
static int check_arrived(const orc_scene *s, const double *box, const double *dest_box, double *area_out) { /* :164-170 */
    double ua = orc_quad_intersection_area(box, dest_box);
    if (area_out) *area_out = ua;
    return ua / orc_quad_area(dest_box) > 0.95;
}

/* action: physical (steer, speed) or NULL (reset's action-less step). */
void orc_env_step_physical(orc_scene *s, const double *action, orc_obs *o, int with_rs) {
    orc_init();
    double prev_state[3] = {s->pose[0], s->pose[1], s->pose[2]};
    double dest_box[8], box[8];
    orc_create_box(s->dest, dest_box);
    int arrive = 0;
    o->substeps = 0;
    o->collided_substep = -1;
    if (action) {
        for (int k = 0; k < NUM_STEP; k++) {
            double prev_info[3] = {s->pose[0], s->pose[1], s->pose[2]};
            orc_ks_step(s->pose, action, 0);
            orc_create_box(s->pose, box);
            o->substeps = k + 1;
            if (check_arrived(s, box, dest_box, 0)) { arrive = 1; break; }
            if (orc_detect_collision(box, s->verts, s->nvert, s->n_obst)) {
                s->pose[0] = prev_info[0]; s->pose[1] = prev_info[1]; s->pose[2] = prev_info[2]; /* retreat */
                o->substeps = k;
                o->collided_substep = k;
                break;
            }
        }
    }
    s->t += 1;
    /* render(): lidar, action mask, target (car_parking_base.py:399-407) */
    orc_lidar_observation(s->pose, s->verts, s->nvert, s->n_obst, o->lidar);
    orc_get_steps(o->lidar, 0, 0, o->mask);
    orc_target_repr(s->pose, s->dest, o->target);
    int status;
    orc_create_box(s->pose, box);
    if (arrive) status = ST_ARRIVED;
    else { /* _check_status :175-184 */
        if (orc_detect_collision(box, s->verts, s->nvert, s->n_obst)) status = ST_COLLIDED;
        else if (detect_outbound(s)) status = ST_OUTBOUND;
        else if (check_arrived(s, box, dest_box, 0)) status = ST_ARRIVED;
        else if (s->t > TOLERANT_TIME) status = ST_OUTTIME;
        else status = ST_CONTINUE;
    }
    for (int i = 0; i < 5; i++) o->reward_info[i] = 0;
    if (status == ST_CONTINUE) {
        double ua = orc_quad_intersection_area(box, dest_box);
        orc_reward_terms(prev_state, s->pose, s->dest, s->start, s->t, ua, orc_quad_area(dest_box),
                         &s->accum_arrive_reward, o->reward_info);
    }
    o->rs_found = 0;
    o->rs_nseg = 0;
    o->rs_L = 0;
    for (int k = 0; k < 5; k++) { o->rs_ctypes[k] = -1; o->rs_lengths[k] = 0; }
    if (with_rs && s->t > 1 && status == ST_CONTINUE &&
        pdist(s->pose[0], s->pose[1], s->dest[0], s->dest[1]) < RS_MAX_DIST) {
        o->rs_found = orc_find_rs_path(s->pose, s->dest, s->verts, s->nvert, s->n_obst, s->bbox, &o->rs_nseg,
                                       o->rs_ctypes, o->rs_lengths, &o->rs_L, 0);
    }
    o->status = status;
    o->reward = orc_reward_shaping(o->reward_info, status);
    o->done = status != ST_CONTINUE;
}

/* CarParkingWrapper.step: action in [-1,1]^2 -> rescale -> step -> shaped reward */
void orc_env_step(orc_scene *s, const double *action, orc_obs *o, int with_rs) {
    if (action) {
        double phys[2];
        orc_action_rescale(action, phys);
        orc_env_step_physical(s, phys, o, with_rs);
    } else orc_env_step_physical(s, 0, o, with_rs);
}

/* CarParking.reset (:127-138): zero accumulators, pose = start, then the action-less step */
void orc_env_reset(orc_scene *s, orc_obs *o) {
    s->accum_arrive_reward = 0.0;
    s->t = 0.0;
    s->pose[0] = s->start[0]; s->pose[1] = s->start[1]; s->pose[2] = s->start[2];
    orc_env_step(s, 0, o, 1);
}

/* ---- flat batch interface (ctypes; also the timed cpu_baseline loop) -------------------
 * Scenes are stored with a fixed stride of max_obst obstacles. */
void orc_batch_step(int n, int max_obst, const int32_t *n_obst, const double *verts, const int32_t *nvert,
                    const double *start, const double *dest, const double *bbox, double *pose, double *t,
                    double *accum, const double *actions /* [n][2] in [-1,1] or NULL */, int with_rs,
                    double *lidar, double *mask, double *target, double *reward_info, double *reward,
                    int32_t *status, int32_t *rs_found, int32_t *rs_ctypes, double *rs_lengths,
                    int32_t *substeps) {
    orc_init();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 2)
#endif
    for (int i = 0; i < n; i++) {
        orc_scene s;
        orc_obs o;
        s.n_obst = n_obst[i];
        s.verts = verts + (size_t)i * max_obst * 8;
        s.nvert = nvert + (size_t)i * max_obst;
        memcpy(s.start, start + 3 * i, 24);
        memcpy(s.dest, dest + 3 * i, 24);
        memcpy(s.bbox, bbox + 4 * i, 32);
        memcpy(s.pose, pose + 3 * i, 24);
        s.t = t[i];
        s.accum_arrive_reward = accum[i];
        orc_env_step(&s, actions ? actions + 2 * i : 0, &o, with_rs);
        memcpy(pose + 3 * i, s.pose, 24);
        t[i] = s.t;
        accum[i] = s.accum_arrive_reward;
        if (lidar) memcpy(lidar + (size_t)NBEAM * i, o.lidar, sizeof(o.lidar));
        if (mask) memcpy(mask + (size_t)NACT * i, o.mask, sizeof(o.mask));
        if (target) memcpy(target + 5 * i, o.target, sizeof(o.target));
        if (reward_info) memcpy(reward_info + 5 * i, o.reward_info, sizeof(o.reward_info));
        if (reward) reward[i] = o.reward;
        if (status) status[i] = o.status;
        if (rs_found) rs_found[i] = o.rs_found;
        if (rs_ctypes) memcpy(rs_ctypes + 5 * i, o.rs_ctypes, sizeof(o.rs_ctypes));
        if (rs_lengths) memcpy(rs_lengths + 5 * i, o.rs_lengths, sizeof(o.rs_lengths));
        if (substeps) substeps[i] = o.substeps;
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* the OpenMP build's team size for the following batch calls (bench.py's all-core baseline: every CPU the process may use) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* elementary-function test hook: fn 0 sin, 1 cos, 2 tan, 3 atan2(a,b), 4 asin, 5 acos, 6 hypot(a,b), 7 fmod(a,b), 8 tanh,
 * 9 exp, 10 sqrt, 11 a/b */
void orc_math(int fn, int n, const double *a, const double *b, double *out) {
    for (int i = 0; i < n; i++) {
        double x = a[i], y = b ? b[i] : 0.0;
        switch (fn) {
            case 0: out[i] = hm_sin(x); break;
            case 1: out[i] = hm_cos(x); break;
            case 2: out[i] = hm_tan(x); break;
            case 3: out[i] = hm_atan2(x, y); break;
            case 4: out[i] = hm_asin(x); break;
            case 5: out[i] = hm_acos(x); break;
            case 6: out[i] = hm_hypot(x, y); break;
            case 7: out[i] = hm_fmod(x, y); break;
            case 8: out[i] = hm_tanh(x); break;
            case 9: out[i] = hm_exp(x); break;
            case 10: out[i] = sqrt(x); break;
            default: out[i] = x / y; break;
        }
    }
}
