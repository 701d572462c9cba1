/*
 * hope_math.h -- deterministic float64 elementary functions, shared by the HIP kernels and the CPU oracle.
 *
 * Why: the reference's Reeds-Shepp search is ill-conditioned in two structural ways (equal-length twin words;
 * tolerance-free hits on exactly axis-aligned edges, see tests/rs_illcond.py), so two correct implementations that
 * use different libm's (OCML on the GPU, glibc in the oracle, numpy's SIMD loops in the reference) disagree on
 * ~3e-5 of the searches purely through the last bit of sin/cos/atan2.  These functions use ONLY operations IEEE-754
 * defines exactly (+ - * / sqrt, rint, trunc, fabs, copysign, ldexp, fma), in a fixed order, with contraction off
 * (-ffp-contract=off on both compilers; every fused multiply-add is an EXPLICIT fma(), v_fma_f64 on gfx950 and
 * vfmadd / glibc's correctly rounded fma() on x86-64): the same source gives the same bits on both, so device and
 * oracle agree bit-for-bit and every remaining difference is a bug.  The polynomials are Horner chains of fma: half
 * the instructions of separate multiplies and adds, and one rounding less per step.
 *
 * Accuracy: <= ~2 ulp on the ranges the env uses (|x| < 1e5 for sin/cos); validated against libm in
 * tests/test_math.py.  They are NOT bit-identical to any libm -- nothing can be, the reference's own numbers
 * depend on its platform's libm.  Constants from tools/gen_math_consts.py (exact rational / 90-digit arithmetic).
 */
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define HM_FN __host__ __device__ __forceinline__
#else
#define HM_FN static inline
#endif

#define HM_PI 3.141592653589793
#define HM_PI_LO 1.2246467991473532e-16
#define HM_PIO2 1.5707963267948966
#define HM_PIO2_LO 6.123233995736766e-17
#define HM_PIO4 0.7853981633974483

/* ---- sin / cos --------------------------------------------------------------------------------------------- */
/* kernels on |r| <= pi/4 (Taylor to r^21 / r^22: truncation < 1e-19) */
HM_FN double hm_ksin(double r) {
    const double z = r * r;
    double p = 1.9572941063391263e-20;
    p = fma(p, z, -8.22063524662433e-18);
    p = fma(p, z, 2.8114572543455206e-15);
    p = fma(p, z, -7.647163731819816e-13);
    p = fma(p, z, 1.6059043836821613e-10);
    p = fma(p, z, -2.505210838544172e-08);
    p = fma(p, z, 2.7557319223985893e-06);
    p = fma(p, z, -0.0001984126984126984);
    p = fma(p, z, 0.008333333333333333);
    p = fma(p, z, -0.16666666666666666);
    return fma(r, z * p, r);
}
HM_FN double hm_kcos(double r) {
    const double z = r * r;
    double p = -8.896791392450574e-22;
    p = fma(p, z, 4.110317623312165e-19);
    p = fma(p, z, -1.5619206968586225e-16);
    p = fma(p, z, 4.779477332387385e-14);
    p = fma(p, z, -1.1470745597729725e-11);
    p = fma(p, z, 2.08767569878681e-09);
    p = fma(p, z, -2.755731922398589e-07);
    p = fma(p, z, 2.48015873015873e-05);
    p = fma(p, z, -0.001388888888888889);
    p = fma(p, z, 0.041666666666666664);
    const double hz = 0.5 * z;
    return fma(z, z * p, 1.0 - hz);
}
/* |x| = k*pi/2 + r, |r| <= pi/4 (Cody-Waite, pi/2 in three parts; k*part exact for |k| < 2^20) */
HM_FN void hm_sincos(double x, double* s, double* c) {
    const double ax = fabs(x);
    const double kd = rint(ax * 0.6366197723675814);
    double r = fma(-kd, 1.5707963267341256, ax);
    r = fma(-kd, 6.077100506303966e-11, r);
    r = fma(-kd, 2.0222662487959506e-21, r);
    /* quadrant: one 32-bit conversion instruction on the GPU (v_cvt_i32_f64 SATURATES: kd >= 2^31 -> INT_MAX, NaN -> 0).  On the host
     * an out-of-range conversion is undefined behaviour (x86 gives INT_MIN), so the host spells the GPU's saturation out: both sides
     * return the same bits for EVERY input, also beyond the reduction's accurate range (|x| < 2^20 pi/2; the env's angles are < 1e3) */
#if defined(__HIP_DEVICE_COMPILE__)
    const int q = (int)kd & 3;
#else
    const int q = (kd != kd ? 0 : (kd >= 2147483647.0 ? 2147483647 : (int)kd)) & 3;
#endif
    const double ks = hm_ksin(r), kc = hm_kcos(r);
    double ss = (q & 1) ? kc : ks;
    double cc = (q & 1) ? ks : kc;
    if (q == 2 || q == 3) ss = -ss;
    if (q == 1 || q == 2) cc = -cc;
    *s = x < 0 ? -ss : ss;          /* exactly odd / even */
    *c = cc;
}
HM_FN double hm_sin(double x) { double s, c; hm_sincos(x, &s, &c); return s; }
HM_FN double hm_cos(double x) { double s, c; hm_sincos(x, &s, &c); return c; }
HM_FN double hm_tan(double x) { double s, c; hm_sincos(x, &s, &c); return s / c; }

/* ---- atan / atan2 / asin / acos ------------------------------------------------------------------------------ */
/* atan(t) for |t| <= 0.2679 (Taylor to t^31) */
HM_FN double hm_katan(double t) {
    const double z = t * t;
    double p = -0.03225806451612903;
    p = fma(p, z, 0.034482758620689655);
    p = fma(p, z, -0.037037037037037035);
    p = fma(p, z, 0.04);
    p = fma(p, z, -0.043478260869565216);
    p = fma(p, z, 0.047619047619047616);
    p = fma(p, z, -0.05263157894736842);
    p = fma(p, z, 0.058823529411764705);
    p = fma(p, z, -0.06666666666666667);
    p = fma(p, z, 0.07692307692307693);
    p = fma(p, z, -0.09090909090909091);
    p = fma(p, z, 0.1111111111111111);
    p = fma(p, z, -0.14285714285714285);
    p = fma(p, z, 0.2);
    p = fma(p, z, -0.3333333333333333);
    return fma(t, z * p, t);
}
/* atan(a) for a >= 0 */
HM_FN double hm_atan_pos(double a) {
    if (a > 1.0) {                                   /* atan(a) = pi/2 - atan(1/a) */
        const double b = 1.0 / a;
        double v;
        if (b < 0.25) v = hm_katan(b);
        else if (b < 0.75) v = 0.4636476090008061 + (hm_katan((b - 0.5) / (1.0 + 0.5 * b)) + 2.2698777452961687e-17);
        else v = HM_PIO4 + (hm_katan((b - 1.0) / (1.0 + b)) + 3.061616997868383e-17);
        return HM_PIO2 - (v - HM_PIO2_LO);
    }
    if (a < 0.25) return hm_katan(a);
    if (a < 0.75) return 0.4636476090008061 + (hm_katan((a - 0.5) / (1.0 + 0.5 * a)) + 2.2698777452961687e-17);
    return HM_PIO4 + (hm_katan((a - 1.0) / (1.0 + a)) + 3.061616997868383e-17);
}
HM_FN double hm_atan2(double y, double x) {
    if (x == 0.0 && y == 0.0) return copysign((copysign(1.0, x) < 0) ? HM_PI : 0.0, y);   /* C semantics for zeros */
    const double ax = fabs(x), ay = fabs(y);
    double v;
    if (ax >= ay) v = hm_atan_pos(ay / ax);          /* [0, pi/4] */
    else v = HM_PIO2 - (hm_atan_pos(ax / ay) - HM_PIO2_LO);
    if (x < 0.0 || (x == 0.0 && copysign(1.0, x) < 0)) v = HM_PI - (v - HM_PI_LO);
    return y < 0.0 || (y == 0.0 && copysign(1.0, y) < 0) ? -v : v;
}
HM_FN double hm_asin(double x) { return hm_atan2(x, sqrt((1.0 - x) * (1.0 + x))); }
HM_FN double hm_acos(double x) { return hm_atan2(sqrt((1.0 - x) * (1.0 + x)), x); }

/* ---- hypot, fmod --------------------------------------------------------------------------------------------- */
HM_FN double hm_hypot(double x, double y) { return sqrt(x * x + y * y); }   /* env magnitudes: no over/underflow */
/* C fmod for |x/y| < 2^52: the remainder x - q*y is exactly representable, fma delivers it with one rounding */
HM_FN double hm_fmod(double x, double y) {
    const double ay = fabs(y), ax = fabs(x);
    if (ax < ay) return x;
    const double q = trunc(ax / ay);
    double r = fma(-q, ay, ax);
    if (r < 0.0) r = r + ay;                         /* quotient was rounded up across an integer */
    if (r >= ay) r = r - ay;
    return x < 0 ? -r : r;
}

/* ---- exp, tanh ------------------------------------------------------------------------------------------------ */
HM_FN double hm_exp(double x) {
    const double kd = rint(x * 1.4426950408889634);
    double r = fma(-kd, 0.6931471803691238, x);
    r = fma(-kd, 1.9082149288430703e-10, r);
    r = fma(-kd, 4.275175589747649e-20, r);
    double p = 1.1470745597729725e-11;
    p = fma(p, r, 1.6059043836821613e-10);
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    const double e = 1.0 + fma(r, r * p, r);
    return ldexp(e, (int)kd);
}
HM_FN double hm_tanh(double x) {
    const double ax = fabs(x);
    double v;
    if (ax < 0.15) {                                 /* the env only needs |x| <= 0.101 (t / 2000, t <= 202) */
        const double z = ax * ax;
        double p = -0.00023912911424355248;
        p = fma(p, z, 0.000590027440945586);
        p = fma(p, z, -0.0014558343870513183);
        p = fma(p, z, 0.003592128036572481);
        p = fma(p, z, -0.008863235529902197);
        p = fma(p, z, 0.021869488536155203);
        p = fma(p, z, -0.05396825396825397);
        p = fma(p, z, 0.13333333333333333);
        p = fma(p, z, -0.3333333333333333);
        v = fma(ax, z * p, ax);
    } else if (ax > 22.0) v = 1.0;
    else {
        const double e = hm_exp(2.0 * ax);
        v = (e - 1.0) / (e + 1.0);
    }
    return x < 0 ? -v : v;
}
