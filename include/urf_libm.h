/*
 * urf_libm.h -- the three libm functions on the urban_road_filter hot path,
 * as ONE shared source for the CPU oracle and the gfx950 kernels.
 *
 * The reference calls glibc's float acos/asin/atan2 at
 *   src/lidar_segmentation.cpp:162,165   (vertical angle:  acos / asin)
 *   src/lidar_segmentation.cpp:256-268   (azimuth:         asin)
 *   src/x_zero_method.cpp:58             (triangle angle:  acos)
 *   src/z_zero_method.cpp:63             (vector angle:    acos)
 *   src/star_shaped_search.cpp:166       (polar angle:     atan2)
 * (all `float` overloads).  glibc's float versions are
 * not correctly rounded, differ between releases and do not exist on the GPU,
 * so this project DEFINES the three functions as "evaluate in binary64 with
 * the fixed operation sequence below, round once to binary32".  Every
 * operation is an IEEE-754 basic operation (+ - * / sqrt fma), written out
 * explicitly, so host (gcc/clang, any -O level, with or without -mfma) and
 * device (hipcc, gfx950) produce bit-identical results.  Measured against
 * glibc 2.35 the float results agree to <= 1 ulp (tests/test_libm.py).
 *
 * Coefficients: tools/gen_libm_coeffs.py (Chebyshev interpolation at 60
 * digits; max relative error 9e-18 (asin) / 1.2e-17 (atan) before the final
 * rounding, i.e. the float result is the correctly rounded one except for
 * arguments within ~1e-9 ulp of a rounding boundary).
 *
 * Plain C99 / C++17 / HIP.  No dependency on <math.h>.
 */
#ifndef URF_LIBM_H
#define URF_LIBM_H

#if defined(__HIPCC__)
#define URF_HD __host__ __device__
#else
#define URF_HD
#endif

#define URF_PI_D      0x1.921fb54442d18p+1 /* (double)pi   == M_PI   */
#define URF_PIO2_D    0x1.921fb54442d18p+0 /* (double)pi/2 == M_PI_2 */
#define URF_PIO4_D    0x1.921fb54442d18p-1 /* (double)pi/4 == M_PI_4 */
#define URF_SQRT2M1_D 0x1.a827999fcef32p-2 /* sqrt(2) - 1 */

/* s + s*w*P(w) with w = s*s in [0, 0.25]:  asin(s) for s in [0, 0.5]. */
static inline URF_HD double urf__asin_poly(double s, double w)
{
    double p = 0x1.d72b2bc8155f8p-6;
    p = __builtin_fma(p, w, -0x1.e6aaa8a0a04ccp-7);
    p = __builtin_fma(p, w, 0x1.1d189408314eep-6);
    p = __builtin_fma(p, w, 0x1.65a9c4dfcf8b2p-8);
    p = __builtin_fma(p, w, 0x1.52420b04b37bep-7);
    p = __builtin_fma(p, w, 0x1.782651caa6547p-7);
    p = __builtin_fma(p, w, 0x1.c9cf07674736ap-7);
    p = __builtin_fma(p, w, 0x1.1c4d35cf95421p-6);
    p = __builtin_fma(p, w, 0x1.6e8bb1c8209a2p-6);
    p = __builtin_fma(p, w, 0x1.f1c71c1db0623p-6);
    p = __builtin_fma(p, w, 0x1.6db6db6e31f13p-5);
    p = __builtin_fma(p, w, 0x1.3333333332ecap-4);
    p = __builtin_fma(p, w, 0x1.5555555555556p-3);
    double sw = s * w;
    return __builtin_fma(sw, p, s);
}

/* t + t*v*Q(v) with v = t*t, |t| <= sqrt(2)-1:  atan(t). */
static inline URF_HD double urf__atan_poly(double t)
{
    double v = t * t;
    double q = -0x1.3a2b7a07caea9p-6;
    q = __builtin_fma(q, v, 0x1.41603647c7a7cp-5);
    q = __builtin_fma(q, v, -0x1.a098bb6ba4941p-5);
    q = __builtin_fma(q, v, 0x1.dfe61e80903d2p-5);
    q = __builtin_fma(q, v, -0x1.10fa75382537fp-4);
    q = __builtin_fma(q, v, 0x1.3b1262d95579ep-4);
    q = __builtin_fma(q, v, -0x1.745d0b26b83e7p-4);
    q = __builtin_fma(q, v, 0x1.c71c7185314cbp-4);
    q = __builtin_fma(q, v, -0x1.24924924360cbp-3);
    q = __builtin_fma(q, v, 0x1.999999999934ap-3);
    q = __builtin_fma(q, v, -0x1.5555555555555p-2);
    double tv = t * v;
    return __builtin_fma(tv, q, t);
}

/* asin on [0,1] in binary64.  One polynomial evaluation serves both ranges
 * (a <= 0.5 directly, else through asin(a) = pi/2 - 2 asin(sqrt((1-a)/2))), so a
 * wave whose lanes fall on both sides does not pay for two. */
static inline URF_HD double urf__asin01(double a)
{
    const int small = a <= 0.5;
    double w = a * a, s = a;
    if (!small) {                       /* a real branch: a wave of small arguments skips the sqrt */
        w = (1.0 - a) * 0.5;            /* exact */
        s = __builtin_sqrt(w);
    }
    const double r = urf__asin_poly(s, w);
    return small ? r : __builtin_fma(-2.0, r, URF_PIO2_D);
}

/* replaces glibc asinf (float overload of asin) */
static inline URF_HD float urf_asinf(float x)
{
    double a = __builtin_fabs((double)x);
    if (!(a <= 1.0))
        return __builtin_nanf("");
    double r = urf__asin01(a);
    return (float)(x < 0.0f ? -r : r);
}

/* replaces glibc acosf (float overload of acos) */
static inline URF_HD float urf_acosf(float x)
{
    const double a = __builtin_fabs((double)x);
    if (!(a <= 1.0))
        return __builtin_nanf("");
    const int small = a <= 0.5;
    double w = a * a, s = a;
    if (!small) {
        w = (1.0 - a) * 0.5;            /* exact */
        s = __builtin_sqrt(w);
    }
    const double p = urf__asin_poly(s, w);
    double r;
    if (small)
        r = x < 0.0f ? URF_PIO2_D + p : URF_PIO2_D - p;   /* pi/2 -+ asin(|x|) */
    else {
        const double t = 2.0 * p;                         /* acos(|x|) = 2 asin(sqrt((1-|x|)/2)) */
        r = x < 0.0f ? URF_PI_D - t : t;
    }
    return (float)r;
}

/* replaces glibc atan2f (float overload of atan2); result in (-pi, pi] */
static inline URF_HD float urf_atan2f(float y, float x)
{
    if (x != x || y != y)
        return __builtin_nanf("");
    double ax = __builtin_fabs((double)x), ay = __builtin_fabs((double)y);
    double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    double r;
    if (mx == 0.0) {
        r = 0.0;
    } else {
        if (mx > 0x1.fffffffffffffp+1023) {              /* infinite operand */
            mn = (mn > 0x1.fffffffffffffp+1023) ? 1.0 : 0.0;
            mx = 1.0;
        }
        /* one division either way: atan(mn/mx) directly, or, above tan(pi/8),
         * pi/4 + atan((mn-mx)/(mn+mx)) */
        const int big = mn > mx * URF_SQRT2M1_D;
        const double num = big ? mn - mx : mn;
        const double den = big ? mn + mx : mx;
        const double t = num / den;
        r = urf__atan_poly(t);
        if (big)
            r = URF_PIO4_D + r;
        if (ay > ax)
            r = URF_PIO2_D - r;
    }
    /* sign of x: note -0.0f counts as negative, as in IEEE atan2 */
    if (__builtin_signbit(x))
        r = URF_PI_D - r;
    if (__builtin_signbit(y))
        r = -r;
    return (float)r;
}

#endif /* URF_LIBM_H */
