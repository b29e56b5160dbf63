/*
 * urf_device.hpp -- device-side helpers: the per-point expressions of the
 * reference with its exact float/double mix, and the
 * wave64 primitives the kernels share.  gfx950 only; compile with
 * -ffp-contract=off (the reference is built without contraction).
 */
#ifndef URF_DEVICE_HPP
#define URF_DEVICE_HPP

#include <hip/hip_runtime.h>

#include "urf_internal.hpp"
#include "urf_libm.h"

#define URF_WAVE 64

__device__ __forceinline__ unsigned urf_lane() { return __lane_id(); }

/* Mask of the lanes of this wave that hold the same key.  Every lane of the
 * wave must be active.  nbits = number of key bits that can differ. */
__device__ __forceinline__ unsigned long long urf_match_any(unsigned key, unsigned nbits)
{
    unsigned long long m = ~0ull;
    for (unsigned b = 0; b < nbits; b++) {
        const bool bit = (key >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

/* Same, with two one-ballot shortcuts for the patterns an organised sweep produces in
 * firing order: every lane holds the same key (one firing = one star sector), or lane l
 * holds key0 + l (one firing = every ring once). */
__device__ __forceinline__ unsigned long long urf_match_any_fast(unsigned key, unsigned nbits)
{
    const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
    if (__ballot(key != first) == 0)
        return ~0ull;
    if (__ballot(key != first + urf_lane()) == 0)
        return 1ull << urf_lane();
    return urf_match_any(key, nbits);
}

/* The same among the lanes that hold a key (`on`); the mask of a lane without one is unspecified.
 * The shortcuts look at those lanes only: a region of interest that drops part of a firing, or
 * points that match no ring, must not push the whole wave onto the bit-by-bit path. */
__device__ __forceinline__ unsigned long long urf_match_any_on(unsigned key, bool on, unsigned nbits)
{
    const unsigned long long vm = __ballot(on);
    if (vm == 0)
        return 0ull;
    const unsigned f = (unsigned)__ffsll((long long)vm) - 1u;
    const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)key, (int)f);
    if (__ballot(on && key != first) == 0)
        return vm;
    if (__ballot(on && key != first + (urf_lane() - f)) == 0)
        return 1ull << urf_lane();
    return urf_match_any(on ? key : (1u << nbits) - 1u, nbits) & vm;   /* the all-ones key is no valid key */
}

__device__ __forceinline__ unsigned urf_popc_below(unsigned long long m)
{
    return __popcll(m & ((1ull << urf_lane()) - 1ull));
}

__device__ __forceinline__ bool urf_is_leader(unsigned long long m)
{
    return (unsigned)__ffsll((long long)m) - 1u == urf_lane();
}

/* Wave-wide minimum / maximum of an unsigned value, result in every lane.  DPP row shifts and the
 * gfx9 row broadcasts (the cross-lane data path of the VALU) instead of six ds_bpermute round trips
 * through the LDS crossbar: lanes without a source keep the identity passed as `old`. */
#define URF_DPP_STEP(op, ident, ctrl, rowmask)                                                         \
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)(ident), (int)v, (ctrl), (rowmask), 0xf, false))
__device__ __forceinline__ unsigned urf_umin(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned urf_umax(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned urf_wave_min(unsigned v)
{
    URF_DPP_STEP(urf_umin, 0xffffffffu, 0x111, 0xf);   /* row_shr:1 */
    URF_DPP_STEP(urf_umin, 0xffffffffu, 0x112, 0xf);   /* row_shr:2 */
    URF_DPP_STEP(urf_umin, 0xffffffffu, 0x114, 0xf);   /* row_shr:4 */
    URF_DPP_STEP(urf_umin, 0xffffffffu, 0x118, 0xf);   /* row_shr:8: lane 15 of a row holds the row's result */
    URF_DPP_STEP(urf_umin, 0xffffffffu, 0x142, 0xa);   /* row_bcast:15 into rows 1 and 3 */
    URF_DPP_STEP(urf_umin, 0xffffffffu, 0x143, 0xc);   /* row_bcast:31 into rows 2 and 3: lane 63 holds all */
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned urf_wave_max(unsigned v)
{
    URF_DPP_STEP(urf_umax, 0u, 0x111, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x112, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x114, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x118, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x142, 0xa);
    URF_DPP_STEP(urf_umax, 0u, 0x143, 0xc);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

/* inclusive prefix sum / prefix maximum over the lanes of the wave (same data path; the sequence
 * is the one LLVM's atomic optimizer emits for gfx9) */
__device__ __forceinline__ unsigned urf_uadd(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned urf_wave_scan_add(unsigned v)
{
    URF_DPP_STEP(urf_uadd, 0u, 0x111, 0xf);
    URF_DPP_STEP(urf_uadd, 0u, 0x112, 0xf);
    URF_DPP_STEP(urf_uadd, 0u, 0x114, 0xf);
    URF_DPP_STEP(urf_uadd, 0u, 0x118, 0xf);
    URF_DPP_STEP(urf_uadd, 0u, 0x142, 0xa);
    URF_DPP_STEP(urf_uadd, 0u, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ unsigned urf_wave_scan_max(unsigned v)
{
    URF_DPP_STEP(urf_umax, 0u, 0x111, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x112, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x114, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x118, 0xf);
    URF_DPP_STEP(urf_umax, 0u, 0x142, 0xa);
    URF_DPP_STEP(urf_umax, 0u, 0x143, 0xc);
    return v;
}

/* LDS hand-over between the lanes of ONE wave (its LDS operations execute in order): nothing but the compiler has to be held back */
__device__ __forceinline__ void urf_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* scan s occupies [off, off+len) of the CALLER's per-point arrays (x, y, z, labels); a scan longer
 * than the max_len the host sized the grids and tables for is cut there */
__device__ __forceinline__ void urf_scan_range(const urf_kargs& a, unsigned s, unsigned& off, unsigned& len)
{
    if (a.offsets) {
        off = a.offsets[s];
        len = a.offsets[s + 1] - off;
        len = len > a.max_len ? a.max_len : len;
    } else {
        off = s * a.n_per_scan;
        len = a.n_per_scan;
    }
}

/* first scratch element of scan s (urf_internal.hpp: scratch layout); urf_create keeps
 * max_batch * sstride below 2^32 */
__device__ __forceinline__ unsigned urf_sbase(const urf_kargs& a, unsigned s) { return s * a.sstride; }

/* Ring c of scan s, position p inside the ring -> index of the point in the tile-local ring-sorted
 * arrays (rx, ry, rz, rsrc), relative to the scan's scratch base.  Bisection in the ring's prefix
 * over the tiles; for the kernels off the hot path (k_ring keeps the table in LDS). */
__device__ __forceinline__ unsigned urf_ring_slot(const urf_kargs& a, unsigned s, unsigned C, unsigned c, unsigned ntiles,
                                                  unsigned p)
{
    const unsigned* P = a.rpre + ((size_t)s * C + c) * (a.tiles + 1);
    unsigned lo = 0, hi = ntiles;   /* largest t with P[t] <= p; P[0] = 0 */
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (P[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo * URF_TILE + a.rstart[((size_t)s * C + c) * a.tiles + lo] + (p - P[lo]);
}

/* ---- per-point expressions ------------------------------------------------ */

/* a / M_PI for a = (double)(float in [0, 600]): the reference's "* 180 / M_PI".
 * q = a*RN(1/pi); r = a - q*pi (exact, fma); q + r*RN(1/pi) is the correctly
 * rounded quotient (Markstein); equality with the IEEE division was checked
 * exhaustively for all 1 142 292 481 floats in [0, 600] on the host
 * (tools/check_div_pi.c) and is re-checked on the device by urf_selftest().
 * Three fma-class operations instead of a ~30-instruction f64 division. */
__device__ __forceinline__ double urf_div_pi(double a)
{
    const double rpi = 0x1.45f306dc9c883p-2;   /* RN(1/pi) */
    const double q = a * rpi;
    const double r = __builtin_fma(-q, URF_PI_D, a);
    return __builtin_fma(r, rpi, q);
}

/* lidar_segmentation.cpp:106-113 */
__device__ __forceinline__ bool urf_in_roi(const urf_params& p, float x, float y, float z)
{
    return x >= p.min_X && x <= p.max_X && y >= p.min_Y && y <= p.max_Y &&
           z >= p.min_Z && z <= p.max_Z && (x + y) + z != 0.0f;
}

/* lidar_segmentation.cpp:148-166: 3-D range in double, vertical angle in degrees */
__device__ __forceinline__ float urf_vertical_angle(float x, float y, float z)
{
    const float d = (float)__builtin_sqrt((double)x * (double)x + (double)y * (double)y + (double)z * (double)z);
    float b = __builtin_fabsf(z) / d;
    if (b < -1.0f)
        b = -1.0f;
    else if (b > 1.0f)
        b = 1.0f;
    if (z < 0.0f)
        return (float)urf_div_pi((double)(urf_acosf(b) * 180.0f));
    return (float)(urf_div_pi((double)(urf_asinf(b) * 180.0f)) + 90.0);
}

/* lidar_segmentation.cpp:245-269: planar range and azimuth in degrees
 * (0 at -y, 90 at +x, 180 at +y, 270 at -x) */
__device__ __forceinline__ float urf_azimuth(float x, float y, float* d_out)
{
    const float d = (float)__builtin_sqrt((double)x * (double)x + (double)y * (double)y);
    *d_out = d;
    float b = __builtin_fabsf(x) / d;
    if (b < -1.0f)
        b = -1.0f;
    else if (b > 1.0f)
        b = 1.0f;
    const double t = urf_div_pi((double)(urf_asinf(b) * 180.0f));
    if (x >= 0.0f && y <= 0.0f)
        return (float)t;
    if (x >= 0.0f && y > 0.0f)
        return (float)(180.0 - t);
    if (x < 0.0f && y >= 0.0f)
        return (float)(180.0 + t);
    return (float)(360.0 - t);
}

/* star_shaped_search.cpp:164-171: sector of a point (index == sectors wraps to 0,
 * where the reference dereferences a null pointer) */
__device__ __forceinline__ unsigned urf_sector(float x, float y, float Kfi, unsigned sectors)
{
    float fi = urf_atan2f(y, x);
    if (fi < 0.0f)
        fi = (float)((double)fi + 2.0 * URF_PI_D);
    const int f = (int)(fi * Kfi);
    return ((unsigned)f >= sectors) ? 0u : (unsigned)f;
}

/* ---- float fast paths ---------------------------------------------------------
 * Ring and sector of a point are DECISIONS (which table entry lies within `interval`, which
 * integer the scaled polar angle truncates to).  A float approximation of the angle with a known
 * error bound settles them whenever the approximation is farther from every decision boundary
 * than the bound; only the rare point inside such a margin takes the reference's exact
 * float/double sequence.  The result is identical by construction, the f64 square root, division
 * and polynomials are skipped for almost every point.
 *
 * urf_fast_atan2f: |result - atan2(y, x)| <= 6e-7 rad (v_rcp_f32 is 1 ulp, degree-4 minimax of
 * (atan t - t)/t^3 in t^2 is 1e-9, the rest is float rounding); urf_selftest measures it on the
 * device and tests/ assert it stays below a third of the margins used. */
#define URF_FAST_ATAN_ERR 2.0e-6f            /* claimed bound, rad */
__device__ __forceinline__ float urf_fast_atan2f(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const bool big = t > 0.41421357f;
    const float tt = big ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
    const float v = tt * tt;
    float q = -0.06451718869358958f;
    q = __builtin_fmaf(q, v, 0.10743668324830098f);
    q = __builtin_fmaf(q, v, -0.14263949753605723f);
    q = __builtin_fmaf(q, v, 0.1999954031495797f);
    q = __builtin_fmaf(q, v, -0.33333331760434554f);
    float r = __builtin_fmaf(tt * v, q, tt);
    if (big)
        r += 0.78539816339744831f;
    if (ay > ax)
        r = 1.57079632679489662f - r;
    if (x < 0.0f)
        r = 3.14159265358979324f - r;
    return y < 0.0f ? -r : r;
}

/* Vertical angle [deg] of lidar_segmentation.cpp:148-166 (angle from the -z axis), approximately.
 * The reference's own value deviates from the true angle by up to ~1.2e-7 * |z|/rho rad (it rounds
 * |z|/d to float before the acos), hence the restriction to |z| <= 4 rho; within it
 * |approx - reference| <= 3e-4 deg (URF_FAST_VALPHA_ERR, measured by urf_selftest). */
#define URF_FAST_VALPHA_ERR 3.0e-4f
/* magnitudes for which the float products, sums and reciprocals of the fast paths stay normal
 * numbers with full precision (squares within [1e-30, 1e36]) */
#define URF_FAST_MIN 1.0e-15f
#define URF_FAST_MAX 1.0e18f
__device__ __forceinline__ bool urf_fast_vertical_angle(float x, float y, float z, float* out)
{
    /* no early exit: the callers run whole waves through it and select on the result.  The
     * hardware square root (1 ulp) instead of the correctly rounded sequence (15 instructions):
     * its relative error of 1.2e-7 moves the angle by less than 4e-6 deg, far inside the margin. */
    const float rho = __builtin_amdgcn_sqrtf(x * x + y * y);
    *out = urf_fast_atan2f(rho, -z) * 57.295779513082323f;
    return (rho >= URF_FAST_MIN) & (rho <= URF_FAST_MAX) & (__builtin_fabsf(z) <= 4.0f * rho);
}

/* The ring of a point is decided without any arc tangent: u = -z / rho = cot(vertical angle) falls
 * strictly as the angle grows, so "the angle lies in [lo, hi] deg" is "u lies in [cot hi, cot lo]", and
 * k_ring_table turns every table entry's window into thresholds on u once per scan
 * (urf_ring_thresholds).  u from one v_rsq_f32 (1 ulp) and one product: relative error <= 3e-7 (three
 * roundings of x*x + y*y, half of which reaches the root), i.e. <= 3e-7 * |u| / (1 + u*u) <= 1.5e-7
 * rad = 9e-6 deg of angle; the reference's own value deviates from the true angle by up to
 * 1.2e-7 * |z| / rho rad (it rounds |z| / d to float before the acos) + half an ulp of the result,
 * hence the restriction to |u| <= 4 as before.  Together < 5e-5 deg; the margin stays
 * URF_FAST_VALPHA_ERR = 3e-4 deg (urf_selftest_fast measures |angle(u) - reference|). */
#define URF_FAST_MIN2 2.0e-30f   /* max(|x|, |y|) >= 1e-15 = URF_FAST_MIN */
#define URF_FAST_MAX2 1.0e36f
__device__ __forceinline__ bool urf_fast_cot(float x, float y, float z, float* u_out, bool* planar_ok)
{
    const float rho2 = x * x + y * y;
    const float u = -z * __builtin_amdgcn_rsqf(rho2);
    *u_out = u;
    *planar_ok = (rho2 >= URF_FAST_MIN2) & (rho2 <= URF_FAST_MAX2);   /* x, y of a magnitude the fast paths accept (NaN: false) */
    return *planar_ok & (__builtin_fabsf(u) <= URF_LUT_UMAX);
}

/* cot of an angle given in degrees, clamped to [1, 179] deg (|cot| = 57 there: beyond every u the fast
 * path accepts, so the clamp changes no decision), in binary64: tan(t), t = pi/2 - angle, from the
 * Taylor series of sin / cos at t/2 (|t/2| <= 0.78: the first neglected terms are below 1e-19) and
 * the double-angle formula.  Error ~1e-15 relative; needs 1e-7. */
__host__ __device__ __forceinline__ double urf_cot_deg(double deg)
{
    deg = deg < 1.0 ? 1.0 : (deg > 179.0 ? 179.0 : deg);
    const double h = (90.0 - deg) * (URF_PI_D / 360.0);
    const double v = h * h;
    double sn = -1.0 / 121645100408832000.0;             /* 1/19! */
    sn = __builtin_fma(sn, v, 1.0 / 355687428096000.0);   /* 1/17! */
    sn = __builtin_fma(sn, v, -1.0 / 1307674368000.0);
    sn = __builtin_fma(sn, v, 1.0 / 6227020800.0);
    sn = __builtin_fma(sn, v, -1.0 / 39916800.0);
    sn = __builtin_fma(sn, v, 1.0 / 362880.0);
    sn = __builtin_fma(sn, v, -1.0 / 5040.0);
    sn = __builtin_fma(sn, v, 1.0 / 120.0);
    sn = __builtin_fma(sn, v, -1.0 / 6.0);
    sn = __builtin_fma(sn * v, h, h);
    double cs = 1.0 / 6402373705728000.0;                 /* 1/18! */
    cs = __builtin_fma(cs, v, -1.0 / 20922789888000.0);
    cs = __builtin_fma(cs, v, 1.0 / 87178291200.0);
    cs = __builtin_fma(cs, v, -1.0 / 479001600.0);
    cs = __builtin_fma(cs, v, 1.0 / 3628800.0);
    cs = __builtin_fma(cs, v, -1.0 / 40320.0);
    cs = __builtin_fma(cs, v, 1.0 / 720.0);
    cs = __builtin_fma(cs, v, -1.0 / 24.0);
    cs = __builtin_fma(cs, v, 0.5);
    cs = __builtin_fma(-cs, v, 1.0);
    const double th = sn / cs;
    return 2.0 * th / (1.0 - th * th);
}

/* Thresholds on u for the table entry `angle` (deg) with the margin e (deg) on either side of its
 * window [angle - interval, angle + interval]:
 *   u <  .x                the entry lies surely below the point's window: fl(angle - alpha) < -interval
 *   .y <= u <= .z          the entry surely matches:                       |fl(angle - alpha)| <= interval
 *   u >  .w                the entry lies surely above the point's window: fl(angle - alpha) > interval
 * (.x < .y <= .z < .w; with a window narrower than 2 e nothing "surely matches": .y > .z) */
struct urf_ring_thr { float x, y, z, w; };
__device__ __forceinline__ urf_ring_thr urf_ring_thresholds(float angle, float interval, float e)
{
    const double a = (double)angle, iv = (double)interval, ee = (double)e;
    urf_ring_thr t;
    t.x = (float)urf_cot_deg(a + iv + ee);
    t.y = (float)urf_cot_deg(a + iv - ee);
    t.z = (float)urf_cot_deg(a - iv + ee);
    t.w = (float)urf_cot_deg(a - iv - ee);
    return t;
}

/* one of the four (which = 0..3: .x .y .z .w) */
__device__ __forceinline__ float urf_ring_threshold(float angle, float interval, float e, unsigned which)
{
    const double a = (double)angle, iv = (double)interval, ee = (double)e;
    const double arg = which == 0 ? a + iv + ee : (which == 1 ? a + iv - ee : (which == 2 ? a - iv + ee : a - iv - ee));
    return (float)urf_cot_deg(arg);
}

/* Sector of star_shaped_search.cpp:164-171 when the scaled polar angle u = fi * Kfi is clear of an
 * integer by more than `margin`; -1 = undecided.  Both error sources grow with the number of sectors:
 * the reference's own roundings (two of the angle, one of the product: <= 3 ulp of u, u < sectors) and
 * the approximation (URF_FAST_ATAN_ERR * Kfi, Kfi = sectors / 2 pi).  At the reference's 360 sectors
 * they add up to 5e-5 + 1.4e-4 < URF_FAST_SECTOR_ERR; upload_params scales the margin with
 * sectors / 360 beyond that (urf_dev_params::sector_margin), and urf_selftest_fast measures the
 * approximation's part at the configured Kfi. */
#define URF_FAST_SECTOR_ERR 2.5e-4f
__device__ __forceinline__ float urf_fast_polar(float x, float y)   /* polar angle in [0, 2 pi), approximately */
{
    const float fi = urf_fast_atan2f(y, x);
    return fi < 0.0f ? fi + 6.28318530717958648f : fi;
}
__device__ __forceinline__ int urf_fast_sector_of(float fi, float x, float y, float Kfi, unsigned sectors, float margin)
{
    const float mx = __builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y));
    const float u = fi * Kfi;
    const float f = __builtin_floorf(u);
    const float fr = u - f;
    const bool ok = (mx >= URF_FAST_MIN) & (mx <= URF_FAST_MAX) & (fr > margin) & (fr < 1.0f - margin) & (f >= 0.0f) &
                    (f < (float)sectors);
    return ok ? (int)f : -1;
}
/* the same when the caller has already checked the magnitudes: x*x + y*y within [URF_FAST_MIN2,
 * URF_FAST_MAX2] puts max(|x|, |y|) within [URF_FAST_MIN, URF_FAST_MAX]; fi >= 0 makes the floor non-negative, a NaN fails the comparisons of its fraction */
__device__ __forceinline__ int urf_fast_sector_ranged(float fi, float Kfi, unsigned sectors, float margin)
{
    const float u = fi * Kfi;
    const float f = __builtin_floorf(u);
    const float fr = u - f;
    const bool ok = (fr > margin) & (fr < 1.0f - margin) & (f < (float)sectors);
    return ok ? (int)f : -1;
}
__device__ __forceinline__ int urf_fast_sector(float x, float y, float Kfi, unsigned sectors, float margin)
{
    return urf_fast_sector_of(urf_fast_polar(x, y), x, y, Kfi, sectors, margin);
}

/* Azimuth [deg] of lidar_segmentation.cpp:245-269 (0 at -y, 90 at +x, 180 at +y, 270 at -x),
 * approximately: the polar angle (the one the sector comes from: k_split evaluates the arc tangent
 * once per point) turned by a quarter.  The reference takes asin(|x| / d) with |x| / d rounded to
 * float, which is ill-conditioned towards the x axis: its own deviation from the true angle is up
 * to 1.2e-7 * |x|/|y| rad = 3.94e-4 / delta deg, delta = the azimuth's distance from the x axis (90 /
 * 270 deg) in degrees.  The margin follows it (urf_fast_az_eps): |approx - reference| <=
 * URF_FAST_AZ_ERR + 6e-4 / delta (measured by urf_selftest_fast as a fraction of the margin), down
 * to |y| = |x| / 1024 (delta = 0.056 deg, margin 0.011 deg); only closer to the axis the users take
 * the exact sequence (urf_fast_az_ok).  Not valid across the 0/360 seam, which the users treat as
 * undecided anyway (the approximation is then within the margin of an integer). */
#define URF_FAST_AZ_ERR 5.0e-4f
#define URF_FAST_AZ_RATIO 1024.0f
__device__ __forceinline__ float urf_fast_az_eps(float az)
{
    const float delta = __builtin_fminf(__builtin_fabsf(az - 90.0f), __builtin_fabsf(az - 270.0f));
    return URF_FAST_AZ_ERR + 6.0e-4f * __builtin_amdgcn_rcpf(__builtin_fmaxf(delta, 0.04f));
}
__device__ __forceinline__ float urf_fast_azimuth_of(float fi)
{
    const float az = fi * 57.295779513082323f + 90.0f;
    return az >= 360.0f ? az - 360.0f : az;
}
/* the azimuth field of a slot record (urf_internal.hpp: URF_REC_*): az in [0, 360) -> 18-bit code and back
 * (the code's centre); the all-ones code is "unknown" */
static_assert(URF_TILE == URF_REC_SRC_MASK + 1u, "the record's source field holds an index inside a tile");
__device__ __forceinline__ unsigned urf_az_code(float az)
{
    const unsigned c = (unsigned)(az * URF_REC_AZ_SCALE);
    return c < URF_REC_AZ_UNKNOWN - 1u ? c : URF_REC_AZ_UNKNOWN - 1u;
}
__device__ __forceinline__ float urf_az_decode(unsigned code)
{
    return code == URF_REC_AZ_UNKNOWN ? URF_AZ_UNKNOWN : ((float)code + 0.5f) * URF_REC_AZ_STEP;
}
__device__ __forceinline__ bool urf_fast_az_ok(float x, float y)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    return (ay * URF_FAST_AZ_RATIO >= ax) & (ay >= URF_FAST_MIN) & (ax <= URF_FAST_MAX) & (ay <= URF_FAST_MAX);
}
__device__ __forceinline__ bool urf_fast_azimuth(float x, float y, float* out)
{
    *out = urf_fast_azimuth_of(urf_fast_polar(x, y));
    return urf_fast_az_ok(x, y);
}

/* star_shaped_search.cpp:73-107: is the point inside the rectangular beam of its sector */
__device__ __forceinline__ bool urf_in_beam(const urf_beam& b, float x, float y)
{
    if (b.yx) {
        const float c = b.d * y;
        return (c - b.o) < x && x < (c + b.o);
    }
    const float c = b.d * x;
    return (c - b.o) < y && y < (c + b.o);
}

/* The correctly rounded square root of a float in [2^-90, 2^126] (what the compiler's own expansion of sqrtf does between its
 * scaling of tiny arguments and its test for zero / infinity: the hardware's root is within one ulp, the two neighbours are
 * tried with one fma each).  k_front's planar range (star_shaped_search.cpp:164) for the points of the fast path: seven
 * instructions instead of eighteen; anything outside the interval takes __builtin_sqrtf. */
__device__ __forceinline__ float urf_sqrt_rn_normal(float x)
{
    float r = __builtin_amdgcn_sqrtf(x);
    const float rd = __uint_as_float(__float_as_uint(r) - 1u), ru = __uint_as_float(__float_as_uint(r) + 1u);
    const float ed = __builtin_fmaf(-rd, r, x), eu = __builtin_fmaf(-ru, r, x);
    r = ed <= 0.0f ? rd : r;
    r = eu > 0.0f ? ru : r;
    return r;
}

/* non-negative floats order like their bit patterns */
__device__ __forceinline__ unsigned urf_fbits(float f) { return __float_as_uint(f); }

#endif /* URF_DEVICE_HPP */
