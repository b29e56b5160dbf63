/*
 * synth.cpp -- synthetic LiDAR sweeps for parity fixtures and the benchmark
 * (SURVEY.md section 8d).  Host code, no device work.
 *
 * The reference ships no data (README.md:36-46 points at a rosbag that is not
 * in the repository), so the benchmark clouds are analytic: a spinning sensor
 * 1.8 m above a flat ground plane or a street with two curbs, sampled on a
 * rings x cols grid and stored in firing order.
 *
 * Scenes 0-2 are the analytic benchmark clouds of SURVEY.md section 8d:
 *   - no two points of one star-shaped sector share the same float planar
 *     range r = sqrtf(x*x+y*y) (the survey's rule for the benchmark clouds; the
 *     reference orders equal r with std::sort, star_shaped_search.cpp:109);
 *   - no azimuth in (-5e-7, 0) rad (sector index 360, a null dereference at
 *     star_shaped_search.cpp:171-173) and no point with x == y == 0.
 * Scenes 3 and 4 (r5) are the SENSOR-LIKE versions of scenes 1 and 2: what a
 * spinning LiDAR's driver delivers -- Gaussian range noise (sigma 1 cm), the
 * range quantised to 2 mm along the ray, 1.5 % of the returns dropped -- and
 * nothing removed afterwards: a sweep holds ~10 000 exact planar-range ties
 * inside its star sectors (neighbouring firings of a ring on flat ground),
 * whose order under the reference's std::sort decides labels.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "urf_test_hooks.h"
#include "urf_libm.h"

namespace {

inline uint64_t splitmix64(uint64_t seed, uint64_t k)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (k + 1) * 0xD1B54A32D192ED03ULL;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

/* uniform in [-1, 1) */
inline double unit(uint64_t seed, uint64_t k)
{
    return (double)(splitmix64(seed, k) >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

/* sin and cos from IEEE basic operations only (fma, +, *), so that the clouds
 * are bit-identical on every host regardless of its libm.  |a| < ~1e3;
 * accuracy ~1e-16 absolute, far beyond what the generator needs. */
inline void det_sincos(double a, double* s, double* c)
{
    const double two_over_pi = 0x1.45f306dc9c883p-1;
    const double pio2_hi = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
    const double kf = std::floor(a * two_over_pi + 0.5);
    double r = __builtin_fma(-kf, pio2_hi, a);
    r = __builtin_fma(-kf, pio2_lo, r);
    const double r2 = r * r;
    /* Taylor series on |r| <= pi/4 */
    double ps = 1.0 / 355687428096000.0;              /* 1/17! */
    ps = __builtin_fma(ps, r2, -1.0 / 1307674368000.0);   /* 1/15! */
    ps = __builtin_fma(ps, r2, 1.0 / 6227020800.0);
    ps = __builtin_fma(ps, r2, -1.0 / 39916800.0);
    ps = __builtin_fma(ps, r2, 1.0 / 362880.0);
    ps = __builtin_fma(ps, r2, -1.0 / 5040.0);
    ps = __builtin_fma(ps, r2, 1.0 / 120.0);
    ps = __builtin_fma(ps, r2, -1.0 / 6.0);
    const double sn = __builtin_fma(r * r2, ps, r);
    double pc = -1.0 / 6402373705728000.0;            /* 1/18! */
    pc = __builtin_fma(pc, r2, 1.0 / 20922789888000.0);   /* 1/16! */
    pc = __builtin_fma(pc, r2, -1.0 / 87178291200.0);
    pc = __builtin_fma(pc, r2, 1.0 / 479001600.0);
    pc = __builtin_fma(pc, r2, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, r2, 1.0 / 40320.0);
    pc = __builtin_fma(pc, r2, -1.0 / 720.0);
    pc = __builtin_fma(pc, r2, 1.0 / 24.0);
    pc = __builtin_fma(pc, r2, -0.5);
    const double cs = __builtin_fma(r2, pc, 1.0);
    switch ((long long)kf & 3) {
    case 0: *s = sn;  *c = cs;  break;
    case 1: *s = cs;  *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}

/* the sector a point falls into, exactly as the classification computes it
 * (star_shaped_search.cpp:164-171, 360 sectors) */
inline int sector_of(float x, float y)
{
    const float Kfi = (float)(360.0 / (2 * URF_PI_D));
    float fi = urf_atan2f(y, x);
    if (fi < 0)
        fi = (float)((double)fi + 2 * URF_PI_D);
    int f = (int)(fi * Kfi);
    return f >= 360 ? 0 : f;
}

}   // namespace

extern "C" int urf_synth_cloud(uint32_t rings, uint32_t cols, int scene, uint64_t seed,
                               float* x, float* y, float* z)
{
    if (!x || !y || !z || rings == 0 || cols == 0 || scene < 0 || scene > 4)
        return URF_ERR_INVALID_ARG;
    const bool sensor_like = scene >= 3;
    if (sensor_like)
        scene -= 2;
    const double h = 1.8;             /* sensor height above the road */
    const double curb_y = scene == 2 ? 3.0 : 4.0;   /* |y| of the curb faces */
    const double curb_h = 0.15;       /* curb height */
    const double max_range = 120.0;   /* beyond: no return -> (0,0,0) */
    const double deg = URF_PI_D / 180.0;
    const double e_lo = (rings == 16) ? -15.0 : -24.8;
    const double e_hi = (rings == 16) ? -1.0 : -2.0;
    const size_t n = (size_t)rings * cols;

    for (uint32_t c = 0; c < cols; c++) {
        const double th = ((double)c + 0.5) * (2 * URF_PI_D) / (double)cols;
        double ct, st;
        det_sincos(th, &st, &ct);
        for (uint32_t r = 0; r < rings; r++) {
            const double e = (rings > 1 ? e_lo + (e_hi - e_lo) * (double)r / (double)(rings - 1) : e_lo) * deg;
            double ce, se;
            det_sincos(e, &se, &ce);
            const double dx = ce * ct, dy = ce * st, dz = se;
            double t = -h / dz;   /* ground hit */
            if (scene != 0) {
                const double yg = t * dy;
                if (std::fabs(yg) >= curb_y) {
                    const double tc = curb_y / std::fabs(dy);   /* reaches the curb plane first */
                    const double zc = tc * dz;
                    if (zc < -h + curb_h)
                        t = tc;                        /* vertical curb face */
                    else
                        t = -(h - curb_h) / dz;        /* sidewalk */
                }
            }
            const size_t idx = (size_t)c * rings + r;
            bool dropped = false;
            if (sensor_like) {
                /* Gaussian by Irwin-Hall (12 uniforms: basic operations only, bit-identical on every host), sigma 1 cm;
                 * quantised to 2 mm along the ray; 1.5 % drop-outs */
                double g = 0.0;
                for (uint64_t u = 0; u < 12; u++)
                    g += unit(seed ^ 0x5eed5eedULL, idx * 16 + u);
                t += 0.01 * (g * 0.5);   /* the sum of 12 uniforms on [-1, 1) has variance 4 */
                t = std::floor(t / 0.002 + 0.5) * 0.002;
                dropped = unit(seed ^ 0xd509ULL, idx) > 0.97;   /* P = 0.015 */
            } else {
                t *= 1.0 + 1e-4 * unit(seed, idx);
            }
            if (!(t < max_range) || dropped) {
                x[idx] = y[idx] = z[idx] = 0.0f;
            } else {
                x[idx] = (float)(t * dx);
                y[idx] = (float)(t * dy);
                z[idx] = (float)(t * dz);
            }
        }
    }

    if (sensor_like)
        return URF_OK;   /* ties stay in */
    /* remove radial ties inside every sector: scale the later point of a tie
     * by (1 + k*2^-21) and re-check, until every sector is tie-free */
    struct Ent { float r; uint32_t idx; };
    std::vector<std::vector<Ent>> sec(360);
    for (int round = 0; round < 64; round++) {
        for (auto& s : sec)
            s.clear();
        for (size_t i = 0; i < n; i++) {
            if (x[i] == 0.0f && y[i] == 0.0f && z[i] == 0.0f)
                continue;
            sec[sector_of(x[i], y[i])].push_back({ std::sqrt(x[i] * x[i] + y[i] * y[i]), (uint32_t)i });
        }
        size_t ties = 0;
        for (auto& s : sec) {
            std::sort(s.begin(), s.end(), [](const Ent& a, const Ent& b) {
                return a.r < b.r || (a.r == b.r && a.idx < b.idx);
            });
            for (size_t i = 1, k = 1; i < s.size(); i++) {
                if (s[i].r == s[i - 1].r) {
                    const float f = 1.0f + (float)k * 4.76837158203125e-07f;   /* 2^-21 */
                    const uint32_t j = s[i].idx;
                    x[j] *= f; y[j] *= f; z[j] *= f;
                    ties++;
                    k++;
                } else {
                    k = 1;
                }
            }
        }
        if (ties == 0)
            return URF_OK;
    }
    return URF_ERR_INVALID_ARG;   /* could not make the cloud tie-free */
}
