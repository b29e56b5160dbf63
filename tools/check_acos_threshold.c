/* r5: "alpha <= angleFilter" for alpha = (float)((double)(acosf(b) * 180.0f) / M_PI) (x_zero_method.cpp:58-61, z_zero_method.cpp:63-66)
 * is "b >= T" for a threshold T that depends on angleFilter only -- IF alpha falls (weakly) as b grows.  This checks exactly that for
 * include/urf_libm.h's urf_acosf, over EVERY float b in [-1, 1] (2 130 706 433 of them), and then, for a set of filter angles, that the
 * threshold found by bisection (the code of urf_api.hip: urf_angle_threshold) reproduces the predicate for every b.
 *   gcc -O2 -ffp-contract=off -fopenmp -I include tools/check_acos_threshold.c -o /tmp/check_acos -lm && /tmp/check_acos [quick]
 * Full run (r5, 2 m 40 s on 8 cores): alpha rises at 0 of 2 130 706 433 neighbouring pairs; 0 of 2 130 706 434 floats decided differently for
 * each of the fourteen angles. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "urf_libm.h"

static float alpha_of(float b) { return (float)((double)(urf_acosf(b) * 180.0f) / URF_PI_D); }
/* floats of [-1, 1] in ascending order: key 0 = -1.0f ... */
static float from_key(uint32_t k)
{
    const uint32_t neg = 0x3f800000u;   /* bits of 1.0f */
    uint32_t u = k <= neg ? 0x80000000u | (neg - k) : k - neg - 1u;   /* -1 .. -0 | +0 .. 1 */
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define NKEYS (2u * 0x3f800000u + 2u)

static float threshold(float A)   /* smallest b (as a float of [-1, 1]) with alpha(b) <= A; 2.0f: none */
{
    if (!(alpha_of(1.0f) <= A))
        return 2.0f;
    uint32_t lo = 0, hi = NKEYS - 1;   /* alpha(from_key(hi)) <= A */
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (alpha_of(from_key(mid)) <= A)
            hi = mid;
        else
            lo = mid + 1;
    }
    return from_key(lo);
}

int main(int argc, char** argv)
{
    /* "quick" (tests/test_libm.py): every 61st neighbouring pair and, per filter angle, the 400 000 floats around the threshold plus
     * every 61st of the rest -- seconds instead of minutes; the full run was done once (r5: 0 / 0, header of this file) */
    const uint32_t step = (argc > 1 && strcmp(argv[1], "quick") == 0) ? 61u : 1u;
    unsigned long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (uint32_t k = 1; k < NKEYS; k += step)
        if (alpha_of(from_key(k)) > alpha_of(from_key(k - 1)))
            bad++;
    printf("alpha(b) rises somewhere: %llu of %u neighbouring pairs\n", bad, (NKEYS - 1) / step);
    const float As[] = { 150.0f, 140.0f, 120.0f, 175.0f, 100.0f, 170.0f, 0.0f, 180.0f, 179.99998f, 90.0f, 60.000004f, 1e-3f, -1.0f, 200.0f };
    for (unsigned a = 0; a < sizeof(As) / sizeof(As[0]); a++) {
        const float A = As[a], T = threshold(A);
        unsigned long long wrong = 0;
#pragma omp parallel for reduction(+ : wrong) schedule(static)
        for (uint32_t k = 0; k < NKEYS; k += step) {
            const float b = from_key(k);
            wrong += (alpha_of(b) <= A) != (b >= T);
        }
        if (step > 1 && T <= 1.0f) {   /* densely around the threshold */
            uint32_t kt = 0, lo = 0, hi = NKEYS - 1;
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (from_key(mid) >= T)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            kt = lo;
            const uint32_t k0 = kt > 200000u ? kt - 200000u : 0u, k1 = kt + 200000u < NKEYS ? kt + 200000u : NKEYS;
            for (uint32_t k = k0; k < k1; k++) {
                const float b = from_key(k);
                wrong += (alpha_of(b) <= A) != (b >= T);
            }
        }
        printf("angleFilter %.9g: threshold %.9g (%a), %llu of %u floats decided differently\n", A, T, T, wrong, NKEYS);
        bad += wrong;
    }
    return bad != 0;
}
