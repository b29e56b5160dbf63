/*
 * urf_test_hooks.h -- test and benchmark hooks.  NOT part of the product ABI.
 *
 * liburf_hip.so (what a node links) exports include/urf.h and nothing declared here.  These entry points exist only
 * in liburf_hip_test.so: the same sources compiled with -DURF_ENABLE_TEST_HOOKS plus the synthetic-sweep generator
 * (urban_road_filter_amd/build.py).  tests/ and bench.py load that library for these calls; a context created in one
 * library must be used with that library's entry points only.
 */
#ifndef URF_TEST_HOOKS_H
#define URF_TEST_HOOKS_H

#include "urf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic sweeps -------------------------------
 * Host-side generator of the benchmark clouds: `rings` x `cols` rays from a
 * sensor 1.8 m above ground, scene 0 = flat plane, 1 = street with 0.15 m
 * curbs at |y| = 4 m, 2 = narrow street (curbs at |y| = 3 m, inside the reach
 * of the innermost rings, so that the blind-spot logic of blind_spots.cpp:17-99
 * engages); column-major "firing order" (idx = col*rings + ring);
 * per-sector radial ties removed.  Scenes 3 / 4 = scenes 1 / 2 as a sensor's driver
 * delivers them: range noise (sigma 1 cm), range quantised to 2 mm, 1.5 % drop-outs,
 * the planar-range ties LEFT IN (~10 000 per 64 x 2048 sweep).
 * Writes n = rings*cols floats to x, y, z. */
int urf_synth_cloud(uint32_t rings, uint32_t cols, int scene, uint64_t seed,
                    float* x, float* y, float* z);

/* Benchmark helper: the submit / collect loop of a C / C++ client of the asynchronous path (what a node's
 * subscriber callback and publisher do, lidar_segmentation.cpp:53,95,612-621), timed inside the library:
 * n_sweeps messages taken round robin from msgs[0..n_msgs) (host buffers of n_points records each), at most
 * in_flight (1..URF_MAX_IN_FLIGHT) submitted before the oldest is collected into labels_out (may be NULL).
 * producer_pinned != 0: the messages are produced in the library's pinned buffers (urf_pinned_input; each
 * buffer is filled once, outside the producer's cost).  *seconds = wall time of the whole loop. */
int urf_bench_callback_stream(urf_ctx* ctx, const uint8_t* const* msgs, uint32_t n_msgs, uint32_t n_points,
                              uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                              uint32_t n_sweeps, uint32_t in_flight, int producer_pinned, uint8_t* labels_out,
                              double* seconds);

/* Device self test of the arithmetic shortcuts the kernels take (currently: the
 * 3-operation division by pi against the IEEE division, exhaustively over all
 * floats in [0, 600]).  *n_mismatches must come back 0.  Synchronous. */
int urf_selftest(urf_ctx* ctx, uint64_t* n_mismatches);
/* Measured error of the float fast paths that settle ring and sector decisions (k_split) over
 * n_samples pseudo-random points: err[0] = max |approx - exact| of the vertical angle [deg] (k_split: the
 * angle whose cotangent its u = -z / rho is; k_ring_table's look-ahead: a float arc tangent),
 * err[1] of the polar angle [rad], err[2] of the scaled polar angle (at the configured number of
 * sectors), err[3] of the azimuth AS A FRACTION of its margin (which grows towards the x axis, where
 * the reference's own value is ill-conditioned; k_split / k_label).  The first three must stay below
 * the margins the kernels use (3e-4, 2e-6, 2.5e-4 * max(1, sectors / 360)), the last below 1.  err
 * has room for 4 floats.  Synchronous. */
int urf_selftest_fast(urf_ctx* ctx, uint64_t n_samples, float* err);
/* Test hook: bit 2 (value 4) forces the general (comparison network) path of the star-shaped sort
 * for every sector; 0 in production.  Takes effect with the next classify call. */
int urf_set_debug_flags(urf_ctx* ctx, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* URF_TEST_HOOKS_H */
