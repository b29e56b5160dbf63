/*
 * urf_kernels.hpp -- the gfx950 kernels of the per-scan classification.
 *
 * One launch covers a whole batch: blockIdx.y (or .x for per-scan kernels) is
 * the scan, the other grid dimension the tile / ring / sector inside it.
 * Pipeline (reference lines each kernel replaces; DESIGN.md has the full map):
 *
 *   k_ring_table   first-fit ring-angle table + sort, straight from x/y/z; per-entry thresholds on
 *                  cot(vertical angle) and the lookup table over it            lidar_segmentation.cpp:145-196,205
 *   k_split        ONE pass over x/y/z per 2048-point tile: ROI test, ring and star sector of every point,
 *                  stable split of the tile by ring and by sector into its own region of the scratch
 *                  arrays      lidar_segmentation.cpp:106-166,226-242,276, star_shaped_search.cpp:162-174
 *   k_index        piece < 30 test, per-ring run tables over the tiles, sector sizes and first runs
 *                                                                              lidar_segmentation.cpp:120-126
 *   k_star_sort_*  per-sector sort by range, slopes             star_shaped_search.cpp:109-129
 *   k_star_walk    per-sector running-mean slope test           star_shaped_search.cpp:123-149
 *   k_ring         x_zero, z_zero, azimuth, maxDistance, per-degree curb tables,
 *                  blind-spot quadrants   x_zero_method.cpp, z_zero_method.cpp,
 *                  lidar_segmentation.cpp:245-274, blind_spots.cpp:17-57
 *   k_beams        first blocked ring of each of the 2 x 331 beams     blind_spots.cpp:65-283
 *   k_label        road acceptance per point, label bytes       blind_spots.cpp:124-130,164-170, lidar_segmentation.cpp:354-367
 *
 * The kernels live in one header per family (r6), included below in this order: urf_k_table.hpp, urf_k_split.hpp,
 * urf_k_star.hpp, urf_k_ring.hpp, urf_k_beams_label.hpp, urf_k_outputs.hpp.  A batch of sweeps in firing order takes the
 * fused front end of urf_front.hpp (k_front, k_front_finish, k_label_front) instead of k_split / k_ring / k_label.
 *
 * The reference's per-ring azimuth quicksort (lidar_segmentation.cpp:70-93,
 * 289-291; 56 % of its run time) has no counterpart: blindSpots only ever asks
 * "is there a curb point with azimuth in [lo, hi] on ring k" and "is this point
 * inside an accepted window", both of which are answered from per-degree
 * min/max tables without ordering the ring.
 */
#ifndef URF_KERNELS_HPP
#define URF_KERNELS_HPP

#include "urf_device.hpp"

#define URF_INT_NONE_MIN 0x7fffffff
/* (float)sqrt(s) < 5.0 (x_zero_method.cpp:35-40, z_zero_method.cpp:23-28) holds exactly for the
 * doubles s below this one: sqrt and the rounding to float are monotone, the threshold is the
 * smallest double whose rounded root reaches 5.0f (found by bisection, tools/check_dist5.c). */
#define URF_DIST5_SQ 0x1.8ffffd800000fp+4
#ifndef URF_RING_THREADS
#define URF_RING_THREADS 128
#endif
#define URF_LABEL_THREADS 384
#define URF_STAR_THREADS 64
#ifndef URF_STAR_LOG_NB
#define URF_STAR_LOG_NB 9u          /* k_star_sort_small: 512 range buckets */
#endif
#define URF_STAR_NB (1u << URF_STAR_LOG_NB)
#define URF_INGEST_THREADS 256       /* tile kernels that need no big LDS tile run 8 workgroups per CU */
#define URF_LABEL_TILE_THREADS 256   /* k_label: one tile per workgroup, 16 slots per thread, 8 workgroups per CU */
#define URF_STAR_MID_CAP_ 2048
#define URF_STAR_SMALL_CAP 384       /* k_star_sort_small: sectors of up to 6 x 64 points, one wave each */

#include "urf_k_table.hpp"
#include "urf_k_split.hpp"
#include "urf_k_star.hpp"
#include "urf_k_ring.hpp"
#include "urf_k_beams_label.hpp"
#include "urf_k_outputs.hpp"

#endif /* URF_KERNELS_HPP */
