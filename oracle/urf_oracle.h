/*
 * urf_oracle.h -- CPU oracle ("oracle B") for the urban_road_filter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (urban_road_filter_amd/) never links, imports or falls back to it.
 *
 * A plain-C restatement of the reference's per-scan classification, statement
 * by statement in behaviour (see urf_oracle.c for the file:line map).  Pinned
 * against "oracle A" = the reference's own unmodified sources compiled against
 * stand-in ROS/PCL/Boost headers (oracle/Makefile -> oracle/_ref/) by
 * tests/test_oracle_vs_reference.py and the golden vectors in tests/golden/.
 */
#ifndef URF_ORACLE_H
#define URF_ORACLE_H

#include <stdint.h>
#include "urf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional per-stage outputs; every pointer may be NULL.  Arrays "per point"
 * have n entries and are indexed by INPUT point index. */
typedef struct urf_oracle_debug {
    float*   valpha;      /* per point; -1 outside the ROI */
    int16_t* ring;        /* per point; -1 if not bucketed */
    float*   azimuth;     /* per point (bucketed points only, else 0) */
    float*   range2d;     /* per point (bucketed points only, else 0) */
    uint8_t* detect;      /* per point on a ring: bit0 star, bit1 x_zero, bit2 z_zero (as carried into array3D) */
    int16_t* sector;      /* per point; -1 outside the ROI, star disabled, or removed by the beam filter */
    float*   angle_table; /* [channels] sorted table, zero padded */
    float*   max_dist;    /* [channels] */
    float*   quadrants;   /* [4] q1..q4 */
    int16_t* beam_stop;   /* [2*361] first blocked ring per fwd/bwd beam; n_rings = free; -1 = not cast */
    /* the published clouds as input-index sequences in the reference's order (ring-major,
     * azimuth ascending, lidar_segmentation.cpp:354-367, 605-608); n entries of room each;
     * the lengths are info.n_road / n_curb / n_ring10 */
    uint32_t* road_order;
    uint32_t* curb_order;
    uint32_t* ring10_order;
    /* lidar_segmentation.cpp:295-351: marker points x,y,z,red (room for 361*4 floats) and their number */
    float*    marker_pts;
    uint32_t* n_marker_pts;
} urf_oracle_debug;

/* ---- road_marker line strips (lidar_segmentation.cpp:369-602) ---------------- */
typedef struct urf_oracle_marker {
    int32_t  id, action, type;     /* action 0 = ADD, 2 = DELETE; type 4 = LINE_STRIP */
    float    r, g, b, a;
    uint32_t first_point, n_points;   /* into urf_oracle_markers.pts (x,y,z doubles) */
} urf_oracle_marker;
typedef struct urf_oracle_markers {
    urf_oracle_marker* markers;    /* caller: room for cap_markers */
    uint32_t n_markers, cap_markers;
    double*  pts;                  /* caller: room for 3*cap_points doubles */
    uint32_t n_points, cap_points;
    int32_t  published;            /* 0: nothing published (cM <= 2) */
} urf_oracle_markers;
/* state the reference keeps between callbacks: ghostcount (lidar_segmentation.cpp:23) and the
 * member linestring `line` (data_structures.hpp:139), which keeps the points of a strip that was
 * started but not closed */
typedef struct urf_oracle_marker_state {
    int32_t ghostcount;
    float   line_x[1024], line_y[1024];
    int32_t line_n;
} urf_oracle_marker_state;
int urf_oracle_marker_strips(const float* marker_pts, uint32_t n_marker_pts, const urf_marker_params* mp,
                             urf_oracle_marker_state* state, urf_oracle_markers* out);

/* Classifies one scan given as SoA.  labels: n bytes (urf.h label byte).
 * Returns URF_OK, URF_TOO_FEW_POINTS, or a negative error. */
int urf_oracle_classify(const float* x, const float* y, const float* z, uint32_t n,
                        const urf_params* params, uint8_t* labels,
                        urf_scan_info* info, urf_oracle_debug* dbg);

/* star_shaped_search.cpp:109: libstdc++'s std::sort by r (urf_stdsort.h) on n (r, id) records, in place; for
 * tests/test_stdsort.py */
void urf_oracle_std_sort(float* r, int* id, int n);
long urf_oracle_std_sort_heap_sorts(void);   /* calls of the heap-sort fallback so far (depth limit reached) */

/* The three libm replacements, exported for tests/test_libm.py. */
float urf_oracle_acosf(float x);
float urf_oracle_asinf(float x);
float urf_oracle_atan2f(float y, float x);

#ifdef __cplusplus
}
#endif
#endif
