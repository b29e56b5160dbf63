/*
 * urf.h -- C ABI of the MI355X-native urban_road_filter hot path.
 *
 * Drop-in boundary.  The reference processes every LiDAR
 * sweep inside one C++ callback,
 *     void Detector::filtered(const pcl::PointCloud<pcl::PointXYZI>& cloud)
 *         include/urban_road_filter/data_structures.hpp:118
 *         src/lidar_segmentation.cpp:95-622
 * fed by a sensor_msgs/PointCloud2 subscription (src/lidar_segmentation.cpp:53)
 * and answering with the clouds "road", "curb", "roi", "road_probably"
 * (src/lidar_segmentation.cpp:55-58, 354-367, 605-608, 618-621).
 *
 * This library replaces the geometric classification in the middle
 * (lidar_segmentation.cpp:100-293,353-367,605-608 + star_shaped_search.cpp +
 * x_zero_method.cpp + z_zero_method.cpp + blind_spots.cpp) by hand-written
 * gfx950 HIP kernels.  Points go in as PointCloud2-layout bytes (or as SoA
 * x/y/z device arrays for resident batches); what comes out is ONE BYTE PER
 * INPUT POINT that encodes the reference's per-point result:
 *
 *     bits 0-1  isCurbPoint of the reference (data_structures.hpp:44):
 *               0 = none, 1 = road (blind_spots.cpp:128,168,237,277),
 *               2 = curb (star_shaped_search.cpp:146, x_zero_method.cpp:66,
 *               z_zero_method.cpp:71)
 *     bit 2     point passed the ROI filter, i.e. is in the "roi" cloud
 *               (lidar_segmentation.cpp:106-117, 620)
 *     bit 3     point was assigned to a ring (lidar_segmentation.cpp:226-277);
 *               only such points can appear in "road"/"curb"
 *     bit 4     point lies on sorted ring index 10, i.e. is in the
 *               "road_probably" cloud (lidar_segmentation.cpp:605-608)
 *
 * so   road = {i : (label[i] & 3) == 1},  curb = {i : (label[i] & 3) == 2},
 *      roi  = {i : label[i] & 4},  road_probably = {i : label[i] & 16}.
 * The C++ adapter (urban_road_filter_amd/csrc/detector.hpp) re-materialises
 * the four clouds from these sets.  Set membership is the contract; the order
 * of points inside the reference's published clouds is not.
 *
 * All functions return 0 on success, a negative urf_status on error.  One
 * context owns one device, one HIP stream and all scratch memory; a context is
 * not thread-safe, any number of contexts may coexist (the reference keeps its
 * state in globals and allows one Detector per process).
 */
#ifndef URF_H
#define URF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): URF_MAX_IN_FLIGHT sweeps on the asynchronous path (tickets map to slots modulo it),
 * urf_result_labels() validates its ticket, URF_NUM_KERNELS / kernel names as listed below,
 * urf_enable_stage_capture() takes a mode 0..2, urf_scan_info::n_nan_azimuth is written.
 * 3 (round 4): the test / benchmark hooks (urf_synth_cloud, urf_bench_callback_stream, urf_selftest*,
 * urf_set_debug_flags) left this header and the product library (include/urf_test_hooks.h, liburf_hip_test.so);
 * rings that hold a point with x == y == 0 follow the reference (deviation D5 of earlier versions is gone);
 * urf_read_stage / urf_ordered_indices / urf_marker_points answer URF_ERR_BUSY for a sweep whose scratch row has
 * been resubmitted; urf_callback_path_state reports sequence bits 2 and 3.
 * 4 (round 5): equal planar ranges inside a star sector are ordered as libstdc++'s std::sort orders them
 * (star_shaped_search.cpp:109) and equal azimuths inside a ring as the reference's Lomuto quicksort does
 * (lidar_segmentation.cpp:70-93): "deviation D2" of earlier versions is gone, labels on real sensor data (range ties in
 * every sector) and the published order equal the reference's; urf_callback_path_state reports sequence bit 4.
 * 5 (round 6): urf_set_front_mode / urf_front_scans (the fused front end for batches of organised sweeps: firing order and
 * row-major); the entry points that read ring-sorted intermediate results may run the last batch call again, see there;
 * urf_callback_path_preset.
 * WHICH std::sort (4 above): the one of libstdc++ as shipped with GCC 5 .. 13 (bits/stl_algo.h: __sort = __introsort_loop with
 * _S_threshold 16, __move_median_to_first on (first + 1, mid, last - 1), __unguarded_partition, depth limit 2 * floor(log2 n),
 * __partial_sort as the fallback, then __final_insertion_sort); tests/test_stdsort.py pins the restatement against the std::sort
 * of the build host and FAILS when they disagree.  A reference built against another standard library (libc++) or a future
 * libstdc++ with another __sort orders equal planar ranges differently, and its labels on tied data then differ from this
 * library's at the points concerned -- nothing at run time can tell. */
#define URF_ABI_VERSION 5

/* ---- label byte --------------------------------------------------------- */
#define URF_LABEL_MASK   0x03u
#define URF_LABEL_NONE   0u
#define URF_LABEL_ROAD   1u
#define URF_LABEL_CURB   2u
#define URF_FLAG_ROI     0x04u
#define URF_FLAG_RING    0x08u
#define URF_FLAG_RING10  0x10u

/* ---- status codes ------------------------------------------------------- */
typedef enum urf_status {
    URF_OK = 0,
    /* per-scan status (positive: not an error) */
    URF_TOO_FEW_POINTS = 1,      /* < 30 ROI points: the reference returns without
                                    publishing (lidar_segmentation.cpp:124-126);
                                    every label of the scan is 0 */
    /* errors */
    URF_ERR_INVALID_ARG = -1,
    URF_ERR_NO_DEVICE = -2,      /* no HIP device / runtime error at create */
    URF_ERR_HIP = -3,            /* HIP runtime failure, see urf_last_error() */
    URF_ERR_CAPACITY = -4,       /* scan or batch larger than urf_create() sizes */
    URF_ERR_OOM = -5,
    URF_ERR_PARAMS = -6,         /* parameter outside the supported range */
    URF_ERR_BUSY = -7            /* every slot of the asynchronous single-scan path is in flight (or the asked-for
                                    result is still being written) */
} urf_status;

/* ---- parameters ----------------------------------------------------------
 * The hot-path subset of the reference's 27 dynamic_reconfigure fields
 * (cfg/LidarFilters.cfg:10-84 -> src/main.cpp:4-34 -> namespace params,
 * data_structures.hpp:66-88) plus its three compile-time globals.  Field names
 * follow namespace params.  Types follow the reference (float/int/bool), so
 * e.g. interval is 0.18f, not 0.18.
 */
typedef struct urf_params {
    uint32_t size;               /* = sizeof(urf_params); ABI versioning */
    int32_t  x_zero_method;      /* cfg:16  bool */
    int32_t  z_zero_method;      /* cfg:17  bool */
    int32_t  star_shaped_method; /* cfg:18  bool */
    int32_t  blind_spots;        /* cfg:19  bool */
    int32_t  xDirection;         /* cfg:27  0 both, 1 +X, 2 -X */
    float    interval;           /* cfg:30  ring-angle tolerance [deg] */
    float    curbHeight;         /* cfg:33  curb_height [m] */
    int32_t  curbPoints;         /* cfg:36  curb_points, 1..30 */
    float    beamZone;           /* cfg:39  [deg] */
    float    min_X, max_X;       /* cfg:42-43 */
    float    min_Y, max_Y;       /* cfg:46-47 */
    float    min_Z, max_Z;       /* cfg:50-51 */
    float    angleFilter1;       /* cfg:54  cylinder_deg_x */
    float    angleFilter2;       /* cfg:57  cylinder_deg_z */
    float    angleFilter3;       /* cfg:60  curb_slope_deg */
    float    kdev_param;         /* cfg:63 */
    float    kdist_param;        /* cfg:66 */
    int32_t  starbeam_filter;    /* cfg:69  bool */
    int32_t  dmin_param;         /* cfg:72 */
    int32_t  channels;           /* lidar_segmentation.cpp:4   (64), 1..128 */
    int32_t  sectors;            /* star_shaped_search.cpp:8   rep = 360 */
    float    beam_width;         /* star_shaped_search.cpp:9   width = 0.2 */
} urf_params;

/* Fills *p with the reference defaults (cfg/LidarFilters.cfg) and the
 * reference's compile-time globals (channels 64, rep 360, width 0.2). */
int urf_default_params(urf_params* p);

/* ---- the live parameter surface ---------------------------------------------
 * One descriptor per gen.add() of cfg/LidarFilters.cfg:10-84 (the node's dynamic_reconfigure
 * interface): the reference's parameter name, type, default and range, the xDirection enum
 * (:22-27), and where the value lives in urf_params / urf_marker_params (`field`, `offset`).
 * fixed_frame / topic_name configure the ROS node, not the classification (URF_PARAM_NODE_ONLY).
 * urf_clamp_params() does what the dynamic_reconfigure server does to a request before
 * paramsCallback (src/main.cpp:4-34) sees it: every value is clamped to [min, max], bools become
 * 0/1; *n_clamped (optional) = number of values it changed.  mp may be NULL.  (urf_set_params still
 * rejects what the kernels cannot run: channels, sectors, curbPoints outside their limits.) */
typedef enum urf_param_type { URF_PARAM_BOOL = 0, URF_PARAM_INT = 1, URF_PARAM_DOUBLE = 2, URF_PARAM_STR = 3 } urf_param_type;
typedef enum urf_param_where { URF_PARAM_IN_PARAMS = 0, URF_PARAM_IN_MARKER_PARAMS = 1, URF_PARAM_NODE_ONLY = 2 } urf_param_where;
typedef struct urf_param_desc {
    const char* cfg_name;    /* name in cfg/LidarFilters.cfg */
    const char* field;       /* member of urf_params / urf_marker_params ("" = node only) */
    int32_t     where;       /* urf_param_where */
    uint32_t    offset;      /* byte offset of the member (float for double_t, int32_t for int_t / bool_t) */
    int32_t     type;        /* urf_param_type (the cfg's type) */
    double      def, min, max;
    const char* def_str;     /* default of a str_t, else NULL */
    const char* enum_values; /* "name=value,..." where the cfg defines an enum, else NULL */
    int32_t     cfg_line;    /* line of the gen.add() */
} urf_param_desc;
struct urf_marker_params;
int urf_param_count(void);
const urf_param_desc* urf_param_table(void);
int urf_clamp_params(urf_params* p, struct urf_marker_params* mp, uint32_t* n_clamped);

/* ---- per-scan summary ---------------------------------------------------- */
typedef struct urf_scan_info {
    int32_t  status;     /* URF_OK or URF_TOO_FEW_POINTS */
    uint32_t n_roi;      /* "piece",  lidar_segmentation.cpp:120 */
    uint32_t n_rings;    /* "index",  lidar_segmentation.cpp:139,194 */
    uint32_t n_ring_pts; /* points assigned to a ring */
    uint32_t n_road;     /* size of the "road" cloud */
    uint32_t n_curb;     /* size of the "curb" cloud */
    uint32_t n_ring10;   /* size of "road_probably" */
    uint32_t n_nan_azimuth; /* ring points with x == y == 0 (azimuth NaN): 0 on every real sweep; followed as the
                               reference treats them, see below */
} urf_scan_info;
/* A ring point with x == y == 0 has d = 0 and azimuth asin(0 / 0) = NaN (lidar_segmentation.cpp:245-269).  In the
 * reference that NaN goes through the per-ring Lomuto quicksort (:70-93), where every comparison with it is false: the
 * point ends up at an input-order-dependent place of the sorted ring, and the beam scans of blind_spots.cpp
 * (:107,146,216,255) end there -- forward beams see only what stands in front of the ring's first NaN, backward beams only
 * what stands behind its last one.  Deterministic, hence followed: for such a ring the library runs the reference's
 * quicksort literally (k_nan_rings) and limits the beams accordingly; labels, counters and the published order
 * (urf_ordered_indices) equal the reference's.  (Up to round 3 this was "deviation D5": such a point simply never became
 * road and never cut a beam short.)  n_nan_azimuth counts these points (0 on every real sweep: a return at the sensor's
 * own axis).  One residue: two points of such a ring with bit-identical azimuths on either side of a NaN's place are told
 * apart by position in the reference and by value here.
 * COST: the quicksort is the reference's own -- Lomuto, pivot = last element, O(n^2) on the nearly sorted rings of an
 * organised sweep (56 % of the reference's CPU time goes there) -- run by ONE wave per such ring: about 1 ms for a ring of
 * 2 048 points, several ms for a ring beyond 6 144 points (sorted in global memory), against ~0.13 ms for the whole sweep
 * otherwise; on the callback path the first such sweep is additionally run twice (urf_callback_path_state).  A region of
 * interest that excludes the sensor's own axis (the reference's "x + y + z != 0" filter already drops (0, 0, 0) filler
 * points) never gets here. */

typedef struct urf_ctx urf_ctx;

/* ---- lifetime ------------------------------------------------------------
 * Replaces Detector::Detector (lidar_segmentation.cpp:51-65): allocates every
 * scratch buffer once (the reference allocates channels x piece x 64 B per
 * scan, lidar_segmentation.cpp:207) and runs beam_init()
 * (star_shaped_search.cpp:32-66).  max_points bounds the points of one scan,
 * max_batch the scans of one batch call. */
int urf_create(urf_ctx** ctx, int device_id, uint32_t max_points, uint32_t max_batch);
int urf_destroy(urf_ctx* ctx);

/* Replaces paramsCallback (src/main.cpp:4-34): may be called between scans. */
int urf_set_params(urf_ctx* ctx, const urf_params* p);
int urf_get_params(const urf_ctx* ctx, urf_params* p);

/* Run on a caller-owned hipStream_t (e.g. the framework's current stream);
 * NULL restores the context's own stream. */
int urf_set_stream(urf_ctx* ctx, void* hip_stream);
int urf_synchronize(urf_ctx* ctx);

/* ---- single scan, host buffers, PointCloud2 layout ------------------------
 * Replaces the body of Detector::filtered for one sensor_msgs/PointCloud2:
 * `data` holds n_points records of point_step bytes; x/y/z are little-endian
 * FLOAT32 at byte offsets off_x/off_y/off_z (pcl::fromROSMsg resolves the
 * fields by name; every other field is ignored, as pcl::PointXYZI ignores
 * them).  labels_out (host, n_points bytes) receives the label bytes, *info
 * (optional) the summary.  Synchronous.  Returns URF_OK or an error; the
 * "too few points" condition is reported in info->status. */
int urf_classify_pc2(urf_ctx* ctx, const uint8_t* data, uint32_t n_points,
                     uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                     uint8_t* labels_out, urf_scan_info* info);

/* The same, asynchronously: the message's x / y / z are gathered into pinned memory (three planes: 12 bytes
 * per point cross PCIe whatever the point_step) and sent to the device, classified by ONE graph launch (the
 * kernel sequence of a sweep of this shape is captured once and replayed), and the labels come back to pinned
 * memory.  URF_MAX_IN_FLIGHT sweeps may be in flight; with a context created for max_batch >= 2 they are spread
 * over min(max_batch, URF_MAX_IN_FLIGHT) scratch rows, each with its own stream (copy in, kernels, copy out), so
 * that the copies and kernels of different sweeps overlap -- create the context with max_batch >= 4 for the
 * callback path:
 *     urf_classify_pc2_async(ctx, msg_a, ..., &ta);
 *     urf_classify_pc2_async(ctx, msg_b, ..., &tb);      // ... the fifth in a row returns URF_ERR_BUSY
 *     urf_classify_pc2_wait(ctx, ta, labels_a, &info_a);  // blocks until sweep a is done
 * Sweeps must be waited for in the order they were submitted if the results are to be looked at with
 * urf_read_stage / urf_ordered_indices / urf_marker_points (those see the sweep waited for last).
 * SHARED ROWS: with max_batch < URF_MAX_IN_FLIGHT the slots share min(max_batch, URF_MAX_IN_FLIGHT) scratch rows
 * (slot i uses row i modulo that number; slots on one row are serialised).  Labels and summary of every sweep are
 * right whatever the sharing -- each slot has result buffers of its own --, but the three entry points just named
 * read the sweep's ROW: once a later sweep has been submitted on that row they return URF_ERR_BUSY instead of mixing
 * two sweeps' intermediate results.  Create the context with max_batch >= the number of sweeps kept in flight, or
 * read a sweep's intermediate results before submitting on its row again.
 * Every other entry point of the context that touches its scratch memory (the batch calls, the three
 * just named, urf_compact_indices*) is ordered behind the sweeps still in flight; urf_synchronize()
 * waits for them as well.
 * labels_out may be NULL: urf_result_labels() then gives read access to the pinned result buffer of a
 * ticket that has been waited for, valid until the ticket's slot is used again (URF_MAX_IN_FLIGHT
 * submissions later); a ticket never issued or overtaken is refused with URF_ERR_INVALID_ARG, one
 * still in flight with URF_ERR_BUSY.  A producer that can fill a
 * buffer of the library's choosing (a driver, a deserialiser) saves the staging copy: it asks for
 * the pinned input buffer of the NEXT submission with urf_pinned_input() (URF_ERR_BUSY while that
 * slot is still in flight), writes the message there and passes that very pointer as `data` (a message
 * that starts inside that buffer but is not exactly it, or is longer than the size asked for, is
 * refused with URF_ERR_INVALID_ARG).  Reference: the subscriber callback, lidar_segmentation.cpp:53,95-100, and
 * the publishers, :612-621. */
#define URF_MAX_IN_FLIGHT 4
int urf_classify_pc2_async(urf_ctx* ctx, const uint8_t* data, uint32_t n_points,
                           uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                           uint32_t* ticket);
int urf_classify_pc2_wait(urf_ctx* ctx, uint32_t ticket, uint8_t* labels_out, urf_scan_info* info);
int urf_result_labels(urf_ctx* ctx, uint32_t ticket, const uint8_t** labels);
int urf_pinned_input(urf_ctx* ctx, size_t bytes, uint8_t** ptr);

/* ---- batch of scans, device-resident ---------------------------------------
 * n_scans independent scans of n_per_scan points each; scan s owns elements
 * [s*n_per_scan, (s+1)*n_per_scan) of d_x/d_y/d_z (SoA, device memory) and of
 * d_labels.  d_info (device, n_scans entries) is optional.  Asynchronous on
 * the context's stream. */
int urf_classify_batch_soa(urf_ctx* ctx, const float* d_x, const float* d_y, const float* d_z,
                           uint32_t n_per_scan, uint32_t n_scans,
                           uint8_t* d_labels, urf_scan_info* d_info);

/* Same with ragged scans: scan s owns [d_offsets[s], d_offsets[s+1]) (device
 * array of n_scans+1 uint32).  max_len >= the longest scan (host value). */
int urf_classify_batch_soa_ragged(urf_ctx* ctx, const float* d_x, const float* d_y, const float* d_z,
                                  const uint32_t* d_offsets, uint32_t max_len, uint32_t n_scans,
                                  uint8_t* d_labels, urf_scan_info* d_info);

/* Batch of PointCloud2-layout scans in device memory (n_per_scan records of
 * point_step bytes per scan, scans back to back). */
int urf_classify_batch_pc2(urf_ctx* ctx, const uint8_t* d_data, uint32_t n_per_scan, uint32_t n_scans,
                           uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                           uint8_t* d_labels, urf_scan_info* d_info);

/* ---- index-set outputs -----------------------------------------------------
 * Compacts the label bytes of ONE scan (device) into ascending index lists
 * (device, each with room for n_points entries; any may be NULL) and writes
 * the four counts to d_counts[0..3] = {road, curb, roi, road_probably}. */
int urf_compact_indices(urf_ctx* ctx, const uint8_t* d_labels, uint32_t n_points,
                        uint32_t* d_road, uint32_t* d_curb, uint32_t* d_roi, uint32_t* d_ring10,
                        uint32_t* d_counts);
/* The same for every scan of a batch in one launch (grid over tiles x scans, asynchronous on the
 * context's stream): scan s owns [s*n_per_scan, (s+1)*n_per_scan) of d_labels and of every list
 * (indices are relative to the scan), and d_counts[4*s .. 4*s+3].  The reference builds these sets
 * for every sweep (lidar_segmentation.cpp:354-367, 605-608). */
int urf_compact_indices_batch(urf_ctx* ctx, const uint8_t* d_labels, uint32_t n_per_scan, uint32_t n_scans,
                              uint32_t* d_road, uint32_t* d_curb, uint32_t* d_roi, uint32_t* d_ring10,
                              uint32_t* d_counts);

/* ---- the published clouds, in the reference's order ------------------------
 * The reference fills "road", "curb" and "road_probably" ring by ring (sorted
 * ring index ascending), each ring in ascending azimuth (its per-ring quicksort,
 * lidar_segmentation.cpp:70-93, 289-291, 354-367, 605-608); "roi" is in input
 * order.  After a classify call this returns the input indices of scan `scan`
 * in exactly that order (points of one ring with bit-identical azimuths in the
 * order the reference's quicksort leaves them in: it is run literally for such a
 * ring).  Host buffers with room
 * for n_points entries each (any may be NULL); counts[3] = {road, curb,
 * road_probably}.  Synchronous; costs one extra per-ring sort.
 * COST of bit-identical azimuths: a ring that holds ONE such pair is sorted a second time by the reference's own quicksort, run
 * literally by a single wave -- the ring is nearly sorted by then, Lomuto's scheme is quadratic on that: about 1 ms per ring of
 * 2048 points (the same holds for urf_marker_points and for a ring with a NaN azimuth).  A sensor that delivers exact duplicates
 * (dual returns written twice) in many rings makes these two entry points tens of milliseconds slower; the labels are not affected.
 * LIFETIME: urf_ordered_indices*, urf_marker_points* and urf_read_stage run kernels over the LAST
 * classify call's results, which include the caller's own buffers of that call: d_labels (all three)
 * and, for calls with ragged offsets, nothing else -- the x / y / z these kernels need were copied
 * into the context's scratch by the call itself.  d_labels of the last classify call must therefore
 * stay allocated and unmodified until the last of these calls on it has completed (for a sweep of the
 * callback path the library owns that buffer: nothing to keep alive). */
int urf_ordered_indices(urf_ctx* ctx, uint32_t scan, uint32_t* road, uint32_t* curb, uint32_t* ring10,
                        uint32_t* counts);
/* Every scan of the last classify call at once, results on the DEVICE (asynchronous on the
 * context's stream): list l of scan s at d_l + s*stride (stride >= the call's scan length; any list
 * may be NULL), d_counts[3*s .. 3*s+2] = {road, curb, road_probably}. */
int urf_ordered_indices_batch(urf_ctx* ctx, uint32_t* d_road, uint32_t* d_curb, uint32_t* d_ring10,
                              uint32_t stride, uint32_t* d_counts);

/* ---- road_marker: the marker points ------------------------------------------
 * lidar_segmentation.cpp:295-351: for every integer degree i = 0..360 the farthest road point
 * with azimuth in [i, i+1), scanning ring by ring (each ring in ascending azimuth) until the first
 * point of that degree that is not road; `red` tells whether such a point was met.  After a
 * classify call this returns the marker points of scan `scan`: pts[4*k + 0..3] = x, y, z, red
 * (host buffer with room for 361 points), *count = their number (the reference's cM).
 * The line strips the reference builds from them (colour grouping, Douglas-Peucker
 * simplification, ghost deletion, :369-602) are host code: urf::Detector::road_marker().
 * Synchronous; costs one extra per-ring sort. */
int urf_marker_points(urf_ctx* ctx, uint32_t scan, float* pts, uint32_t* count);
/* Every scan of the last classify call at once, results on the DEVICE (asynchronous): the marker
 * points of scan s at d_pts + s*361*4 (x, y, z, red per point), their number in d_counts[s]. */
int urf_marker_points_batch(urf_ctx* ctx, float* d_pts, uint32_t* d_counts);

/* The polygon parameters of the reference (cfg/LidarFilters.cfg:75-84 -> main.cpp:29-32). */
typedef struct urf_marker_params {
    uint32_t size;              /* = sizeof(urf_marker_params) */
    int32_t  simple_poly_allow; /* cfg:75  bool, default true */
    float    poly_s_param;      /* cfg:78  Douglas-Peucker distance, default 0.7 */
    float    poly_z_manual;     /* cfg:81  default -1.5 */
    int32_t  poly_z_avg_allow;  /* cfg:84  bool, default true */
} urf_marker_params;
int urf_default_marker_params(urf_marker_params* p);

/* The line simplification of the polygon step (lidar_segmentation.cpp:475,512,548 call
 * boost::geometry::simplify(line, out, poly_s_param)): Douglas-Peucker with the distance of a point
 * to the SEGMENT between the span's ends, first and last point kept, an interior point kept iff it
 * is the farthest of its span and strictly farther than max_distance; float coordinates
 * (point_xy<float>, data_structures.hpp:38).  xy = n points (x0, y0, x1, y1, ...); keep[i] = 1 for
 * the points of the simplified line.  Boost.Geometry itself is not in the reference checkout:
 * the behaviour is pinned by the worked example of its documentation (tests/test_simplify_kat.py). */
int urf_simplify_line(const float* xy, uint32_t n, float max_distance, uint8_t* keep);

/* ---- stage-wise inspection (parity tests) ----------------------------------
 * After a classify call, copies one intermediate array of scan `scan` to host
 * memory.  Arrays indexed by input point have n_points entries; values of
 * points outside the ROI / not on a ring are unspecified unless noted. */
typedef enum urf_stage {
    URF_STAGE_VALPHA = 1,     /* float  per point: vertical angle, lidar_segmentation.cpp:151-166;
                                 negative for points outside the ROI */
    URF_STAGE_RING = 2,       /* int16  per point: sorted ring index or -1, lidar_segmentation.cpp:226-233 */
    URF_STAGE_AZIMUTH = 3,    /* float  per point: azimuth alpha [deg], lidar_segmentation.cpp:248-269 */
    URF_STAGE_RANGE2D = 4,    /* float  per point: planar d, lidar_segmentation.cpp:245 */
    URF_STAGE_DETECT = 5,     /* uint8  per point on a ring: bit0 star, bit1 x_zero, bit2 z_zero hit (0 elsewhere) */
    URF_STAGE_SECTOR = 6,     /* int16  per point: star sector or -1, star_shaped_search.cpp:171 */
    URF_STAGE_ANGLE_TABLE = 7,/* float[channels]: sorted ring-angle table, lidar_segmentation.cpp:205 */
    URF_STAGE_MAXDIST = 8,    /* float[channels]: maxDistance, lidar_segmentation.cpp:271-274 */
    URF_STAGE_QUADRANTS = 9,  /* float[4]: q1..q4, blind_spots.cpp:13-57 */
    URF_STAGE_BEAM_STOP = 10  /* int16[2*361]: first blocked ring per forward / backward beam
                                 (n_rings = not blocked, -1 = beam not cast) */
} urf_stage;
int urf_read_stage(urf_ctx* ctx, urf_stage what, uint32_t scan, void* host_dst, size_t bytes);
/* Capture mode (default 0 = off: the production path records nothing per input point):
 *   1  every point takes the reference's exact arithmetic and the values are stored: all stages
 *      can be read (URF_STAGE_VALPHA, URF_STAGE_AZIMUTH and URF_STAGE_RANGE2D only in this mode --
 *      without it the pipeline settles most ring / sector / road decisions on float approximations
 *      of these angles and never evaluates the reference's exact value for such a point);
 *   2  the production decisions, with the ring and sector of every input point recorded
 *      (URF_STAGE_RING, URF_STAGE_SECTOR; mode 0 keeps them only in sorted order).
 * The other stages can be read in every mode.  urf_read_stage, urf_ordered_indices and
 * urf_marker_points look at the LAST classify call of the context with the parameters that call
 * ran with; the label buffer handed to that call must still be alive (see LIFETIME above). */
int urf_enable_stage_capture(urf_ctx* ctx, int mode);

/* ---- the fused front end for batches of sweeps in firing order (round 6) -------
 * A batch call (urf_classify_batch_*) whose scans arrive as a spinning LiDAR's driver delivers them -- firing after
 * firing, the 64 lasers of a firing in one fixed order (any order), returns missing where there were none -- is classified
 * by a front end that keeps no ring-sorted copy of the sweep (urban_road_filter_amd/csrc/urf_front.hpp: one lane per
 * laser, the detectors' windows of lidar_segmentation.cpp:280-283 / x_zero_method.cpp:30-67 / z_zero_method.cpp:21-72
 * in registers).  Decided per scan on the device; a scan without that shape takes the general kernels in the same
 * call; labels and summaries are identical either way.  Applies with channels == 64, curbPoints == 5, no stage
 * capture, at most 128 x 2048 points per scan.  mode 0: never; 1 (default): batch calls of at least 192 scans (below that the general kernels are faster: a sweep's fused kernels are
 * a few long dependent chains, which need many sweeps side by side to fill the device; a context whose sweeps have turned out to be
 * row-major takes the fused kernels at any batch size -- the general kernels are 2.3 x slower on that layout even for four sweeps --
 * and on the callback path: urf_classify_pc2(_async) of a row-major sweep, from the context's second or third such sweep on); 2: every
 * batch call it applies to.  urf_read_stage / urf_ordered_indices* / urf_marker_points* read ring-sorted intermediate
 * results: after a call that took the fused front end they first run that call again through the general kernels
 * (the call's INPUT arrays must then still be alive, like its label buffer), and the context stays with the general
 * kernels afterwards (until urf_set_front_mode is called again with a mode other than 0).  A context that has handed a
 * scan back launches the general kernels as full grids next to the fused ones from then on, and one that has handed a
 * whole batch back (unorganised clouds) stops trying in mode 1; urf_set_params with other parameters and
 * urf_set_front_mode with another mode forget both.  urf_front_scans: how many scans of the last batch call took the
 * fused front end (synchronises). */
int urf_set_front_mode(urf_ctx* ctx, int mode);
int urf_front_scans(urf_ctx* ctx, uint32_t* n_fused);

/* ---- per-kernel timing (benchmark) ------------------------------------------
 * With timing on, every classify call brackets each kernel of the pipeline
 * with hipEvents on the context's stream.  urf_kernel_timing() synchronises,
 * adds the elapsed milliseconds of all calls since the last query to
 * ms_sum[0..URF_NUM_KERNELS) and the number of calls to *n_calls, then resets. */
#define URF_NUM_KERNELS 8
int urf_enable_kernel_timing(urf_ctx* ctx, int on);
int urf_kernel_timing(urf_ctx* ctx, double* ms_sum, uint32_t* n_calls);
const char* urf_kernel_name(int index);

/* ---- diagnostics --------------------------------------------------------- */
/* The cotangent (of an angle in degrees, clamped to [1, 179]) from which k_ring_table derives the
 * thresholds on u = -z / rho that decide a point's ring: the same source evaluated on the host, so that
 * its accuracy (1e-15; needed: 1e-7) can be checked without a GPU. */
double urf_ring_threshold_cot(double angle_deg);
/* The host-side gather of the callback path by itself (no context, no device): x / y / z of a PointCloud2-layout
 * message into three arrays of n_points floats.  urf_classify_pc2_async() does this into pinned memory with every
 * message it has to stage (12 bytes per point then cross PCIe, whatever the point_step); records whose x, y, z lie
 * side by side with a fourth word behind them inside the record go four at a time through a 4 x 4 transpose. */
int urf_pc2_to_planes(const uint8_t* data, uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                      uint32_t off_z, float* x, float* y, float* z);
/* Diagnostics of the callback path.  It launches a short kernel sequence first (no repair kernels behind the
 * speculative ring table, none for the work lists of star sectors of more than 384 points, none for rings that hold a
 * point with a NaN azimuth, none for star sectors with equal planar ranges); a sweep that needed what
 * was left out is run again inside urf_classify_pc2_wait() with the full sequence, and so is every later one.
 * n_rerun: sweeps run again so far (per cause the first one and those in flight beside it); sequence: bit 0 the ring table is still speculative,
 * bit 1 the work-list kernels are part of the sequence, bit 2 so is the kernel for rings with NaN azimuths, bit 3 the ring table also
 * stops at the ring count of the previous sweep (a stream of sweeps from one sensor shows the same rings), bit 4 the kernel that orders
 * equal planar ranges of a star sector as std::sort does is part of the sequence (a real sensor's first sweep switches it on).
 * A stream from a real sensor therefore pays one sweep run twice at its start (and the sweeps in flight beside it) and two more
 * near-empty launches per sweep from then on: bench.py's e2e_latency_ms_sensor_like is that steady state, e2e_latency_ms the
 * tie-free one.
 * Either pointer may be NULL. */
int urf_callback_path_state(const urf_ctx* ctx, uint32_t* n_rerun, uint32_t* sequence);
/* Puts kernels into the callback path's sequence before a sweep has asked for them: sequence_bits = 2 (work lists of large star
 * sectors) | 4 (rings with NaN azimuths) | 16 (std::sort's order of equal planar ranges) -- a node that knows its sensor (every real
 * one delivers equal ranges) calls it with 16 once and saves the stream's first sweeps their second run.  Bits are only ever added. */
int urf_callback_path_preset(urf_ctx* ctx, uint32_t sequence_bits);
const char* urf_strerror(int status);
const char* urf_last_error(const urf_ctx* ctx);   /* text of the last HIP failure */
int urf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* URF_H */
