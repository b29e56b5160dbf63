/*
 * urf_internal.hpp -- limits, device-side views and the context shared by the
 * host API (urf_api.hip) and the kernels (urf_kernels.hpp).
 */
#ifndef URF_INTERNAL_HPP
#define URF_INTERNAL_HPP

#include <cstddef>
#include <cstdint>

#include "urf.h"

/* ---- limits --------------------------------------------------------------- */
#define URF_MAX_CHANNELS    128    /* ring keys fit 7 bits + "none" */
#define URF_MAX_SECTORS     1022   /* sector keys fit 10 bits + "none" */
#define URF_MAX_CURB_POINTS 30     /* cfg/LidarFilters.cfg:36 */
/* ring lookup table (k_ring_table -> k_split), over u = -z / rho = cot(vertical angle) in [-4, 4] */
#define URF_LUT_UMAX        4.0f
#define URF_LUT_SCALE       512.0f /* cells per unit of u: 1/512 = 0.11 deg at the horizon, less elsewhere */
#define URF_LUT_CELLS       4100   /* 8 * 512 + 1 cells (u == 4 has its own), padded to a multiple of 4 */
#define URF_DEG_CELLS       361    /* integer degrees 0..360 (blind_spots.cpp:68,177) */

/* one tile = the unit of the stable multi-split by ring / by sector */
#define URF_TILE            2048
#ifndef URF_TILE_THREADS
#define URF_TILE_THREADS    512     /* k_split: one wave per 256 points of the tile */
#endif
#define URF_TILE_GROUPS     (URF_TILE / 64)   /* wave-sized groups per tile */
#define URF_MAX_TILES       4096    /* tiles per scan (k_ring keeps one table entry per tile in LDS) */
#define URF_SCAN_PAD        512     /* scratch elements per scan beyond its tiles: rings start at multiples of 4 */
#define URF_SLOT_NONE       0xFFFFu
#define URF_SLOT_OFF        0x8000u   /* bit 15 of an sslot entry: the point lies on no ring (URF_SLOT_NONE; the fused front end keeps the point's place in its tile below it) */
#define URF_TABLE_LOOKAHEAD 8192    /* k_ring_table gives up waiting for a new ring after this many points (speculation) */

/* The record of a ring-sorted slot (k_split -> k_ring -> k_label), ONE word per point:
 *   bits  0-10  index of the point inside its input tile (where its label byte goes)
 *   bits 11-13  detector hits, OR-ed in by k_ring: 1 star-shaped, 2 x_zero, 4 z_zero
 *   bits 14-31  k_split's float approximation of the azimuth in steps of 360 / 262143 deg
 *               (URF_REC_AZ_UNKNOWN: the point lies too close to the x axis for the approximation,
 *               urf_fast_az_ok -- exact azimuth on demand)
 * The exact azimuth of lidar_segmentation.cpp:245-269 is only ever needed for curb points (beam tables,
 * k_ring) and for the few points whose road decision the approximation leaves open (k_label); both
 * compute it from the slot's x / y.  (r2 kept a float azimuth, a source index and a flag byte per
 * slot: 7 bytes written and 7 read per point instead of 4 and 4.) */
#define URF_REC_SRC_MASK    0x7FFu
#define URF_REC_FLAG_SHIFT  11
#define URF_REC_AZ_SHIFT    14
#define URF_REC_AZ_UNKNOWN  0x3FFFFu
#define URF_REC_AZ_SCALE    (262143.0f / 360.0f)
#define URF_REC_AZ_STEP     (360.0f / 262143.0f)
#define URF_REC_AZ_QERR     9.0e-4f   /* |decoded - encoded azimuth| <= 0.52 steps = 7.2e-4 deg, plus what the decoded value's
                                         distance from the x axis adds to urf_fast_az_eps (1.5e-4) */
/* (urf_kargs::optimistic) */
#define URF_OPT_NO_REPAIR 1u   /* k_table_repair / k_split_repair do not follow k_split */
#define URF_OPT_NO_LISTS 2u    /* k_star_sort_mid / k_star_sort_big do not follow k_star_sort_small */
#define URF_OPT_NO_NAN 4u      /* k_nan_rings does not follow k_ring */
#define URF_OPT_NO_TIES 8u     /* k_star_ties does not follow the sort kernels */
/* internal values of urf_scan_info::status: never seen by a caller */
#define URF_STATUS_REDO_TABLE 0x7f000001
#define URF_STATUS_REDO_LISTS 0x7f000002
#define URF_STATUS_REDO_NAN 0x7f000003
#define URF_STATUS_REDO_HINT 0x7f000004   /* the table was incomplete because the walk stopped at the previous call's ring count */
#define URF_STATUS_REDO_TIES 0x7f000005   /* a star sector holds equal planar ranges where the walk looks: needs k_star_ties */
/* star_first[] carries flags above the index (a scan holds at most 2^23 points).  Equal planar ranges of a sector are ordered
 * as libstdc++'s std::sort orders them (star_shaped_search.cpp:109), which only k_star_ties knows how to do; the sort kernels
 * order them by position and say what they saw:
 *   URF_TIE_FLAG  the sorted prefix the walk may look at holds two equal ranges with DIFFERENT heights: the slopes around them
 *                 depend on their order -- k_star_ties sorts the sector again before the walk and clears the bit;
 *   URF_TIE_NEXT  the point behind the walk's last one has its range (and height): see URF_TIE_POST;
 *   URF_TIE_DONE  k_star_ties has put the sector into std::sort's order.
 * Equal ranges with EQUAL heights (what a sensor's neighbouring firings of one ring on flat ground produce: ~30 per sector)
 * leave the sequence of (range, height) pairs -- all the walk computes with -- the same in any order; only WHICH point stands
 * at the index the walk stops at depends on it, and only if that point has a twin behind it:
 *   URF_TIE_POST  set by the walk kernels together with that index: k_star_ties' second pass (behind the walk) sorts the
 *                 sector as std::sort does and reports the point at that index instead. */
#define URF_TIE_FLAG 0x80000000u
#define URF_TIE_NEXT 0x40000000u
#define URF_TIE_DONE 0x20000000u
#define URF_TIE_POST 0x10000000u
#define URF_TIE_INDEX 0x00ffffffu
#define URF_TIE_CAP 2048u    /* k_star_ties: sectors of up to this many points in LDS; beyond: in global memory */
#define URF_AZ_UNKNOWN      -1.0f     /* decoded value of URF_REC_AZ_UNKNOWN */
#define URF_RING_NONE       0xFFu
#define URF_SEC_NONE        0x3FFu

int urf_validate_params(const urf_params* p);

/* Per-sector constants of the rectangular star beam
 * (star_shaped_search.cpp:32-66 beam_init). */
struct urf_beam {
    int32_t yx;
    float   d;
    float   o;
};

/* Parameters as the kernels see them: the reference's scalars plus the values
 * the reference derives once per scan or per start-up. */
struct urf_dev_params {
    urf_params p;
    float    slope_param;   /* star_shaped_search.cpp:160 */
    float    Kfi;           /* star_shaped_search.cpp:65  */
    float    sector_margin; /* urf_fast_sector: URF_FAST_SECTOR_ERR scaled with sectors / 360 */
    float    fwd_limit;     /* 360 - beamZone  (blind_spots.cpp:68)  */
    float    bwd_limit;     /* 0 + beamZone    (blind_spots.cpp:177) */
    float    inv_cp;        /* 1 / (float)curbPoints (z_zero_method.cpp:52) */
    float    x_angle_thr;   /* "alpha <= angleFilter1" (x_zero_method.cpp:58-61) as a threshold on the cosine: bracket >= x_angle_thr */
    float    z_angle_thr;   /* ... angleFilter2 (z_zero_method.cpp:63-66); urf_api.hip: urf_angle_threshold */
    uint32_t sec_keybits;   /* bits needed for sector keys incl. "none" */
    uint32_t ring_keybits;
    uint32_t exp_flags;     /* test hook (urf_set_debug_flags): bit 2 forces the general star sort path; 0 in production */
};

/* Everything a kernel needs to find a scan's data.  All pointers are device
 * memory owned by the context except x/y/z/labels (caller's).
 *
 * Scratch layout ("tile-local"): scan s owns the scratch elements
 * [s * sstride, (s + 1) * sstride), sstride = tiles * URF_TILE + URF_SCAN_PAD, independent of where
 * the caller keeps the scan (offsets[]).  Tile t of the scan (input points [t * URF_TILE, ...)) owns
 * [s * sstride + t * URF_TILE, ... + URF_TILE) of every per-point array:
 *   ring-sorted   rx ry rz rec    slot j = the tile's points that lie on a ring, ordered by ring,
 *                                 input order inside a ring (stable); rec = URF_REC_* word
 *   sector-sorted sr sz sslot     the tile's points that take part in the star-shaped search,
 *                                 ordered by sector, input order inside; sslot = the point's
 *                                 ring-sorted slot (URF_SLOT_NONE if it lies on no ring)
 * so that ONE pass over x/y/z (k_split) can write both without knowing any total.  Ring c of the
 * scan = the concatenation over the tiles of run [troff[t][c], troff[t][c+1]); k_index turns the
 * rings' per-tile run tables into per-ring tables (prefix over the tiles, start inside the tile);
 * a sector's runs are read from tsoff directly (it meets few tiles).  What k_ring produces per point
 * (detector hits) goes into the point's ring-sorted record; what the star sort produces is
 * contiguous per sector (wsg, ssrt16 / ssrt: at s * sstride + sec_off[k] + i). */
/* A sector's points sit in one run per tile (k_split's sector-sorted order).  The first two non-empty
 * runs: point i of the sector is element a0 + i of the sector-sorted arrays for i < c0, a1 + (i - c0)
 * beyond (indices relative to the scan's scratch); nruns > 2: the sort walks the per-tile tables. */
#define URF_RUNS_FLAG 0x40000000u   /* in nruns (k_index): more than two runs, but one short run per tile: k_star_sort_runs */
struct urf_sec_run { uint32_t a0, c0, a1, nruns; };

/* k_star_sort_* -> k_star_walk, per point of a sector in sorted order */
struct alignas(8) urf_sg { float slp, g; };
struct alignas(8) urf_wu { float w, u; };   /* (float)(i - 1), 1 / (float)i: the factors of step i of a walk */

/* What the reference's beam scans see of a ring whose azimuth-sorted array holds NaN entries: the forward scans
 * (blind_spots.cpp:107,124,146,164: "alpha <= window end" ends the loop, and is false for a NaN) only the points in
 * front of the first NaN -- the smallest azimuths, up to f_hi --, the backward scans (:216,233,255,273) only those
 * behind the last one -- the largest, from b_lo.  A ring without such a point: (+inf, -inf). */
struct alignas(8) urf_vis { float f_hi, b_lo; };

/* k_front -> k_front_finish: (index of the point inside its scan, URF_FC_*); k_front_finish's list of curb points: (azimuth bits, ring) */
struct alignas(8) urf_u2 { uint32_t x, y; };

/* k_beams -> k_label, per (ring, integer degree) */
struct urf_win { float hi, lo; };

struct urf_kargs {
    /* input */
    const float* x;
    const float* y;
    const float* z;
    const uint32_t* offsets;    /* ragged: [n_scans+1]; else NULL */
    uint32_t n_per_scan;
    uint32_t n_scans;
    uint32_t max_len;
    uint32_t tiles;             /* tiles per scan = ceil(max_len / URF_TILE): stride of the per-tile tables */
    uint32_t sstride;           /* scratch elements per scan */
    uint32_t table_lookahead;   /* k_ring_table: stop after this many points without a new leader (0: never), see there */
    uint32_t table_hint;        /* ... and as soon as the table holds as many rings as the row's previous call found (*ring_hint), see there */
    uint32_t optimistic;        /* the callback path's short launch sequence: URF_OPT_* of what it leaves out; k_index turns a scan that
                                   needed it into URF_STATUS_REDO_*, which urf_classify_pc2_wait() answers with the full sequence */
    uint32_t capture;           /* 0 production; 1 every point takes the exact sequence, values recorded;
                                   2 production decisions, ring / sector keys recorded */
    uint32_t  front;            /* this call launches it (urf_api.hip decides: 64 channels, curbPoints 5, no stage capture, a batch) */
    uint32_t  front_tpb;        /* tiles per block of k_front */
    uint32_t  front_cand_cap;   /* entries per scan of front_cand / front_all */
    uint32_t  front_pad_;
    /* output */
    uint8_t* labels;
    urf_scan_info* info;        /* context copy, [n_scans] */
    /* per point, input order (scratch indexing), stage capture only */
    float*    valpha;
    uint16_t* seckey;
    uint8_t*  ringkey;
    float*    caz;              /* capture mode 1: exact azimuth per ring-sorted slot (k_ring) */
    /* per point, ring-sorted inside the tile */
    float*    rx;
    float*    ry;
    float*    rz;
    uint32_t* rec;              /* URF_REC_*: source index | detector hits | approximate azimuth */
    float*    rd2;              /* planar range, stage capture only (may be NULL) */
    /* per point, sector-sorted inside the tile */
    float*    sr;
    float*    sz;
    uint16_t* sslot;
    /* per point, sector-major */
    uint16_t* ssrt16;           /* sectors of at most two runs and at most URF_STAR_MID_CAP points (every sector of an
                                 * organised sweep): position inside the sector (input order) of the i-th point in sorted order;
                                 * the walk turns the one it needs into a slot through sec_run and sslot */
    uint32_t* ssrt;             /* all other sectors: tile-local ring-sorted index (t * URF_TILE + slot) of the i-th point in
                                 * sorted order, 0xffffffff = on no ring */
    urf_sg*   wsg;              /* .x slope between the (i-1)-th and i-th point of the sector in sorted order, .y (r_i - r_{i-1}) * kdist:
                                 * side by side, so that the walk fetches both with one 8-byte load (16 steps of a sector = one 128-byte line) */
    const urf_wu* walk_tab;     /* [max_points + 32] ((float)(i - 1), 1 / (float)i): the walk's wave-uniform factors (k_walk_table, once per context) */
    /* per scan x tile (k_split) */
    uint32_t* tile_roi;         /* [S][tiles] ROI points of the tile */
    uint16_t* troff;            /* [S][tiles][C+1] first ring-sorted slot of ring c in the tile; [C] = ring points of the tile */
    uint16_t* tsoff;            /* [S][tiles][K+1] same for star sectors */
    unsigned long long* tmaxs;  /* [S][tiles][C] largest x*x + y*y (binary64 bits) among the tile's points of ring c (k_split -> k_ring: maxDistance) */
    /* per scan x key x tile (k_index) */
    uint32_t* rpre;             /* [S][C][tiles+1] points of ring c in the tiles before t; [ntiles] = ring_cnt */
    uint16_t* rstart;           /* [S][C][tiles]   = troff[t][c] */
    /* per scan */
    float*    angle;            /* [S][channels] sorted ring-angle table */
    uint8_t*  ring_lut;         /* [S][URF_LUT_CELLS] number of table entries no point of the cell (of u) can match */
    float*    ring_thr;         /* [S][channels][4] per table entry, in u = cot(vertical angle), which FALLS as the angle
                                 * grows: u < .x the entry lies surely below the point's window (skip it), u in
                                 * [.y, .z] it surely matches, u > .w it surely lies above the window
                                 * (urf_device.hpp: urf_ring_thresholds) */
    uint32_t* ring_cnt;         /* [S][channels] */
    uint32_t* ring_off;         /* [S][channels+1] ring points of the scan in front of ring c (exclusive scan of ring_cnt) */
    uint32_t* sec_cnt;          /* [S][sectors] */
    urf_sec_run* sec_run;       /* [S][sectors] the sector's first two runs (k_index -> k_star_sort_small) */
    uint32_t* sec_off;          /* [S][sectors+1] */
    int32_t*  star_hit;         /* [S][sectors] ring-major position of the sector's curb point; -1 = none or on no ring */
    uint32_t* star_first;       /* [S][sectors] last sorted index the walk may visit */
    uint32_t* star_list_mid;    /* [S*sectors] work list: scan*sectors+sector of sectors with 385..2048 points */
    uint32_t* star_list_big;    /* [S*sectors] ... with more than 2048 points */
    uint32_t* star_list_runs;   /* [S*sectors] ... of many short runs (URF_RUNS_FLAG; star_count[7]) */
    uint32_t* tie_list;         /* [S*sectors] scan*sectors+sector of the sectors that carry URF_TIE_FLAG (sort kernels -> k_star_ties, first pass) */
    uint32_t* tie_post;         /* [S*sectors] ... URF_TIE_POST (walk kernels -> second pass) */
    uint32_t* star_count;       /* [8] lengths of the two lists, [2] = length of redo_list, [3] = length of nan_list, [4] = length of
                                 * tie_list, [5] = of tie_post, [6] = of front_list (zeroed per call) */
    uint32_t* table_upto;       /* [S] first point a speculative k_ring_table did not look at (0xffffffff: none) */
    uint32_t* table_redo;       /* [S] k_split: the speculative table of the scan is incomplete */
    uint32_t* redo_list;        /* [S] such scans (k_table_repair) */
    uint32_t* table_cause;      /* [S] which rule ended a speculative walk: 1 the quiet look-ahead, 2 the ring-count hint, 3 the rows' first points */
    uint32_t* ring_hint;        /* [1] per scratch row: the largest n_rings of the row's previous call (k_ring_table reads it, k_split
                                 * zeroes it, k_index collects the new one) */
    uint32_t* spec_failed;      /* host-mapped flags: [0] a look-ahead speculation failed, [1] a ring-count hint did */
    float*    big_r;            /* sector-major copies of the sectors on the "big" list (sorted in place) */
    float*    big_z;
    uint32_t* big_i;
    /* rings that hold a point with x == y == 0, whose azimuth is NaN (normally none): the reference's per-ring quicksort
     * parks such a point at an input-order-dependent place and its beam scans stop there -- k_nan_rings reproduces both */
    uint32_t* nan_mask;         /* [S][4] bit c: ring c of the scan holds such a point (k_split's exact pass; zeroed by k_ring_table) */
    uint32_t* nan_list;         /* [2 * S * channels] scan * channels + ring of the rings whose bit was newly set (k_split -> k_nan_rings) */
    urf_vis*  vis;              /* [S][channels] what of ring c the beam scans see (k_ring: everything; k_nan_rings; -> k_beams) */
    float*    maxdist;          /* [S][channels] */
    float*    quad;             /* [S][4] */
    uint32_t* curb_cnt;         /* [S][channels] curb points of the ring with a valid azimuth; 0xffffffff: more than k_ring's list holds, see sufmin / premax */
    float*    curb_az;          /* [S][channels][URF_CURB_LIST] their exact azimuths (k_ring -> k_beams) */
    float*    sufmin;           /* [S][channels][361] only for rings whose list overflowed */
    float*    premax;           /* [S][channels][361] */
    int16_t*  stop_f;           /* [S][361] */
    int16_t*  stop_b;           /* [S][361] */
    unsigned long long* roi_bits; /* [S][tiles][32] bit i of a tile: input point i lies in the region of interest (k_split -> k_label) */
    urf_win*  win;              /* [S][channels][361] k_beams -> k_label.  .x: upper end of the window of the
                                 * nearest forward beam at or below degree d that reached beyond the ring
                                 * (-inf: none); .y: lower end for the nearest backward beam at or above d (+inf) */
    /* tables */
    const float*    newY;       /* [max_points] x_zero_method.cpp:24-27 */
    const urf_beam* beams;      /* [sectors] */
    /* the fused front end (urf_front.hpp): scans that arrive firing by firing skip k_split / k_ring / k_label */
    uint32_t* front_ok;         /* [S] 1: the scan has the shape so far (k_ring_table sets it, k_front / k_table_repair clear it); the legacy
                                 * kernels skip a scan whose flag is set, the fused ones a scan whose flag is clear */
    uint32_t* front_pres;       /* [S][tiles][64] bit j of word (t, l): lane l of firing j of tile t is a ring point */
    unsigned long long* front_maxs; /* [S][tiles][64] per block and lane the largest x*x + y*y of its ring points (binary64 bits) */
    uint32_t* front_lane_ring;  /* [S][64] the ring of lane l (0xffffffff: none yet) */
    uint32_t* front_ring_lane;  /* [S][channels] the lane of ring r */
    urf_u2*   front_cand;       /* [S][front_cand_cap] (input index, URF_FC_*): points whose detector decision is pending (k_front -> k_front_finish) */
    urf_u2*   front_all;        /* [S][front_cand_cap] (azimuth bits, ring) of every curb point (k_front_finish: rings whose list overflowed) */
    uint32_t* front_ncand;      /* [S] */
    uint32_t* front_st;         /* [S][URF_FRONT_ST_WORDS] k_front_finish part 1 -> part 2 */
    uint32_t* front_list;       /* [S] the scans whose flag is clear (k_front_collect; star_count[6] = how many): the list-driven legacy kernels' work */
    uint32_t* front_state;      /* host-mapped: [0] some scan of some call was handed back, [1] every scan of some call was, [2] a scan looked
                                 * row-major (URF_FRONT_ROWS), [3] the row-major speculation failed on one, [4] a scan took the row-major layout */
    uint32_t  front_lists;      /* this call launches the legacy kernels list-driven (k_split_list, k_ring_list, k_label_list) */
    /* row-major organised sweeps (height = the sensor's 64 lasers, width = firings: point l * F + f): front_ok[s] == URF_FRONT_ROWS,
     * k_transpose writes the firing-order copy the fused kernels read instead of x / y / z, k_label_front stores the labels
     * where the points came from.  Everything in between is indexed by firing * 64 + laser. */
    uint32_t  front_rows;       /* this call's sequence holds k_rows_probe and k_transpose: k_ring_table may choose the layout (else it only reports
                                 * that it saw such a scan: front_state[2], the next call's sequence holds the kernels) */
    uint32_t  front_sight;      /* this call takes the general kernels but k_ring_table still reports a row-major sighting (a batch below mode 1's threshold:
                                 * row-major sweeps gain from the fused kernels at ANY batch size, tools/r6_min_scans.py --rows) */
    float*    rows_v;           /* [S][64] k_rows_probe: the vertical angles of the rows' first region-of-interest points, in row order (the table's leaders) */
    uint32_t* rows_ok;          /* [S] ... how many + 1; 0: the scan is not row-major */
    float*    tx;               /* [S * sstride] firing-order copies (scratch stride) */
    float*    ty;
    float*    tz;
};
#define URF_FRONT_ROWS 2u

#endif /* URF_INTERNAL_HPP */
