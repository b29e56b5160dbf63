/*
 * urf_api.hip -- host side of the C ABI (include/urf.h): context, scratch
 * memory, kernel sequencing.  No CPU fallback: every entry point that
 * classifies fails with URF_ERR_NO_DEVICE / URF_ERR_HIP when there is no GPU.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#include <xmmintrin.h>
#endif

#include "urf.h"
#ifdef URF_ENABLE_TEST_HOOKS
#include "urf_test_hooks.h"
#endif
#include "urf_internal.hpp"
#include "urf_kernels.hpp"
#include "urf_front.hpp"

#define URF_ASYNC_SLOTS 4
static_assert(URF_ASYNC_SLOTS == URF_MAX_IN_FLIGHT, "include/urf.h documents the number of sweeps in flight");

struct urf_ctx {
    int device = 0;
    uint32_t max_points = 0, max_batch = 0;
    size_t total = 0;               /* scratch elements: sstride * max_batch */
    uint32_t max_tiles = 0;
    uint32_t sstride = 0;           /* scratch elements per scan: max_tiles * URF_TILE + URF_SCAN_PAD */
    uint32_t debug_flags = 0;       /* urf_set_debug_flags */
    unsigned n_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    urf_params params;
    urf_dev_params dp;
    urf_kargs k;                    /* device pointers (context-owned part) */
    /* owned device memory */
    std::vector<void*> allocs;
    float *sx = nullptr, *sy = nullptr, *sz = nullptr;   /* SoA staging for PointCloud2 input */
    /* The single-scan (callback) path: URF_ASYNC_SLOTS sweeps in flight, so that -- with a context created
     * for several scans -- the copies and kernels of several sweeps overlap on the device (a single sweep's
     * kernels are a few dozen workgroups each).  Per slot: pinned host staging for the message bytes and for the
     * results, device copies of both, the captured launch sequence. */
    struct slot_t {
        uint8_t* h_in = nullptr;        /* pinned, h_in_cap bytes */
        size_t h_in_cap = 0;
        uint8_t* d_raw = nullptr;       /* device, d_raw_cap bytes */
        size_t d_raw_cap = 0;
        uint8_t* h_labels = nullptr;    /* pinned, max_points */
        uint8_t* d_labels = nullptr;    /* device, max_points */
        urf_scan_info* h_info = nullptr;   /* pinned */
        hipEvent_t ev_done = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t key[3] = { 0, 0, 0 };  /* what the captured sequence was built for */
        urf_kargs cap_a;                /* ... and the kernel arguments / parameters it runs with */
        urf_dev_params cap_dp;
        bool pending = false, used = false;
        uint64_t gen = 0;               /* the row's submission number of the sweep in this slot */
        uint32_t n_points = 0, ticket = 0;
        uint32_t point_step = 0, off_x = 0, off_y = 0, off_z = 0;   /* layout of the message in d_raw */
        bool planes = false;            /* d_raw holds x[n] y[n] z[n] (a staged message) instead of the records */
    } slots[URF_ASYNC_SLOTS];
    bool streams_made = false;
    /* the callback path launches a short sequence first: no repair kernels behind the speculative ring table, no
     * kernels for the work lists of oversized star sectors (run_pipeline); a sweep that needed one of them comes back
     * with an internal status and is run again with it, as are all later ones (urf_classify_pc2_wait) */
    bool slot_lists = false;
    bool slot_nan = false;          /* ... k_nan_rings (a sweep with a ring point on the sensor's axis) */
    bool slot_ties = false;         /* ... k_star_ties (a sweep with equal planar ranges in a star sector: any real sensor's) */
    uint32_t n_rerun = 0;           /* sweeps urf_classify_pc2_wait had to run again */
    /* Slot i works on scratch row i % rows, rows = min(max_batch, URF_ASYNC_SLOTS); row 0 runs on the
     * context's stream, every other row on a stream of its own (slots that share a row share its
     * stream: they are serialised).  Everything else the context launches runs on `stream`; the two
     * kinds of work are ordered against each other by events (order_after_slots / order_row_after_main). */
    hipStream_t row_stream[URF_ASYNC_SLOTS] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev_main = nullptr;          /* recorded on `stream` for a row stream to wait on */
    uint64_t main_seq = 1;                 /* bumped by every launch on `stream` that touches scratch rows >= 1 or the staging */
    uint64_t row_seen[URF_ASYNC_SLOTS] = { 0, 0, 0, 0 };
    /* Slots that share a scratch row (max_batch < URF_ASYNC_SLOTS) are serialised on the row's stream, and a later
     * sweep overwrites the row.  Labels and summary of every sweep are safe (each slot has its own result buffers,
     * filled in stream order); what reads the ROW afterwards (urf_read_stage / urf_ordered_indices /
     * urf_marker_points) checks that the sweep published as "the last call" is still the row's latest submission. */
    uint64_t row_gen[URF_ASYNC_SLOTS] = { 0, 0, 0, 0 };   /* submissions on the row so far */
    bool last_is_slot = false;             /* "the last call" is a sweep of the callback path ... */
    uint32_t last_row = 0;                 /* ... on this row ... */
    uint64_t last_gen = 0;                 /* ... which was the row's submission number last_gen */
    uint32_t next_ticket = 0;
    uint64_t epoch = 1;             /* bumped by everything a captured sequence depends on */
    /* lazily, sized for the largest number of scans asked for so far: scratch of the index-list and
     * marker-point outputs (sstride entries resp. channels x 361 cells per scan) */
    unsigned long long* ord_keys = nullptr;
    uint32_t* ord_pos = nullptr;
    uint32_t* ord_cls = nullptr;   /* [scans][URF_MAX_CHANNELS][2] road / curb points per ring (k_ring_order -> k_ordered_lists) */
    uint32_t ord_scans = 0;
    uint32_t* ord_lists = nullptr;  /* single-scan entry point: 3 x sstride + 4 */
    float* mk_d = nullptr;
    uint32_t* mk_pos = nullptr;
    uint8_t* mk_red = nullptr;
    uint32_t mk_scans = 0;
    float* mk_out = nullptr;        /* single-scan entry point: 361 x 4 floats + 1 count */
    uint32_t* compact_cnt = nullptr;   /* [max_batch][max_tiles][4] */
    float* d_newY = nullptr;
    urf_beam* d_beams = nullptr;
    uint32_t beams_cap = 0;
    int capture = 0;                /* urf_enable_stage_capture */
    bool timing = false;
    std::vector<std::vector<hipEvent_t>> timing_events;   /* sets of URF_NUM_KERNELS+1 events, created once and reused */
    size_t timing_used = 0;         /* sets recorded since the last urf_kernel_timing() */
    uint32_t* offsets_copy = nullptr;   /* [max_batch + 1] the ragged offsets of the last call (context-owned) */
    /* k_ring_table speculates (stops when no new ring has shown up for a while, k_split checks);
     * a scan that proves it wrong is repaired in the same call and raises this host-visible flag,
     * after which the context builds its tables the long way */
    uint32_t* h_spec_failed = nullptr;  /* pinned, device-mapped: [0] look-ahead, [1] ring-count hint */
    /* the fused front end (urf_front.hpp, urf_set_front_mode): 0 never, 1 batches of at least URF_FRONT_MIN_SCANS scans (default),
     * 2 every batch call it applies to.  The entry points that read ring-sorted intermediate results (urf_read_stage,
     * urf_ordered_indices*, urf_marker_points*) run the last call again through the legacy kernels when it took the fused ones,
     * and the context keeps to the legacy kernels from then on (want_ring_sorted). */
    int front_mode = 1;
    uint32_t front_tpb = 0;         /* tiles per block of k_front; 0: by batch size (URF_FRONT_TPB_*), else what URF_FRONT_TPB says */
    bool want_ring_sorted = false;
    /* k_front hands a scan without the shape back to the legacy kernels.  As long as no call has done so, those are launched
     * list-driven (a few persistent workgroups that find an empty list) instead of as full grids of workgroups that look at the
     * scan's flag and leave; after the first such scan (h_spec_failed[2]) they come as full grids, and once a whole batch has been
     * handed back (h_spec_failed[3]: unorganised clouds) the context stops trying.  urf_set_params / urf_set_front_mode start over. */
    bool front_direct = false, front_off = false;
    /* row-major organised sweeps (k_ring_table's third rule): once one has been sighted (h_spec_failed[4]) the batch calls' sequence
     * holds k_transpose and k_ring_table may choose the layout */
    bool front_rows = false, rows_oom = false;
    /* ... and once a scan has really taken the layout (h_spec_failed[6]) -- or for a few calls after the sighting -- batches below mode 1's threshold
     * take the fused kernels too (row-major sweeps gain from them at any batch size, sweeps in firing order only from 192 per call on) */
    bool rows_used = false;
    uint32_t rows_probation = 0;
    /* k_front_finish's first part runs on a stream of its own next to the star-shaped search (run_pipeline) */
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool speculate = true;
    bool use_hint = true;           /* k_ring_table also stops at the ring count of the row's previous call (until that fails once) */
    /* last call, for the entry points that read its intermediate results (urf_read_stage,
     * urf_ordered_indices, urf_marker_points): the kernel arguments and parameters it ran with */
    uint32_t last_scans = 0;
    urf_kargs last_a;
    urf_dev_params last_dp;
    std::string last_error;
    /* host-side cost of the callback path, phase by phase (only the build with the test hooks fills them:
     * URF_HOST_TIMES=1, printed by its benchmark loop; the context's layout is the same in both builds) */
    double ht[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    bool ht_on = false;
};

#define URF_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return e_ == hipErrorOutOfMemory ? URF_ERR_OOM : URF_ERR_HIP;               \
        }                                                                               \
    } while (0)

template <class T>
static int dev_alloc(urf_ctx* c, T** p, size_t count)
{
    void* v = nullptr;
    URF_HIP(c, hipMalloc(&v, (count ? count : 1) * sizeof(T)));
    c->allocs.push_back(v);
    *p = (T*)v;
    return URF_OK;
}

/* star_shaped_search.cpp:32-66 beam_init: per-sector constants of the
 * rectangular beam; `fi` is a float, so tan/sin/cos(fi) are the float
 * overloads and tan(0.5*M_PI - fi) is the double one. */
static void beam_init(std::vector<urf_beam>& beams, int rep, float width)
{
    beams.resize((size_t)rep);
    const float off = (float)(0.5 * (double)width);
    for (int i = 0; i < rep; i++) {
        const float fi = (float)((double)(i * 2) * M_PI / (double)rep);
        if (std::fabs(std::tan(fi)) > 1) {
            beams[i].yx = 1;
            beams[i].d = (float)std::tan(0.5 * M_PI - (double)fi);
            beams[i].o = std::fabs(off / std::sin(fi));
        } else {
            beams[i].yx = 0;
            beams[i].d = std::tan(fi);
            beams[i].o = std::fabs(off / std::cos(fi));
        }
    }
}

/* x_zero_method.cpp:58-61 / z_zero_method.cpp:63-66 test "alpha <= angleFilter" with alpha = (float)((double)(acosf(b) * 180.0f) /
 * M_PI), b the clamped cosine.  alpha falls (weakly) as b grows -- checked for EVERY float of [-1, 1] against include/urf_libm.h's
 * acosf by tools/check_acos_threshold.c, together with the bisection below for fourteen filter angles -- so the test is "b >= T" with
 * T = the smallest float of [-1, 1] whose alpha passes: no arc cosine on the device (a fifth of the candidate chain of k_ring).  A NaN
 * cosine fails either form.  Returns 2 when no b passes. */
static float urf_angle_threshold(float angle_filter)
{
    auto alpha_of = [](float b) { return (float)((double)(urf_acosf(b) * 180.0f) / M_PI); };   /* (IEEE division: what urf_div_pi equals) */
    auto from_key = [](uint32_t k) {   /* the floats of [-1, 1] in ascending order */
        const uint32_t one = 0x3f800000u;
        const uint32_t u = k <= one ? 0x80000000u | (one - k) : k - one - 1u;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    if (!(alpha_of(1.0f) <= angle_filter))
        return 2.0f;
    uint32_t lo = 0, hi = 2u * 0x3f800000u + 1u;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (alpha_of(from_key(mid)) <= angle_filter)
            hi = mid;
        else
            lo = mid + 1;
    }
    return from_key(lo);
}

static int upload_params(urf_ctx* c)
{
    const urf_params& p = c->params;
    urf_dev_params& dp = c->dp;
    dp.p = p;
    dp.slope_param = (float)((double)p.angleFilter3 * (M_PI / 180));   /* star_shaped_search.cpp:160 */
    dp.Kfi = (float)((double)p.sectors / (2 * M_PI));                  /* star_shaped_search.cpp:65 */
    dp.fwd_limit = 360.0f - p.beamZone;                                /* blind_spots.cpp:68 */
    dp.bwd_limit = 0.0f + p.beamZone;                                  /* blind_spots.cpp:177 */
    dp.inv_cp = 1.0f / (float)p.curbPoints;                            /* z_zero_method.cpp:52 */
    dp.x_angle_thr = urf_angle_threshold(p.angleFilter1);
    dp.z_angle_thr = urf_angle_threshold(p.angleFilter2);
    /* keys 0..K-1 plus "none" (mapped to K) must be distinguishable */
    dp.sec_keybits = 1;
    while ((1u << dp.sec_keybits) <= (unsigned)p.sectors)
        dp.sec_keybits++;
    dp.ring_keybits = 1;
    while ((1u << dp.ring_keybits) <= (unsigned)p.channels)
        dp.ring_keybits++;
    dp.exp_flags = c->debug_flags;
    dp.sector_margin = URF_FAST_SECTOR_ERR * (p.sectors > 360 ? (float)p.sectors / 360.0f : 1.0f);
    std::vector<urf_beam> beams;
    beam_init(beams, p.sectors, p.beam_width);
    URF_HIP(c, hipMemcpyAsync(c->d_beams, beams.data(), beams.size() * sizeof(urf_beam), hipMemcpyHostToDevice, c->stream));
    URF_HIP(c, hipStreamSynchronize(c->stream));   /* `beams` is a stack object */
    return URF_OK;
}

/* the firing-order copies of row-major organised sweeps (k_transpose): allocated when the first such sweep has been sighted */
static int ensure_rows_arrays(urf_ctx* c)
{
    if (c->k.tx)
        return URF_OK;
    int rc;
    if ((rc = dev_alloc(c, &c->k.rows_v, (size_t)c->max_batch * 64)) != URF_OK || (rc = dev_alloc(c, &c->k.rows_ok, (size_t)c->max_batch)) != URF_OK ||
        (rc = dev_alloc(c, &c->k.ty, c->total)) != URF_OK || (rc = dev_alloc(c, &c->k.tz, c->total)) != URF_OK ||
        (rc = dev_alloc(c, &c->k.tx, c->total)) != URF_OK)   /* (tx last: its pointer says that all of them are there) */
        return rc;
    return URF_OK;
}

static int ensure_capture_arrays(urf_ctx* c)
{
    if (c->k.valpha)
        return URF_OK;
    int rc;
    if ((rc = dev_alloc(c, &c->k.valpha, c->total)) != URF_OK || (rc = dev_alloc(c, &c->k.seckey, c->total)) != URF_OK ||
        (rc = dev_alloc(c, &c->k.ringkey, c->total)) != URF_OK || (rc = dev_alloc(c, &c->k.rd2, c->total)) != URF_OK ||
        (rc = dev_alloc(c, &c->k.caz, c->total)) != URF_OK)
        return rc;
    return URF_OK;
}

extern "C" int urf_create(urf_ctx** out, int device_id, uint32_t max_points, uint32_t max_batch)
{
    if (!out || max_points == 0 || max_batch == 0)
        return URF_ERR_INVALID_ARG;
    *out = nullptr;
    const unsigned long long max_tiles = ((unsigned long long)max_points + URF_TILE - 1) / URF_TILE;
    const unsigned long long sstride = max_tiles * URF_TILE + URF_SCAN_PAD;
    if (max_tiles > URF_MAX_TILES || sstride * max_batch >= (1ull << 32))
        return URF_ERR_CAPACITY;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
        return URF_ERR_NO_DEVICE;
    urf_ctx* c = new urf_ctx();
    c->device = device_id;
    c->max_points = max_points;
    c->max_batch = max_batch;
    c->max_tiles = (uint32_t)max_tiles;
    c->sstride = (uint32_t)sstride;
    c->total = (size_t)sstride * max_batch;
    int rc = URF_OK;
    auto fail = [&](int code) {
        urf_destroy(c);
        return code;
    };
    if (hipSetDevice(device_id) != hipSuccess)
        return fail(URF_ERR_NO_DEVICE);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess)
        return fail(URF_ERR_HIP);
    c->stream = c->own_stream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
            c->n_cus = (unsigned)prop.multiProcessorCount;
    }
    urf_default_params(&c->params);
    std::memset(&c->k, 0, sizeof(c->k));
    std::memset(&c->last_a, 0, sizeof(c->last_a));
    urf_kargs& k = c->k;
    const size_t T = c->total, S = max_batch, tiles = c->max_tiles;
    const size_t C = URF_MAX_CHANNELS, K = URF_MAX_SECTORS;
#define A(ptr, count)                                     \
    if ((rc = dev_alloc(c, &(ptr), (count))) != URF_OK)   \
        return fail(rc);
    A(k.rx, T) A(k.ry, T) A(k.rz, T) A(k.rec, T)
    A(k.sr, T) A(k.sz, T) A(k.sslot, T) A(k.ssrt16, T) A(k.ssrt, T) A(k.wsg, T)
    A(k.big_r, T) A(k.big_z, T) A(k.big_i, T)
    A(k.tile_roi, S * tiles) A(k.roi_bits, S * tiles * (URF_TILE / 64)) A(k.troff, S * tiles * (C + 1)) A(k.tsoff, S * tiles * (K + 1)) A(k.tmaxs, S * tiles * C)
    A(k.rpre, S * C * (tiles + 1)) A(k.rstart, S * C * tiles)
    A(k.angle, S * C) A(k.ring_thr, S * C * 4) A(k.ring_lut, S * URF_LUT_CELLS) A(k.ring_cnt, S * C) A(k.ring_off, S * (C + 1))
    A(k.sec_cnt, S * K) A(k.sec_run, S * K) A(k.sec_off, S * (K + 1)) A(k.star_hit, S * K)
    A(k.star_first, S * K) A(k.star_list_mid, S * K) A(k.star_list_big, S * K) A(k.star_list_runs, S * K) A(k.tie_list, S * K) A(k.tie_post, S * K) A(k.star_count, 8 * URF_ASYNC_SLOTS)   /* eight counters per scratch row in use at once */
    A(k.table_upto, S) A(k.table_redo, S) A(k.redo_list, S) A(k.table_cause, S) A(k.ring_hint, URF_ASYNC_SLOTS)
    A(k.nan_mask, S * 4) A(k.nan_list, 2 * S * C) A(k.vis, S * C)
    A(k.maxdist, S * C) A(k.quad, S * 4)
    A(k.curb_cnt, S * C) A(k.curb_az, S * C * URF_CURB_LIST)
    A(k.sufmin, S * C * URF_DEG_CELLS) A(k.premax, S * C * URF_DEG_CELLS)
    A(k.stop_f, S * URF_DEG_CELLS) A(k.stop_b, S * URF_DEG_CELLS)
    A(k.win, S * C * URF_DEG_CELLS)
    A(k.info, S)
    k.front_cand_cap = max_points / 8 > 4096 ? max_points / 8 : 4096;
    A(k.front_ok, S) A(k.front_pres, S * tiles * 64) A(k.front_maxs, S * tiles * 64) A(k.front_lane_ring, S * 64) A(k.front_ring_lane, S * C)
    A(k.front_cand, S * k.front_cand_cap) A(k.front_all, S * k.front_cand_cap) A(k.front_ncand, S) A(k.front_list, S) A(k.front_st, S * 72)
    A(c->offsets_copy, S + 1)
    A(c->compact_cnt, S * tiles * 4)
    A(c->d_newY, (size_t)max_points) A(c->d_beams, K)
#undef A
    {
        urf_wu* tab = nullptr;
        const size_t nt = (size_t)max_points + 32;
        if ((rc = dev_alloc(c, &tab, nt)) != URF_OK)
            return fail(rc);
        hipLaunchKernelGGL(k_walk_table, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, tab, (unsigned)nt);
        k.walk_tab = tab;
    }
    k.sstride = c->sstride;
    if (const char* e = std::getenv("URF_FRONT_TPB"))   /* tuning experiments: tiles per block of k_front */
        c->front_tpb = (uint32_t)std::atoi(e) > 0 ? (uint32_t)std::atoi(e) : c->front_tpb;
    {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 8 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess)
            return fail(URF_ERR_HIP);
        c->h_spec_failed = (uint32_t*)hp;   /* [2], [3]: the fused front end's two flags (front_direct, front_off); [4], [5]: row-major sweeps sighted / failed */
        for (int i = 0; i < 8; i++)
            c->h_spec_failed[i] = 0;
        if (hipMemset(k.ring_hint, 0, URF_ASYNC_SLOTS * sizeof(uint32_t)) != hipSuccess)
            return fail(URF_ERR_HIP);
        void* dp_ = nullptr;
        if (hipHostGetDevicePointer(&dp_, hp, 0) != hipSuccess)
            return fail(URF_ERR_HIP);
        k.spec_failed = (uint32_t*)dp_;
        k.front_state = k.spec_failed + 2;
    }
    /* x_zero_method.cpp:24-27: newY[j] = newY[j-1] + 0.0100 (float += double), a
     * data-independent table shared by all rings */
    {
        std::vector<float> newY(max_points);
        newY[0] = 0.0f;
        for (uint32_t j = 1; j < max_points; j++)
            newY[j] = (float)((double)newY[j - 1] + 0.0100);
        if (hipMemcpy(c->d_newY, newY.data(), newY.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return fail(URF_ERR_HIP);
    }
    k.newY = c->d_newY;
    k.beams = c->d_beams;
    if ((rc = upload_params(c)) != URF_OK)
        return fail(rc);
    *out = c;
    return URF_OK;
}

static void free_lazy(urf_ctx* c)
{
    if (c->h_spec_failed)
        (void)hipHostFree(c->h_spec_failed);
    for (auto& sl : c->slots) {
        if (sl.exec)
            (void)hipGraphExecDestroy(sl.exec);
        if (sl.graph)
            (void)hipGraphDestroy(sl.graph);
        if (sl.ev_done)
            (void)hipEventDestroy(sl.ev_done);
        if (sl.h_in)
            (void)hipHostFree(sl.h_in);
        if (sl.h_labels)
            (void)hipHostFree(sl.h_labels);
        if (sl.h_info)
            (void)hipHostFree(sl.h_info);
        if (sl.d_raw)
            (void)hipFree(sl.d_raw);
        if (sl.d_labels)
            (void)hipFree(sl.d_labels);
    }
    for (hipStream_t st : c->row_stream)
        if (st)
            (void)hipStreamDestroy(st);
    if (c->ev_main)
        (void)hipEventDestroy(c->ev_main);
    for (void* p : { (void*)c->mk_d, (void*)c->mk_pos, (void*)c->mk_red, (void*)c->mk_out, (void*)c->ord_keys,
                     (void*)c->ord_pos, (void*)c->ord_cls, (void*)c->ord_lists })
        if (p)
            (void)hipFree(p);
    if (c->sx) {
        (void)hipFree(c->sx);
        (void)hipFree(c->sy);
        (void)hipFree(c->sz);
    }
}

extern "C" int urf_destroy(urf_ctx* c)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->own_stream)
        (void)hipStreamSynchronize(c->own_stream);
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
        if (c->ev_fork)
            (void)hipEventDestroy(c->ev_fork);
        if (c->ev_join)
            (void)hipEventDestroy(c->ev_join);
    }
    for (hipStream_t st : c->row_stream)
        if (st)
            (void)hipStreamSynchronize(st);
    for (auto& set : c->timing_events)
        for (hipEvent_t e : set)
            (void)hipEventDestroy(e);
    for (void* p : c->allocs)
        (void)hipFree(p);
    free_lazy(c);
    if (c->own_stream)
        (void)hipStreamDestroy(c->own_stream);
    delete c;
    return URF_OK;
}

extern "C" int urf_set_params(urf_ctx* c, const urf_params* p)
{
    if (!c || !p)
        return URF_ERR_INVALID_ARG;
    const int rc = urf_validate_params(p);
    if (rc != URF_OK)
        return rc;
    URF_HIP(c, hipSetDevice(c->device));
    for (hipStream_t st : c->row_stream)
        if (st)
            URF_HIP(c, hipStreamSynchronize(st));   /* a sweep in flight on another row keeps its parameters */
    /* the ring counts of earlier calls say nothing about sweeps classified with OTHER parameters (region of interest, interval) */
    if (std::memcmp(&c->params, p, sizeof(*p)) != 0) {
        URF_HIP(c, hipMemsetAsync(c->k.ring_hint, 0, URF_ASYNC_SLOTS * sizeof(uint32_t), c->stream));
        c->front_direct = c->front_off = false;   /* ... nor does what the fused front end made of them */
        c->h_spec_failed[2] = c->h_spec_failed[3] = 0;
    }
    c->params = *p;
    c->epoch++;
    return upload_params(c);
}

extern "C" int urf_get_params(const urf_ctx* c, urf_params* p)
{
    if (!c || !p)
        return URF_ERR_INVALID_ARG;
    *p = c->params;
    return URF_OK;
}

extern "C" int urf_set_stream(urf_ctx* c, void* hip_stream)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    c->epoch++;
    return URF_OK;
}

extern "C" int urf_synchronize(urf_ctx* c)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    URF_HIP(c, hipStreamSynchronize(c->stream));
    for (hipStream_t st : c->row_stream)   /* sweeps of the callback path in flight on other scratch rows */
        if (st)
            URF_HIP(c, hipStreamSynchronize(st));
    return URF_OK;
}

extern "C" int urf_enable_stage_capture(urf_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 2)
        return URF_ERR_INVALID_ARG;
    if (mode) {
        URF_HIP(c, hipSetDevice(c->device));
        const int rc = ensure_capture_arrays(c);
        if (rc != URF_OK)
            return rc;
    }
    c->capture = mode;
    c->epoch++;
    return URF_OK;
}

extern "C" int urf_callback_path_state(const urf_ctx* c, uint32_t* n_rerun, uint32_t* sequence)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    if (n_rerun)
        *n_rerun = c->n_rerun;
    if (sequence)
        *sequence = (c->speculate ? 1u : 0u) | (c->slot_lists ? 2u : 0u) | (c->slot_nan ? 4u : 0u) | (c->speculate && c->use_hint ? 8u : 0u) | (c->slot_ties ? 16u : 0u);
    return URF_OK;
}

extern "C" int urf_callback_path_preset(urf_ctx* c, uint32_t sequence_bits)
{
    if (!c || (sequence_bits & ~(2u | 4u | 16u)))
        return URF_ERR_INVALID_ARG;
    const bool lists = c->slot_lists || (sequence_bits & 2u), nan = c->slot_nan || (sequence_bits & 4u), ties = c->slot_ties || (sequence_bits & 16u);
    if (lists != c->slot_lists || nan != c->slot_nan || ties != c->slot_ties) {
        c->slot_lists = lists;
        c->slot_nan = nan;
        c->slot_ties = ties;
        c->epoch++;   /* the captured sequences are rebuilt */
    }
    return URF_OK;
}

extern "C" double urf_ring_threshold_cot(double angle_deg)
{
    return urf_cot_deg(angle_deg);
}

extern "C" int urf_set_front_mode(urf_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 2)
        return URF_ERR_INVALID_ARG;
    if (mode != c->front_mode && mode != 0) {   /* (a new start: what earlier calls made of the fused front end is forgotten) */
        c->front_direct = c->front_off = false;
        c->h_spec_failed[2] = c->h_spec_failed[3] = 0;
    }
    c->front_mode = mode;
    c->epoch++;   /* (the callback path's captured sequences depend on it) */
    if (mode != 0)
        c->want_ring_sorted = false;   /* (a caller that asks for ring-sorted results again pays for them again) */
    return URF_OK;
}

extern "C" int urf_front_scans(urf_ctx* c, uint32_t* n_fused)
{
    if (!c || !n_fused)
        return URF_ERR_INVALID_ARG;
    *n_fused = 0;
    if (!c->last_a.front || c->last_scans == 0)
        return URF_OK;
    URF_HIP(c, hipSetDevice(c->device));
    URF_HIP(c, hipStreamSynchronize(c->stream));   /* (a sweep of the callback path has been waited for: its row is at rest) */
    std::vector<uint32_t> ok(c->last_is_slot ? 1u : c->last_scans);
    URF_HIP(c, hipMemcpy(ok.data(), c->last_a.front_ok, ok.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t v : ok)
        *n_fused += v ? 1u : 0u;
    return URF_OK;
}

extern "C" int urf_enable_kernel_timing(urf_ctx* c, int on)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    c->timing = on != 0;
    c->epoch++;
    return URF_OK;
}

extern "C" const char* urf_kernel_name(int i)
{
    static const char* names[URF_NUM_KERNELS] = { "k_ring_table", "k_split", "k_index", "k_star_sort",
                                                  "k_star_walk", "k_ring", "k_beams", "k_label" };
    return (i >= 0 && i < URF_NUM_KERNELS) ? names[i] : "";
}

extern "C" int urf_kernel_timing(urf_ctx* c, double* ms_sum, uint32_t* n_calls)
{
    if (!c || !ms_sum || !n_calls)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    URF_HIP(c, hipStreamSynchronize(c->stream));
    for (hipStream_t st : c->row_stream)   /* (with timing on, a sweep of the callback path records its events on its row's stream) */
        if (st)
            URF_HIP(c, hipStreamSynchronize(st));
    for (size_t i = 0; i < c->timing_used; i++) {
        auto& set = c->timing_events[i];
        for (int k = 0; k < URF_NUM_KERNELS; k++) {
            float ms = 0.f;
            URF_HIP(c, hipEventElapsedTime(&ms, set[k], set[k + 1]));
            ms_sum[k] += (double)ms;
        }
        (*n_calls)++;
    }
    c->timing_used = 0;   /* the events stay allocated and are recorded again by the next calls */
    return URF_OK;
}

extern "C" const char* urf_last_error(const urf_ctx* c) { return c ? c->last_error.c_str() : ""; }

/* The context's scratch with every per-scan array advanced by `row` scans (allocation strides): slot i of the
 * callback path (URF_MAX_IN_FLIGHT slots) runs on row i % min(max_batch, URF_MAX_IN_FLIGHT) and on that row's own
 * stream, so that the sweeps in flight overlap.  Row 0 = the context's own arguments. */
static urf_kargs kargs_row(const urf_ctx* c, uint32_t row)
{
    urf_kargs k = c->k;
    if (row == 0)
        return k;
    const size_t P = (size_t)row * c->sstride, tiles = c->max_tiles, C = URF_MAX_CHANNELS, K = URF_MAX_SECTORS, r = row;
    k.rx += P; k.ry += P; k.rz += P; k.rec += P;
    k.sr += P; k.sz += P; k.sslot += P; k.ssrt16 += P; k.ssrt += P; k.wsg += P;
    k.big_r += P; k.big_z += P; k.big_i += P;
    if (k.valpha) {
        k.valpha += P; k.seckey += P; k.ringkey += P; k.rd2 += P; k.caz += P;
    }
    k.tile_roi += r * tiles; k.roi_bits += r * tiles * (URF_TILE / 64); k.troff += r * tiles * (C + 1); k.tsoff += r * tiles * (K + 1); k.tmaxs += r * tiles * C;
    k.rpre += r * C * (tiles + 1); k.rstart += r * C * tiles;
    k.angle += r * C; k.ring_thr += r * C * 4; k.ring_lut += r * URF_LUT_CELLS; k.ring_cnt += r * C; k.ring_off += r * (C + 1);
    k.sec_cnt += r * K; k.sec_run += r * K; k.sec_off += r * (K + 1); k.star_hit += r * K;
    k.star_first += r * K; k.star_list_mid += r * K; k.star_list_big += r * K; k.star_list_runs += r * K; k.tie_list += r * K; k.tie_post += r * K; k.star_count += 8 * r;
    k.table_upto += r; k.table_redo += r; k.redo_list += r; k.table_cause += r; k.ring_hint += r;
    k.nan_mask += r * 4; k.nan_list += 2 * r * C; k.vis += r * C;
    k.maxdist += r * C; k.quad += r * 4;
    k.curb_cnt += r * C; k.curb_az += r * C * URF_CURB_LIST;
    k.sufmin += r * C * URF_DEG_CELLS; k.premax += r * C * URF_DEG_CELLS;
    k.stop_f += r * URF_DEG_CELLS; k.stop_b += r * URF_DEG_CELLS;
    k.win += r * C * URF_DEG_CELLS;
    k.info += r;
    k.front_ok += r; k.front_pres += r * tiles * 64; k.front_maxs += r * tiles * 64; k.front_lane_ring += r * 64; k.front_ring_lane += r * C;
    if (k.tx) {
        k.tx += P; k.ty += P; k.tz += P; k.rows_v += r * 64; k.rows_ok += r;
    }
    k.front_cand += r * (size_t)k.front_cand_cap; k.front_all += r * (size_t)k.front_cand_cap; k.front_ncand += r; k.front_list += r; k.front_st += r * 72;
    return k;
}

/* Work on the context's stream that reads or writes scratch (any row), the SoA staging or the results of
 * the last call must come after the sweeps of the callback path that are still in flight on OTHER
 * streams (a batch call overwrites their rows; urf_read_stage / urf_ordered_indices /
 * urf_marker_points read them) ... */
static int order_after_slots(urf_ctx* c)
{
    for (auto& sl : c->slots)
        if (sl.pending && sl.ev_done)
            URF_HIP(c, hipStreamWaitEvent(c->stream, sl.ev_done, 0));   /* (a no-op for a slot that ran on `stream` itself) */
    c->main_seq++;
    return URF_OK;
}
/* ... and a sweep launched on a row's own stream must come after whatever the context's stream still
 * has to do with that row or the staging arrays. */
static int order_row_after_main(urf_ctx* c, uint32_t row, hipStream_t st)
{
    if (st == c->stream || c->row_seen[row] == c->main_seq)
        return URF_OK;
    if (!c->ev_main)
        URF_HIP(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    URF_HIP(c, hipEventRecord(c->ev_main, c->stream));
    URF_HIP(c, hipStreamWaitEvent(st, c->ev_main, 0));
    c->row_seen[row] = c->main_seq;
    return URF_OK;
}

/* Row-major organised sweeps (urf_front.hpp, k_ring_table's third rule): what the host makes of the device's flags.  h_spec_failed[4]: a scan
 * looked row-major (a sighting) -- once per context the firing-order copies are allocated and the launch sequences hold k_rows_probe and
 * k_transpose from then on; [6]: a scan TOOK the layout.  Everything a captured sequence depends on bumps the epoch.  Never called inside a
 * stream capture (it synchronises). */
static int rows_state_update(urf_ctx* c, hipStream_t st)
{
    if (c->front_mode != 0 && !c->front_rows && !c->rows_oom && c->h_spec_failed[4]) {
        /* the calls in flight finish first -- what they handed back (such sweeps, possibly all of them) says nothing about the calls to come */
        URF_HIP(c, hipStreamSynchronize(st));
        if (ensure_rows_arrays(c) == URF_OK) {
            c->front_rows = true;
            c->rows_probation = 16;
            c->front_direct = c->front_off = false;
            c->h_spec_failed[2] = c->h_spec_failed[3] = 0;
        } else {
            c->rows_oom = true;   /* (such sweeps keep to the general kernels) */
            c->last_error.clear();
        }
        c->epoch++;
    }
    if (c->front_rows && !c->rows_used && c->h_spec_failed[6]) {
        c->rows_used = true;
        c->epoch++;
    }
    return URF_OK;
}

/* ---- the pipeline ---------------------------------------------------------- */
/* on_stream == nullptr: a call of the public batch entry points (row 0 onwards, the context's stream,
 * published as "the last call"); otherwise one sweep of the callback path on its row and stream (the
 * caller publishes it when it is waited for). */
static int run_pipeline(urf_ctx* c, const float* d_x, const float* d_y, const float* d_z,
                        const uint32_t* d_offsets, uint32_t n_per_scan, uint32_t max_len, uint32_t n_scans,
                        uint8_t* d_labels, urf_scan_info* d_info, uint32_t row = 0, hipStream_t on_stream = nullptr,
                        urf_kargs* a_out = nullptr, urf_dev_params* dp_out = nullptr, const urf_dev_params* dp_in = nullptr,
                        int capture_in = -1, bool legacy_only = false)
{
    /* dp_in / capture_in: the parameters and capture mode a sweep was SUBMITTED with (urf_classify_pc2_wait runs a voided
     * sweep again: "a sweep in flight keeps its parameters", include/urf.h) */
    if (!d_x || !d_y || !d_z || !d_labels)
        return URF_ERR_INVALID_ARG;
    if (n_scans == 0)
        return URF_OK;
    if (row + n_scans > c->max_batch || max_len > c->max_points)
        return URF_ERR_CAPACITY;
    URF_HIP(c, hipSetDevice(c->device));
    hipStream_t st = on_stream ? on_stream : c->stream;
    if (!on_stream) {
        const int orc = order_after_slots(c);
        if (orc != URF_OK)
            return orc;
    }
    urf_kargs a = kargs_row(c, row);
    a.x = d_x;
    a.y = d_y;
    a.z = d_z;
    a.offsets = nullptr;
    if (d_offsets) {
        /* the context keeps its own copy: the entry points that look at this call's results later
         * (urf_read_stage, urf_ordered_indices, urf_marker_points) must not depend on the caller
         * keeping d_offsets alive.  Scratch memory is indexed by scan, never by these offsets. */
        if (d_offsets != c->offsets_copy)   /* (last_row_intact runs the last call again with the copy itself) */
            URF_HIP(c, hipMemcpyAsync(c->offsets_copy, d_offsets, ((size_t)n_scans + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        a.offsets = c->offsets_copy;
    }
    a.n_per_scan = n_per_scan;
    a.n_scans = n_scans;
    a.max_len = max_len;
    a.tiles = (max_len + URF_TILE - 1) / URF_TILE;
    if (a.tiles == 0)
        a.tiles = 1;
    a.sstride = c->sstride;
    if (c->speculate && c->h_spec_failed[0]) {   /* an earlier call had to repair a speculative ring table */
        c->speculate = false;
        c->epoch++;
    }
    if (c->use_hint && c->h_spec_failed[1]) {    /* ... or one that stopped at the previous call's ring count */
        c->use_hint = false;
        c->epoch++;
    }
    a.table_lookahead = c->speculate ? URF_TABLE_LOOKAHEAD : 0u;
    a.table_hint = (c->speculate && c->use_hint) ? 1u : 0u;
    /* A sweep of the callback path is waited for by the host before anybody sees its result: the kernels that
     * normally find nothing to do -- the two repair kernels behind the speculative ring table, the two for the work
     * lists of oversized star sectors, 20 of a sweep's 200 microseconds -- are left out, k_index voids a sweep that
     * needed them, and urf_classify_pc2_wait() runs it again with them. */
    a.optimistic = on_stream ? ((c->speculate ? URF_OPT_NO_REPAIR : 0u) | (c->slot_lists ? 0u : URF_OPT_NO_LISTS) | (c->slot_nan ? 0u : URF_OPT_NO_NAN) |
                               (c->slot_ties ? 0u : URF_OPT_NO_TIES)) : 0u;
    a.capture = (uint32_t)(capture_in >= 0 ? capture_in : c->capture);
    a.labels = d_labels;
    if (a.capture != 1) {
        a.rd2 = nullptr;
        a.caz = nullptr;
    }
    const urf_dev_params dp = dp_in ? *dp_in : c->dp;
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    const bool star = dp.p.star_shaped_method != 0;
    const dim3 g_tiles(a.tiles, n_scans), g_scan(n_scans);
    /* The fused front end (urf_front.hpp) for batches of sweeps in firing order: k_front tries every scan, the legacy kernels
     * skip the scans it kept.  64 lasers = 64 lanes, the detectors' window of curbPoints == 5 in registers, no stage capture
     * (its values are the legacy kernels'), not for the single sweeps of the callback path (sixteen waves on the whole device). */
    if (!on_stream) {   /* (a sweep of the callback path: urf_classify_pc2_async has done it, outside its stream capture) */
        const int rrc = rows_state_update(c, st);
        if (rrc != URF_OK)
            return rrc;
    }
    if (c->k.tx && !a.tx) {   /* (allocated after this call's arguments were copied: row 0's) */
        a.tx = c->k.tx;
        a.ty = c->k.ty;
        a.tz = c->k.tz;
        a.rows_v = c->k.rows_v;
        a.rows_ok = c->k.rows_ok;
    }
    if (c->h_spec_failed[2])
        c->front_direct = true;
    if (c->h_spec_failed[3] && c->front_mode != 2)
        c->front_off = true;
    /* (a context that has sighted row-major sweeps takes the fused kernels at any batch size: the general kernels need 0.64 ms for four
     * such sweeps, the fused ones 0.26 -- tools/r6_min_scans.py --rows; sweeps in firing order gain from 192 per call on) */
    const bool front_shape = c->front_mode != 0 && !c->front_off && !legacy_only && !c->want_ring_sorted && a.capture == 0 &&
                             C == URF_FRONT_LANES && dp.p.curbPoints == 5 && a.tiles <= URF_FRONT_MAX_TILES;
    const bool small_ok = c->front_rows && (c->rows_used || c->rows_probation > 0);
    /* (a single sweep of the callback path: only in a context whose sweeps come row-major -- a sweep in firing order is faster through the
     * general kernels, tools/r6_single_sweep.py) */
    a.front = (front_shape && (on_stream ? small_ok : (c->front_mode == 2 || small_ok || n_scans >= URF_FRONT_MIN_SCANS))) ? 1u : 0u;
    if (a.front && !on_stream && c->front_mode != 2 && n_scans < URF_FRONT_MIN_SCANS && !c->rows_used && c->rows_probation)
        c->rows_probation--;   /* (a sighting that no scan confirms -- a sweep in firing order whose region of interest begins with a single laser -- lapses;
                                *  the callback path counts its submissions: urf_classify_pc2_async) */
    a.front_sight = (front_shape && !a.front && !c->front_rows && !c->rows_oom) ? 1u : 0u;
    a.front_tpb = c->front_tpb ? c->front_tpb : (n_scans >= URF_FRONT_TPB_SCANS ? URF_FRONT_TPB_LARGE : (n_scans >= 16u ? URF_FRONT_TPB_SMALL : 1u));
    a.front_lists = (a.front && !c->front_direct && !on_stream) ? 1u : 0u;   /* (the callback path's sequence holds the general kernels as grids anyway) */
    a.front_rows = (a.front && c->front_rows) ? 1u : 0u;   /* (the rows' rule does not depend on the two other speculations: the repair kernels below come with it) */

    std::vector<hipEvent_t>* ev = nullptr;
    if (c->timing) {
        if (c->timing_used == c->timing_events.size()) {
            c->timing_events.emplace_back(URF_NUM_KERNELS + 1);
            for (hipEvent_t& e : c->timing_events.back())
                URF_HIP(c, hipEventCreate(&e));
        }
        ev = &c->timing_events[c->timing_used++];
    }
    int stage = 0;
    auto mark = [&]() {
        if (ev)
            (void)hipEventRecord((*ev)[stage], st);
        stage++;
    };
    mark();
    if (a.front_rows)
        hipLaunchKernelGGL(k_rows_probe, g_scan, dim3(256), 0, st, a, dp);
    hipLaunchKernelGGL(k_ring_table, g_scan, dim3(URF_TABLE_THREADS), 0, st, a, dp);
    mark();
    if (a.front_rows)
        hipLaunchKernelGGL(k_transpose, g_tiles, dim3(256), 0, st, a);
    if (a.front)
        hipLaunchKernelGGL(k_front, dim3((a.tiles + a.front_tpb - 1) / a.front_tpb, n_scans), dim3(64), 0, st, a, dp);
    if (a.front_lists) {   /* what k_front handed back (normally nothing) */
        hipLaunchKernelGGL(k_table_repair, g_scan, dim3(URF_TABLE_THREADS), 0, st, a, dp, 1u);
        hipLaunchKernelGGL(k_split_list, dim3(c->n_cus * 2), dim3(URF_TILE_THREADS), urf_split_lds_bytes(C, K, star), st, a, dp);
    } else {
        hipLaunchKernelGGL(k_split, g_tiles, dim3(URF_TILE_THREADS), urf_split_lds_bytes(C, K, star), st, a, dp);
    }
    if ((a.table_lookahead || a.front_rows) && !(a.optimistic & URF_OPT_NO_REPAIR)) {   /* normally both find nothing to do */
        hipLaunchKernelGGL(k_table_repair, g_scan, dim3(URF_TABLE_THREADS), 0, st, a, dp, 0u);
        hipLaunchKernelGGL(k_split_repair, dim3(c->n_cus), dim3(URF_TILE_THREADS), urf_split_lds_bytes(C, K, star), st, a, dp);
    }
    mark();
    /* k_front_finish's first part (positions, the detectors' candidates and their marks) depends on nothing behind k_front: it goes to a
     * stream of its own and runs NEXT TO k_index, the star-shaped search's sort and walk -- it waits for scattered loads (0.18 ms on its
     * own), they are bound by vector issue.  With the per-kernel event brackets on (urf_enable_kernel_timing) everything stays on one
     * stream, so that the brackets add up to the step. */
    bool side = false, part1 = false;
    const size_t finish_lds = (size_t)a.tiles * 384 + 2 * URF_FINISH_CHUNK * sizeof(urf_u2);
    if (a.front && !ev && !on_stream) {
        if (!c->side_stream) {
            if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
                c->side_stream = nullptr;
        }
        if (c->side_stream && hipEventRecord(c->ev_fork, st) == hipSuccess && hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) == hipSuccess) {
            hipLaunchKernelGGL(k_front_finish, g_scan, dim3(URF_FINISH_THREADS), finish_lds, c->side_stream, a, dp, 1u);
            part1 = true;
            side = hipEventRecord(c->ev_join, c->side_stream) == hipSuccess;
            if (!side)   /* (cannot be joined by an event: wait for it here) */
                (void)hipStreamSynchronize(c->side_stream);
        }
    }
    hipLaunchKernelGGL(k_index, g_scan, dim3(256), 0, st, a, dp);
    mark();
    /* k_star_ties: persistent one-wave workgroups (32 KB of LDS: four per CU) over a list that holds one sector in a hundred of
     * a sensor's sweep and nothing of a benchmark cloud */
    const unsigned tie_grid = K * n_scans < c->n_cus * 4u ? K * n_scans : c->n_cus * 4u;
    const unsigned tie_grid_small = K * n_scans < c->n_cus * 16u ? K * n_scans : c->n_cus * 16u;   /* (8 KB of LDS per wave) */
    if (star) {
        const dim3 g_sec(K, n_scans);
        hipLaunchKernelGGL(k_star_sort_small, g_sec, dim3(URF_STAR_THREADS), 0, st, a, dp);
        /* persistent workgroups over the (normally empty) work lists of oversized sectors */
        if (!(a.optimistic & URF_OPT_NO_LISTS)) {
            hipLaunchKernelGGL(k_star_sort_mid, dim3(c->n_cus * (URF_MID_WAVES * 256 / URF_STAR_MID_THREADS)), dim3(URF_STAR_MID_THREADS), 0, st,
                               a, dp);   /* as many workgroups as are resident */
            hipLaunchKernelGGL(k_star_sort_big, dim3(c->n_cus * 2), dim3(256), 0, st, a, dp);
            hipLaunchKernelGGL(k_star_sort_runs, dim3(c->n_cus * 20), dim3(URF_STAR_THREADS), 0, st, a, dp);   /* (five waves per SIMD) */
        }
        /* sectors whose sorted prefix holds equal planar ranges of different heights (URF_TIE_FLAG): the order libstdc++'s std::sort
         * leaves them in.  Benchmark clouds hold none (the kernel returns at once); a real sensor's sweep holds equal ranges in
         * every sector, but nearly all of them between twins (one height), which only the second pass below cares about. */
        if (!(a.optimistic & URF_OPT_NO_TIES)) {
            hipLaunchKernelGGL((k_star_ties<false, URF_TIE_SMALL>), dim3(tie_grid_small), dim3(64), 0, st, a, dp);
            hipLaunchKernelGGL((k_star_ties<false, URF_TIE_CAP>), dim3(tie_grid), dim3(64), 0, st, a, dp);
        }
    }
    mark();   /* "k_star_sort" = the three sort kernels (mid / big run over normally empty work lists) */
    if (star) {
        if (n_scans <= URF_WALK_FEW_SCANS)   /* an empty device: three waves per 64 sectors, one chunk apart */
            hipLaunchKernelGGL(k_star_walk_few, dim3((K + 63) / 64, n_scans), dim3(URF_WALK_FEW_THREADS), 0, st, a, dp);
        else
            hipLaunchKernelGGL(k_star_walk, dim3((K + 63) / 64, n_scans), dim3(64), 0, st, a, dp);
        /* second pass of k_star_ties: the sectors in which the walk stopped at a point with a twin behind it (URF_TIE_POST) */
        if (!(a.optimistic & URF_OPT_NO_TIES)) {
            hipLaunchKernelGGL((k_star_ties<true, URF_TIE_SMALL>), dim3(tie_grid_small), dim3(64), 0, st, a, dp);
            hipLaunchKernelGGL((k_star_ties<true, URF_TIE_CAP>), dim3(tie_grid), dim3(64), 0, st, a, dp);
        }
    }
    mark();
    const dim3 g_ring(C, n_scans);
    if (a.front_lists)
        hipLaunchKernelGGL(k_ring_list, dim3(c->n_cus * 8), dim3(URF_RING_THREADS), (2 * (size_t)a.tiles + 1) * sizeof(unsigned), st, a, dp);
    else if (dp.p.curbPoints == 5)   /* the reference's default: four points per thread, z only */
        hipLaunchKernelGGL(k_ring, g_ring, dim3(URF_RING_THREADS), (2 * (size_t)a.tiles + 1) * sizeof(unsigned), st, a, dp);
    else
        hipLaunchKernelGGL(k_ring_general, g_ring, dim3(URF_RING_THREADS), (2 * (size_t)a.tiles + 1) * sizeof(unsigned), st, a, dp);
    if (a.front) {
        if (side && hipStreamWaitEvent(st, c->ev_join, 0) != hipSuccess)
            (void)hipStreamSynchronize(c->side_stream);
        hipLaunchKernelGGL(k_front_finish, g_scan, dim3(URF_FINISH_THREADS), finish_lds, st, a, dp, part1 ? 2u : 0u);   /* (2: the star-shaped hits, the hand-over to k_beams) */
    }
    /* the rings that hold a point with a NaN azimuth (k_split listed them: normally none, the kernel returns at once) */
    if (!(a.optimistic & URF_OPT_NO_NAN))
        hipLaunchKernelGGL(k_nan_rings, dim3(32), dim3(256), URF_NAN_LDS * sizeof(unsigned long long), st, a, dp);
    mark();
    hipLaunchKernelGGL(k_beams, g_scan, dim3(URF_BEAM_THREADS), (size_t)C * (24 * sizeof(unsigned) + URF_CURB_LIST * sizeof(float)), st, a, dp);
    mark();
    if (a.front_lists)
        hipLaunchKernelGGL(k_label_list, dim3(c->n_cus * 4), dim3(URF_LABEL_TILE_THREADS), 0, st, a, dp);
    else
        hipLaunchKernelGGL(k_label, g_tiles, dim3(URF_LABEL_TILE_THREADS), 0, st, a, dp);
    if (a.front)
        hipLaunchKernelGGL(k_label_front, g_tiles, dim3(URF_LABEL_TILE_THREADS), 0, st, a, dp);
    mark();
    URF_HIP(c, hipGetLastError());
    if (d_info)
        URF_HIP(c, hipMemcpyAsync(d_info, a.info, (size_t)n_scans * sizeof(urf_scan_info), hipMemcpyDeviceToDevice, st));
    if (on_stream) {
        *a_out = a;
        *dp_out = dp;
    } else {
        c->last_scans = n_scans;
        c->last_a = a;
        c->last_dp = dp;
        c->last_is_slot = false;
    }
    return URF_OK;
}

extern "C" int urf_classify_batch_soa(urf_ctx* c, const float* d_x, const float* d_y, const float* d_z,
                                      uint32_t n_per_scan, uint32_t n_scans, uint8_t* d_labels, urf_scan_info* d_info)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    return run_pipeline(c, d_x, d_y, d_z, nullptr, n_per_scan, n_per_scan, n_scans, d_labels, d_info);
}

extern "C" int urf_classify_batch_soa_ragged(urf_ctx* c, const float* d_x, const float* d_y, const float* d_z,
                                             const uint32_t* d_offsets, uint32_t max_len, uint32_t n_scans,
                                             uint8_t* d_labels, urf_scan_info* d_info)
{
    if (!c || !d_offsets)
        return URF_ERR_INVALID_ARG;
    return run_pipeline(c, d_x, d_y, d_z, d_offsets, 0, max_len, n_scans, d_labels, d_info);
}

static int ensure_soa_staging(urf_ctx* c)
{
    if (c->sx)
        return URF_OK;
    const size_t n = (size_t)c->max_points * c->max_batch;
    void *px = nullptr, *py = nullptr, *pz = nullptr;
    URF_HIP(c, hipMalloc(&px, n * sizeof(float)));
    URF_HIP(c, hipMalloc(&py, n * sizeof(float)));
    URF_HIP(c, hipMalloc(&pz, n * sizeof(float)));
    c->sx = (float*)px;
    c->sy = (float*)py;
    c->sz = (float*)pz;
    return URF_OK;
}

/* field offsets of a PointCloud2 record: every FLOAT32 field must lie inside the record
 * (evaluated in 64 bits: the offsets come from an untrusted wire message) */
static bool pc2_layout_ok(uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z)
{
    const uint64_t ps = point_step;
    return ps >= 4 && (uint64_t)off_x + 4 <= ps && (uint64_t)off_y + 4 <= ps && (uint64_t)off_z + 4 <= ps;
}

extern "C" int urf_classify_batch_pc2(urf_ctx* c, const uint8_t* d_data, uint32_t n_per_scan, uint32_t n_scans,
                                      uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                                      uint8_t* d_labels, urf_scan_info* d_info)
{
    if (!c || !d_data || !pc2_layout_ok(point_step, off_x, off_y, off_z))
        return URF_ERR_INVALID_ARG;
    if (n_scans > c->max_batch || n_per_scan > c->max_points)
        return URF_ERR_CAPACITY;
    if (n_scans == 0)
        return URF_OK;
    URF_HIP(c, hipSetDevice(c->device));
    int rc = ensure_soa_staging(c);
    if (rc != URF_OK)
        return rc;
    rc = order_after_slots(c);   /* the staging arrays are shared with the callback path */
    if (rc != URF_OK)
        return rc;
    const unsigned long long total = (unsigned long long)n_per_scan * n_scans;
    hipLaunchKernelGGL(k_pc2_to_soa, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_data, total,
                       point_step, off_x, off_y, off_z, c->sx, c->sy, c->sz);
    return run_pipeline(c, c->sx, c->sy, c->sz, nullptr, n_per_scan, n_per_scan, n_scans, d_labels, d_info);
}

/* ---- the callback path: one sweep, host buffers ------------------------------- */
/* Slot i of the callback path works on scratch row i % rows, rows = min(max_batch, URF_ASYNC_SLOTS), and on
 * that row's compute stream (row 0: the context's stream): with a context created for several scans the
 * kernels of as many sweeps overlap on the device (a single sweep's kernels are a few dozen workgroups
 * each).  With max_batch == 1 all slots share row 0 and the context's stream: only the copies overlap. */
static uint32_t slot_row(const urf_ctx* c, const urf_ctx::slot_t& sl)
{
    const uint32_t rows = c->max_batch < URF_ASYNC_SLOTS ? c->max_batch : URF_ASYNC_SLOTS;
    return (uint32_t)(&sl - c->slots) % rows;
}
static hipStream_t slot_stream(urf_ctx* c, const urf_ctx::slot_t& sl)
{
    const uint32_t row = slot_row(c, sl);
    return row ? c->row_stream[row] : c->stream;
}

static int slot_prepare(urf_ctx* c, urf_ctx::slot_t& sl, size_t bytes)
{
    if (!c->streams_made) {
        for (uint32_t r = 1; r < URF_ASYNC_SLOTS && r < c->max_batch; r++)
            URF_HIP(c, hipStreamCreateWithFlags(&c->row_stream[r], hipStreamNonBlocking));
        c->streams_made = true;
    }
    if (!sl.ev_done) {
        URF_HIP(c, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        void *hl = nullptr, *hi = nullptr, *dl = nullptr;
        URF_HIP(c, hipHostMalloc(&hl, c->max_points, hipHostMallocDefault));
        URF_HIP(c, hipHostMalloc(&hi, sizeof(urf_scan_info), hipHostMallocDefault));
        URF_HIP(c, hipMalloc(&dl, c->max_points));
        sl.h_labels = (uint8_t*)hl;
        sl.h_info = (urf_scan_info*)hi;
        sl.d_labels = (uint8_t*)dl;
    }
    if (bytes > sl.h_in_cap) {   /* grows to the largest message seen (a new buffer invalidates the captured sequence) */
        URF_HIP(c, hipStreamSynchronize(slot_stream(c, sl)));   /* the slot's last sweep may still read d_raw */
        if (sl.h_in)
            (void)hipHostFree(sl.h_in);
        if (sl.d_raw)
            (void)hipFree(sl.d_raw);
        sl.h_in = nullptr;
        sl.d_raw = nullptr;
        sl.h_in_cap = sl.d_raw_cap = 0;
        sl.key[0] = 0;
        void *h = nullptr, *d = nullptr;
        URF_HIP(c, hipHostMalloc(&h, bytes, hipHostMallocDefault));
        sl.h_in = (uint8_t*)h;
        sl.h_in_cap = bytes;
        URF_HIP(c, hipMalloc(&d, bytes));
        sl.d_raw = (uint8_t*)d;
        sl.d_raw_cap = bytes;
    }
    return URF_OK;
}

#ifdef URF_ENABLE_TEST_HOOKS
static inline double ht_now()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define HT_START double ht0 = c->ht_on ? ht_now() : 0.0
#define HT(k, t0) do { if (c->ht_on) { const double ht_t = ht_now(); c->ht[k] += ht_t - (t0); (t0) = ht_t; } } while (0)
#else
#define HT_START
#define HT(k, t0)
#endif

/* A message that has to be staged anyway is staged as three planes x[n4] y[n4] z[n4], n4 = n rounded up to 4: 12 of
 * its (typically) 32 bytes per point cross PCIe (84 -> 35 us for a 64 x 2048 sweep), and the device needs no
 * records -> SoA kernel.  Records whose x, y, z lie side by side with a fourth word behind them inside the record
 * (pcl::PointXYZI and every PointCloud2 layout of the lidar drivers) go four at a time through a 4 x 4 transpose and
 * leave with non-temporal stores (the planes are 16-byte aligned; on the test box's EPYC 9575F: memcpy of the 4 MiB
 * message 108 us, the transpose with ordinary stores 105, with streaming stores 67). */
static void pc2_to_planes(const uint8_t* data, uint32_t i0, uint32_t n, uint32_t step, uint32_t ox, uint32_t oy, uint32_t oz, float* X,
                          float* Y, float* Z)   /* points [i0, n); i0 a multiple of 4 */
{
    uint32_t i = i0;
#if defined(__SSE2__)
    if (oy == ox + 4 && oz == ox + 8 && (uint64_t)ox + 16 <= step) {
        const uint8_t* p = data + (size_t)i0 * step + ox;
        if ((((uintptr_t)(X + i0) | (uintptr_t)(Y + i0) | (uintptr_t)(Z + i0)) & 15u) == 0) {   /* (the pinned planes: always) */
            for (; i + 4 <= n; i += 4, p += 4 * (size_t)step) {
                __m128 r0 = _mm_loadu_ps((const float*)p), r1 = _mm_loadu_ps((const float*)(p + step));
                __m128 r2 = _mm_loadu_ps((const float*)(p + 2 * (size_t)step)), r3 = _mm_loadu_ps((const float*)(p + 3 * (size_t)step));
                _MM_TRANSPOSE4_PS(r0, r1, r2, r3);
                _mm_stream_ps(X + i, r0);
                _mm_stream_ps(Y + i, r1);
                _mm_stream_ps(Z + i, r2);
            }
            _mm_sfence();   /* before the DMA engine is told to read them */
        } else {
            for (; i + 4 <= n; i += 4, p += 4 * (size_t)step) {
                __m128 r0 = _mm_loadu_ps((const float*)p), r1 = _mm_loadu_ps((const float*)(p + step));
                __m128 r2 = _mm_loadu_ps((const float*)(p + 2 * (size_t)step)), r3 = _mm_loadu_ps((const float*)(p + 3 * (size_t)step));
                _MM_TRANSPOSE4_PS(r0, r1, r2, r3);
                _mm_storeu_ps(X + i, r0);
                _mm_storeu_ps(Y + i, r1);
                _mm_storeu_ps(Z + i, r2);
            }
        }
    }
#endif
    for (; i < n; i++) {
        const uint8_t* p = data + (size_t)i * step;
        std::memcpy(X + i, p + ox, 4);
        std::memcpy(Y + i, p + oy, 4);
        std::memcpy(Z + i, p + oz, 4);
    }
}

/* the gather by itself (host only, no context): what urf_classify_pc2_async does with a message it stages */
extern "C" int urf_pc2_to_planes(const uint8_t* data, uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                                 uint32_t off_z, float* x, float* y, float* z)
{
    if (!data || !x || !y || !z || !pc2_layout_ok(point_step, off_x, off_y, off_z))
        return URF_ERR_INVALID_ARG;
    pc2_to_planes(data, 0, n_points, point_step, off_x, off_y, off_z, x, y, z);
    return URF_OK;
}

/* what one sweep of the callback path launches on the compute stream: records -> SoA (unless the message was
 * staged as planes), the pipeline, results to the pinned host buffers */
static int slot_launch(urf_ctx* c, urf_ctx::slot_t& sl, uint32_t n_points, uint32_t point_step, uint32_t off_x,
                       uint32_t off_y, uint32_t off_z, const urf_dev_params* dp_in = nullptr, int capture_in = -1)
{
    const uint32_t row = slot_row(c, sl);
    hipStream_t st = slot_stream(c, sl);
    float *sx, *sy, *sz;
    if (sl.planes) {
        const size_t n4 = ((size_t)n_points + 3) & ~(size_t)3;
        sx = (float*)sl.d_raw;
        sy = sx + n4;
        sz = sy + n4;
    } else {
        sx = c->sx + (size_t)row * c->max_points;   /* (ensure_soa_staging: the caller) */
        sy = c->sy + (size_t)row * c->max_points;
        sz = c->sz + (size_t)row * c->max_points;
        hipLaunchKernelGGL(k_pc2_to_soa, dim3((n_points + 255) / 256), dim3(256), 0, st, sl.d_raw, (unsigned long long)n_points,
                           point_step, off_x, off_y, off_z, sx, sy, sz);
    }
    urf_kargs a_run;
    urf_dev_params dp_run;   /* (dp_in may point at sl.cap_dp) */
    const int rc = run_pipeline(c, sx, sy, sz, nullptr, n_points, n_points, 1, sl.d_labels, nullptr, row, st, &a_run, &dp_run, dp_in, capture_in);
    if (rc != URF_OK)
        return rc;
    sl.cap_a = a_run;
    sl.cap_dp = dp_run;
    URF_HIP(c, hipMemcpyAsync(sl.h_labels, sl.d_labels, n_points, hipMemcpyDeviceToHost, st));
    URF_HIP(c, hipMemcpyAsync(sl.h_info, kargs_row(c, row).info, sizeof(urf_scan_info), hipMemcpyDeviceToHost, st));
    return URF_OK;
}

extern "C" int urf_pinned_input(urf_ctx* c, size_t bytes, uint8_t** ptr)
{
    if (!c || !ptr || bytes == 0)
        return URF_ERR_INVALID_ARG;
    urf_ctx::slot_t& sl = c->slots[c->next_ticket % URF_ASYNC_SLOTS];   /* the slot the next submission uses */
    if (sl.pending)
        return URF_ERR_BUSY;
    URF_HIP(c, hipSetDevice(c->device));
    const int rc = slot_prepare(c, sl, bytes);
    if (rc != URF_OK)
        return rc;
    *ptr = sl.h_in;
    return URF_OK;
}

extern "C" int urf_classify_pc2_async(urf_ctx* c, const uint8_t* data, uint32_t n_points, uint32_t point_step,
                                      uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t* ticket)
{
    if (!c || !data || !ticket || n_points == 0 || !pc2_layout_ok(point_step, off_x, off_y, off_z))
        return URF_ERR_INVALID_ARG;   /* before any byte of the message is copied */
    if (n_points > c->max_points)
        return URF_ERR_CAPACITY;
    urf_ctx::slot_t& sl = c->slots[c->next_ticket % URF_ASYNC_SLOTS];
    if (sl.pending)
        return URF_ERR_BUSY;          /* every slot in flight: urf_classify_pc2_wait() the oldest one first */
    URF_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n_points * point_step;
    HT_START;
    /* a message inside the slot's pinned buffer must be the buffer urf_pinned_input() handed out, and
     * fit it: a larger one would make slot_prepare() free the very memory it is about to read */
    if (sl.h_in && data >= sl.h_in && data < sl.h_in + sl.h_in_cap && (data != sl.h_in || bytes > sl.h_in_cap))
        return URF_ERR_INVALID_ARG;
    const bool planes = data != sl.h_in;   /* (a producer that filled the pinned buffer itself wrote records) */
    const size_t n4 = ((size_t)n_points + 3) & ~(size_t)3;
    const size_t plane_bytes = 3 * sizeof(float) * n4;
    int rc = slot_prepare(c, sl, planes && plane_bytes > bytes ? plane_bytes : bytes);
    if (rc == URF_OK && !planes)
        rc = ensure_soa_staging(c);   /* where the device gathers the records' x / y / z */
    if (rc != URF_OK)
        return rc;
    hipStream_t st = slot_stream(c, sl);
    rc = order_row_after_main(c, slot_row(c, sl), st);
    if (rc != URF_OK)
        return rc;
    /* The message goes to the device on the slot's own stream, in front of the sweep's kernels.  (r2 / r3 used a copy
     * stream of its own and an event per sweep for the slot's stream to wait on: 7 870 -> 8 670 sweeps/s with a pinned
     * producer, 6 460 -> 7 540 staged without them.  The four streams still map to three hardware queues -- kernel
     * trace, profiles/r3_callback_trace_after.txt --, but more queues made things worse, see profiles/README.md.)
     * Sweeps of other slots run beside the copy as before. */
    if (planes) {
        /* gathered into the pinned planes (pc2_to_planes) in two halves, so that the first one is on its way while the
         * second one is gathered (one 2-D copy per half: its columns of the three planes; 35 us for the whole, 22 per
         * half).  (urf_pinned_input() lets a producer write its records into the pinned buffer directly: no staging.) */
        float* X = (float*)sl.h_in;
        const uint32_t mid = n_points >= 32768 ? (uint32_t)((n4 / 2) & ~(size_t)3) : 0u;
        const uint32_t cut[3] = { 0u, mid, n_points };
        for (int h = mid ? 0 : 1; h < 2; h++) {
            pc2_to_planes(data, cut[h], cut[h + 1], point_step, off_x, off_y, off_z, X, X + n4, X + 2 * n4);
            const size_t w = (h == 1 ? n4 - cut[1] : cut[1]) * sizeof(float), o = cut[h] * sizeof(float);
            URF_HIP(c, hipMemcpy2DAsync(sl.d_raw + o, n4 * sizeof(float), sl.h_in + o, n4 * sizeof(float), w, 3, hipMemcpyHostToDevice, st));
        }
    } else {
        URF_HIP(c, hipMemcpyAsync(sl.d_raw, sl.h_in, bytes, hipMemcpyHostToDevice, st));
    }
    if (planes != sl.planes)
        sl.key[0] = 0;   /* the captured sequence reads the other format */
    sl.planes = planes;
    HT(0, ht0);   /* staging + H2D enqueue */
    /* a sweep that defeated the speculative ring table (k_table_repair raised the host-visible flag) ends
     * the speculation for replayed sequences as well: the captured ones are rebuilt without it */
    if (c->speculate && c->h_spec_failed[0]) {
        c->speculate = false;
        c->epoch++;
    }
    if (c->use_hint && c->h_spec_failed[1]) {
        c->use_hint = false;
        c->epoch++;
    }
    /* row-major organised sweeps: sighted by the general kernels, then the fused ones in this path's sequence as well (rows_state_update); a
     * sighting that no sweep confirms lapses after sixteen submissions */
    rc = rows_state_update(c, st);
    if (rc != URF_OK)
        return rc;
    if (c->front_rows && !c->rows_used && c->rows_probation && --c->rows_probation == 0)
        c->epoch++;
    /* the launch sequence of a sweep of this shape is captured once and replayed (one graph launch
     * instead of a dozen kernel launches per callback); anything it depends on bumps the epoch */
    const uint64_t key[3] = { c->epoch, ((uint64_t)n_points << 32) | point_step,
                              ((uint64_t)off_x << 42) ^ ((uint64_t)off_y << 21) ^ off_z };
    const bool use_graph = !c->timing && !(c->debug_flags & 8u);
    HT(1, ht0);   /* event record, ordering, stream wait */
    if (use_graph && (sl.key[0] != key[0] || sl.key[1] != key[1] || sl.key[2] != key[2] || !sl.exec)) {
        if (sl.exec)
            (void)hipGraphExecDestroy(sl.exec);
        if (sl.graph)
            (void)hipGraphDestroy(sl.graph);
        sl.exec = nullptr;
        sl.graph = nullptr;
        URF_HIP(c, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        rc = slot_launch(c, sl, n_points, point_step, off_x, off_y, off_z);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(st, &g);
        if (rc != URF_OK || e != hipSuccess) {
            if (g)
                (void)hipGraphDestroy(g);
            if (rc == URF_OK)
                URF_HIP(c, e);
            return rc;
        }
        sl.graph = g;
        URF_HIP(c, hipGraphInstantiate(&sl.exec, sl.graph, nullptr, nullptr, 0));
        sl.key[0] = key[0];
        sl.key[1] = key[1];
        sl.key[2] = key[2];
    }
    if (use_graph) {
        URF_HIP(c, hipGraphLaunch(sl.exec, st));
    } else {
        rc = slot_launch(c, sl, n_points, point_step, off_x, off_y, off_z);
        if (rc != URF_OK)
            return rc;
    }
    HT(2, ht0);   /* graph launch (or the launches themselves) */
    URF_HIP(c, hipEventRecord(sl.ev_done, st));
    HT(3, ht0);
    sl.pending = true;
    sl.used = true;
    sl.n_points = n_points;
    sl.point_step = point_step;
    sl.off_x = off_x;
    sl.off_y = off_y;
    sl.off_z = off_z;
    sl.ticket = c->next_ticket;
    sl.gen = ++c->row_gen[slot_row(c, sl)];
    *ticket = c->next_ticket++;
    return URF_OK;
}

extern "C" int urf_classify_pc2_wait(urf_ctx* c, uint32_t ticket, uint8_t* labels_out, urf_scan_info* info)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    urf_ctx::slot_t& sl = c->slots[ticket % URF_ASYNC_SLOTS];
    if (!sl.pending || sl.ticket != ticket)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    HT_START;
    URF_HIP(c, hipEventSynchronize(sl.ev_done));
    HT(4, ht0);
    /* the short launch sequence left out something this sweep needed (run_pipeline): once more, with it --
     * the message is still in the slot's device buffer -- and from now on for every sweep */
    auto redo = [](int st) {
        return st == URF_STATUS_REDO_TABLE || st == URF_STATUS_REDO_LISTS || st == URF_STATUS_REDO_NAN || st == URF_STATUS_REDO_HINT ||
               st == URF_STATUS_REDO_TIES;
    };
    for (int tries = 0; tries < 6 && redo(sl.h_info->status); tries++) {
        if (sl.h_info->status == URF_STATUS_REDO_TABLE)
            c->speculate = false;
        else if (sl.h_info->status == URF_STATUS_REDO_HINT)
            c->use_hint = false;
        else if (sl.h_info->status == URF_STATUS_REDO_LISTS)
            c->slot_lists = true;
        else if (sl.h_info->status == URF_STATUS_REDO_TIES)
            c->slot_ties = true;
        else
            c->slot_nan = true;
        c->epoch++;   /* the captured sequences are rebuilt */
        c->n_rerun++;
        /* with the parameters and capture mode the sweep was submitted with (urf_set_params may have been called since),
         * behind whatever the context's stream still does with the row */
        const urf_dev_params dp_sub = sl.cap_dp;
        /* the rerun is the row's LATEST submission: with fewer rows than sweeps in flight a later sweep shares this row, and
         * what it left there is overwritten now -- its read-backs of the row must answer URF_ERR_BUSY, not this sweep's data */
        sl.gen = ++c->row_gen[slot_row(c, sl)];
        int rc = order_row_after_main(c, slot_row(c, sl), slot_stream(c, sl));
        if (rc == URF_OK)
            rc = slot_launch(c, sl, sl.n_points, sl.point_step, sl.off_x, sl.off_y, sl.off_z, &dp_sub, (int)sl.cap_a.capture);
        if (rc == URF_OK && hipStreamSynchronize(slot_stream(c, sl)) != hipSuccess) {
            c->last_error = "hipStreamSynchronize (rerun of a voided sweep)";
            rc = URF_ERR_HIP;
        }
        if (rc != URF_OK) {
            sl.pending = false;   /* the slot must not stay busy for ever */
            return rc;
        }
    }
    if (redo(sl.h_info->status)) {
        sl.pending = false;
        return URF_ERR_HIP;   /* (cannot happen: the full sequence raises neither) */
    }
    /* only now is the sweep "the last call": urf_read_stage / urf_marker_points / urf_ordered_indices look at
     * its scratch row, which stays untouched until the slot (or a batch call) is used again */
    c->last_scans = 1;
    c->last_a = sl.cap_a;
    c->last_dp = sl.cap_dp;
    c->last_is_slot = true;
    c->last_row = slot_row(c, sl);
    c->last_gen = sl.gen;
    if (labels_out)
        std::memcpy(labels_out, sl.h_labels, sl.n_points);
    if (info)
        *info = *sl.h_info;
    sl.pending = false;
    HT(5, ht0);
    return URF_OK;
}

extern "C" int urf_result_labels(urf_ctx* c, uint32_t ticket, const uint8_t** labels)
{
    if (!c || !labels)
        return URF_ERR_INVALID_ARG;
    *labels = nullptr;
    const urf_ctx::slot_t& sl = c->slots[ticket % URF_ASYNC_SLOTS];
    if (!sl.used || sl.ticket != ticket || !sl.h_labels)
        return URF_ERR_INVALID_ARG;   /* never issued, or its slot has been used again since */
    if (sl.pending)
        return URF_ERR_BUSY;          /* not waited for yet: the buffer is still being written */
    *labels = sl.h_labels;
    return URF_OK;
}

extern "C" int urf_classify_pc2(urf_ctx* c, const uint8_t* data, uint32_t n_points, uint32_t point_step,
                                uint32_t off_x, uint32_t off_y, uint32_t off_z, uint8_t* labels_out, urf_scan_info* info)
{
    if (!c || !data || !labels_out || !pc2_layout_ok(point_step, off_x, off_y, off_z))
        return URF_ERR_INVALID_ARG;   /* before any byte of the message is copied */
    if (n_points > c->max_points)
        return URF_ERR_CAPACITY;
    if (n_points == 0) {
        if (info) {
            std::memset(info, 0, sizeof(*info));
            info->status = URF_TOO_FEW_POINTS;
        }
        return URF_OK;
    }
    uint32_t ticket = 0;
    const int rc = urf_classify_pc2_async(c, data, n_points, point_step, off_x, off_y, off_z, &ticket);
    if (rc != URF_OK)
        return rc;
    return urf_classify_pc2_wait(c, ticket, labels_out, info);
}

/* urf_read_stage / urf_ordered_indices / urf_marker_points read the scratch ROW of the last call.  For a sweep of the
 * callback path that row is only intact while no later sweep has been submitted on it (slots that share a row:
 * max_batch < URF_MAX_IN_FLIGHT and more sweeps in flight than rows). */
static int last_row_intact(urf_ctx* c)
{
    if (c->last_is_slot && c->row_gen[c->last_row] != c->last_gen) {
        c->last_error = "the scratch row of the sweep waited for last has been resubmitted (create the context with max_batch >= "
                        "the number of sweeps in flight, or read its intermediate results before submitting on its row again)";
        return URF_ERR_BUSY;
    }
    if (c->last_is_slot && c->last_a.front) {
        /* ... and so did the sweep of the callback path that was waited for last (a context whose sweeps come row-major): once more on its
         * row through the general kernels -- the message is still in the slot's device buffer, the row has not been resubmitted (checked
         * above) --, as a batch call of one scan with every repair kernel in the sequence; the context stays with the general kernels */
        c->want_ring_sorted = true;
        c->epoch++;   /* (the captured sequences are rebuilt without the fused kernels) */
        const urf_kargs a = c->last_a;
        const urf_dev_params dp = c->last_dp;
        const int rc = run_pipeline(c, a.x, a.y, a.z, nullptr, a.n_per_scan, a.max_len, 1, a.labels, nullptr, c->last_row, nullptr, nullptr, nullptr, &dp,
                                    (int)a.capture, true);
        if (rc != URF_OK)
            return rc;
        c->last_is_slot = false;   /* (what the read-backs look at now is that call's) */
        return URF_OK;
    }
    if (!c->last_is_slot && c->last_a.front) {
        /* the last batch call went through the fused front end (urf_front.hpp), which keeps no ring-sorted copies: once more through
         * the legacy kernels (same inputs -- the caller's arrays must still be alive --, same parameters, same labels), and the
         * context stays with them: a caller that reads ring-sorted results pays for them once, not per call */
        c->want_ring_sorted = true;
        const urf_kargs a = c->last_a;
        const urf_dev_params dp = c->last_dp;
        return run_pipeline(c, a.x, a.y, a.z, a.offsets, a.n_per_scan, a.max_len, a.n_scans, a.labels, nullptr, 0, nullptr, nullptr, nullptr, &dp,
                            (int)a.capture, true);
    }
    return URF_OK;
}

/* ---- index-set and marker outputs: every scan of a batch in one launch sequence ------------- */
extern "C" int urf_compact_indices_batch(urf_ctx* c, const uint8_t* d_labels, uint32_t n_per_scan, uint32_t n_scans,
                                         uint32_t* d_road, uint32_t* d_curb, uint32_t* d_roi, uint32_t* d_ring10,
                                         uint32_t* d_counts)
{
    if (!c || !d_labels)
        return URF_ERR_INVALID_ARG;
    if (n_scans > c->max_batch || n_per_scan > c->max_points)
        return URF_ERR_CAPACITY;
    if (n_scans == 0 || n_per_scan == 0)
        return URF_OK;
    URF_HIP(c, hipSetDevice(c->device));
    {
        const int orc = order_after_slots(c);   /* compact_cnt is shared scratch */
        if (orc != URF_OK)
            return orc;
    }
    const unsigned tiles = (n_per_scan + URF_TILE - 1) / URF_TILE;
    const dim3 grid(tiles, n_scans);
    hipLaunchKernelGGL(k_compact_count, grid, dim3(URF_COMPACT_THREADS), 0, c->stream, d_labels, n_per_scan, tiles, c->compact_cnt);
    hipLaunchKernelGGL(k_compact_write, grid, dim3(URF_COMPACT_THREADS), 0, c->stream, d_labels, n_per_scan, tiles, c->compact_cnt,
                       d_road, d_curb, d_roi, d_ring10, d_counts);
    URF_HIP(c, hipGetLastError());
    return URF_OK;
}

extern "C" int urf_compact_indices(urf_ctx* c, const uint8_t* d_labels, uint32_t n_points,
                                   uint32_t* d_road, uint32_t* d_curb, uint32_t* d_roi, uint32_t* d_ring10,
                                   uint32_t* d_counts)
{
    return urf_compact_indices_batch(c, d_labels, n_points, 1, d_road, d_curb, d_roi, d_ring10, d_counts);
}

static int ensure_order_scratch(urf_ctx* c, uint32_t n_scans)
{
    if (n_scans <= c->ord_scans)
        return URF_OK;
    URF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ord_keys) {
        (void)hipFree(c->ord_keys);
        (void)hipFree(c->ord_pos);
        (void)hipFree(c->ord_cls);
        c->ord_keys = nullptr;
        c->ord_pos = nullptr;
        c->ord_cls = nullptr;
        c->ord_scans = 0;
    }
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    URF_HIP(c, hipMalloc(&p0, (size_t)n_scans * c->sstride * sizeof(unsigned long long)));
    URF_HIP(c, hipMalloc(&p1, (size_t)n_scans * c->sstride * sizeof(uint32_t)));
    URF_HIP(c, hipMalloc(&p2, (size_t)n_scans * URF_MAX_CHANNELS * 2 * sizeof(uint32_t)));
    c->ord_keys = (unsigned long long*)p0;
    c->ord_pos = (uint32_t*)p1;
    c->ord_cls = (uint32_t*)p2;
    c->ord_scans = n_scans;
    return URF_OK;
}

/* scans [s0, s0 + n) of the last classify call, lists of `stride` entries per scan on the device */
static int launch_ordered(urf_ctx* c, uint32_t s0, uint32_t n, uint32_t* d_road, uint32_t* d_curb, uint32_t* d_r10,
                          uint32_t stride, uint32_t* d_counts)
{
    int rc = last_row_intact(c);
    if (rc == URF_OK)
        rc = order_after_slots(c);
    if (rc == URF_OK)
        rc = ensure_order_scratch(c, n);
    if (rc != URF_OK)
        return rc;
    const urf_kargs a = c->last_a;   /* the call's own arguments and parameters, whatever was set since */
    const urf_dev_params dp = c->last_dp;
    hipLaunchKernelGGL(k_ring_order, dim3((unsigned)dp.p.channels, n), dim3(256), (2 * (size_t)a.tiles + 1) * sizeof(unsigned), c->stream, a, dp,
                       s0, c->ord_keys, c->ord_pos, c->ord_cls);
    hipLaunchKernelGGL(k_ordered_lists, dim3((unsigned)dp.p.channels, n), dim3(256), 0, c->stream, a, dp, s0, c->ord_pos, c->ord_cls, d_road,
                       d_curb, d_r10, stride, d_counts);
    URF_HIP(c, hipGetLastError());
    return URF_OK;
}

extern "C" int urf_ordered_indices_batch(urf_ctx* c, uint32_t* d_road, uint32_t* d_curb, uint32_t* d_ring10, uint32_t stride,
                                         uint32_t* d_counts)
{
    if (!c || !d_counts || c->last_scans == 0 || !c->last_a.labels || stride < c->last_a.max_len)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    return launch_ordered(c, 0, c->last_scans, d_road, d_curb, d_ring10, stride, d_counts);
}

extern "C" int urf_ordered_indices(urf_ctx* c, uint32_t scan, uint32_t* road, uint32_t* curb, uint32_t* ring10,
                                   uint32_t* counts)
{
    if (!c || !counts || scan >= c->last_scans || !c->last_a.labels)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    const size_t mp = c->sstride;
    if (!c->ord_lists) {
        void* p2 = nullptr;
        URF_HIP(c, hipMalloc(&p2, (mp * 3 + 4) * sizeof(uint32_t)));
        c->ord_lists = (uint32_t*)p2;
    }
    uint32_t* d_road = c->ord_lists;
    uint32_t* d_curb = d_road + mp;
    uint32_t* d_r10 = d_curb + mp;
    uint32_t* d_cnt = d_r10 + mp;
    hipStream_t st = c->stream;
    const int rc = launch_ordered(c, scan, 1, d_road, d_curb, d_r10, (uint32_t)mp, d_cnt);
    if (rc != URF_OK)
        return rc;
    uint32_t h[3] = { 0, 0, 0 };
    URF_HIP(c, hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    URF_HIP(c, hipStreamSynchronize(st));
    if (road && h[0])
        URF_HIP(c, hipMemcpy(road, d_road, h[0] * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (curb && h[1])
        URF_HIP(c, hipMemcpy(curb, d_curb, h[1] * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (ring10 && h[2])
        URF_HIP(c, hipMemcpy(ring10, d_r10, h[2] * sizeof(uint32_t), hipMemcpyDeviceToHost));
    counts[0] = h[0];
    counts[1] = h[1];
    counts[2] = h[2];
    return URF_OK;
}

static int launch_markers(urf_ctx* c, uint32_t s0, uint32_t n, float* d_pts, uint32_t* d_counts)
{
    {
        const int irc = last_row_intact(c);
        if (irc != URF_OK)
            return irc;
        const int orc = order_after_slots(c);
        if (orc != URF_OK)
            return orc;
    }
    const size_t cells = (size_t)URF_MAX_CHANNELS * URF_DEG_CELLS;
    if (n > c->mk_scans) {
        URF_HIP(c, hipStreamSynchronize(c->stream));
        for (void* p : { (void*)c->mk_d, (void*)c->mk_pos, (void*)c->mk_red })
            if (p)
                (void)hipFree(p);
        c->mk_d = nullptr;
        c->mk_pos = nullptr;
        c->mk_red = nullptr;
        c->mk_scans = 0;
        void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
        URF_HIP(c, hipMalloc(&p0, n * cells * sizeof(float)));
        URF_HIP(c, hipMalloc(&p1, n * cells * sizeof(uint32_t)));
        URF_HIP(c, hipMalloc(&p2, n * (cells + URF_MAX_CHANNELS)));   /* + one flag per ring: its order decides (k_marker_ring_literal) */
        c->mk_d = (float*)p0;
        c->mk_pos = (uint32_t*)p1;
        c->mk_red = (uint8_t*)p2;
        c->mk_scans = n;
    }
    const urf_kargs a = c->last_a;
    const urf_dev_params dp = c->last_dp;
    uint8_t* const mk_lit = c->mk_red + (size_t)c->mk_scans * cells;
    hipLaunchKernelGGL(k_marker_ring, dim3((unsigned)dp.p.channels, n), dim3(256), (2 * (size_t)a.tiles + 1) * sizeof(unsigned), c->stream, a, dp, s0,
                       c->mk_d, c->mk_pos, c->mk_red, mk_lit);
    /* rings in which the ORDER of equal azimuths decides a marker point (normally none: every workgroup returns at once) */
    hipLaunchKernelGGL(k_marker_ring_literal, dim3((unsigned)dp.p.channels, n), dim3(256), 0, c->stream, a, dp, s0, mk_lit, c->mk_d, c->mk_pos, c->mk_red);
    hipLaunchKernelGGL(k_marker_bins, dim3(n), dim3(384), 0, c->stream, a, dp, s0, c->mk_d, c->mk_pos, c->mk_red, d_pts, d_counts);
    URF_HIP(c, hipGetLastError());
    return URF_OK;
}

extern "C" int urf_marker_points_batch(urf_ctx* c, float* d_pts, uint32_t* d_counts)
{
    if (!c || !d_pts || !d_counts || c->last_scans == 0 || !c->last_a.labels)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    return launch_markers(c, 0, c->last_scans, d_pts, d_counts);
}

extern "C" int urf_marker_points(urf_ctx* c, uint32_t scan, float* pts, uint32_t* count)
{
    if (!c || !pts || !count || scan >= c->last_scans || !c->last_a.labels)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    if (!c->mk_out) {
        void* p3 = nullptr;
        URF_HIP(c, hipMalloc(&p3, (URF_DEG_CELLS * 4 + 4) * sizeof(float)));
        c->mk_out = (float*)p3;
    }
    hipStream_t st = c->stream;
    unsigned* d_cnt = (unsigned*)(c->mk_out + URF_DEG_CELLS * 4);
    const int rc = launch_markers(c, scan, 1, c->mk_out, d_cnt);
    if (rc != URF_OK)
        return rc;
    std::vector<float> h(URF_DEG_CELLS * 4 + 4);
    URF_HIP(c, hipMemcpyAsync(h.data(), c->mk_out, h.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    URF_HIP(c, hipStreamSynchronize(st));
    uint32_t n = 0;
    std::memcpy(&n, &h[URF_DEG_CELLS * 4], sizeof(n));
    if (n > URF_DEG_CELLS)
        return URF_ERR_HIP;
    std::memcpy(pts, h.data(), (size_t)n * 4 * sizeof(float));
    *count = n;
    return URF_OK;
}

/* ---- stage-wise inspection -------------------------------------------------- */
template <class T>
static int fetch(urf_ctx* c, std::vector<T>& dst, const T* src, size_t count)
{
    dst.resize(count);
    if (count)
        URF_HIP(c, hipMemcpy(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost));
    return URF_OK;
}

/* The ring-sorted slots of scan `scan` that hold a point (index relative to the scan's scratch
 * base) and the input index of each, rebuilt on the host from k_split's per-tile tables: tile t
 * fills its first troff[t][C] slots. */
static int ring_slot_sources(urf_ctx* c, uint32_t scan, uint32_t len, std::vector<uint32_t>& slot, std::vector<uint32_t>& src)
{
    const urf_kargs& k = c->last_a;
    const unsigned C = (unsigned)c->last_dp.p.channels;
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    std::vector<uint16_t> troff;
    std::vector<uint32_t> rec;
    int rc;
    if ((rc = fetch(c, troff, k.troff + (size_t)scan * k.tiles * (C + 1), (size_t)ntiles * (C + 1))) != URF_OK) return rc;
    if ((rc = fetch(c, rec, k.rec + (size_t)scan * k.sstride, (size_t)ntiles * URF_TILE)) != URF_OK) return rc;
    slot.clear();
    src.clear();
    for (unsigned t = 0; t < ntiles; t++)
        for (unsigned j = 0; j < troff[(size_t)t * (C + 1) + C]; j++) {
            slot.push_back(t * URF_TILE + j);
            src.push_back(t * URF_TILE + (rec[(size_t)t * URF_TILE + j] & URF_REC_SRC_MASK));
        }
    return URF_OK;
}

extern "C" int urf_read_stage(urf_ctx* c, urf_stage what, uint32_t scan, void* host_dst, size_t bytes)
{
    if (!c || !host_dst || scan >= c->last_scans)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    {
        const int irc = last_row_intact(c);
        if (irc != URF_OK)
            return irc;
        const int orc = order_after_slots(c);   /* (sweeps still in flight on other rows' streams) */
        if (orc != URF_OK)
            return orc;
    }
    URF_HIP(c, hipStreamSynchronize(c->stream));
    const urf_kargs& k = c->last_a;   /* the arguments and parameters of the call whose results are read */
    uint32_t len;
    if (k.offsets) {
        uint32_t o2[2];
        URF_HIP(c, hipMemcpy(o2, k.offsets + scan, sizeof(o2), hipMemcpyDeviceToHost));
        len = o2[1] - o2[0];
        if (len > k.max_len)
            len = k.max_len;
    } else {
        len = k.n_per_scan;
    }
    const unsigned C = (unsigned)c->last_dp.p.channels;
    const size_t sb = (size_t)scan * k.sstride;
    urf_scan_info in;
    URF_HIP(c, hipMemcpy(&in, k.info + scan, sizeof(in), hipMemcpyDeviceToHost));
    int rc;
    switch (what) {
    case URF_STAGE_VALPHA:
        if (k.capture != 1) return URF_ERR_INVALID_ARG;
        if (bytes < len * sizeof(float)) return URF_ERR_INVALID_ARG;
        URF_HIP(c, hipMemcpy(host_dst, k.valpha + sb, len * sizeof(float), hipMemcpyDeviceToHost));
        return URF_OK;
    case URF_STAGE_RING: {
        if (k.capture == 0) return URF_ERR_INVALID_ARG;
        if (bytes < len * sizeof(int16_t)) return URF_ERR_INVALID_ARG;
        std::vector<uint8_t> rk;
        if ((rc = fetch(c, rk, k.ringkey + sb, len)) != URF_OK) return rc;
        int16_t* o = (int16_t*)host_dst;
        for (uint32_t i = 0; i < len; i++)
            o[i] = (in.status != URF_OK || rk[i] == URF_RING_NONE) ? (int16_t)-1 : (int16_t)rk[i];
        return URF_OK;
    }
    case URF_STAGE_SECTOR: {
        if (k.capture == 0) return URF_ERR_INVALID_ARG;
        if (bytes < len * sizeof(int16_t)) return URF_ERR_INVALID_ARG;
        std::vector<uint16_t> sk;
        if ((rc = fetch(c, sk, k.seckey + sb, len)) != URF_OK) return rc;
        int16_t* o = (int16_t*)host_dst;
        for (uint32_t i = 0; i < len; i++)
            o[i] = sk[i] == URF_SEC_NONE ? (int16_t)-1 : (int16_t)sk[i];
        return URF_OK;
    }
    case URF_STAGE_AZIMUTH:
    case URF_STAGE_RANGE2D:
    case URF_STAGE_DETECT: {
        if (what != URF_STAGE_DETECT && k.capture != 1) return URF_ERR_INVALID_ARG;   /* exact values need the capture */
        const size_t esz = what == URF_STAGE_DETECT ? 1 : 4;
        if (bytes < len * esz) return URF_ERR_INVALID_ARG;
        std::memset(host_dst, 0, len * esz);
        if (in.status != URF_OK) return URF_OK;
        std::vector<uint32_t> slot, src;
        if ((rc = ring_slot_sources(c, scan, len, slot, src)) != URF_OK) return rc;
        const size_t span = (size_t)((len + URF_TILE - 1) / URF_TILE) * URF_TILE;
        if (what == URF_STAGE_DETECT) {
            std::vector<uint32_t> rec;   /* the detector hits of a slot's record */
            if ((rc = fetch(c, rec, k.rec + sb, span)) != URF_OK) return rc;
            uint8_t* o = (uint8_t*)host_dst;
            for (size_t p = 0; p < slot.size(); p++)
                o[src[p]] = (uint8_t)((rec[slot[p]] >> URF_REC_FLAG_SHIFT) & 7u);
        } else {
            std::vector<float> v;
            if ((rc = fetch(c, v, (what == URF_STAGE_AZIMUTH ? k.caz : k.rd2) + sb, span)) != URF_OK) return rc;
            float* o = (float*)host_dst;
            for (size_t p = 0; p < slot.size(); p++)
                o[src[p]] = v[slot[p]];
        }
        return URF_OK;
    }
    case URF_STAGE_ANGLE_TABLE: {
        if (bytes < C * sizeof(float)) return URF_ERR_INVALID_ARG;
        std::memset(host_dst, 0, C * sizeof(float));
        if (in.status != URF_OK) return URF_OK;
        URF_HIP(c, hipMemcpy(host_dst, k.angle + (size_t)scan * C, in.n_rings * sizeof(float), hipMemcpyDeviceToHost));
        return URF_OK;
    }
    case URF_STAGE_MAXDIST: {
        if (bytes < C * sizeof(float)) return URF_ERR_INVALID_ARG;
        std::memset(host_dst, 0, C * sizeof(float));
        if (in.status != URF_OK) return URF_OK;
        URF_HIP(c, hipMemcpy(host_dst, k.maxdist + (size_t)scan * C, in.n_rings * sizeof(float), hipMemcpyDeviceToHost));
        return URF_OK;
    }
    case URF_STAGE_QUADRANTS:
        if (bytes < 4 * sizeof(float)) return URF_ERR_INVALID_ARG;
        URF_HIP(c, hipMemcpy(host_dst, k.quad + (size_t)scan * 4, 4 * sizeof(float), hipMemcpyDeviceToHost));
        return URF_OK;
    case URF_STAGE_BEAM_STOP:
        if (bytes < 2 * URF_DEG_CELLS * sizeof(int16_t)) return URF_ERR_INVALID_ARG;
        URF_HIP(c, hipMemcpy(host_dst, k.stop_f + (size_t)scan * URF_DEG_CELLS, URF_DEG_CELLS * sizeof(int16_t), hipMemcpyDeviceToHost));
        URF_HIP(c, hipMemcpy((int16_t*)host_dst + URF_DEG_CELLS, k.stop_b + (size_t)scan * URF_DEG_CELLS,
                             URF_DEG_CELLS * sizeof(int16_t), hipMemcpyDeviceToHost));
        return URF_OK;
    }
    return URF_ERR_INVALID_ARG;
}

/* ---- test and benchmark hooks (include/urf_test_hooks.h) --------------------------------------------
 * Compiled only into liburf_hip_test.so (-DURF_ENABLE_TEST_HOOKS, urban_road_filter_amd/build.py): the product
 * library liburf_hip.so exports none of them. */
#ifdef URF_ENABLE_TEST_HOOKS
extern "C" int urf_set_debug_flags(urf_ctx* c, uint32_t flags)
{
    if (!c)
        return URF_ERR_INVALID_ARG;
    c->debug_flags = flags;
    c->dp.exp_flags = flags;
    c->epoch++;
    return URF_OK;
}

extern "C" int urf_selftest(urf_ctx* c, uint64_t* n_mismatches)
{
    if (!c || !n_mismatches)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    URF_HIP(c, hipMalloc((void**)&d, sizeof(*d)));
    URF_HIP(c, hipMemsetAsync(d, 0, sizeof(*d), c->stream));
    hipLaunchKernelGGL(k_selftest_div_pi, dim3(c->n_cus * 8), dim3(256), 0, c->stream, d);
    unsigned long long h = 0;
    hipError_t e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    URF_HIP(c, e);
    *n_mismatches = h;
    return URF_OK;
}

extern "C" int urf_selftest_fast(urf_ctx* c, uint64_t n_samples, float* err)
{
    if (!c || !err)
        return URF_ERR_INVALID_ARG;
    URF_HIP(c, hipSetDevice(c->device));
    unsigned* d = nullptr;
    URF_HIP(c, hipMalloc((void**)&d, 4 * sizeof(unsigned)));
    URF_HIP(c, hipMemsetAsync(d, 0, 4 * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_selftest_fast, dim3(c->n_cus * 8), dim3(256), 0, c->stream, (unsigned long long)n_samples, c->dp.Kfi, d);
    hipError_t e = hipMemcpyAsync(err, d, 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    URF_HIP(c, e);
    return URF_OK;
}

/* Benchmark helper: the submit / collect loop of a C or C++ client of the callback path (a ROS node's
 * subscriber callback and publisher), timed natively -- `n_sweeps` messages (taken round robin from
 * `msgs`), at most `in_flight` of them submitted before the oldest is collected. */
extern "C" int urf_bench_callback_stream(urf_ctx* c, const uint8_t* const* msgs, uint32_t n_msgs, uint32_t n_points,
                                         uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                                         uint32_t n_sweeps, uint32_t in_flight, int producer_pinned, uint8_t* labels_out,
                                         double* seconds)
{
    if (!c || !msgs || n_msgs == 0 || !seconds || in_flight == 0 || in_flight > URF_ASYNC_SLOTS)
        return URF_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < n_msgs; k++)
        if (!msgs[k])
            return URF_ERR_INVALID_ARG;
    const size_t bytes = (size_t)n_points * point_step;
    uint32_t tickets[URF_ASYNC_SLOTS];
    uint32_t head = 0, count = 0;   /* ring of tickets in flight */
    urf_scan_info info;
    struct timespec t0, t1;
    c->ht_on = getenv("URF_HOST_TIMES") != nullptr;
    for (double& v : c->ht)
        v = 0.0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto loop = [&]() -> int {
        for (uint32_t k = 0; k < n_sweeps; k++) {
            if (count == in_flight) {
                const int rc = urf_classify_pc2_wait(c, tickets[head], labels_out, &info);
                if (rc != URF_OK)
                    return rc;
                head = (head + 1) % URF_ASYNC_SLOTS;
                count--;
            }
            const uint8_t* data = msgs[k % n_msgs];
            if (producer_pinned) {   /* the producer fills the library's pinned buffer itself (each slot's once: producing the data is not what is timed) */
                uint8_t* pin = nullptr;
                const int rc = urf_pinned_input(c, bytes, &pin);
                if (rc != URF_OK)
                    return rc;
                if (k < URF_ASYNC_SLOTS)
                    std::memcpy(pin, data, bytes);
                data = pin;
            }
            uint32_t t = 0;
            const int rc = urf_classify_pc2_async(c, data, n_points, point_step, off_x, off_y, off_z, &t);
            if (rc != URF_OK)
                return rc;
            tickets[(head + count) % URF_ASYNC_SLOTS] = t;
            count++;
        }
        while (count) {
            const int rc = urf_classify_pc2_wait(c, tickets[head], labels_out, &info);
            if (rc != URF_OK)
                return rc;
            head = (head + 1) % URF_ASYNC_SLOTS;
            count--;
        }
        return URF_OK;
    };
    const int lrc = loop();
    if (lrc != URF_OK) {
        c->ht_on = false;   /* (every exit path: a later call must not pay for the clock reads) */
        return lrc;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (c->ht_on) {
        fprintf(stderr, "host us per sweep (pinned %d, in flight %u): total %.1f | stage+h2d %.1f order %.1f launch %.1f record %.1f wait %.1f rest-of-wait %.1f\n",
                producer_pinned, in_flight, 1e6 * *seconds / n_sweeps, 1e6 * c->ht[0] / n_sweeps, 1e6 * c->ht[1] / n_sweeps, 1e6 * c->ht[2] / n_sweeps,
                1e6 * c->ht[3] / n_sweeps, 1e6 * c->ht[4] / n_sweeps, 1e6 * c->ht[5] / n_sweeps);
        c->ht_on = false;
    }
    return URF_OK;
}

#endif   /* URF_ENABLE_TEST_HOOKS */

