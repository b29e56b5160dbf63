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

/* ------------------------------------------------------------------------- */
/* PointCloud2 records -> SoA                                                  */
/* ------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_pc2_to_soa(const uint8_t* __restrict__ data, unsigned long long n_total,
                                                    unsigned step, unsigned ox, unsigned oy, unsigned oz,
                                                    float* __restrict__ x, float* __restrict__ y, float* __restrict__ z)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total)
        return;
    const uint8_t* p = data + i * step;
    float fx, fy, fz;
    if ((((unsigned long long)(p + ox) | (unsigned long long)(p + oy) | (unsigned long long)(p + oz)) & 3ull) == 0) {
        fx = *(const float*)(p + ox);
        fy = *(const float*)(p + oy);
        fz = *(const float*)(p + oz);
    } else {
        unsigned bx = 0, by = 0, bz = 0;
        for (int b = 3; b >= 0; b--) {
            bx = (bx << 8) | p[ox + b];
            by = (by << 8) | p[oy + b];
            bz = (bz << 8) | p[oz + b];
        }
        fx = __uint_as_float(bx);
        fy = __uint_as_float(by);
        fz = __uint_as_float(bz);
    }
    x[i] = fx;
    y[i] = fy;
    z[i] = fz;
}

/* ------------------------------------------------------------------------- */
/* k_ring_table                                                                */
/* ------------------------------------------------------------------------- */
/* The reference walks the ROI points in order and appends a point's vertical
 * angle to the table when no earlier entry lies within `interval`
 * (lidar_segmentation.cpp:168-196).  Equivalent formulation used here: leader
 * k+1 is the first point after leader k that matches none of the leaders 0..k.
 * It runs first, straight from x/y/z, so that the one pass over the points that
 * follows (k_split) can already assign rings.
 *
 * One workgroup per scan, two alternating modes:
 *   serial   one wave takes the next 64 points; a new leader costs one ballot
 *            (an organised sweep fills the table within its first firing);
 *   scan     when a 64-point step brought no new leader, all waves look
 *            ahead 2048 points at a time for the first point that no leader
 *            matches (usually there is none: sweeps whose region of interest
 *            cuts off the outer rings never fill the table).
 * Matching "is there a leader within interval" is a bisection in a sorted copy
 * of the leaders: fl(leader - alpha) is monotone in the leader.  The walk ends as
 * soon as the table is full.  The `angle[j] == 0` end-of-table sentinel (:176)
 * is honoured: once a leader equal to 0 has been stored, only the leaders in
 * front of it take part in matching. */
__device__ __forceinline__ bool urf_leader_match(const float* SL, unsigned nmatch, float v, float interval)
{
    unsigned lo = 0, hi = nmatch;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (SL[mid] - v >= -interval)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo < nmatch && __builtin_fabsf(SL[lo] - v) <= interval;
}

/* Same question for a point, settled on the float approximation of its vertical angle whenever
 * that is clear of the +-interval boundaries by the approximation's error (urf_device.hpp). */
__device__ __forceinline__ bool urf_leader_match_point(const float* SL, unsigned nmatch, float x, float y, float z,
                                                       float interval)
{
    float vt;
    if (urf_fast_vertical_angle(x, y, z, &vt)) {
        const float e = URF_FAST_VALPHA_ERR + 2.0e-5f;
        unsigned lo = 0, hi = nmatch;
        while (lo < hi) {
            const unsigned mid = (lo + hi) >> 1;
            if (SL[mid] - vt >= -(interval + e))
                hi = mid;
            else
                lo = mid + 1;
        }
        if (lo == nmatch || SL[lo] - vt > interval + e)
            return false;
        if (__builtin_fabsf(SL[lo] - vt) <= interval - e)
            return true;
    }
    return urf_leader_match(SL, nmatch, urf_vertical_angle(x, y, z), interval);
}

#define URF_TABLE_SCAN_PPT 8
/* threads of k_ring_table: the look-ahead takes URF_TABLE_THREADS x 8 points per round trip.  (1024 threads -- two
 * rounds instead of eight for a sweep of the reference's default region of interest, which looks at ~14 700 points before
 * the speculation gives up -- gain a single sweep 2 us and cost a batch of 1024 sweeps 0.05 ms: the work is the same and
 * sixteen-wave workgroups wait longer at their barriers.  r4, measured.) */
#ifndef URF_TABLE_THREADS
#define URF_TABLE_THREADS 256
#endif
/* Speculation (lookahead > 0): when `lookahead` points in a row brought no new leader the walk stops
 * and hands the rest of the scan to k_split, which classifies every point against the table anyway:
 * a region-of-interest point behind the stop that matches no entry of a table that is not full
 * would have become a leader -- k_split then raises table_redo[s], k_table_repair builds the table
 * again without the shortcut and k_split_repair splits the scan again (and the context stops
 * speculating).  A sweep whose region of interest cuts the outer rings off (the reference's default)
 * otherwise pays a full extra pass over its points just to learn that no further ring shows up.
 * Not taken once a leader equal to 0 has been seen (matching is order dependent then). */
struct urf_table_shared {
    float L[URF_MAX_CHANNELS];    /* leaders in insertion order (the reference's angle[]) */
    float SL[URF_MAX_CHANNELS];   /* the matchable ones, ascending */
    unsigned nL, nmatch, zero, fresh;
    unsigned mins[URF_TABLE_THREADS / 64];
};
__device__ void urf_ring_table_scan(const urf_kargs& a, const urf_dev_params& dp, unsigned s, unsigned lookahead, urf_table_shared& T)
{
    float* const L = T.L;
    float* const SL = T.SL;
    unsigned& sh_nL = T.nL;
    unsigned& sh_nmatch = T.nmatch;
    unsigned& sh_zero = T.zero;
    unsigned& sh_new = T.fresh;
    unsigned* const sh_min = T.mins;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels;
    if (tid == 0) {
        urf_scan_info in;
        in.status = URF_OK;      /* k_offsets turns it into URF_TOO_FEW_POINTS when piece < 30 */
        in.n_roi = 0;
        in.n_rings = 0;
        in.n_ring_pts = 0;
        in.n_road = 0;
        in.n_curb = 0;
        in.n_ring10 = 0;
        in.n_nan_azimuth = 0;
        a.info[s] = in;
        sh_nL = 0;
        sh_nmatch = 0;
        sh_zero = 0;
    }
    if (tid < 4)
        a.nan_mask[(size_t)s * 4 + tid] = 0;   /* (k_table_repair: the bits k_split set against the old table are void) */
    __syncthreads();

    const float interval = dp.p.interval;
    /* Second speculation: a stream of sweeps from one sensor shows the same rings sweep after sweep.  Once the table holds as
     * many entries as the row's previous call found, the walk stops as it would after a quiet look-ahead -- a sweep of the
     * reference's default region finds its 61st and last ring 4 200 points in and then looked at 8 192 more for nothing.
     * k_split checks the rest of the scan either way; a failure of THIS rule only switches the rule off (urf_api.hip). */
    const unsigned hint = (lookahead && a.table_hint) ? *a.ring_hint : 0u;
    unsigned pos = 0, upto = 0xffffffffu;   /* upto: first point the walk did not look at (speculation) */
    unsigned cause = 0;
    while (pos < len && sh_nL < C) {
        if (hint && pos && sh_nL >= hint && !sh_zero) {   /* (uniform: LDS values behind a barrier) */
            upto = pos;
            cause = 2;
            break;
        }
        /* ---- serial step: wave 0, points [pos, pos + 64) ---- */
        if (wave == 0) {
            const unsigned i = pos + lane;
            float v = -1.0f;
            if (i < len) {
                const float x = a.x[off + i], y = a.y[off + i], z = a.z[off + i];
                if (urf_in_roi(dp.p, x, y, z))
                    v = urf_vertical_angle(x, y, z);
            }
            unsigned nL = sh_nL, nmatch = sh_nmatch;
            bool zero_seen = sh_zero != 0;
            const unsigned nL0 = nL;
            bool un = v >= 0.0f && !urf_leader_match(SL, nmatch, v, interval);
            unsigned long long m;
            /* One firing of an organised sweep = one point of every ring, steepest beam first: the unmatched points
             * ascend by more than `interval` from one to the next, so none of them matches another (fl(a - b) is
             * monotone in a) and ALL of them become leaders, in lane order -- in one step instead of one ballot,
             * shuffle and insertion per leader (64 dependent rounds: a quarter of this kernel's time on the first
             * firing).  Anything else (a zero angle, an angle below the largest entry, too many) takes the loop. */
            m = __ballot(un);
            if (m != 0 && !zero_seen) {
                const unsigned long long below = m & ((1ull << lane) - 1ull);
                const unsigned rank = (unsigned)__popcll(below);
                const int prev = below ? 63 - __clzll((long long)below) : (int)lane;
                const float pv = __shfl(v, prev);
                const float floor_v = nmatch ? SL[nmatch - 1] : -1.0f;   /* (angles are >= 0) */
                const bool ok = !un || (v != 0.0f && (below ? (v > pv && v - pv > interval) : v > floor_v));
                const unsigned cnt = (unsigned)__popcll(m);
                if (__ballot(!ok) == 0 && nL + cnt <= C) {
                    if (un) {
                        L[nL + rank] = v;
                        SL[nmatch + rank] = v;
                    }
                    nL += cnt;
                    nmatch += cnt;
                    un = false;
                }
            }
            while ((m = __ballot(un)) != 0 && nL < C) {
                const unsigned f = (unsigned)__ffsll((long long)m) - 1u;
                const float lv = __shfl(v, (int)f);
                bool matchable = false;
                if (lane == 0) {
                    L[nL] = lv;
                    if (!zero_seen && lv != 0.0f) {   /* insert into the sorted copy */
                        unsigned k = nmatch;
                        while (k > 0 && SL[k - 1] > lv) {
                            SL[k] = SL[k - 1];
                            k--;
                        }
                        SL[k] = lv;
                    }
                }
                if (!zero_seen) {
                    if (lv == 0.0f)
                        zero_seen = true;
                    else {
                        nmatch++;
                        matchable = true;
                    }
                }
                nL++;
                if (lane <= f)
                    un = false;
                else if (matchable && __builtin_fabsf(lv - v) <= interval)
                    un = false;
            }
            if (lane == 0) {
                sh_nL = nL;
                sh_nmatch = nmatch;
                sh_zero = zero_seen ? 1u : 0u;
                sh_new = nL != nL0;
            }
        }
        __syncthreads();
        pos += 64;
        if (sh_new || sh_nL >= C)
            continue;
        /* ---- scan mode: first point in [pos, len) that no leader matches ---- */
        const unsigned nmatch = sh_nmatch;
        unsigned quiet = 0;   /* points looked at since the last new leader */
        while (pos < len) {
            unsigned first = 0xffffffffu;
            float px[URF_TABLE_SCAN_PPT], py[URF_TABLE_SCAN_PPT], pz[URF_TABLE_SCAN_PPT];
#pragma unroll
            for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++) {   /* all loads in flight first */
                const unsigned i = pos + q * URF_TABLE_THREADS + tid;
                const bool on = i < len;
                px[q] = on ? a.x[off + i] : 0.f;
                py[q] = on ? a.y[off + i] : 0.f;
                pz[q] = on ? a.z[off + i] : 0.f;
            }
            {
                /* urf_leader_match_point for the thread's eight points together: the bisections in the sorted
                 * leaders step by step side by side (their LDS reads in flight together; point after point, each
                 * behind a short-circuit, the look-ahead cost 9 us per 2048 points) */
                const float e = URF_FAST_VALPHA_ERR + 2.0e-5f;
                float vt[URF_TABLE_SCAN_PPT];
                unsigned lb[URF_TABLE_SCAN_PPT], okm = 0, roim = 0;
#pragma unroll
                for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++) {
                    const unsigned i = pos + q * URF_TABLE_THREADS + tid;
                    roim |= (unsigned)((i < len) & urf_in_roi(dp.p, px[q], py[q], pz[q])) << q;
                }
                /* (a wave none of whose 512 points lies in the region of interest has nothing to match: the reference's
                 * default region drops whole azimuth ranges of a sweep, i.e. whole firings) */
                if (__ballot(roim != 0u) != 0ull) {
#pragma unroll
                for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++) {
                    okm |= (unsigned)urf_fast_vertical_angle(px[q], py[q], pz[q], &vt[q]) << q;
                    lb[q] = 0;
                }
#pragma unroll
                for (unsigned step = URF_MAX_CHANNELS / 2; step > 0; step >>= 1) {   /* first entry with SL - vt >= -(interval + e) */
                    float sv[URF_TABLE_SCAN_PPT];
#pragma unroll
                    for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++)
                        sv[q] = SL[(lb[q] + step - 1) & (URF_MAX_CHANNELS - 1)];
#pragma unroll
                    for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++)
                        lb[q] += (lb[q] + step - 1 < nmatch && !(sv[q] - vt[q] >= -(interval + e))) ? step : 0u;
                }
                float cv[URF_TABLE_SCAN_PPT];
#pragma unroll
                for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++)
                    cv[q] = SL[lb[q] & (URF_MAX_CHANNELS - 1)];
#pragma unroll
                for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++)   /* (the steps add up to 127: a full table of 128 entries all below) */
                    lb[q] += (lb[q] == URF_MAX_CHANNELS - 1 && lb[q] < nmatch && !(cv[q] - vt[q] >= -(interval + e))) ? 1u : 0u;
#pragma unroll
                for (unsigned q = 0; q < URF_TABLE_SCAN_PPT; q++) {
                    const unsigned i = pos + q * URF_TABLE_THREADS + tid;
                    if (!((roim >> q) & 1u))
                        continue;
                    /* (lb can reach nmatch only through entries < nmatch, so lb <= nmatch <= 128; an index of 128
                     * wraps to entry 0 and is not looked at: lb == nmatch) */
                    const float d = cv[q] - vt[q];
                    const bool fast = (okm >> q) & 1u;
                    const bool none = lb[q] >= nmatch || d > interval + e;
                    const bool sure = !none && __builtin_fabsf(d) <= interval - e;
                    bool matched;
                    if (fast && (none || sure))
                        matched = sure;
                    else
                        matched = urf_leader_match(SL, nmatch, urf_vertical_angle(px[q], py[q], pz[q]), interval);
                    if (!matched && i < first)
                        first = i;
                }
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned w = __shfl_xor(first, o);
                first = w < first ? w : first;
            }
            if (lane == 0)
                sh_min[wave] = first;
            __syncthreads();
            unsigned m = sh_min[0];
            for (int w = 1; w < URF_TABLE_THREADS / 64; w++)
                m = sh_min[w] < m ? sh_min[w] : m;
            __syncthreads();
            if (m != 0xffffffffu) {
                pos = m;   /* the serial step resumes exactly there */
                break;
            }
            pos += URF_TABLE_THREADS * URF_TABLE_SCAN_PPT;
            quiet += URF_TABLE_THREADS * URF_TABLE_SCAN_PPT;
            if (lookahead && quiet >= lookahead && !sh_zero && pos < len) {
                upto = pos;
                cause = 1;
                break;
            }
        }
        if (upto != 0xffffffffu)
            break;
    }
    __syncthreads();
    if (tid == 0) {
        a.table_upto[s] = upto;
        a.table_redo[s] = 0;
        a.table_cause[s] = cause;
    }

    /* std::sort(angle, angle + index), lidar_segmentation.cpp:205 (rank sort) */
    const unsigned n = sh_nL;
    if (tid < n) {
        const float v = L[tid];
        unsigned rank = 0;
        for (unsigned j = 0; j < n; j++) {
            const float w = L[j];
            rank += (w < v) || (w == v && j < tid);
        }
        a.angle[(size_t)s * C + rank] = v;
        SL[rank] = v;   /* the matching copy is no longer needed */
    }
    if (tid == 0)
        a.info[s].n_rings = n;
    __syncthreads();
    /* k_split decides rings on u = -z / rho = cot(vertical angle) (urf_device.hpp: urf_fast_cot): per
     * table entry the thresholds on u (urf_ring_thresholds), and a lookup table over u: cell c covers
     * [c / 512 - 4, (c + 1) / 512 - 4); lut[c] = number of entries that lie surely below the window of
     * every point of the cell (u < their .x; the .x fall with the entry's index).  A point starts its
     * search at lut[cell] and usually ends it there or one entry later, instead of bisecting the table.
     * (A cell index that float rounding pushes up by one only lowers the count: still valid.) */
    {
        const float e = URF_FAST_VALPHA_ERR + 2.0e-5f;
        /* one cotangent (a binary64 polynomial and two divisions) per thread: entry tid / 4, threshold tid % 4 */
        for (unsigned k = tid; k < 4 * n; k += URF_TABLE_THREADS) {
            const float t = urf_ring_threshold(SL[k >> 2], interval, e, k & 3u);
            a.ring_thr[((size_t)s * C) * 4 + k] = t;
            if ((k & 3u) == 0)
                L[k >> 2] = t;   /* .x; the leaders in insertion order are no longer needed */
        }
        __syncthreads();
        uint8_t* lut = a.ring_lut + (size_t)s * URF_LUT_CELLS;
        /* four cells per thread at a time, their bisections step by step together (the dependent LDS reads of
         * one cell after the other were the longest chain of this kernel) */
        for (unsigned c0 = tid; c0 < URF_LUT_CELLS; c0 += 4 * URF_TABLE_THREADS) {
            float u1[4];
            unsigned lo[4];
#pragma unroll
            for (unsigned q = 0; q < 4; q++) {
                u1[q] = (float)(c0 + q * URF_TABLE_THREADS + 1) * (1.0f / URF_LUT_SCALE) - URF_LUT_UMAX;   /* exact */
                lo[q] = 0;
            }
#pragma unroll
            for (unsigned step = URF_MAX_CHANNELS; step > 0; step >>= 1) {
                float lv[4];
#pragma unroll
                for (unsigned q = 0; q < 4; q++)
                    lv[q] = L[(lo[q] + step - 1) & (URF_MAX_CHANNELS - 1)];
#pragma unroll
                for (unsigned q = 0; q < 4; q++)
                    lo[q] += (lo[q] + step - 1 < n && lv[q] >= u1[q]) ? step : 0u;
            }
#pragma unroll
            for (unsigned q = 0; q < 4; q++)
                if (c0 + q * URF_TABLE_THREADS < URF_LUT_CELLS)
                    lut[c0 + q * URF_TABLE_THREADS] = (uint8_t)lo[q];
        }
    }
}

__global__ __launch_bounds__(URF_TABLE_THREADS) void k_ring_table(urf_kargs a, urf_dev_params dp)
{
    __shared__ urf_table_shared T;
    if (blockIdx.x == 0 && threadIdx.x < 8)
        a.star_count[threadIdx.x] = 0;   /* the call's work-list lengths (k_table_repair, k_index): first kernel of the sequence */
    {   /* the fused front end's per-scan state (urf_front.hpp): every scan is a candidate until k_front finds otherwise */
        const unsigned s = blockIdx.x, tid = threadIdx.x;
        if (tid == 0) {
            a.front_ok[s] = a.front;
            a.front_ncand[s] = 0;
        }
        if (a.front) {
            if (tid < 64)
                a.front_lane_ring[(size_t)s * 64 + tid] = 0xffffffffu;
            if (tid < (unsigned)dp.p.channels)
                a.front_ring_lane[(size_t)s * dp.p.channels + tid] = 0xffffffffu;
        }
    }
    urf_ring_table_scan(a, dp, blockIdx.x, a.table_lookahead, T);
}

/* the scans whose speculative table k_split found incomplete: the whole walk, listed for k_split_repair */
__global__ __launch_bounds__(URF_TABLE_THREADS) void k_table_repair(urf_kargs a, urf_dev_params dp, unsigned collect)
{
    __shared__ urf_table_shared T;
    const unsigned s = blockIdx.x;
    const bool redo = a.table_redo[s] != 0u;
    /* collect (the launch behind k_front, urf_front.hpp): the scans the fused front end handed back -- or whose speculative table it
     * found incomplete -- are listed for the list-driven legacy kernels; host-visible: was there one, were they all */
    if (collect && threadIdx.x == 0 && (redo || a.front_ok[s] == 0u)) {
        a.front_ok[s] = 0u;
        const unsigned e = atomicAdd(&a.star_count[6], 1u);
        a.front_list[e] = s;
        a.front_state[0] = 1u;
        if (e + 1u == a.n_scans)
            a.front_state[1] = 1u;
    }
    if (!redo)
        return;
    if (a.front && threadIdx.x == 0)
        a.front_ok[s] = 0u;   /* k_split_repair splits the scan the legacy way: the legacy kernels take it from here (urf_front.hpp) */
    const unsigned cause = a.table_cause[s];   /* (before the walk below overwrites it) */
    urf_ring_table_scan(a, dp, s, 0, T);
    if (threadIdx.x == 0) {
        if (!collect)   /* (a collected scan is split by k_split_list) */
            a.redo_list[atomicAdd(&a.star_count[2], 1u)] = s;
        a.spec_failed[cause == 2u ? 1 : 0] = 1u;   /* host-visible: the context stops using the rule that failed */
    }
}

/* ------------------------------------------------------------------------- */
/* k_split                                                                     */
/* ------------------------------------------------------------------------- */
/* ONE pass over x/y/z per tile of 2048 input points: ROI test, ring of every point (float fast path
 * with margins, exact sequence for the rare open point), star sector, then a stable multi-split of
 * the tile by ring and by sector, written into the tile's own region of the scratch arrays
 * (urf_internal.hpp: scratch layout) -- no total over the scan is needed before writing, so the
 * points are read from HBM once.  Input order is preserved inside every ring, which x_zero /
 * z_zero rely on (lidar_segmentation.cpp:280-283 run before the azimuth sort :289).
 *
 * Wave w of the workgroup owns the 256 consecutive points [w*256, w*256+256) of the tile and walks
 * them 64 at a time.  A point's rank inside its key within the tile =
 *     points of that key in earlier waves of this tile       (LDS matrix wcnt[wave][key], scanned per key)
 *   + ... in earlier 64-point steps of its own wave          (running value of wcnt[wave][key])
 *   + ... in lower lanes of its own step                     (match_any + popcount).
 *
 * Ring-sorted stores: in firing order the 64 lanes of a wave belong to 64 different rings.  The
 * tile is therefore transposed through LDS (slot = position in the tile's ring-sorted order,
 * row-padded against bank conflicts) and written out slot by slot, 2048 consecutive elements per
 * array.  Sector-sorted stores go out directly (a firing shares one sector: consecutive ranks). */
/* (one pad word per 32 slots: the lanes of a step of an organised 64-ring tile write slots 32 apart -- lane * 33 + c are 32
 * different banks per half wave; with one pad word per 64 slots, r2-r4, lanes 2k and 2k + 1 shared a bank: every staging
 * store took twice its cycles) */
#ifdef URF_EXP_SLOT_PAD64
#define URF_SLOT(lp) ((lp) + ((lp) >> 6))
#define URF_SLOTS (URF_TILE + URF_TILE / 64)
#else
#define URF_SLOT(lp) ((lp) + ((lp) >> 5))
#define URF_SLOTS (URF_TILE + URF_TILE / 32)
#endif
#define URF_TILE_WAVES (URF_TILE_THREADS / 64)
#define URF_WAVE_PTS (URF_TILE / URF_TILE_WAVES)   /* consecutive points of the tile a wave owns */

__host__ __device__ inline unsigned urf_align16(unsigned v) { return (v + 15u) & ~15u; }

/* LDS carve of k_split: tab | ul | thr | lut | koff[C+1] | soff[Ks+1] | misc[64] | tmax[C] u64 |
 * union { keyr[T] u8, keys[T] u16, pending[T] u16, wcnt_r[W][C] u16, wcnt_s[W][Ks] u16 ;
 *         staging x y z record [URF_SLOTS] u32 } */
/* k_split's small words: [0] ROI points, [1] pending, [2 + wave] wave sums, [30] "not organised with holes", [31] "not organised",
 * [32 + step] step keys, [64 + step] points of the step that take part in the star-shaped search, [96 + step] ... in the steps
 * before it ([128]: in the tile), [130 + step] step keys with the empty steps filled in (key + 1; 0: none yet), [162 + step] the
 * step keys of a tile with holes (sector of the step's first lane that has one), [200 ...) one byte per (ring, wave): points of the
 * ring among the wave's four firings */
#define URF_SPLIT_MISC_WORDS 336
__host__ __device__ inline size_t urf_split_lds_bytes(unsigned C, unsigned K, bool star)
{
    const unsigned Ks = star ? K : 0;
    const size_t fixed = URF_MAX_CHANNELS * (4 + 4 + 16) + urf_align16(URF_LUT_CELLS) + urf_align16((C + 1) * 4) +
                         urf_align16((Ks + 1) * 4) + URF_SPLIT_MISC_WORDS * 4 + urf_align16(C * 8);
    const size_t phase_a = 5 * (size_t)URF_TILE + urf_align16(2 * URF_TILE_WAVES * (C + Ks));
    const size_t phase_b = 4 * (size_t)URF_SLOTS * 4;
    return fixed + (phase_a > phase_b ? phase_a : phase_b);
}

/* The reference's exact sequence for one point: vertical angle (lidar_segmentation.cpp:148-166), first
 * sorted table entry within `interval` (:226-233; fl(angle[j] - alpha) is monotone in angle[j], so
 * the matching entries are contiguous and the first one is found by bisection with the very same
 * float predicate), star sector (star_shaped_search.cpp:164-171; sectors == 0: not wanted).
 * Deliberately NOT inlined: only the rare point the float approximations leave open gets here, and
 * inlined its f64 polynomials would dictate the register allocation (and with it the occupancy)
 * of the whole kernel. */
struct urf_exact_key {
    unsigned ring, sector;
    float valpha;
};
#ifdef URF_EXP_INLINE_EXACT   /* (the r2 build whose parity gate failed at 8 waves per SIMD: kept buildable for the race screen) */
#define URF_EXACT_INLINE __forceinline__
#else
#define URF_EXACT_INLINE __noinline__
#endif
__device__ __forceinline__ urf_exact_key urf_exact_keys_body(const float* tab, unsigned nR, float interval, float x, float y, float z,
                                                            unsigned sectors, float Kfi)
{
    urf_exact_key r;
    r.valpha = urf_vertical_angle(x, y, z);
    r.ring = URF_RING_NONE;
    r.sector = URF_SEC_NONE;
    unsigned l2 = 0, h2 = nR;
    while (l2 < h2) {
        const unsigned mid = (l2 + h2) >> 1;
        if (tab[mid] - r.valpha >= -interval)
            h2 = mid;
        else
            l2 = mid + 1;
    }
    if (l2 < nR && __builtin_fabsf(tab[l2] - r.valpha) <= interval)
        r.ring = l2;
    if (sectors)
        r.sector = urf_sector(x, y, Kfi, sectors);
    return r;
}
__device__ URF_EXACT_INLINE urf_exact_key urf_exact_keys(const float* tab, unsigned nR, float interval, float x, float y, float z,
                                                     unsigned sectors, float Kfi)
{
    return urf_exact_keys_body(tab, nR, interval, x, y, z, sectors, Kfi);
}

/* timing experiment (tools/ab_noparity.sh): wave 0 of a few workgroups in the middle of the grid prints
 * the shader-clock cycles between the kernel's barriers */
#ifdef URF_EXP_PHASE_CLOCK
#define URF_PHASE_DECL unsigned long long ph_t[16]; unsigned ph_n = 0; ph_t[ph_n++] = __builtin_amdgcn_s_memtime()
#define URF_PHASE_MARK ph_t[ph_n++] = __builtin_amdgcn_s_memtime()
#define URF_PHASE_DUMP(name)                                                                                  \
    if (threadIdx.x == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.x < 4) {                                   \
        for (unsigned ph_i = 1; ph_i < ph_n; ph_i++)                                                           \
            printf("%s wg %u phase %u: %llu cycles\n", name, blockIdx.x, ph_i, ph_t[ph_i] - ph_t[ph_i - 1]); \
    }
#define URF_PHASE_ACC(k) do { const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); ph_t[k] += ph_now - ph_last; ph_last = ph_now; } while (0)
#define URF_PHASE_ACC_DECL unsigned long long ph_t[12] = { 0 }, ph_last = __builtin_amdgcn_s_memtime()
#define URF_PH_PARAMS , unsigned long long* ph_t, unsigned long long& ph_last
#define URF_PH_ARGS , ph_t, ph_last
#define URF_PHASE_ACC_DUMP(name, n)                                                                           \
    if (threadIdx.x == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.x < 4) {                                   \
        for (unsigned ph_i = 0; ph_i < (n); ph_i++)                                                            \
            printf("%s wg %u phase %u: %llu cycles\n", name, blockIdx.x, ph_i, ph_t[ph_i]);                   \
    }
#else
#define URF_PHASE_DECL
#define URF_PHASE_MARK
#define URF_PHASE_DUMP(name)
#define URF_PHASE_ACC(k)
#define URF_PHASE_ACC_DECL
#define URF_PHASE_ACC_DUMP(name, n)
#define URF_PH_PARAMS
#define URF_PH_ARGS
#endif
#ifndef URF_SPLIT_WAVES_PER_EU
#define URF_SPLIT_WAVES_PER_EU 6   /* 73 VGPRs without spills; A/B on one box: 4 -> 1.39 ms, 6 -> 1.04 ms, 8 (32 B of scratch) -> 1.12 ms */
#endif
/* ORGANISED WITH HOLES (r5; 64 rings): an organised tile with points MISSING -- a real sensor's drop-outs, rings a region of
 * interest cuts off -- every point that is there sits on its expected ring (C == 64: ring = lane), the points of a step that take
 * part in the star-shaped search share one sector, the sectors of the steps that have one do not fall.  The slots are then counts
 * of the points that are there: ring c starts behind the rings in front of it and holds its points in firing order (per ring and
 * wave one byte of counts, read back as one 8-byte word per ring); the sector-sorted order is the input order of the points that
 * take part (per step a count, one scan over the 32 of them).  Two ballots per step, one scan per family, no match_any, no counter
 * matrix, one barrier instead of three.  Called by every thread of the workgroup for a tile that failed the first test; NOT inlined:
 * inside k_split its few registers tipped the kernel over its 80 (the general path spilled 16 bytes).  misc: URF_SPLIT_MISC_WORDS
 * of LDS (k_split's layout), koff: the ring run table.  Returns ok = 0 when the tile does not have the shape. */
struct urf_holey_slots {
    unsigned lp[4], sp[4], ok;
};
__device__ __noinline__ urf_holey_slots urf_split_holey(unsigned r0, unsigned r1, unsigned r2, unsigned r3, unsigned s0, unsigned s1, unsigned s2,
                                                        unsigned s3, unsigned wave, unsigned lane, bool star, unsigned* misc, unsigned* koff)
{
    unsigned* const stepcnt = misc + 64;
    unsigned* const stepbase = misc + 96;
    unsigned* const stepfk = misc + 130;
    unsigned* const stepkh = misc + 162;
    uint8_t* const ringcnt = (uint8_t*)(misc + 200);
    const unsigned rk[4] = { r0, r1, r2, r3 }, sk[4] = { s0, s1, s2, s3 };
    urf_holey_slots out;
    bool mine_h = true;
    unsigned own = 0;   /* points of this lane's ring among the wave's four firings */
#pragma unroll
    for (unsigned q = 0; q < 4; q++) {
        const bool onr = rk[q] != URF_RING_NONE, ons = sk[q] != URF_SEC_NONE;
        mine_h = mine_h & (!onr | (rk[q] == lane));
        own += onr ? 1u : 0u;
        if (star) {
            const unsigned long long psm = __ballot(ons);   /* lanes of the step that take part in the star-shaped search */
            const unsigned src = psm ? (unsigned)__ffsll((long long)psm) - 1u : 0u;
            const unsigned f = psm ? (unsigned)__builtin_amdgcn_readlane((int)sk[q], (int)src) : URF_SEC_NONE;   /* the step's sector */
            mine_h = mine_h & (!ons | (sk[q] == f));
            if (lane == 0) {
                stepkh[wave * 4 + q] = f;
                stepcnt[wave * 4 + q] = (unsigned)__popcll(psm);
            }
        }
    }
    ringcnt[lane * URF_TILE_WAVES + wave] = (uint8_t)own;
    if (__ballot(!mine_h) != 0ull && lane == 0)
        misc[30] = 1u;   /* (every writer writes the same value) */
    __syncthreads();
    out.ok = 0;
#pragma unroll
    for (unsigned q = 0; q < 4; q++)
        out.lp[q] = out.sp[q] = 0xffffffffu;
    if (misc[30] != 0u)
        return out;   /* (uniform) */
    /* the steps' keys must not fall, steps without a key skipped: key + 1 against the largest in front of it */
    const unsigned k1 = (star && lane < URF_TILE_GROUPS && stepkh[lane & 31u] != URF_SEC_NONE) ? stepkh[lane & 31u] + 1u : 0u;
    const unsigned fk = urf_wave_scan_max(k1);
    unsigned exc = (unsigned)__shfl_up((int)fk, 1);
    exc = lane == 0 ? 0u : exc;
    if (__ballot(k1 != 0u && k1 < exc) != 0ull)
        return out;   /* (uniform) */
    out.ok = 1;
    /* ring `lane`: its points in the waves in front of this one, in the whole tile; first slot = the rings in front of it */
    const unsigned long long w8 = ((const unsigned long long*)ringcnt)[lane];
    unsigned before = 0, total = 0;
#pragma unroll
    for (unsigned w = 0; w < URF_TILE_WAVES; w++) {
        const unsigned b = (unsigned)(w8 >> (8u * w)) & 0xffu;
        total += b;
        before += w < wave ? b : 0u;
    }
    const unsigned kinc = urf_wave_scan_add(total);
    unsigned run = kinc - total + before;
    /* the points of the steps in front of each of the wave's four that take part in the star-shaped search */
    const unsigned sc = (star && lane < URF_TILE_GROUPS) ? stepcnt[lane & 31u] : 0u;
    const unsigned sinc = urf_wave_scan_add(sc);
#pragma unroll
    for (unsigned q = 0; q < 4; q++) {
        const bool onr = rk[q] != URF_RING_NONE, ons = sk[q] != URF_SEC_NONE;
        out.lp[q] = onr ? run : 0xffffffffu;
        run += onr ? 1u : 0u;
        const unsigned sb0 = (unsigned)__shfl((int)(sinc - sc), (int)(wave * 4 + q));
        out.sp[q] = ons ? sb0 + urf_popc_below(__ballot(ons)) : 0xffffffffu;
    }
    if (wave == 0) {   /* the run tables (the sector table: k_split, once stepbase / stepfk can be read) */
        koff[lane] = kinc - total;
        if (lane == 63)
            koff[64] = kinc;
        if (lane < URF_TILE_GROUPS) {
            stepbase[lane] = sinc - sc;
            stepfk[lane] = fk;   /* the step keys with the empty steps filled in from the left (key + 1; 0: none yet) */
        }
        if (lane == URF_TILE_GROUPS - 1u)
            stepbase[URF_TILE_GROUPS] = sinc;
    }
    return out;
}

__device__ __forceinline__ void urf_split_tile(const urf_kargs& a, const urf_dev_params& dp, unsigned s, unsigned t, unsigned char* sh_raw,
                                               const unsigned tid)
{
#ifdef URF_EXP_WAVE_RFL
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63;
#else
    const unsigned wave = tid >> 6, lane = tid & 63;
#endif
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned tbase = t * URF_TILE;
    if (tbase >= len)
        return;
    URF_PHASE_DECL;
    /* (uniform 64-bit bases + small per-lane offsets: per-lane 64-bit addresses cost register pairs) */
    const float* __restrict__ const gx = a.x + ((size_t)off + tbase);
    const float* __restrict__ const gy = a.y + ((size_t)off + tbase);
    const float* __restrict__ const gz = a.z + ((size_t)off + tbase);
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    const bool star = dp.p.star_shaped_method != 0;
    const unsigned Ks = star ? K : 0;
    float* tab = (float*)sh_raw;                       /* the sorted ring angles (exact pass) */
    float* ul = tab + URF_MAX_CHANNELS;                /* urf_ring_thr::x of every entry: the probes */
    urf_ring_thr* thr = (urf_ring_thr*)(ul + URF_MAX_CHANNELS);
    uint8_t* lut = (uint8_t*)(thr + URF_MAX_CHANNELS);
    unsigned* koff = (unsigned*)(lut + urf_align16(URF_LUT_CELLS));
    unsigned* soff = koff + urf_align16((C + 1) * 4) / 4;
    unsigned* misc = soff + urf_align16((Ks + 1) * 4) / 4;   /* [0] ROI points, [1] pending, [2 + wave] wave sums, [31] "not organised", [32 + step] step keys */
    static_assert(2 + URF_TILE_WAVES <= 30 && URF_TILE_GROUPS <= 32 && 200 + URF_TILE_WAVES * 64 / 4 <= URF_SPLIT_MISC_WORDS, "misc[2 + wave], misc[32 + step], ringcnt");
    unsigned* const stepkey = misc + 32;
    unsigned* const stepbase = misc + 96;
    unsigned* const stepfk = misc + 130;
    unsigned long long* tmax = (unsigned long long*)(misc + URF_SPLIT_MISC_WORDS);   /* largest x*x + y*y per ring (binary64 bits: non-negative doubles order like integers) */
    unsigned char* un = (unsigned char*)(tmax + urf_align16(C * 8) / 8);
    uint8_t* keyr = un;
    uint16_t* keys = (uint16_t*)(un + URF_TILE);
    uint16_t* pending = keys + URF_TILE;
    uint16_t* wcnt_r = pending + URF_TILE;
    uint16_t* wcnt_s = wcnt_r + (size_t)URF_TILE_WAVES * C;
    const unsigned sb = urf_sbase(a, s);

    /* Everything the workgroup needs from memory is requested in ONE round trip, none of it depending
     * on another load's result: the scan's tables (unconditionally -- entries at or beyond n_rings are
     * never looked at) first, then the tile's points.  (With the table loads depending on n_rings the
     * workgroup spent a third of its life, 12 000 of 34 000 cycles, in front of its first barrier.) */
    constexpr unsigned Q = URF_TILE / URF_TILE_THREADS;
    constexpr unsigned LUT_WORDS = URF_LUT_CELLS / 4, LUT_PT = (LUT_WORDS + URF_TILE_THREADS - 1) / URF_TILE_THREADS;
    const unsigned trow = tid < URF_MAX_CHANNELS ? tid : 0u;
    const float tab_v = a.angle[(size_t)s * C + trow];
    const float4 thr_v = ((const float4*)a.ring_thr)[(size_t)s * C + trow];
    unsigned lut_v[LUT_PT];
#pragma unroll
    for (unsigned e = 0; e < LUT_PT; e++) {
        const unsigned i = tid + e * URF_TILE_THREADS;
        lut_v[e] = ((const unsigned*)(a.ring_lut + (size_t)s * URF_LUT_CELLS))[i < LUT_WORDS ? i : 0u];
    }
    float px[Q], py[Q], pz[Q];
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const unsigned li = wave * URF_WAVE_PTS + q * 64 + lane;
        const bool valid = tbase + li < len;
        px[q] = valid ? gx[li] : 0.f;
        py[q] = valid ? gy[li] : 0.f;
        pz[q] = valid ? gz[li] : 0.f;
    }
    const unsigned nR = a.info[s].n_rings;
    /* a speculative ring table (k_ring_table) is checked here: a region-of-interest point at or behind
     * `upto` that matches none of its entries would have been a new leader */
    const unsigned upto_v = a.table_upto[s];
    const unsigned upto = nR < C ? upto_v : 0xffffffffu;
    for (unsigned k = tid; k < URF_TILE_WAVES * (C + Ks) / 2; k += URF_TILE_THREADS)
        ((unsigned*)wcnt_r)[k] = 0;
    if (tid < 32)
        misc[tid] = 0;
    if (tid < C)
        tmax[tid] = 0;
    if (tid < URF_MAX_CHANNELS) {
        tab[tid] = tab_v;
        ((float4*)thr)[tid] = thr_v;
        ul[tid] = thr_v.x;
    }
#pragma unroll
    for (unsigned e = 0; e < LUT_PT; e++) {
        const unsigned i = tid + e * URF_TILE_THREADS;
        if (i < LUT_WORDS)
            ((unsigned*)lut)[i] = lut_v[e];
    }
    __syncthreads();
    URF_PHASE_MARK;
#ifdef URF_EXP_PHASE_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* x / y / z have arrived */
    URF_PHASE_MARK;
#endif

    const float interval = dp.p.interval;
    /* Ring, float fast path (urf_device.hpp: urf_fast_cot, urf_ring_thresholds) unless the stage
     * capture wants the exact angle: u = cot(vertical angle) is compared with the thresholds
     * k_ring_table derived from every entry's window -- no arc tangent, no square root.
     *
     * Main pass: only what the approximations decide, and without per-lane branches (every test is
     * a select: the wave executes both sides of a divergent branch anyway, and the exec-mask
     * bookkeeping of the branches cost more scalar instructions than the tests cost vector ones).
     * A point whose ring or sector the approximations leave open (or every point, when the stage
     * capture wants exact angles) is listed and takes the reference's exact sequence in a second,
     * dense pass: the exact code exists once instead of four times in the unrolled loop, and its
     * f64 chains never run with two lanes of a wave. */
    const bool exact_all = a.capture == 1;
    unsigned rkey[Q], skey[Q];
    unsigned azc[Q];         /* approximate azimuth (urf_device.hpp) as the code of the slot record (URF_REC_*), consumed by k_label */
    unsigned openmask = 0;   /* bit q: point q of this thread is on the pending list */
    /* Written phase by phase over the thread's four points, so that the four dependent LDS reads of
     * the ring search (lookup cell, two probes, the entry's thresholds) are in flight for all four
     * points at once: point after point the wave sat through sixteen LDS round trips here. */
    float uu[Q];
    unsigned lo[Q], roim = 0, fastm = 0, openm = 0;
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const unsigned i = tbase + wave * URF_WAVE_PTS + q * 64 + lane;
        roim |= (unsigned)((i < len) & urf_in_roi(dp.p, px[q], py[q], pz[q])) << q;
    }
    /* A wave none of whose 256 points lies in the region of interest has nothing to classify (uniform
     * branch): the reference's default region drops whole azimuth ranges of a sweep, i.e. whole tiles in
     * firing order (cfg/LidarFilters.cfg:42-51, lidar_segmentation.cpp:100-117). */
    const bool wave_on = __ballot(roim != 0u) != 0ull;
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        rkey[q] = URF_RING_NONE;
        skey[q] = URF_SEC_NONE;
        azc[q] = URF_REC_AZ_UNKNOWN;
    }
    if (wave_on) {
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const float x = px[q], y = py[q], z = pz[q];
        const bool roi = (roim >> q) & 1u;
        float u;
        bool planar;
        const bool fast = urf_fast_cot(x, y, z, &u, &planar) & roi & !exact_all;
        uu[q] = fast ? u : 0.f;
        fastm |= (unsigned)fast << q;
    }
    /* THE EXPECTED RING FIRST.  In firing order point li of the tile belongs to ring li mod C (C a power of two: what an
     * organised tile is made of, see below).  "u lies surely inside the window of entry e, and surely above the window of
     * entry e - 1" settles ring e -- the entries in front of e - 1 lie lower still -- with two LDS reads and four
     * comparisons instead of the lookup cell, its two probes and the five comparisons of the search.  A wave in which some
     * point of the fast path does not pass (another sensor layout, a cut firing, rings closer together than `interval`)
     * takes the search, for all its points. */
    bool searched = true;
#ifndef URF_EXP_NO_EXPECTED
    if ((C & (C - 1u)) == 0u) {   /* (uniform) */
        unsigned okm = 0;
#pragma unroll
        for (unsigned q = 0; q < Q; q++) {
            const unsigned e = (wave * URF_WAVE_PTS + q * 64 + lane) & (C - 1u);
            const urf_ring_thr tv = thr[e];
            const float below = ul[(e - 1u) & (URF_MAX_CHANNELS - 1)];   /* (entry e - 1 lies surely below the window of every u < its .x) */
            const float u = uu[q];
            okm |= (unsigned)((e < nR) & (u >= tv.y) & (u <= tv.z) & ((e == 0u) | (u < below))) << q;
        }
        if (__ballot((fastm & ~okm) != 0u) == 0ull) {   /* (uniform) every point of the fast path sits on its expected ring */
            searched = false;
#pragma unroll
            for (unsigned q = 0; q < Q; q++) {
                const bool fast = (fastm >> q) & 1u;
                rkey[q] = fast ? ((wave * URF_WAVE_PTS + q * 64 + lane) & (C - 1u)) : URF_RING_NONE;
            }
            openm |= roim & ~fastm;
        }
    }
#endif
    if (searched) {
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        /* lo = number of table entries surely below the point's window: the cell's count from the
         * lookup table, plus up to two entries between the cell's end and u (a third one is rare
         * and left to the exact pass) */
        const unsigned cell = (unsigned)((uu[q] + URF_LUT_UMAX) * URF_LUT_SCALE);   /* u in [-4, 4]: cell <= 4096 */
        lo[q] = lut[cell];
    }
#pragma unroll
    for (unsigned q = 0; q < Q; q++)
        lo[q] += (lo[q] < nR) & (uu[q] < ul[lo[q] & (URF_MAX_CHANNELS - 1)]);
#pragma unroll
    for (unsigned q = 0; q < Q; q++)
        lo[q] += (lo[q] < nR) & (uu[q] < ul[lo[q] & (URF_MAX_CHANNELS - 1)]);
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const float u = uu[q];
        const urf_ring_thr tv = thr[lo[q] & (URF_MAX_CHANNELS - 1)];
        const bool unsettled = (lo[q] < nR) & (u < tv.x);
        const bool none = (lo[q] >= nR) | (u > tv.w);              /* no entry can match */
        const bool match = !none & (u >= tv.y) & (u <= tv.z);      /* the first candidate surely matches */
        const bool roi = (roim >> q) & 1u, fast = (fastm >> q) & 1u;
        openm |= (unsigned)(roi & (!fast | unsettled | !(none | match))) << q;
        rkey[q] = match ? lo[q] : URF_RING_NONE;
    }
    }
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const unsigned li = wave * URF_WAVE_PTS + q * 64 + lane, i = tbase + li;
        const float x = px[q], y = py[q];
        const bool roi = (roim >> q) & 1u;
        bool open = (openm >> q) & 1u;
        const unsigned rk = rkey[q];
        unsigned sk = URF_SEC_NONE;
        const float fi = urf_fast_polar(x, y);   /* one arc tangent: the star sector and the azimuth */
        azc[q] = urf_fast_az_ok(x, y) ? urf_az_code(urf_fast_azimuth_of(fi)) : URF_REC_AZ_UNKNOWN;   /* (too close to the x axis: exact on demand) */
        if (star) {
            /* (decided on the approximation only where the ring was: magnitudes checked there; the few
             * points steeper than |z| = 4 rho take the exact sequence for both) */
            const int fs = ((fastm >> q) & 1u) ? urf_fast_sector_ranged(fi, dp.Kfi, K, dp.sector_margin) : -1;
            open = open | (roi & (fs < 0));
            sk = (unsigned)fs;
            if (dp.p.starbeam_filter && !urf_in_beam(a.beams[fs < 0 ? 0 : fs], x, y))
                sk = URF_SEC_NONE;
        }
        const bool settled = roi & !open;
        rkey[q] = settled ? rk : URF_RING_NONE;
        skey[q] = settled ? sk : URF_SEC_NONE;
        if (settled & (rk == URF_RING_NONE) & (i >= upto))
            a.table_redo[s] = 1u;   /* (rare; every writer writes the same value) */
        if (open) {
            pending[atomicAdd(&misc[1], 1u)] = (uint16_t)li;
            openmask |= 1u << q;
        }
    }
    }
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const unsigned li = wave * URF_WAVE_PTS + q * 64 + lane, i = tbase + li;
        const bool roi = (roim >> q) & 1u;
        if (i < len && exact_all && !roi)
            a.valpha[sb + i] = -1.0f;   /* stage capture only */
        /* the label bytes are k_label's: it gets the region-of-interest bits of the tile, 64 per word */
        const unsigned long long rb = __ballot(roi);
        if (lane == 0) {
            a.roi_bits[((size_t)s * a.tiles + t) * (URF_TILE / 64) + (li >> 6)] = rb;
            if (rb)
                atomicAdd(&misc[0], (unsigned)__popcll(rb));
        }
    }
    __syncthreads();
    URF_PHASE_MARK;
    if (misc[0] == 0) {
        /* no point of the tile lies in the region of interest (uniform): empty run tables, and the
         * remaining six phases (exact pass, ranks, scans, transposes) have nothing to do */
        const size_t row0 = (size_t)s * a.tiles + t;
        for (unsigned k = tid; k <= C; k += URF_TILE_THREADS)
            a.troff[row0 * (C + 1) + k] = 0;
        for (unsigned k = tid; k < C; k += URF_TILE_THREADS)
            a.tmaxs[row0 * C + k] = 0ull;
        if (star)
            for (unsigned k = tid; k <= K; k += URF_TILE_THREADS)
                a.tsoff[row0 * (K + 1) + k] = 0;
        if (tid == 0)
            a.tile_roi[row0] = 0;
        if (a.capture)   /* stage capture only */
#pragma unroll
            for (unsigned q = 0; q < Q; q++) {
                const unsigned i = tbase + wave * URF_WAVE_PTS + q * 64 + lane;
                if (i < len) {
                    a.ringkey[sb + i] = (uint8_t)URF_RING_NONE;
                    a.seckey[sb + i] = (uint16_t)URF_SEC_NONE;
                }
            }
        return;
    }
    const unsigned np = misc[1];
    for (unsigned k = tid; k < np; k += URF_TILE_THREADS) {
        const unsigned li = pending[k], i = tbase + li;
        const float x = gx[li], y = gy[li], z = gz[li];
        const urf_exact_key ek = urf_exact_keys(tab, nR, interval, x, y, z, star ? K : 0u, dp.Kfi);
        unsigned sk = ek.sector;
        if (star && dp.p.starbeam_filter && !urf_in_beam(a.beams[sk], x, y))
            sk = URF_SEC_NONE;
        if (exact_all)
            a.valpha[sb + i] = ek.valpha;   /* stage capture only */
        if (ek.ring == URF_RING_NONE && i >= upto)
            a.table_redo[s] = 1u;
        /* a ring point straight above or below the sensor: its azimuth is NaN, and the reference's per-ring quicksort and
         * beam scans treat the ring in a way of their own (k_nan_rings).  Such a point always gets here (urf_fast_cot
         * refuses x == y == 0), so the test costs the main pass nothing. */
        if (x == 0.0f && y == 0.0f && ek.ring != URF_RING_NONE) {
            const unsigned bit = 1u << (ek.ring & 31u);
            const unsigned old = atomicOr(&a.nan_mask[(size_t)s * 4 + (ek.ring >> 5)], bit);
            if (!(old & bit))
                a.nan_list[atomicAdd(&a.star_count[3], 1u)] = s * C + ek.ring;
        }
        keyr[li] = (uint8_t)ek.ring;
        keys[li] = (uint16_t)sk;
    }
    __syncthreads();
    URF_PHASE_MARK;

    /* the points the exact pass decided; stage capture */
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const unsigned li = wave * URF_WAVE_PTS + q * 64 + lane, i = tbase + li;
        if (openmask & (1u << q)) {
            rkey[q] = (unsigned)keyr[li];
            skey[q] = star ? (unsigned)keys[li] : URF_SEC_NONE;
        }
        if (a.capture && i < len) {   /* stage capture only */
            a.ringkey[sb + i] = (uint8_t)rkey[q];
            a.seckey[sb + i] = (uint16_t)skey[q];
        }
    }
    /* ORGANISED TILE (every benchmark sweep, every spinning LiDAR that reports in firing order): point li of the tile
     * lies on ring li mod C (C a power of two: a firing holds every ring once, in order) and -- where the star-shaped
     * search runs -- every 64-point step shares ONE sector, the steps' sectors not falling along the tile.  Then both
     * sorted orders are closed-form functions of the input order:
     *     ring-sorted slot   = (li mod C) * (2048 / C) + li / C         (a 64 x 32 transpose for C = 64)
     *     sector-sorted slot = li                                       (the stable split by sector is the identity)
     * and the whole ranking machinery below -- match_any per step, per-wave counters, the scans over waves and
     * keys, three barriers -- has nothing to compute.  Decided per tile from the keys themselves (one ballot per step
     * and family, the step keys compared across the tile), so any other tile simply takes the general path. */
    unsigned mode = 0;   /* 0: the general path, 1: organised tile, 2: organised with holes */
    unsigned hlp[Q] = { 0, 0, 0, 0 }, hsp[Q] = { 0, 0, 0, 0 };
#ifndef URF_EXP_NO_ORGANISED
    {
        const bool shape = (C & (C - 1u)) == 0u && tbase + URF_TILE <= len;   /* (uniform) */
        bool mine = shape;
#pragma unroll
        for (unsigned q = 0; q < Q; q++) {
            const unsigned li = wave * URF_WAVE_PTS + q * 64 + lane;
            mine = mine & (rkey[q] == (li & (C - 1u)));
            if (star) {
                const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane((int)skey[q]);
                mine = mine & (skey[q] == f) & (f != URF_SEC_NONE);
                if (lane == 0)
                    stepkey[wave * Q + q] = f;
            }
        }
        if (__ballot(!mine) != 0ull && lane == 0)
            misc[31] = 1u;   /* (every writer writes the same value) */
        __syncthreads();
        static_assert(URF_TILE_GROUPS == 32, "one lane per step of the tile compares it with the next");
        const bool falls = star && lane < URF_TILE_GROUPS - 1u && stepkey[lane] > stepkey[lane + 1];
        mode = (misc[31] == 0u && __ballot(falls) == 0ull) ? 1u : 0u;   /* (uniform over the workgroup) */
#ifndef URF_EXP_NO_HOLEY
        /* (a second look only at a tile that failed the first: the fully organised tile pays nothing for it, the tile with
         * holes one barrier and a call) */
        if (mode == 0u && shape && C == 64u) {   /* (uniform) */
            const urf_holey_slots hs = urf_split_holey(rkey[0], rkey[1], rkey[2], rkey[3], skey[0], skey[1], skey[2], skey[3], wave, lane, star,
                                                       misc, koff);
            if (hs.ok) {
                mode = 2u;
#pragma unroll
                for (unsigned q = 0; q < Q; q++) {
                    hlp[q] = hs.lp[q];
                    hsp[q] = hs.sp[q];
                }
            }
        }
#endif
    }
#endif
    unsigned lp[Q], sp[Q];
    const unsigned logC = 31u - (unsigned)__clz((int)C);
    if (mode == 1u) {
        const unsigned P = URF_TILE >> logC;
        /* (the slots are computed where they are used, below; so is the rings' largest range) */
#pragma unroll
        for (unsigned q = 0; q < Q; q++)
            lp[q] = sp[q] = 0;
        /* the run tables: ring c starts at slot c * P; sector k at the first step whose sector is >= k (the step keys
         * do not fall: bisection over the 32 of them) */
        for (unsigned k = tid; k <= C; k += URF_TILE_THREADS)
            koff[k] = k * P;
        if (star)
            for (unsigned k = tid; k <= K; k += URF_TILE_THREADS) {
                unsigned lo = 0;   /* number of steps with a sector < k */
#pragma unroll
                for (unsigned step = URF_TILE_GROUPS / 2; step > 0; step >>= 1)
                    lo += stepkey[lo + step - 1] < k ? step : 0u;
                lo += (lo == URF_TILE_GROUPS - 1u && stepkey[lo] < k) ? 1u : 0u;
                soff[k] = lo * 64u;
            }
    } else if (mode == 2u) {
#pragma unroll
        for (unsigned q = 0; q < Q; q++) {
            lp[q] = hlp[q];
            sp[q] = hsp[q];
        }
    } else {
    /* step 1: ranks inside the wave's own 256 points.  The lanes of a step that share a key read
     * the key's running count (one LDS address: a broadcast), the first of them adds the group's
     * size; a wave touches only its own row, in program order. */
    unsigned rrank[Q], srank[Q];
    uint16_t* my_r = wcnt_r + wave * C;
    uint16_t* my_s = wcnt_s + wave * K;
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        {
            const bool on = rkey[q] != URF_RING_NONE;
            const unsigned long long m = urf_match_any_on(rkey[q], on, dp.ring_keybits);
            const unsigned old = my_r[on ? rkey[q] : 0];
            if (on && urf_is_leader(m))
                my_r[rkey[q]] = (uint16_t)(old + (unsigned)__popcll(m));
            rrank[q] = old + urf_popc_below(m);
            /* maxDistance (lidar_segmentation.cpp:271-274) is the largest float(sqrt(double s)), s = x^2 + y^2,
             * of a ring: both roundings are monotone, so the largest s is tracked -- here, where x and y
             * are at hand, per tile and ring; k_ring takes the maximum over the tiles and streams only z */
            if (on) {
                const double s2 = (double)px[q] * (double)px[q] + (double)py[q] * (double)py[q];
                atomicMax(&tmax[rkey[q]], (unsigned long long)__double_as_longlong(s2));
            }
        }
        srank[q] = 0;
        if (star) {
            const bool on = skey[q] != URF_SEC_NONE;
            const unsigned long long m = urf_match_any_on(skey[q], on, dp.sec_keybits);
            const unsigned old = my_s[on ? skey[q] : 0];
            if (on && urf_is_leader(m))
                my_s[skey[q]] = (uint16_t)(old + (unsigned)__popcll(m));
            srank[q] = old + urf_popc_below(m);
        }
    }
    __syncthreads();
    URF_PHASE_MARK;
    /* steps 2 + 3: per key the exclusive scan of its counts over the waves (four at a time read before they are
     * written back), then -- by the same thread, no barrier in between -- the scan across the keys: first
     * slot of every ring (C <= 128: the last wave, two keys per lane) and of every sector (K <= 1022: two
     * keys per thread) inside the tile */
    auto column = [&](uint16_t* col, unsigned stride) -> unsigned {
        unsigned run = 0;
#pragma unroll
        for (unsigned w0 = 0; w0 < URF_TILE_WAVES; w0 += 4) {   /* four at a time: the kernel has no registers to spare */
            unsigned c[4];
#pragma unroll
            for (unsigned w = 0; w < 4; w++)
                c[w] = col[(w0 + w) * stride];
#pragma unroll
            for (unsigned w = 0; w < 4; w++) {
                col[(w0 + w) * stride] = (uint16_t)run;
                run += c[w];
            }
        }
        return run;
    };
    unsigned sv0 = 0, sv1 = 0, sinc = 0;
    if (star) {
        sv0 = 2 * tid < K ? column(wcnt_s + 2 * tid, K) : 0;
        sv1 = 2 * tid + 1 < K ? column(wcnt_s + 2 * tid + 1, K) : 0;
        sinc = urf_wave_scan_add(sv0 + sv1);
        if (lane == 63)
            misc[2 + wave] = sinc;
    }
    if (wave == URF_TILE_WAVES - 1) {
        const unsigned v0 = lane < C ? column(wcnt_r + lane, C) : 0, v1 = lane + 64 < C ? column(wcnt_r + lane + 64, C) : 0;
        const unsigned i0 = urf_wave_scan_add(v0), i1 = urf_wave_scan_add(v1);
        const unsigned total0 = __shfl(i0, 63), total1 = __shfl(i1, 63);
        if (lane < C)
            koff[lane] = i0 - v0;
        if (lane + 64 < C)
            koff[lane + 64] = total0 + i1 - v1;
        if (lane == 0)
            koff[C] = total0 + total1;
    }
    __syncthreads();
    URF_PHASE_MARK;
    if (star) {
        /* the waves' totals: lane l reads wave l's, one scan, two broadcasts (every thread summing all of them took
         * as many registers as there are waves) */
        const unsigned mine = lane < URF_TILE_WAVES ? misc[2 + lane] : 0u;
        const unsigned minc = urf_wave_scan_add(mine);
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)minc, URF_TILE_WAVES - 1);
        const unsigned wb = (unsigned)__shfl((int)(minc - mine), (int)wave);
        const unsigned run = wb + sinc - (sv0 + sv1);
        if (2 * tid < K)
            soff[2 * tid] = run;
        if (2 * tid + 1 < K)
            soff[2 * tid + 1] = run + sv0;
        if (tid == 0)
            soff[K] = total;
    }
    __syncthreads();
    URF_PHASE_MARK;

    /* step 4: slot in the tile's ring-sorted order (lp) and sector-sorted order (sp) */
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {   /* unconditional reads (entry 0 for "none"), then a select */
        const bool ron = rkey[q] != URF_RING_NONE, son = skey[q] != URF_SEC_NONE;
        const unsigned rk = ron ? rkey[q] : 0, sk = son ? skey[q] : 0;
        const unsigned lr = koff[rk] + my_r[rk] + rrank[q];
        lp[q] = ron ? lr : 0xffffffffu;
        sp[q] = 0xffffffffu;
        if (star) {
            const unsigned ls = soff[sk] + my_s[sk] + srank[q];
            sp[q] = son ? ls : 0xffffffffu;
        }
    }
    }   /* (general path) */
    __syncthreads();   /* keys and wcnt are dead: their memory becomes the staging buffers */
    URF_PHASE_MARK;
    if (mode == 2u && star)   /* (uniform) sector k starts with the first step whose key is >= k: its points in front of it */
        for (unsigned k = tid; k <= K; k += URF_TILE_THREADS) {
            unsigned lo = 0;   /* number of steps whose (filled-in) key + 1 is < k + 1 */
#pragma unroll
            for (unsigned step = URF_TILE_GROUPS / 2; step > 0; step >>= 1)
                lo += stepfk[lo + step - 1] < k + 1u ? step : 0u;
            lo += (lo == URF_TILE_GROUPS - 1u && stepfk[lo] < k + 1u) ? 1u : 0u;
            soff[k] = stepbase[lo];
        }
    unsigned* stx = (unsigned*)un;
    unsigned* sty = stx + URF_SLOTS;
    unsigned* stz = sty + URF_SLOTS;
    unsigned* str = stz + URF_SLOTS;
    const unsigned tb = sb + tbase;
    float* __restrict__ const o_sr = a.sr + tb;
    float* __restrict__ const o_sz = a.sz + tb;
    uint16_t* __restrict__ const o_ss = a.sslot + tb;
#pragma unroll
    for (unsigned q = 0; q < Q; q++) {
        const unsigned li = wave * URF_WAVE_PTS + q * 64 + lane;
        const float x = px[q], y = py[q], z = pz[q];
        if (mode == 1u) {   /* (uniform) closed-form slots; maxDistance as on the general path (lidar_segmentation.cpp:271-274) */
            lp[q] = (li & (C - 1u)) * (URF_TILE >> logC) + (li >> logC);
            sp[q] = star ? li : 0xffffffffu;
            const double s2 = (double)x * (double)x + (double)y * (double)y;
            atomicMax(&tmax[li & (C - 1u)], (unsigned long long)__double_as_longlong(s2));
        }
        if (mode == 2u && lp[q] != 0xffffffffu) {   /* (uniform) maxDistance as on the other paths; C == 64: the ring is the lane */
            const double s2 = (double)x * (double)x + (double)y * (double)y;
            atomicMax(&tmax[lane], (unsigned long long)__double_as_longlong(s2));
        }
        if (lp[q] != 0xffffffffu) {
            const unsigned sl = URF_SLOT(lp[q]);
            stx[sl] = __float_as_uint(x);
            sty[sl] = __float_as_uint(y);
            stz[sl] = __float_as_uint(z);
            str[sl] = (azc[q] << URF_REC_AZ_SHIFT) | li;   /* the slot's record: no detector hit so far */
        }
        if (sp[q] != 0xffffffffu) {
            const unsigned so = sp[q] & (URF_TILE - 1u);   /* (a slot inside the tile) */
            __builtin_nontemporal_store(__builtin_sqrtf(x * x + y * y), &o_sr[so]);   /* star_shaped_search.cpp:164 */
            __builtin_nontemporal_store(z, &o_sz[so]);
            /* where a star-shaped hit on this point has to be reported: its ring-sorted slot (none
             * if the point lies on no ring: such a hit ends the walk but marks nothing that
             * reaches the output, lidar_segmentation.cpp:235-242) */
            __builtin_nontemporal_store((uint16_t)(lp[q] != 0xffffffffu ? lp[q] : URF_SLOT_NONE), &o_ss[so]);
        }
    }
    __syncthreads();
    URF_PHASE_MARK;
    const unsigned tile_ring_pts = koff[C];
    {
        float* __restrict__ const o_rx = a.rx + tb;
        float* __restrict__ const o_ry = a.ry + tb;
        float* __restrict__ const o_rz = a.rz + tb;
        uint32_t* __restrict__ const o_rec = a.rec + tb;
        for (unsigned j = tid; j < tile_ring_pts; j += URF_TILE_THREADS) {
            const unsigned sl = URF_SLOT(j);
            __builtin_nontemporal_store(__uint_as_float(stx[sl]), &o_rx[j]);
            __builtin_nontemporal_store(__uint_as_float(sty[sl]), &o_ry[j]);
            __builtin_nontemporal_store(__uint_as_float(stz[sl]), &o_rz[j]);
            __builtin_nontemporal_store(str[sl], &o_rec[j]);
        }
    }
    const size_t row = (size_t)s * a.tiles + t;
    for (unsigned k = tid; k <= C; k += URF_TILE_THREADS)
        a.troff[row * (C + 1) + k] = (uint16_t)koff[k];
    for (unsigned k = tid; k < C; k += URF_TILE_THREADS)
        a.tmaxs[row * C + k] = tmax[k];
    if (star)
        for (unsigned k = tid; k <= K; k += URF_TILE_THREADS)
            a.tsoff[row * (K + 1) + k] = (uint16_t)soff[k];
    if (tid == 0)
        a.tile_roi[row] = misc[0];
    URF_PHASE_MARK;
    URF_PHASE_DUMP("k_split");
}

__global__ __launch_bounds__(URF_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(URF_SPLIT_WAVES_PER_EU, URF_SPLIT_WAVES_PER_EU))) void k_split(urf_kargs a, urf_dev_params dp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_split[];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *a.ring_hint = 0;   /* k_ring_table has read the previous call's ring count; k_index collects this call's */
    if (a.front && __builtin_amdgcn_readfirstlane((int)((const uint32_t* __restrict__)a.front_ok)[blockIdx.y]))
        return;   /* (uniform) a scan of the fused front end (urf_front.hpp) */
    urf_split_tile(a, dp, blockIdx.y, blockIdx.x, sh_split, threadIdx.x);
}

/* the scans k_table_repair listed (normally none): their tiles once more, with the complete table.
 * Persistent workgroups over list x tiles. */
__global__ __launch_bounds__(URF_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(URF_SPLIT_WAVES_PER_EU, URF_SPLIT_WAVES_PER_EU))) void k_split_repair(urf_kargs a, urf_dev_params dp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_split[];
    const unsigned n = a.star_count[2];
    for (unsigned w = blockIdx.x; w < n * a.tiles; w += gridDim.x) {
        unsigned tid_i = threadIdx.x;   /* opaque per iteration: nothing thread-derived is hoisted out of the loop (and spilled) */
        asm volatile("" : "+v"(tid_i));
        urf_split_tile(a, dp, a.redo_list[w / a.tiles], w % a.tiles, sh_split, tid_i);
        __syncthreads();   /* the LDS carve is reused by the next tile */
    }
}

/* the scans the fused front end handed back (urf_front.hpp: front_list; normally none), and -- that list holds them too -- the
 * scans whose speculative ring table k_table_repair has rebuilt: split the legacy way */
__global__ __launch_bounds__(URF_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(URF_SPLIT_WAVES_PER_EU, URF_SPLIT_WAVES_PER_EU))) void k_split_list(urf_kargs a, urf_dev_params dp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_split[];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *a.ring_hint = 0;   /* (k_split's duty) */
    const unsigned n = a.star_count[6];
    for (unsigned w = blockIdx.x; w < n * a.tiles; w += gridDim.x) {
        unsigned tid_i = threadIdx.x;
        asm volatile("" : "+v"(tid_i));
        urf_split_tile(a, dp, a.front_list[w / a.tiles], w % a.tiles, sh_split, tid_i);
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------- */
/* k_index                                                                     */
/* ------------------------------------------------------------------------- */
/* exclusive scan of cnt[0..K) (K <= 1024) by 256 threads -> offs[0..K] */
__device__ void urf_scan_keys_256(const unsigned* cnt, unsigned* offs, unsigned K, unsigned* sh /* [8] */)
{
    const unsigned tid = threadIdx.x;
    unsigned v[4], sum = 0;
    for (int e = 0; e < 4; e++) {
        const unsigned k = tid * 4 + e;
        v[e] = k < K ? cnt[k] : 0;
        sum += v[e];
    }
    const unsigned inc = urf_wave_scan_add(sum);
    __syncthreads();
    if (urf_lane() == 63)
        sh[tid >> 6] = inc;
    __syncthreads();
    unsigned wbase = 0;
    for (unsigned w = 0; w < (tid >> 6); w++)
        wbase += sh[w];
    unsigned run = wbase + inc - sum;
    for (int e = 0; e < 4; e++) {
        const unsigned k = tid * 4 + e;
        if (k < K)
            offs[k] = run;
        run += v[e];
        if (k + 1 == K)
            offs[K] = run;
    }
    __syncthreads();
}

/* One family of keys (rings or sectors) of one scan: turns k_split's per-tile run tables
 * toff[tile][key] (first slot of the key's run inside the tile, row-major, rows of nkeys + 1 u16)
 * into per-key tables: pre[key][tile] = points of the key in the tiles before (u32, [ntiles] =
 * total), start[key][tile] = toff[tile][key], and the totals cnt[key], for the 64 keys from k0.  Blocks of 64 keys x 64 tiles are transposed through LDS so that both the
 * reads (rows of toff) and the writes (rows of pre / start) are contiguous. */
struct urf_index_shared {
    uint16_t cnt[64][66];
    uint16_t st[64][66];
    unsigned carry[64];
};
__device__ void urf_index_family(urf_index_shared& L, const uint16_t* toff, unsigned nkeys, unsigned k0, unsigned ntiles,
                                 unsigned tstride, unsigned* pre, uint16_t* start, unsigned* cnt)
{
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned rowlen = nkeys + 1;
    if (k0 < nkeys) {   /* the block of keys [k0, k0 + 64) */
        const unsigned key = k0 + lane;
        if (tid < 64)
            L.carry[tid] = 0;
        for (unsigned t0 = 0; t0 < ntiles; t0 += 64) {
            /* rows of toff -> counts and starts (wave w: tiles t0 + 4 * pass + w; lane = key).  All 32 loads of a
             * thread are in flight at once: unconditional, from clamped addresses (behind a condition the compiler
             * waits for every single one). */
            unsigned fv0[16], fv1[16];
#pragma unroll
            for (unsigned pass = 0; pass < 16; pass++) {
                const unsigned tt = t0 + pass * 4 + wave;
                const bool in = tt < ntiles && key < nkeys;
                const unsigned idx = in ? tt * rowlen + key : 0u;
                fv0[pass] = toff[idx];
                fv1[pass] = toff[idx + 1];
                fv0[pass] = in ? fv0[pass] : 0u;
                fv1[pass] = in ? fv1[pass] : 0u;
            }
#pragma unroll
            for (unsigned pass = 0; pass < 16; pass++) {
                const unsigned u = pass * 4 + wave;
                L.cnt[lane][u] = (uint16_t)(fv1[pass] - fv0[pass]);
                L.st[lane][u] = (uint16_t)fv0[pass];
            }
            __syncthreads();
            /* prefix along the tiles and the rows of pre / start (wave w: keys k0 + 4 * pass + w; lane = tile): one
             * wave scan per key (r2 / r3: 64 serial steps of one wave through LDS, 8 000 of the kernel's 48 000
             * cycles on a single sweep).  A wave touches only its own keys' carries. */
#pragma unroll 4
            for (unsigned pass = 0; pass < 16; pass++) {
                const unsigned r = pass * 4 + wave, kk = k0 + r, tt = t0 + lane;
                const unsigned c = L.cnt[r][lane];
                const unsigned incl = urf_wave_scan_add(c);
                const unsigned base = L.carry[r];
                if (kk < nkeys && tt < ntiles) {
                    pre[(size_t)kk * (tstride + 1) + tt] = base + incl - c;
                    start[(size_t)kk * tstride + tt] = L.st[r][lane];
                }
                if (lane == 63)
                    L.carry[r] = base + incl;
            }
            __syncthreads();
        }
        if (tid < 64 && key < nkeys) {
            pre[(size_t)key * (tstride + 1) + ntiles] = L.carry[tid];
            cnt[key] = L.carry[tid];
        }
        __syncthreads();
    }
}

/* piece = number of ROI points of the scan (lidar_segmentation.cpp:120), summed by all 256 threads */
__device__ __forceinline__ unsigned urf_scan_piece(const urf_kargs& a, unsigned s, unsigned ntiles, unsigned* sh /* [4] */)
{
    const unsigned tid = threadIdx.x;
    unsigned piece = 0;
    for (unsigned t = tid; t < ntiles; t += 256)
        piece += a.tile_roi[(size_t)s * a.tiles + t];
    for (int o = 32; o > 0; o >>= 1)
        piece += __shfl_xor(piece, o);
    if (urf_lane() == 0)
        sh[tid >> 6] = piece;
    __syncthreads();
    piece = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return piece;
}

/* size and first two runs of NK sectors per thread (k, k + 256, ...), 16 tiles of each per round: NK x 32 loads in
 * flight */
template <unsigned NK>
__device__ __forceinline__ void urf_index_sectors(const urf_kargs& a, unsigned s, unsigned K, unsigned ntiles)
{
    const unsigned tid = threadIdx.x;
    const uint16_t* toff = a.tsoff + (size_t)s * a.tiles * (K + 1);
    for (unsigned kp = 0; kp < K; kp += 256 * NK) {
        unsigned run[NK], mx[NK];
        urf_sec_run sr[NK];
#pragma unroll
        for (unsigned h = 0; h < NK; h++) {
            run[h] = 0;
            mx[h] = 0;
            sr[h] = urf_sec_run{ 0u, 0u, 0u, 0u };
        }
        for (unsigned t0 = 0; t0 < ntiles; t0 += 16) {
            unsigned v0[NK][16], v1[NK][16];
#pragma unroll
            for (unsigned h = 0; h < NK; h++)
#pragma unroll
                for (unsigned u = 0; u < 16; u++) {
                    /* (unconditional loads from clamped addresses: behind a condition the compiler waits for every
                     * single one) */
                    const unsigned k = kp + h * 256 + tid;
                    const bool in = t0 + u < ntiles && k < K;
                    const unsigned idx = in ? (t0 + u) * (K + 1) + k : 0u;
                    v0[h][u] = (unsigned)toff[idx];
                    v1[h][u] = (unsigned)toff[idx + 1];
                    v0[h][u] = in ? v0[h][u] : 0u;
                    v1[h][u] = in ? v1[h][u] : 0u;
                }
#pragma unroll
            for (unsigned h = 0; h < NK; h++)
#pragma unroll
                for (unsigned u = 0; u < 16; u++) {
                    /* (selects: as branches these 64 steps per key were 12 000 of the kernel's 48 000 cycles on a
                     * single sweep) */
                    const unsigned c = v1[h][u] - v0[h][u], ad = (t0 + u) * URF_TILE + v0[h][u];
                    const bool first = (c != 0u) & (sr[h].nruns == 0u), second = (c != 0u) & (sr[h].nruns == 1u);
                    sr[h].a0 = first ? ad : sr[h].a0;
                    sr[h].c0 = first ? c : sr[h].c0;
                    sr[h].a1 = second ? ad : sr[h].a1;
                    sr[h].nruns += c != 0u;
                    run[h] += c;
                    mx[h] = c > mx[h] ? c : mx[h];
                }
        }
#pragma unroll
        for (unsigned h = 0; h < NK; h++) {
            const unsigned k = kp + h * 256 + tid;
            if (k < K) {
                /* many short runs, one per tile of at most 64: k_star_sort_runs (one lane per run) instead of the workgroup kernel */
                if (sr[h].nruns > 2u && ntiles <= 64u && mx[h] <= URF_STAR_SMALL_CAP / 64u)
                    sr[h].nruns |= URF_RUNS_FLAG;
                a.sec_cnt[(size_t)s * K + k] = run[h];
                a.sec_run[(size_t)s * K + k] = sr[h];
            }
        }
    }
}

/* One workgroup per scan: piece < 30 test (lidar_segmentation.cpp:120-126); the per-ring run tables
 * (urf_index_family) and where every ring starts in the ring-major arrays; the size of every sector
 * (the sort kernels read a sector's runs straight from k_split's per-tile tables: a sector meets
 * only a few tiles) and where it starts in the sector-major arrays; the work lists of the oversized
 * sectors. */
__device__ __forceinline__ void urf_index_body(const urf_kargs& a, const urf_dev_params& dp, urf_index_shared& L, unsigned* sh)
{
    const unsigned s = blockIdx.x, tid = threadIdx.x;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    if ((a.optimistic & URF_OPT_NO_REPAIR) && a.table_redo[s]) {
        /* the speculative ring table was incomplete and nothing has repaired it (callback path): the scan is void,
         * every later kernel skips it, the host runs it again without the speculation */
        if (tid == 0)
            a.info[s].status = a.table_cause[s] == 2u ? URF_STATUS_REDO_HINT : URF_STATUS_REDO_TABLE;
        return;
    }
    if (a.optimistic & URF_OPT_NO_NAN) {
        /* a ring holds a point with a NaN azimuth and nothing will run the reference's quicksort for it (callback path):
         * void, run again with the full sequence */
        const uint4 nm = *(const uint4*)(a.nan_mask + (size_t)s * 4);
        if (nm.x | nm.y | nm.z | nm.w) {   /* (uniform) */
            if (tid == 0)
                a.info[s].status = URF_STATUS_REDO_NAN;
            return;
        }
    }
    {
        const unsigned piece = urf_scan_piece(a, s, ntiles, sh);
        if (tid == 0) {
            a.info[s].n_roi = piece;
            if (piece < 30) {
                a.info[s].status = URF_TOO_FEW_POINTS;
                a.info[s].n_rings = 0;
            }
        }
        if (piece < 30)
            return;
    }
    if (dp.p.star_shaped_method) {
        /* sector sizes: column sums of the per-tile tables (rows are read contiguously: thread = key,
         * 16 tiles in flight.  Two keys per thread and twice the loads in flight bought a single sweep nothing:
         * 0.0180 vs 0.0183 ms) */
        urf_index_sectors<1u>(a, s, K, ntiles);
    }
    const bool fused = a.front && a.front_ok[s];   /* (uniform) urf_front.hpp: no ring-sorted copies, k_front_finish counts the rings' points */
    if (fused && tid == 0)
        atomicMax(a.ring_hint, a.info[s].n_rings);
    if (!fused) {
    for (unsigned k0 = 0; k0 < C; k0 += 64)
        urf_index_family(L, a.troff + (size_t)s * a.tiles * (C + 1), C, k0, ntiles, a.tiles, a.rpre + (size_t)s * C * (a.tiles + 1),
                         a.rstart + (size_t)s * C * a.tiles, a.ring_cnt + (size_t)s * C);
    urf_scan_keys_256(&a.ring_cnt[(size_t)s * C], &a.ring_off[(size_t)s * (C + 1)], C, sh);
    }
    if (!fused) {   /* lidar_segmentation.cpp:605-608: road_probably = every point of sorted ring 10 */
        unsigned tot = 0;
        for (unsigned k = tid; k < C; k += 256)
            tot += a.ring_cnt[(size_t)s * C + k];
        for (int o = 32; o > 0; o >>= 1)
            tot += __shfl_xor(tot, o);
        if (urf_lane() == 0)
            sh[4 + (tid >> 6)] = tot;
        __syncthreads();
        if (tid == 0) {
            urf_scan_info* o = &a.info[s];
            o->n_ring_pts = sh[4] + sh[5] + sh[6] + sh[7];
            o->n_ring10 = o->n_rings > 10 ? a.ring_cnt[(size_t)s * C + 10] : 0;
            atomicMax(a.ring_hint, o->n_rings);   /* for the row's next call (k_ring_table) */
        }
    }
    if (!dp.p.star_shaped_method)
        return;
    __syncthreads();   /* sec_cnt / sec_run are other threads' stores (a fused scan has passed no barrier since urf_index_sectors) */
    urf_scan_keys_256(&a.sec_cnt[(size_t)s * K], &a.sec_off[(size_t)s * (K + 1)], K, sh);
    /* sectors too large for one wave's LDS tile go on the work lists of k_star_mid / k_star_big
     * (one atomic per wave, not per sector) */
    for (unsigned k0 = 0; k0 < K; k0 += 256) {
        const unsigned k = k0 + tid;
        const unsigned c = k < K ? a.sec_cnt[(size_t)s * K + k] : 0;
        /* (the wave-per-sector kernel takes sectors of at most two runs: one scattered over more tiles -- an
         * unorganised cloud -- goes the workgroup path whatever its size) */
        const unsigned nr = k < K ? a.sec_run[(size_t)s * K + k].nruns : 0;   /* (this thread's own store above) */
        const bool runs = (nr & URF_RUNS_FLAG) != 0u && c >= 2 && c <= URF_STAR_SMALL_CAP;
        const bool mid = (c > URF_STAR_SMALL_CAP || (nr > 2 && c >= 2 && !runs)) && c <= URF_STAR_MID_CAP_, big = c > URF_STAR_MID_CAP_;
        const unsigned long long bm = __ballot(mid), bb = __ballot(big), br = __ballot(runs);
        unsigned pm = 0, pb = 0, pr = 0;
        if ((a.optimistic & URF_OPT_NO_LISTS) && (bm | bb | br) && urf_lane() == 0)
            a.info[s].status = URF_STATUS_REDO_LISTS;   /* nobody sorts the lists in this launch sequence (every writer writes the same value) */
        if (urf_lane() == 0) {
            if (bm)
                pm = atomicAdd(&a.star_count[0], (unsigned)__popcll(bm));
            if (bb)
                pb = atomicAdd(&a.star_count[1], (unsigned)__popcll(bb));
            if (br)
                pr = atomicAdd(&a.star_count[7], (unsigned)__popcll(br));
        }
        pm = __shfl(pm, 0);
        pb = __shfl(pb, 0);
        pr = __shfl(pr, 0);
        if (runs)
            a.star_list_runs[pr + urf_popc_below(br)] = s * K + k;
        if (mid)
            a.star_list_mid[pm + urf_popc_below(bm)] = s * K + k;
        if (big)
            a.star_list_big[pb + urf_popc_below(bb)] = s * K + k;
    }
}

__global__ __launch_bounds__(256) void k_index(urf_kargs a, urf_dev_params dp)
{
    __shared__ urf_index_shared L;
    __shared__ unsigned sh[8];
    urf_index_body(a, dp, L, sh);
}

/* ------------------------------------------------------------------------- */
/* k_star_*                                                                    */
/* ------------------------------------------------------------------------- */
/* One sector = star_shaped_search.cpp:109-150: order the sector's points by planar
 * range, walk outwards, stop at the first point whose slope gives the curb away.
 *
 * Split in two so that neither half idles 63 of 64 lanes:
 *   k_star_sort_*  one wave (or workgroup) per sector: sort, then ALL lanes compute
 *                  what the walk needs and does not depend on the running mean:
 *                    slp[i] = (z_i - z_{i-1}) / (r_i - r_{i-1})            (:129)
 *                    g[i]   = (r_i - r_{i-1}) * kdist                      (:143)
 *                    first i with slp[i] > slope_param (walk stops there)  (:142)
 *                  written over the sector-major arrays in sorted order.
 *   k_star_walk    one LANE per sector: the sequential running mean /
 *                  mean-absolute-deviation recurrence (:135-140), 64 sectors
 *                  per wave.
 *
 * Sort key = (range bits << 32 | position in the sector-major array); the
 * position grows with the input index (the split is stable).  Equal ranges are
 * thereby in input order, which is NOT the order the reference's std::sort
 * leaves them in (:109): a sector whose sorted prefix holds equal neighbours is
 * flagged and sorted again by k_star_ties, below.
 * Small sectors (<= 384 points, <= 6 per lane): every 64-element block is
 * sorted in registers by an in-wave bitonic network (shuffles, no LDS traffic),
 * then each element finds its final rank by binary search in the other blocks
 * (multiway merge by ranking).  Larger sectors: bitonic network in LDS
 * ("normalised": all comparators ascending, so slots >= n act as +inf and need no
 * padding), or in global memory for sizes beyond LDS. */

/* 64 keys, one per lane, ascending by lane */
__device__ __forceinline__ unsigned long long urf_wave_sort64(unsigned long long key)
{
    const unsigned lane = urf_lane();
#pragma unroll
    for (unsigned kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (unsigned j = kk >> 1; j > 0; j >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)key, (int)j);
            const unsigned hi = __shfl_xor((unsigned)(key >> 32), (int)j);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            const bool lower = (lane & j) == 0;
            const bool up = (lane & kk) == 0 || kk == 64;
            const bool take_min = lower == up;
            const bool other_less = other < key;
            key = (take_min == other_less) ? other : key;
        }
    }
    return key;
}

/* number of entries of the sorted 64-entry block `blk` that are < key */
__device__ __forceinline__ unsigned urf_count_less64(const unsigned long long* blk, unsigned long long key)
{
    unsigned pos = 0;
#pragma unroll
    for (unsigned step = 32; step > 0; step >>= 1)
        if (blk[pos + step - 1] < key)
            pos += step;
    if (blk[pos] < key)   /* pos <= 63 */
        pos++;
    return pos;
}

/* The runs of sector k: the non-empty pieces (tile, first slot, count) of the sector in tile order
 * (k_index tables).  runP[r] = position inside the sector of the run's first point, runA[r] = index
 * of that point in the sector-sorted arrays (relative to the scan) minus runP[r], so that point i
 * of the sector lives at runA[r] + i.  One wave builds the list; returns the number of runs
 * (<= number of points of the sector). */
struct urf_run_row {   /* lane t: the sector's run in tile t (first 64 tiles): first slot and size */
    unsigned st, cnt;
};
__device__ __forceinline__ urf_run_row urf_sector_run_row(const urf_kargs& a, unsigned s, unsigned K, unsigned k, unsigned t0)
{
    const unsigned t = t0 + urf_lane();
    const uint16_t* row = a.tsoff + ((size_t)s * a.tiles + (t < a.tiles ? t : 0)) * (K + 1) + k;
    urf_run_row r;
    r.st = (unsigned)row[0];
    r.cnt = (unsigned)row[1] - r.st;
    return r;
}
/* One wave builds the list from column k of k_split's per-tile tables (a sector of an organised
 * sweep meets two or three tiles; the 64 two-byte reads of a column block hit 64 cache lines, all of
 * them shared with the neighbouring sectors' waves) and a prefix sum over the tiles. */
__device__ __forceinline__ unsigned urf_sector_runs(const urf_kargs& a, unsigned s, unsigned K, unsigned k, unsigned ntiles,
                                                    const urf_run_row& first, unsigned* runP, unsigned* runA)
{
    const unsigned lane = urf_lane();
    unsigned nr = 0, carry = 0;
    for (unsigned t0 = 0; t0 < ntiles; t0 += 64) {
        const unsigned t = t0 + lane;
        urf_run_row r = first;   /* requested by the caller along with its other inputs */
        if (t0)
            r = urf_sector_run_row(a, s, K, k, t0);
        const unsigned c = t < ntiles ? r.cnt : 0;
        const unsigned inc = urf_wave_scan_add(c);
        const unsigned p0 = carry + inc - c;
        const unsigned long long m = __ballot(c != 0);
        if (c) {
            const unsigned idx = nr + urf_popc_below(m);
            runP[idx] = p0;
            runA[idx] = t * URF_TILE + r.st - p0;
        }
        nr += (unsigned)__popcll(m);
        carry += (unsigned)__shfl((int)inc, 63);
    }
    return nr;
}

/* sectors with at most 384 points: one wave per (sector, scan).
 * Fast path: distribution sort.  The range bits are quantised monotonically
 * into URF_STAR_NB buckets ((bits - min) >> shift), a counting sort by bucket places
 * every key next to the few keys sharing its bucket, and each key then counts
 * the smaller keys inside its own bucket -- exact for any input, and about five
 * times fewer instructions than a comparison network when the ranges are
 * spread out (they are: a sector holds ~6 firings x 64 rings; the firings of
 * one ring share a bucket, a curb face puts a dozen keys into one).  If some
 * bucket collects more than 64 keys (heavily clustered ranges) the wave falls back to the
 * general path: every 64-key block is sorted in registers by an in-wave
 * bitonic network and the blocks are merged by ranking. */
/* RUNS (r6): a sector that meets MANY tiles with a few points in each -- every sector of a sweep stored ring by ring (row-major H x W:
 * tile t = ring t, ~6 of its points per sector) -- used to go to the workgroup kernel of the oversized sectors (12 barriers per
 * sector: 4.7 ms per 1024 such sweeps).  Here lane t takes the run of tile t (its address and length come from the caller): the
 * points of one ring again sit in the registers of one lane, which is what the ranking below is built for.  Such a sector publishes
 * tile-local ring-sorted indices (ssrt), as the workgroup kernels do for every sector of more than two runs. */
template <unsigned MAXB, bool RUNS = false>
__device__ __forceinline__ bool urf_star_sort_sector(const urf_kargs& a, const urf_dev_params& dp, unsigned sb, unsigned obase, unsigned n,
                                                     const urf_sec_run& two, unsigned long long* A, unsigned* cnt,
                                                     unsigned* sh_first, uint32_t* star_first_out, unsigned run_adr = 0, unsigned run_cnt = 0)
{
    constexpr unsigned NB = URF_STAR_NB, PL = NB / 64;
    /* (r5, measured: one pad word per PL counters -- a lane scans PL consecutive counters, lanes PL words apart meet in 32 / PL
     * banks -- takes the kernel's SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE from 0.38 to 0.33 and makes it 4 % SLOWER: the index
     * arithmetic costs more vector instructions than the conflicts cost cycles, profiles/r5_lds_ab.txt) */
    auto CI = [](unsigned c) { return c; };
    const unsigned lane = threadIdx.x;
    const unsigned Bn = (n + 63) >> 6;                           /* rounds over the sector's positions */
    const unsigned B = RUNS ? urf_wave_max(run_cnt) : Bn;        /* rounds over the lanes' elements */
    URF_PHASE_ACC_DECL;
    unsigned long long key[MAXB];
    float zreg[MAXB];      /* the height travels with the key: the tail then needs no dependent gathers from memory */
    /* What the walk finally needs of the sorted sector is ONE point: its curb point.  A sector of at most two
     * runs (every sector of an organised sweep, and the only kind this kernel sees) therefore publishes, per
     * sorted index, only the point's position inside the sector (2 bytes; the walk turns the one it wants into a
     * ring-sorted slot through sec_run and sslot) and never reads the slots.  Sectors scattered over more tiles take
     * the workgroup path, which carries the slot with the key and publishes tile-local ring-sorted indices. */
    unsigned sreg[MAXB];
    unsigned rmin = 0xffffffffu, rmax = 0;
    {
        /* The low half of a key is the point's index in the sector-sorted arrays: it grows with the
         * position inside the sector (tiles in order, input order inside), i.e. it breaks ties
         * exactly as the position would, and it finds the point's companions again. */
        const unsigned a1m = two.a1 - two.c0;
        /* straight-line: elements past the sector's end repeat its last one (valid addresses) and
         * are dropped afterwards; all loads of the lane are in flight together */
        unsigned adr[MAXB], rbv[MAXB], slv[MAXB];
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++) {
            const unsigned i = q * 64 + lane, ic = i < n ? i : n - 1u;
            adr[q] = RUNS ? run_adr + (q < run_cnt ? q : 0u) : ic + (ic < two.c0 ? two.a0 : a1m);
        }
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++) {
            rbv[q] = 0;
            zreg[q] = 0.f;
            slv[q] = 0;
            if (q < B) {   /* uniform */
                rbv[q] = urf_fbits(a.sr[sb + adr[q]]);
                zreg[q] = a.sz[sb + adr[q]];
                if (RUNS)
                    slv[q] = a.sslot[sb + adr[q]];
            }
        }
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++) {
            const bool valid = RUNS ? q < run_cnt : q * 64 + lane < n;
            sreg[q] = RUNS ? (slv[q] == URF_SLOT_NONE ? 0xffffffffu : (adr[q] & ~(URF_TILE - 1u)) + slv[q]) : q * 64 + lane;
            key[q] = valid ? ((unsigned long long)rbv[q] << 32) | adr[q] : ~0ull;
            rmin = valid && rbv[q] < rmin ? rbv[q] : rmin;
            rmax = valid && rbv[q] > rmax ? rbv[q] : rmax;
        }
    }
    for (unsigned c = lane; c <= NB; c += 64)
        cnt[c] = 0;
    rmin = urf_wave_min(rmin);
    rmax = urf_wave_max(rmax);
    const unsigned range = rmax - rmin;
    const unsigned sh = range < NB ? 0u : (unsigned)(32 - __clz((int)range)) - URF_STAR_LOG_NB;   /* (range >> sh) < NB */
    __syncthreads();
    URF_PHASE_ACC(0);

    unsigned bkt[MAXB], wq[MAXB];
#pragma unroll
    for (unsigned q = 0; q < MAXB; q++) {
        bkt[q] = 0;
        wq[q] = 0;
        if (q < B && key[q] != ~0ull) {
            bkt[q] = ((unsigned)(key[q] >> 32) - rmin) >> sh;
            wq[q] = atomicAdd(&cnt[CI(bkt[q])], 1u);   /* arrival order inside the bucket: resolved below */
        }
    }
    __syncthreads();
    URF_PHASE_ACC(1);
    /* exclusive scan of the counts: NB / 64 consecutive counters per lane */
    unsigned maxc = 0;
    {
        unsigned c8[PL], sum = 0;
#pragma unroll
        for (unsigned e = 0; e < PL; e++) {
            c8[e] = cnt[CI(lane * PL + e)];
            sum += c8[e];
            maxc = c8[e] > maxc ? c8[e] : maxc;
        }
        unsigned inc = sum;   /* (the DPP scan measured slower here than the shuffles: 0.77 -> 0.89 ms) */
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned w = __shfl_up(inc, o);
            if ((int)lane >= o)
                inc += w;
        }
        unsigned run = inc - sum;
#pragma unroll
        for (unsigned e = 0; e < PL; e++) {
            cnt[CI(lane * PL + e)] = run;
            run += c8[e];
        }
        if (lane == 63)
            cnt[CI(NB)] = run;   /* == n */
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned w = __shfl_xor(maxc, o);
            maxc = w > maxc ? w : maxc;
        }
    }
    __syncthreads();
    URF_PHASE_ACC(2);

    unsigned rank[MAXB];
    if (maxc <= 64 && !(dp.exp_flags & 4u)) {
        /* Rank inside the bucket = number of smaller keys in it.  In an organised sweep the keys
         * that share a bucket are the firings of ONE ring inside the sector, and those sit in the
         * registers of one lane (element q * 64 + lane = firing q, ring lane): every lane first
         * ranks its own keys against each other (15 register comparisons for 6 keys).  A key whose
         * bucket holds nothing but keys of its own lane is done; only the others read the bucket
         * from LDS. */
        unsigned ol[MAXB];   /* keys of this lane in the same bucket (incl. itself) | smaller ones among them << 8 */
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++) {
            ol[q] = 1;
            if (!(q < B && key[q] != ~0ull))
                bkt[q] = 0xffff0000u + q;   /* matches nothing */
        }
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++)
#pragma unroll
            for (unsigned r = q + 1; r < MAXB; r++) {
                const unsigned same = bkt[q] == bkt[r];
                const unsigned lt = key[q] < key[r];   /* keys are distinct (the index is part of them) */
                ol[q] += same + ((same & (lt ^ 1u)) << 8);
                ol[r] += same + ((same & lt) << 8);
            }
        bool need = false;   /* does any key of this lane share its bucket with another lane? */
        unsigned bb[MAXB];   /* bucket start | bucket size << 16 */
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++) {
            bb[q] = 0;
            if (q < B && key[q] != ~0ull) {
                const unsigned b0 = cnt[CI(bkt[q])];
                bb[q] = b0 | ((cnt[CI(bkt[q] + 1)] - b0) << 16);
                need = need || (bb[q] >> 16) != (ol[q] & 0xffu);
            }
            rank[q] = (bb[q] & 0xffffu) + (ol[q] >> 8);
        }
        if (__any(need)) {
            /* The keys of the buckets that mix lanes (a wall, a curb face: a few per cent of the keys) are
             * ranked by POSITION in the bucket-ordered copy: lane l takes positions 64 q + l, so the lanes
             * that have work in a step sit in the same one or two buckets and the step takes as many
             * trips as THAT bucket is large -- ranked by their owners, every one of the six steps had
             * some lane in the largest bucket (21 trips of four keys per sector on average instead of 5).
             * The rank travels back through the unused tail of A (n <= 384 of its 512 entries).  (Keeping it in the
             * upper halves of the bucket offsets instead makes room for a 7th wave per SIMD, which then spills 12
             * bytes at its 72 registers: 0.557 ms instead of 0.519.) */
            static_assert(MAXB * 64 <= 384, "the rank slots live behind the keys in A");
            uint16_t* RK = (uint16_t*)(A + 384);
#pragma unroll
            for (unsigned q = 0; q < MAXB; q++)
                if (q < B && key[q] != ~0ull) {
                    const unsigned pos = (bb[q] & 0xffffu) + wq[q];
                    A[pos] = key[q];
                    RK[pos] = (bb[q] >> 16) != (ol[q] & 0xffu) ? (uint16_t)0xffffu : (uint16_t)0;
                }
            __syncthreads();
#pragma unroll
            for (unsigned q = 0; q < MAXB; q++) {
                const unsigned pos = q * 64 + lane;
                if (q < Bn && pos < n && RK[pos] == 0xffffu) {
                    const unsigned long long kk = A[pos];
                    const unsigned bk = ((unsigned)(kk >> 32) - rmin) >> sh;
                    const unsigned b0 = cnt[CI(bk)], b1 = cnt[CI(bk + 1)];
                    unsigned r = b0, t = b0;
                    for (; t + 3 < b1; t += 4) {   /* four bucket-mates per trip */
                        const unsigned long long k0 = A[t], k1 = A[t + 1], k2 = A[t + 2], k3 = A[t + 3];
                        r += (k0 < kk) + (k1 < kk) + (k2 < kk) + (k3 < kk);
                    }
                    if (t + 1 < b1) {
                        const unsigned long long k0 = A[t], k1 = A[t + 1];
                        r += (k0 < kk) + (k1 < kk);
                        t += 2;
                    }
                    if (t < b1)
                        r += A[t] < kk;
                    RK[pos] = (uint16_t)r;
                }
            }
            __syncthreads();
#pragma unroll
            for (unsigned q = 0; q < MAXB; q++)
                if (q < B && key[q] != ~0ull && (bb[q] >> 16) != (ol[q] & 0xffu))
                    rank[q] = RK[(bb[q] & 0xffffu) + wq[q]];
        }
    } else {
        /* general path: in-register block sorts + multiway merge by ranking */
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++)
            if (q < B) {
                key[q] = urf_wave_sort64(key[q]);
                A[q * 64 + lane] = key[q];
                if (key[q] != ~0ull) {   /* the key moved to another lane: fetch its companions again */
                    const unsigned adr = (unsigned)key[q];
                    zreg[q] = a.sz[sb + adr];
                    /* at most two runs: the position inside the sector from the address */
                    if (RUNS) {
                        const unsigned sl = a.sslot[sb + adr];
                        sreg[q] = sl == URF_SLOT_NONE ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
                    } else {
                        sreg[q] = (two.nruns == 2 && adr >= two.a1) ? two.c0 + (adr - two.a1) : adr - two.a0;
                    }
                }
            }
        __syncthreads();
#pragma unroll
        for (unsigned q = 0; q < MAXB; q++) {
            rank[q] = lane;
            if (q < B && key[q] != ~0ull)
                for (unsigned p = 0; p < B; p++)
                    if (p != q)
                        rank[q] += urf_count_less64(A + p * 64, key[q]);
        }
    }
    __syncthreads();   /* every lane has its ranks: A and cnt may be overwritten */
    URF_PHASE_ACC(3);
    uint2* RZ = (uint2*)A;           /* (range bits, height) side by side, position / ring-sorted index, in sorted order */
    unsigned* S = cnt;
    static_assert(URF_STAR_NB + 1 >= MAXB * 64, "S reuses the bucket counters: one word per point of the sector");
#pragma unroll
    for (unsigned q = 0; q < MAXB; q++)
        if (q < B && key[q] != ~0ull) {
            RZ[rank[q]] = make_uint2((unsigned)(key[q] >> 32), __float_as_uint(zreg[q]));   /* one 8-byte scatter instead of two 4-byte ones */
            S[rank[q]] = sreg[q];
        }
    __syncthreads();
    URF_PHASE_ACC(4);
    /* tail: slopes / distance terms / ring positions in sorted order; the walk can never pass the first
     * "static" hit (slope > slope_param): stop after the 64-element chunk that holds it */
    const float slope_param = dp.slope_param, kdist = dp.p.kdist_param;
    bool tie = false;   /* two equal planar ranges next to each other where the walk may look: their order is std::sort's (k_star_ties) */
#pragma unroll
    for (unsigned q = 0; q < MAXB; q++) {
        const unsigned i = q * 64 + lane;
        if (q * 64 >= n)
            break;
        if (i < n) {
            float slp = 0.f, g = 0.f;
            if (i >= 1) {
                const uint2 pa = RZ[i - 1], pb = RZ[i];
                const float ax = __uint_as_float(pa.x), bx = __uint_as_float(pb.x);
                slp = (__uint_as_float(pb.y) - __uint_as_float(pa.y)) / (bx - ax);   /* star_shaped_search.cpp:27-30 */
                g = (bx - ax) * kdist;
                tie = tie || (pa.x == pb.x && pa.y != pb.y);   /* equal ranges, different heights: the order decides slopes */
                if (slp > slope_param)
                    atomicMin(sh_first, i);
            }
            if (RUNS)
                a.ssrt[obase + i] = S[i];
            else
                a.ssrt16[obase + i] = (uint16_t)S[i];
            a.wsg[obase + i] = urf_sg{ slp, g };
        }
        __syncthreads();
        if (*sh_first < (q + 1) * 64)
            break;
    }
    const unsigned first = *sh_first;
    /* (the points behind the walk's last one that share its range may take its place: one of another height changes the
     * slope there; twins only the identity of the point, URF_TIE_NEXT) */
    unsigned next = 0;
    if (lane == 0 && first < n)
        for (unsigned j = first + 1; j < n && RZ[j].x == RZ[first].x; j++) {
            next = URF_TIE_NEXT;
            tie = tie || RZ[j].y != RZ[first].y;
        }
    const bool any_tie = __any(tie);
    if (lane == 0)
        *star_first_out = (first < n - 1 ? first : n - 1) | (any_tie ? URF_TIE_FLAG : 0u) | next;   /* last index the walk may visit */
    URF_PHASE_ACC(5);
#ifdef URF_EXP_PHASE_CLOCK
    if (threadIdx.x == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.x >= 100 && blockIdx.x < 104)
        printf("k_star_sort_small sector %u n %u: load %llu count %llu scan %llu rank %llu place %llu tail %llu\n", blockIdx.x, n, ph_t[0], ph_t[1], ph_t[2], ph_t[3], ph_t[4], ph_t[5]);
#endif
    return any_tie;
}

/* a sector was flagged with URF_TIE_FLAG: tell k_star_ties' instance for its size that there is work -- or, in a launch
 * sequence without it (callback path), void the sweep: urf_classify_pc2_wait() runs it again with the kernel (every writer
 * writes the same value) */
__device__ __forceinline__ void urf_tie_found(const urf_kargs& a, unsigned s, unsigned sk)
{
    a.tie_list[atomicAdd(&a.star_count[4], 1u)] = sk;   /* (one sector in a hundred of a sensor's sweep; none of a benchmark cloud) */
    if (a.optimistic & URF_OPT_NO_TIES)
        a.info[s].status = URF_STATUS_REDO_TIES;
}


/* amdgpu_waves_per_eu(6, 6): 6 KB of LDS allow 26 waves per CU; without the cap the register ranking
 * below takes 98 VGPRs and halves the occupancy (0.75 -> 0.92 ms instead of 0.70) */
#ifndef URF_SMALL_WAVES
#define URF_SMALL_WAVES 6
#endif
__global__ __launch_bounds__(URF_STAR_THREADS) __attribute__((amdgpu_waves_per_eu(URF_SMALL_WAVES, URF_SMALL_WAVES))) void k_star_sort_small(urf_kargs a, urf_dev_params dp)
{
    constexpr unsigned NB = URF_STAR_NB;            /* buckets */
    __shared__ unsigned long long A[8 * 64];        /* keys by bucket, then range / height of the sorted sector */
    __shared__ unsigned cnt[NB + 1];                /* bucket counts, then exclusive offsets, then ring positions */
    __shared__ unsigned sh_first;
    const unsigned k = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    const unsigned K = (unsigned)dp.p.sectors;
    /* status, both ends of the sector and its run table in ONE round trip */
    const int status = a.info[s].status;
    const unsigned so0 = a.sec_off[(size_t)s * (K + 1) + k], so1 = a.sec_off[(size_t)s * (K + 1) + k + 1];
    const urf_sec_run two = a.sec_run[(size_t)s * K + k];
    if (status != URF_OK)
        return;
    const unsigned n = so1 - so0;
    /* a sector of an organised sweep meets one or two tiles, and k_index described those runs: that is the only case
     * this kernel handles.  One scattered over more tiles (an unorganised cloud) or of more than 384 points is on a
     * work list of the workgroup kernels (k_index). */
    if (n > URF_STAR_SMALL_CAP || (two.nruns > 2 && n >= 2))
        return;
    if (n < 2) {
        if (lane == 0)
            a.star_first[(size_t)s * K + k] = 0;   /* nothing to walk */
        return;
    }
    const unsigned sb = urf_sbase(a, s), obase = sb + so0;
    if (lane == 0)
        sh_first = n;
    /* per-lane element count fixed at compile time: 6 covers a sector of a 64 x 2048 sweep.  (An
     * 8-per-lane instance for sectors of up to 512 points made the kernel spill 68 bytes per lane at
     * its 80 registers; such sectors take the workgroup path now.) */
    const bool tie = urf_star_sort_sector<URF_STAR_SMALL_CAP / 64>(a, dp, sb, obase, n, two, A, cnt, &sh_first, &a.star_first[(size_t)s * K + k]);
    if (tie && lane == 0)
        urf_tie_found(a, s, s * K + k);
}

/* the sectors k_index listed as "many short runs" (URF_RUNS_FLAG: more than two runs, at most 64 tiles, at most six points per
 * run -- a sweep stored ring by ring; normally none): persistent waves over the list */
__global__ __launch_bounds__(URF_STAR_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_star_sort_runs(urf_kargs a, urf_dev_params dp)
{
    constexpr unsigned NB = URF_STAR_NB;
    __shared__ unsigned long long A[8 * 64];
    __shared__ unsigned cnt[NB + 1];
    __shared__ unsigned sh_first;
    const unsigned count = a.star_count[7], lane = threadIdx.x;
    const unsigned K = (unsigned)dp.p.sectors;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        const unsigned sk = a.star_list_runs[w], s = sk / K, k = sk % K;
        if (a.info[s].status != URF_OK)
            continue;
        unsigned off, len;
        urf_scan_range(a, s, off, len);
        const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
        const unsigned so0 = a.sec_off[(size_t)s * (K + 1) + k], so1 = a.sec_off[(size_t)s * (K + 1) + k + 1];
        const urf_sec_run two = a.sec_run[sk];
        const unsigned n = so1 - so0;
        unsigned c0 = 0, c1 = 0;
        if (lane < ntiles) {   /* (ntiles <= 64: k_index) */
            const uint16_t* row = a.tsoff + ((size_t)s * a.tiles + lane) * (K + 1) + k;
            c0 = row[0];
            c1 = row[1];
        }
        const unsigned sb = urf_sbase(a, s), obase = sb + so0;
        if (lane == 0)
            sh_first = n;
        const bool tie = urf_star_sort_sector<URF_STAR_SMALL_CAP / 64, true>(a, dp, sb, obase, n, two, A, cnt, &sh_first, &a.star_first[sk],
                                                                              c1 > c0 ? lane * URF_TILE + c0 : 0u, c1 - c0);   /* (a lane without a run reads the scan's first element, never a tile behind its last) */
        if (tie && lane == 0)
            urf_tie_found(a, s, sk);
        __syncthreads();   /* the LDS is reused by the next sector */
    }
}

template <int NT>
__device__ __forceinline__ void urf_bitonic_keys(unsigned long long* keys, unsigned n)
{
    unsigned P = 1;
    while (P < n)
        P <<= 1;
    for (unsigned kk = 2; kk <= P; kk <<= 1) {
        for (unsigned j = kk >> 1; j > 0; j >>= 1) {
            const bool flip = (j == (kk >> 1));
            for (unsigned tt = threadIdx.x; tt < (P >> 1); tt += NT) {
                const unsigned lo = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
                /* flip step: partner of lo inside its block of size kk is block_end - (lo - block_start) */
                const unsigned hi = flip ? ((lo & ~(kk - 1)) + (kk - 1) - (lo & (kk - 1))) : lo + j;
                if (hi < n) {
                    const unsigned long long ka = keys[lo], kb = keys[hi];
                    if (ka > kb) {
                        keys[lo] = kb;
                        keys[hi] = ka;
                    }
                }
            }
            __syncthreads();
        }
    }
}

/* Workgroup-wide sort of up to NT*EPT 64-bit keys (element tid + e*NT in key[e], ~0 = none):
 * the distribution sort of k_star_sort_small with NB buckets and workgroup-wide reductions,
 * the normalised bitonic network in LDS as the fallback for clustered keys.  The sorted keys
 * end up in A[0..n). */
struct urf_sort_shared {
    unsigned rmin, rmax, maxc;
    unsigned w[8];
};
#define URF_BLOCK_CNT(NB, NT) ((NB) + 1)   /* words of the counter array */
/* rank[e] = number of keys of the workgroup smaller than key[e] (keys are distinct).  The keys come
 * back PERMUTED among the threads (every key exactly once, each with its rank). */
template <int NT, int EPT, int NB>
__device__ __forceinline__ void urf_block_rank_keys(unsigned long long (&key)[EPT], unsigned n, unsigned long long* A,
                                                    unsigned* cnt, urf_sort_shared* sh, bool force_general, unsigned (&rank)[EPT] URF_PH_PARAMS)
{
    static_assert(NB % NT == 0 && NT / 64 <= 8, "bucket scan layout");
    auto CI = [](unsigned c) { return c; };   /* (padded counters: measured slower, see urf_star_sort_sector) */
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        /* (materialised here: hoisted out of the persistent loop of k_star_sort_mid, these three constants sat in registers
         * the kernel does not have and went through scratch memory) */
        unsigned ones = 0xffffffffu, zero = 0u;
        asm volatile("" : "+v"(ones), "+v"(zero));
        sh->rmin = ones;
        sh->rmax = zero;
        sh->maxc = zero;
    }
    for (unsigned c = tid; c <= NB; c += NT)
        cnt[c] = 0;
    __syncthreads();
    URF_PHASE_ACC(4);
    unsigned rmin = 0xffffffffu, rmax = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (key[e] != ~0ull) {
            const unsigned rb = (unsigned)(key[e] >> 32);
            rmin = rb < rmin ? rb : rmin;
            rmax = rb > rmax ? rb : rmax;
        }
    rmin = urf_wave_min(rmin);   /* DPP: no bpermute addresses / lane masks for the compiler to hoist out of */
    rmax = urf_wave_max(rmax);   /* the persistent loop (they cost the kernel registers it does not have) */
    if (lane == 0) {
        atomicMin(&sh->rmin, rmin);
        atomicMax(&sh->rmax, rmax);
    }
    __syncthreads();
    URF_PHASE_ACC(5);
    rmin = sh->rmin;
    const unsigned range = sh->rmax - rmin;
    unsigned shf = 0;
    while ((range >> shf) >= (unsigned)NB)   /* (range >> shf) < NB */
        shf++;
    unsigned bkt[EPT], wq[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        bkt[e] = 0;
        wq[e] = 0;
        if (key[e] != ~0ull) {
            bkt[e] = ((unsigned)(key[e] >> 32) - rmin) >> shf;
            wq[e] = atomicAdd(&cnt[CI(bkt[e])], 1u);
        }
    }
    __syncthreads();
    URF_PHASE_ACC(6);
    {   /* exclusive scan of the NB counts: NB/NT consecutive counters per thread */
        unsigned c8[NB / NT], sum = 0, maxc = 0;
#pragma unroll
        for (int e = 0; e < NB / NT; e++) {
            c8[e] = cnt[CI(tid * (NB / NT) + e)];
            sum += c8[e];
            maxc = c8[e] > maxc ? c8[e] : maxc;
        }
        const unsigned inc = urf_wave_scan_add(sum);
        if (lane == 63)
            sh->w[wave] = inc;
        maxc = urf_wave_max(maxc);
        if (lane == 0)
            atomicMax(&sh->maxc, maxc);
        __syncthreads();
        unsigned run = inc - sum;
        for (unsigned v = 0; v < wave; v++)
            run += sh->w[v];
#pragma unroll
        for (int e = 0; e < NB / NT; e++) {
            cnt[CI(tid * (NB / NT) + e)] = run;
            run += c8[e];
        }
        if (tid == NT - 1)
            cnt[CI(NB)] = run;
    }
    __syncthreads();
    URF_PHASE_ACC(7);
    /* in-bucket ranking is quadratic in the bucket size, but up to a few hundred keys per bucket it is
     * still cheaper than the bitonic network below (128 x 4096 sweeps: 2.13 -> 1.74 ms with 256 instead of 64) */
    if (sh->maxc <= 256 && !force_general) {
#pragma unroll
        for (int e = 0; e < EPT; e++)
            if (key[e] != ~0ull)
                A[cnt[CI(bkt[e])] + wq[e]] = key[e];
        __syncthreads();
        URF_PHASE_ACC(8);
        /* From here on a thread owns the keys at POSITIONS tid + e * NT of the bucket-ordered array
         * instead of the ones it loaded: the lanes of a wave then sit in the same few buckets and loop
         * equally long.  (A wall puts 70 or 100 keys of a 128 x 4096 sweep's sector into one bucket; owned by
         * 70 threads spread over all eight waves, every wave looped as long as that bucket is large.)
         * The caller goes on with the (key, rank) pairs it gets back. */
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const unsigned pos = tid + (unsigned)e * NT;
            key[e] = pos < n ? A[pos] : ~0ull;
        }
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            rank[e] = 0;
            if (key[e] != ~0ull) {
                const unsigned b = ((unsigned)(key[e] >> 32) - rmin) >> shf;
                const unsigned b0 = cnt[CI(b)], b1 = cnt[CI(b + 1)];
                unsigned r = b0, t = b0;
                for (; t + 1 < b1; t += 2) {   /* two bucket-mates per trip */
                    const unsigned long long k0 = A[t], k1 = A[t + 1];
                    r += (k0 < key[e]) + (k1 < key[e]);
                }
                if (t < b1)
                    r += A[t] < key[e];
                rank[e] = r;
            }
        }
        __syncthreads();
        URF_PHASE_ACC(9);
    } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const unsigned i = tid + (unsigned)e * NT;
            if (i < n)
                A[i] = key[e];
        }
        __syncthreads();
        urf_bitonic_keys<NT>(A, n);
#pragma unroll
        for (int e = 0; e < EPT; e++) {   /* where did the key end up? */
            unsigned lo = 0, hi = n;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if (A[mid] < key[e])
                    lo = mid + 1;
                else
                    hi = mid;
            }
            rank[e] = lo;
        }
        __syncthreads();
    }
}

/* ... and the sorted keys in A[0..n) */
template <int NT, int EPT, int NB>
__device__ __forceinline__ void urf_block_sort_keys(unsigned long long (&key)[EPT], unsigned n, unsigned long long* A,
                                                    unsigned* cnt, urf_sort_shared* sh, bool force_general)
{
    unsigned rank[EPT];
    URF_PHASE_ACC_DECL;
    urf_block_rank_keys<NT, EPT, NB>(key, n, A, cnt, sh, force_general, rank URF_PH_ARGS);
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (key[e] != ~0ull)
            A[rank[e]] = key[e];
    __syncthreads();
}

/* sectors with 385..2048 points (e.g. 128 rings x 4096 columns): persistent
 * workgroups of 256 threads walk the work list built by k_index. */
#ifndef URF_STAR_MID_THREADS
#define URF_STAR_MID_THREADS 512   /* A/B on 256 x 128x4096 sweeps: 256 threads x 4 waves/SIMD 1.93 ms, 512 x 6 1.91 ms, 512 x 8 1.72 ms */
#endif
#define URF_STAR_MID_CAP 2048
#ifndef URF_MID_WAVES
#define URF_MID_WAVES 8
#endif
__global__ __launch_bounds__(URF_STAR_MID_THREADS) __attribute__((amdgpu_waves_per_eu(URF_MID_WAVES, URF_MID_WAVES))) void k_star_sort_mid(urf_kargs a, urf_dev_params dp)
{
    constexpr unsigned NT = URF_STAR_MID_THREADS, NB = 2048, EPT = URF_STAR_MID_CAP / NT;
    __shared__ unsigned long long A[URF_STAR_MID_CAP];
    __shared__ unsigned cnt[URF_BLOCK_CNT(NB, NT)];
    __shared__ urf_sort_shared ssh;
    __shared__ unsigned sh_first, sh_nruns, sh_tie;
    const unsigned K = (unsigned)dp.p.sectors;
    const unsigned count = a.star_count[0];
    const unsigned tid = threadIdx.x;
    URF_PHASE_ACC_DECL;
    /* The description of a sector (list entry -> size, place, first two runs: two dependent round
     * trips) is fetched one iteration ahead, into scalar registers: at the top of an iteration it
     * has long arrived.  (Fetched on the spot, with the run list built from the per-tile tables by
     * one wave, this cost 7 000 of the 32 000 cycles a sector took.) */
    auto list_entry = [&](unsigned w) -> unsigned { return w < count ? a.star_list_mid[w] : 0u; };
    auto rfl = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    /* (r5) The list entry is fetched TWO iterations ahead and the description one, and both are taken into scalar registers
     * in front of the tail's stores: loads and stores share one in-order counter on this chip, so a load still pending when
     * the tail's barrier comes makes the workgroup wait for the acknowledgement of every store issued before it -- 6 100 of a
     * sector's 20 300 cycles went there, and another round trip into the list entry at the top of every iteration. */
    struct urf_mid_desc {
        unsigned sk, n, so, a0, c0, a1, nruns;
    };
    urf_mid_desc cur;
    cur.sk = rfl(list_entry(blockIdx.x));
    {
        const urf_sec_run t = a.sec_run[cur.sk];
        cur.n = rfl(a.sec_cnt[cur.sk]);
        cur.so = rfl(a.sec_off[(size_t)(cur.sk / K) * (K + 1) + cur.sk % K]);
        cur.a0 = rfl(t.a0);
        cur.c0 = rfl(t.c0);
        cur.a1 = rfl(t.a1);
        cur.nruns = rfl(t.nruns);
    }
    unsigned sk_next = rfl(list_entry(blockIdx.x + gridDim.x));
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        const unsigned sk = cur.sk;
        const unsigned s = sk / K, k = sk % K;
        unsigned off, len;
        urf_scan_range(a, s, off, len);
        const unsigned n = cur.n;
        const unsigned sb = urf_sbase(a, s);
        const unsigned obase = sb + cur.so;
        const unsigned two_a0 = cur.a0, two_c0 = cur.c0, two_a1 = cur.a1;
        const bool simple = cur.nruns <= 2u;
        /* a sector scattered over more than two tiles: its runs (<= n <= 2048 of them) are listed in A's
         * memory until the keys are in registers */
        unsigned* runP = (unsigned*)A;
        unsigned* runA = runP + URF_STAR_MID_CAP;
        if (!simple && tid < 64) {
            const unsigned nr = urf_sector_runs(a, s, K, k, (len + URF_TILE - 1) / URF_TILE, urf_sector_run_row(a, s, K, k, 0), runP, runA);
            if (tid == 0)
                sh_nruns = nr;
        }
        if (tid == 0) {
            sh_first = n;
            sh_tie = 0;
        }
        __syncthreads();
        URF_PHASE_ACC(0);
        const unsigned nruns = simple ? 0u : sh_nruns;
        unsigned long long key[EPT];
        float zreg[EPT];      /* height and ring-sorted index of the keys (fetched after the ranking) */
        unsigned sreg[EPT];
        unsigned r = 0;
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const unsigned i = tid + e * NT;
            key[e] = ~0ull;
            zreg[e] = 0.f;
            sreg[e] = 0;
            if (i < n) {
                unsigned adr = i < two_c0 ? two_a0 + i : two_a1 + (i - two_c0);   /* grows with i: the tie-break */
                if (!simple) {
                    while (r + 1 < nruns && i >= runP[r + 1])
                        r++;
                    adr = runA[r] + i;
                }
                key[e] = ((unsigned long long)urf_fbits(a.sr[sb + adr]) << 32) | adr;
            }
        }
        /* the next sector's description and the list entry behind it: requested now, taken in front of the tail */
        const unsigned n_nx = a.sec_cnt[sk_next];
        const unsigned so_nx = a.sec_off[(size_t)(sk_next / K) * (K + 1) + sk_next % K];
        const urf_sec_run two_nx = a.sec_run[sk_next];
        const unsigned sk_nx2 = list_entry(w + 2 * gridDim.x);
        __syncthreads();   /* the run list has been read: A is free */
        URF_PHASE_ACC(1);
        unsigned rank[EPT];
        urf_block_rank_keys<NT, EPT, NB>(key, n, A, cnt, &ssh, (dp.exp_flags & 4u) != 0, rank URF_PH_ARGS);
        /* height and ring-sorted index are fetched once the ranks are known (the low half of a key is
         * the point's place in the sector-sorted arrays): carried along from the start they did not
         * fit the 64 registers of 8 waves per SIMD and went through scratch memory */
        /* (at most two runs: the point's position inside the sector instead of its slot, see urf_star_sort_sector) */
#pragma unroll
        for (unsigned e = 0; e < EPT; e++)
            if (key[e] != ~0ull) {
                const unsigned adr = (unsigned)key[e];
                zreg[e] = a.sz[sb + adr];
                if (simple) {
                    sreg[e] = adr >= two_a1 && two_a1 > two_a0 ? two_c0 + (adr - two_a1) : adr - two_a0;
                } else {
                    const unsigned sl = a.sslot[sb + adr];
                    sreg[e] = sl == URF_SLOT_NONE ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
                }
            }
        URF_PHASE_ACC(2);
        /* range bits, height, ring-sorted index in sorted order (A and cnt are free again) */
        unsigned* R = (unsigned*)A;
        float* Z = (float*)A + URF_STAR_MID_CAP;
        unsigned* S = cnt;
#pragma unroll
        for (unsigned e = 0; e < EPT; e++)
            if (key[e] != ~0ull) {
                R[rank[e]] = (unsigned)(key[e] >> 32);
                Z[rank[e]] = zreg[e];
                S[rank[e]] = sreg[e];
            }
        {   /* every load of the iteration has arrived by now: none is pending when the tail's stores go out */
            urf_mid_desc nx;
            nx.sk = sk_next;
            nx.n = rfl(n_nx);
            nx.so = rfl(so_nx);
            nx.a0 = rfl(two_nx.a0);
            nx.c0 = rfl(two_nx.c0);
            nx.a1 = rfl(two_nx.a1);
            nx.nruns = rfl(two_nx.nruns);
            sk_next = rfl(sk_nx2);
            cur = nx;
        }
        __syncthreads();
        /* tail: slopes / distance terms / ring-sorted indices in sorted order; the walk can never pass the
         * first "static" hit (slope > slope_param), so stop after the chunk of NT elements that holds it */
        const float slope_param = dp.slope_param, kdist = dp.p.kdist_param;
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const unsigned i = tid + e * NT;
            if (e * NT >= n)
                break;
            if (i < n) {
                float slp = 0.f, g = 0.f;
                if (i >= 1) {
                    const float ax = __uint_as_float(R[i - 1]), bx = __uint_as_float(R[i]);
                    slp = (Z[i] - Z[i - 1]) / (bx - ax);   /* star_shaped_search.cpp:27-30 */
                    g = (bx - ax) * kdist;
                    if (R[i - 1] == R[i] && __float_as_uint(Z[i - 1]) != __float_as_uint(Z[i]))
                        sh_tie = 1u;   /* equal planar ranges of different heights where the walk may look: k_star_ties */
                    if (slp > slope_param)
                        atomicMin(&sh_first, i);
                }
                if (simple)
                    a.ssrt16[obase + i] = (uint16_t)S[i];
                else
                    a.ssrt[obase + i] = S[i];
                a.wsg[obase + i] = urf_sg{ slp, g };
            }
            __syncthreads();
            if (sh_first < (e + 1) * NT)
                break;
        }
        const unsigned first = sh_first;
        if (tid == 0) {
            bool tie = sh_tie != 0u;
            unsigned next = 0;
            if (first < n)
                for (unsigned j = first + 1; j < n && R[j] == R[first]; j++) {
                    next = URF_TIE_NEXT;
                    tie = tie || __float_as_uint(Z[j]) != __float_as_uint(Z[first]);
                }
            a.star_first[sk] = (first < n - 1 ? first : n - 1) | (tie ? URF_TIE_FLAG : 0u) | next;
            if (tie)
                urf_tie_found(a, s, sk);
        }
        __syncthreads();
        URF_PHASE_ACC(3);
    }
#ifdef URF_EXP_PHASE_CLOCK
    if (threadIdx.x == 0 && blockIdx.x < 3)
        printf("k_star_sort_mid wg %u: runs %llu load %llu rank-rest %llu tail %llu | zero %llu minmax %llu count %llu scan %llu scatter %llu loop %llu cycles, %u sectors\n", blockIdx.x, ph_t[0], ph_t[1], ph_t[2], ph_t[3],
               ph_t[4], ph_t[5], ph_t[6], ph_t[7], ph_t[8], ph_t[9], (count + gridDim.x - 1 - blockIdx.x) / gridDim.x);
#endif
}

/* sectors with more than 2048 points (adversarial clouds): gathered into sector-major
 * copies and sorted there, in global memory, by one workgroup each, same network, keys (range,
 * position in the sector = input order); then slopes in a second sweep. */
__global__ __launch_bounds__(256) void k_star_sort_big(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned sh_first, sh_tie;
    __shared__ unsigned P[URF_MAX_TILES + 1];   /* the sector's points in the tiles before t */
    __shared__ uint16_t ST[URF_MAX_TILES];      /* first slot of its run in tile t */
    const unsigned K = (unsigned)dp.p.sectors;
    const unsigned count = a.star_count[1];
    const float slope_param = dp.slope_param, kdist = dp.p.kdist_param;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        const unsigned sk = a.star_list_big[w];
        const unsigned s = sk / K, k = sk % K;
        unsigned off, len;
        urf_scan_range(a, s, off, len);
        const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
        const unsigned n = a.sec_cnt[(size_t)s * K + k];
        const unsigned sb = urf_sbase(a, s), base = sb + a.sec_off[(size_t)s * (K + 1) + k];
        if (threadIdx.x < 64) {   /* one wave: column k of the per-tile tables, prefix over the tiles */
            unsigned carry = 0;
            for (unsigned t0 = 0; t0 < ntiles; t0 += 64) {
                const unsigned t = t0 + threadIdx.x;
                const urf_run_row r = urf_sector_run_row(a, s, K, k, t0);
                const unsigned c = t < ntiles ? r.cnt : 0;
                const unsigned inc = urf_wave_scan_add(c);
                if (t < ntiles) {
                    P[t] = carry + inc - c;
                    ST[t] = (uint16_t)r.st;
                }
                carry += (unsigned)__shfl((int)inc, 63);
            }
            if (threadIdx.x == 0)
                P[ntiles] = carry;
        }
        __syncthreads();
        float* R = a.big_r + base;
        float* Z = a.big_z + base;
        unsigned* I = a.big_i + base;
        unsigned* Pq = a.ssrt + base;   /* original position in the sector = input order: the tie-break */
        if (threadIdx.x == 0) {
            sh_first = n;
            sh_tie = 0;
        }
        for (unsigned i = threadIdx.x; i < n; i += 256) {
            unsigned lo = 0, hi = ntiles;   /* largest tile t with P[t] <= i (its run is not empty) */
            while (hi - lo > 1) {
                const unsigned mid = (lo + hi) >> 1;
                if (P[mid] <= i)
                    lo = mid;
                else
                    hi = mid;
            }
            const unsigned adr = sb + lo * URF_TILE + ST[lo] + (i - P[lo]);
            const unsigned sl = a.sslot[adr];
            R[i] = a.sr[adr];
            Z[i] = a.sz[adr];
            I[i] = sl == URF_SLOT_NONE ? 0xffffffffu : lo * URF_TILE + sl;
            Pq[i] = i;
        }
        __threadfence_block();
        __syncthreads();
        unsigned P2 = 1;
        while (P2 < n)
            P2 <<= 1;
        for (unsigned kk = 2; kk <= P2; kk <<= 1) {
            for (unsigned j = kk >> 1; j > 0; j >>= 1) {
                const bool flip = (j == (kk >> 1));
                for (unsigned tt = threadIdx.x; tt < (P2 >> 1); tt += 256) {
                    const unsigned lo = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
                    const unsigned hi = flip ? ((lo & ~(kk - 1)) + (kk - 1) - (lo & (kk - 1))) : lo + j;
                    if (hi < n) {
                        const unsigned long long ka = ((unsigned long long)urf_fbits(R[lo]) << 32) | Pq[lo];
                        const unsigned long long kb = ((unsigned long long)urf_fbits(R[hi]) << 32) | Pq[hi];
                        if (ka > kb) {
                            const float r0 = R[lo], z0 = Z[lo];
                            const unsigned i0 = I[lo], p0 = Pq[lo];
                            R[lo] = R[hi]; Z[lo] = Z[hi]; I[lo] = I[hi]; Pq[lo] = Pq[hi];
                            R[hi] = r0; Z[hi] = z0; I[hi] = i0; Pq[hi] = p0;
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        unsigned first = n;
        for (unsigned i = threadIdx.x; i < n; i += 256) {
            float slp = 0.f, g = 0.f;
            if (i >= 1) {
                slp = (Z[i] - Z[i - 1]) / (R[i] - R[i - 1]);
                g = (R[i] - R[i - 1]) * kdist;
                if (R[i] == R[i - 1])
                    sh_tie = 1u;   /* equal planar ranges anywhere in the sector: k_star_ties */
                if (slp > slope_param && i < first)
                    first = i;
            }
            a.wsg[base + i] = urf_sg{ slp, g };
        }
        __syncthreads();   /* every Pq (= ssrt) has been read for the last time */
        for (unsigned i = threadIdx.x; i < n; i += 256)
            a.ssrt[base + i] = I[i];
        atomicMin(&sh_first, first);
        __syncthreads();
        if (threadIdx.x == 0) {
            a.star_first[sk] = (sh_first < n - 1 ? sh_first : n - 1) | (sh_tie ? URF_TIE_FLAG : 0u);
            if (sh_tie)
                urf_tie_found(a, s, sk);
        }
        __syncthreads();
    }
}

/* ---- equal planar ranges: the order std::sort leaves them in ------------------------------------------------
 * star_shaped_search.cpp:109 sorts a sector with std::sort(.., ptcmpr), ptcmpr(a, b) = a.r < b.r (:22-25).  Where two
 * points of a sector share their float range the result depends on the ALGORITHM -- libstdc++'s introsort is not stable
 * but it is deterministic, the walk divides by the difference of neighbouring ranges (a tie is +-inf or NaN, and which of
 * the two points comes second decides the sign), so the reference's labels depend on that order.  The benchmark clouds
 * are tie-free by construction; a real sensor's sweep -- ranges quantised to millimetres,
 * neighbouring firings of a ring on flat ground -- holds such pairs in every sector.  The sort kernels above order equal
 * ranges by position (the stable order) and list a sector whose sorted prefix, as far as the walk may look plus one, holds
 * equal neighbours of DIFFERENT heights (URF_TIE_FLAG in star_first, tie_list); this kernel then sorts the listed sector
 * AGAIN, as libstdc++ does (bits/stl_algo.h of GCC 5 .. 13: __sort -> __introsort_loop -> __unguarded_partition_pivot /
 * __partial_sort, __final_insertion_sort), and rewrites everything the sort kernels wrote for it.  Equal neighbours of one
 * height (twins: nearly all of a sensor's) do not change what the walk computes, only which of them stands where it stops:
 * the walk kernels list such a sector (URF_TIE_POST, tie_post) and the second pass picks that one point (urf_tie_select).
 *
 * One wave per sector.  What has to be followed literally is the introsort loop: only it moves equal elements past each
 * other.  (a) __move_median_to_first on (first + 1, mid, last - 1).  (b) __unguarded_partition against the pivot now at
 * `first`, in two data-parallel passes: with L = the positions of [first + 1, last) holding an element >= pivot in
 * ascending order and R = those holding one <= pivot in descending order, the sequential loop swaps exactly the pairs
 * (L[k], R[k]) with L[k] < R[k] -- what lies between the two pointers is untouched until they get there, and such k form a
 * prefix k < k* -- and returns cut = min(L[k*], R[k* - 1]).  (c) [cut, last) and [first, cut) go on while longer than 16
 * elements, depth limit 2 * floor(log2 n); a segment that reaches the limit is heap sorted (__partial_sort = __make_heap +
 * __sort_heap) by ONE lane, statement by statement -- an adversarial input's business.  (d) __final_insertion_sort is a
 * stable sort of what the loop leaves, and that is a sequence of segments of at most 16 elements (or heap sorted ones),
 * each <= the next: every element's final place is its segment's start + the smaller elements of the segment + the equal
 * ones in front of it.  tests/test_stdsort.py pins the same formulation on the CPU against the real std::sort.
 *
 * The arrays (range bits, the points, two work arrays of positions) live in LDS for sectors of up to URF_TIE_CAP points
 * (32 KB) and beyond that in the sector's stretch of big_r / big_i / big_z / ssrt. */
__device__ __forceinline__ int urf_walk_slot_to_ring_pos(const urf_kargs& a, unsigned s, unsigned C, unsigned v);
/* two memory policies: LDS (sectors of up to URF_TIE_CAP points) and global memory; the index arrays hold the points'
 * addresses in the sector-sorted arrays */
struct urf_tie_lds {
    typedef unsigned* rptr;
    typedef unsigned* iptr;
    static __device__ __forceinline__ void sync() { urf_wave_lds_sync(); }
};
struct urf_tie_glb {
    typedef volatile unsigned* rptr;   /* (volatile: one lane writes what the others read next) */
    typedef volatile unsigned* iptr;
    static __device__ __forceinline__ void sync()
    {
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
};

/* stl_heap.h: __adjust_heap + __push_heap on the segment starting at f (one lane) */
template <class RP_, class IP_>
__device__ void urf_tie_adjust_heap(RP_ R, IP_ P, unsigned f, int hole, int len, unsigned v, unsigned pv)
{
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (R[f + second] < R[f + second - 1])
            second--;
        R[f + hole] = R[f + second];
        P[f + hole] = P[f + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        R[f + hole] = R[f + second - 1];
        P[f + hole] = P[f + second - 1];
        hole = second - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && R[f + parent] < v) {
        R[f + hole] = R[f + parent];
        P[f + hole] = P[f + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    R[f + hole] = v;
    P[f + hole] = pv;
}
/* stl_algo.h __partial_sort(first, last, last): __make_heap, then __sort_heap */
template <class RP_, class IP_>
__device__ __noinline__ void urf_tie_heap_sort(RP_ R, IP_ P, unsigned f, unsigned l)
{
    const int len = (int)(l - f);
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            urf_tie_adjust_heap(R, P, f, parent, len, R[f + parent], P[f + parent]);
            if (parent == 0)
                break;
            parent--;
        }
    }
    for (int last = len - 1; last >= 1; last--) {
        const unsigned v = R[f + last], pv = P[f + last];
        R[f + last] = R[f];
        P[f + last] = P[f];
        urf_tie_adjust_heap(R, P, f, 0, last, v, pv);
    }
}

/* __unguarded_partition_pivot(first, last) on [f, l), l - f > 16, by one wave: returns the cut.  R = range bits, P = the
 * points (moved along), LP / RP: work arrays (the stretch [f + 1, l) of each is used). */
template <class MEM>
__device__ __forceinline__ unsigned urf_tie_partition(typename MEM::rptr R, typename MEM::iptr P, typename MEM::iptr LP, typename MEM::iptr RP,
                                                      unsigned f, unsigned l)
{
    const unsigned lane = threadIdx.x;
    {   /* __move_median_to_first(first, first + 1, mid, last - 1) */
        const unsigned mid = f + (l - f) / 2;
        const unsigned va = R[f + 1], vb = R[mid], vc = R[l - 1];
        unsigned m;
        if (va < vb)
            m = vb < vc ? mid : (va < vc ? l - 1 : f + 1);
        else if (va < vc)
            m = f + 1;
        else if (vb < vc)
            m = l - 1;
        else
            m = mid;
        if (lane == 0) {
            const unsigned r0 = R[f], p0 = P[f];
            R[f] = R[m];
            P[f] = P[m];
            R[m] = r0;
            P[m] = p0;
        }
        MEM::sync();
    }
    const unsigned pv = R[f];
    /* __unguarded_partition(first + 1, last, first): where the left pointer can stop (>= pivot), where the right one (<= pivot) */
    unsigned tL = 0, tR = 0;
    for (unsigned c0 = f + 1; c0 < l; c0 += 64) {
        const unsigned p = c0 + lane;
        const bool in = p < l;
        const unsigned v = in ? R[p] : 0u;
        const bool isL = in && v >= pv, isR = in && v <= pv;
        const unsigned long long mL = __ballot(isL), mR = __ballot(isR);
        if (isL)
            LP[f + 1 + tL + urf_popc_below(mL)] = p;
        if (isR)
            RP[f + 1 + tR + urf_popc_below(mR)] = p;   /* ascending; the k-th from the right is entry tR - 1 - k */
        tL += (unsigned)__popcll(mL);
        tR += (unsigned)__popcll(mR);
    }
    MEM::sync();
    const unsigned mn = tL < tR ? tL : tR;
    unsigned ks = 0;
    for (unsigned k0 = 0; k0 < mn; k0 += 64) {
        const unsigned kk = k0 + lane;
        const bool in = kk < mn;
        const unsigned lp = in ? LP[f + 1 + kk] : 0u, rp = in ? RP[f + tR - kk] : 0u;
        const bool ok = in && lp < rp;
        const unsigned long long mo = __ballot(ok), mi = __ballot(in);
        if (ok) {   /* iter_swap: the positions of all pairs are distinct */
            const unsigned r0 = R[lp], p0 = P[lp], r1 = R[rp], p1 = P[rp];
            R[lp] = r1;
            P[lp] = p1;
            R[rp] = r0;
            P[rp] = p0;
        }
        ks += (unsigned)__popcll(mo);
        if (mo != mi)
            break;
    }
    MEM::sync();
    const unsigned Lk = ks < tL ? LP[f + 1 + ks] : 0xffffffffu;
    const unsigned Rk = ks > 0 ? RP[f + 1 + tR - ks] : 0xffffffffu;
    return Lk < Rk ? Lk : Rk;
}

/* __introsort_loop on R (range bits) with P (the points) moved along; LP / RP: work arrays of n entries each.  Leaves, for
 * every element j, the segment [LP[j], RP[j]) the final insertion sort will keep it in. */
template <class MEM>
__device__ __forceinline__ void urf_tie_introsort_loop(unsigned n, typename MEM::rptr R, typename MEM::iptr P, typename MEM::iptr LP,
                                                       typename MEM::iptr RP, int* stk)
{
    const unsigned lane = threadIdx.x;
    const unsigned limit = 2u * (31u - (unsigned)__clz((int)n));
    int top = 0;
    unsigned f = 0, l = n, d = 0;
    for (;;) {
        while (l - f > 16u) {
            if (d == limit) {
                if (lane == 0)
                    urf_tie_heap_sort<typename MEM::rptr, typename MEM::iptr>(R, P, f, l);
                for (unsigned j = f + lane; j < l; j += 64) {   /* sorted: every element a segment of its own */
                    LP[j] = j;
                    RP[j] = j + 1;
                }
                MEM::sync();
                f = l;
                break;
            }
            d++;
            const unsigned cut = urf_tie_partition<MEM>(R, P, LP, RP, f, l);
            if (l - cut > 16u) {   /* __introsort_loop(cut, last, depth_limit): later */
                if (lane == 0) {
                    stk[3 * top] = (int)cut;
                    stk[3 * top + 1] = (int)l;
                    stk[3 * top + 2] = (int)d;
                }
                top++;
            } else {
                const unsigned j = cut + lane;
                if (j < l) {
                    LP[j] = cut;
                    RP[j] = l;
                }
            }
            l = cut;
        }
        if (l > f) {   /* at most 16 elements: left to the final insertion sort */
            const unsigned j = f + lane;
            if (j < l) {
                LP[j] = f;
                RP[j] = l;
            }
        }
        if (top == 0)
            break;
        top--;
        urf_wave_lds_sync();
        f = (unsigned)stk[3 * top];
        l = (unsigned)stk[3 * top + 1];
        d = (unsigned)stk[3 * top + 2];
    }
    MEM::sync();
}

/* WHICH point std::sort leaves at sorted index `target`: the partitions of the segment that holds that index, and only
 * those -- what the introsort loop does to the other side of a cut never reaches it (the second pass of k_star_ties needs one
 * point, not the order: n + n / 2 + n / 4 ... elements looked at instead of n log n).  Returns the point (P's entry). */
template <class MEM>
__device__ __forceinline__ unsigned urf_tie_select(unsigned n, unsigned target, typename MEM::rptr R, typename MEM::iptr P, typename MEM::iptr LP,
                                                   typename MEM::iptr RP)
{
    const unsigned lane = threadIdx.x;
    const unsigned limit = 2u * (31u - (unsigned)__clz((int)n));
    unsigned f = 0, l = n, d = 0;
    while (l - f > 16u) {
        if (d == limit) {   /* __partial_sort: the segment is sorted when it returns */
            if (lane == 0)
                urf_tie_heap_sort<typename MEM::rptr, typename MEM::iptr>(R, P, f, l);
            MEM::sync();
            return P[target];
        }
        d++;
        const unsigned cut = urf_tie_partition<MEM>(R, P, LP, RP, f, l);
        if (target < cut)
            l = cut;
        else
            f = cut;
    }
    /* __final_insertion_sort keeps the segment's elements inside it, stably: the one whose place is `target` */
    const unsigned j = f + lane;
    unsigned rank = 0xffffffffu;
    if (j < l) {
        const unsigned v = R[j];
        rank = f;
        for (unsigned i = f; i < l; i++) {
            const unsigned u = R[i];
            rank += (u < v || (u == v && i < j)) ? 1u : 0u;
        }
    }
    const unsigned long long m = __ballot(rank == target);   /* exactly one lane */
    return P[f + (unsigned)__ffsll((long long)m) - 1u];
}

/* __final_insertion_sort: stable, and every element stays inside its segment -- its final place */
template <class MEM>
__device__ __forceinline__ unsigned urf_tie_final_rank(unsigned j, typename MEM::rptr R, typename MEM::iptr LP, typename MEM::iptr RP)
{
    const unsigned sa = LP[j], se = RP[j], v = R[j];
    unsigned rank = sa;
    for (unsigned i = sa; i < se; i++) {
        const unsigned u = R[i];
        rank += (u < v || (u == v && i < j)) ? 1u : 0u;
    }
    return rank;
}

template <class MEM, bool POST>
__device__ __forceinline__ void urf_tie_sector_body(const urf_kargs& a, const urf_dev_params& dp, unsigned sk, unsigned s, unsigned k, unsigned n,
                                                    unsigned hit_i, typename MEM::rptr R, typename MEM::iptr P, typename MEM::iptr LP,
                                                    typename MEM::iptr RP, int* stk)
{
    const unsigned lane = threadIdx.x, K = (unsigned)dp.p.sectors;
    const unsigned sb = urf_sbase(a, s), base = sb + a.sec_off[(size_t)s * (K + 1) + k];
    const urf_sec_run two = a.sec_run[sk];
    const bool simple = two.nruns <= 2;
    /* the sector in the reference's order (ROI order = tiles in order, input order inside): range bits and the point's
     * address in the sector-sorted arrays */
    {
        unsigned nruns = 0;
        if (!simple) {
            unsigned off, len;
            urf_scan_range(a, s, off, len);
            nruns = urf_sector_runs(a, s, K, k, (len + URF_TILE - 1) / URF_TILE, urf_sector_run_row(a, s, K, k, 0), (unsigned*)LP, (unsigned*)RP);
            MEM::sync();
        }
        unsigned r = 0;
        for (unsigned i = lane; i < n; i += 64) {
            unsigned adr = i < two.c0 ? two.a0 + i : two.a1 + (i - two.c0);
            if (!simple) {
                while (r + 1 < nruns && i >= LP[r + 1])
                    r++;
                adr = RP[r] + i;
            }
            R[i] = urf_fbits(a.sr[sb + adr]);
            P[i] = adr;
        }
    }
    MEM::sync();
    if constexpr (POST) {
        /* behind the walk: the point std::sort leaves at the index the walk stopped at (urf_walk_report's conversion of a
         * point's address in the sector-sorted arrays into its place in the ring-major ones) */
        const unsigned adr = urf_tie_select<MEM>(n, hit_i, R, P, LP, RP);
        const unsigned sl = a.sslot[sb + adr];
        const unsigned v = sl == URF_SLOT_NONE ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
        const int hit = urf_walk_slot_to_ring_pos(a, s, (unsigned)dp.p.channels, v);
        if (lane == 0) {
            a.star_hit[sk] = hit;
            a.star_first[sk] = 0;   /* (the flag is consumed) */
        }
        MEM::sync();
        return;
    }
    urf_tie_introsort_loop<MEM>(n, R, P, LP, RP, stk);
    /* sorted: RP = addresses, P = range bits, R = heights */
    for (unsigned j = lane; j < n; j += 64)
        LP[j] = urf_tie_final_rank<MEM>(j, R, LP, RP);
    MEM::sync();
    for (unsigned j = lane; j < n; j += 64)
        RP[LP[j]] = P[j];            /* address of the i-th point in sorted order */
    MEM::sync();
    for (unsigned j = lane; j < n; j += 64)
        P[LP[j]] = R[j];             /* its range bits */
    MEM::sync();
    for (unsigned i = lane; i < n; i += 64)
        R[i] = __float_as_uint(a.sz[sb + RP[i]]);   /* its height */
    MEM::sync();
    /* ---- what the sort kernels publish: slopes, distance terms, the point's position / ring-sorted index ---- */
    const float slope_param = dp.slope_param, kdist = dp.p.kdist_param;
    const bool fmt16 = simple && n <= URF_STAR_MID_CAP_;   /* (urf_walk_report) */
    unsigned first = n;
    for (unsigned i0 = 0; i0 < n; i0 += 64) {
        const unsigned i = i0 + lane;
        bool hit = false;
        if (i < n) {
            float slp = 0.f, g = 0.f;
            if (i >= 1) {
                const float ax = __uint_as_float(P[i - 1]), bx = __uint_as_float(P[i]);
                slp = (__uint_as_float(R[i]) - __uint_as_float(R[i - 1])) / (bx - ax);   /* star_shaped_search.cpp:27-30 */
                g = (bx - ax) * kdist;
                hit = slp > slope_param;
            }
            const unsigned adr = RP[i];
            if (fmt16) {
                a.ssrt16[base + i] = (uint16_t)((two.nruns == 2 && adr >= two.a1) ? two.c0 + (adr - two.a1) : adr - two.a0);
            } else {
                const unsigned sl = a.sslot[sb + adr];
                a.ssrt[base + i] = sl == URF_SLOT_NONE ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;   /* (RP may BE this stretch of ssrt: own element) */
            }
            a.wsg[base + i] = urf_sg{ slp, g };
        }
        const unsigned long long mh = __ballot(hit);
        if (mh) {
            first = i0 + (unsigned)__ffsll((long long)mh) - 1u;
            break;   /* the walk can never pass the first slope above the threshold */
        }
    }
    if (lane == 0)
        a.star_first[sk] = (first < n - 1 ? first : n - 1) | URF_TIE_DONE;   /* (the walk need not ask for the second pass) */
    MEM::sync();
}

/* POST = false: in front of the walk, the sectors the sort kernels flagged (URF_TIE_FLAG); POST = true: behind it, the sectors
 * in which the walk stopped at a point with a twin behind it (URF_TIE_POST | index): sorted as std::sort does, the point that
 * stands at that index is reported instead (the walk itself does not change: the twins have one range and one height).
 * One wave per sector, persistent over the list the sort / walk kernels appended the sector to.  (Until the twins were told
 * apart, EVERY sector of a sensor's sweep came through here -- flags scanned instead of a list appended to by 368 000 atomics
 * on one counter -- and a second instance with 16-bit index arrays, 5 KB of LDS and six waves per SIMD carried the load:
 * 65 k -> 135 k sweeps/s; with one sector in a hundred left, one instance with 32 KB does: the time is the latency of one
 * wave's chain.) */
/* Two instances per pass (r6): sectors of at most URF_TIE_SMALL points -- every sector of a 64 x 2048 sweep -- in 8 KB of LDS per
 * wave, as many waves resident as the list of a 1024-sweep batch has sectors (one round: a pass is the latency of ONE wave's
 * chain, ~40 us per sector; with 32 KB per wave and four waves per CU such a batch took two or three rounds per pass); the larger
 * ones as before. */
#define URF_TIE_SMALL 512u
template <bool POST, unsigned CAP>
__global__ __launch_bounds__(64) void k_star_ties(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned W[4 * CAP];   /* R, P, LP, RP */
    __shared__ int stk[3 * 64];
    const unsigned count = a.star_count[POST ? 5 : 4];   /* (uniform; 0 for every tie-free sweep: the kernel returns at once) */
    const unsigned K = (unsigned)dp.p.sectors;
    const uint32_t* const list = POST ? a.tie_post : a.tie_list;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        {
            const unsigned sk = list[w];
            const unsigned sf = a.star_first[sk];
            if (!(sf & (POST ? URF_TIE_POST : URF_TIE_FLAG)))
                continue;
            const unsigned hit_i = sf & URF_TIE_INDEX;   /* (POST) */
            const unsigned s = sk / K, k = sk % K;
            if (a.info[s].status != URF_OK)
                continue;   /* (a void scan's entries are leftovers of an earlier call) */
            const unsigned n = a.sec_cnt[sk];
            if (n < 2)
                continue;
            if (CAP == URF_TIE_SMALL ? n > URF_TIE_SMALL : n <= URF_TIE_SMALL)
                continue;   /* (the other instance's) */
            if (n <= CAP) {
                urf_tie_sector_body<urf_tie_lds, POST>(a, dp, sk, s, k, n, hit_i, W, W + CAP, W + 2 * CAP, W + 3 * CAP, stk);
            } else {
                const unsigned base = urf_sbase(a, s) + a.sec_off[(size_t)s * (K + 1) + k];
                urf_tie_sector_body<urf_tie_glb, POST>(a, dp, sk, s, k, n, hit_i, (unsigned*)a.big_r + base, a.big_i + base,
                                                       (unsigned*)a.big_z + base, a.ssrt + base, stk);
            }
        }
    }
}

/* star_shaped_search.cpp:123-149, one LANE per (sector, scan), one wave per 64
 * sectors.  wsg = (slope, distance term) pairs in sorted order; the walk of a
 * sector visits i = 1..star_first.  A lane reading its own sector directly
 * would touch 64 different cache lines per load, so the wave fetches the next
 * 16 steps of all its sectors cooperatively (16 consecutive pairs = one line
 * per sector) into LDS and every lane then reads its own row.
 *
 * With one wave per SIMD (a single sweep: six waves on the whole device) every
 * instruction of the chain costs 5-8 cycles (profiles/r4_valubench.txt, W = 1),
 * so the chunk is written for the fewest instructions: the running mean and
 * deviation advance unconditionally (what follows a sector's curb point or its
 * last step is never looked at), the sixteen hit tests leave as lane masks and
 * the first one is picked afterwards; the wave-uniform (i - 1, 1 / i) come from a
 * table in global memory through scalar loads (a.walk_tab); a NaN slope shows as
 * a NaN mean at the end of the chunk, which is then walked again by the general
 * version from the state it started with. */
#define URF_WALK_CHUNK 16
/* the running state of one sector's walk (star_shaped_search.cpp:123-149) */
struct urf_walk_state {
    float avg, dev, nan;
    unsigned hit_i;   /* sorted index of the sector's curb point, 0 = none */
    unsigned lim;     /* last index this lane still walks; 0 = done */
};
/* a * b, rounded once, out of reach of the SLP vectoriser (which pairs the multiplications of the hit test into
 * v_pk_mul_f32 and pays two v_mov per pair to line the operands up) */
__device__ __forceinline__ float urf_mul_f32(float x, float y)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float urf_add_f32(float x, float y)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
/* x + |y - z| in two instructions, each rounded once */
__device__ __forceinline__ float urf_add_absdiff_f32(float x, float y, float z)
{
    float d, r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(y), "v"(z));
    asm("v_add_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(x), "v"(d));
    return r;
}
/* One chunk of URF_WALK_CHUNK steps from index c0 (wave-uniform) of every lane's sector, no NaN slope so far in any
 * sector of the wave: pairs sg[], the wave-uniform (i - 1, 1 / i) in wu[].  ABOVE: every step of the chunk is past
 * dmin_param (and c0 != 0), which leaves the test without its wave-uniform part.  The hit tests do not look at the
 * sector's end: the first raw hit either lies inside the walk, and is the walk's, or nothing inside does.  Returns
 * false (state untouched) if a lane that is still walking ended the chunk with a NaN mean. */
template <bool ABOVE>
__device__ __forceinline__ bool urf_walk_chunk_fast(urf_walk_state& w_, unsigned c0, const urf_sg (&sg)[URF_WALK_CHUNK],
                                                    const urf_wu* __restrict__ wu, float kdev, float slope_param, int dmin)
{
    float avg = w_.avg, dev = w_.dev;
    const unsigned lim = w_.lim;
    bool h[URF_WALK_CHUNK];
#pragma unroll
    for (unsigned j = 0; j < URF_WALK_CHUNK; j++) {
        const unsigned i = c0 + j;
        h[j] = false;
        if (!ABOVE && j == 0 && c0 == 0)
            continue;   /* the walk starts at 1 */
        const float slp = sg[j].slp;
        const float w = wu[j].w, u = wu[j].u;                  /* (float)i - 0 - 1 and 1 / (float)i */
        float na = avg * w;                                    /* star_shaped_search.cpp:135-140 */
        na = na + slp;
        na = na * u;
        float nd = dev * w;
        nd = nd + __builtin_fabsf(slp - na);
        nd = nd * u;
        avg = na;
        dev = nd;
        const bool dyn = urf_mul_f32(urf_mul_f32(urf_mul_f32(slp, slp) - urf_mul_f32(na, na), kdev), sg[j].g) > nd;
        h[j] = (slp > slope_param) | ((ABOVE || (int)i > dmin) & dyn);   /* :142-143 */
    }
    if (__any(lim != 0u && avg != avg))
        return false;
    unsigned hit = 0;
#pragma unroll
    for (int j = URF_WALK_CHUNK - 1; j >= 0; j--)
        hit = h[j] ? c0 + (unsigned)j : hit;                   /* the first one: :146 */
    w_.avg = avg;
    w_.dev = dev;
    if (hit && hit <= lim) {
        w_.hit_i = hit;
        w_.lim = 0;
    }
    return true;
}
/* The same chunk with NaN slopes in it (or before it): they are counted and skipped, star_shaped_search.cpp:131-132. */
__device__ __forceinline__ void urf_walk_chunk_general(urf_walk_state& w_, unsigned c0, const urf_sg* sg /* the lane's row of the LDS tile */, float kdev,
                                                       float slope_param, int dmin)
{
    float avg = w_.avg, dev = w_.dev, nan = w_.nan;
    unsigned hit_i = w_.hit_i, lim = w_.lim;
#pragma unroll 1
    for (unsigned j = 0; j < URF_WALK_CHUNK; j++) {
        const unsigned i = c0 + j;
        const bool active = i >= 1 && i <= lim;
        const float slp = sg[j].slp;
        if (active) {
            if (slp != slp) {
                nan += 1.0f;                               /* :131-132 */
            } else {
                const float w = (float)(int)i - nan - 1.0f;
                const float u = 1.0f / ((float)(int)i - nan);
                avg *= w;
                avg += slp;
                avg *= u;
                dev *= w;
                dev += __builtin_fabsf(slp - avg);
                dev *= u;
            }
        }
        const bool h = slp > slope_param ||
                       ((int)i > dmin && (slp * slp - avg * avg) * kdev * sg[j].g > dev);
        if (active && h) {
            hit_i = i;
            lim = 0;
        }
    }
    w_.avg = avg;
    w_.dev = dev;
    w_.nan = nan;
    w_.hit_i = hit_i;
    w_.lim = lim;
}

/* The curb point of sector k (sorted index hit_i, 0 = none), reported where k_ring looks for it: as a position
 * in the ring-major arrays (-1: none, or on no ring).  Its tile-local ring-sorted index t * URF_TILE + slot: for a
 * sector of at most two runs the sort left the point's position inside the sector (ssrt16), which sec_run turns
 * into its place in the sector-sorted arrays, where its slot stands; other sectors hold the index itself (ssrt).
 * The ring is the run of tile t that contains the slot (bisection in the tile's run table). */
/* tile-local ring-sorted index v = t * URF_TILE + slot (0xffffffff: on no ring) -> the point's position in the ring-major arrays, -1: none */
__device__ __forceinline__ int urf_walk_slot_to_ring_pos(const urf_kargs& a, unsigned s, unsigned C, unsigned v)
{
    int hit = -1;
    if (a.front && a.front_ok[s])   /* a scan of the fused front end (urf_front.hpp): sslot holds the index inside the input tile, k_front_finish wants the input index */
        return (int)v;
    if (v != 0xffffffffu) {
        const unsigned t = v / URF_TILE, j = v % URF_TILE;
        const uint16_t* row = a.troff + ((size_t)s * a.tiles + t) * (C + 1);
        unsigned lo = 0, hi = C;   /* largest c with row[c] <= j (its run is not empty) */
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if ((unsigned)row[mid] <= j)
                lo = mid;
            else
                hi = mid;
        }
        const unsigned p = a.rpre[((size_t)s * C + lo) * (a.tiles + 1) + t] + (j - (unsigned)row[lo]);
        hit = (int)(a.ring_off[(size_t)s * (C + 1) + lo] + p);   /* relative to the scan's scratch base */
    }
    return hit;
}
__device__ __forceinline__ int urf_walk_report(const urf_kargs& a, unsigned s, unsigned K, unsigned C, unsigned k, unsigned n, unsigned base,
                                               unsigned hit_i)
{
    int hit = -1;
    unsigned v = 0xffffffffu;
    if (hit_i && hit_i < n) {   /* (an index behind the sector would read behind its stretch of the arrays) */
        const urf_sec_run two = a.sec_run[(size_t)s * K + k];
        if (two.nruns <= 2 && n <= URF_STAR_MID_CAP_) {
            const unsigned i0 = a.ssrt16[base + hit_i];
            const unsigned adr = i0 < two.c0 ? two.a0 + i0 : two.a1 + (i0 - two.c0);
            const unsigned sl = a.sslot[urf_sbase(a, s) + adr];
            v = sl == URF_SLOT_NONE ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
        } else {
            v = a.ssrt[base + hit_i];
        }
    }
    hit = urf_walk_slot_to_ring_pos(a, s, C, v);
    return hit;
}

/* (float)(i - 1) and 1 / (float)i for every step a walk can take, star_shaped_search.cpp:137; filled once per context */
__global__ __launch_bounds__(256) void k_walk_table(urf_wu* tab, unsigned n)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        tab[i] = i ? urf_wu{ (float)(int)i - 1.0f, 1.0f / (float)(int)i } : urf_wu{ 0.f, 0.f };   /* [0]: no step of a walk; k_star_walk_few runs it as a no-op */
}

typedef urf_sg urf_walk_tile[64][URF_WALK_CHUNK + 2];   /* rows of 36 words: 16-byte reads of a lane's own row without bank conflicts */
/* LDS of the two walk kernels: the chunk of pairs, a row per sector, and the sectors' places */
__shared__ urf_walk_tile walk_tile;
__shared__ unsigned walk_sbase[64], walk_slast[64];

/* One wave walks its 64 sectors from chunk c_start (W = the state in front of it) to the end, fetching for itself:
 * the next 16 steps of all 64 sectors are 16 eight-byte loads in flight, parked in registers until the LDS tile is
 * free again (chunk c+1 loads while chunk c is walked; the walk itself runs out of registers, so one tile suffices
 * -- a second one would halve the resident workgroups).  A step past the sector's last one reads the last one again
 * (the same line: no traffic), never past the sector.  The other waves of the workgroup may have left already: only
 * wave-level ordering is used (k_star_walk is this loop from chunk 0 for a workgroup of one wave). */
__device__ __forceinline__ void urf_walk_sequential(urf_walk_state& W, unsigned c_start, unsigned last, unsigned maxlast, const urf_sg* __restrict__ wsg,
                                                    const urf_wu* __restrict__ tab, unsigned lane, float kdev, float slope_param, int dmin)
{
    urf_walk_tile& tile = walk_tile;
    const unsigned* sbase = walk_sbase;
    const unsigned* slast = walk_slast;
    urf_sg v[16];
    unsigned fo[16], fe[16];   /* the 16 sectors this lane loads for: its own step of the chunk / the sector's last step, as indices into wsg */
#pragma unroll
    for (unsigned r = 0; r < 16; r++) {
        const unsigned sec = r * 4 + (lane >> 4);
        fo[r] = sbase[sec] + (lane & 15);
        fe[r] = sbase[sec] + slast[sec];
    }
    auto fetch = [&](unsigned c0) {
#pragma unroll
        for (unsigned r = 0; r < 16; r++) {
            const unsigned e = fo[r] + c0;
            v[r] = wsg[e < fe[r] ? e : fe[r]];   /* slope and distance term: one 8-byte load */
        }
    };
    auto park = [&]() {
#pragma unroll
        for (unsigned r = 0; r < 16; r++)
            tile[r * 4 + (lane >> 4)][lane & 15] = v[r];
    };
    fetch(c_start);
    park();
    urf_wave_lds_sync();
    for (unsigned c0 = c_start; c0 <= maxlast; c0 += URF_WALK_CHUNK) {
        if (!__any(W.lim != 0))
            break;
        const bool more = c0 + URF_WALK_CHUNK <= maxlast;
        if (more)
            fetch(c0 + URF_WALK_CHUNK);
        urf_sg sg[URF_WALK_CHUNK];
        {
            const float4* row = reinterpret_cast<const float4*>(&tile[lane][0]);
#pragma unroll
            for (unsigned j = 0; j < URF_WALK_CHUNK / 2; j++) {
                const float4 q = row[j];
                sg[2 * j] = urf_sg{ q.x, q.y };
                sg[2 * j + 1] = urf_sg{ q.z, q.w };
            }
        }
        bool walked = false;
        if (!__any(W.nan != 0.0f))
            walked = (c0 != 0 && (int)c0 > dmin) ? urf_walk_chunk_fast<true>(W, c0, sg, tab + c0, kdev, slope_param, dmin)
                                                 : urf_walk_chunk_fast<false>(W, c0, sg, tab + c0, kdev, slope_param, dmin);
        if (!walked)
            urf_walk_chunk_general(W, c0, &tile[lane][0], kdev, slope_param, dmin);
        if (c0 + URF_WALK_CHUNK > last)
            W.lim = 0;
        urf_wave_lds_sync();   /* the tile has been read: it may take the next chunk */
        if (more)
            park();
        urf_wave_lds_sync();
    }
}

/* what every wave of a walk kernel starts with: its 64 sectors' places (sbase / slast in LDS, by wave 0) and the longest walk among them */
struct urf_walk_sectors {
    unsigned k, n, base, last, maxlast, sf;   /* sf: star_first with its URF_TIE_* flags */
    bool have;
};
/* The walk stopped at sorted index hit_i.  If the point there has a twin behind it (same range, same height: the sort kernels
 * let such pairs pass, URF_TIE_*), WHICH of them stands at that index is a matter of std::sort's order: the sector goes to
 * k_star_ties' second pass.  "The next point has the same range" = its distance term is zero (wsg holds it up to the sort's
 * last index; for the last index itself the sort kernel left URF_TIE_NEXT).  (A distance term that is zero for another reason
 * -- kdist == 0, underflow -- only costs the second pass a sector it did not have to look at.) */
__device__ __forceinline__ void urf_walk_twins(const urf_kargs& a, unsigned s, unsigned K, unsigned k, unsigned n, unsigned base, unsigned sf,
                                               unsigned hit_i)
{
    if (hit_i == 0 || (sf & URF_TIE_DONE) || hit_i + 1 >= n)
        return;
    const float gn = a.wsg[base + hit_i + 1].g;
    /* (not greater, not smaller: a NaN -- 0 * inf with a non-finite kdist_param -- counts as a twin and costs one sector of the second pass) */
    const bool twin = hit_i == (sf & URF_TIE_INDEX) ? (sf & URF_TIE_NEXT) != 0u : !(gn > 0.0f || gn < 0.0f);
    if (!twin)
        return;
    a.star_first[(size_t)s * K + k] = URF_TIE_POST | hit_i;
    a.tie_post[atomicAdd(&a.star_count[5], 1u)] = s * K + k;
    if (a.optimistic & URF_OPT_NO_TIES)
        a.info[s].status = URF_STATUS_REDO_TIES;   /* nobody runs the second pass in this launch sequence: once more, with it */
}
__device__ __forceinline__ urf_walk_sectors urf_walk_prologue(const urf_kargs& a, unsigned s, unsigned K, unsigned lane, bool publish)
{
    unsigned* sbase = walk_sbase;
    unsigned* slast = walk_slast;
    urf_walk_sectors q;
    q.k = blockIdx.x * 64 + lane;
    q.have = q.k < K;
    q.n = q.have ? a.sec_cnt[(size_t)s * K + q.k] : 0;
    const unsigned rel = q.have ? a.sec_off[(size_t)s * (K + 1) + q.k] : 0;
    q.base = urf_sbase(a, s) + rel;
    q.sf = q.n >= 2 ? a.star_first[(size_t)s * K + q.k] : 0;
    q.last = q.sf & URF_TIE_INDEX;
    if (publish) {
        sbase[lane] = q.last ? rel : 0u;   /* a sector without a walk reads the scan's first pair, whatever it is */
        slast[lane] = q.last;
    }
    unsigned maxlast = q.last;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned w = __shfl_xor(maxlast, o);
        maxlast = w > maxlast ? w : maxlast;
    }
    q.maxlast = (unsigned)__builtin_amdgcn_readfirstlane((int)maxlast);
    return q;
}

__global__ __launch_bounds__(64) void k_star_walk(urf_kargs a, urf_dev_params dp)
{
    urf_walk_tile& tile = walk_tile;
    unsigned* sbase = walk_sbase;
    unsigned* slast = walk_slast;
    const unsigned K = (unsigned)dp.p.sectors;
    const unsigned s = blockIdx.y, lane = threadIdx.x;
    const unsigned k = blockIdx.x * 64 + lane;
    if (a.info[s].status != URF_OK)
        return;
    const unsigned C = (unsigned)dp.p.channels;
    const bool have = k < K;
    const unsigned n = have ? a.sec_cnt[(size_t)s * K + k] : 0;
    const unsigned rel = have ? a.sec_off[(size_t)s * (K + 1) + k] : 0;
    const unsigned base = urf_sbase(a, s) + rel;
    const unsigned sf = n >= 2 ? a.star_first[(size_t)s * K + k] : 0;
    const unsigned last = sf & URF_TIE_INDEX;
    sbase[lane] = last ? rel : 0u;   /* a sector without a walk reads the scan's first pair, whatever it is */
    slast[lane] = last;
    unsigned maxlast = last;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned w = __shfl_xor(maxlast, o);
        maxlast = w > maxlast ? w : maxlast;
    }
    maxlast = (unsigned)__builtin_amdgcn_readfirstlane((int)maxlast);
    __syncthreads();

    /* urf_walk_sequential from chunk 0, written out: the compiler keeps this form in 149 registers (three waves
     * per SIMD) and the inlined function in 211 */
    const urf_sg* __restrict__ wsg = a.wsg + urf_sbase(a, s);
    urf_sg v[16];
    unsigned fo[16], fe[16];
#pragma unroll
    for (unsigned r = 0; r < 16; r++) {
        const unsigned sec = r * 4 + (lane >> 4);
        fo[r] = sbase[sec] + (lane & 15);
        fe[r] = sbase[sec] + slast[sec];
    }
    auto fetch = [&](unsigned c0) {
#pragma unroll
        for (unsigned r = 0; r < 16; r++) {
            const unsigned e = fo[r] + c0;
            v[r] = wsg[e < fe[r] ? e : fe[r]];   /* slope and distance term: one 8-byte load */
        }
    };
    auto park = [&]() {
#pragma unroll
        for (unsigned r = 0; r < 16; r++)
            tile[r * 4 + (lane >> 4)][lane & 15] = v[r];
    };

    const float kdev = dp.p.kdev_param, slope_param = dp.slope_param;
    const int dmin = dp.p.dmin_param;
    urf_walk_state W = { 0.f, 0.f, 0.f, 0u, last };
    fetch(0);
    park();
    __syncthreads();
    for (unsigned c0 = 0; c0 <= maxlast; c0 += URF_WALK_CHUNK) {
        if (!__any(W.lim != 0))
            break;
        const bool more = c0 + URF_WALK_CHUNK <= maxlast;
        if (more)
            fetch(c0 + URF_WALK_CHUNK);
        urf_sg sg[URF_WALK_CHUNK];
        {
            const float4* row = reinterpret_cast<const float4*>(&tile[lane][0]);
#pragma unroll
            for (unsigned j = 0; j < URF_WALK_CHUNK / 2; j++) {
                const float4 q = row[j];
                sg[2 * j] = urf_sg{ q.x, q.y };
                sg[2 * j + 1] = urf_sg{ q.z, q.w };
            }
        }
        bool walked = false;
        if (!__any(W.nan != 0.0f))
            walked = (c0 != 0 && (int)c0 > dmin) ? urf_walk_chunk_fast<true>(W, c0, sg, a.walk_tab + c0, kdev, slope_param, dmin)
                                                 : urf_walk_chunk_fast<false>(W, c0, sg, a.walk_tab + c0, kdev, slope_param, dmin);
        if (!walked)
            urf_walk_chunk_general(W, c0, &tile[lane][0], kdev, slope_param, dmin);
        if (c0 + URF_WALK_CHUNK > last)
            W.lim = 0;
        __syncthreads();   /* the tile has been read: it may take the next chunk (one wave per workgroup: no s_barrier in the code) */
        if (more)
            park();
        __syncthreads();
    }
    const int hit = urf_walk_report(a, s, K, C, k, n, base, W.hit_i);
    if (have) {
        a.star_hit[(size_t)s * K + k] = hit;
        urf_walk_twins(a, s, K, k, n, base, sf, W.hit_i);
    }
}

/* The same walk for a handful of sweeps (the callback path: one), where the device is empty and the time is the
 * chain of one wave's instructions: a wave that has its SIMD to itself issues an independent instruction every 4.7
 * cycles, a dependent one every 9.3, a mix like the walk's every 7 (tools/bench_micro/lonewave.hip,
 * profiles/r4_lonewave.txt): the two chains of a step (mean: three instructions, deviation: four) take 49 cycles, the
 * whole step of k_star_walk 130.  Five waves share the 64 sectors' chunk instead, one chunk apart, one barrier per
 * chunk, tile / X / hits double-buffered:
 *   wave 0       nothing but the chains of chunk c: running mean and deviation after each step, left in LDS (X);
 *   waves 1..4   the hit tests of chunk c - 1, four steps each, from X and the pairs they kept from the tile; a
 *                quarter of the loads of chunk c + 2 and of the parking of chunk c + 1 each;
 *   wave 1       also keeps the walks' state: merges the published hits (chunk c - 2) and tells the others through
 *                ctl when no sector is walking any more (they leave one chunk later, all in the same iteration).
 * A NaN mean in a sector with a walk (a NaN slope: star_shaped_search.cpp:131-132) ends the pipeline: wave 0 goes on
 * alone from the chunk it appeared in, with the state it had in front of it and the hits merged so far
 * (urf_walk_sequential). */
#define URF_WALK_FEW_THREADS 320
#ifndef URF_WALK_FEW_SCANS
#define URF_WALK_FEW_SCANS 32u
#endif
/* The hit tests of four steps, stage by stage: the opaque multiplications stay in the order they are written in, and a
 * lone wave waits 9 cycles for a result it needs at once, 5 for one it needs a few instructions later. */
template <bool ABOVE>
__device__ __forceinline__ unsigned urf_walk_tests_quarter(unsigned i0, const urf_sg (&sg)[4], const float4 (&x)[2], float kdev, float slope_param, int dmin)
{
    float t[4], q[4];
#pragma unroll
    for (unsigned j = 0; j < 4; j++)
        t[j] = urf_mul_f32(sg[j].slp, sg[j].slp);
#pragma unroll
    for (unsigned j = 0; j < 4; j++) {
        const float na = (j & 1) ? x[j >> 1].z : x[j >> 1].x;
        q[j] = urf_mul_f32(na, na);
    }
#pragma unroll
    for (unsigned j = 0; j < 4; j++)
        t[j] = t[j] - q[j];
#pragma unroll
    for (unsigned j = 0; j < 4; j++)
        t[j] = urf_mul_f32(t[j], kdev);
#pragma unroll
    for (unsigned j = 0; j < 4; j++)
        t[j] = urf_mul_f32(t[j], sg[j].g);
    bool h[4];
#pragma unroll
    for (unsigned j = 0; j < 4; j++) {
        const unsigned i = i0 + j;
        const float nd = (j & 1) ? x[j >> 1].w : x[j >> 1].y;
        h[j] = (ABOVE || i != 0) & ((sg[j].slp > slope_param) | ((ABOVE || (int)i > dmin) & (t[j] > nd)));   /* :142-143; the walk starts at 1 */
    }
    unsigned hit = 0;
#pragma unroll
    for (int j = 3; j >= 0; j--)
        hit = h[j] ? i0 + (unsigned)j : hit;
    return hit;
}

__global__ __launch_bounds__(URF_WALK_FEW_THREADS) void k_star_walk_few(urf_kargs a, urf_dev_params dp)
{
    const unsigned* sbase = walk_sbase;
    const unsigned* slast = walk_slast;
    __shared__ urf_walk_tile tile2;                                 /* chunks c odd (walk_tile: c even) */
    __shared__ __attribute__((aligned(16))) float X[2][64][2 * URF_WALK_CHUNK + 4];   /* (mean, deviation) after each step of the chunk, a row per sector */
    __shared__ __attribute__((aligned(16))) unsigned hitb[2][64][4];   /* first raw hit of each quarter of the chunk, 0 = none */
    __shared__ unsigned ctl[2], nan_at, fin_hit[64], fin_lim[64];
    const unsigned K = (unsigned)dp.p.sectors;
    const unsigned s = blockIdx.y, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    /* (ONE decision for the workgroup: another workgroup of the scan may void it -- urf_walk_twins, URF_STATUS_REDO_TIES -- between the
     * loads of this one's five waves, and a wave that carried on alone would read what nobody wrote) */
    if (__syncthreads_or(a.info[s].status != URF_OK))
        return;
    const urf_walk_sectors q = urf_walk_prologue(a, s, K, lane, wave == 0);
    if (threadIdx.x == 0) {
        nan_at = 0xffffffffu;
        ctl[0] = 1u;
        ctl[1] = 1u;
    }
    __syncthreads();
    const urf_sg* __restrict__ wsg = a.wsg + urf_sbase(a, s);
    const float kdev = dp.p.kdev_param, slope_param = dp.slope_param;
    const int dmin = dp.p.dmin_param;
    const unsigned last = q.last, nchunks = q.maxlast / URF_WALK_CHUNK + 1;
    const unsigned b = wave ? wave - 1u : 0u;   /* waves 1..4: quarter b of every chunk */

    /* waves 1..4: four of the sixteen loads of a chunk each */
    urf_sg v[4];
    unsigned fo[4], fe[4];
#pragma unroll
    for (unsigned r = 0; r < 4; r++) {
        const unsigned sec = (4 * b + r) * 4 + (lane >> 4);
        fo[r] = sbase[sec] + (lane & 15);
        fe[r] = sbase[sec] + slast[sec];
    }
    auto fetch = [&](unsigned c0) {
#pragma unroll
        for (unsigned r = 0; r < 4; r++) {
            const unsigned e = fo[r] + c0;
            v[r] = wsg[e < fe[r] ? e : fe[r]];
        }
    };
    auto park = [&](urf_walk_tile& t) {
#pragma unroll
        for (unsigned r = 0; r < 4; r++)
            t[(4 * b + r) * 4 + (lane >> 4)][lane & 15] = v[r];
    };
    if (wave) {
        fetch(0);
        park(walk_tile);
        if (nchunks > 1)
            fetch(URF_WALK_CHUNK);
    }
    /* wave 0: the (i - 1, 1 / i) of its next chunk, asked for as soon as the previous chunk's are used up (they come from
     * the far side of the L2); [0] = (0, 0), which with a slope of 0 leaves the state at 0: step 0 needs no exception */
    urf_wu wu[URF_WALK_CHUNK];
#pragma unroll
    for (unsigned j = 0; j < URF_WALK_CHUNK; j++)
        wu[j] = a.walk_tab[j];
    float avg = 0.f, dev = 0.f;      /* wave 0: the state behind the chunk it walked last */
    unsigned lim = last, hit_i = 0;  /* wave 1: the walks' state */
    urf_sg sgp[4];                   /* waves 1..4: their steps' pairs of the chunk they test next */
#pragma unroll
    for (unsigned j = 0; j < 4; j++)
        sgp[j] = urf_sg{ 0.f, 0.f };
    unsigned stop = 0xffffffffu;
    for (unsigned c = 0;; c++) {
        __syncthreads();   /* tile[c & 1] holds chunk c, X[(c - 1) & 1] chunk c - 1, hitb[(c - 1) & 1] the hits of chunk c - 2 */   /* tile[c & 1] holds chunk c, X[(c - 1) & 1] chunk c - 1, hitb[(c - 1) & 1] the hits of chunk c - 2 */
        /* (wave 0 may write nan_at = c while a late wave is still here: only a word of an EARLIER iteration counts, so that
         * all five see the same thing in the same iteration) */
        stop = nan_at;
        stop = stop < c ? stop : 0xffffffffu;
        const unsigned walking = ctl[(c - 1) & 1];   /* wave 1's word of the iteration before */
        if (wave == 1 && c >= 2) {
            const uint4 h = *reinterpret_cast<const uint4*>(&hitb[(c - 1) & 1][lane][0]);
            const unsigned hit = h.x ? h.x : h.y ? h.y : h.z ? h.z : h.w;
            if (lim != 0 && hit != 0 && hit <= lim) {
                hit_i = hit;
                lim = 0;
            }
            if ((c - 1) * URF_WALK_CHUNK > last)
                lim = 0;
        }
        if (!walking || stop != 0xffffffffu || c > nchunks + 2)
            break;
        if (wave == 1) {
            const unsigned alive = __any(lim != 0) ? 1u : 0u;
            if (lane == 0)
                ctl[c & 1] = alive;
        }
        urf_walk_tile& tc = (c & 1) ? tile2 : walk_tile;
        if (wave == 0) {
            if (c < nchunks) {
                float sl[URF_WALK_CHUNK];
                {
                    const float4* row = reinterpret_cast<const float4*>(&tc[lane][0]);
#pragma unroll
                    for (unsigned j = 0; j < URF_WALK_CHUNK / 2; j++) {
                        const float4 t = row[j];
                        sl[2 * j] = t.x;
                        sl[2 * j + 1] = t.z;
                    }
                }
                if (c == 0)
                    sl[0] = 0.f;   /* the walk starts at 1 */
                const float avg0 = avg, dev0 = dev;
                float4* xw = reinterpret_cast<float4*>(&X[c & 1][lane][0]);
                float na_[URF_WALK_CHUNK], nd_[URF_WALK_CHUNK];
#pragma unroll
                for (unsigned j = 0; j < URF_WALK_CHUNK; j++) {
                    /* star_shaped_search.cpp:135-140; through urf_mul_f32 & co. because the SLP vectoriser otherwise pairs
                     * mean and deviation into v_pk_* chains (three dependent packed operations and four v_mov per step) */
                    const float na = urf_mul_f32(urf_add_f32(urf_mul_f32(avg, wu[j].w), sl[j]), wu[j].u);
                    const float nd = urf_mul_f32(urf_add_absdiff_f32(urf_mul_f32(dev, wu[j].w), sl[j], na), wu[j].u);
                    avg = na;
                    dev = nd;
                    na_[j] = na;
                    nd_[j] = nd;
                }
#pragma unroll
                for (unsigned j = 0; j < URF_WALK_CHUNK; j++)
                    wu[j] = a.walk_tab[(c + 1) * URF_WALK_CHUNK + j];   /* the table is longer than any walk by two chunks */
#pragma unroll
                for (unsigned j = 0; j < URF_WALK_CHUNK / 2; j++)
                    xw[j] = make_float4(na_[2 * j], nd_[2 * j], na_[2 * j + 1], nd_[2 * j + 1]);
                /* (a sector that has found its curb point walks on over real pairs, one without a walk over whatever the
                 * scan's first pair holds: only the former's mean says anything) */
                if (__any(last != 0u && avg != avg)) {
                    avg = avg0;
                    dev = dev0;
                    if (lane == 0)
                        nan_at = c;
                }
            }
        } else {
            urf_sg sgn[4];
            float4 x[2];
            {
                const float4* row = reinterpret_cast<const float4*>(&tc[lane][4 * b]);
                const float4* xr = reinterpret_cast<const float4*>(&X[(c - 1) & 1][lane][8 * b]);
#pragma unroll
                for (unsigned j = 0; j < 2; j++) {
                    const float4 t = row[j];
                    sgn[2 * j] = urf_sg{ t.x, t.y };
                    sgn[2 * j + 1] = urf_sg{ t.z, t.w };
                    x[j] = xr[j];
                }
            }
            if (c + 1 < nchunks) {
                park((c & 1) ? walk_tile : tile2);   /* chunk c + 1 */
                if (c + 2 < nchunks)
                    fetch((c + 2) * URF_WALK_CHUNK);
            }
            if (c >= 1 && c <= nchunks) {
                const unsigned i0 = (c - 1) * URF_WALK_CHUNK + 4 * b;
                const unsigned hit = (i0 != 0 && (int)i0 > dmin) ? urf_walk_tests_quarter<true>(i0, sgp, x, kdev, slope_param, dmin)
                                                                 : urf_walk_tests_quarter<false>(i0, sgp, x, kdev, slope_param, dmin);
                hitb[c & 1][lane][b] = hit;
            }
#pragma unroll
            for (unsigned j = 0; j < 4; j++)
                sgp[j] = sgn[j];
        }
    }
    /* all five left in the same iteration: the walks' state goes from wave 1 to wave 0 */
    if (wave == 1) {
        fin_hit[lane] = hit_i;
        fin_lim[lane] = lim;
    }
    __syncthreads();
    if (wave != 0)
        return;
    urf_walk_state W = { avg, dev, 0.f, fin_hit[lane], fin_lim[lane] };
    if (stop != 0xffffffffu && __any(W.lim != 0))
        urf_walk_sequential(W, stop * URF_WALK_CHUNK, last, q.maxlast, wsg, a.walk_tab, lane, kdev, slope_param, dmin);
    const int hit = urf_walk_report(a, s, K, (unsigned)dp.p.channels, q.k, q.n, q.base, W.hit_i);
    if (q.have) {
        a.star_hit[(size_t)s * K + q.k] = hit;
        urf_walk_twins(a, s, K, q.k, q.n, q.base, q.sf, W.hit_i);
    }
}

/* ------------------------------------------------------------------------- */
/* k_ring                                                                      */
/* ------------------------------------------------------------------------- */
/* One workgroup per (ring, scan).  The ring's points (input order) stream
 * through LDS in chunks of 512 with a halo of curbPoints on both sides; every
 * thread owns four points per chunk and evaluates for each
 *   - x_zero for the triple (p - cp/2, p, p - cp/2 + cp) that marks p,
 *   - z_zero for the centre p,
 *   - azimuth and planar range of p,
 * then feeds the per-degree curb tables used by the beam march.
 * The star-shaped hits arrive as ring-major positions (k_scatter stores them in
 * the sector-major records); the ring collects the handful that fall into its
 * own range and turns them into one bit per point of the current chunk.
 *
 * Two point-to-thread mappings.  With the default curbPoints (5) a thread owns
 * four CONSECUTIVE points: the 16 z values around them are read once (four
 * 16-byte LDS loads) and every window maximum of z_zero and both z of x_zero
 * come out of registers; chunks start at a multiple of four in the global
 * ring-major index so that azimuth and flags leave as 16- and 4-byte stores.
 * Any other curbPoints takes the generic mapping (points strided by the
 * workgroup size, windows read from LDS). */
#define URF_RING_PPT 4
#define URF_RING_CHUNK (URF_RING_THREADS * URF_RING_PPT)
#define URF_RING_CAND 1024   /* capacity of the candidate list; flushed when a chunk might not fit */
#define URF_RING_PAD 32   /* LDS slots in front of a chunk, >= URF_MAX_CURB_POINTS, multiple of 4 */
#define URF_RING_HITS 62  /* star-shaped hits of one ring kept in LDS (more: rescanned per chunk) */
#define URF_CURB_LIST 48  /* curb points of one ring handed to k_beams as a list of azimuths (64 x 2048 street sweeps: <= 30); more: per-degree tables */
#define URF_CURB_DENSE 0xffffffffu

/* (the instance for curbPoints == 5 keeps no x / y windows and a shorter candidate list: 13 KB instead of
 * 18, twelve resident workgroups per CU instead of eight) */
template <bool QUADS>
struct urf_ring_shared_t {
    float xs[QUADS ? 4 : URF_RING_CHUNK + 2 * URF_RING_PAD + 4] __attribute__((aligned(16)));
    float ys[QUADS ? 4 : URF_RING_CHUNK + 2 * URF_RING_PAD + 4] __attribute__((aligned(16)));
    /* z window of a chunk; the four-points-per-thread instance has two and alternates, so that the next chunk
     * can be parked while the slower wave still evaluates the current one (one barrier per chunk less) */
    float zsb[QUADS ? 2 : 1][URF_RING_CHUNK + 2 * URF_RING_PAD + 4] __attribute__((aligned(16)));
    int cmin[URF_DEG_CELLS], cmax[URF_DEG_CELLS];
    int q[4];
    unsigned long long maxs;
    unsigned hits[URF_RING_HITS];
    unsigned n_hits, n_runs;
    float curb[URF_CURB_LIST];   /* exact azimuths of the ring's curb points (the first URF_CURB_LIST of them) */
    unsigned n_curb;
    unsigned hb[3][URF_RING_CHUNK / 32];   /* star-hit bit per point of the chunk; three in rotation: the one of chunk c + 1
                                            * is cleared while chunk c is parked and c - 1 may still be read */
    /* quad mapping: the ring's points that need one of the expensive evaluations, compacted */
    static constexpr unsigned CAND = QUADS ? URF_RING_CAND - 128 : URF_RING_CAND;
    unsigned cand[CAND];                   /* ring-relative position | URF_CAND_* << URF_CAND_SHIFT */
    unsigned n_cand;
};
typedef urf_ring_shared_t<false> urf_ring_shared;
#define URF_CAND_SHIFT 28   /* position below, URF_CAND_* above */
#define URF_CAND_XZERO 1u   /* passed the height tests of x_zero: angle test pending */
#define URF_CAND_ZZERO 2u   /* same for z_zero */
#define URF_CAND_EXACT 4u   /* no float approximation of the azimuth (near the x axis, stage capture) */
#define URF_CAND_STAR 8u    /* star-shaped hit */

/* Ring position -> index of the point in the tile-local ring-sorted arrays (rx, ry, rz).  P[t] =
 * points of the ring in the tiles before t (P[ntiles] = n), radd[t] = scratch index of the first
 * point of the ring's run in tile t, minus P[t]: position j of the ring lives at radd[tile(j)] + j.
 * Both tables sit in LDS.  The tile is guessed from the ring's average run length (exact for an
 * organised sweep: every firing adds one point to every ring) and found by bisection otherwise.
 * (A 256-entry inverse table + forward walk instead of the bisection was measured: no gain on a
 * sweep cut by the default region of interest, 9 % slower on a full one -- registers.) */
struct urf_ring_map {
    const unsigned* P;
    const unsigned* radd;
    unsigned ntiles;   /* entries (k_ring lists the ring's non-empty runs only: then "tile" = index of the run) */
    float scale;   /* ntiles / n */
    __device__ __forceinline__ unsigned tile(unsigned j) const
    {
        unsigned t = (unsigned)((float)j * scale);
        t = t < ntiles ? t : ntiles - 1;
        if (P[t] <= j && j < P[t + 1])
            return t;
        /* one entry off (runs of unequal length: a region of interest that cuts firings apart) */
        const unsigned t1 = j < P[t] ? (t ? t - 1 : 0u) : (t + 1 < ntiles ? t + 1 : t);
        if (P[t1] <= j && j < P[t1 + 1])
            return t1;
        unsigned lo = 0, hi = ntiles;   /* largest t with P[t] <= j */
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (P[mid] <= j)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    }
    __device__ __forceinline__ unsigned at(unsigned j) const { return radd[tile(j)] + j; }
};

/* x_zero_method.cpp:30-68 for the triple (j, p, j + cp), j = p - cp / 2, given the cheap height
 * tests passed.  (xj, yj) / (x3, y3): planar coordinates of the points j and j + cp. */
/* (The angle tests and the exact azimuth are NOT inlined: only the few points that pass the cheap
 * height tests get here, and inlined their f64 code dictates the kernel's register allocation --
 * k_ring spilled 52..80 bytes per lane with them inside.) */
__device__ __forceinline__ bool urf_x_zero_angle_body(float nyj, float ny2, float ny3, float angleFilter1, float x_angle_thr, float xj, float yj, float x3, float y3,
                                                      float zj, float pz, float z3)
{
    const double dx = (double)(x3 - xj), dy = (double)(y3 - yj);
    if (!(dx * dx + dy * dy < URF_DIST5_SQ))                                    /* :35-40 */
        return false;
    double u, v;
    u = (double)(ny2 - nyj); v = (double)(pz - zj);
    const float x1 = (float)__builtin_sqrt(u * u + v * v);
    u = (double)(ny3 - ny2); v = (double)(z3 - pz);
    const float x2 = (float)__builtin_sqrt(u * u + v * v);
    u = (double)(ny3 - nyj); v = (double)(z3 - zj);
    const float x3s = (float)__builtin_sqrt(u * u + v * v);
    const double num = (double)x3s * (double)x3s - (double)x1 * (double)x1 - (double)x2 * (double)x2;
    const float den = (-2.0f * x1) * x2;
    float br = (float)(num / (double)den);                                      /* :52 */
    if (br < -1.0f)
        br = -1.0f;
    else if (br > 1.0f)
        br = 1.0f;
#ifdef URF_EXP_ACOS
    const float alpha = (float)urf_div_pi((double)(urf_acosf(br) * 180.0f));   /* :58 */
    return alpha <= angleFilter1;                                               /* :61 */
#else
    return br >= x_angle_thr;   /* :58-61 "alpha <= angleFilter1", alpha = acos(br) in degrees: urf_api.hip urf_angle_threshold */
#endif
}

__device__ __noinline__ bool urf_x_zero_angle(const float* newY, float angleFilter1, float x_angle_thr, float xj, float yj, float x3, float y3,
                                              int j, int p, int cp, float zj, float pz, float z3)
{
    return urf_x_zero_angle_body(newY[j], newY[p], newY[j + cp], angleFilter1, x_angle_thr, xj, yj, x3, y3, zj, pz, z3);
}
/* (k_front_finish requests the three table values together with the points: its own instance) */
__device__ __noinline__ bool urf_x_zero_angle_vals(float nyj, float ny2, float ny3, float angleFilter1, float x_angle_thr, float xj, float yj, float x3,
                                                   float y3, float zj, float pz, float z3)
{
    return urf_x_zero_angle_body(nyj, ny2, ny3, angleFilter1, x_angle_thr, xj, yj, x3, y3, zj, pz, z3);
}

/* z_zero_method.cpp:21-66 for the centre p, given the height tests passed.  xy(r, x, y) delivers the
 * planar coordinates of ring position r (an LDS window or a gather from the ring-sorted arrays). */
template <class FXY>
__device__ __forceinline__ bool urf_z_zero_angle(float inv_cp, float angleFilter2, float z_angle_thr, FXY xy, int p, int cp, float px, float py)
{
    float xa, ya, xb, yb;
    xy(p + cp, xb, yb);
    xy(p - cp, xa, ya);
    const double dx = (double)(xb - xa), dy = (double)(yb - ya);
    if (!(dx * dx + dy * dy < URF_DIST5_SQ))                                    /* :23-28 */
        return false;
    float va1 = 0.f, va2 = 0.f, vb1 = 0.f, vb2 = 0.f;
    for (int k = 1; k <= cp; k++) {                                             /* :35-38 */
        float x, y;
        xy(p - k, x, y);
        va1 = va1 + (x - px);
        va2 = va2 + (y - py);
    }
    for (int k = 1; k <= cp; k++) {                                             /* :44-47 */
        float x, y;
        xy(p + k, x, y);
        vb1 = vb1 + (x - px);
        vb2 = vb2 + (y - py);
    }
    va1 = inv_cp * va1;                                                         /* :52-55 */
    va2 = inv_cp * va2;
    vb1 = inv_cp * vb1;
    vb2 = inv_cp * vb2;
    const float num = va1 * vb1 + va2 * vb2;
    const double na = __builtin_sqrt((double)va1 * (double)va1 + (double)va2 * (double)va2);
    const double nb = __builtin_sqrt((double)vb1 * (double)vb1 + (double)vb2 * (double)vb2);
    float br = (float)((double)num / (na * nb));                                /* :57 */
    if (br < -1.0f)
        br = -1.0f;
    else if (br > 1.0f)
        br = 1.0f;
#ifdef URF_EXP_ACOS
    const float alpha = (float)urf_div_pi((double)(urf_acosf(br) * 180.0f));   /* :63 */
    return alpha <= angleFilter2;                                               /* :66 */
#else
    return br >= z_angle_thr;   /* :63-66, as in urf_x_zero_angle */
#endif
}

/* the two instances k_ring uses: operands gathered from the ring-sorted arrays through the ring's
 * map (quad mapping), or read from an LDS window whose element 0 is ring position `origin` */
__device__ __noinline__ bool urf_z_zero_angle_gather(const float* rx, const float* ry, const urf_ring_map map, float inv_cp,
                                                     float angleFilter2, float z_angle_thr, int p, int cp, float px, float py)
{
    auto gxy = [&](int r, float& x, float& y) {
        const unsigned idx = map.at((unsigned)r);
        x = rx[idx];
        y = ry[idx];
    };
    return urf_z_zero_angle(inv_cp, angleFilter2, z_angle_thr, gxy, p, cp, px, py);
}
__device__ __noinline__ bool urf_z_zero_angle_window(const float* xs, const float* ys, int origin, float inv_cp, float angleFilter2,
                                                     float z_angle_thr, int p, int cp, float px, float py)
{
    auto lxy = [&](int r, float& x, float& y) {
        x = xs[r - origin];
        y = ys[r - origin];
    };
    return urf_z_zero_angle(inv_cp, angleFilter2, z_angle_thr, lxy, p, cp, px, py);
}

/* exact azimuth (and planar range when captured) of one point and its entry in the curb tables
 * (lidar_segmentation.cpp:245-269, blind_spots.cpp:19-56); returns the azimuth.
 * maxDistance (:271-274) is the largest float(sqrt(double s)), s = x^2 + y^2: both roundings
 * are monotone, so the callers track the largest s instead. */
template <class SHARED>
__device__ __noinline__ float urf_ring_point(float* rd2, float* caz, SHARED& S, unsigned gpos, float px, float py,
                                             unsigned flag, bool want_quad)
{
    float d2;
    const float az = urf_azimuth(px, py, &d2);
    if (rd2) {   /* stage capture */
        rd2[gpos] = d2;
        caz[gpos] = az;
    }
    if (flag && az == az) {
        /* curb point: listed for the beam march (k_beams) ... */
        const unsigned e = atomicAdd(&S.n_curb, 1u);
        if (e < URF_CURB_LIST)
            S.curb[e] = az;
        /* ... and entered in the per-degree tables, which stand in for the list when it overflows.  The
         * azimuth lies in [0,360]; cell_lo = largest integer <= az, cell_hi = smallest integer >= az. */
        int cl = (int)__builtin_floorf(az), ch = (int)__builtin_ceilf(az);
        cl = cl < 0 ? 0 : (cl > 360 ? 360 : cl);
        ch = ch < 0 ? 0 : (ch > 360 ? 360 : ch);
        const int ab = (int)urf_fbits(az);
        atomicMin(&S.cmin[cl], ab);
        atomicMax(&S.cmax[ch], ab);
        if (want_quad) {   /* blind_spots.cpp:19-56 */
            if (az >= 0.f && az < 90.f)
                atomicMax(&S.q[0], ab);
            else if (az >= 90.f && az < 180.f)
                atomicMin(&S.q[1], ab);
            else if (az >= 180.f && az < 270.f)
                atomicMax(&S.q[2], ab);
            else if (az < 360.f)   /* "alpha < q4" with q4 starting at 360 */
                atomicMin(&S.q[3], ab);
        }
    }
    return az;
}

/* upper / lower end of beam i's window on ring k (blind_spots.cpp:107,136-143 / :216,245-252) */
__device__ __forceinline__ float urf_fwd_hi(const urf_dev_params& dp, int i, unsigned k, double qk)
{
    const float fi = (float)i;
    const float far = fi == dp.fwd_limit ? 360.0f : (float)((double)i + qk);   /* selects, not branches */
    return k == 0 ? fi + dp.p.beamZone : far;
}
__device__ __forceinline__ float urf_bwd_lo(const urf_dev_params& dp, int i, unsigned k, double qk)
{
    const float fi = (float)i;
    const float far = fi == dp.bwd_limit ? 0.0f : (float)((double)i - qk);
    return k == 0 ? fi - dp.p.beamZone : far;
}
/* arcDistance / ((maxDistance[k] * M_PI) / 180), blind_spots.cpp:65,142 */
__device__ __forceinline__ double urf_arc_ratio(const urf_dev_params& dp, float maxd0, float maxdk)
{
    const float arc = (float)((((double)maxd0 * URF_PI_D) / 180.0) * (double)dp.p.beamZone);
    return (double)arc / (((double)maxdk * URF_PI_D) / 180.0);
}

/* 4 waves per SIMD (<= 128 VGPRs): the kernel hides its barrier and memory latencies with resident
 * workgroups, measured 1.19 -> 1.00 ms against the compiler's own choice of 155 VGPRs */
/* QUADS: curbPoints == 5 (the reference's default), four consecutive points per thread on z alone;
 * otherwise the general path with x / y / z windows.  Two instances, so that the common one does not
 * carry the other's registers. */
template <bool QUADS>
__device__ __forceinline__ void urf_ring_body(const urf_kargs& a, const urf_dev_params& dp, const unsigned c, const unsigned s)
{
    constexpr int CH = URF_RING_CHUNK, PAD = URF_RING_PAD;
    __shared__ urf_ring_shared_t<QUADS> S;
    extern __shared__ unsigned sh_ring_tab[];   /* P[tiles + 1], radd[tiles] (urf_ring_map) */
    int* const cmin = S.cmin;
    int* const cmax = S.cmax;
    int* const sh_q = S.q;
    const unsigned tid = threadIdx.x;
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    const bool star = dp.p.star_shaped_method != 0;
    /* Everything the workgroup needs before it can start is requested at once (scan summary, the
     * ring's size and place, the first 128 entries of its run table, the first 384 star-shaped
     * hits): a chain of dependent round trips cost a fifth of a workgroup's life. */
    URF_PHASE_ACC_DECL;
    const urf_scan_info in = a.info[s];
    const int n = (int)a.ring_cnt[(size_t)s * C + c];
    const unsigned ro = a.ring_off[(size_t)s * (C + 1) + c];   /* the ring's first position among the scan's ring points (star hits) */
    const unsigned* gp = a.rpre + ((size_t)s * C + c) * (a.tiles + 1);
    const uint16_t* gs = a.rstart + ((size_t)s * C + c) * a.tiles;
    const unsigned pt0 = tid <= a.tiles ? gp[tid] : 0;
    const unsigned st0 = tid < a.tiles ? (unsigned)gs[tid] : 0;
    /* the largest x*x + y*y of the ring's points per tile (k_split): maxDistance without reading x / y here */
    const unsigned long long tm0 = a.tmaxs[((size_t)s * a.tiles + (tid < a.tiles ? tid : 0u)) * C + c];
    unsigned h0[3];
#pragma unroll
    for (unsigned u = 0; u < 3; u++)
        h0[u] = star && tid + u * URF_RING_THREADS < K ? (unsigned)a.star_hit[(size_t)s * K + tid + u * URF_RING_THREADS] : 0xffffffffu;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    if (in.status != URF_OK || c >= in.n_rings)
        return;
    if (a.front && a.front_ok[s])
        return;   /* (uniform) a scan of the fused front end (urf_front.hpp: k_front_finish) */
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    const unsigned sb = urf_sbase(a, s);
    const int cp = dp.p.curbPoints;
    const bool want_quad = (c == 1) && dp.p.blind_spots;
    unsigned* const mapP = sh_ring_tab;
    unsigned* const mapA = sh_ring_tab + a.tiles + 1;
    if (tid <= ntiles) {
        mapP[tid] = pt0;
        if (tid < ntiles)
            mapA[tid] = sb + tid * URF_TILE + st0 - pt0;
    }
    for (unsigned t = tid + URF_RING_THREADS; t <= ntiles; t += URF_RING_THREADS) {
        const unsigned pt = gp[t];
        mapP[t] = pt;
        if (t < ntiles)
            mapA[t] = sb + t * URF_TILE + gs[t] - pt;
    }
    for (unsigned i = tid; i < URF_DEG_CELLS; i += URF_RING_THREADS) {
        cmin[i] = URF_INT_NONE_MIN;
        cmax[i] = -1;
    }
    if (tid < 2 * (CH / 32))
        (&S.hb[0][0])[tid] = 0;
    if (tid == 0) {
        a.vis[(size_t)s * C + c] = urf_vis{ __builtin_inff(), -__builtin_inff() };   /* the beam scans see the whole ring (k_nan_rings) */
        sh_q[0] = (int)urf_fbits(0.f);
        sh_q[1] = (int)urf_fbits(180.f);
        sh_q[2] = (int)urf_fbits(180.f);
        sh_q[3] = (int)urf_fbits(360.f);
        S.maxs = 0;
        S.n_cand = 0;
        S.n_hits = 0;
        S.n_curb = 0;
    }
    __syncthreads();
    URF_PHASE_ACC(0);
    /* The ring's map lists its NON-EMPTY runs only: the guess "run = position / average run length" is then
     * exact for an organised sweep whatever azimuth ranges the region of interest removes (with the empty
     * tiles of the reference's default region in the table almost every lookup fell through to the bisection:
     * k_ring took longer on 40 % of the points than on all of them).  One wave compacts the table in place,
     * 64 entries at a time in ascending order (an entry never moves up), before the barrier below. */
    if (tid < 64) {
        unsigned m = 0;
        for (unsigned t0 = 0; t0 < ntiles; t0 += 64) {
            const unsigned t = t0 + tid;
            const unsigned p0 = t < ntiles ? mapP[t] : 0u, p1 = t < ntiles ? mapP[t + 1] : 0u;
            const unsigned ad = t < ntiles ? mapA[t] : 0u;
            const unsigned long long bm = __ballot(p1 > p0);
            if (p1 > p0) {
                const unsigned e = m + urf_popc_below(bm);
                mapP[e] = p0;
                mapA[e] = ad;
            }
            m += (unsigned)__popcll(bm);
        }
        if (tid == 0) {
            mapP[m] = (unsigned)n;
            S.n_runs = m;
        }
    }
    /* lidar_segmentation.cpp:241-242: the star-shaped hits that lie on this ring, as ring positions.  A short
     * list (a ring rarely holds more than a handful of the scan's <= 1022 hits; a list for all of them cost
     * 4 KB of LDS, i.e. resident workgroups); if it overflows, every chunk scans the scan's hits again. */
    if (star) {
#pragma unroll
        for (unsigned u = 0; u < 3; u++)
            if (h0[u] >= ro && h0[u] < ro + (unsigned)n) {
                const unsigned e = atomicAdd(&S.n_hits, 1u);
                if (e < URF_RING_HITS)
                    S.hits[e] = h0[u] - ro;
            }
        for (unsigned k = tid + 3 * URF_RING_THREADS; k < K; k += URF_RING_THREADS) {
            const unsigned h = (unsigned)a.star_hit[(size_t)s * K + k];
            if (h >= ro && h < ro + (unsigned)n) {
                const unsigned e = atomicAdd(&S.n_hits, 1u);
                if (e < URF_RING_HITS)
                    S.hits[e] = h - ro;
            }
        }
    }
    __syncthreads();
    URF_PHASE_ACC(1);
    const unsigned nh = S.n_hits;
    const unsigned nruns = S.n_runs;
    const urf_ring_map map = { mapP, mapA, nruns, (float)nruns / (float)(n > 0 ? n : 1) };
    constexpr bool quads = QUADS;
    double maxs = 0.0;
    {   /* (rows of tiles behind the scan's last one hold whatever an earlier call left there) */
        unsigned long long tm = tid < ntiles ? tm0 : 0ull;
        for (unsigned t = tid + URF_RING_THREADS; t < ntiles; t += URF_RING_THREADS) {
            const unsigned long long v = a.tmaxs[((size_t)s * a.tiles + t) * C + c];
            tm = v > tm ? v : tm;
        }
        maxs = __longlong_as_double((long long)tm);
    }
    const int cs0 = 0;
    const int zpad = PAD + (cp & 3);   /* z slot of chunk point 0: puts p - cp of a quad on a 16-byte boundary for cp = 5 */
    unsigned buf = 0, hbi = 0;   /* z window / hit bitmap of the chunk at hand */

    /* The next chunk's points are requested from memory before the current chunk is evaluated and
     * parked in LDS after it: the evaluation hides the latency.  A thread fetches quads of
     * consecutive ring positions [cs - PAD + 4q, +4): inside one run of the tile-local layout and
     * 16-byte aligned (an organised sweep: always) that is one 16-byte load per array, otherwise
     * four mapped ones. */
    constexpr int NSQ = (CH + 2 * PAD + 4 * URF_RING_THREADS - 1) / (4 * URF_RING_THREADS);
    /* (the four-points-per-thread instance works on z alone and needs five points in front of a chunk and ten behind it:
     * ONE quad per thread -- its own four points -- plus one halo value in each of 16 lanes, five registers per chunk in
     * flight instead of eight: 0.42 -> 0.40 ms, r5) */
    /* (a sweep with drop-outs -- every real one -- has runs of uneven length, and a quad of ring positions then starts at any
     * slot of its tile's run: global memory takes a 16-byte load at any 4-byte boundary, so only a quad that straddles two runs
     * falls back to four mapped loads.  With the loads restricted to 16-byte boundaries, r2-r4, three quads in four fell back on
     * such a sweep: k_ring 0.59 ms per 1024 sensor-like sweeps against 0.39 on the drop-out-free benchmark clouds.) */
    struct __attribute__((packed, aligned(4))) urf_f4u {
        float x, y, z, w;
    };
    struct zbuf {
        float4 q;
        float h;
    };
    constexpr int QH = 8;   /* halo values on either side */
    auto fetchq = [&](int cs, zbuf& b) {
        const int j = cs + 4 * (int)tid;
        b.q = make_float4(0.f, 0.f, 0.f, 0.f);
        b.h = 0.f;
        if (j < n) {
            const unsigned t = map.tile((unsigned)j);
            const unsigned end = mapP[t + 1], idx = mapA[t] + (unsigned)j;
            if ((unsigned)j + 3 < end) {   /* four consecutive slots of one run: one 16-byte load, aligned or not */
                const urf_f4u v = *(const urf_f4u*)(a.rz + idx);
                b.q = make_float4(v.x, v.y, v.z, v.w);
            } else {
                /* the quad straddles the end of its run (a sweep with drop-outs: one quad in eight; a wave takes both branches, so
                 * this one must be short): the rest lies at the start of the next run -- or, runs of fewer than three points,
                 * wherever the map says */
                const unsigned t1 = t + 1 < nruns ? t + 1 : t;
                const unsigned end1 = mapP[t1 + 1], base1 = mapA[t1];
                float ez[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    const unsigned pos = (unsigned)(j + e4);
                    if (pos < (unsigned)n) {
                        unsigned ie = pos < end ? idx + (unsigned)e4 : base1 + pos;
                        if (pos >= end && pos >= end1)
                            ie = map.at(pos);
                        ez[e4] = a.rz[ie];
                    }
                }
                b.q = make_float4(ez[0], ez[1], ez[2], ez[3]);
            }
        }
        if (tid < 2 * QH) {
            const int hp = (int)tid < QH ? cs - QH + (int)tid : cs + CH + ((int)tid - QH);
            if (hp >= 0 && hp < n)
                b.h = a.rz[map.at((unsigned)hp)];
        }
    };
    float4 fx[NSQ], fy[NSQ], fzA[NSQ];
    [[maybe_unused]] zbuf zA;
    auto fetch = [&](int cs, float4 (&fz)[NSQ]) {
#pragma unroll
        for (int m = 0; m < NSQ; m++) {
            const int j = cs - PAD + 4 * ((int)tid + m * URF_RING_THREADS);
            fx[m] = fy[m] = fz[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j + 3 < 0 || j >= n || j + 3 < cs - cp || j >= cs + CH + cp)
                continue;   /* outside the ring or outside the chunk's halo */
            bool wide = false;
            if (j >= 0 && j + 3 < n) {
                const unsigned t = map.tile((unsigned)j);
                const unsigned idx = mapA[t] + (unsigned)j;
                if ((unsigned)j + 3 < mapP[t + 1]) {   /* (one run: a 16-byte load at any 4-byte boundary, see fetchq) */
                    wide = true;
                    if (!quads) {   /* (uniform) the four-points-per-thread path works on z alone */
                        const urf_f4u vx = *(const urf_f4u*)(a.rx + idx), vy = *(const urf_f4u*)(a.ry + idx);
                        fx[m] = make_float4(vx.x, vx.y, vx.z, vx.w);
                        fy[m] = make_float4(vy.x, vy.y, vy.z, vy.w);
                    }
                    const urf_f4u vz = *(const urf_f4u*)(a.rz + idx);
                    fz[m] = make_float4(vz.x, vz.y, vz.z, vz.w);
                }
            }
            if (!wide) {
                float ex[4] = { 0.f, 0.f, 0.f, 0.f }, ey[4] = { 0.f, 0.f, 0.f, 0.f }, ez[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++)
                    if (j + e4 >= 0 && j + e4 < n) {
                        const unsigned ie = map.at((unsigned)(j + e4));
                        if (!quads) {
                            ex[e4] = a.rx[ie];
                            ey[e4] = a.ry[ie];
                        }
                        ez[e4] = a.rz[ie];
                    }
                fx[m] = make_float4(ex[0], ex[1], ex[2], ex[3]);
                fy[m] = make_float4(ey[0], ey[1], ey[2], ey[3]);
                fz[m] = make_float4(ez[0], ez[1], ez[2], ez[3]);
            }
        }
    };
    /* one chunk: `fz` / `zq` hold its z values (requested while the chunk before was evaluated); parked, then the buffer is
     * refilled with chunk `cs_next` */
    auto chunk = [&](const int cs, float4 (&fz)[NSQ], zbuf& zq, const int cs_next) {
        /* park [cs - PAD, cs + CH + PAD) (positions outside the ring hold zeros nobody reads), mark
         * the star hits of the chunk, clear the other bitmap */
        {
            if constexpr (QUADS) {
                float* const zw = S.zsb[buf] + zpad;   /* slot of position cs */
                zw[4 * tid] = zq.q.x;                  /* (zpad = PAD + 1: the evaluation's 16-byte reads, one slot lower, are the aligned ones) */
                zw[4 * tid + 1] = zq.q.y;
                zw[4 * tid + 2] = zq.q.z;
                zw[4 * tid + 3] = zq.q.w;
                if (tid < 2 * QH)
                    zw[(int)tid < QH ? (int)tid - QH : CH + ((int)tid - QH)] = zq.h;
            }
#pragma unroll
            for (int m = 0; m < (QUADS ? 0 : NSQ); m++) {
                const int li = 4 * ((int)tid + m * URF_RING_THREADS);   /* slot of position cs - PAD + li */
                if (li < CH + 2 * PAD) {
                    if (!quads) {
                        *(float4*)(S.xs + li) = fx[m];
                        *(float4*)(S.ys + li) = fy[m];
                    }
                    float* const zw = S.zsb[QUADS ? buf : 0u];
                    zw[li + zpad - PAD] = fz[m].x;
                    zw[li + zpad - PAD + 1] = fz[m].y;
                    zw[li + zpad - PAD + 2] = fz[m].z;
                    zw[li + zpad - PAD + 3] = fz[m].w;
                }
            }
            if (nh <= URF_RING_HITS) {
                for (unsigned i = tid; i < nh; i += URF_RING_THREADS) {
                    const unsigned h = S.hits[i] - (unsigned)cs;
                    if (h < (unsigned)CH)
                        atomicOr(&S.hb[hbi][h >> 5], 1u << (h & 31));
                }
            } else {   /* (a ring that collects more hits than the list holds: pathological input) */
                for (unsigned k = tid; k < K; k += URF_RING_THREADS) {
                    const unsigned h = (unsigned)a.star_hit[(size_t)s * K + k] - ro - (unsigned)cs;
                    if (h < (unsigned)CH && h + (unsigned)cs < (unsigned)n)
                        atomicOr(&S.hb[hbi][h >> 5], 1u << (h & 31));
                }
            }
            if (tid < CH / 32)
                S.hb[hbi == 2u ? 0u : hbi + 1u][tid] = 0;
        }
        __syncthreads();
        URF_PHASE_ACC(2);
        if (cs_next < n) {
            if constexpr (QUADS)
                fetchq(cs_next, zq);
            else
                fetch(cs_next, fz);
        }
        if (quads) {
            /* ---- four consecutive points per thread, curbPoints == 5 ----
             * Cheap tests and the float azimuth for every point, stored at once; the points that
             * need an angle test of a detector or the exact azimuth go to the ring's candidate
             * list, which is worked off densely (all lanes busy instead of the two or three that
             * sit on a curb) when it fills up and at the end of the ring. */
            const int q0 = cs + 4 * (int)tid;
#ifdef URF_EXP_SKIP_EVAL
            if (false) {
#else
            if (q0 < n) {
#endif
                const float4* zp = (const float4*)(S.zsb[QUADS ? buf : 0u] + 4 * tid + PAD - 4);   /* slot of q0 - 5 */
                float w[16];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const float4 t = zp[v];
                    w[4 * v] = t.x; w[4 * v + 1] = t.y; w[4 * v + 2] = t.z; w[4 * v + 3] = t.w;
                }
                /* M[j] = max |z| over window slots j..j+5 (z_zero_method.cpp:39-40, :48-49: centre included) */
                float T[12], M[9];
#pragma unroll
                for (int k = 0; k < 12; k++)
                    T[k] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(w[k]), __builtin_fabsf(w[k + 1])), __builtin_fabsf(w[k + 2]));
#pragma unroll
                for (int j = 0; j < 9; j++)
                    M[j] = __builtin_fmaxf(T[j], T[j + 3]);
                const unsigned hbits = (S.hb[hbi][tid >> 3] >> ((tid & 7u) * 4u)) & 15u;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int p = q0 + i;
                    if (p < 0 || p >= n)
                        continue;
                    const float pz = w[5 + i];
                    unsigned t = ((hbits >> i) & 1u) ? URF_CAND_STAR : 0u;
                    if (dp.p.x_zero_method && p - 2 >= 5 && p - 2 <= (n - 1) - 5) {   /* j = p - cp/2 in [cp, n-1-cp] */
                        const float zj = w[3 + i], z3 = w[8 + i];
                        if ((__builtin_fabsf(zj - pz) >= dp.p.curbHeight || __builtin_fabsf(z3 - pz) >= dp.p.curbHeight) &&
                            (double)__builtin_fabsf(zj - z3) >= 0.05)                            /* x_zero_method.cpp:62-64 */
                            t |= URF_CAND_XZERO;
                    }
                    if (dp.p.z_zero_method && p >= 5 && p <= (n - 1) - 5) {
                        const float az = __builtin_fabsf(pz), max1 = M[i], max2 = M[5 + i];
                        if ((max1 - az >= dp.p.curbHeight || max2 - az >= dp.p.curbHeight) &&
                            (double)__builtin_fabsf(max1 - max2) >= 0.05)                        /* z_zero_method.cpp:67-69 */
                            t |= URF_CAND_ZZERO;
                    }
                    /* (a point too close to the x axis for k_split's approximate azimuth carries URF_REC_AZ_UNKNOWN
                     * and gets its exact azimuth in k_label; the ring's largest range comes from k_split's
                     * per-tile maxima: this path reads neither x nor y) */
                    if (a.rd2)   /* stage capture: exact azimuth and planar range of every point */
                        t |= URF_CAND_EXACT;
                    if (t)
                        S.cand[atomicAdd(&S.n_cand, 1u)] = (unsigned)p | (t << URF_CAND_SHIFT);
                }
                /* (nothing is stored per point here: k_split left every slot's record as "no detector hit,
                 * approximate azimuth"; the candidate pass ORs the hits of the few curb points into theirs) */
            }
            __syncthreads();
            URF_PHASE_ACC(3);
#ifdef URF_EXP_SKIP_CAND
            if (false) {
#else
            if (S.n_cand > urf_ring_shared_t<QUADS>::CAND - CH || cs + CH >= n) {   /* the next chunk might not fit / last chunk */
#endif
                const unsigned nc = S.n_cand;
                for (unsigned e = tid; e < nc; e += URF_RING_THREADS) {
                    const unsigned v = S.cand[e], t = v >> URF_CAND_SHIFT;
                    const int p = (int)(v & ((1u << URF_CAND_SHIFT) - 1u));
                    const unsigned ip = map.at((unsigned)p);
                    const float px = a.rx[ip], py = a.ry[ip];
                    unsigned flag = (t & URF_CAND_STAR) ? 1u : 0u;
                    /* (r5, measured: the two angle tests as separate work items -- more lanes, one f64 chain per lane -- cost a barrier,
                     * an atomic per hit and the gathers twice: 0.40 -> 0.44 ms, profiles/r5_ring_ab.txt) */
                    if (t & URF_CAND_XZERO) {   /* j = p - 2 and j + cp = p + 3 exist (height tests passed) */
                        const unsigned ij = map.at((unsigned)(p - 2)), i3 = map.at((unsigned)(p + 3));
                        if (urf_x_zero_angle(a.newY, dp.p.angleFilter1, dp.x_angle_thr, a.rx[ij], a.ry[ij], a.rx[i3], a.ry[i3], p - 2, p, 5, a.rz[ij],
                                             a.rz[ip], a.rz[i3]))
                            flag |= 2u;
                    }
                    if ((t & URF_CAND_ZZERO) &&   /* operands come from the ring-sorted arrays (L2) */
                        urf_z_zero_angle_gather(a.rx, a.ry, map, dp.inv_cp, dp.p.angleFilter2, dp.z_angle_thr, p, 5, px, py))
                        flag |= 4u;
                    if (flag || (t & URF_CAND_EXACT)) {
                        const float az = urf_ring_point(a.rd2, a.caz, S, ip, px, py, flag, want_quad);
                        if (flag) {
                            atomicOr(&a.rec[ip], flag << URF_REC_FLAG_SHIFT);
                            if (!(az == az))   /* x == y == 0: a NaN azimuth (include/urf.h: n_nan_azimuth), counted per scan (the others: k_label) */
                                atomicAdd(&a.info[s].n_nan_azimuth, 1u);
                        }
                    }
                }
                __syncthreads();
                URF_PHASE_ACC(4);
                if (tid == 0)
                    S.n_cand = 0;
            }
        } else {
#pragma unroll
            for (int e = 0; e < URF_RING_PPT; e++) {
                const int lc = e * URF_RING_THREADS + (int)tid;   /* chunk-relative index */
                const int p = cs + lc;
                if (p >= n)
                    continue;
                const int lp = lc + PAD, lz = lc + zpad;
                const float px = S.xs[lp], py = S.ys[lp], pz = S.zsb[0][lz];
                unsigned flag = (S.hb[hbi][lc >> 5] >> (lc & 31)) & 1u;

                /* Both detectors are an && of an angle test (f64 sqrt/div, acos) and cheap float
                 * height tests.  The height tests run first: on road surface they fail for
                 * whole waves, which then skip the expensive part.  (Reordering an && chain of
                 * side-effect-free tests does not change its value.) */
                if (dp.p.x_zero_method) {   /* x_zero_method.cpp:30-68, evaluated for the point it marks */
                    const int j = p - cp / 2;
                    if (j >= cp && j <= (n - 1) - cp) {
                        const float zj = S.zsb[0][lz - cp / 2], z3 = S.zsb[0][lz - cp / 2 + cp];
                        const bool heights = (__builtin_fabsf(zj - pz) >= dp.p.curbHeight ||
                                              __builtin_fabsf(z3 - pz) >= dp.p.curbHeight) &&
                                             (double)__builtin_fabsf(zj - z3) >= 0.05;          /* :62-64 */
                        if (heights && urf_x_zero_angle(a.newY, dp.p.angleFilter1, dp.x_angle_thr, S.xs[j - cs + PAD], S.ys[j - cs + PAD],
                                                        S.xs[j + cp - cs + PAD], S.ys[j + cp - cs + PAD], j, p, cp, zj, pz, z3))
                            flag |= 2u;
                    }
                }
                if (dp.p.z_zero_method) {   /* z_zero_method.cpp:21-72 */
                    if (p >= cp && p <= (n - 1) - cp) {
                        const float az = __builtin_fabsf(pz);
                        float max1 = az, max2 = az;
                        for (int k = 1; k <= cp; k++) {                                         /* :39-40, :48-49 */
                            const float za = __builtin_fabsf(S.zsb[0][lz - k]), zb = __builtin_fabsf(S.zsb[0][lz + k]);
                            if (za > max1)
                                max1 = za;
                            if (zb > max2)
                                max2 = zb;
                        }
                        const bool heights = (max1 - az >= dp.p.curbHeight || max2 - az >= dp.p.curbHeight) &&
                                             (double)__builtin_fabsf(max1 - max2) >= 0.05;      /* :67-69 */
                        if (heights && urf_z_zero_angle_window(S.xs, S.ys, cs - PAD, dp.inv_cp, dp.p.angleFilter2, dp.z_angle_thr, p, cp, px, py))
                            flag |= 4u;
                    }
                }
                const double s2 = (double)px * (double)px + (double)py * (double)py;
                maxs = s2 > maxs ? s2 : maxs;
                if (flag || a.rd2) {   /* the exact azimuth: curb points (beam tables) and the stage capture */
                    const unsigned ip = map.at((unsigned)p);
                    const float az = urf_ring_point(a.rd2, a.caz, S, ip, px, py, flag, want_quad);
                    if (flag) {
                        atomicOr(&a.rec[ip], flag << URF_REC_FLAG_SHIFT);
                        if (!(az == az))
                            atomicAdd(&a.info[s].n_nan_azimuth, 1u);
                    }
                }
            }
        }
        if (!QUADS)   /* (two z windows: the next chunk is parked into the other one; n_cand / the hit bitmaps are ordered by the barrier after the parking) */
            __syncthreads();
        URF_PHASE_ACC(5);
        buf ^= 1u;
        hbi = hbi == 2u ? 0u : hbi + 1u;
    };
    if constexpr (QUADS) {
        /* (two chunks of lead, two register buffers in rotation: 0.402 ms against 0.399 -- the parking does not wait for data,
         * profiles/r5_ring_ab.txt) */
        fetchq(cs0, zA);
        for (int cs = cs0; cs < n; cs += CH)
            chunk(cs, fzA, zA, cs + CH);
    } else {
        fetch(cs0, fzA);
        for (int cs = cs0; cs < n; cs += CH)
            chunk(cs, fzA, zA, cs + CH);
    }

#ifdef URF_EXP_SKIP_EPILOGUE
    return;
#endif
    {   /* the ring's largest squared range: wave maximum first, one LDS atomic per wave
         * (non-negative doubles order like integers) */
        unsigned long long m = (unsigned long long)__double_as_longlong(maxs);
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)m, o), hi = (unsigned)__shfl_xor((int)(unsigned)(m >> 32), o);
            const unsigned long long w = ((unsigned long long)hi << 32) | lo;
            m = w > m ? w : m;
        }
        if ((tid & 63) == 0)
            atomicMax(&S.maxs, m);
    }
    __syncthreads();   /* S.maxs is complete (and so is the ring's list of curb points: every candidate pass ended with a barrier) */
    if (tid == 0)
        a.maxdist[(size_t)s * C + c] = (float)__builtin_sqrt(__longlong_as_double((long long)S.maxs));
    if (want_quad && tid < 4)
        a.quad[(size_t)s * 4 + tid] = __uint_as_float((unsigned)sh_q[tid]);
    /* What the beam march asks of a ring is "is there a curb point with azimuth in [lo, hi]" (blind_spots.cpp:
     * 112-116, 151-155).  A ring holds a few dozen curb points at most on real ground, so it hands k_beams their
     * exact azimuths (<= URF_CURB_LIST floats) instead of two per-degree tables of 361 floats each: 185 KB per
     * 64-ring scan written here and read there, which made k_beams a bandwidth-bound kernel.  Only a ring with more
     * curb points than the list holds (rough ground) builds the tables. */
    const unsigned ncurb = S.n_curb;
    if (tid == 0)
        a.curb_cnt[(size_t)s * C + c] = ncurb <= URF_CURB_LIST ? ncurb : URF_CURB_DENSE;
    if (ncurb <= URF_CURB_LIST) {   /* (uniform) */
        if (tid < ncurb)
            a.curb_az[((size_t)s * C + c) * URF_CURB_LIST + tid] = S.curb[tid];
        URF_PHASE_ACC(6);
        URF_PHASE_ACC_DUMP("k_ring", 7);
        return;
    }
    /* sufmin[i] = min curb azimuth >= i ; premax[i] = max curb azimuth <= i ; NaN = none.  Both are
     * prefix maxima: premax over the cells in order, sufmin over the cells in REVERSE order of the
     * bit-flipped values (a minimum is the maximum of the complements).  Three consecutive cells per
     * thread, one DPP scan across the wave, the first wave's total handed to the second. */
    static_assert(URF_RING_THREADS == 128 && 3 * URF_RING_THREADS >= URF_DEG_CELLS, "three cells per thread, two waves");
    static_assert(URF_CURB_LIST <= URF_RING_THREADS, "one listed azimuth per thread");
    __shared__ unsigned wtot[2];
    unsigned up[3], dn[3];
#pragma unroll
    for (unsigned e = 0; e < 3; e++) {
        const unsigned i = 3 * tid + e;                 /* cell of the prefix maximum */
        up[e] = i < URF_DEG_CELLS ? (unsigned)(cmax[i] + 1) : 0u;                          /* none (-1) -> 0 */
        dn[e] = i < URF_DEG_CELLS ? ~(unsigned)cmin[URF_DEG_CELLS - 1 - i] : 0u;           /* none (0x7fffffff) -> 0x80000000, below every value */
        if (e) {
            up[e] = up[e] > up[e - 1] ? up[e] : up[e - 1];
            dn[e] = dn[e] > dn[e - 1] ? dn[e] : dn[e - 1];
        }
    }
    const unsigned iu = urf_wave_scan_max(up[2]), id = urf_wave_scan_max(dn[2]);
    if (tid == 63) {
        wtot[0] = iu;
        wtot[1] = id;
    }
    unsigned pu = (unsigned)__shfl_up((int)iu, 1), pd = (unsigned)__shfl_up((int)id, 1);
    if ((tid & 63) == 0)
        pu = pd = 0;
    __syncthreads();   /* wtot is complete */
    if (tid >= 64) {
        pu = pu > wtot[0] ? pu : wtot[0];
        pd = pd > wtot[1] ? pd : wtot[1];
    }
    float* sm = a.sufmin + ((size_t)s * C + c) * URF_DEG_CELLS;
    float* pm = a.premax + ((size_t)s * C + c) * URF_DEG_CELLS;
#pragma unroll
    for (unsigned e = 0; e < 3; e++) {
        const unsigned i = 3 * tid + e;
        if (i < URF_DEG_CELLS) {
            const unsigned u = up[e] > pu ? up[e] : pu, d = dn[e] > pd ? dn[e] : pd;
            pm[i] = u == 0 ? __builtin_nanf("") : __uint_as_float(u - 1u);
            sm[URF_DEG_CELLS - 1 - i] = (d == 0x80000000u || d == 0u) ? __builtin_nanf("") : __uint_as_float(~d);
        }
    }
    URF_PHASE_ACC(6);
    URF_PHASE_ACC_DUMP("k_ring", 7);
}

#ifndef URF_RING_WAVES
#define URF_RING_WAVES 6   /* 77 registers, 13.6 KB of LDS: twelve workgroups per CU (A/B: 4 -> 0.640 ms, 5 -> 0.547, 6 -> 0.51) */
#endif
__global__ __launch_bounds__(URF_RING_THREADS) __attribute__((amdgpu_waves_per_eu(URF_RING_WAVES, URF_RING_WAVES))) void k_ring(urf_kargs a, urf_dev_params dp)
{
    urf_ring_body<true>(a, dp, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(URF_RING_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ring_general(urf_kargs a, urf_dev_params dp)
{
    urf_ring_body<false>(a, dp, blockIdx.x, blockIdx.y);
}
/* The scans the fused front end handed back (urf_front.hpp: front_list, normally none -- the kernels return at once): persistent
 * workgroups over list x rings, so that a batch whose scans all took the fused front end does not pay for 65 536 workgroups
 * that look at a flag and leave. */
__global__ __launch_bounds__(URF_RING_THREADS) void k_ring_list(urf_kargs a, urf_dev_params dp)
{
    const unsigned n = a.star_count[6], C = (unsigned)dp.p.channels;
    for (unsigned w = blockIdx.x; w < n * C; w += gridDim.x) {
        const unsigned s = a.front_list[w / C], c = w % C;
        if (dp.p.curbPoints == 5)
            urf_ring_body<true>(a, dp, c, s);
        else
            urf_ring_body<false>(a, dp, c, s);
        __syncthreads();   /* the LDS is reused by the next ring */
    }
}

/* ------------------------------------------------------------------------- */
/* k_nan_rings                                                                 */
/* ------------------------------------------------------------------------- */
/* A ring point with x == y == 0 has the azimuth asin(0 / 0) = NaN (lidar_segmentation.cpp:245-269).  The reference
 * sorts every ring with a Lomuto quicksort (:70-93) whose only comparison, alpha < pivot, is false for a NaN on either
 * side: the non-NaN azimuths still come out ascending, but every NaN ends up at a place that depends on the order of
 * the input, and the beam scans of blind_spots.cpp (:107,124,146,164 forwards, :216,233,255,273 backwards) end at the
 * first NaN they meet -- the forward beams see only what lies in front of the first NaN of the sorted ring, the backward
 * beams only what lies behind the last one.  Deterministic, hence part of the contract: for the listed rings (k_split:
 * normally none, and this kernel returns at once) the quicksort is run LITERALLY -- same pivot, same comparison,
 * same swaps, on (exact azimuth, position in the ring) pairs -- and what comes out of it is
 *   vis[ring]    = (largest azimuth in front of the first NaN, smallest azimuth behind the last NaN), with which
 *                  k_beams limits what a curb point of the ring can stop and what a beam can mark on it;
 *   ssrt[ring..] = the ring in the reference's final order (positions), for the published order (k_ring_order).
 * (Two non-NaN points of one ring with bit-identical azimuths on either side of such a boundary would need the
 * positions themselves; the limits are compared as values.)
 *
 * One workgroup per listed ring, persistent over the list.  The pairs live in LDS (rings of up to URF_NAN_LDS
 * points) or in the ring's stretch of wsg, which nobody reads after k_star_walk.  Wave 0 runs the partition loop 64
 * elements at a time: a step in which no element is smaller than the pivot moves nothing, one in which all are and
 * the block of not-smaller elements is empty only swaps elements with themselves -- an ascending run with its
 * largest element as the pivot (what the recursion meets on an organised sweep) costs n / 64 steps per partition;
 * anything else goes element by element, exactly as written in the reference. */
#define URF_NAN_LDS 6144u   /* pairs of 8 bytes: 48 KB */
__device__ __forceinline__ float urf_pair_alpha(unsigned long long e) { return __uint_as_float((unsigned)(e >> 32)); }

/* lidar_segmentation.cpp:69-82 partition(low, high) on A, by one wave (all 64 lanes call it with uniform arguments) */
__device__ int urf_lomuto_partition(volatile unsigned long long* A, int low, int high)   /* (volatile: one lane writes what the others read next, LDS or global memory) */
{
    const int lane = (int)urf_lane();
    const float pivot = urf_pair_alpha(A[high]);
    int i = low - 1;
    for (int j0 = low; j0 <= high - 1; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j <= high - 1;
        const unsigned long long e = in ? A[j] : 0ull;
        const bool less = in && urf_pair_alpha(e) < pivot;
        const unsigned long long m = __ballot(less), vm = __ballot(in);
        if (m == 0ull)
            continue;                                   /* no element of the step is swapped */
        if (m == vm && i == j0 - 1) {
            i += (int)__popcll(vm);                     /* every swap of the step is a swap with itself */
            continue;
        }
        if (lane == 0) {
            const int jend = j0 + 63 < high - 1 ? j0 + 63 : high - 1;
            for (int jj = j0; jj <= jend; jj++) {
                const unsigned long long ej = A[jj];
                if (urf_pair_alpha(ej) < pivot) {
                    i++;
                    const unsigned long long ei = A[i];
                    A[i] = ej;
                    A[jj] = ei;
                }
            }
        }
        __threadfence_block();
        i = __shfl(i, 0);
    }
    if (lane == 0) {
        const unsigned long long t = A[i + 1];
        A[i + 1] = A[high];
        A[high] = t;
    }
    __threadfence_block();
    return i + 1;
}

/* quickSort(0, n - 1), lidar_segmentation.cpp:85-93, by one wave (all 64 lanes call it): the two halves of a partition are
 * disjoint, so the order in which they are sorted does not matter -- the smaller one first, the larger one on a stack
 * (<= log2 n deep; stk: 2 * 64 ints of LDS) */
__device__ __noinline__ void urf_lomuto_sort(volatile unsigned long long* A, unsigned n, int* stk)
{
    const unsigned lane = urf_lane();
    int top = 0, low = 0, high = (int)n - 1;
    for (;;) {
        while (low < high) {
            const int pi = urf_lomuto_partition(A, low, high);
            const int l0 = low, h0 = pi - 1, l1 = pi + 1, h1 = high;
            const bool left_small = (h0 - l0) < (h1 - l1);
            const int pl = left_small ? l1 : l0, ph = left_small ? h1 : h0;   /* pushed */
            low = left_small ? l0 : l1;
            high = left_small ? h0 : h1;
            if (pl < ph && top < 64) {
                if (lane == 0) {
                    stk[2 * top] = pl;
                    stk[2 * top + 1] = ph;
                }
                top++;
            }
        }
        if (top == 0)
            break;
        top--;
        __threadfence_block();
        low = stk[2 * top];
        high = stk[2 * top + 1];
    }
}

__global__ __launch_bounds__(256) void k_nan_rings(urf_kargs a, urf_dev_params dp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long sh_pairs[];
    __shared__ int stk[2 * 64];
    __shared__ unsigned n_nan, first_nan, last_nan, claimed;
    const unsigned n_list = a.star_count[3];
    if (n_list == 0)
        return;
    const unsigned tid = threadIdx.x, C = (unsigned)dp.p.channels;
    for (unsigned w = blockIdx.x; w < n_list; w += gridDim.x) {
        const unsigned ent = a.nan_list[w], s = ent / C, c = ent % C;
        const urf_scan_info in = a.info[s];
        if (in.status != URF_OK || c >= in.n_rings)
            continue;   /* (uniform) */
        unsigned off, len;
        urf_scan_range(a, s, off, len);
        const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
        const unsigned sb = urf_sbase(a, s);
        const unsigned n = a.ring_cnt[(size_t)s * C + c], ro = a.ring_off[(size_t)s * (C + 1) + c];
        volatile unsigned long long* const A = n <= URF_NAN_LDS ? sh_pairs : (unsigned long long*)(a.wsg + sb + ro);
        /* A ring can stand on the list twice (k_split listed it, k_table_repair cleared the mask, k_split_repair listed it
         * again): the first workgroup to get here claims it -- k_ring left vis = (+inf, -inf), the claim turns f_hi into a NaN
         * pattern until the real value is written below -- and the other one leaves: two workgroups sorting one ring's stretch
         * of global memory in place (rings beyond URF_NAN_LDS points) would race. */
        if (tid == 0) {
            unsigned* const claim = (unsigned*)&a.vis[(size_t)s * C + c].f_hi;
            claimed = atomicCAS(claim, 0x7f800000u, 0x7fc00001u) == 0x7f800000u ? 1u : 0u;
            n_nan = 0;
            first_nan = 0xffffffffu;
            last_nan = 0;
        }
        __syncthreads();
        if (!claimed) {   /* (uniform) */
            __syncthreads();
            continue;
        }
        /* the ring in bucket order (input order: what the reference sorts), exact azimuths */
        unsigned mine = 0;
        for (unsigned j = tid; j < n; j += 256) {
            const unsigned slot = sb + urf_ring_slot(a, s, C, c, ntiles, j);
            float d2;
            const float az = urf_azimuth(a.rx[slot], a.ry[slot], &d2);
            A[j] = ((unsigned long long)__float_as_uint(az) << 32) | j;
            mine += !(az == az);
        }
        if (mine)
            atomicAdd(&n_nan, mine);
        __threadfence_block();
        __syncthreads();
        if (n_nan == 0) {   /* (uniform) a bit set against a ring table that was rebuilt afterwards */
            if (tid == 0) {
                atomicAnd(&a.nan_mask[(size_t)s * 4 + (c >> 5)], ~(1u << (c & 31u)));
                a.vis[(size_t)s * C + c] = urf_vis{ __builtin_inff(), -__builtin_inff() };   /* (the claim above) */
            }
            __syncthreads();
            continue;
        }
        if (tid < 64 && n >= 2)
            urf_lomuto_sort(A, n, stk);
        __threadfence_block();
        __syncthreads();
        for (unsigned j = tid; j < n; j += 256) {
            const unsigned long long e = A[j];
            a.ssrt[sb + ro + j] = (unsigned)e;   /* the ring in the reference's final order */
            const float az = urf_pair_alpha(e);
            if (!(az == az)) {
                atomicMin(&first_nan, j);
                atomicMax(&last_nan, j);
            }
        }
        __syncthreads();
        if (tid == 0) {
            urf_vis v;
            v.f_hi = first_nan > 0 ? urf_pair_alpha(A[first_nan - 1]) : -__builtin_inff();
            v.b_lo = last_nan + 1 < n ? urf_pair_alpha(A[last_nan + 1]) : __builtin_inff();
            a.vis[(size_t)s * C + c] = v;
        }
        __syncthreads();   /* the pairs / the counters are reused by the next listed ring */
    }
}

/* ------------------------------------------------------------------------- */
/* k_beams                                                                     */
/* ------------------------------------------------------------------------- */
/* blind_spots.cpp:72-99 / :181-208 */
__device__ __forceinline__ bool urf_blind(const urf_params& p, const float* q, int i)
{
    if (!p.blind_spots)
        return false;
    const float fi = (float)i;
    if (p.xDirection == 0)
        return (q[0] != 0.f && q[3] != 360.f && (fi <= q[0] || fi >= q[3])) ||
               (q[1] != 180.f && q[2] != 180.f && fi >= q[1] && fi <= q[2]);
    if (p.xDirection == 1)
        return (q[1] != 180.f && fi >= q[1] && i <= 270) || (q[0] != 0.f && (fi <= q[0] || i >= 270));
    return (q[3] != 360.f && (fi >= q[3] || i <= 90)) || (q[2] != 180.f && fi <= q[2] && i >= 90);
}

/* One thread per integer degree casts the forward and the backward beam that
 * start there and finds the first ring whose window holds a curb point.
 * URF_BEAM_PARTS groups of 384 threads share the rings of a scan (group h takes the rings k = h mod URF_BEAM_PARTS in
 * every ring loop): one workgroup per scan is all a sweep of the callback path has, and its loops over the rings are
 * chains of LDS round trips and dependent instructions. */
#ifndef URF_BEAM_PARTS
#define URF_BEAM_PARTS 2
#endif
#define URF_BEAM_THREADS (384 * URF_BEAM_PARTS)
__global__ __launch_bounds__(URF_BEAM_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_beams(urf_kargs a, urf_dev_params dp)
{
    constexpr unsigned NH = URF_BEAM_PARTS;
    __shared__ int16_t xs[2][NH][384];   /* the groups' first stopping rings, per degree */
    __shared__ double qk[URF_MAX_CHANNELS];
    __shared__ float q[4];
    __shared__ unsigned long long mf[URF_MAX_CHANNELS * 6], mb[URF_MAX_CHANNELS * 6];
    __shared__ int16_t pf[URF_MAX_CHANNELS * 6], nb[URF_MAX_CHANNELS * 6];
    __shared__ unsigned lcnt[URF_MAX_CHANNELS];   /* curb points of ring k (URF_CURB_DENSE: see its per-degree tables) */
    __shared__ unsigned lpre[URF_MAX_CHANNELS + 1];   /* listed curb points on the rings in front of ring k */
    __shared__ unsigned n_dense;                       /* rings whose list overflowed */
    /* rings that hold a point with a NaN azimuth (k_nan_rings; normally none): what the forward / backward scans see of them */
    __shared__ float vfh[URF_MAX_CHANNELS], vbl[URF_MAX_CHANNELS];
    extern __shared__ unsigned sh_beams[];            /* sfm[channels][12] | sbm[channels][12] | lst[channels][URF_CURB_LIST] */
    const unsigned s = blockIdx.x, tid = threadIdx.x;
    const unsigned part = tid / 384u, dt = tid % 384u;   /* (a wave lies in one group: 384 = 6 x 64) */
    const unsigned C = (unsigned)dp.p.channels;
    unsigned* const sfm = sh_beams;                   /* per ring: the degrees whose forward / backward beam it stops (bit d) */
    unsigned* const sbm = sfm + C * 12;
    float* const lst = (float*)(sbm + C * 12);        /* the rings' lists of curb azimuths (k_ring) */
    /* the scan's summary, the rings' curb counts and their lists are requested together */
    URF_PHASE_DECL;
    const urf_scan_info in = a.info[s];
    const unsigned v_cnt = tid < C ? a.curb_cnt[(size_t)s * C + tid] : 0u;
    const uint4 nanm = *(const uint4*)(a.nan_mask + (size_t)s * 4);
    const bool has_nan = (nanm.x | nanm.y | nanm.z | nanm.w) != 0u;   /* (uniform) */
    constexpr unsigned LPT = (URF_MAX_CHANNELS * URF_CURB_LIST + URF_BEAM_THREADS - 1) / URF_BEAM_THREADS;
    float v_lst[LPT];
#pragma unroll
    for (unsigned e = 0; e < LPT; e++) {
        const unsigned idx = tid + e * URF_BEAM_THREADS;
        v_lst[e] = idx < C * URF_CURB_LIST ? a.curb_az[(size_t)s * C * URF_CURB_LIST + idx] : 0.f;   /* (entries behind a ring's count: never looked at) */
    }
    if (in.status != URF_OK)
        return;
    const unsigned nR = in.n_rings;
    const float* maxd = a.maxdist + (size_t)s * C;
    if (tid < 4) {
        const float init[4] = { 0.f, 180.f, 180.f, 360.f };
        /* q1..q4 come from sorted ring 1 (blind_spots.cpp:19) */
        q[tid] = (dp.p.blind_spots && nR > 1) ? a.quad[(size_t)s * 4 + tid] : init[tid];
    }
    for (unsigned k = tid; k < nR; k += URF_BEAM_THREADS)
        qk[k] = urf_arc_ratio(dp, maxd[0], maxd[k]);
    if (tid < C)
        lcnt[tid] = v_cnt;
    if (has_nan && tid < nR && tid < C) {
        const urf_vis v = a.vis[(size_t)s * C + tid];
        vfh[tid] = v.f_hi;
        vbl[tid] = v.b_lo;
    }
#pragma unroll
    for (unsigned e = 0; e < LPT; e++) {
        const unsigned idx = tid + e * URF_BEAM_THREADS;
        if (idx < C * URF_CURB_LIST)
            lst[idx] = v_lst[e];
    }
    for (unsigned e = tid; e < nR * 12; e += URF_BEAM_THREADS) {
        sfm[e] = 0u;
        sbm[e] = 0u;
    }
    /* where each ring's listed points start in the scan's flat numbering: one wave, two rings per lane */
    if (tid < 64) {
        static_assert(URF_MAX_CHANNELS <= 128, "two rings per lane");
        const unsigned c0 = tid < nR && tid < C && v_cnt != URF_CURB_DENSE ? v_cnt : 0u;
        const unsigned v1 = tid + 64 < C ? a.curb_cnt[(size_t)s * C + tid + 64] : 0u;
        const unsigned c1 = tid + 64 < nR && v1 != URF_CURB_DENSE ? v1 : 0u;
        const unsigned i0 = urf_wave_scan_add(c0), i1 = urf_wave_scan_add(c1);
        const unsigned t0 = (unsigned)__shfl((int)i0, 63), t1 = (unsigned)__shfl((int)i1, 63);   /* (every lane takes part in the shuffles) */
        lpre[tid] = i0 - c0;
        lpre[tid + 64] = t0 + i1 - c1;
        const unsigned long long dm0 = __ballot(tid < nR && tid < C && v_cnt == URF_CURB_DENSE), dm1 = __ballot(tid + 64 < nR && v1 == URF_CURB_DENSE);
        if (tid == 0) {
            lpre[URF_MAX_CHANNELS] = t0 + t1;
            n_dense = (unsigned)__popcll(dm0) + (unsigned)__popcll(dm1);
        }
    }
    __syncthreads();
    URF_PHASE_MARK;
    if (tid < 4 && !(dp.p.blind_spots && nR > 1))
        a.quad[(size_t)s * 4 + tid] = q[tid];
    /* The forward beam of degree i stops at the first ring k that holds a curb point with azimuth in [i, hi_k(i)]
     * (blind_spots.cpp:107-155: the sorted scan from the first point >= i finds one <= hi), the backward beam at
     * the first with one in [lo_k(i), i] (:216-273).  Turned round: hi_k and lo_k do not fall as the degree grows,
     * so a curb point (k, az) stops exactly the forward beams of the degrees [dmin, floor(az)], dmin = the smallest
     * degree with hi_k(dmin) >= az, and the backward beams of [ceil(az), dmax] -- plus the one beam whose window is
     * stretched to the end of the circle (fi == limit, rings k >= 1).  One thread per listed curb point finds dmin /
     * dmax (an estimate from the window's width, corrected with the reference's own predicate) and sets the bits of
     * the interval in the ring's mask: ~700 points per 64 x 2048 sweep, a few predicate tests and two or three
     * LDS atomics each.  (r2: two per-degree tables of 361 floats per ring, 185 KB per sweep through memory, this
     * kernel bandwidth-bound; a march through per-ring lists degree by degree compares every degree with every
     * curb point: 490 000 tests, 0.07 -> 0.16 ms.) */
    {
        const bool fl_int = dp.fwd_limit >= 0.0f && dp.fwd_limit <= 360.0f && (float)(int)dp.fwd_limit == dp.fwd_limit;
        const bool bl_int = dp.bwd_limit >= 0.0f && dp.bwd_limit <= 360.0f && (float)(int)dp.bwd_limit == dp.bwd_limit;
        auto set_bits = [&](unsigned* m12, int d0, int d1) {   /* degrees d0..d1 (inclusive), 0 <= d0 <= d1 <= 360 */
            for (int w = d0 >> 5; w <= (d1 >> 5); w++) {
                const int lo_b = d0 > w * 32 ? d0 - w * 32 : 0, hi_b = d1 < w * 32 + 31 ? d1 - w * 32 : 31;
                atomicOr(&m12[w], (0xffffffffu >> (31 - hi_b)) & (0xffffffffu << lo_b));
            }
        };
        const unsigned n_ent = lpre[URF_MAX_CHANNELS];
        for (unsigned idx = tid; idx < n_ent; idx += URF_BEAM_THREADS) {
            /* the ring of flat entry idx: the last ring whose start is <= idx and that lists something (bisection
             * over the starts; rings without entries share their successor's start and are stepped over) */
            unsigned lo_k = 0, hi_k = URF_MAX_CHANNELS;
#pragma unroll
            for (unsigned step = 0; step < 7; step++) {
                const unsigned mid = (lo_k + hi_k) >> 1;
                if (lpre[mid] <= idx)
                    lo_k = mid;
                else
                    hi_k = mid;
            }
            const unsigned k = lo_k;                       /* lpre[k] <= idx < lpre[k + 1] */
            const float az = lst[k * URF_CURB_LIST + (idx - lpre[k])];   /* in [0, 360] */
            const double qq = qk[k];
            const float wd = k == 0 ? dp.p.beamZone : (float)qq;   /* width of the window on this ring */
            if (!(wd == wd))
                continue;   /* (NaN: no comparison with such a window end holds) */
            const int a0 = (int)__builtin_floorf(az), a1 = (int)__builtin_ceilf(az);
            /* the window ends away from the limit beams (which are added below): monotone in the degree */
            auto hi_of = [&](int d) { return k == 0 ? (float)d + dp.p.beamZone : (float)((double)d + qq); };
            auto lo_of = [&](int d) { return k == 0 ? (float)d - dp.p.beamZone : (float)((double)d - qq); };
            /* (a ring with NaN azimuths: the forward scans end at its first NaN, the backward scans at its last) */
            const bool seen_f = !has_nan || az <= vfh[k], seen_b = !has_nan || az >= vbl[k];
            if (seen_f) {   /* forward: [dmin, a0] */
                const float est = __builtin_ceilf(az - wd);
                int d = !(est >= 0.0f) ? 0 : (est > (float)(a0 + 1) ? a0 + 1 : (int)est);
                int guard = 0;
                while (d > 0 && az <= hi_of(d - 1) && guard++ < 400)
                    d--;
                while (d <= a0 && !(az <= hi_of(d)) && guard++ < 800)
                    d++;
                if (d <= a0)
                    set_bits(&sfm[k * 12], d, a0);
                if (k != 0 && fl_int && (int)dp.fwd_limit <= a0)   /* fi == limit: the window reaches 360 >= az */
                    set_bits(&sfm[k * 12], (int)dp.fwd_limit, (int)dp.fwd_limit);
            }
            if (seen_b) {   /* backward: [a1, dmax] */
                const float est = __builtin_floorf(az + wd);
                int d = !(est <= 360.0f) ? 360 : (est < (float)(a1 - 1) ? a1 - 1 : (int)est);
                int guard = 0;
                while (d < 360 && az >= lo_of(d + 1) && guard++ < 400)
                    d++;
                while (d >= a1 && !(az >= lo_of(d)) && guard++ < 800)
                    d--;
                if (d >= a1)
                    set_bits(&sbm[k * 12], a1, d);
                if (k != 0 && bl_int && (int)dp.bwd_limit >= a1)   /* fi == limit: the window reaches 0 <= az */
                    set_bits(&sbm[k * 12], (int)dp.bwd_limit, (int)dp.bwd_limit);
            }
        }
    }
    __syncthreads();
    URF_PHASE_MARK;
    const int i = (int)dt;
    const bool inrange = i <= 360;
    const float fi = (float)i;
    const bool blind = !inrange || urf_blind(dp.p, q, i);
    const bool cast_f = fi <= dp.fwd_limit && !blind;   /* blind_spots.cpp:68 */
    const bool cast_b = fi >= dp.bwd_limit && !blind;   /* blind_spots.cpp:177 */
    int sf = cast_f ? (int)nR : -1, sb = cast_b ? (int)nR : -1;
    {
        const unsigned w = inrange ? dt >> 5 : 11u;
        const unsigned bit = 1u << (dt & 31);
        const bool dense_rings = n_dense != 0;   /* (uniform; normally none) */
        for (unsigned k0 = 0; k0 < nR; k0 += 8 * NH) {   /* (the words of eight rings in flight) */
            unsigned wf[8], wb[8];
#pragma unroll
            for (unsigned u = 0; u < 8; u++) {
                const unsigned k = k0 + u * NH + part < nR ? k0 + u * NH + part : nR - 1;
                wf[u] = sfm[k * 12 + w];
                wb[u] = sbm[k * 12 + w];
            }
#pragma unroll
            for (unsigned u = 0; u < 8; u++) {
                const unsigned k = k0 + u * NH + part;
                bool hf = k < nR && (wf[u] & bit) != 0, hb = k < nR && (wb[u] & bit) != 0;
                if (dense_rings && k < nR && lcnt[k] == URF_CURB_DENSE) {   /* (uniform) the ring's list overflowed: its per-degree tables */
                    float whi = urf_fwd_hi(dp, i, k, qk[k]), wlo = urf_bwd_lo(dp, i, k, qk[k]);
                    if (has_nan) {   /* (selects that keep a NaN window end a NaN) */
                        whi = vfh[k] < whi ? vfh[k] : whi;
                        wlo = vbl[k] > wlo ? vbl[k] : wlo;
                    }
                    hf = cast_f && sf == (int)nR && a.sufmin[((size_t)s * C + k) * URF_DEG_CELLS + i] <= whi;
                    hb = cast_b && sb == (int)nR && a.premax[((size_t)s * C + k) * URF_DEG_CELLS + i] >= wlo;
                }
                sf = (sf == (int)nR && hf) ? (int)k : sf;
                sb = (sb == (int)nR && hb) ? (int)k : sb;
            }
        }
    }
    if (NH > 1) {   /* the first stopping ring over all groups ("not cast" = -1 in every group, "none" = n_rings) */
        xs[0][part][dt] = (int16_t)sf;
        xs[1][part][dt] = (int16_t)sb;
        __syncthreads();
#pragma unroll
        for (unsigned h = 0; h < NH; h++) {
            const int of = xs[0][h][dt], ob = xs[1][h][dt];
            sf = of < sf ? of : sf;
            sb = ob < sb ? ob : sb;
        }
    }
    if (inrange && part == 0) {
        a.stop_f[(size_t)s * URF_DEG_CELLS + i] = (int16_t)sf;
        a.stop_b[(size_t)s * URF_DEG_CELLS + i] = (int16_t)sb;
    }
    /* Per ring k: bit i of mf / mb <=> the forward / backward beam that starts at degree i reached
     * beyond ring k.  From the masks, for every (ring, degree d): the window end of the nearest such
     * forward beam at or below d and of the nearest backward beam at or above d -- all k_label needs
     * to decide a point (windows [i, hi_k(i)] and [lo_k(i), i] move monotonically with i). */
    /* (lane u of a wave stores the words of ring k0 + u: 64 rings per round instead of one) */
    for (unsigned k0 = 0; k0 < nR; k0 += 64) {
        unsigned long long myf = 0ull, myb = 0ull;
        const unsigned kn = nR - k0 < 64u ? nR - k0 : 64u;
        for (unsigned u = part; u < kn; u += NH) {   /* (uniform per wave) */
            const unsigned long long bf = __ballot(sf > (int)(k0 + u)), bb = __ballot(sb > (int)(k0 + u));
            myf = urf_lane() == u ? bf : myf;
            myb = urf_lane() == u ? bb : myb;
        }
        if (urf_lane() < kn && urf_lane() % NH == part) {
            mf[(k0 + urf_lane()) * 6 + (dt >> 6)] = myf;
            mb[(k0 + urf_lane()) * 6 + (dt >> 6)] = myb;
        }
    }
    __syncthreads();
    URF_PHASE_MARK;
    /* highest set forward bit in the words below word w / lowest set backward bit in the words above */
    for (unsigned e = tid; e < nR * 6; e += URF_BEAM_THREADS) {
        const unsigned k = e / 6, w = e % 6;
        int below = -1, above = -1;
        for (unsigned v = 0; v < w; v++)
            if (mf[k * 6 + v])
                below = (int)(v * 64 + 63 - __clzll((long long)mf[k * 6 + v]));
        for (unsigned v = 5; v > w; v--)
            if (mb[k * 6 + v])
                above = (int)(v * 64 + __ffsll((long long)mb[k * 6 + v]) - 1);
        pf[e] = (int16_t)below;
        nb[e] = (int16_t)above;
    }
    __syncthreads();
    URF_PHASE_MARK;
    if (inrange) {
        const unsigned w = dt >> 6, b = dt & 63;
        const unsigned long long le = b == 63 ? ~0ull : ((2ull << b) - 1ull), ge = ~0ull << b;
        urf_win* win = a.win + (size_t)s * C * URF_DEG_CELLS + i;
        for (unsigned k0 = 0; k0 < nR; k0 += 4 * NH) {   /* four rings at a time: their masks and ratios read before any is used */
            unsigned long long f4[4], g4[4];
            int p4[4], n4[4];
            double q4[4];
#pragma unroll
            for (unsigned u = 0; u < 4; u++) {
                const unsigned k = k0 + u * NH + part < nR ? k0 + u * NH + part : nR - 1;
                f4[u] = mf[k * 6 + w] & le;
                g4[u] = mb[k * 6 + w] & ge;
                p4[u] = (int)pf[k * 6 + w];
                n4[u] = (int)nb[k * 6 + w];
                q4[u] = qk[k];
            }
#pragma unroll
            for (unsigned u = 0; u < 4; u++) {
                const unsigned k = k0 + u * NH + part;
                if (k >= nR)
                    break;
                const int jf = f4[u] ? (int)(w * 64 + 63 - __clzll((long long)f4[u])) : p4[u];
                const int jb = g4[u] ? (int)(w * 64 + __ffsll((long long)g4[u]) - 1) : n4[u];
                urf_win o;
                o.hi = jf >= 0 ? urf_fwd_hi(dp, jf, k, q4[u]) : -__builtin_inff();
                o.lo = jb >= 0 ? urf_bwd_lo(dp, jb, k, q4[u]) : __builtin_inff();
                if (has_nan) {   /* (uniform) a beam marks only what its scan of the ring sees (blind_spots.cpp:124,164,233,273) */
                    o.hi = vfh[k] < o.hi ? vfh[k] : o.hi;
                    o.lo = vbl[k] > o.lo ? vbl[k] : o.lo;
                }
                win[(size_t)k * URF_DEG_CELLS] = o;
            }
        }
    }
    URF_PHASE_MARK;
    URF_PHASE_DUMP("k_beams");
}

/* ------------------------------------------------------------------------- */
/* k_label                                                                     */
/* ------------------------------------------------------------------------- */
/* A point of ring k is road iff it is no curb point and lies in the window of
 * a beam that reached beyond ring k.  Windows [i, hi_k(i)] grow with i, so it
 * suffices to test the largest such forward beam with i <= azimuth (and the
 * smallest such backward beam with i >= azimuth).
 *
 * One workgroup per input tile, so that the label bytes leave as whole cache
 * lines: the ring-major runs that belong to the tile (the split is stable, so
 * each ring contributes one contiguous run: tile_ring) are read run by run,
 * the labels are placed by input index into an LDS image of the tile and the
 * image is written out in input order. */
/* byte image of the tile's labels; consecutive ring-major slots of an organised sweep lie 64
 * bytes apart in input order, so the row (i >> 6) rotates the column (i & 63) to spread the
 * byte stores over the LDS banks */
/* Is a non-curb point with azimuth az road?  win: the point's ring's row of k_beams' window table.
 * With eps > 0 the azimuth is only known to within eps: `unsure` is set when a decision taken
 * here (floor, ceil, either window comparison) could come out differently for the true value.
 * A NaN azimuth fails both comparisons (blind_spots.cpp:128,237 compare it the same way). */
__device__ __forceinline__ bool urf_road_test(const urf_win* __restrict__ win, float az, float eps, bool& unsure)
{
    const float fl = __builtin_floorf(az), ce = __builtin_ceilf(az);
    const bool num = az == az;
    int cf = num ? (int)fl : 0, cb = num ? (int)ce : 0;
    cf = cf < 0 ? 0 : (cf > 360 ? 360 : cf);
    cb = cb < 0 ? 0 : (cb > 360 ? 360 : cb);
    const float hi = win[cf].hi, lo = win[cb].lo;
    const bool road = az <= hi || az >= lo;
    unsure = eps > 0.0f && (az - fl <= eps || (fl + 1.0f) - az <= eps || __builtin_fabsf(az - hi) <= eps ||
                            __builtin_fabsf(az - lo) <= eps);
    return road;
}

/* the same decision on the exact azimuth of the point in ring-sorted slot `slot`: bit 0 = road, bit 1 =
 * the azimuth is NaN (x == y == 0: counted by the caller, urf_scan_info::n_nan_azimuth) */
__device__ __noinline__ unsigned urf_road_exact(const urf_kargs& a, const urf_win* win, unsigned slot)
{
    float d2;
    bool unsure;
    const float az = urf_azimuth(a.rx[slot], a.ry[slot], &d2);
    return (urf_road_test(win, az, 0.0f, unsure) ? 1u : 0u) | (az == az ? 0u : 2u);
}

#define URF_LABEL_UNSURE 256   /* capacity of the list of points decided on the exact azimuth */
/* byte image of the tile's labels in input order, every 64-byte row followed by four spare bytes:
 * consecutive ring-major slots of an organised sweep lie 64 bytes apart in input order and so land
 * in different LDS banks, while four consecutive labels still form one aligned word */
#define URF_IMG(i) ((i) + 4u * ((i) >> 6))
#ifndef URF_LABEL_WAVES
#define URF_LABEL_WAVES 8
#endif
__global__ __launch_bounds__(URF_LABEL_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(URF_LABEL_WAVES, URF_LABEL_WAVES))) void k_label(urf_kargs a, urf_dev_params dp);   /* (below: the tile's body first) */
__device__ __forceinline__ void urf_label_tile(const urf_kargs& a, const urf_dev_params& dp, const unsigned s, const unsigned t)
{
    __shared__ unsigned koff[URF_MAX_CHANNELS + 1];
    __shared__ uint8_t img[URF_TILE + URF_TILE / 16 + 4] __attribute__((aligned(8)));   /* + a spare byte for the slots past the tile's last */
    __shared__ uint8_t ring_of[URF_TILE] __attribute__((aligned(8)));
    __shared__ unsigned wave_max[URF_LABEL_TILE_THREADS / 64];
    __shared__ unsigned cnt_road, cnt_curb, n_unsure;
    __shared__ unsigned un_pos[URF_LABEL_UNSURE], un_key[URF_LABEL_UNSURE];   /* points to decide on the exact azimuth */
    const unsigned tid = threadIdx.x;
    if (a.front && a.front_ok[s])
        return;   /* (uniform) a scan of the fused front end (urf_front.hpp: k_label_front) */
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned tbase = t * URF_TILE;
    if (tbase >= len)
        return;
    const unsigned C = (unsigned)dp.p.channels;
    const size_t row = (size_t)s * a.tiles + t;
    const unsigned sb = urf_sbase(a, s);
    constexpr unsigned Q = URF_TILE / URF_LABEL_TILE_THREADS;
    /* The scan summary and the tile's run table are requested together (the run of ring `tid` inside
     * the tile's ring-sorted order starts at slot troff[tid]).  (Also requesting the beam masks and
     * the slots' input indices up front was measured: no gain, and beyond 72 registers the kernel
     * loses a resident workgroup.) */
    const urf_scan_info in = a.info[s];
    const unsigned troi = a.tile_roi[row];
    const unsigned v_koff = tid <= C ? (unsigned)a.troff[row * (C + 1) + tid] : 0;
    /* ... and so are the records of the thread's eight slots (their addresses depend on nothing but the
     * thread's index: requested behind the tables they arrive while the rings of the slots are worked out) */
    const unsigned slot0 = sb + tbase + tid;   /* ring-sorted slot of point q: slot0 + 256 q (record, x, y) */
    unsigned rec[Q];   /* URF_REC_*: index inside the tile | detector hits | approximate azimuth */
#pragma unroll
    for (unsigned q = 0; q < Q; q++)
        rec[q] = a.rec[slot0 + q * URF_LABEL_TILE_THREADS];
    if (in.status != URF_OK || troi == 0) {
        /* nothing is published for this scan (lidar_segmentation.cpp:124-126), or no point of the tile
         * lies in the region of interest (k_split; the reference's default region drops whole azimuth
         * ranges of a sweep): all labels 0 */
        uint8_t* out0 = a.labels + off + tbase;
        if (tbase + URF_TILE <= len && ((uintptr_t)out0 & 7u) == 0) {   /* uniform */
            ((uint2*)out0)[tid] = make_uint2(0u, 0u);
        } else {
            for (unsigned i = tbase + tid; i < len && i < tbase + URF_TILE; i += URF_LABEL_TILE_THREADS)
                a.labels[off + i] = 0;
        }
        return;
    }
    if (tid <= C)
        koff[tid] = v_koff;
    const urf_win* win = a.win + (size_t)s * C * URF_DEG_CELLS;
    {
        /* the image starts as the labels of points on no ring: the region-of-interest flag or nothing
         * (k_split left the tile's 2048 bits); this thread's byte covers input points 8 tid .. 8 tid + 7 */
        static_assert(URF_TILE == 8 * URF_LABEL_TILE_THREADS, "one byte of the bitmap per thread");
        const unsigned bits = ((const uint8_t*)(a.roi_bits + row * (URF_TILE / 64)))[tid];
        unsigned* i32 = (unsigned*)img + 2 * tid + (tid >> 3);
        i32[0] = (((bits & 15u) * 0x00204081u) & 0x01010101u) * URF_FLAG_ROI;
        i32[1] = (((bits >> 4) * 0x00204081u) & 0x01010101u) * URF_FLAG_ROI;
        ((unsigned*)ring_of)[tid] = 0;
        ((unsigned*)ring_of)[tid + URF_LABEL_TILE_THREADS] = 0;
    }
    if (tid == 0) {
        cnt_road = 0;
        cnt_curb = 0;
        n_unsure = 0;
    }
    /* An ORGANISED tile (k_split: every ring holds 2048 / C consecutive slots; every tile of a sweep in firing order) needs no
     * table of its slots' rings: slot j lies on ring j / (2048 / C).  Decided from the run table itself; such a tile skips the
     * marks, the prefix maximum and two of the three barriers (r5). */
    const unsigned logP = 11u - (31u - (unsigned)__clz((int)C));   /* log2(2048 / C) for a power of two */
#ifdef URF_EXP_LABEL_NO_ORG
    const bool organised = (__syncthreads_or(1) != 0) && false;
#else
    const bool organised = __syncthreads_or(((C & (C - 1u)) != 0u) || (tid <= C && v_koff != (tid << logP))) == 0;
#endif
    const unsigned npts = koff[C];
    /* Ring of every slot of the tile's ring-sorted order: each non-empty run marks its first slot
     * with ring + 1, a prefix maximum over the slots spreads the marks (runs are in ring order).
     * Eight consecutive slots per thread, shuffles across the wave, LDS across the four waves --
     * a seventh of the instructions of a bisection in koff per point. */
    if (!organised) {   /* (uniform) */
    if (tid < C && koff[tid + 1] > koff[tid])
        ring_of[koff[tid]] = (uint8_t)(tid + 1);
    __syncthreads();
    {
        static_assert(URF_TILE == 8 * URF_LABEL_TILE_THREADS, "eight slots per thread");
        unsigned* w32 = (unsigned*)ring_of;
        const unsigned w0 = w32[2 * tid], w1 = w32[2 * tid + 1];
        unsigned m[8];
#pragma unroll
        for (unsigned e = 0; e < 4; e++) {
            m[e] = (w0 >> (8 * e)) & 0xffu;
            m[4 + e] = (w1 >> (8 * e)) & 0xffu;
        }
#pragma unroll
        for (unsigned e = 1; e < 8; e++)
            m[e] = m[e] > m[e - 1] ? m[e] : m[e - 1];
        const unsigned inc = urf_wave_scan_max(m[7]);
        if ((tid & 63) == 63)
            wave_max[tid >> 6] = inc;
        unsigned pre = __shfl_up(inc, 1);
        if ((tid & 63) == 0)
            pre = 0;
        __syncthreads();
        static_assert(URF_LABEL_TILE_THREADS == 256, "four waves");
#pragma unroll
        for (unsigned w = 0; w < 3; w++) {
            const unsigned m = w < (tid >> 6) ? wave_max[w] : 0u;
            pre = m > pre ? m : pre;
        }
        unsigned o0 = 0, o1 = 0;
#pragma unroll
        for (unsigned e = 0; e < 4; e++) {
            o0 |= (m[e] > pre ? m[e] : pre) << (8 * e);
            o1 |= (m[4 + e] > pre ? m[4 + e] : pre) << (8 * e);
        }
        w32[2 * tid] = o0;
        w32[2 * tid + 1] = o1;
    }
    __syncthreads();
    }
    unsigned my_road = 0, my_curb = 0;
    if (npts != 0) {   /* uniform; 0: no point of the tile lies on a ring */
    /* Straight-line per point: slots past the tile's last one read whatever the scratch holds there
     * (the tile's 2048 slots are allocated, every table index is clamped) and drop their result into a
     * spare byte of the image. */
    /* the window ends of all eight points are requested before any of them is looked at (point after
     * point the workgroup sat through eight dependent round trips to the table here) */
#ifndef URF_LABEL_QB
#define URF_LABEL_QB 4   /* r2 (three arrays per slot): 2 at 8 waves per SIMD 0.382 ms, 4 at 7 waves (72 registers) 0.390, 4 at 8 waves (48 B of
                          * scratch) 0.477, point by point 0.425; r3 (one record per slot, 63 registers at 4): 2 -> 0.373, 4 -> 0.359 */
#endif
    constexpr unsigned QB = URF_LABEL_QB;   /* points per batch of table requests */
#pragma unroll
    for (unsigned q0 = 0; q0 < Q; q0 += QB) {
    float whi[QB], wlo[QB];
    unsigned cc[QB];
#pragma unroll
    for (unsigned qq = 0; qq < QB; qq++) {
        const unsigned q = q0 + qq;
        const unsigned j = tid + q * URF_LABEL_TILE_THREADS;
        cc[qq] = organised ? j >> logP : (unsigned)ring_of[j] - 1u;   /* (past the last slot: the last ring, from the prefix maximum) */
        const float az = urf_az_decode(rec[q] >> URF_REC_AZ_SHIFT);
        const bool num = az == az;
        int cf = num ? (int)__builtin_floorf(az) : 0, cb = num ? (int)__builtin_ceilf(az) : 0;
        cf = cf < 0 ? 0 : (cf > 360 ? 360 : cf);
        cb = cb < 0 ? 0 : (cb > 360 ? 360 : cb);
        whi[qq] = win[cc[qq] * URF_DEG_CELLS + cf].hi;
        wlo[qq] = win[cc[qq] * URF_DEG_CELLS + cb].lo;
    }
#pragma unroll
    for (unsigned qq = 0; qq < QB; qq++) {
        const unsigned q = q0 + qq;
        const unsigned j = tid + q * URF_LABEL_TILE_THREADS;
        const bool valid = j < npts;
        const unsigned c = cc[qq];
        const unsigned src = rec[q] & URF_REC_SRC_MASK;
        const bool curb = ((rec[q] >> URF_REC_FLAG_SHIFT) & 7u) != 0;
        /* The record holds k_split's float approximation of the azimuth, quantised (error <=
         * urf_fast_az_eps + URF_REC_AZ_QERR).  Every decision that the approximation clears by that
         * margin is the reference's decision; the rare point that does not is listed and decided
         * below on the exact azimuth. */
        const float az = urf_az_decode(rec[q] >> URF_REC_AZ_SHIFT), eps = urf_fast_az_eps(az) + URF_REC_AZ_QERR;
        const float fl = __builtin_floorf(az);
        bool road = az <= whi[qq] || az >= wlo[qq];   /* urf_road_test with the window ends at hand */
        /* (URF_AZ_UNKNOWN = -1: k_split had no usable approximation -- the point lies too close to the x axis) */
        const bool unsure = az < 0.0f || az - fl <= eps || (fl + 1.0f) - az <= eps ||
                            __builtin_fabsf(az - whi[qq]) <= eps || __builtin_fabsf(az - wlo[qq]) <= eps;
        if (unsure && valid && !curb) {
            const unsigned e = atomicAdd(&n_unsure, 1u);
            if (e < URF_LABEL_UNSURE) {
                un_pos[e] = slot0 + q * URF_LABEL_TILE_THREADS;
                un_key[e] = src | (c << 16);
                road = false;   /* placeholder, corrected after the tile is written */
            } else {
                const unsigned re = urf_road_exact(a, win + c * URF_DEG_CELLS, slot0 + q * URF_LABEL_TILE_THREADS);   /* list full (pathological input) */
                road = re & 1u;
                if (re & 2u)
                    atomicAdd(&a.info[s].n_nan_azimuth, 1u);
            }
        }
        road = road && !curb;
        const unsigned lab = URF_FLAG_ROI | URF_FLAG_RING | (c == 10 ? URF_FLAG_RING10 : 0) |
                             (curb ? URF_LABEL_CURB : 0) | (road ? URF_LABEL_ROAD : 0);
        my_curb += (valid && curb) ? 1u : 0u;
        my_road += (valid && road) ? 1u : 0u;
        img[valid ? URF_IMG(src & (URF_TILE - 1u)) : URF_TILE + URF_TILE / 16] = (uint8_t)lab;
    }
    }
    }
    __syncthreads();
    static_assert(URF_LABEL_UNSURE <= URF_LABEL_TILE_THREADS, "one listed point per thread");
    /* The listed points: their coordinates are requested now and used after the tile has been
     * written, so that the round trip hides behind the stores (URF_LABEL_UNSURE <= workgroup size). */
    const unsigned nu = n_unsure < URF_LABEL_UNSURE ? n_unsure : URF_LABEL_UNSURE;
    const bool tail = tid < nu;
    float tx = 0.f, ty = 0.f;
    unsigned tkey = 0;
    if (tail) {
        tkey = un_key[tid];
        tx = a.rx[un_pos[tid]];
        ty = a.ry[un_pos[tid]];
    }
    {
        uint8_t* out = a.labels + off + tbase;
        if (tbase + URF_TILE <= len && ((uintptr_t)out & 3u) == 0) {   /* uniform */
#pragma unroll
            for (unsigned r = 0; r < URF_TILE / 4 / URF_LABEL_TILE_THREADS; r++) {
                const unsigned k = tid + r * URF_LABEL_TILE_THREADS;
                ((unsigned*)out)[k] = ((const unsigned*)img)[k + (k >> 4)];
            }
        } else {
            for (unsigned i = tid; i < URF_TILE && tbase + i < len; i += URF_LABEL_TILE_THREADS)
                out[i] = img[URF_IMG(i)];
        }
    }
    __syncthreads();   /* the tile's stores come first, the corrections second */
    if (tail) {
        const unsigned c = tkey >> 16, li = tkey & 0xffffu;
        bool unsure;
        float d2;
        const float az = urf_azimuth(tx, ty, &d2);
        if (urf_road_test(win + c * URF_DEG_CELLS, az, 0.0f, unsure)) {
            a.labels[off + tbase + li] = URF_FLAG_ROI | URF_FLAG_RING | (c == 10 ? URF_FLAG_RING10 : 0) | URF_LABEL_ROAD;
            my_road++;
        }
        if (!(az == az))   /* x == y == 0: a NaN azimuth (include/urf.h: n_nan_azimuth), counted per scan */
            atomicAdd(&a.info[s].n_nan_azimuth, 1u);
    }
    if (my_road)
        atomicAdd(&cnt_road, my_road);
    if (my_curb)
        atomicAdd(&cnt_curb, my_curb);
    __syncthreads();
    if (tid == 0) {
        urf_scan_info* o = &a.info[s];
        if (cnt_road)
            atomicAdd(&o->n_road, cnt_road);
        if (cnt_curb)
            atomicAdd(&o->n_curb, cnt_curb);
    }
}

__global__ __launch_bounds__(URF_LABEL_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(URF_LABEL_WAVES, URF_LABEL_WAVES))) void k_label(urf_kargs a, urf_dev_params dp)
{
    /* Workgroups are handed to the eight XCDs round robin, each XCD with its own L2.  The tiles of a
     * scan all read the scan's window table (185 KB): spread over the XCDs every L2 fetched most of it
     * (5.9 B per point of this kernel's 13.9); with the mapping below the tiles of one scan run on ONE
     * XCD (eight scans at a time, one per XCD) and the table comes from memory once. */
    unsigned s = blockIdx.y, t = blockIdx.x;
    {
        const unsigned T = gridDim.x, lin = blockIdx.y * T + blockIdx.x;
        const unsigned grp = lin / (8u * T), r = lin - grp * (8u * T);
        if ((grp + 1u) * 8u <= gridDim.y) {   /* a complete group of eight scans */
            s = grp * 8u + (r & 7u);
            t = r >> 3;
        }
    }
    urf_label_tile(a, dp, s, t);
}
/* the scans the fused front end handed back (k_ring_list): persistent workgroups over list x tiles */
__global__ __launch_bounds__(URF_LABEL_TILE_THREADS) void k_label_list(urf_kargs a, urf_dev_params dp)
{
    const unsigned n = a.star_count[6];
    for (unsigned w = blockIdx.x; w < n * a.tiles; w += gridDim.x) {
        urf_label_tile(a, dp, a.front_list[w / a.tiles], w % a.tiles);
        __syncthreads();   /* the LDS is reused by the next tile */
    }
}

/* exact azimuth of the point in ring-sorted slot `slot` (its record holds an approximation) */
__device__ __forceinline__ float urf_exact_az(const urf_kargs& a, unsigned slot)
{
    float d2;
    return urf_azimuth(a.rx[slot], a.ry[slot], &d2);
}

/* ------------------------------------------------------------------------- */
/* index lists                                                                 */
/* ------------------------------------------------------------------------- */
/* lidar_segmentation.cpp:354-367, 605-608, 620 as index sets: for every scan of a batch the
 * ascending lists of the input indices of its road / curb / roi / road_probably points.
 * Workgroup (t, s) = tile t (2048 labels) of scan s.  k_compact_count: the tile's four counts;
 * k_compact_write: the tile's first position in each list = the counts of the tiles before it, then
 * ranks inside the tile by ballot + prefix, eight rounds of 256 labels (ascending order kept). */
#define URF_COMPACT_THREADS 256
__device__ __forceinline__ unsigned urf_label_classes(unsigned l)
{
    return ((l & URF_LABEL_MASK) == URF_LABEL_ROAD ? 1u : 0u) | ((l & URF_LABEL_MASK) == URF_LABEL_CURB ? 2u : 0u) |
           ((l & URF_FLAG_ROI) ? 4u : 0u) | ((l & URF_FLAG_RING10) ? 8u : 0u);
}
__global__ __launch_bounds__(URF_COMPACT_THREADS) void k_compact_count(const uint8_t* __restrict__ labels, unsigned n_per_scan,
                                                                         unsigned tiles, unsigned* __restrict__ tile_cnt)
{
    __shared__ unsigned sh[4];
    const unsigned t = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const uint8_t* L = labels + (size_t)s * n_per_scan;
    if (tid < 4)
        sh[tid] = 0;
    __syncthreads();
    unsigned c[4] = { 0, 0, 0, 0 };
    for (unsigned r = 0; r < URF_TILE / URF_COMPACT_THREADS; r++) {
        const unsigned i = t * URF_TILE + r * URF_COMPACT_THREADS + tid;
        const unsigned f = i < n_per_scan ? urf_label_classes(L[i]) : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            c[k] += (unsigned)__popcll(__ballot((f >> k) & 1u));
    }
    if (urf_lane() == 0)
#pragma unroll
        for (int k = 0; k < 4; k++)
            atomicAdd(&sh[k], c[k]);
    __syncthreads();
    if (tid < 4)
        tile_cnt[((size_t)s * tiles + t) * 4 + tid] = sh[tid];
}
__global__ __launch_bounds__(URF_COMPACT_THREADS) void k_compact_write(const uint8_t* __restrict__ labels, unsigned n_per_scan,
                                                                         unsigned tiles, const unsigned* __restrict__ tile_cnt,
                                                                         unsigned* road, unsigned* curb, unsigned* roi,
                                                                         unsigned* ring10, unsigned* counts)
{
    __shared__ unsigned run[4], wsum[4][URF_COMPACT_THREADS / 64];
    const unsigned t = blockIdx.x, s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* L = labels + (size_t)s * n_per_scan;
    if (tid < 64) {   /* the tiles before this one: lane k + 4 j sums every 16th tile of class k */
        const unsigned k = tid & 3u;
        unsigned sum = 0;
        for (unsigned u = tid >> 2; u < t; u += 16)
            sum += tile_cnt[((size_t)s * tiles + u) * 4 + k];
        sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 8);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (tid < 4)
            run[tid] = sum;
    }
    __syncthreads();
    unsigned* outs[4] = { road, curb, roi, ring10 };
    for (unsigned r = 0; r < URF_TILE / URF_COMPACT_THREADS; r++) {
        const unsigned i = t * URF_TILE + r * URF_COMPACT_THREADS + tid;
        const unsigned f = i < n_per_scan ? urf_label_classes(L[i]) : 0u;
        unsigned below[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot((f >> k) & 1u);
            below[k] = urf_popc_below(m);
            if (lane == 0)
                wsum[k][wave] = (unsigned)__popcll(m);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned pre = run[k];
            for (unsigned w = 0; w < wave; w++)
                pre += wsum[k][w];
            if (((f >> k) & 1u) && outs[k])
                outs[k][(size_t)s * n_per_scan + pre + below[k]] = i;
        }
        __syncthreads();
        if (tid < 4)
            run[tid] += wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
        __syncthreads();
    }
    if (t + 1 == tiles && tid < 4 && counts)
        counts[(size_t)s * 4 + tid] = run[tid];
}

/* ------------------------------------------------------------------------- */
/* published order                                                             */
/* ------------------------------------------------------------------------- */
/* The reference sorts every ring by azimuth (lidar_segmentation.cpp:70-93, 289-291) and fills
 * its road / curb / road_probably clouds ring by ring in that order (:354-367, 605-608).  The
 * labels do not need that sort; callers that want the clouds in the reference's order do.
 * k_ring_order: one workgroup per ring of ONE scan sorts (azimuth bits, position in the ring)
 * and writes the ring-major position of the i-th point of the ring in azimuth order; a ring in
 * which two points share their azimuth bit for bit is then sorted AGAIN, literally as the
 * reference's Lomuto quicksort does it, whose order of equal azimuths is what gets published
 * (r5).  Rings of up to 2048 points sort in LDS, longer ones in global memory. */
__global__ __launch_bounds__(256) void k_ring_order(urf_kargs a, urf_dev_params dp, unsigned s0,
                                                    unsigned long long* gkeys_all, unsigned* rord_all, unsigned* rcls_all)
{
    constexpr unsigned NT = 256, NB = 2048, EPT = 8, CAP = NT * EPT;
    __shared__ unsigned long long A[CAP];
    __shared__ unsigned cnt[URF_BLOCK_CNT(NB, NT)];
    __shared__ urf_sort_shared ssh;
    __shared__ unsigned ncls[2], sh_tie;
    __shared__ int lom_stk[2 * 64];
    extern __shared__ unsigned sh_ord_tab[];   /* P[tiles + 1], radd[tiles] (urf_ring_map) */
    const unsigned c = blockIdx.x, s = s0 + blockIdx.y, tid = threadIdx.x;
    unsigned long long* gkeys = gkeys_all + (size_t)blockIdx.y * a.sstride;   /* per scan of the launch: sstride entries */
    unsigned* rord = rord_all + (size_t)blockIdx.y * a.sstride;
    unsigned* rcls = rcls_all + ((size_t)blockIdx.y * URF_MAX_CHANNELS + c) * 2;   /* road / curb points of the ring */
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK || c >= in.n_rings) {
        if (tid < 2)
            rcls[tid] = 0;
        return;
    }
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels;
    const unsigned n = a.ring_cnt[(size_t)s * C + c];
    const unsigned rel = a.ring_off[(size_t)s * (C + 1) + c];   /* scan-relative start of the ring */
    const unsigned sb = urf_sbase(a, s);
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    /* the ring's run table in LDS (k_ring's map): position in the ring -> ring-sorted slot without a
     * bisection in global memory */
    unsigned* const mapP = sh_ord_tab;
    unsigned* const mapA = sh_ord_tab + a.tiles + 1;
    {
        const unsigned* gp = a.rpre + ((size_t)s * C + c) * (a.tiles + 1);
        const uint16_t* gs = a.rstart + ((size_t)s * C + c) * a.tiles;
        for (unsigned t = tid; t <= ntiles; t += NT) {
            const unsigned pt = gp[t];
            mapP[t] = pt;
            if (t < ntiles)
                mapA[t] = t * URF_TILE + gs[t] - pt;   /* relative to the scan's scratch base */
        }
    }
    if (tid < 2)
        ncls[tid] = 0;
    if (tid == 0)
        sh_tie = 0;
    __syncthreads();
    const urf_ring_map map = { mapP, mapA, ntiles, (float)ntiles / (float)(n > 0 ? n : 1) };
    /* Two points of the ring with bit-identical azimuths: their order is the one the reference's Lomuto quicksort
     * (lidar_segmentation.cpp:70-93; deterministic, not stable) leaves.  SORTED = the (azimuth, position) keys in
     * ascending order; if two neighbours share their azimuth the keys go back into bucket order (the ring's stretch of wsg,
     * which nobody reads after k_star_walk), one wave runs the quicksort literally (urf_lomuto_sort, as k_nan_rings does for
     * rings with NaN azimuths) and the ring is published in that order. */
    volatile unsigned long long* const LIT = (volatile unsigned long long*)(a.wsg + sb + rel);
    auto literal_order = [&](const unsigned long long* SORTED) -> bool {
        for (unsigned j = tid; j + 1 < n; j += NT)
            if ((unsigned)(SORTED[j] >> 32) == (unsigned)(SORTED[j + 1] >> 32))
                sh_tie = 1u;
        __syncthreads();
        if (!sh_tie)
            return false;   /* (uniform) */
        for (unsigned j = tid; j < n; j += NT) {
            const unsigned long long k = SORTED[j];
            LIT[(unsigned)k] = k;
        }
        __threadfence_block();
        __syncthreads();
        if (tid < 64)
            urf_lomuto_sort(LIT, n, lom_stk);
        __threadfence_block();
        __syncthreads();
        return true;
    };
    /* what is published for position i of the ring: the point's input index | its class << 30 */
    auto entry_of = [&](unsigned i, unsigned& cls) {
        const unsigned slot = map.at(i);
        const unsigned src = (slot & ~(URF_TILE - 1u)) + (a.rec[sb + slot] & URF_REC_SRC_MASK);
        cls = a.labels[off + src] & URF_LABEL_MASK;
        return src | (cls << 30);
    };
    unsigned my_road = 0, my_curb = 0;
    if ((a.nan_mask[(size_t)s * 4 + (c >> 5)] >> (c & 31u)) & 1u) {
        /* (uniform) a ring with NaN azimuths: k_nan_rings ran the reference's quicksort literally and left the ring in
         * its final order -- where a NaN lands is no function of the azimuths */
        for (unsigned j = tid; j < n; j += NT) {
            unsigned cls;
            rord[rel + j] = entry_of(a.ssrt[sb + rel + j], cls);
            my_road += cls == URF_LABEL_ROAD;
            my_curb += cls == URF_LABEL_CURB;
        }
    } else if (n <= CAP) {
        unsigned long long key[EPT];
        unsigned slot[EPT];
        float px[EPT], py[EPT];
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {   /* coordinates of the thread's eight points in flight together */
            const unsigned i = tid + e * NT;
            slot[e] = i < n ? map.at(i) : 0u;
            px[e] = a.rx[sb + slot[e]];
            py[e] = a.ry[sb + slot[e]];
        }
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const unsigned i = tid + e * NT;
            key[e] = ~0ull;
            if (i < n) {
                float d2;   /* the slot's record holds an approximation: the published order is that of the exact azimuth */
                key[e] = ((unsigned long long)urf_fbits(urf_azimuth(px[e], py[e], &d2)) << 32) | i;
            }
        }
        urf_block_sort_keys<NT, EPT, NB>(key, n, A, cnt, &ssh, false);
        const bool lit = literal_order(A);
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const unsigned j = tid + e * NT;
            if (j < n) {
                unsigned cls;
                rord[rel + j] = entry_of(lit ? (unsigned)LIT[j] : (unsigned)A[j], cls);
                my_road += cls == URF_LABEL_ROAD;
                my_curb += cls == URF_LABEL_CURB;
            }
        }
    } else {
        unsigned long long* G = gkeys + rel;
        for (unsigned i = tid; i < n; i += NT)
            G[i] = ((unsigned long long)urf_fbits(urf_exact_az(a, sb + map.at(i))) << 32) | i;
        __threadfence_block();
        __syncthreads();
        unsigned P = 1;
        while (P < n)
            P <<= 1;
        for (unsigned kk = 2; kk <= P; kk <<= 1)
            for (unsigned j = kk >> 1; j > 0; j >>= 1) {
                const bool flip = (j == (kk >> 1));
                for (unsigned tt = tid; tt < (P >> 1); tt += NT) {
                    const unsigned lo = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
                    const unsigned hi = flip ? ((lo & ~(kk - 1)) + (kk - 1) - (lo & (kk - 1))) : lo + j;
                    if (hi < n) {
                        const unsigned long long ka = G[lo], kb = G[hi];
                        if (ka > kb) {
                            G[lo] = kb;
                            G[hi] = ka;
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        const bool lit = literal_order(G);
        for (unsigned j = tid; j < n; j += NT) {
            unsigned cls;
            rord[rel + j] = entry_of(lit ? (unsigned)LIT[j] : (unsigned)G[j], cls);
            my_road += cls == URF_LABEL_ROAD;
            my_curb += cls == URF_LABEL_CURB;
        }
    }
    if (my_road)
        atomicAdd(&ncls[0], my_road);
    if (my_curb)
        atomicAdd(&ncls[1], my_curb);
    __syncthreads();
    if (tid < 2)
        rcls[tid] = ncls[tid];
}

/* The lists of one scan = its rings in order, every ring in azimuth order (k_ring_order left, per ring
 * position, the point's input index and class, and per ring the number of road / curb points): workgroup
 * (ring, scan) finds where its ring starts in each list (the counts of the rings in front of it) and
 * appends its points in order -- ballot + prefix per 256 entries.  (One workgroup per SCAN walking all
 * ring points with three barriers per 1024 of them took 1.5 ms per 1024 sweeps.) */
__global__ __launch_bounds__(256) void k_ordered_lists(urf_kargs a, urf_dev_params dp, unsigned s0, const unsigned* rord_all,
                                                       const unsigned* rcls_all, unsigned* road_all, unsigned* curb_all,
                                                       unsigned* ring10_all, unsigned stride, unsigned* counts_all)
{
    const unsigned c = blockIdx.x, s = s0 + blockIdx.y;
    const unsigned* rord = rord_all + (size_t)blockIdx.y * a.sstride;
    const unsigned* rcls = rcls_all + (size_t)blockIdx.y * URF_MAX_CHANNELS * 2;
    unsigned* road = road_all ? road_all + (size_t)blockIdx.y * stride : nullptr;
    unsigned* curb = curb_all ? curb_all + (size_t)blockIdx.y * stride : nullptr;
    unsigned* ring10 = ring10_all ? ring10_all + (size_t)blockIdx.y * stride : nullptr;
    unsigned* counts = counts_all + (size_t)blockIdx.y * 3;
    __shared__ unsigned wsum[2][4];
    __shared__ unsigned base[2];
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const urf_scan_info in = a.info[s];
    const unsigned C = (unsigned)dp.p.channels;
    const unsigned nR = in.status == URF_OK ? in.n_rings : 0;
    if (c >= nR) {
        if (c == 0 && tid < 3)
            counts[tid] = 0;   /* nothing is published for this scan */
        return;
    }
    if (tid < 2)
        base[tid] = 0;
    __syncthreads();
    if (tid < c) {   /* c <= 127 rings in front */
        atomicAdd(&base[0], rcls[2 * tid]);
        atomicAdd(&base[1], rcls[2 * tid + 1]);
    }
    __syncthreads();
    unsigned run0 = base[0], run1 = base[1];
    const unsigned n = a.ring_cnt[(size_t)s * C + c];
    const unsigned rel = a.ring_off[(size_t)s * (C + 1) + c];
    for (unsigned j0 = 0; j0 < n; j0 += 256) {
        const unsigned j = j0 + tid;
        const unsigned ent = j < n ? rord[rel + j] : 0u;
        const unsigned src = ent & 0x3fffffffu, cls = j < n ? ent >> 30 : 0u;
        const unsigned long long m0 = __ballot(cls == URF_LABEL_ROAD), m1 = __ballot(cls == URF_LABEL_CURB);
        if (lane == 0) {
            wsum[0][wave] = (unsigned)__popcll(m0);
            wsum[1][wave] = (unsigned)__popcll(m1);
        }
        __syncthreads();
        unsigned p0 = run0 + urf_popc_below(m0), p1 = run1 + urf_popc_below(m1);
#pragma unroll
        for (unsigned w = 0; w < 4; w++) {
            p0 += w < wave ? wsum[0][w] : 0u;
            p1 += w < wave ? wsum[1][w] : 0u;
            run0 += wsum[0][w];
            run1 += wsum[1][w];
        }
        if (cls == URF_LABEL_ROAD && road)
            road[p0] = src;
        if (cls == URF_LABEL_CURB && curb)
            curb[p1] = src;
        if (c == 10 && j < n && ring10)   /* lidar_segmentation.cpp:605-608: every point of sorted ring 10 */
            ring10[j] = src;
        __syncthreads();
    }
    if (tid == 0) {
        if (c + 1 == nR) {
            counts[0] = run0;
            counts[1] = run1;
            if (nR <= 10)
                counts[2] = 0;
        }
        if (c == 10)
            counts[2] = n;
    }
}

/* ------------------------------------------------------------------------- */
/* road_marker: marker points                                                  */
/* ------------------------------------------------------------------------- */
/* lidar_segmentation.cpp:305-351 scans, for every integer degree, all rings in order and every ring
 * in ascending azimuth, remembers the farthest road point of that degree and stops at the first point
 * of that degree that is not road.  Per ring and degree that is: the smallest azimuth of a non-road
 * point (where the scan of this ring stops, and with it the whole scan), and the farthest road point
 * in front of it (ties: the first in azimuth order).  k_marker_ring builds these two tables per ring
 * in LDS (no sort needed), k_marker_bins walks the rings per degree. */
__global__ __launch_bounds__(256) void k_marker_ring(urf_kargs a, urf_dev_params dp, unsigned s0,
                                                     float* m_d_all, unsigned* m_pos_all, uint8_t* m_red_all, uint8_t* m_lit_all)
{
    __shared__ int nrmin[URF_DEG_CELLS];
    __shared__ unsigned long long best[URF_DEG_CELLS];
    __shared__ unsigned bestpos[URF_DEG_CELLS];
    __shared__ unsigned need_lit;   /* the ring's order decides (k_marker_ring_literal) */
    const unsigned c = blockIdx.x, s = s0 + blockIdx.y, tid = threadIdx.x;
    const size_t cells = (size_t)URF_MAX_CHANNELS * URF_DEG_CELLS;   /* per scan of the launch */
    float* m_d = m_d_all + blockIdx.y * cells;
    unsigned* m_pos = m_pos_all + blockIdx.y * cells;
    uint8_t* m_red = m_red_all + blockIdx.y * cells;
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK || c >= in.n_rings) {
        if (tid == 0)
            m_lit_all[(size_t)blockIdx.y * URF_MAX_CHANNELS + c] = 0;
        return;
    }
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels;
    const unsigned n = a.ring_cnt[(size_t)s * C + c];
    const unsigned sb = urf_sbase(a, s);
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    /* input index of the point in ring-sorted slot `slot` (relative to the scan) */
    auto src_of = [&](unsigned slot) { return (slot & ~(URF_TILE - 1u)) + (a.rec[sb + slot] & URF_REC_SRC_MASK); };
    for (unsigned i = tid; i < URF_DEG_CELLS; i += 256) {
        nrmin[i] = URF_INT_NONE_MIN;
        best[i] = 0;
        bestpos[i] = 0xffffffffu;
    }
    if (tid == 0)
        need_lit = 0;
    constexpr unsigned EPT = 8;
    if (n <= 256 * EPT) {
        /* The usual ring (at most 2048 points): every point is looked at ONCE -- its slot through the
         * ring's run table in LDS (k_ring's map) instead of a bisection in global memory, slot records,
         * then labels, eight points per thread in flight at a time -- and azimuth, label and slot stay
         * in registers for the three passes.  (Pass by pass, with seven dependent loads per point and
         * the exact azimuth worked out three times, this kernel took longer than the whole
         * classification: 3.9 ms per 1024 sweeps.) */
        extern __shared__ unsigned sh_mark_tab[];   /* P[tiles + 1], radd[tiles] (urf_ring_map) */
        unsigned* const mapP = sh_mark_tab;
        unsigned* const mapA = sh_mark_tab + a.tiles + 1;
        {
            const unsigned* gp = a.rpre + ((size_t)s * C + c) * (a.tiles + 1);
            const uint16_t* gs = a.rstart + ((size_t)s * C + c) * a.tiles;
            for (unsigned t = tid; t <= ntiles; t += 256) {
                const unsigned pt = gp[t];
                mapP[t] = pt;
                if (t < ntiles)
                    mapA[t] = t * URF_TILE + gs[t] - pt;   /* relative to the scan's scratch base */
            }
        }
        __syncthreads();
        const urf_ring_map map = { mapP, mapA, ntiles, (float)ntiles / (float)(n > 0 ? n : 1) };
        unsigned slot[EPT], sr[EPT];
        float az[EPT], px[EPT], py[EPT];
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const unsigned p = tid + e * 256;
            slot[e] = p < n ? map.at(p) : 0u;
            sr[e] = a.rec[sb + slot[e]] & URF_REC_SRC_MASK;
            px[e] = a.rx[sb + slot[e]];
            py[e] = a.ry[sb + slot[e]];
        }
        unsigned labs = 0;   /* two bits per point */
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const unsigned lab = a.labels[off + (slot[e] & ~(URF_TILE - 1u)) + sr[e]] & URF_LABEL_MASK;
            labs |= lab << (2 * e);
        }
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {   /* (the slot's record holds an approximation of the azimuth) */
            float d2;
            az[e] = urf_azimuth(px[e], py[e], &d2);
        }
        /* pass 1: where does the scan of this ring stop in each degree (:318) */
        int bin[EPT];
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            const int b = (int)__builtin_floorf(az[e]);
            bin[e] = b < 0 ? 0 : (b > 360 ? 360 : b);
            if (tid + e * 256 < n && az[e] == az[e] && ((labs >> (2 * e)) & 3u) != URF_LABEL_ROAD)
                atomicMin(&nrmin[bin[e]], (int)urf_fbits(az[e]));
        }
        __syncthreads();
        /* pass 2: farthest road point in front of it (:325-335); key = (d, first in azimuth order) */
        unsigned long long key[EPT];
#pragma unroll
        for (unsigned e = 0; e < EPT; e++) {
            key[e] = 0;
            const bool road = tid + e * 256 < n && az[e] == az[e] && ((labs >> (2 * e)) & 3u) == URF_LABEL_ROAD;
            if (road && (int)urf_fbits(az[e]) == nrmin[bin[e]])
                need_lit = 1u;   /* the very azimuth of the degree's first non-road point: in front of it or behind? */
            if (road && (int)urf_fbits(az[e]) < nrmin[bin[e]]) {
                const float x = px[e], y = py[e];
                const float d = (float)__builtin_sqrt((double)(0.f - x) * (double)(0.f - x) + (double)(0.f - y) * (double)(0.f - y));
                if (d > 0.0f) {   /* "d > maxDistanceRoad" with maxDistanceRoad starting at 0 */
                    key[e] = ((unsigned long long)urf_fbits(d) << 32) | (0xffffffffu - urf_fbits(az[e]));
                    atomicMax(&best[bin[e]], key[e]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (unsigned e = 0; e < EPT; e++)
            if (key[e] != 0 && key[e] == best[bin[e]])
                if (atomicMin(&bestpos[bin[e]], tid + e * 256) != 0xffffffffu)
                    need_lit = 1u;   /* two road points with this distance AND azimuth: which comes first? */
        __syncthreads();
        if (tid == 0)
            m_lit_all[(size_t)blockIdx.y * URF_MAX_CHANNELS + c] = (uint8_t)need_lit;
        for (unsigned i = tid; i < URF_DEG_CELLS; i += 256) {
            const size_t o = (size_t)c * URF_DEG_CELLS + i;
            m_d[o] = __uint_as_float((unsigned)(best[i] >> 32));
            m_pos[o] = bestpos[i] == 0xffffffffu ? 0xffffffffu : sb + map.at(bestpos[i]);
            m_red[o] = nrmin[i] != URF_INT_NONE_MIN;
        }
        return;
    }
    __syncthreads();
    /* pass 1: where does the scan of this ring stop in each degree (:318) */
    for (unsigned p = tid; p < n; p += 256) {
        const unsigned slot = urf_ring_slot(a, s, C, c, ntiles, p);
        const float az = urf_exact_az(a, sb + slot);
        const unsigned lab = a.labels[off + src_of(slot)] & URF_LABEL_MASK;
        if (az == az && lab != URF_LABEL_ROAD) {
            int bin = (int)__builtin_floorf(az);
            bin = bin < 0 ? 0 : (bin > 360 ? 360 : bin);
            atomicMin(&nrmin[bin], (int)urf_fbits(az));
        }
    }
    __syncthreads();
    /* pass 2: farthest road point in front of it (:325-335); key = (d, first in azimuth order) */
    for (int pass = 0; pass < 2; pass++) {
        for (unsigned p = tid; p < n; p += 256) {
            const unsigned slot = urf_ring_slot(a, s, C, c, ntiles, p);
            const float az = urf_exact_az(a, sb + slot);
            const unsigned lab = a.labels[off + src_of(slot)] & URF_LABEL_MASK;
            if (az == az && lab == URF_LABEL_ROAD) {
                int bin = (int)__builtin_floorf(az);
                bin = bin < 0 ? 0 : (bin > 360 ? 360 : bin);
                if ((int)urf_fbits(az) == nrmin[bin])
                    need_lit = 1u;
                if ((int)urf_fbits(az) < nrmin[bin]) {
                    const float x = a.rx[sb + slot], y = a.ry[sb + slot];
                    const float d = (float)__builtin_sqrt((double)(0.f - x) * (double)(0.f - x) + (double)(0.f - y) * (double)(0.f - y));
                    if (d > 0.0f) {   /* "d > maxDistanceRoad" with maxDistanceRoad starting at 0 */
                        const unsigned long long key = ((unsigned long long)urf_fbits(d) << 32) | (0xffffffffu - urf_fbits(az));
                        if (pass == 0)
                            atomicMax(&best[bin], key);
                        else if (key == best[bin] && atomicMin(&bestpos[bin], p) != 0xffffffffu)
                            need_lit = 1u;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0)
        m_lit_all[(size_t)blockIdx.y * URF_MAX_CHANNELS + c] = (uint8_t)need_lit;
    for (unsigned i = tid; i < URF_DEG_CELLS; i += 256) {
        const size_t o = (size_t)c * URF_DEG_CELLS + i;
        m_d[o] = __uint_as_float((unsigned)(best[i] >> 32));
        m_pos[o] = bestpos[i] == 0xffffffffu ? 0xffffffffu : sb + urf_ring_slot(a, s, C, c, ntiles, bestpos[i]);
        m_red[o] = nrmin[i] != URF_INT_NONE_MIN;
    }
}

/* The same tables when the ring's ORDER decides -- a road point shares its azimuth, bit for bit, with the first non-road
 * point of its degree, or two road points share azimuth and distance: which comes first is what the reference's Lomuto
 * quicksort (lidar_segmentation.cpp:70-93) leaves.  k_marker_ring flags such a ring (m_lit); this kernel, launched behind
 * it on the same grid, returns at once for every other ring.  The ring is sorted literally (urf_lomuto_sort on (azimuth,
 * position) pairs in the ring's stretch of wsg, as k_nan_rings / k_ring_order do) and the three passes compare places in
 * that order instead of azimuths.  Rare: never on a spinning sensor's sweep. */
__global__ __launch_bounds__(256) void k_marker_ring_literal(urf_kargs a, urf_dev_params dp, unsigned s0, const uint8_t* m_lit_all,
                                                             float* m_d_all, unsigned* m_pos_all, uint8_t* m_red_all)
{
    const unsigned c = blockIdx.x, s = s0 + blockIdx.y, tid = threadIdx.x;
    if (!m_lit_all[(size_t)blockIdx.y * URF_MAX_CHANNELS + c])
        return;
    __shared__ int nrmin[URF_DEG_CELLS];
    __shared__ unsigned long long best[URF_DEG_CELLS];
    __shared__ unsigned bestpos[URF_DEG_CELLS];
    __shared__ int stk[2 * 64];
    const size_t cells = (size_t)URF_MAX_CHANNELS * URF_DEG_CELLS;
    float* m_d = m_d_all + blockIdx.y * cells;
    unsigned* m_pos = m_pos_all + blockIdx.y * cells;
    uint8_t* m_red = m_red_all + blockIdx.y * cells;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels;
    const unsigned n = a.ring_cnt[(size_t)s * C + c];
    const unsigned sb = urf_sbase(a, s);
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    const unsigned rel = a.ring_off[(size_t)s * (C + 1) + c];
    volatile unsigned long long* const LIT = (volatile unsigned long long*)(a.wsg + sb + rel);
    for (unsigned i = tid; i < URF_DEG_CELLS; i += 256) {
        nrmin[i] = URF_INT_NONE_MIN;
        best[i] = 0;
        bestpos[i] = 0xffffffffu;
    }
    for (unsigned p = tid; p < n; p += 256)
        LIT[p] = ((unsigned long long)urf_fbits(urf_exact_az(a, sb + urf_ring_slot(a, s, C, c, ntiles, p))) << 32) | p;
    __threadfence_block();
    __syncthreads();
    if (tid < 64 && n >= 2)
        urf_lomuto_sort(LIT, n, stk);
    __threadfence_block();
    __syncthreads();
    for (int pass = 0; pass < 3; pass++) {
        for (unsigned j = tid; j < n; j += 256) {
            const unsigned long long e = LIT[j];
            const float az = urf_pair_alpha(e);
            const unsigned p = (unsigned)e, slot = urf_ring_slot(a, s, C, c, ntiles, p);
            const unsigned lab = a.labels[off + (slot & ~(URF_TILE - 1u)) + (a.rec[sb + slot] & URF_REC_SRC_MASK)] & URF_LABEL_MASK;
            if (!(az == az))
                continue;
            int bin = (int)__builtin_floorf(az);
            bin = bin < 0 ? 0 : (bin > 360 ? 360 : bin);
            if (pass == 0) {
                if (lab != URF_LABEL_ROAD)
                    atomicMin(&nrmin[bin], (int)j);   /* :318 the scan of this ring stops here */
            } else if (lab == URF_LABEL_ROAD && (int)j < nrmin[bin]) {
                const float x = a.rx[sb + slot], y = a.ry[sb + slot];
                const float d = (float)__builtin_sqrt((double)(0.f - x) * (double)(0.f - x) + (double)(0.f - y) * (double)(0.f - y));
                if (d > 0.0f) {
                    const unsigned long long key = ((unsigned long long)urf_fbits(d) << 32) | (0xffffffffu - j);   /* (d, first in the ring's order) */
                    if (pass == 1)
                        atomicMax(&best[bin], key);
                    else if (key == best[bin])
                        bestpos[bin] = p;
                }
            }
        }
        __syncthreads();
    }
    for (unsigned i = tid; i < URF_DEG_CELLS; i += 256) {
        const size_t o = (size_t)c * URF_DEG_CELLS + i;
        m_d[o] = __uint_as_float((unsigned)(best[i] >> 32));
        m_pos[o] = bestpos[i] == 0xffffffffu ? 0xffffffffu : sb + urf_ring_slot(a, s, C, c, ntiles, bestpos[i]);
        m_red[o] = nrmin[i] != URF_INT_NONE_MIN;
    }
}

__global__ __launch_bounds__(384) void k_marker_bins(urf_kargs a, urf_dev_params dp, unsigned s0, const float* m_d_all,
                                                     const unsigned* m_pos_all, const uint8_t* m_red_all, float* out_all,
                                                     unsigned* count_all)
{
    __shared__ unsigned wsum[6];
    const unsigned s = s0 + blockIdx.x;
    const size_t cells = (size_t)URF_MAX_CHANNELS * URF_DEG_CELLS;
    const float* m_d = m_d_all + blockIdx.x * cells;
    const unsigned* m_pos = m_pos_all + blockIdx.x * cells;
    const uint8_t* m_red = m_red_all + blockIdx.x * cells;
    float* out = out_all + (size_t)blockIdx.x * URF_DEG_CELLS * 4;
    unsigned* count = count_all + blockIdx.x;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const urf_scan_info in = a.info[s];
    const unsigned nR = in.status == URF_OK ? in.n_rings : 0;
    unsigned id = 0xffffffffu;
    float red = 0.f;
    if (tid <= 360) {
        float maxd = 0.f;
        for (unsigned j = 0; j < nR; j++) {
            const size_t o = (size_t)j * URF_DEG_CELLS + tid;
            if (m_pos[o] != 0xffffffffu && m_d[o] > maxd) {   /* :329 */
                maxd = m_d[o];
                id = m_pos[o];
            }
            if (m_red[o]) {                                  /* :318-321, 338-339 */
                red = 1.f;
                break;
            }
        }
    }
    const bool valid = id != 0xffffffffu;                    /* :343 */
    const unsigned long long m = __ballot(valid);
    if (lane == 0)
        wsum[wave] = __popcll(m);
    __syncthreads();
    unsigned pre = __popcll(m & ((1ull << lane) - 1ull));
    for (unsigned w = 0; w < wave; w++)
        pre += wsum[w];
    if (valid) {
        out[4 * pre + 0] = a.rx[id];
        out[4 * pre + 1] = a.ry[id];
        out[4 * pre + 2] = a.rz[id];
        out[4 * pre + 3] = red;
    }
    if (tid == 0)
        *count = wsum[0] + wsum[1] + wsum[2] + wsum[3] + wsum[4] + wsum[5];
}

#ifdef URF_ENABLE_TEST_HOOKS   /* liburf_hip_test.so only (include/urf_test_hooks.h) */
/* ------------------------------------------------------------------------- */
/* self test                                                                   */
/* ------------------------------------------------------------------------- */
/* urf_div_pi(a) == a / M_PI for every float a in [0, 600] (bit patterns 0..0x44160000) */
__global__ __launch_bounds__(256) void k_selftest_div_pi(unsigned long long* mismatches)
{
    const unsigned top = 0x44160000u;   /* 600.0f */
    unsigned long long bad = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= top;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const double a = (double)__uint_as_float((unsigned)b);
        if (urf_div_pi(a) != a / URF_PI_D)
            bad++;
    }
    /* urf_sqrt_rn_normal(x) == sqrtf(x) for every float of [2^-90, 2^126] (k_front's planar range) */
    for (unsigned long long b = 0x12800000ull + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7e800000ull;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        if (__float_as_uint(urf_sqrt_rn_normal(x)) != __float_as_uint(__builtin_sqrtf(x)))
            bad++;
    }
    if (bad)
        atomicAdd(mismatches, bad);
}

/* max |fast - exact| of the float fast paths over pseudo-random points: out[0] = vertical angle
 * [deg] (float bits), out[1] = polar angle [rad], out[2] = scaled polar angle fi*Kfi, out[3] =
 * azimuth [deg] */
__global__ __launch_bounds__(256) void k_selftest_fast(unsigned long long n, float Kfi, unsigned* out)
{
    float ev = 0.f, ea = 0.f, eu = 0.f, ez = 0.f;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
        float c[3];
        for (int k = 0; k < 3; k++) {
            h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
            c[k] = ((float)(h >> 40) * (1.0f / 16777216.0f) - 0.5f) * ((i & 3) == 0 ? 400.0f : 20.0f);
        }
        /* an eighth of the samples at arbitrary magnitudes (2^-70 .. 2^70), with independent
         * exponents per coordinate: the range guards of the fast paths have to hold there too */
        if ((i & 7) == 1) {
            for (int k = 0; k < 3; k++) {
                h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27;
                c[k] = __builtin_ldexpf(c[k], (int)(h % 141u) - 70);
            }
        }
        /* another eighth close to the x axis (|y| / |x| between 1 / 2048 and 1 / 8), where the azimuth's
         * margin grows with 1 / delta and its end (urf_fast_az_ok) lies */
        if ((i & 7) == 2) {
            h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27;
            c[1] = c[0] * __builtin_ldexpf(1.0f + (float)(h >> 41) * (1.0f / 8388608.0f), -(int)(4 + (h & 7u))) * ((h & 8u) ? -1.0f : 1.0f);
        }
        const float x = c[0], y = c[1], z = c[2] * 0.25f;
        float vt;
        if (urf_fast_vertical_angle(x, y, z, &vt)) {   /* k_ring_table's look-ahead */
            const float d = __builtin_fabsf(vt - urf_vertical_angle(x, y, z));
            ev = d > ev ? d : ev;
        }
        float uc;
        bool planar_ok;
        if (urf_fast_cot(x, y, z, &uc, &planar_ok)) {   /* k_split: the angle whose cotangent uc is (atan2 rounded to float: +-1e-5 deg) */
            const float au = (float)((double)urf_atan2f(1.0f, uc) * (180.0 / URF_PI_D));
            const float d = __builtin_fabsf(au - urf_vertical_angle(x, y, z));
            ev = d > ev ? d : ev;
        }
        float azt;
        if (urf_fast_azimuth(x, y, &azt)) {
            float d2;
            const float d = __builtin_fabsf(azt - urf_azimuth(x, y, &d2)) / urf_fast_az_eps(azt);   /* as a fraction of the margin */
            if (d < 1000.0f)   /* the 0/360 seam is never decided on the approximation */
                ez = d > ez ? d : ez;
        }
        if (x != 0.f || y != 0.f) {
            float fe = urf_atan2f(y, x);
            const float fa = urf_fast_atan2f(y, x);
            const float da = __builtin_fabsf(fa - fe);
            ea = da > ea ? da : ea;
            if (fe < 0.0f)
                fe = (float)((double)fe + 2.0 * URF_PI_D);
            float ff = fa < 0.0f ? fa + 6.28318530717958648f : fa;
            /* near the wrap the two may sit on opposite ends: the fast path never decides there */
            const float du = __builtin_fabsf(ff * Kfi - fe * Kfi);
            if (du < 180.0f)
                eu = du > eu ? du : eu;
        }
    }
    atomicMax(&out[0], __float_as_uint(ev));
    atomicMax(&out[1], __float_as_uint(ea));
    atomicMax(&out[2], __float_as_uint(eu));
    atomicMax(&out[3], __float_as_uint(ez));
}
#endif   /* URF_ENABLE_TEST_HOOKS */

#endif /* URF_KERNELS_HPP */
