/*
 * urf_k_table.hpp -- PointCloud2 records -> SoA; k_ring_table / k_table_repair: the first-fit ring-angle table (lidar_segmentation.cpp:145-205).
 * One of the kernel families of urf_kernels.hpp (r6: split by family, zero behaviour change); included from there, in order.
 */
#ifndef URF_K_TABLE_HPP
#define URF_K_TABLE_HPP

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
    unsigned nL, nmatch, zero, fresh, first_nl;
    unsigned mins[URF_TABLE_THREADS / 64];
};
/* k_ring_table's third rule (below), the search: the first region-of-interest point of every row of a row-major organised sweep ->
 * their vertical angles, and through the reference's insertion in row order: rows_v[s][0 .. n) the leaders, rows_ok[s] = n + 1.  Given up
 * (rows_ok[s] = 0) when one of them would not become a leader, or as soon as the 64 points from a row's first one on show a second
 * ring: a sweep in firing order.  A kernel of its own, in the sequence of a context that has
 * sighted such a sweep: inside k_ring_table its registers cost that kernel a wave per SIMD (1024 scans no longer resident at once). */
__global__ __launch_bounds__(256) void k_rows_probe(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned sh_alive;
    __shared__ float rowv[64];
    const unsigned s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    if (len < 128u || (len & 63u) != 0u) {
        if (tid == 0)
            a.rows_ok[s] = 0u;
        return;
    }
    const unsigned F = len >> 6;
    volatile unsigned* const alive = &sh_alive;
    if (tid == 0)
        sh_alive = 1u;
    __syncthreads();
    /* wave w takes rows 16 w .. 16 w + 15, four at a time */
    const float tol = 2.0f * dp.p.interval * 0.017453292f;
    for (unsigned g = 0; g < 16u && *alive; g += 4u) {
        const unsigned r0 = wave * 16u + g;
        unsigned found = 0;   /* bit q: row r0 + q is settled */
        for (unsigned c0 = 0; c0 < F && found != 15u && *alive; c0 += 64u) {
            float px[4], py[4], pz[4];
            const bool in = c0 + lane < F;
#pragma unroll
            for (unsigned q = 0; q < 4; q++) {
                const size_t i = (size_t)off + (size_t)(r0 + q) * F + c0 + (in ? lane : 0u);
                px[q] = a.x[i];
                py[q] = a.y[i];
                pz[q] = a.z[i];
            }
#pragma unroll
            for (unsigned q = 0; q < 4; q++) {
                if ((found >> q) & 1u)
                    continue;   /* (uniform) */
                const bool roi = in && urf_in_roi(dp.p, px[q], py[q], pz[q]);
                const unsigned long long m = __ballot(roi);
                if (m == 0ull)
                    continue;
                const int src = (int)__ffsll((long long)m) - 1;
                /* one ring?  (cot of the vertical angle, to twice the interval: a hint -- k_front is the check) */
                const float u = -pz[q] * __builtin_amdgcn_rsqf(px[q] * px[q] + py[q] * py[q]), u0 = __shfl(u, src);
                if (__ballot(roi && !(__builtin_fabsf(u - u0) <= tol * (1.0f + u0 * u0))) != 0ull) {
                    if (lane == 0)
                        *alive = 0u;
                    break;
                }
                const float v = urf_vertical_angle(__shfl(px[q], src), __shfl(py[q], src), __shfl(pz[q], src));
                if (lane == 0)
                    rowv[r0 + q] = v;
                found |= 1u << q;
            }
        }
#pragma unroll
        for (unsigned q = 0; q < 4; q++)
            if (!((found >> q) & 1u) && lane == 0)
                rowv[r0 + q] = -1.0f;
    }
    __syncthreads();
    /* (a) of the rule: put through the reference's insertion in row order, every one of them becomes a leader -- none matches an earlier one
     * (lidar_segmentation.cpp:176-190: |angle[j] - alpha| <= interval), none is the table's end marker 0 (:176) */
    if (wave == 0) {
        const float v = rowv[lane];
        bool bad = v == 0.0f;
        for (unsigned j = 0; j < 63u; j++) {
            const float w = rowv[j];
            bad = bad || (j < lane && v >= 0.0f && w >= 0.0f && __builtin_fabsf(w - v) <= dp.p.interval);
        }
        const unsigned long long have = __ballot(v >= 0.0f);
        const bool ok = sh_alive != 0u && __ballot(bad) == 0ull;
        if (ok && v >= 0.0f)
            a.rows_v[(size_t)s * 64u + urf_popc_below(have)] = v;   /* compacted: the reference's angle[] before its sort */
        if (lane == 0)
            a.rows_ok[s] = ok ? (unsigned)__popcll(have) + 1u : 0u;
    }
}

__device__ void urf_ring_table_scan(const urf_kargs& a, const urf_dev_params& dp, unsigned s, unsigned lookahead, urf_table_shared& T, bool first_walk)
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
        T.first_nl = 0;
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
    /* wave 0: the lanes' angles (-1: none), in lane order, through the reference's insertion */
    auto insert64 = [&](const float v) {
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
            /* The same for lasers in ANY order inside the firing (r6: a Velodyne's laser numbering interleaves two blocks; the loop below
             * cost such a batch 0.11 ms): into an empty table all unmatched points become leaders, in lane order, iff no two of them
             * match each other and none is the end marker.  64 broadcast reads; the sorted copy by rank. */
            if ((m = __ballot(un)) != 0 && !zero_seen && nmatch == 0 && nL + (unsigned)__popcll(m) <= C) {
                float* const stage = SL + 64;   /* (the sorted copy is empty: its upper half carries the step's angles) */
                stage[lane] = un ? v : -1.0f;
                urf_wave_lds_sync();
                bool clash = un && v == 0.0f;
                unsigned srank = 0;
                for (unsigned j = 0; j < 64; j++) {
                    const float w = stage[j];
                    const bool there = w >= 0.0f;
                    clash = clash || (there && un && j < lane && __builtin_fabsf(w - v) <= interval);
                    srank += (there && (w < v || (w == v && j < lane))) ? 1u : 0u;
                }
                if (__ballot(clash) == 0) {
                    const unsigned cnt = (unsigned)__popcll(m);
                    if (un) {
                        L[nL + urf_popc_below(m)] = v;
                        SL[srank] = v;
                    }
                    nL += cnt;
                    nmatch += cnt;
                    un = false;
                }
                urf_wave_lds_sync();
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
                if (nL0 == 0u && nL != 0u)
                    T.first_nl = zero_seen ? 0xffffu : nL;   /* what the first step that met a ring found (the rows' sighting, below) */
            }
    };
    /* Third speculation (r6): a ROW-MAJOR organised sweep -- height = the sensor's lasers, width = firings, point l * F + f: what
     * the drivers that publish organised clouds deliver.  The walk below meets a new ring every F points (0.7 ms per 1024 sweeps of
     * 64 x 2048).  Rule: the leaders are the first region-of-interest points of the 64 rows, in row order.  That IS the reference's
     * table when (a) those points, inserted in row order, all become leaders (checked here, by the reference's own insertion) and
     * (b) every other region-of-interest point of a row lies on its row's entry (then it matched that leader when the reference
     * met it: it is behind it in its row): k_front checks (b) point by point for the scan's lanes anyway -- a point on another
     * entry or on none raises table_redo[s], k_table_repair walks the scan the long way and the legacy kernels take it.  Only
     * for a scan the fused front end will look at; a call whose sequence lacks k_transpose only reports the sighting.
     * Tried when the walk's first step (64 points: one row's) has shown at most one ring; given up as soon as the 64 points around
     * a row's first one show a second ring (a sweep in firing order whose first firing lies outside the region of interest). */
    bool rows = false;
    bool rows_try = first_walk && (a.front || a.front_sight) && C == 64u && len >= 128u && (len & 63u) == 0u;   /* (not k_table_repair's walk: that one follows a failure) */
    if (rows_try && a.front_rows) {
        /* k_rows_probe has found the rows' first points and put them through the reference's insertion: rows_ok[s] - 1 leaders, in row order */
        const unsigned nr = a.rows_ok[s];   /* (uniform) */
        if (nr) {
            if (tid < nr - 1u)
                L[tid] = a.rows_v[(size_t)s * 64u + tid];
            if (tid == 0) {
                sh_nL = nr - 1u;
                a.front_ok[s] = URF_FRONT_ROWS;
                a.front_state[4] = 1u;   /* host-visible: this context's sweeps DO come row-major (urf_api.hip: the fused kernels at any batch size) */
            }
            rows = true;
            upto = 0;
            cause = 3;
            __syncthreads();
        }
    }
    while (!rows && pos < len && sh_nL < C) {
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
            insert64(v);
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
    if (rows_try && !a.front_rows && !rows) {
        /* The first 64-point step that met a ring: a sweep in firing order shows many at once there (wherever its region of interest
         * begins), a row-major one as many as rows of len / 64 points fit into the step.  A sighting: */
        const unsigned nl = T.first_nl;
        if (nl != 0u && nl * (len >> 6) <= 63u + (len >> 6)) {   /* nl <= ceil(64 / F) */
            if (tid == 0)
                a.front_state[2] = 1u;   /* host-visible: the next call's sequence holds k_rows_probe and k_transpose */
            cause |= 4u;   /* (a row-major sweep defeats the look-ahead wherever its region of interest drops a few rows in succession:
                            * such a failure must not switch ALL speculation off, this rule included -- k_table_repair) */
        }
    }
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
    urf_ring_table_scan(a, dp, blockIdx.x, a.table_lookahead, T, true);
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
    urf_ring_table_scan(a, dp, s, 0, T, false);
    if (threadIdx.x == 0) {
        if (!collect)   /* (a collected scan is split by k_split_list) */
            a.redo_list[atomicAdd(&a.star_count[2], 1u)] = s;
        if (cause == 3u)
            a.front_state[3] = 1u;   /* (the rows' rule: counted, not switched off -- a failure costs that scan the long walk) */
        else if (!(cause & 4u))     /* (4: a scan that looked row-major to a call whose sequence lacked the kernels for it) */
            a.spec_failed[cause == 2u ? 1 : 0] = 1u;   /* host-visible: the context stops using the rule that failed */
    }
}


#endif /* URF_K_TABLE_HPP */
