/*
 * urf_k_beams_label.hpp -- k_beams: the blind-spot beam march (blind_spots.cpp:7-284); k_label / k_label_list: road / curb / road_probably (lidar_segmentation.cpp:354-367, 605-608).
 * One of the kernel families of urf_kernels.hpp (r6: split by family, zero behaviour change); included from there, in order.
 */
#ifndef URF_K_BEAMS_LABEL_HPP
#define URF_K_BEAMS_LABEL_HPP

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
    const bool organised = __syncthreads_or(((C & (C - 1u)) != 0u) || (tid <= C && v_koff != (tid << logP))) == 0;
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


#endif /* URF_K_BEAMS_LABEL_HPP */
