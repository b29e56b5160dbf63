/*
 * urf_k_ring.hpp -- k_ring / k_ring_general / k_ring_list: x_zero and z_zero along a ring (x_zero_method.cpp, z_zero_method.cpp); k_nan_rings.
 * One of the kernel families of urf_kernels.hpp (r6: split by family, zero behaviour change); included from there, in order.
 */
#ifndef URF_K_RING_HPP
#define URF_K_RING_HPP

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
    return br >= x_angle_thr;   /* :58-61 "alpha <= angleFilter1", alpha = acos(br) in degrees: urf_api.hip urf_angle_threshold */
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
    return br >= z_angle_thr;   /* :63-66, as in urf_x_zero_angle */
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
            if (q0 < n) {
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
            if (S.n_cand > urf_ring_shared_t<QUADS>::CAND - CH || cs + CH >= n) {   /* the next chunk might not fit / last chunk */
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


#endif /* URF_K_RING_HPP */
