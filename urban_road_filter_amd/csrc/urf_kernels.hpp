/*
 * urf_kernels.hpp -- the gfx950 kernels of the per-scan classification.
 *
 * One launch covers a whole batch: blockIdx.y (or .x for per-scan kernels) is
 * the scan, the other grid dimension the tile / ring / sector inside it.
 * Pipeline (reference lines each kernel replaces; DESIGN.md has the full map):
 *
 *   k_ingest       ROI test, vertical angle, star sector + per-tile sector
 *                  histogram            lidar_segmentation.cpp:106-166, star_shaped_search.cpp:162-174
 *   k_ring_table   first-fit ring-angle table + sort            lidar_segmentation.cpp:124-126,168-196,205
 *   k_ring_assign  ring of every point + per-tile ring histogram lidar_segmentation.cpp:226-233
 *   k_offsets      exclusive scans of both histograms
 *   k_scatter      stable split into ring-major and sector-major order   lidar_segmentation.cpp:238-242,276
 *   k_star         per-sector sort by range + slope scan        star_shaped_search.cpp:109-150
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
#define URF_RING_THREADS 256
#define URF_LABEL_THREADS 384
#define URF_STAR_THREADS 64

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
/* k_ingest                                                                    */
/* ------------------------------------------------------------------------- */
__global__ __launch_bounds__(URF_TILE_THREADS) void k_ingest(urf_kargs a, urf_dev_params dp)
{
    extern __shared__ unsigned sh_hist[];   /* [sectors] sector histogram, [sectors] = ROI count */
    const unsigned s = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned tbase = t * URF_TILE;
    if (tbase >= len)
        return;
    const unsigned K = (unsigned)dp.p.sectors;
    const bool star = dp.p.star_shaped_method != 0;
    for (unsigned k = tid; k <= K; k += URF_TILE_THREADS)
        sh_hist[k] = 0;
    __syncthreads();

    for (unsigned q = 0; q < URF_TILE / URF_TILE_THREADS; q++) {
        const unsigned i = tbase + q * URF_TILE_THREADS + tid;
        const bool valid = i < len;
        float x = 0.f, y = 0.f, z = 0.f;
        if (valid) {
            x = a.x[off + i];
            y = a.y[off + i];
            z = a.z[off + i];
        }
        const bool roi = valid && urf_in_roi(dp.p, x, y, z);
        float va = -1.0f;
        unsigned key = URF_SEC_NONE;
        if (roi) {
            va = urf_vertical_angle(x, y, z);
            if (star) {
                key = urf_sector(x, y, dp.Kfi, K);
                if (dp.p.starbeam_filter && !urf_in_beam(a.beams[key], x, y))
                    key = URF_SEC_NONE;
            }
        }
        if (valid) {
            a.valpha[off + i] = va;
            a.seckey[off + i] = (uint16_t)key;
        }
        if (star) {
            const unsigned long long m = urf_match_any(key, dp.sec_keybits);
            if (key != URF_SEC_NONE && urf_is_leader(m))
                atomicAdd(&sh_hist[key], (unsigned)__popcll(m));
        }
        const unsigned long long rb = __ballot(roi);
        if (urf_lane() == 0 && rb)
            atomicAdd(&sh_hist[K], (unsigned)__popcll(rb));
    }
    __syncthreads();
    const size_t row = (size_t)s * a.tiles + t;
    if (star)
        for (unsigned k = tid; k < K; k += URF_TILE_THREADS)
            a.tile_sec[row * K + k] = sh_hist[k];
    if (tid == 0)
        a.tile_roi[row] = sh_hist[K];
}

/* ------------------------------------------------------------------------- */
/* k_ring_table                                                                */
/* ------------------------------------------------------------------------- */
__device__ __forceinline__ unsigned urf_block_min_256(unsigned v, unsigned* sh4)
{
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    __syncthreads();
    if (urf_lane() == 0)
        sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned r = sh4[0];
    for (int w = 1; w < 4; w++)
        r = sh4[w] < r ? sh4[w] : r;
    return r;
}

/* The reference walks the ROI points in order and appends a point's vertical
 * angle to the table when no earlier entry lies within `interval`
 * (lidar_segmentation.cpp:168-196).  Equivalent formulation used here: leader
 * k+1 is the first point after leader k that matches none of the leaders
 * 0..k; the search for it is a parallel min-reduction over 1024 points per
 * step, restarted behind every new leader.  The `angle[j] == 0` end-of-table
 * sentinel (:176) is honoured: once a leader equal to 0 has been stored, only
 * the leaders in front of it take part in matching. */
__global__ __launch_bounds__(256) void k_ring_table(urf_kargs a, urf_dev_params dp)
{
    __shared__ float L[URF_MAX_CHANNELS];
    __shared__ unsigned sh4[4];
    __shared__ unsigned sh_n, sh_nmatch, sh_zero;
    const unsigned s = blockIdx.x, tid = threadIdx.x;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels;
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;

    /* piece = number of ROI points (lidar_segmentation.cpp:120) */
    unsigned cnt = 0;
    for (unsigned t = tid; t < ntiles; t += 256)
        cnt += a.tile_roi[(size_t)s * a.tiles + t];
    for (int o = 32; o > 0; o >>= 1)
        cnt += __shfl_xor(cnt, o);
    if (urf_lane() == 0)
        sh4[tid >> 6] = cnt;
    if (tid == 0) {
        sh_n = 0;
        sh_nmatch = 0;
        sh_zero = 0;
    }
    __syncthreads();
    const unsigned piece = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    const bool too_few = piece < 30;   /* lidar_segmentation.cpp:124 */
    if (tid == 0) {
        urf_scan_info in;
        in.status = too_few ? URF_TOO_FEW_POINTS : URF_OK;
        in.n_roi = piece;
        in.n_rings = 0;
        in.n_ring_pts = 0;
        in.n_road = 0;
        in.n_curb = 0;
        in.n_ring10 = 0;
        in.reserved = 0;
        a.info[s] = in;
    }
    if (too_few)
        return;

    const float interval = dp.p.interval;
    unsigned base = 0;
    while (base < len) {
        const unsigned nmatch = sh_nmatch;
        unsigned first = 0xffffffffu;
        for (unsigned q = 0; q < 4 && first == 0xffffffffu; q++) {
            const unsigned i = base + q * 256 + tid;
            if (i < len) {
                const float v = a.valpha[off + i];
                if (v >= 0.0f) {   /* ROI point */
                    bool matched = false;
                    for (unsigned j = 0; j < nmatch; j++) {
                        if (__builtin_fabsf(L[j] - v) <= interval) {
                            matched = true;
                            break;
                        }
                    }
                    if (!matched)
                        first = i;
                }
            }
        }
        const unsigned m = urf_block_min_256(first, sh4);
        if (m == 0xffffffffu) {
            base += 1024;
        } else {
            if (tid == 0) {
                const float v = a.valpha[off + m];
                const unsigned n = sh_n;
                L[n] = v;
                sh_n = n + 1;
                if (!sh_zero) {
                    if (v == 0.0f)
                        sh_zero = 1;
                    else
                        sh_nmatch = n + 1;
                }
            }
            base = m + 1;
        }
        __syncthreads();
        if (sh_n >= C)
            break;
    }

    /* std::sort(angle, angle + index), lidar_segmentation.cpp:205 (rank sort) */
    const unsigned n = sh_n;
    if (tid < n) {
        const float v = L[tid];
        unsigned rank = 0;
        for (unsigned j = 0; j < n; j++) {
            const float w = L[j];
            rank += (w < v) || (w == v && j < tid);
        }
        a.angle[(size_t)s * C + rank] = v;
    }
    if (tid == 0)
        a.info[s].n_rings = n;
}

/* ------------------------------------------------------------------------- */
/* k_ring_assign                                                               */
/* ------------------------------------------------------------------------- */
/* lidar_segmentation.cpp:226-233: first sorted table entry within `interval`.
 * fl(angle[j] - alpha) is monotone in angle[j], so the matching entries are
 * contiguous and the first one is found by bisection with the very same float
 * predicate. */
__global__ __launch_bounds__(URF_TILE_THREADS) void k_ring_assign(urf_kargs a, urf_dev_params dp)
{
    __shared__ float tab[URF_MAX_CHANNELS];
    __shared__ unsigned hist[URF_MAX_CHANNELS];
    const unsigned s = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned tbase = t * URF_TILE;
    if (tbase >= len)
        return;
    const unsigned C = (unsigned)dp.p.channels;
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK) {
        /* nothing is published for this scan: all labels 0 */
        for (unsigned q = 0; q < URF_TILE / URF_TILE_THREADS; q++) {
            const unsigned i = tbase + q * URF_TILE_THREADS + tid;
            if (i < len)
                a.labels[off + i] = 0;
        }
        return;
    }
    const unsigned nR = in.n_rings;
    if (tid < C) {
        tab[tid] = tid < nR ? a.angle[(size_t)s * C + tid] : 0.f;
        hist[tid] = 0;
    }
    __syncthreads();
    const float interval = dp.p.interval;
    for (unsigned q = 0; q < URF_TILE / URF_TILE_THREADS; q++) {
        const unsigned i = tbase + q * URF_TILE_THREADS + tid;
        const bool valid = i < len;
        unsigned key = URF_RING_NONE;
        uint8_t lab = 0;
        if (valid) {
            const float v = a.valpha[off + i];
            if (v >= 0.0f) {
                lab = URF_FLAG_ROI;
                unsigned lo = 0, hi = nR;
                while (lo < hi) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (tab[mid] - v >= -interval)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                if (lo < nR && __builtin_fabsf(tab[lo] - v) <= interval)
                    key = lo;
            }
            a.ringkey[off + i] = (uint8_t)key;
            a.labels[off + i] = lab;
        }
        const unsigned long long m = urf_match_any(key, dp.ring_keybits);
        if (key != URF_RING_NONE && urf_is_leader(m))
            atomicAdd(&hist[key], (unsigned)__popcll(m));
    }
    __syncthreads();
    if (tid < C)
        a.tile_ring[((size_t)s * a.tiles + t) * C + tid] = hist[tid];
}

/* ------------------------------------------------------------------------- */
/* k_offsets                                                                   */
/* ------------------------------------------------------------------------- */
/* exclusive scan of cnt[0..K) (K <= 1024) by 256 threads -> off[0..K] */
__device__ void urf_scan_keys_256(const unsigned* cnt, unsigned* offs, unsigned K, unsigned* sh /* [256+8] */)
{
    const unsigned tid = threadIdx.x;
    unsigned v[4], sum = 0;
    for (int e = 0; e < 4; e++) {
        const unsigned k = tid * 4 + e;
        v[e] = k < K ? cnt[k] : 0;
        sum += v[e];
    }
    /* inclusive scan of `sum` over the block */
    unsigned inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned w = __shfl_up(inc, o);
        if ((int)urf_lane() >= o)
            inc += w;
    }
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

__global__ __launch_bounds__(256) void k_offsets(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned sh[8];
    const unsigned s = blockIdx.x, tid = threadIdx.x;
    if (a.info[s].status != URF_OK)
        return;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    /* rings */
    for (unsigned k = tid; k < C; k += 256) {
        unsigned run = 0;
        for (unsigned t = 0; t < ntiles; t++) {
            unsigned* p = &a.tile_ring[((size_t)s * a.tiles + t) * C + k];
            const unsigned c = *p;
            *p = run;
            run += c;
        }
        a.ring_cnt[(size_t)s * C + k] = run;
    }
    __syncthreads();
    urf_scan_keys_256(&a.ring_cnt[(size_t)s * C], &a.ring_off[(size_t)s * (C + 1)], C, sh);
    if (!dp.p.star_shaped_method)
        return;
    for (unsigned k = tid; k < K; k += 256) {
        unsigned run = 0;
        for (unsigned t = 0; t < ntiles; t++) {
            unsigned* p = &a.tile_sec[((size_t)s * a.tiles + t) * K + k];
            const unsigned c = *p;
            *p = run;
            run += c;
        }
        a.sec_cnt[(size_t)s * K + k] = run;
    }
    __syncthreads();
    urf_scan_keys_256(&a.sec_cnt[(size_t)s * K], &a.sec_off[(size_t)s * (K + 1)], K, sh);
}

/* ------------------------------------------------------------------------- */
/* k_scatter                                                                   */
/* ------------------------------------------------------------------------- */
/* Stable multi-split of one tile by ring and by sector.  A point's rank
 * inside its key = (points of that key in earlier tiles: tile_ring/tile_sec)
 * + (in earlier wave-sized groups of this tile: LDS matrix gcnt[group][key])
 * + (in lower lanes of its own group: match_any + popcount).  Input order is
 * preserved inside every ring, which x_zero / z_zero rely on
 * (lidar_segmentation.cpp:280-283 run before the azimuth sort :289). */
__global__ __launch_bounds__(URF_TILE_THREADS) void k_scatter(urf_kargs a, urf_dev_params dp)
{
    extern __shared__ unsigned sh_dyn[];
    const unsigned s = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned tbase = t * URF_TILE;
    if (tbase >= len)
        return;
    if (a.info[s].status != URF_OK)
        return;
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    const bool star = dp.p.star_shaped_method != 0;
    /* LDS carve: base_r[C] base_s[K] (uint32) | gcnt_r[G][C] gcnt_s[G][K] (uint16) */
    unsigned* base_r = sh_dyn;
    unsigned* base_s = base_r + C;
    uint16_t* gcnt_r = (uint16_t*)(base_s + (star ? K : 0));
    uint16_t* gcnt_s = gcnt_r + (size_t)URF_TILE_GROUPS * C;
    const size_t row = (size_t)s * a.tiles + t;

    for (unsigned k = tid; k < C; k += URF_TILE_THREADS)
        base_r[k] = off + a.ring_off[(size_t)s * (C + 1) + k] + a.tile_ring[row * C + k];
    if (star)
        for (unsigned k = tid; k < K; k += URF_TILE_THREADS)
            base_s[k] = off + a.sec_off[(size_t)s * (K + 1) + k] + a.tile_sec[row * K + k];
    {
        const unsigned tot = URF_TILE_GROUPS * (C + (star ? K : 0));
        unsigned* z32 = (unsigned*)gcnt_r;
        for (unsigned k = tid; k < (tot + 1) / 2; k += URF_TILE_THREADS)
            z32[k] = 0;
    }
    __syncthreads();

    unsigned rkey[URF_TILE / URF_TILE_THREADS], skey[URF_TILE / URF_TILE_THREADS];
    unsigned rrank[URF_TILE / URF_TILE_THREADS], srank[URF_TILE / URF_TILE_THREADS];
    const unsigned wave = tid >> 6;
#pragma unroll
    for (unsigned q = 0; q < URF_TILE / URF_TILE_THREADS; q++) {
        const unsigned i = tbase + q * URF_TILE_THREADS + tid;
        const bool valid = i < len;
        const unsigned g = q * (URF_TILE_THREADS / 64) + wave;
        rkey[q] = valid ? (unsigned)a.ringkey[off + i] : URF_RING_NONE;
        const unsigned long long mr = urf_match_any(rkey[q], dp.ring_keybits);
        rrank[q] = urf_popc_below(mr);
        if (rkey[q] != URF_RING_NONE && urf_is_leader(mr))
            gcnt_r[g * C + rkey[q]] = (uint16_t)__popcll(mr);
        skey[q] = URF_SEC_NONE;
        srank[q] = 0;
        if (star) {
            skey[q] = valid ? (unsigned)a.seckey[off + i] : URF_SEC_NONE;
            const unsigned long long ms = urf_match_any(skey[q], dp.sec_keybits);
            srank[q] = urf_popc_below(ms);
            if (skey[q] != URF_SEC_NONE && urf_is_leader(ms))
                gcnt_s[g * K + skey[q]] = (uint16_t)__popcll(ms);
        }
    }
    __syncthreads();
    /* exclusive scan over the groups, one thread per key */
    for (unsigned k = tid; k < C; k += URF_TILE_THREADS) {
        unsigned run = 0;
        for (unsigned g = 0; g < URF_TILE_GROUPS; g++) {
            const unsigned c = gcnt_r[g * C + k];
            gcnt_r[g * C + k] = (uint16_t)run;
            run += c;
        }
    }
    if (star)
        for (unsigned k = tid; k < K; k += URF_TILE_THREADS) {
            unsigned run = 0;
            for (unsigned g = 0; g < URF_TILE_GROUPS; g++) {
                const unsigned c = gcnt_s[g * K + k];
                gcnt_s[g * K + k] = (uint16_t)run;
                run += c;
            }
        }
    __syncthreads();
#pragma unroll
    for (unsigned q = 0; q < URF_TILE / URF_TILE_THREADS; q++) {
        const unsigned i = tbase + q * URF_TILE_THREADS + tid;
        const unsigned g = q * (URF_TILE_THREADS / 64) + wave;
        if (rkey[q] == URF_RING_NONE && skey[q] == URF_SEC_NONE)
            continue;
        const float x = a.x[off + i], y = a.y[off + i], z = a.z[off + i];
        if (rkey[q] != URF_RING_NONE) {
            const unsigned dst = base_r[rkey[q]] + gcnt_r[g * C + rkey[q]] + rrank[q];
            a.rx[dst] = x;
            a.ry[dst] = y;
            a.rz[dst] = z;
            a.rsrc[dst] = i;
        }
        if (skey[q] != URF_SEC_NONE) {
            const unsigned dst = base_s[skey[q]] + gcnt_s[g * K + skey[q]] + srank[q];
            a.sr[dst] = __builtin_sqrtf(x * x + y * y);   /* star_shaped_search.cpp:164 */
            a.sz[dst] = z;
            a.ssrc[dst] = i;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* k_star                                                                      */
/* ------------------------------------------------------------------------- */
/* star_shaped_search.cpp:123-149: walk the sector outwards, return the position
 * (in sorted order) of the first point whose slope gives the curb away. */
template <class GetR, class GetZ>
__device__ __forceinline__ int urf_slope_scan(unsigned n, const urf_dev_params& dp, GetR get_r, GetZ get_z)
{
    const float kdev = dp.p.kdev_param, kdist = dp.p.kdist_param, slope_param = dp.slope_param;
    const int dmin = dp.p.dmin_param;
    float avg = 0.f, dev = 0.f, nan = 0.f;
    float bx = get_r(0), by = get_z(0);
    for (unsigned i = 1; i < n; i++) {
        const float ax = bx, ay = by;
        bx = get_r(i);
        by = get_z(i);
        const float slp = (by - ay) / (bx - ax);
        if (slp != slp) {
            nan += 1.0f;
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
        if (slp > slope_param ||
            ((int)i > dmin && (slp * slp - avg * avg) * kdev * ((bx - ax) * kdist) > dev))
            return (int)i;
    }
    return -1;
}

/* One wave per (sector, scan).  MODE 0: sector fits CAP entries of LDS.
 * MODE 1: any size, sorted in place in global memory (adversarial inputs).
 * Sort key = (range bits, input index): unique, so the order is total where
 * the reference's std::sort leaves ties unspecified (star_shaped_search.cpp:109).
 * The network is the "normalised" bitonic sorter: every comparator orders
 * (lower index, higher index) ascending, so slots >= n behave as +inf padding
 * and comparators touching them are skipped. */
template <int CAP_LO, int CAP_HI, bool IN_LDS>
__global__ __launch_bounds__(URF_STAR_THREADS) void k_star(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned long long keys[IN_LDS ? CAP_HI : 1];
    __shared__ float zs[IN_LDS ? CAP_HI : 1];
    const unsigned k = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    if (a.info[s].status != URF_OK)
        return;
    const unsigned K = (unsigned)dp.p.sectors;
    const unsigned n = a.sec_cnt[(size_t)s * K + k];
    if ((int)n <= CAP_LO || (IN_LDS && (int)n > CAP_HI))
        return;   /* another instantiation owns this sector */
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned base = off + a.sec_off[(size_t)s * (K + 1) + k];
    int hit = -1;
    if (n >= 2) {
        unsigned P = 1;
        while (P < n)
            P <<= 1;
        if (IN_LDS) {
            for (unsigned i = lane; i < n; i += URF_STAR_THREADS) {
                keys[i] = ((unsigned long long)urf_fbits(a.sr[base + i]) << 32) | a.ssrc[base + i];
                zs[i] = a.sz[base + i];
            }
            __syncthreads();
            for (unsigned kk = 2; kk <= P; kk <<= 1) {
                for (unsigned j = kk >> 1; j > 0; j >>= 1) {
                    const bool flip = (j == (kk >> 1));
                    for (unsigned tt = lane; tt < (P >> 1); tt += URF_STAR_THREADS) {
                        const unsigned lo = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
                        /* flip step: partner of lo inside its block of size kk is block_end - (lo - block_start) */
                        const unsigned hi2 = flip ? ((lo & ~(kk - 1)) + (kk - 1) - (lo & (kk - 1))) : lo + j;
                        if (hi2 < n) {
                            const unsigned long long ka = keys[lo], kb = keys[hi2];
                            if (ka > kb) {
                                keys[lo] = kb;
                                keys[hi2] = ka;
                                const float za = zs[lo];
                                zs[lo] = zs[hi2];
                                zs[hi2] = za;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            if (lane == 0) {
                const int pos = urf_slope_scan(
                    n, dp, [&](unsigned i) { return __uint_as_float((unsigned)(keys[i] >> 32)); },
                    [&](unsigned i) { return zs[i]; });
                if (pos >= 0)
                    hit = (int)(unsigned)(keys[pos] & 0xffffffffull);
            }
        } else {
            float* R = a.sr + base;
            float* Z = a.sz + base;
            unsigned* I = a.ssrc + base;
            for (unsigned kk = 2; kk <= P; kk <<= 1) {
                for (unsigned j = kk >> 1; j > 0; j >>= 1) {
                    const bool flip = (j == (kk >> 1));
                    for (unsigned tt = lane; tt < (P >> 1); tt += URF_STAR_THREADS) {
                        const unsigned lo = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
                        const unsigned hi2 = flip ? ((lo & ~(kk - 1)) + (kk - 1) - (lo & (kk - 1))) : lo + j;
                        if (hi2 < n) {
                            const unsigned long long ka = ((unsigned long long)urf_fbits(R[lo]) << 32) | I[lo];
                            const unsigned long long kb = ((unsigned long long)urf_fbits(R[hi2]) << 32) | I[hi2];
                            if (ka > kb) {
                                const float r0 = R[lo], z0 = Z[lo];
                                const unsigned i0 = I[lo];
                                R[lo] = R[hi2]; Z[lo] = Z[hi2]; I[lo] = I[hi2];
                                R[hi2] = r0; Z[hi2] = z0; I[hi2] = i0;
                            }
                        }
                    }
                    __threadfence_block();
                    __syncthreads();
                }
            }
            if (lane == 0) {
                const int pos = urf_slope_scan(
                    n, dp, [&](unsigned i) { return R[i]; }, [&](unsigned i) { return Z[i]; });
                if (pos >= 0)
                    hit = (int)I[pos];
            }
        }
    }
    if (lane == 0)
        a.star_hit[(size_t)s * K + k] = hit;
}

/* Instantiations (each sector is owned by exactly one): <-1,512,LDS> for
 * n <= 512 (also writes "no hit" for sectors with 0 or 1 point), <512,2048,LDS>
 * and <2048,-,global> for everything larger. */

/* ------------------------------------------------------------------------- */
/* k_ring                                                                      */
/* ------------------------------------------------------------------------- */
/* One workgroup per (ring, scan).  The ring's points (input order) stream
 * through LDS in chunks of 256 with a halo of curbPoints on both sides; every
 * thread owns one point and evaluates
 *   - x_zero for the triple (p - cp/2, p, p - cp/2 + cp) that marks p,
 *   - z_zero for the centre p,
 *   - azimuth and planar range of p,
 * then feeds the per-degree curb tables used by the beam march. */
__global__ __launch_bounds__(URF_RING_THREADS) void k_ring(urf_kargs a, urf_dev_params dp)
{
    constexpr int HALO = URF_MAX_CURB_POINTS;
    __shared__ float xs[URF_RING_THREADS + 2 * HALO], ys[URF_RING_THREADS + 2 * HALO], zs[URF_RING_THREADS + 2 * HALO];
    __shared__ int cmin[URF_DEG_CELLS], cmax[URF_DEG_CELLS];
    __shared__ int sh_q[4];
    __shared__ int sh_maxd;
    const unsigned c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK || c >= in.n_rings)
        return;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    const int n = (int)a.ring_cnt[(size_t)s * C + c];
    const unsigned base = off + a.ring_off[(size_t)s * (C + 1) + c];
    const int cp = dp.p.curbPoints;
    const bool star = dp.p.star_shaped_method != 0;
    const bool want_quad = (c == 1) && dp.p.blind_spots;

    for (unsigned i = tid; i < URF_DEG_CELLS; i += URF_RING_THREADS) {
        cmin[i] = URF_INT_NONE_MIN;
        cmax[i] = -1;
    }
    if (tid == 0) {
        sh_q[0] = (int)urf_fbits(0.f);
        sh_q[1] = (int)urf_fbits(180.f);
        sh_q[2] = (int)urf_fbits(180.f);
        sh_q[3] = (int)urf_fbits(360.f);
        sh_maxd = 0;
    }
    int maxd_bits = 0;
    __syncthreads();

    for (int cs = 0; cs < n; cs += URF_RING_THREADS) {
        /* stage [cs - cp, cs + 256 + cp) */
        const int lo = cs - cp < 0 ? 0 : cs - cp;
        const int hi = cs + URF_RING_THREADS + cp > n ? n : cs + URF_RING_THREADS + cp;
        for (int j = lo + (int)tid; j < hi; j += URF_RING_THREADS) {
            const int li = j - cs + cp;
            xs[li] = a.rx[base + j];
            ys[li] = a.ry[base + j];
            zs[li] = a.rz[base + j];
        }
        __syncthreads();
        const int p = cs + (int)tid;
        if (p < n) {
            const int lp = (int)tid + cp;   /* LDS slot of p */
            const float px = xs[lp], py = ys[lp], pz = zs[lp];
            unsigned flag = 0;

            if (star) {   /* lidar_segmentation.cpp:241-242: carry the star-shaped hit over */
                const unsigned src = a.rsrc[base + p];
                const unsigned sk = a.seckey[off + src];
                if (sk != URF_SEC_NONE && a.star_hit[(size_t)s * K + sk] == (int)src)
                    flag |= 1u;
            }

            if (dp.p.x_zero_method) {   /* x_zero_method.cpp:30-68, evaluated for the point it marks */
                const int j = p - cp / 2;
                if (j >= cp && j <= (n - 1) - cp) {
                    const int lj = lp - cp / 2, l3 = lj + cp;
                    const double dx = (double)(xs[l3] - xs[lj]), dy = (double)(ys[l3] - ys[lj]);
                    const float d = (float)__builtin_sqrt(dx * dx + dy * dy);
                    if ((double)d < 5.0) {
                        const float nyj = a.newY[j], ny2 = a.newY[p], ny3 = a.newY[j + cp];
                        const float zj = zs[lj], z3 = zs[l3];
                        double u, v;
                        u = (double)(ny2 - nyj); v = (double)(pz - zj);
                        const float x1 = (float)__builtin_sqrt(u * u + v * v);
                        u = (double)(ny3 - ny2); v = (double)(z3 - pz);
                        const float x2 = (float)__builtin_sqrt(u * u + v * v);
                        u = (double)(ny3 - nyj); v = (double)(z3 - zj);
                        const float x3 = (float)__builtin_sqrt(u * u + v * v);
                        const double num = (double)x3 * (double)x3 - (double)x1 * (double)x1 - (double)x2 * (double)x2;
                        const float den = (-2.0f * x1) * x2;
                        float br = (float)(num / (double)den);
                        if (br < -1.0f)
                            br = -1.0f;
                        else if (br > 1.0f)
                            br = 1.0f;
                        const float alpha = (float)((double)(urf_acosf(br) * 180.0f) / URF_PI_D);
                        if (alpha <= dp.p.angleFilter1 &&
                            (__builtin_fabsf(zj - pz) >= dp.p.curbHeight || __builtin_fabsf(z3 - pz) >= dp.p.curbHeight) &&
                            (double)__builtin_fabsf(zj - z3) >= 0.05)
                            flag |= 2u;
                    }
                }
            }

            if (dp.p.z_zero_method) {   /* z_zero_method.cpp:21-72 */
                if (p >= cp && p <= (n - 1) - cp) {
                    const double dx = (double)(xs[lp + cp] - xs[lp - cp]), dy = (double)(ys[lp + cp] - ys[lp - cp]);
                    const float d = (float)__builtin_sqrt(dx * dx + dy * dy);
                    if ((double)d < 5.0) {
                        const float az = __builtin_fabsf(pz);
                        float max1 = az, max2 = az;
                        float va1 = 0.f, va2 = 0.f, vb1 = 0.f, vb2 = 0.f;
                        for (int k = 1; k <= cp; k++) {
                            va1 = va1 + (xs[lp - k] - px);
                            va2 = va2 + (ys[lp - k] - py);
                            const float zk = __builtin_fabsf(zs[lp - k]);
                            if (zk > max1)
                                max1 = zk;
                        }
                        for (int k = 1; k <= cp; k++) {
                            vb1 = vb1 + (xs[lp + k] - px);
                            vb2 = vb2 + (ys[lp + k] - py);
                            const float zk = __builtin_fabsf(zs[lp + k]);
                            if (zk > max2)
                                max2 = zk;
                        }
                        va1 = dp.inv_cp * va1;
                        va2 = dp.inv_cp * va2;
                        vb1 = dp.inv_cp * vb1;
                        vb2 = dp.inv_cp * vb2;
                        const float num = va1 * vb1 + va2 * vb2;
                        const double na = __builtin_sqrt((double)va1 * (double)va1 + (double)va2 * (double)va2);
                        const double nb = __builtin_sqrt((double)vb1 * (double)vb1 + (double)vb2 * (double)vb2);
                        float br = (float)((double)num / (na * nb));
                        if (br < -1.0f)
                            br = -1.0f;
                        else if (br > 1.0f)
                            br = 1.0f;
                        const float alpha = (float)((double)(urf_acosf(br) * 180.0f) / URF_PI_D);
                        if (alpha <= dp.p.angleFilter2 &&
                            (max1 - az >= dp.p.curbHeight || max2 - az >= dp.p.curbHeight) &&
                            (double)__builtin_fabsf(max1 - max2) >= 0.05)
                            flag |= 4u;
                    }
                }
            }

            float d2;
            const float az = urf_azimuth(px, py, &d2);   /* lidar_segmentation.cpp:245-269 */
            a.raz[base + p] = az;
            a.rflag[base + p] = (uint8_t)flag;
            if (a.rd2)
                a.rd2[base + p] = d2;
            const int db = (int)urf_fbits(d2);           /* :271-274, d2 >= 0 */
            maxd_bits = db > maxd_bits ? db : maxd_bits;

            if (flag && az == az) {
                /* curb point: per-degree tables for the beam march.  The azimuth
                 * lies in [0,360]; cell_lo = largest integer <= az, cell_hi =
                 * smallest integer >= az. */
                int cl = (int)__builtin_floorf(az), ch = (int)__builtin_ceilf(az);
                cl = cl < 0 ? 0 : (cl > 360 ? 360 : cl);
                ch = ch < 0 ? 0 : (ch > 360 ? 360 : ch);
                const int ab = (int)urf_fbits(az);
                atomicMin(&cmin[cl], ab);
                atomicMax(&cmax[ch], ab);
            }
            if (want_quad && flag) {   /* blind_spots.cpp:19-56 */
                const int ab = (int)urf_fbits(az);
                if (az >= 0.f && az < 90.f)
                    atomicMax(&sh_q[0], ab);
                else if (az >= 90.f && az < 180.f)
                    atomicMin(&sh_q[1], ab);
                else if (az >= 180.f && az < 270.f)
                    atomicMax(&sh_q[2], ab);
                else if (az < 360.f)   /* "alpha < q4" with q4 starting at 360; NaN fails */
                    atomicMin(&sh_q[3], ab);
            }
        }
        __syncthreads();
    }

    atomicMax(&sh_maxd, maxd_bits);
    __syncthreads();
    if (tid == 0)
        a.maxdist[(size_t)s * C + c] = __uint_as_float((unsigned)sh_maxd);
    if (want_quad && tid < 4)
        a.quad[(size_t)s * 4 + tid] = __uint_as_float((unsigned)sh_q[tid]);

    /* sufmin[i] = min curb azimuth >= i ; premax[i] = max curb azimuth <= i ; NaN = none */
    float* sm = a.sufmin + ((size_t)s * C + c) * URF_DEG_CELLS;
    float* pm = a.premax + ((size_t)s * C + c) * URF_DEG_CELLS;
    for (unsigned i = tid; i < URF_DEG_CELLS; i += URF_RING_THREADS) {
        int mn = URF_INT_NONE_MIN;
        for (unsigned j = i; j < URF_DEG_CELLS; j++)
            mn = cmin[j] < mn ? cmin[j] : mn;
        int mx = -1;
        for (int j = (int)i; j >= 0; j--)
            mx = cmax[j] > mx ? cmax[j] : mx;
        sm[i] = mn == URF_INT_NONE_MIN ? __builtin_nanf("") : __uint_as_float((unsigned)mn);
        pm[i] = mx < 0 ? __builtin_nanf("") : __uint_as_float((unsigned)mx);
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

/* upper / lower end of beam i's window on ring k (blind_spots.cpp:107,136-143 / :216,245-252) */
__device__ __forceinline__ float urf_fwd_hi(const urf_dev_params& dp, int i, unsigned k, double qk)
{
    const float fi = (float)i;
    if (k == 0)
        return fi + dp.p.beamZone;
    if (fi == dp.fwd_limit)
        return 360.0f;
    return (float)((double)i + qk);
}
__device__ __forceinline__ float urf_bwd_lo(const urf_dev_params& dp, int i, unsigned k, double qk)
{
    const float fi = (float)i;
    if (k == 0)
        return fi - dp.p.beamZone;
    if (fi == dp.bwd_limit)
        return 0.0f;
    return (float)((double)i - qk);
}
/* arcDistance / ((maxDistance[k] * M_PI) / 180), blind_spots.cpp:65,142 */
__device__ __forceinline__ double urf_arc_ratio(const urf_dev_params& dp, float maxd0, float maxdk)
{
    const float arc = (float)((((double)maxd0 * URF_PI_D) / 180.0) * (double)dp.p.beamZone);
    return (double)arc / (((double)maxdk * URF_PI_D) / 180.0);
}

/* One thread per integer degree casts the forward and the backward beam that
 * start there and finds the first ring whose window holds a curb point. */
__global__ __launch_bounds__(URF_LABEL_THREADS) void k_beams(urf_kargs a, urf_dev_params dp)
{
    __shared__ double qk[URF_MAX_CHANNELS];
    __shared__ float q[4];
    const unsigned s = blockIdx.x, tid = threadIdx.x;
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK)
        return;
    const unsigned C = (unsigned)dp.p.channels, nR = in.n_rings;
    const float* maxd = a.maxdist + (size_t)s * C;
    if (tid < 4) {
        const float init[4] = { 0.f, 180.f, 180.f, 360.f };
        /* q1..q4 come from sorted ring 1 (blind_spots.cpp:19) */
        q[tid] = (dp.p.blind_spots && nR > 1) ? a.quad[(size_t)s * 4 + tid] : init[tid];
    }
    for (unsigned k = tid; k < nR; k += URF_LABEL_THREADS)
        qk[k] = urf_arc_ratio(dp, maxd[0], maxd[k]);
    __syncthreads();
    if (tid < 4 && !(dp.p.blind_spots && nR > 1))
        a.quad[(size_t)s * 4 + tid] = q[tid];
    const int i = (int)tid;
    if (i > 360)
        return;
    const float fi = (float)i;
    const bool blind = urf_blind(dp.p, q, i);
    int sf = -1, sb = -1;
    if (fi <= dp.fwd_limit && !blind) {   /* blind_spots.cpp:68 */
        sf = (int)nR;
        for (unsigned k = 0; k < nR; k++) {
            const float m = a.sufmin[((size_t)s * C + k) * URF_DEG_CELLS + i];
            if (m <= urf_fwd_hi(dp, i, k, qk[k])) {
                sf = (int)k;
                break;
            }
        }
    }
    if (fi >= dp.bwd_limit && !blind) {   /* blind_spots.cpp:177 */
        sb = (int)nR;
        for (unsigned k = 0; k < nR; k++) {
            const float m = a.premax[((size_t)s * C + k) * URF_DEG_CELLS + i];
            if (m >= urf_bwd_lo(dp, i, k, qk[k])) {
                sb = (int)k;
                break;
            }
        }
    }
    a.stop_f[(size_t)s * URF_DEG_CELLS + i] = (int16_t)sf;
    a.stop_b[(size_t)s * URF_DEG_CELLS + i] = (int16_t)sb;
}

/* ------------------------------------------------------------------------- */
/* k_label                                                                     */
/* ------------------------------------------------------------------------- */
/* A point of ring k is road iff it is no curb point and lies in the window of
 * a beam that reached beyond ring k.  Windows [i, hi_k(i)] grow with i, so it
 * suffices to test the largest such forward beam with i <= azimuth (and the
 * smallest such backward beam with i >= azimuth). */
__global__ __launch_bounds__(URF_LABEL_THREADS) void k_label(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned long long actf[6], actb[6];
    __shared__ unsigned cnt_road, cnt_curb;
    const unsigned c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK || c >= in.n_rings)
        return;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned C = (unsigned)dp.p.channels;
    const unsigned n = a.ring_cnt[(size_t)s * C + c];
    const unsigned base = off + a.ring_off[(size_t)s * (C + 1) + c];
    {
        const int i = (int)tid;
        const bool af = i <= 360 && (int)a.stop_f[(size_t)s * URF_DEG_CELLS + i] > (int)c;
        const bool ab = i <= 360 && (int)a.stop_b[(size_t)s * URF_DEG_CELLS + i] > (int)c;
        const unsigned long long bf = __ballot(af), bb = __ballot(ab);
        if (urf_lane() == 0) {
            actf[tid >> 6] = bf;
            actb[tid >> 6] = bb;
        }
        if (tid == 0) {
            cnt_road = 0;
            cnt_curb = 0;
        }
    }
    __syncthreads();
    const float* maxd = a.maxdist + (size_t)s * C;
    const double qk = urf_arc_ratio(dp, maxd[0], maxd[c]);
    const uint8_t lab0 = URF_FLAG_ROI | URF_FLAG_RING | (c == 10 ? URF_FLAG_RING10 : 0);
    unsigned my_road = 0, my_curb = 0;
    for (unsigned p = tid; p < n; p += URF_LABEL_THREADS) {
        const unsigned flag = a.rflag[base + p];
        const float az = a.raz[base + p];
        const unsigned src = a.rsrc[base + p];
        uint8_t lab = lab0;
        if (flag) {
            lab |= URF_LABEL_CURB;
            my_curb++;
        } else if (az == az) {
            bool road = false;
            int cf = (int)__builtin_floorf(az);
            cf = cf < 0 ? 0 : (cf > 360 ? 360 : cf);
            {
                int w = cf >> 6;
                const int b = cf & 63;
                unsigned long long mm = actf[w] & (b == 63 ? ~0ull : ((2ull << b) - 1ull));
                while (mm == 0 && w > 0)
                    mm = actf[--w];
                if (mm) {
                    const int i = w * 64 + 63 - __clzll((long long)mm);
                    road = az <= urf_fwd_hi(dp, i, c, qk);
                }
            }
            if (!road) {
                int cb = (int)__builtin_ceilf(az);
                cb = cb < 0 ? 0 : (cb > 360 ? 360 : cb);
                int w = cb >> 6;
                const int b = cb & 63;
                unsigned long long mm = actb[w] & (~0ull << b);
                while (mm == 0 && w < 5)
                    mm = actb[++w];
                if (mm) {
                    const int i = w * 64 + __ffsll((long long)mm) - 1;
                    road = az >= urf_bwd_lo(dp, i, c, qk);
                }
            }
            if (road) {
                lab |= URF_LABEL_ROAD;
                my_road++;
            }
        }
        a.labels[off + src] = lab;
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
        atomicAdd(&o->n_ring_pts, n);
        if (c == 10)
            atomicAdd(&o->n_ring10, n);
    }
}

/* ------------------------------------------------------------------------- */
/* index lists                                                                 */
/* ------------------------------------------------------------------------- */
/* Single workgroup, ascending index order (ballot + prefix per 1024 points). */
__global__ __launch_bounds__(1024) void k_compact(const uint8_t* __restrict__ labels, unsigned n,
                                                  unsigned* road, unsigned* curb, unsigned* roi, unsigned* ring10,
                                                  unsigned* counts)
{
    __shared__ unsigned wsum[4][16];
    __shared__ unsigned run[4];
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 4)
        run[tid] = 0;
    __syncthreads();
    for (unsigned b0 = 0; b0 < n; b0 += 1024) {
        const unsigned i = b0 + tid;
        const unsigned l = i < n ? labels[i] : 0;
        const bool f[4] = { (l & URF_LABEL_MASK) == URF_LABEL_ROAD, (l & URF_LABEL_MASK) == URF_LABEL_CURB,
                            (l & URF_FLAG_ROI) != 0, (l & URF_FLAG_RING10) != 0 };
        unsigned below[4];
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(f[k]);
            below[k] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0)
                wsum[k][wave] = __popcll(m);
        }
        __syncthreads();
        unsigned* outs[4] = { road, curb, roi, ring10 };
        for (int k = 0; k < 4; k++) {
            unsigned pre = run[k];
            for (unsigned w = 0; w < wave; w++)
                pre += wsum[k][w];
            if (f[k] && outs[k])
                outs[k][pre + below[k]] = i;
        }
        __syncthreads();
        if (tid < 4) {
            unsigned t = 0;
            for (int w = 0; w < 16; w++)
                t += wsum[tid][w];
            run[tid] += t;
        }
        __syncthreads();
    }
    if (tid < 4 && counts)
        counts[tid] = run[tid];
}

#endif /* URF_KERNELS_HPP */
