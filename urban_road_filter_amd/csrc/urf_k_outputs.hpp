/*
 * urf_k_outputs.hpp -- the outputs on demand: index lists, the published order, the road_marker points; the device self tests (test hooks).
 * One of the kernel families of urf_kernels.hpp (r6: split by family, zero behaviour change); included from there, in order.
 */
#ifndef URF_K_OUTPUTS_HPP
#define URF_K_OUTPUTS_HPP

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


#endif /* URF_K_OUTPUTS_HPP */
