/*
 * urf_k_star.hpp -- the star-shaped search: k_star_sort_small / _runs / _mid / _big, k_star_ties, k_star_walk / _few (star_shaped_search.cpp:32-181).
 * One of the kernel families of urf_kernels.hpp (r6: split by family, zero behaviour change); included from there, in order.
 */
#ifndef URF_K_STAR_HPP
#define URF_K_STAR_HPP

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
            sreg[q] = RUNS ? ((slv[q] & URF_SLOT_OFF) ? 0xffffffffu : (adr[q] & ~(URF_TILE - 1u)) + slv[q]) : q * 64 + lane;
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
                        sreg[q] = (sl & URF_SLOT_OFF) ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
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
                    sreg[e] = (sl & URF_SLOT_OFF) ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
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
            I[i] = (sl & URF_SLOT_OFF) ? 0xffffffffu : lo * URF_TILE + sl;
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
    if (a.front && a.front_ok[s] == URF_FRONT_ROWS) {   /* (uniform) */
        /* a row-major scan (urf_front.hpp): the fused kernels hold the sector in firing order -- (tile, firing, laser) -- and the
         * reference met its points row by row: (laser, firing).  Rank by that key (a point's place in its tile stands in sslot). */
        for (unsigned i = lane; i < n; i += 64) {
            const unsigned adr = P[i];
            const unsigned sl = (unsigned)a.sslot[sb + adr] & (URF_TILE - 1u);
            LP[i] = ((sl & 63u) << 16) | ((adr / URF_TILE) * (URF_TILE / 64u) + (sl >> 6));
        }
        MEM::sync();
        for (unsigned i = lane; i < n; i += 64) {
            const unsigned key = LP[i];
            unsigned rank = 0;
            for (unsigned j = 0; j < n; j++)
                rank += LP[j] < key ? 1u : 0u;
            RP[i] = rank;
        }
        MEM::sync();
        for (unsigned i = lane; i < n; i += 64)
            LP[RP[i]] = R[i];
        MEM::sync();
        for (unsigned i = lane; i < n; i += 64)
            R[i] = LP[i];
        MEM::sync();
        for (unsigned i = lane; i < n; i += 64)
            LP[RP[i]] = P[i];
        MEM::sync();
        for (unsigned i = lane; i < n; i += 64)
            P[i] = LP[i];
        MEM::sync();
    }
    if constexpr (POST) {
        /* behind the walk: the point std::sort leaves at the index the walk stopped at (urf_walk_report's conversion of a
         * point's address in the sector-sorted arrays into its place in the ring-major ones) */
        const unsigned adr = urf_tie_select<MEM>(n, hit_i, R, P, LP, RP);
        const unsigned sl = a.sslot[sb + adr];
        const unsigned v = (sl & URF_SLOT_OFF) ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
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
                a.ssrt[base + i] = (sl & URF_SLOT_OFF) ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;   /* (RP may BE this stretch of ssrt: own element) */
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
            v = (sl & URF_SLOT_OFF) ? 0xffffffffu : (adr & ~(URF_TILE - 1u)) + sl;
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


#endif /* URF_K_STAR_HPP */
