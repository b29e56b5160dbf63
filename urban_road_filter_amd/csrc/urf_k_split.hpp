/*
 * urf_k_split.hpp -- k_split / k_split_repair / k_split_list: the tile-local stable multi-split; k_index: the per-scan tables (lidar_segmentation.cpp:100-126, 207-278).
 * One of the kernel families of urf_kernels.hpp (r6: split by family, zero behaviour change); included from there, in order.
 */
#ifndef URF_K_SPLIT_HPP
#define URF_K_SPLIT_HPP

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
#define URF_SLOT(lp) ((lp) + ((lp) >> 5))
#define URF_SLOTS (URF_TILE + URF_TILE / 32)
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
    const unsigned wave = tid >> 6, lane = tid & 63;
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
    }
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
            a.info[s].status = (a.table_cause[s] & 3u) == 2u ? URF_STATUS_REDO_HINT : URF_STATUS_REDO_TABLE;
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


#endif /* URF_K_SPLIT_HPP */
