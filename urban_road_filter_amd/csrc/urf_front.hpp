/*
 * urf_front.hpp -- the fused front end for organised sweeps (r6): firing by firing, or row-major (height = the lasers) through k_transpose.
 *
 * What it replaces: k_split + k_ring (+ k_label's ring-sorted reads) for a scan whose points come as a spinning
 * LiDAR's driver delivers them -- firing after firing, every firing holding the sensor's 64 lasers in ONE fixed order
 * (elevation order, laser-number order, any permutation), points missing where there was no return or the region of
 * interest cut them off (lidar_segmentation.cpp:100-117), the points of a firing that take part in the star-shaped
 * search sharing one sector (star_shaped_search.cpp:164-171).
 *
 * The legacy path sorts every tile by ring so that a workgroup per ring can stream its points (k_split: 39 B/point,
 * k_ring: 7 B/point more).  Here LANE l of a wave IS laser l: the wave marches along the firings of its block, 64
 * points (one firing) per step, coalesced 256-byte loads per array, and the lane keeps the last eleven points of ITS
 * ring in registers -- the window both detectors look at for curbPoints == 5 (x_zero_method.cpp:30-67: the triple
 * (j, j + 2, j + 5); z_zero_method.cpp:21-72: the centre and five points on either side).  A lane whose point is
 * missing simply does not shift its window: a hole costs nothing and needs no compaction.  Nothing is sorted by ring,
 * nothing is transposed, x / y / z are read once and only the 4-byte record (input order) and the star-shaped search's
 * sector-sorted copies leave the kernel: 26 B/point instead of 46.
 *
 *   k_front        grid (blocks of URF_FRONT tiles, scans) x 64 threads.  Per step: region of interest, ring (the
 *                  lane's expected table entry first, the reference's exact sequence for the rare point that is not
 *                  surely on it), sector, azimuth code, the cheap height tests of both detectors for the point whose
 *                  window has just become complete; what passes goes to the scan's candidate list.  Per tile: the
 *                  sector run table k_index and the sort kernels read (tsoff), the presence word of every lane.
 *                  A block starts URF_FRONT_HPRE firings early and ends URF_FRONT_HPOST firings late (windows
 *                  across block borders); what a hole in the halo leaves undecided goes to the list as an EDGE item.
 *   k_front_finish one workgroup per scan, behind the star-shaped search: ring sizes and positions from the presence
 *                  words, the angle tests of the candidates (f64 chains, all lanes busy), the star-shaped hits, the
 *                  rings' curb lists, largest ranges and quadrants for k_beams.
 *   k_label_front  k_label in input order: record -> label byte, no LDS image.
 *   k_transpose    (row-major organised sweeps: point l * F + f, front_ok[s] == URF_FRONT_ROWS, decided by k_rows_probe + k_ring_table,
 *                  urf_k_table.hpp) the firing-order copy of x / y / z the three kernels above read instead of the caller's arrays;
 *                  everything between input and output is indexed by firing * 64 + laser, k_label_front stores the labels row-major
 *                  and k_star_ties (urf_k_star.hpp) ranks a sector in the reference's row order before it re-enacts std::sort.
 *
 * A scan that does not have the shape (a lane that meets two rings, two lanes on one ring, a firing in two sectors,
 * sectors that fall inside a tile, a ring point on the sensor's axis, an incomplete speculative ring table) clears
 * front_ok[s] and takes the legacy kernels, which skip every scan whose flag is still set.  Labels are the same
 * either way, bit for bit: the arithmetic is the legacy path's (urf_device.hpp, urf_x_zero_angle, urf_z_zero_angle).
 */
#ifndef URF_FRONT_HPP
#define URF_FRONT_HPP

#define URF_FRONT_HPRE 8u      /* firings in front of a block: 7 fill a window without holes (x_zero's j >= 5 rule) */
#ifndef URF_FRONT_HPOST
#define URF_FRONT_HPOST 8u     /* firings behind it: 5 complete the last centre's window */
#endif
#define URF_FRONT_LANES 64u
#define URF_FRONT_STEPS (URF_TILE / URF_FRONT_LANES)   /* firings per tile */
#define URF_FRONT_MIN_SCANS 192u  /* below (mode 1): the general kernels.  A block of k_front is ONE wave marching 80-144 dependent steps, k_front_finish one
                                   * workgroup per scan: with few scans the device is empty and the chains are the time (tools/r6_min_scans.py, general / fused ms per call:
                                   * 32 sweeps 0.21 / 0.25-0.33, 64: 0.31 / 0.32-0.40, 128: 0.45 / 0.44-0.50, 256: 0.73 / 0.66-0.70, 1024: 2.44 / 2.1) */
#define URF_FRONT_TPB_SMALL 2u    /* tiles per block of k_front for batches below URF_FRONT_TPB_SCANS scans (more, shorter chains), ... */
#define URF_FRONT_TPB_LARGE 4u    /* ... and from there on (less halo): 256 sweeps 0.663 / 0.703 ms at 2 / 4, 1024 sweeps 0.829 / 0.820 */
#define URF_FRONT_TPB_SCANS 512u
#define URF_FRONT_MAX_TILES 128u   /* k_front_finish keeps a presence word per (tile, lane) in LDS */
#define URF_FRONT_RING_NONE 0x7fu  /* ring field of an input-order record */
/* candidate kinds */
#define URF_FC_XZ 1u       /* passed x_zero's height tests as the marked point: angle test pending */
#define URF_FC_ZZ 2u       /* passed z_zero's as the centre */
#define URF_FC_EDGE_X 4u   /* x_zero not evaluated by the march (window incomplete at a block border): everything pending */
#define URF_FC_EDGE_Z 8u
#ifndef URF_FRONT_NT
#define URF_FRONT_NT 0   /* cache policy of k_front's stores (2 = non-temporal).  A sweep with drop-outs compacts its star records, a step's 64 stores then straddle
                          * cache lines that the next step completes: non-temporal, such half-written lines left the L2 at once (the address queue in
                          * front of it full 18 x as often as on a sweep without holes, k_front 1.50 ms per 1024 sensor-like sweeps against 0.98) */
#endif
#ifndef URF_FRONT_WAVES
#define URF_FRONT_WAVES 5   /* 88 registers; at 6 (80) the kernel reloads a spilled constant in every step, behind an s_waitcnt vmcnt(0) that drains its prefetch */
#endif

struct urf_front_thr {
    float y, z, below;   /* (an entry the table does not hold: y = +inf, nothing lies inside) */
};
__device__ __forceinline__ urf_front_thr urf_front_load_thr(const urf_kargs& a, unsigned s, unsigned C, unsigned e, unsigned nR)
{
    urf_front_thr t;
    const bool valid = e < nR;
    const float4 tv = ((const float4*)a.ring_thr)[(size_t)s * C + (valid ? e : 0u)];
    const float bl = a.ring_thr[((size_t)s * C + ((valid && e) ? e - 1u : 0u)) * 4];
    t.y = valid ? tv.y : __builtin_inff();
    t.z = tv.z;
    t.below = e ? bl : __builtin_inff();
    return t;
}

/* raw buffer descriptors (gfx9 word 3: 32-bit untyped): a lane whose byte offset lies at or beyond `bytes` loads zero and stores
 * nothing -- the march's loads behind the scan's end and its predicated stores need neither a select nor a branch, and the
 * compiler sees a straight line of memory operations (its s_waitcnt for the points loaded four firings ago then leaves every
 * younger operation in flight: loads and stores share ONE in-order counter on this chip, and a wait that cannot count the stores
 * in between waits for all of them -- the first version of this kernel spent two thirds of its time there) */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t urf_buf(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
#define URF_OOB 0xffffffffu

/* the arrays the fused kernels read a scan's points from, indexed by firing * 64 + laser: the caller's for a sweep in firing order,
 * k_transpose's copy for a row-major one (front_ok[s] == URF_FRONT_ROWS) */
__device__ __forceinline__ void urf_front_src(const urf_kargs& a, unsigned s, unsigned off, unsigned ok, const float*& gx, const float*& gy, const float*& gz)
{
    const bool rows = ok == URF_FRONT_ROWS;   /* (uniform) */
    const size_t o = rows ? (size_t)urf_sbase(a, s) : (size_t)off;
    gx = (rows ? (const float*)a.tx : a.x) + o;
    gy = (rows ? (const float*)a.ty : a.y) + o;
    gz = (rows ? (const float*)a.tz : a.z) + o;
}

/* Row-major organised sweep -> firing order: tx[f * 64 + l] = x[l * F + f].  Grid (tiles, scans) x 256 threads, a tile = 32 firings:
 * 64 rows x 128 bytes in (a cache line per row), LDS, 8 KB out in one stretch; 24 B/point, HBM-bound. */
__global__ __launch_bounds__(256) void k_transpose(urf_kargs a)
{
    __shared__ float T[3][64][33];
    const unsigned s = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    if (a.front_ok[s] != URF_FRONT_ROWS)
        return;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned F = len >> 6, f0 = t * URF_FRONT_STEPS;
    if (f0 >= F)
        return;
    const unsigned c = tid & 31u, r8 = tid >> 5;
    const bool in = f0 + c < F;
    float vx[8], vy[8], vz[8];
#pragma unroll
    for (unsigned j = 0; j < 8; j++) {
        const size_t i = (size_t)off + (size_t)(j * 8u + r8) * F + f0 + (in ? c : 0u);
        vx[j] = a.x[i];
        vy[j] = a.y[i];
        vz[j] = a.z[i];
    }
#pragma unroll
    for (unsigned j = 0; j < 8; j++) {
        T[0][j * 8u + r8][c] = vx[j];
        T[1][j * 8u + r8][c] = vy[j];
        T[2][j * 8u + r8][c] = vz[j];
    }
    __syncthreads();
    const size_t ob = (size_t)urf_sbase(a, s) + (size_t)f0 * 64u;
    const unsigned nf = F - f0 < URF_FRONT_STEPS ? F - f0 : URF_FRONT_STEPS;
#pragma unroll
    for (unsigned j = 0; j < 8; j++) {
        const unsigned v = j * 256u + tid, f = v >> 6, l = v & 63u;
        if (f < nf) {
            a.tx[ob + v] = T[0][l][f];
            a.ty[ob + v] = T[1][l][f];
            a.tz[ob + v] = T[2][l][f];
        }
    }
}

/* What the fast decisions of a firing leave open -- a point that is not SURELY on its lane's table entry, whose sector is within
 * the margin of a border, or that the approximations refuse: the reference's exact sequence (urf_exact_keys).  Returns bit 0: on
 * the lane's ring; bits 1-11: sector + 1 (0: none wanted); bits 12-18 + URF_FO_ADOPT: the lane has no confirmed entry yet and the
 * point lies on this one; URF_FO_NONE: on no ring; URF_FO_FAIL: the scan does not have the shape (one lane, two rings; a ring point
 * on the sensor's axis, whose azimuth is NaN: k_nan_rings, legacy path).  NOT inlined: a call under a branch that is rarely
 * taken, and nothing the hot path keeps in registers depends on what happens in here. */
#define URF_FO_ADOPT 0x20000000u
#define URF_FO_NONE 0x40000000u
#define URF_FO_FAIL 0x80000000u
__device__ __noinline__ unsigned urf_front_open(const float* tab, unsigned nR, float interval, float x, float y, float z, unsigned sectors, float Kfi,
                                                unsigned E, unsigned econf)
{
    const urf_exact_key ek = urf_exact_keys_body(tab, nR, interval, x, y, z, sectors, Kfi);
    unsigned r = sectors ? ((ek.sector + 1u) & 0x7ffu) << 1 : 0u;
    if (ek.ring == URF_RING_NONE)
        r |= URF_FO_NONE;
    else if (x == 0.0f && y == 0.0f)
        r |= URF_FO_FAIL;
    else if (ek.ring == E)
        r |= 1u;
    else if (!econf)
        r |= 1u | URF_FO_ADOPT | (ek.ring << 12);
    else
        r |= URF_FO_FAIL;
    return r;
}

/* the wave's candidate buffer (LDS; the workgroup IS the wave) and its flush into the scan's list: one atomic per ~150 candidates */
#define URF_FRONT_CBUF 256u
__device__ __forceinline__ void urf_front_flush(const urf_kargs& a, unsigned s, urf_u2* cbuf, unsigned& ncb, bool& overflow)
{
    if (ncb == 0u)
        return;   /* (uniform) */
    urf_wave_lds_sync();
    unsigned base = 0;
    if (urf_lane() == 0u)
        base = atomicAdd(&a.front_ncand[s], ncb);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    for (unsigned j = urf_lane(); j < ncb; j += 64u) {
        if (base + j < a.front_cand_cap)
            a.front_cand[(size_t)s * a.front_cand_cap + base + j] = cbuf[j];
        else
            overflow = true;
    }
    urf_wave_lds_sync();
    ncb = 0;
}
__device__ __forceinline__ void urf_front_push(urf_u2* cbuf, unsigned& ncb, bool has, unsigned idx, unsigned what)
{
    const unsigned long long m = __ballot(has);
    if (has)
        cbuf[ncb + urf_popc_below(m)] = urf_u2{ idx, what };
    ncb += (unsigned)__popcll(m);
}

template <bool STAR, bool BEAM>
__device__ __forceinline__ void urf_front_body(const urf_kargs& a, const urf_dev_params& dp, urf_u2* cbuf)
{
    const unsigned s = blockIdx.y, b = blockIdx.x, lane = threadIdx.x;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned TPB = a.front_tpb;
    const unsigned t_first = b * TPB;
    if (t_first * URF_TILE >= len)
        return;
    const unsigned ok = a.front_ok[s];
    if (ok == 0u)
        return;
    constexpr unsigned C = URF_FRONT_LANES;
    const unsigned K = (unsigned)dp.p.sectors;
    const unsigned nf = (len + 63u) >> 6;                                   /* firings of the scan */
    const unsigned F0 = t_first * URF_FRONT_STEPS;
    const unsigned F1 = F0 + TPB * URF_FRONT_STEPS < nf ? F0 + TPB * URF_FRONT_STEPS : nf;
    const unsigned Fs = F0 > URF_FRONT_HPRE ? F0 - URF_FRONT_HPRE : 0u;
    const unsigned Fe = F1 + URF_FRONT_HPOST < nf ? F1 + URF_FRONT_HPOST : nf;
    const bool from_start = Fs == 0u;   /* the window count IS the ring position + 1 */
    const bool to_end = Fe == nf;       /* no point of the scan lies behind the march */
    static_assert(URF_FRONT_HPRE % 4u == 0u && URF_FRONT_STEPS % 4u == 0u, "the march runs in groups of four firings");
    const unsigned sb = urf_sbase(a, s);
    const float *gx, *gy, *gz;
    urf_front_src(a, s, off, ok, gx, gy, gz);
    const __amdgpu_buffer_rsrc_t brec = urf_buf(a.rec + sb, len * 4u);
    const __amdgpu_buffer_rsrc_t bsr = urf_buf(a.sr + sb, a.tiles * URF_TILE * 4u), bsz = urf_buf(a.sz + sb, a.tiles * URF_TILE * 4u);
    const __amdgpu_buffer_rsrc_t bss = urf_buf(a.sslot + sb, a.tiles * URF_TILE * 2u);
    const unsigned nR = a.info[s].n_rings;
    const unsigned upto_v = a.table_upto[s];
    /* (a row-major scan's table rests on EVERY point lying on its row's entry -- k_ring_table's third rule: a point on none asks for the long walk) */
    const unsigned upto = ok == URF_FRONT_ROWS ? 0u : (nR < C ? upto_v : 0xffffffffu);
    const float* const tab = a.angle + (size_t)s * dp.p.channels;
    const float curbH = dp.p.curbHeight;
    const bool use_x = dp.p.x_zero_method != 0, use_z = dp.p.z_zero_method != 0;

    unsigned E = lane;        /* the table entry this lane's points are expected on */
    bool econf = false;       /* ... and a point of this march has confirmed it */
    urf_front_thr th = urf_front_load_thr(a, s, C, E, nR);
    bool failed = false, overflow = false;
    unsigned long long failed_m = 0;   /* ... what the hot path finds wrong, as lane masks (wave-uniform) */
    unsigned ncb = 0;         /* candidates in the wave's buffer */

    /* the lane's window: w10 = its newest ring point, w0 the one ten before; firing (relative to Fs) of the six newest */
    float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, w5 = 0.f, w6 = 0.f, w7 = 0.f, w8 = 0.f, w9 = 0.f, w10 = 0.f;
    unsigned fwA = 0, fwB = 0, fwC = 0;   /* (f6 << 16 | f5), (f8 << 16 | f7), (f10 << 16 | f9) */
    unsigned wc = 0;                       /* points in the window, at most 11 */
    unsigned tot = 0, nin = 0;             /* ring points of this lane since the block's first firing / inside the block */
    double maxs = 0.0;                     /* largest x*x + y*y among the lane's ring points of the block (maxDistance, lidar_segmentation.cpp:271-274) */
    unsigned pw = 0;                       /* presence bits of the tile at hand */
    /* per tile (wave-uniform) */
    unsigned troi = 0, tstar = 0;
    int stepkey_v = (int)URF_SEC_NONE, stepcnt_v = 0;   /* lane j: sector / participating points of step j of the tile */

    /* one firing.  PH 0: the halo in front of the block (windows fill), 1: the block, 2: the halo behind it (windows complete) */
    auto step = [&](auto ph, const unsigned f, const float x, const float y, const float z) {
        constexpr unsigned PH = decltype(ph)::value;
        const unsigned stp = f % URF_FRONT_STEPS;
        const unsigned i = f * 64u + lane;
        const bool roi = i < len && urf_in_roi(dp.p, x, y, z);
        const unsigned long long roim = __ballot(roi);
        if (PH == 1u && lane == 0u)
            a.roi_bits[((size_t)s * a.tiles + f / URF_FRONT_STEPS) * URF_FRONT_STEPS + stp] = roim;
        if (roim == 0ull)
            return;   /* (uniform) nothing of this firing lies in the region of interest */
        const float rho2 = x * x + y * y;
        const float u = -z * __builtin_amdgcn_rsqf(rho2);   /* urf_fast_cot */
        const bool fast = (rho2 >= URF_FAST_MIN2) & (rho2 <= URF_FAST_MAX2) & (__builtin_fabsf(u) <= URF_LUT_UMAX) & roi;
        const bool on_f = fast & (u >= th.y) & (u <= th.z) & (u < th.below);
        float fi = 0.f;
        int fs = -1;
        if (PH == 1u) {
            fi = urf_fast_polar(x, y);
            if (STAR)
                fs = fast ? urf_fast_sector_ranged(fi, dp.Kfi, K, dp.sector_margin) : -1;
        }
        bool on = on_f;
        const bool open = roi && !(on_f && (PH != 1u || !STAR || fs >= 0));
        /* (r6, vector-issue diet: a ballot of anything but a direct compare costs two vector instructions -- v_cndmask 0 / 1, v_cmp -- to
         * mask it with exec; a plain divergent branch skips its block when no lane takes it for two SCALAR instructions.  Wave-level
         * flags (failed, overflow) are OR-ed up as masks, per-lane state that only rare paths read (econf) likewise.) */
        if (open) {   /* rare: the reference's exact sequence for the lanes that need it */
            const unsigned r = urf_front_open(tab, nR, dp.p.interval, x, y, z, (PH == 1u && STAR) ? K : 0u, dp.Kfi, E, econf ? 1u : 0u);
            on = (r & 1u) != 0u;
            fs = (int)((r >> 1) & 0x7ffu) - 1;
            if (r & URF_FO_FAIL)
                failed = true;
            if (PH == 1u && (r & URF_FO_NONE) && i >= upto)
                a.table_redo[s] = 1u;   /* the speculative ring table is incomplete (k_table_repair, legacy path) */
            if (r & URF_FO_ADOPT) {   /* this lane's laser sits on another table entry: learned from its first point */
                E = (r >> 12) & 0x7fu;
                th = urf_front_load_thr(a, s, C, E, nR);
            }
        }
        if (PH == 1u) {
            /* the record, input order: ring | azimuth code (URF_REC_*; detector hits are OR-ed in by k_front_finish) */
            const unsigned azc_v = urf_az_code(urf_fast_azimuth_of(fi));   /* (unconditionally, then a select: a branch around eight instructions costs more) */
            const unsigned azc = urf_fast_az_ok(x, y) ? azc_v : URF_REC_AZ_UNKNOWN;
            __builtin_amdgcn_raw_buffer_store_b32((azc << URF_REC_AZ_SHIFT) | (on ? E : URF_FRONT_RING_NONE), brec, i * 4u, 0, URF_FRONT_NT);
            if (STAR) {
                /* star-shaped search: the firing's participants share one sector */
                unsigned sk = (unsigned)fs;
                if (BEAM && roi && !urf_in_beam(a.beams[fs < 0 ? 0 : fs], x, y))
                    sk = URF_SEC_NONE;
                /* (every lane is active here: the masks of direct compares need no exec) */
                const unsigned long long psm = roim & __builtin_amdgcn_ballot_w64(sk != URF_SEC_NONE) & __builtin_amdgcn_ballot_w64((int)sk >= 0);
                const bool ons = roi && sk != URF_SEC_NONE && (int)sk >= 0;
                const unsigned src = psm ? (unsigned)__ffsll((long long)psm) - 1u : 0u;
                const unsigned f0 = psm ? (unsigned)__builtin_amdgcn_readlane((int)sk, (int)src) : URF_SEC_NONE;
                failed_m |= psm & __builtin_amdgcn_ballot_w64(sk != f0);
                const unsigned so = (f / URF_FRONT_STEPS) * URF_TILE + tstar + urf_popc_below(psm);
                const unsigned o4 = ons ? so * 4u : URF_OOB;
                const float pr = urf_sqrt_rn_normal(rho2);   /* star_shaped_search.cpp:164: sqrtf(x * x + y * y) */
                failed_m |= psm & ~(__builtin_amdgcn_ballot_w64(rho2 >= 0x1p-90f) & __builtin_amdgcn_ballot_w64(rho2 <= 0x1p126f));   /* (outside the shortcut's interval: the legacy kernels) */
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pr), bsr, o4, 0, URF_FRONT_NT);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z), bsz, o4, 0, URF_FRONT_NT);
                __builtin_amdgcn_raw_buffer_store_b16((short)((stp * 64u + lane) | (on ? 0u : URF_SLOT_OFF)), bss, ons ? so * 2u : URF_OOB, 0, URF_FRONT_NT);
                stepkey_v = lane == stp ? (int)f0 : stepkey_v;
                stepcnt_v = lane == stp ? (int)__popcll(psm) : stepcnt_v;
                tstar += (unsigned)__popcll(psm);
            }
            troi += (unsigned)__popcll(roim);
        }
        /* the lane's window moves on by its new ring point; what has become decidable is decided */
        if (on) {
            if (PH == 1u) {
                const double s2 = (double)x * (double)x + (double)y * (double)y;
                maxs = s2 > maxs ? s2 : maxs;
                pw |= 1u << stp;
            }
            econf = true;
            w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7; w7 = w8; w8 = w9; w9 = w10;
            w10 = z;
            fwA = __builtin_amdgcn_alignbit(fwB, fwA, 16);
            fwB = __builtin_amdgcn_alignbit(fwC, fwB, 16);
            fwC = __builtin_amdgcn_alignbit(f - Fs, fwC, 16);
            wc = wc < 11u ? wc + 1u : 11u;
            if (PH >= 1u)
                tot++;
            if (PH == 1u)
                nin++;
        }
        if (PH == 0u)
            return;
        /* (straight-line: every lane computes, the lanes without a new point are masked out at the end -- branches around the
         * tests put the four results through registers and selects) */
        const bool full = wc == 11u;
        /* the centre (five points back) and the point x_zero marks (three back): are they this block's */
        const bool c_in = on & (tot >= 6u) & (tot - 5u <= nin);
        const bool p_in = on & (tot >= 4u) & (tot - 3u <= nin);
        /* z_zero_method.cpp:39-40, 48-49, 67-69 */
        const float a5 = __builtin_fabsf(w5);
        /* (window values are region-of-interest points' heights, never NaN: v_max3 with |.| modifiers, three instructions per side --
         * the compiler's fmaxf quiets the first two operands of every chain with a v_max x, x each: five) */
        float m1, m2, t1, t2;
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t1) : "v"(w0), "v"(w1), "v"(w2));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(t1) : "v"(t1), "v"(w3), "v"(w4));
        asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m1) : "v"(t1), "v"(w5));
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t2) : "v"(w10), "v"(w9), "v"(w8));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(t2) : "v"(t2), "v"(w7), "v"(w6));
        asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m2) : "v"(t2), "v"(w5));
        const bool hz = ((m1 - a5 >= curbH) | (m2 - a5 >= curbH)) & (__builtin_fabsf(m1 - m2) >= 0.05f);   /* ((double)v >= 0.05 <=> v >= 0.05f: the float above 0.05) */
        /* x_zero_method.cpp:62-64 for the triple (w5, w7, w10) = (j, j + 2, j + 5) */
        const bool hx = ((__builtin_fabsf(w5 - w7) >= curbH) | (__builtin_fabsf(w10 - w7) >= curbH)) & (__builtin_fabsf(w5 - w10) >= 0.05f);
        /* (lane masks combined as masks: `&` on bools mixed with the wave-uniform switches went through 0 / 1 integers in vector
         * registers -- v_cndmask, v_and, v_cmp per term) */
        bool zz = false, xz = false, ez = false, ex = false;
        /* a window that began inside this march (a block border with a hole in the halo, a ring that has just entered the
         * region of interest): positions unknown here, k_front_finish decides */
        if (use_z) {   /* (uniform) */
            zz = c_in && full && hz;
            if (!from_start)
                ez = c_in && !full;
        }
        if (use_x) {
            xz = p_in && full && hx;
            if (!from_start)
                ex = p_in && !full;
        }
        if (__ballot(zz | xz | ez | ex) != 0ull) {   /* (uniform) */
            urf_front_push(cbuf, ncb, zz | ez, ((fwA & 0xffffu) + Fs) * 64u + lane, zz ? URF_FC_ZZ : URF_FC_EDGE_Z);
            urf_front_push(cbuf, ncb, xz | ex, ((fwB & 0xffffu) + Fs) * 64u + lane, xz ? URF_FC_XZ : URF_FC_EDGE_X);
            if (ncb > URF_FRONT_CBUF - 128u)
                urf_front_flush(a, s, cbuf, ncb, overflow);
        }
    };
    auto tile_end = [&](const unsigned t) {
        const size_t row = (size_t)s * a.tiles + t;
        a.front_pres[row * 64u + lane] = pw;
        pw = 0;
        if (lane == 0)
            a.tile_roi[row] = troi;
        if (STAR) {
            /* sector k starts with the first step whose sector is >= k (urf_split_holey's construction: the steps' keys
             * with the empty steps filled in from the left must not fall; bisection over the 32 of them) */
            const unsigned k1 = (lane < URF_FRONT_STEPS && (unsigned)stepkey_v != URF_SEC_NONE) ? (unsigned)stepkey_v + 1u : 0u;
            const unsigned fk = urf_wave_scan_max(k1);
            unsigned exc = (unsigned)__shfl_up((int)fk, 1);
            exc = lane == 0 ? 0u : exc;
            if (__ballot(k1 != 0u && k1 < exc) != 0ull)
                failed = true;   /* (uniform) the sectors fall inside the tile (the sweep's seam, an unorganised cloud) */
            const unsigned sc = lane < URF_FRONT_STEPS ? (unsigned)stepcnt_v : 0u;
            const unsigned sinc = urf_wave_scan_add(sc);
            const unsigned sbase_l = sinc - sc;   /* lane j: participating points of the steps in front of step j; lane 32: all */
            for (unsigned k0 = 0; k0 <= K; k0 += 64u) {
                const unsigned k = k0 + lane;
                unsigned lo = 0;   /* number of steps whose filled-in key + 1 is < k + 1 */
#pragma unroll
                for (unsigned st = URF_FRONT_STEPS / 2; st > 0; st >>= 1) {
                    const unsigned v = (unsigned)__shfl((int)fk, (int)(lo + st - 1u));
                    lo += v < k + 1u ? st : 0u;
                }
                {
                    const unsigned v = (unsigned)__shfl((int)fk, (int)lo);
                    lo += (lo == URF_FRONT_STEPS - 1u && v < k + 1u) ? 1u : 0u;
                }
                const unsigned so = (unsigned)__shfl((int)sbase_l, (int)lo);   /* (lane 32 holds the tile's total) */
                if (k <= K)
                    a.tsoff[row * (K + 1) + k] = (uint16_t)so;
            }
        }
        troi = 0;
        tstar = 0;
        stepkey_v = (int)URF_SEC_NONE;
        stepcnt_v = 0;
    };

    /* The points arrive four firings ahead, in four register sets that are refilled as soon as they have been used: no
     * copies between sets (a copy waits for its source), every wait is for a load issued three firings ago. */
    float px[4], py[4], pz[4];
    auto ld = [&](const unsigned j, const unsigned f) {
        const unsigned i = f * 64u + lane, o = i < len ? i : len - 1u;   /* (a lane behind the scan's end: some point of the scan, never looked at) */
        px[j] = gx[o];
        py[j] = gy[o];
        pz[j] = gz[o];
    };
#pragma unroll
    for (unsigned j = 0; j < 4; j++)
        ld(j, Fs + j);
    for (unsigned f = Fs; f < F0; f += 4) {
#pragma unroll
        for (unsigned j = 0; j < 4; j++) {
            step(std::integral_constant<unsigned, 0u>{}, f + j, px[j], py[j], pz[j]);
            ld(j, f + j + 4u);
        }
    }
    for (unsigned f = F0; f < F1; f += 4) {
#pragma unroll
        for (unsigned j = 0; j < 4; j++) {
            if (f + j < F1)   /* (uniform: the scan's last tile may end anywhere) */
                step(std::integral_constant<unsigned, 1u>{}, f + j, px[j], py[j], pz[j]);
            ld(j, f + j + 4u);
        }
        if (((f + 4u) % URF_FRONT_STEPS) == 0u || f + 4u >= F1) {   /* (uniform) the tile is complete */
            tile_end(f / URF_FRONT_STEPS);
            if (failed_m != 0ull || __ballot(failed | overflow) != 0ull) {   /* (uniform) */
                a.front_ok[s] = 0u;
                if (ok == URF_FRONT_ROWS)
                    a.table_redo[s] = 1u;   /* (nobody has checked the rest of the scan against the rows' table) */
                return;
            }
        }
    }
    for (unsigned f = F1; f < Fe; f += 4) {
#pragma unroll
        for (unsigned j = 0; j < 4; j++) {
            if (f + j < Fe)
                step(std::integral_constant<unsigned, 2u>{}, f + j, px[j], py[j], pz[j]);
            ld(j, f + j + 4u);
        }
    }
    /* the block's last ring points of every lane: their windows reach behind the march */
    if (!to_end) {
        const unsigned fr[5] = { fwA >> 16, fwB & 0xffffu, fwB >> 16, fwC & 0xffffu, fwC >> 16 };   /* entries 6..10 */
#pragma unroll
        for (unsigned e = 0; e < 5; e++) {
            const unsigned back = 4u - e;   /* entry 6 + e is `back` points behind the newest */
            const bool in = tot > back && tot - back <= nin;
            const unsigned what = (use_z ? URF_FC_EDGE_Z : 0u) | ((use_x && e >= 2u) ? URF_FC_EDGE_X : 0u);
            if (ncb > URF_FRONT_CBUF - 64u)
                urf_front_flush(a, s, cbuf, ncb, overflow);
            urf_front_push(cbuf, ncb, in && what != 0u, (fr[e] + Fs) * 64u + lane, what);
        }
    }
    urf_front_flush(a, s, cbuf, ncb, overflow);
    a.front_maxs[((size_t)s * a.tiles + b) * 64u + lane] = (unsigned long long)__double_as_longlong(maxs);
    /* one lane, one ring -- over the whole scan: the blocks agree through the scan's two tables */
    if (econf) {
        const unsigned o1 = atomicCAS(&a.front_lane_ring[(size_t)s * 64u + lane], 0xffffffffu, E);
        const unsigned o2 = atomicCAS(&a.front_ring_lane[(size_t)s * C + E], 0xffffffffu, lane);
        failed = failed | (o1 != 0xffffffffu && o1 != E) | (o2 != 0xffffffffu && o2 != lane);
    }
    if (failed_m != 0ull || __ballot(failed | overflow) != 0ull) {
        a.front_ok[s] = 0u;
        if (ok == URF_FRONT_ROWS)
            a.table_redo[s] = 1u;
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(URF_FRONT_WAVES, URF_FRONT_WAVES))) void k_front(urf_kargs a, urf_dev_params dp)
{
    __shared__ urf_u2 cbuf[URF_FRONT_CBUF];
    if (!dp.p.star_shaped_method)
        urf_front_body<false, false>(a, dp, cbuf);
    else if (!dp.p.starbeam_filter)
        urf_front_body<true, false>(a, dp, cbuf);
    else
        urf_front_body<true, true>(a, dp, cbuf);
}

/* ------------------------------------------------------------------------- */
/* k_front_finish                                                              */
/* ------------------------------------------------------------------------- */
#define URF_FRONT_ST_WORDS 72u   /* k_front_finish part 1 -> part 2: 64 list lengths, 4 quadrant values, the length of the list of marks */
#define URF_FINISH_CHUNK 768u   /* candidates per chunk (each may be listed twice): 12 KB of LDS */
#ifndef URF_FINISH_THREADS
#define URF_FINISH_THREADS 256   /* four waves and <= 40 KB of LDS per scan: four workgroups per CU, a batch of 1024 scans in one round (A/B 256 / 512 /
                                    * 1024 threads: cfg3 0.186 / 0.184 / 0.206 ms, the reference's default region of interest 0.210 / 0.241 / 0.264) */
#endif
struct urf_finish_shared {
    unsigned n[64];              /* ring points of lane l */
    unsigned ring[64];           /* its ring (0xffffffff: the lane holds no ring point) */
    unsigned ncurb[URF_FRONT_LANES];   /* curb points of ring r (the fused front end runs with 64 channels) */
    int q[4];
    unsigned n_pend;             /* entries of the scan's list of points to mark */
    unsigned nx, nz;             /* x_zero / z_zero items of the chunk at hand */
    unsigned qsum[URF_FINISH_THREADS / 64][64];
};

/* firing of the ring point in front of / behind firing f in lane l; 0xffffffff: none */
__device__ __forceinline__ unsigned urf_front_prev(const unsigned* P, unsigned l, unsigned f)
{
    unsigned t = f >> 5;
    unsigned w = P[t * 64u + l] & ((1u << (f & 31u)) - 1u);
    while (w == 0u) {
        if (t == 0u)
            return 0xffffffffu;
        t--;
        w = P[t * 64u + l];
    }
    return t * 32u + 31u - (unsigned)__clz((int)w);
}
__device__ __forceinline__ unsigned urf_front_next(const unsigned* P, unsigned ntiles, unsigned l, unsigned f)
{
    unsigned t = f >> 5;
    unsigned w = (f & 31u) == 31u ? 0u : P[t * 64u + l] & ~((2u << (f & 31u)) - 1u);
    while (w == 0u) {
        t++;
        if (t >= ntiles)
            return 0xffffffffu;
        w = P[t * 64u + l];
    }
    return t * 32u + (unsigned)__ffs((int)w) - 1u;
}

/* part 0: everything.  Parts 1 and 2 (r6): what does not depend on the star-shaped search -- positions, ring sizes, the candidates of the
 * two detectors and their marks -- as a launch of its own on a SECOND stream, next to k_index / k_star_sort_* / k_star_walk (it waits for
 * scattered loads, they for vector issue: urf_api.hip), and the star-shaped hits, the lists' hand-over to k_beams and the overflow tables
 * behind the walk.  The counters travel from part 1 to part 2 through front_st. */
__global__ __launch_bounds__(URF_FINISH_THREADS) void k_front_finish(urf_kargs a, urf_dev_params dp, unsigned part)
{
    __shared__ urf_finish_shared S;
    extern __shared__ unsigned sh_finish[];   /* P[tiles][64] presence words | B[tiles][64] (u16) ring points of the lane in the tiles before */
    const unsigned s = blockIdx.x, tid = threadIdx.x;
    const unsigned ok = a.front_ok[s];
    if (!ok)
        return;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned ntiles = (len + URF_TILE - 1) / URF_TILE;
    const unsigned C = (unsigned)dp.p.channels, K = (unsigned)dp.p.sectors;
    const urf_scan_info in = a.info[s];
    if (in.status != URF_OK)
        return;
    unsigned* const P = sh_finish;
    uint16_t* const B = (uint16_t*)(sh_finish + a.tiles * 64u);
    urf_u2* const chunk = (urf_u2*)(sh_finish + a.tiles * 96u);   /* [2 * URF_FINISH_CHUNK] a chunk of the candidate list, by kind */
    unsigned* const st = a.front_st + (size_t)s * URF_FRONT_ST_WORDS;   /* [0..63] ncurb, [64..67] quadrants, [68] list length: part 1 -> part 2 */
    urf_u2* const pend = a.front_all + (size_t)s * a.front_cand_cap;   /* (index, flags) of what phase M has to mark; later the list of all curb points */
    const unsigned sb = urf_sbase(a, s);
    const float *gx, *gy, *gz;
    urf_front_src(a, s, off, ok, gx, gy, gz);
    auto passed = [&](unsigned idx, unsigned flag) {
        const unsigned e = atomicAdd(&S.n_pend, 1u);
        if (e < a.front_cand_cap)
            pend[e] = urf_u2{ idx, flag };
    };
    unsigned m_from = 0;   /* phase M starts here in the list */
    if (part != 2u) {
    const urf_u2* const cand = a.front_cand + (size_t)s * a.front_cand_cap;
    const unsigned nc_raw = a.front_ncand[s];
    const unsigned nc = nc_raw < a.front_cand_cap ? nc_raw : a.front_cand_cap;
    for (unsigned k = tid; k < ntiles * 64u; k += URF_FINISH_THREADS)
        P[k] = a.front_pres[(size_t)s * a.tiles * 64u + k];
    if (tid < URF_FRONT_LANES)
        S.ncurb[tid] = 0;
    if (tid < 64)
        S.ring[tid] = a.front_lane_ring[(size_t)s * 64u + tid];
    if (tid == 0) {
        st[70] = 0;
        S.q[0] = (int)urf_fbits(0.f);
        S.q[1] = (int)urf_fbits(180.f);
        S.q[2] = (int)urf_fbits(180.f);
        S.q[3] = (int)urf_fbits(360.f);
        S.n_pend = 0;
    }
    __syncthreads();
    /* positions: the tiles in as many stretches as the workgroup has waves, per lane; then the stretches' sums */
    {
        constexpr unsigned NP = URF_FINISH_THREADS / 64u;
        const unsigned l = tid & 63u, prt = tid >> 6;
        const unsigned tq = (ntiles + NP - 1u) / NP, ta = prt * tq < ntiles ? prt * tq : ntiles, tb = ta + tq < ntiles ? ta + tq : ntiles;
        unsigned run = 0;
        for (unsigned t = ta; t < tb; t++)
            run += (unsigned)__popc(P[t * 64u + l]);
        S.qsum[prt][l] = run;
        __syncthreads();
        unsigned add = 0, all = 0;
        for (unsigned p = 0; p < NP; p++) {
            add += p < prt ? S.qsum[p][l] : 0u;
            all += S.qsum[p][l];
        }
        for (unsigned t = ta; t < tb; t++) {
            B[t * 64u + l] = (uint16_t)add;
            add += (unsigned)__popc(P[t * 64u + l]);
        }
        if (prt == 0)
            S.n[l] = all;
    }
    /* ring sizes, the scan's summary, the rings' largest ranges */
    if (tid < C)
        a.ring_cnt[(size_t)s * C + tid] = 0;
    __syncthreads();
    if (tid < 64) {
        const unsigned r = S.ring[tid], n = r != 0xffffffffu ? S.n[tid] : 0u;
        unsigned tot = n;
        for (int o = 32; o > 0; o >>= 1)
            tot += __shfl_xor(tot, o);
        if (r != 0xffffffffu) {
            a.ring_cnt[(size_t)s * C + r] = n;
            unsigned long long m = 0;
            const unsigned nblk = (ntiles + a.front_tpb - 1u) / a.front_tpb;
            for (unsigned b0 = 0; b0 < nblk; b0 += 8u) {   /* (eight blocks' values in flight: one after the other this lane's chain was sixteen round trips) */
                unsigned long long v[8];
#pragma unroll
                for (unsigned b = 0; b < 8; b++)
                    v[b] = a.front_maxs[((size_t)s * a.tiles + (b0 + b < nblk ? b0 + b : b0)) * 64u + tid];
#pragma unroll
                for (unsigned b = 0; b < 8; b++)
                    m = v[b] > m ? v[b] : m;
            }
            a.maxdist[(size_t)s * C + r] = (float)__builtin_sqrt(__longlong_as_double((long long)m));
            a.vis[(size_t)s * C + r] = urf_vis{ __builtin_inff(), -__builtin_inff() };
            if (r == 10u)
                st[70] = n;
        }
        if (tid == 0)
            st[69] = tot;   /* (the summary's two counts are written behind k_index -- below -- which may still find the scan below the 30-point threshold) */
    }
    /* The candidates.  One wave-instruction costs the same with one busy lane as with sixty-four, and every kind of candidate has
     * its own expensive chain (x_zero: three f64 roots and a division; z_zero: ten differences, two roots, a division; a passed
     * test: the reference's azimuth -- a root, a division, an arc sine).  A wave that met all kinds in one iteration ran all
     * chains with a few lanes each: 1 200 instructions per iteration, 0.18 ms per 1024 scans.  So the list is worked off in
     * chunks, a chunk PARTITIONED by kind in LDS, each kind by whole waves, what passed collected and marked by whole waves:
     *   phase X   x_zero items    (XZ, EDGE_X)
     *   phase Z   z_zero items    (ZZ, EDGE_Z)
     *   phase M   the points that got a mark, and the star-shaped hits: record flag, azimuth, ring list, quadrants.
     * An item makes ONE memory round trip for its data: neighbours' firings from the presence words in LDS, then the point, the
     * neighbours and x_zero's table values requested together. */
    constexpr unsigned CH = URF_FINISH_CHUNK;
    for (unsigned c0 = 0; c0 < nc; c0 += CH) {
        const unsigned cn = nc - c0 < CH ? nc - c0 : CH;
        if (tid == 0) {
            S.nx = 0;
            S.nz = 0;
        }
        __syncthreads();
        /* partition: x_zero items from the front, z_zero items from the back (an item of a block's end may be both) */
        for (unsigned e = tid; e < cn; e += URF_FINISH_THREADS) {
            const urf_u2 cd = cand[c0 + e];
            if (cd.y & (URF_FC_XZ | URF_FC_EDGE_X))
                chunk[atomicAdd(&S.nx, 1u)] = cd;
            if (cd.y & (URF_FC_ZZ | URF_FC_EDGE_Z))
                chunk[2u * CH - 1u - atomicAdd(&S.nz, 1u)] = cd;
        }
        __syncthreads();
        const unsigned nx = S.nx, nz = S.nz;
        /* phase X: x_zero_method.cpp:30-68 marks p = j + 2 for j = p - 2 in [curbPoints, n - 1 - curbPoints] */
        if (dp.p.x_zero_method)
            for (unsigned e = tid; e < nx; e += URF_FINISH_THREADS) {
                const urf_u2 cd = chunk[e];
                const unsigned idx = cd.x, l = idx & 63u, f = idx >> 6;
                const unsigned n = S.n[l];
                const unsigned p = (unsigned)B[(f >> 5) * 64u + l] + (unsigned)__popc(P[(f >> 5) * 64u + l] & ((1u << (f & 31u)) - 1u));
                if (!(p >= 7u && p + 3u < n))
                    continue;
                unsigned fj = urf_front_prev(P, l, f);
                fj = urf_front_prev(P, l, fj);
                unsigned f3 = urf_front_next(P, ntiles, l, f);
                f3 = urf_front_next(P, ntiles, l, f3);
                f3 = urf_front_next(P, ntiles, l, f3);
                const unsigned ij = fj * 64u + l, i3 = f3 * 64u + l;
                const float pz = gz[idx], xj = gx[ij], yj = gy[ij], zj = gz[ij], x3 = gx[i3], y3 = gy[i3], z3 = gz[i3];
                const float nyj = a.newY[p - 2u], ny2 = a.newY[p], ny3 = a.newY[p + 3u];
                bool heights = true;
                if (cd.y & URF_FC_EDGE_X)   /* (the march has not looked at the heights) */
                    heights = (__builtin_fabsf(zj - pz) >= dp.p.curbHeight || __builtin_fabsf(z3 - pz) >= dp.p.curbHeight) &&
                              (double)__builtin_fabsf(zj - z3) >= 0.05;
                if (heights && urf_x_zero_angle_vals(nyj, ny2, ny3, dp.p.angleFilter1, dp.x_angle_thr, xj, yj, x3, y3, zj, pz, z3))
                    passed(idx, 2u);
            }
        /* phase Z: z_zero_method.cpp:21-72 for the centre p */
        if (dp.p.z_zero_method)
            for (unsigned e = tid; e < nz; e += URF_FINISH_THREADS) {
                const urf_u2 cd = chunk[2u * CH - 1u - e];
                const unsigned idx = cd.x, l = idx & 63u, f = idx >> 6;
                const unsigned n = S.n[l];
                const unsigned p = (unsigned)B[(f >> 5) * 64u + l] + (unsigned)__popc(P[(f >> 5) * 64u + l] & ((1u << (f & 31u)) - 1u));
                if (!(p >= 5u && p + 5u < n))
                    continue;
                unsigned im[5], ip[5];
                {
                    unsigned g = f;
#pragma unroll
                    for (unsigned k = 0; k < 5; k++) {
                        g = urf_front_prev(P, l, g);
                        im[k] = g * 64u + l;
                    }
                    g = f;
#pragma unroll
                    for (unsigned k = 0; k < 5; k++) {
                        g = urf_front_next(P, ntiles, l, g);
                        ip[k] = g * 64u + l;
                    }
                }
                const bool needz = (cd.y & URF_FC_EDGE_Z) != 0u;   /* (the march has not looked at the heights) */
                float xm[5], ym[5], zm[5] = { 0.f, 0.f, 0.f, 0.f, 0.f }, xp[5], yp[5], zp[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
                const float px = gx[idx], py = gy[idx], pz = gz[idx];
#pragma unroll
                for (unsigned k = 0; k < 5; k++) {
                    xm[k] = gx[im[k]];
                    ym[k] = gy[im[k]];
                    xp[k] = gx[ip[k]];
                    yp[k] = gy[ip[k]];
                }
                /* (what bounds this kernel is the rate at which a CU takes scattered 4-byte loads -- one cache line per lane and
                 * instruction: the heights are only asked for by a wave that holds such an item) */
                if (__ballot(needz) != 0ull) {
#pragma unroll
                    for (unsigned k = 0; k < 5; k++) {
                        zm[k] = gz[needz ? im[k] : idx];
                        zp[k] = gz[needz ? ip[k] : idx];
                    }
                }
                bool heights = true;
                if (needz) {
                    const float azp = __builtin_fabsf(pz);
                    float max1 = azp, max2 = azp;
#pragma unroll
                    for (unsigned k = 0; k < 5; k++) {
                        const float za = __builtin_fabsf(zm[k]), zb = __builtin_fabsf(zp[k]);
                        max1 = za > max1 ? za : max1;
                        max2 = zb > max2 ? zb : max2;
                    }
                    heights = (max1 - azp >= dp.p.curbHeight || max2 - azp >= dp.p.curbHeight) && (double)__builtin_fabsf(max1 - max2) >= 0.05;
                }
                if (heights) {
                    auto xy = [&](int pos, float& xx, float& yy) {   /* pos: ring position; the centre's is p */
                        const int rel = pos - (int)p;
                        xx = rel < 0 ? xm[-rel - 1] : xp[rel - 1];
                        yy = rel < 0 ? ym[-rel - 1] : yp[rel - 1];
                    };
                    if (urf_z_zero_angle(dp.inv_cp, dp.p.angleFilter2, dp.z_angle_thr, xy, (int)p, 5, px, py))
                        passed(idx, 4u);
                }
            }
        __syncthreads();   /* the chunk's buffer is free for the next one */
    }
    } else {
        /* part 2: the counters of part 1 */
        if (tid < URF_FRONT_LANES) {
            S.ncurb[tid] = st[tid];
            S.ring[tid] = a.front_lane_ring[(size_t)s * 64u + tid];
        }
        if (tid < 4)
            S.q[tid] = (int)st[64u + tid];
        if (tid == 0)
            S.n_pend = st[68];
        m_from = st[68];
        __syncthreads();
    }
    /* lidar_segmentation.cpp:235-242: the star-shaped hits (the walk reported them as input indices; -1: none or on no ring) */
    if (part != 1u && dp.p.star_shaped_method)
        for (unsigned k = tid; k < K; k += URF_FINISH_THREADS) {
            const int h = a.star_hit[(size_t)s * K + k];
            if (h >= 0)
                passed((unsigned)h, 1u);
        }
    __syncthreads();
    /* phase M.  A ring point that has a detector's mark: its record's flag (whoever sets the first one lists the point), the
     * reference's azimuth, its ring's list, ring 1's quadrants (urf_ring_point: lidar_segmentation.cpp:245-269, blind_spots.cpp:
     * 19-56).  The atomic is on its way while the azimuth is worked out.  The list of all curb points (the rings whose own list
     * overflows) takes the places of the entries already read: entry e is rewritten by the thread that read it. */
    const unsigned n_pend = S.n_pend < a.front_cand_cap ? S.n_pend : a.front_cand_cap;
    for (unsigned e = m_from + tid; e < n_pend; e += URF_FINISH_THREADS) {
        const urf_u2 pd = pend[e];
        const unsigned idx = pd.x, r = S.ring[idx & 63u];
        const float px = gx[idx], py = gy[idx];
        const unsigned old = atomicOr(&a.rec[sb + idx], pd.y << URF_REC_FLAG_SHIFT);
        float d2;
        const float az = urf_azimuth(px, py, &d2);
        urf_u2 out = urf_u2{ 0u, 0xffffffffu };   /* (ring 0xffffffff: not a list entry) */
        if (((old >> URF_REC_FLAG_SHIFT) & 7u) == 0u) {   /* otherwise: already a curb point, listed by whoever marked it first */
            const unsigned ec = atomicAdd(&S.ncurb[r], 1u);
            if (ec < URF_CURB_LIST)
                a.curb_az[((size_t)s * C + r) * URF_CURB_LIST + ec] = az;
            out = urf_u2{ __float_as_uint(az), r };
            if (r == 1u && dp.p.blind_spots) {
                const int ab = (int)urf_fbits(az);
                if (az >= 0.f && az < 90.f)
                    atomicMax(&S.q[0], ab);
                else if (az >= 90.f && az < 180.f)
                    atomicMin(&S.q[1], ab);
                else if (az >= 180.f && az < 270.f)
                    atomicMax(&S.q[2], ab);
                else if (az < 360.f)
                    atomicMin(&S.q[3], ab);
            }
        }
        pend[e] = out;
    }
    __syncthreads();
    if (part == 1u) {   /* (uniform) the counters for part 2 */
        if (tid < URF_FRONT_LANES)
            st[tid] = S.ncurb[tid];
        if (tid < 4)
            st[64u + tid] = (unsigned)S.q[tid];
        if (tid == 0)
            st[68] = n_pend;
        return;
    }
    /* the scan's summary (lidar_segmentation.cpp:605-608: road_probably = every point of sorted ring 10) */
    if (tid == 0) {
        a.info[s].n_ring_pts = st[69];
        a.info[s].n_ring10 = in.n_rings > 10 ? st[70] : 0u;
    }
    /* what k_beams reads (k_ring's epilogue) */
    if (tid < 4 && dp.p.blind_spots && in.n_rings > 1)
        a.quad[(size_t)s * 4 + tid] = __uint_as_float((unsigned)S.q[tid]);
    for (unsigned r = tid; r < in.n_rings; r += URF_FINISH_THREADS)
        a.curb_cnt[(size_t)s * C + r] = S.ncurb[r] <= URF_CURB_LIST ? S.ncurb[r] : URF_CURB_DENSE;
    /* a ring with more curb points than its list holds (rough ground): the per-degree tables instead, from the scan's
     * list of all curb points -- sufmin[i] = smallest curb azimuth >= i, premax[i] = largest <= i, NaN = none.  (The presence
     * words are no longer needed: their memory holds the two tables of the ring at hand.) */
    const unsigned n_all = n_pend;
    int* const cmin = (int*)sh_finish;
    int* const cmax = cmin + URF_DEG_CELLS;
    for (unsigned r = 0; r < in.n_rings; r++) {
        if (S.ncurb[r] <= URF_CURB_LIST)
            continue;   /* (uniform) */
        __syncthreads();
        for (unsigned i = tid; i < URF_DEG_CELLS; i += URF_FINISH_THREADS) {
            cmin[i] = URF_INT_NONE_MIN;
            cmax[i] = -1;
        }
        __syncthreads();
        for (unsigned e = tid; e < n_all; e += URF_FINISH_THREADS) {
            const urf_u2 v = a.front_all[(size_t)s * a.front_cand_cap + e];
            if (v.y != r)
                continue;
            const float az = __uint_as_float(v.x);
            int cl = (int)__builtin_floorf(az), ch = (int)__builtin_ceilf(az);
            cl = cl < 0 ? 0 : (cl > 360 ? 360 : cl);
            ch = ch < 0 ? 0 : (ch > 360 ? 360 : ch);
            atomicMin(&cmin[cl], (int)v.x);
            atomicMax(&cmax[ch], (int)v.x);
        }
        __syncthreads();
        if (tid < 64) {   /* one wave: running maximum upwards, running minimum downwards */
            float* sm = a.sufmin + ((size_t)s * C + r) * URF_DEG_CELLS;
            float* pm = a.premax + ((size_t)s * C + r) * URF_DEG_CELLS;
            /* six cells per lane, one scan across the wave each way */
            unsigned up[6], dn[6];
#pragma unroll
            for (unsigned e = 0; e < 6; e++) {
                const unsigned i = 6u * tid + e;
                up[e] = i < URF_DEG_CELLS ? (unsigned)(cmax[i] + 1) : 0u;
                dn[e] = i < URF_DEG_CELLS ? ~(unsigned)cmin[URF_DEG_CELLS - 1 - i] : 0u;
                if (e) {
                    up[e] = up[e] > up[e - 1] ? up[e] : up[e - 1];
                    dn[e] = dn[e] > dn[e - 1] ? dn[e] : dn[e - 1];
                }
            }
            const unsigned iu = urf_wave_scan_max(up[5]), id = urf_wave_scan_max(dn[5]);
            unsigned pu = (unsigned)__shfl_up((int)iu, 1), pd = (unsigned)__shfl_up((int)id, 1);
            if (tid == 0)
                pu = pd = 0;
#pragma unroll
            for (unsigned e = 0; e < 6; e++) {
                const unsigned i = 6u * tid + e;
                if (i < URF_DEG_CELLS) {
                    const unsigned u = up[e] > pu ? up[e] : pu, d = dn[e] > pd ? dn[e] : pd;
                    pm[i] = u == 0 ? __builtin_nanf("") : __uint_as_float(u - 1u);
                    sm[URF_DEG_CELLS - 1 - i] = (d == 0x80000000u || d == 0u) ? __builtin_nanf("") : __uint_as_float(~d);
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* k_label_front                                                               */
/* ------------------------------------------------------------------------- */
/* k_label for a scan of the fused front end: the records are in input order, eight consecutive points per thread,
 * the label bytes leave as one 8-byte store.  Same decisions as k_label (urf_road_test on the record's quantised
 * azimuth, the exact azimuth where that leaves a decision open). */
__global__ __launch_bounds__(URF_LABEL_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_label_front(urf_kargs a, urf_dev_params dp)
{
    __shared__ unsigned cnt_road, cnt_curb, n_unsure;
    __shared__ unsigned un_idx[URF_LABEL_UNSURE], un_ring[URF_LABEL_UNSURE];
    __shared__ __attribute__((aligned(8))) uint8_t lab_t[64][40];   /* (row-major scans) [laser][firing of the tile] */
    unsigned s = blockIdx.y, t = blockIdx.x;
    {   /* the tiles of one scan on one XCD (k_label: the scan's window table is fetched by one L2) */
        const unsigned T = gridDim.x, lin = blockIdx.y * T + blockIdx.x;
        const unsigned grp = lin / (8u * T), r = lin - grp * (8u * T);
        if ((grp + 1u) * 8u <= gridDim.y) {
            s = grp * 8u + (r & 7u);
            t = r >> 3;
        }
    }
    const unsigned tid = threadIdx.x;
    const unsigned ok = a.front_ok[s];
    if (!ok)
        return;
    unsigned off, len;
    urf_scan_range(a, s, off, len);
    const unsigned tbase = t * URF_TILE;
    if (tbase >= len)
        return;
    const bool rows = ok == URF_FRONT_ROWS;   /* (uniform) the labels go where the points came from: row l, column f */
    const unsigned F = len >> 6;
    const float *gx, *gy, *gz;
    urf_front_src(a, s, off, ok, gx, gy, gz);
    const unsigned C = (unsigned)dp.p.channels;
    const size_t row = (size_t)s * a.tiles + t;
    const unsigned sb = urf_sbase(a, s);
    const urf_scan_info in = a.info[s];
    const unsigned troi = a.tile_roi[row];
    const unsigned i0 = tbase + tid * 8u;   /* this thread's eight points: one eighth of a firing */
    /* (the firings behind the scan's last one were never visited: their words hold whatever an earlier call left) */
    const unsigned in_scan = i0 + 8u <= len ? 0xffu : (i0 < len ? (1u << (len - i0)) - 1u : 0u);
    const unsigned bits = ((const uint8_t*)(a.roi_bits + row * URF_FRONT_STEPS))[tid] & in_scan;
    uint8_t* const out = a.labels + off + i0;
    const bool whole = tbase + URF_TILE <= len && ((uintptr_t)(a.labels + off + tbase) & 7u) == 0;   /* (uniform) */
    /* row-major: the tile's 32 firings x 64 lasers through LDS, then 8 columns of a row per thread (32 bytes per row and tile) */
    auto store_rows = [&](const unsigned (&lb)[8]) {
        const unsigned stp = tid >> 3, l0 = (tid & 7u) * 8u;
#pragma unroll
        for (unsigned e = 0; e < 8; e++)
            lab_t[l0 + e][stp] = (uint8_t)lb[e];
        __syncthreads();
        const unsigned l = tid >> 2, c0 = (tid & 3u) * 8u, f0 = t * URF_FRONT_STEPS + c0;
        uint8_t* const o = a.labels + off + (size_t)l * F + f0;
        const uint8_t* const src = &lab_t[l][c0];
        if (f0 + 8u <= F && ((uintptr_t)o & 7u) == 0) {
            *(uint2*)o = *(const uint2*)src;
        } else {
            for (unsigned e = 0; e < 8; e++)
                if (f0 + e < F)
                    o[e] = src[e];
        }
    };
    if (in.status != URF_OK || troi == 0) {
        if (rows) {
            const unsigned zero[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
            store_rows(zero);
        } else if (whole) {
            *(uint2*)out = make_uint2(0u, 0u);
        } else {
            for (unsigned e = 0; e < 8; e++)
                if (i0 + e < len)
                    out[e] = 0;
        }
        return;
    }
    if (tid == 0) {
        cnt_road = 0;
        cnt_curb = 0;
        n_unsure = 0;
    }
    __syncthreads();
    const urf_win* win = a.win + (size_t)s * C * URF_DEG_CELLS;
    unsigned rec[8];
    if (bits) {   /* (the records of a firing without a point in the region of interest were never written) */
        const uint4 r0 = *(const uint4*)(a.rec + sb + i0), r1 = *(const uint4*)(a.rec + sb + i0 + 4);
        rec[0] = r0.x; rec[1] = r0.y; rec[2] = r0.z; rec[3] = r0.w;
        rec[4] = r1.x; rec[5] = r1.y; rec[6] = r1.z; rec[7] = r1.w;
    } else {
#pragma unroll
        for (unsigned e = 0; e < 8; e++)
            rec[e] = URF_FRONT_RING_NONE;
    }
    unsigned lab[8];
    unsigned my_road = 0, my_curb = 0;
    /* (four points at a time: the window ends of four are requested before any of them is looked at) */
#pragma unroll
    for (unsigned h = 0; h < 8; h += 4) {
    float whi[4], wlo[4];
#pragma unroll
    for (unsigned q = 0; q < 4; q++) {
        const unsigned e = h + q;
        const bool on = ((bits >> e) & 1u) && (rec[e] & 0x7fu) != URF_FRONT_RING_NONE;
        const unsigned c = on ? (rec[e] & 0x7fu) : 0u;
        const float az = urf_az_decode(rec[e] >> URF_REC_AZ_SHIFT);
        const bool num = az == az;
        int cf = num ? (int)__builtin_floorf(az) : 0, cb = num ? (int)__builtin_ceilf(az) : 0;
        cf = cf < 0 ? 0 : (cf > 360 ? 360 : cf);
        cb = cb < 0 ? 0 : (cb > 360 ? 360 : cb);
        whi[q] = win[c * URF_DEG_CELLS + cf].hi;
        wlo[q] = win[c * URF_DEG_CELLS + cb].lo;
    }
#pragma unroll
    for (unsigned q = 0; q < 4; q++) {
        const unsigned e = h + q;
        const bool roi = (bits >> e) & 1u;
        const bool on = roi && (rec[e] & 0x7fu) != URF_FRONT_RING_NONE;
        const unsigned c = rec[e] & 0x7fu;
        const bool curb = on && ((rec[e] >> URF_REC_FLAG_SHIFT) & 7u) != 0;
        const float az = urf_az_decode(rec[e] >> URF_REC_AZ_SHIFT), eps = urf_fast_az_eps(az) + URF_REC_AZ_QERR;
        const float fl = __builtin_floorf(az);
        bool road = az <= whi[q] || az >= wlo[q];
        const bool unsure = az < 0.0f || az - fl <= eps || (fl + 1.0f) - az <= eps || __builtin_fabsf(az - whi[q]) <= eps ||
                            __builtin_fabsf(az - wlo[q]) <= eps;
        /* (the label first, with a point whose decision is open as "not road"; THEN the branch for such a point: the lane masks above
         * are dead by then -- they used to live across it, and the compiler parked fourteen scalar registers per point in a vector
         * register's lanes, v_writelane by v_writelane) */
        const bool uns = unsure && on && !curb;
        road = road && on && !curb && !uns;
        lab[e] = !roi ? 0u
                      : (URF_FLAG_ROI | (on ? URF_FLAG_RING | (c == 10 ? URF_FLAG_RING10 : 0) | (curb ? URF_LABEL_CURB : 0) | (road ? URF_LABEL_ROAD : 0) : 0u));
        my_curb += curb ? 1u : 0u;
        my_road += road ? 1u : 0u;
        if (uns) {
            const unsigned u = atomicAdd(&n_unsure, 1u);
            if (u < URF_LABEL_UNSURE) {
                un_idx[u] = i0 + e;
                un_ring[u] = c;   /* corrected below */
            } else {   /* list full (pathological input) */
                float d2;
                bool dummy;
                const float xaz = urf_azimuth(gx[i0 + e], gy[i0 + e], &d2);
                if (urf_road_test(win + c * URF_DEG_CELLS, xaz, 0.0f, dummy)) {
                    lab[e] |= URF_LABEL_ROAD;
                    my_road++;
                }
            }
        }
    }
    }
    if (rows) {
        store_rows(lab);
    } else if (whole) {
        *(uint2*)out = make_uint2(lab[0] | lab[1] << 8 | lab[2] << 16 | lab[3] << 24, lab[4] | lab[5] << 8 | lab[6] << 16 | lab[7] << 24);
    } else {
        for (unsigned e = 0; e < 8; e++)
            if (i0 + e < len)
                out[e] = (uint8_t)lab[e];
    }
    __syncthreads();   /* the tile's stores come first, the corrections second */
    const unsigned nu = n_unsure < URF_LABEL_UNSURE ? n_unsure : URF_LABEL_UNSURE;
    if (tid < nu) {
        const unsigned i = un_idx[tid], c = un_ring[tid];
        bool dummy;
        float d2;
        const float az = urf_azimuth(gx[i], gy[i], &d2);
        if (urf_road_test(win + c * URF_DEG_CELLS, az, 0.0f, dummy)) {
            a.labels[off + (rows ? (size_t)(i & 63u) * F + (i >> 6) : (size_t)i)] = URF_FLAG_ROI | URF_FLAG_RING | (c == 10 ? URF_FLAG_RING10 : 0) | URF_LABEL_ROAD;
            my_road++;
        }
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

#endif /* URF_FRONT_HPP */
