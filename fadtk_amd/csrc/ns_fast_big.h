// The iteration products of ns_fast.h for a BATCH of problems (per-song scores, frechet.hip: fast_songs): the same split-float16
// arithmetic and the same outputs -- split planes of the result in both orientations, digit planes of the new Y, residual partials,
// the convergence check riding on the update launch -- on 128 x 128 workgroup tiles staged through LDS.
//
// ns_fast.h's kernels are built for ONE problem: 32 x 32 tiles, the k range split over eight waves, operands straight from L2 into
// registers -- the shortest possible dependent chain, and 128-192 KB of operand traffic per 2 MFLOP tile.  Batched over 32 songs of
// D = 768 that design moved 10.8 GB through the L2s per iteration and ran at 14 % of the float16 MFMA rate (profiles/r03h_c5_*:
// T + U = 0.75 ms per iteration, eight to eleven iterations per song).  Here a workgroup (4 waves, 2 x 2, 64 x 64 per wave) owns a
// 128 x 128 tile and walks the whole k range: per k-step of 16 it needs 16 operand pieces of 1 KiB (four row blocks of A and four
// column blocks of B, hi and lo plane) -- which the fragment-major layout stores exactly as the MFMA reads them, so one
// global_load_lds_dwordx4 per piece moves it into LDS verbatim (no swizzle, no transpose read: the 64 lanes' 16-byte operands back to
// back) and one ds_read_b128 per lane brings it to the MFMA.  A ring of three stages of one k-step; 12 MFMAs (2 x 2 blocks x hi hi,
// hi lo, lo hi) against 8 ds_read_b128 and 4 LDS-DMA instructions per wave and k-step; operand traffic per tile is a quarter of
// the small-tile kernels'.  48 KiB of LDS: three workgroups per CU.  Launches that would leave CUs with ONE such workgroup take
// 128 x 64 tiles instead (nsf_big<MODE, 1>, below); which (problem, tile) a workgroup takes: big_slots.h.
//
// The epilogue is ns_fast.h's, a 32 x 32 block at a time: each wave parks a block of its 64 x 64 result in its own (now free) part
// of the ring and runs the store_tile tasks over it in a few passes.
#pragma once
#include "ns_fast.h"
#include "big_slots.h"

namespace fad {
namespace nsf {

// Ring depth (stages of one k-step = 16 KiB) and with it the workgroups per CU.  Measured at 32 songs x [1500 x 768], T / U launch:
// 4 stages x 2 workgroups 164 / 283 us, 8 stages x 1 workgroup 175 / 318 us, 3 stages x 3 workgroups 142 / 243 us: what hides the
// operand latency here is another workgroup's epilogue and loop, not a deeper ring.
#ifndef FAD_BIG_STAGES
#define FAD_BIG_STAGES 3
#endif
constexpr int kBigStages = FAD_BIG_STAGES;
constexpr int kBigStage = 1024;                                   // uint4 per stage: A pieces [4 rb][2 planes][64 lanes], then B pieces
constexpr size_t kBigLds = (size_t)kBigStages * kBigStage * 16 + 256;   // + the reduction scratch

// NJ = 2: the 128 x 128 tile.  NJ = 1: a 128 x 64 tile (a wave 64 x 32) for launches that would put fewer than two workgroups on a
// CU (big_nj, big_slots.h): a lone workgroup's four waves are one to a SIMD and each pays its LDS-DMA issue, its LDS reads and its
// MFMAs one after the other -- 544 ns per k-step at 16 pairs of D = 512 where the MFMAs need 160 and the L2 -> LDS path 138
// (scripts/probes/r6_l2_paths.hip: 119-131 GB/s per CU from L2) -- while two workgroups per CU fill each other's gaps (SP_U, always two
// products: 182 ns).  Half the tile costs 12 instead of 16 pieces for half the MFMAs (1.5 x the operand traffic per flop).
template <int MODE, int NJ = 2>
__global__ __launch_bounds__(256, (FAD_BIG_STAGES <= 3) ? 3 : ((FAD_BIG_STAGES <= 4) ? 2 : 1)) void nsf_big(SplitArgs g) {
    static_assert(MODE == SP_T || MODE == SP_U || MODE == SP_FIRST, "SP_FIRST: iteration 0 (scale from nsf_i8<A>'s statistics, Y1, Z1)");
    extern __shared__ __attribute__((aligned(16))) uint4 ring[];
    double* red = reinterpret_cast<double*>(ring + kBigStages * kBigStage);
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    // ---- which (song, product, tile): a 1-D grid (big_slot above: one XCD per song, the last few songs cut over all eight); after the
    // product workgroups come the check workgroups of SP_U, one per song.
    constexpr int ZP = (MODE == SP_U) ? 2 : 1;
    const int t = d >> 7, tx_n = (NJ == 2) ? t : 2 * t, tt = t * tx_n;   // tiles of a product: t rows of 128 x tx_n columns of 64 NJ
    const int L = blockIdx.x;
    const int nprod = big_grid(g.nprob, tt * ZP);
    if constexpr (MODE == SP_U) {
        if (L >= nprod) { nsf_check<256>(g, (int64_t)(L - nprod) * g.pstride, red); return; }
    }
    const BigSlot slot = big_slot(L, g.nprob, tt * ZP);
    if (!slot.live) return;
    const int song = slot.song, zi = slot.item / tt, tile = slot.item - zi * tt;
    const int TY = tile / tx_n, TX = tile - TY * tx_n;
    const int64_t po = (int64_t)song * g.pstride;
    const MatHdr* hB = adv(g.hB, po);
    const MatHdr* hA = adv(g.hA, (int64_t)song * g.astride);        // (astride = 0: the songs' shared baseline; pairs: per problem)
    if (hdr_bad(hA, hB, g.gen)) return;
    double inv_c = 0.0, inv_cn = 0.0;
    float m1 = 1.5f, m3 = 0.5f;                                      // SP_FIRST: 1.5 mu_0, 0.5 mu_0^3
    if constexpr (MODE == SP_FIRST) {
        // ---- the scale, as nsf_split<FIRST> derives it (every workgroup, identically, from nsf_i8<A>'s 32 x 32 tile statistics);
        // the ring is not in use yet: it holds the per-tile maxima meanwhile
        const int nb = d >> 5;
        const double* scal = adv(g.statsA, po);
        double* tmax = reinterpret_cast<double*>(ring);              // [2][nb * nb]
        double v2[2] = {0.0, 0.0};
        for (int k = tid; k < nb * nb; k += 256) {
            const double2 r0 = *reinterpret_cast<const double2*>(scal + kTileStats * k), r1 = *reinterpret_cast<const double2*>(scal + kTileStats * k + 2);
            v2[0] += r0.x; v2[1] += r0.y;
            tmax[k] = (r1.x == r1.x) ? r1.x : 1e300; tmax[nb * nb + k] = (r1.y == r1.y) ? r1.y : 1e300;
        }
        __syncthreads();
        double bnd = 0.0;
        if (tid < 2 * nb) {
            const int line = tid % nb; const bool col = tid >= nb;
            for (int q = 0; q < nb; ++q) bnd += col ? tmax[nb * nb + q * nb + line] : tmax[line * nb + q];
        }
        const double inf_b = wg8_max<4>((tid < nb) ? bnd : 0.0, red);
        const double one_b = wg8_max<4>((tid >= nb && tid < 2 * nb) ? bnd : 0.0, red);
        wg8_sum<2, 4>(v2, red);
        const double fro2 = v2[0], trA = v2[1];
        double u = sqrt(fro2);
        if (inf_b < u) u = inf_b;
        if (one_b < u) u = one_b;
        double c = u / 2.9;
        const double wmean = (trA > 0.0) ? fro2 / trA : 0.0;
        if (wmean > c && wmean <= u) c = wmean;
        NsState* const st_p = adv(g.st, po);
        const double mean_term = st_p->mean_term, tr1 = hA->tr, tr2 = hB->tr;
        const bool bad = !(fro2 == fro2) || isinf(fro2) || !(trA == trA) || isinf(trA) || !(mean_term == mean_term) || isinf(mean_term);
        const bool zero = !bad && !(c > 0.0);
        // Scaled steps (g.scaled): a spectrum that is not flat -- participation ratio below 0.8 d: songs of a few D frames, a baseline
        // whose variances differ by dimension -- starts from c = u >= rho(A) (every x = sqrt(lambda / c) in (0, 1]) and lifts the lower
        // end by mu_k per step (ns_check.h); flat spectra keep the start near 1 above, which needs no lifting.
        // Which products the chain serves at all: as nsf_split<SP_FIRST> (ns_fast.h) -- songs (g.lp_wide = 0) keep the participation-ratio
        // rule, pairs (g.lp_wide = 1) go by the x_min estimate.
        const bool flat = trA * trA >= 0.25 * (double)d * fro2;
        const bool want_scaled = g.scaled && !bad && !zero && u > 0.0 && trA * trA < 0.8 * (double)d * fro2;
        double l0 = 1.0;
        if (want_scaled) { l0 = ns_l0_from_participation((float)(trA * trA / fro2), d) * g.l0_scale; if (l0 > 0.5) l0 = 0.5; }      // (||A||_F^2 >= tr A^2: the estimate errs low)
        const bool hopeless = !bad && !zero && (c < 0.0078125 || (g.lp_wide ? (want_scaled && l0 < g.l0_min) || (!want_scaled && !flat) : !flat));
        const bool scaled = want_scaled && !hopeless;
        double mu0 = 1.0;
        if (scaled) {
            c = u;
            double l = l0;
            mu0 = ns_step_scale(l);
            if (tile == 0 && tid == 0) { double ln = l; st_p->mu[0] = mu0; st_p->mu[1] = ns_step_scale(ln); st_p->l_cur = l; }
        } else if (g.scaled && tile == 0 && tid == 0) {
            st_p->mu[0] = 1.0; st_p->mu[1] = 1.0; st_p->l_cur = 1.0;
        }
        if (tile == 0 && tid == 0) {
            NsState* st = st_p; Ns32State* s32 = adv(g.s32, po);
            st->c = zero ? 1.0 : c * hdr_inv_s12(hA, hB);
            st->tr1 = tr1; st->tr2 = tr2;
            st->res_last = 0.0; st->tr_last = 0.0; st->res_min = 1e300; st->tr_safe = 0.0; st->has_safe = 0;
            st->final_iter = zero ? 0 : -1; st->conv = zero ? 1 : 0;
            st->nonfinite = bad ? 1 : 0; st->done = (bad || zero) ? 1 : 0; st->finished = (bad || zero) ? 1 : 0;
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->final_iter = -1; s32->failed = 0;
            s32->upd_skip[0] = 0; s32->upd_skip[1] = 0; s32->skip_corr = 1; s32->decided_at = -1; s32->strict = 0; s32->grew = 0;
            s32->res[0] = 1e300;
            if (bad || zero || hopeless) { s32->done = 1; s32->finished = 1; s32->failed = 1; s32->upd_skip[0] = 1; s32->upd_skip[1] = 1; }
        }
        if (bad || zero || hopeless) return;
        inv_cn = 1.0 / c;
        inv_c = inv_cn / hdr_inv_s12(hA, hB);
        m1 = (float)(1.5 * mu0); m3 = (float)(0.5 * mu0 * mu0 * mu0);
        __syncthreads();                                             // the maxima have been read: the ring may fill
    } else {
        if (g.skip && *adv(g.skip, po) != 0) return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
#ifdef FAD_BIG_ABL_NOLOOP                                          // ablation (scripts/probe_chain8.py): no k loop, the epilogue alone
    const int nks = 0;
#else
    const int nks = d >> 4;
#endif

    // ---- loads: the pieces of a stage are A's four row blocks (hi, lo plane: 2 KiB in a row) and then B's 2 NJ column blocks, 16 or 12
    // of them; wave w moves pieces PP w .. PP w + PP - 1 (NJ = 2: waves 0, 1 the A side, waves 2, 3 the B side).  Wave-uniform 64-bit
    // base in SGPRs + the lane's 16-byte offset (the form that overlaps with the MFMAs, moments_kernels.h: issue_fast).
    const SplitMat Am = adv(g.A[zi], po), Bm = adv(g.B[zi], po);
    constexpr int PP = 2 + NJ;                                       // pieces per wave and stage
    // (the narrow tile's stages are 12 KiB and FOUR of them would fit where the wide tile has three: measured, no gain -- r06p)
    constexpr int STG = kBigStage, NST = kBigStages;
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
    // (all tiles of a product walk k from 0 in step -- letting tile (TY, TX) start (TY t + TX) s k-steps in, s = 1 .. 13, so that they do
    //  not ask for the same pieces at the same moment, measured 1-14 % SLOWER: the six tiles that share a strip want it together)
    auto issue = [&](int ks) {
        const int kk = ks;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int piece = wave * PP + q;                         // (wave-uniform)
            const bool bside = piece >= 8;
            const int blk = (bside ? piece - 8 : piece) >> 1, plane = piece & 1;
            const uint4* src_mat = bside ? Bm.at : Am.a;
            const int blk_global = bside ? 2 * NJ * TX + blk : 4 * TY + blk;
            const uint64_t sb = (uint64_t)(src_mat + fa_idx(blk_global, kk, plane, 0, d));
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            const uint32_t dst = ring_lds + (uint32_t)(((ks % NST) * STG + piece * 64) * 16);
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(dst);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    f32x16 acc0[2][NJ], acc1[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0[i][j][q] = 0.f; acc1[i][j][q] = 0.f; }

    for (int ks = 0; ks < NST - 1 && ks < nks; ++ks) issue(ks);
    for (int ks = 0; ks < nks; ++ks) {
        // stage ks must have landed; up to two younger stages stay in flight (the count has to be an immediate)
        const int ahead = (nks - 1 - ks < NST - 2) ? (nks - 1 - ks) : (NST - 2);
        switch (ahead * PP) {
            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();                          // stage ks is in LDS for every wave; the slot of stage ks - 1 is free
        if (ks + NST - 1 < nks) issue(ks + NST - 1);
        const f16x8* st = reinterpret_cast<const f16x8*>(ring + (ks % NST) * STG) + lane;
        f16x8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = st[((2 * wr + i) * 2) * 64]; al[i] = st[((2 * wr + i) * 2 + 1) * 64]; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { bh[j] = st[(8 + (NJ * wc + j) * 2) * 64]; bl[j] = st[(8 + (NJ * wc + j) * 2 + 1) * 64]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
            }
    }
    __syncthreads();                                           // every wave has left the ring: it is scratch now
#ifdef FAD_BIG_ABL_NOEPI                                           // ablation: the k loop alone
    if (acc0[0][0][0] != 12345.678f) return;
#endif

    // ---- epilogue, per wave: its 2 NJ 32 x 32 blocks one after the other through a [32][33] float area of its own
    float* fin = reinterpret_cast<float*>(ring) + wave * (32 * 33 + 32);
    const int kg = lane >> 5, r = lane & 31;
    float alpha = (MODE == SP_T) ? g.alpha : 1.f, beta = (MODE == SP_T) ? g.beta_eye : 0.f, gamma = g.gamma;
    if (MODE == SP_T && g.scaled) {                                  // this problem's step scale (set by FIRST / the previous check)
        const double m = adv(g.st, po)->mu[g.k];
        alpha = (float)(-0.5 * m * m * m); beta = (float)(1.5 * m); gamma = beta + alpha;
    }
    const SplitMat Cm = adv(g.C[zi], po);
    const bool with_digits = (MODE == SP_U) && zi == 0 && g.Cdig[0];
    uint4* dig = with_digits ? adv(g.Cdig[0], po) : nullptr;
    uint4* dig_t = with_digits ? adv(g.Cdig_t[0], po) : nullptr;
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int by = 4 * TY + 2 * wr + i, bx = 2 * NJ * TX + NJ * wc + j;
            if constexpr (MODE == SP_FIRST) {
                // Y0 = A/c; the product is P P = (c Y0)^2 in normalised units: Y1 = Y0 T0 = 1.5 Y0 - 0.5 Y0^2, Z1 = T0 = 1.5 I - 0.5 Y0
                const double* A64 = adv(g.A64, po) + (int64_t)(32 * by + 4 * kg) * d + 32 * bx + r;
                float z1[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int q = (reg & 3) + 8 * (reg >> 2);
                    const float y0 = (float)(A64[(int64_t)q * d] * inv_c);
                    const float y2 = (float)((double)(acc0[i][j][reg] + acc1[i][j][reg] * kLoInv) * (inv_cn * inv_cn));
                    fin[(q + 4 * kg) * 33 + r] = m1 * y0 - m3 * y2;
                    z1[reg] = ((by == bx && q + 4 * kg == r) ? m1 : 0.f) - m3 * y0;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) store_tile_planes(fin, Cm, by, bx, d, pass * 64 + lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) fin[((reg & 3) + 8 * (reg >> 2) + 4 * kg) * 33 + r] = z1[reg];
                __builtin_amdgcn_wave_barrier();
                const SplitMat Zm1 = adv(g.C[1], po);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) store_tile_planes(fin, Zm1, by, bx, d, pass * 64 + lane);
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            float ssf = 0.f;                                 // (the block's 16 squares in float32: a residual ESTIMATE, and 40 bytes of scratch less)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int q = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                const bool dg = (by == bx) && (q == r);
                const float v = alpha * (acc0[i][j][reg] + acc1[i][j][reg] * kLoInv) + (dg ? beta : 0.f);
                if constexpr (MODE == SP_T) { const float e = v - (dg ? gamma : 0.f); ssf = __builtin_fmaf(e, e, ssf); }
                fin[q * 33 + r] = v;
            }
            ss += (double)ssf;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) store_tile_planes(fin, Cm, by, bx, d, pass * 64 + lane);
            if (with_digits) {
#pragma unroll 2
                for (int pass = 0; pass < 8; ++pass) store_tile_digits(fin, dig, dig_t, by, bx, d, pass * 64 + lane);
            }
            __builtin_amdgcn_wave_barrier();
        }
    if constexpr (MODE == SP_T) {
        double s1[1] = {ss};
        wg8_sum<1, 4>(s1, red);
        if (tid == 0) adv(g.partials, po)[tile] = s1[0];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The two exact products (nsf_i8<A>, nsf_i8<G>) for a batch, D >= 256: workgroup tile 128 x 64, eight waves, each wave ONE 32 x 32
// block over the whole k range -- no k-split, no partial tiles to sum -- digit pieces staged through LDS by LDS-DMA (per k-step of
// 32: four row blocks and two column blocks x six digit planes = 36 pieces of 1 KiB, two stages), one XCD per song.  On 32 x 32
// workgroup tiles a batch of 32 songs of D = 768 pulled 295 KB of digit planes per tile through the L2s (5.4 GB per launch,
// 0.76 / 0.95 ms: profiles/r03o_c5_kernel_stats.csv); here a block costs 28 KB.
constexpr int kI8BigStage = 36 * 64;                               // uint4 per stage
// Ring depth: two stages (two workgroups per CU).  Measured with four (counted waits, two younger stages in flight, one workgroup per
// CU): the same 35-37 us per launch for eight pairs of D = 512 and the same 3.9 ms for 32 songs of [1500 x 768] -- the kernel is bound
// by its 30 / 26 int8 MFMAs per wave and k-step (12.9 us of a launch at the dense int8 rate) and its float64 epilogue, not by the loads.
#ifndef FAD_I8_BIG_STAGES
#define FAD_I8_BIG_STAGES 2
#endif
constexpr int kI8BigStages = FAD_I8_BIG_STAGES;
constexpr size_t kI8BigLds = (size_t)kI8BigStages * kI8BigStage * 16 + 256;

// WITHR (I8_G for pairs on the wide chain): the block's residual R stays in registers and leaves as kVerScale R planes (SP_V2 reads them);
// the songs' launches use the instantiation without it (holding R across the epilogue cost the songs' correction 10 %: r05d)
template <int MODE, bool WITHR = false>
__global__ __launch_bounds__(512) void nsf_i8_big(I8Args g, int nprob) {
    extern __shared__ __attribute__((aligned(16))) uint4 ring[];
    double* red = reinterpret_cast<double*>(ring + kI8BigStages * kI8BigStage);
    constexpr int kUmin = (MODE == I8_A) ? kUminA : kUminG;
    constexpr int kGroups = 2 * (kDigits - 1) - kUmin + 1;
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    const int tr_ = d >> 7, tc_ = d >> 6, tt = tr_ * tc_;           // tiles per song: rows of 128, columns of 64
    const BigSlot slot = big_slot(blockIdx.x, nprob, tt);
    if (!slot.live) return;
    const int song = slot.song, tile = slot.item;
    const int TY = tile / tc_, TX = tile - TY * tc_;
    const int64_t po = (int64_t)song * g.pstride, ho = (int64_t)song * g.hstride;
    const MatHdr* hB = adv(g.hB, po);
    const MatHdr* hA = adv(g.hA, (int64_t)song * g.astride);
    NsState* const st_p = adv(g.st, po);
    const bool bad = hdr_bad(hA, hB, g.gen);
    const bool skipped = bad || (g.skip && *adv(g.skip, po) != 0);
    if constexpr (MODE == I8_G) {
        if (tile == 0 && tid == 0) {                                 // whatever happens, the host finds the state next to the partials
            const NsState* st = st_p; const Ns32State* s = adv(g.s32, po);
            int* hw = adv(g.host_words, ho); double* hv = adv(g.host_vals, ho);
            hw[0] = bad ? 1 : 0; hw[1] = st->done; hw[2] = st->nonfinite; hw[3] = st->too_few[0]; hw[4] = st->too_few[1];
            hw[5] = s->ok; hw[6] = s->failed; hw[7] = s->final_iter; hw[8] = s->decided_at; hw[9] = s->strict; hw[10] = s->finished;
            hw[11] = skipped ? 1 : 0;
            hv[0] = st->c; hv[1] = st->tr1; hv[2] = st->tr2; hv[3] = st->mean_term;
#pragma unroll
            for (int q = 0; q < 16; ++q) hv[4 + q] = s->res[q];
            hw[14] = (g.scaled && st->mu[0] != 1.0) ? 1 : 0;
            hw[12] = g.gen;
        }
    }
    if (skipped) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const bool alt = (MODE == I8_G) && g.sel && (*adv(g.sel, po) & 1);
    const uint4* Ad = (MODE == I8_A) ? adv(g.Adig, (int64_t)song * g.astride) : adv(alt ? g.Adig_alt : g.Adig, po);
    const uint4* Bd = adv(alt ? g.Bdig_alt : g.Bdig, po);
#ifdef FAD_I8_ABL_G_NOLOOP                                       // ablation: the epilogue of the G product alone
    const int nks = (MODE == I8_G) ? 0 : (d >> 5);
#else
    const int nks = d >> 5;
#endif

    // piece q of a stage: q < 24: A side, row block q / 6, digit q % 6; else B side, column block (q - 24) / 6.  Wave w moves pieces
    // w, w + 8, ... (waves 0..3: five, the others four)
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
    auto issue = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = wave + 8 * i;
            if (q < 36) {
                const bool bs = q >= 24;
                const int blk = bs ? (q - 24) / 6 : q / 6, p = bs ? (q - 24) % 6 : q % 6;
                const uint64_t sb = (uint64_t)((bs ? Bd : Ad) + dg_idx((bs ? 2 * TX : 4 * TY) + blk, ks, p, 0, d));
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb);
                const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
                const uint64_t ub = ((uint64_t)hi << 32) | lo;
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)(((ks % kI8BigStages) * kI8BigStage + q * 64) * 16));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
            }
        }
    };
    i32x16 acc[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][q] = 0;
    static_assert(kI8BigStages >= 2 && kI8BigStages <= 4, "the counted waits below cover two younger stages at most");
    for (int ks = 0; ks < kI8BigStages - 1 && ks < nks; ++ks) issue(ks);
    const bool five = wave < 4;                                    // pieces this wave moves per stage: five (waves 0..3) or four
    for (int ks = 0; ks < nks; ++ks) {
        // stage ks must have landed; up to kI8BigStages - 2 younger stages stay in flight (the count has to be an immediate)
        const int ahead = (nks - 1 - ks < kI8BigStages - 2) ? (nks - 1 - ks) : (kI8BigStages - 2);
        if (ahead == 2) { if (five) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (ahead == 1) { if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // stage ks is there for every wave; the slot of stage ks - 1 is free
#ifndef FAD_I8_ABL_NODMA                                           // ablations (scripts/build_variant.sh + scripts/probe_chain16.py, r06p): 16 pairs of D = 512,
        if (ks + kI8BigStages - 1 < nks) issue(ks + kI8BigStages - 1);   // nsf_i8_big<A> 66 us; without the MFMAs 42, without the loads 56: the int8 MFMAs bound it
#endif
        const i32x4* st = reinterpret_cast<const i32x4*>(ring + (ks % kI8BigStages) * kI8BigStage) + lane;
        i32x4 a[kDigits], b[kDigits];
#pragma unroll
        for (int p = 0; p < kDigits; ++p) { a[p] = st[(wr * 6 + p) * 64]; b[p] = st[(24 + wc * 6 + p) * 64]; }
#ifdef FAD_I8_ABL_NOMMA
#pragma unroll
        for (int p = 0; p < kDigits; ++p) { acc[0][p] += a[p][0] ^ b[p][1]; acc[1][p] += a[p][2] ^ b[p][3]; }
#else
#pragma unroll
        for (int p = kDigits - 1; p >= 0; --p)
#pragma unroll
            for (int q = kDigits - 1; q >= 0; --q)
                if (p + q >= kUmin) acc[p + q - kUmin] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[q], acc[p + q - kUmin], 0, 0, 0);
#endif
    }
    __syncthreads();                                               // the ring is scratch now: one [32][33] float area per wave
    double gp[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gp[q] = 0.0;
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const double wgt = __builtin_ldexp(1.0, 7 * (u + kUmin) - 80);
#pragma unroll
        for (int q = 0; q < 16; ++q) gp[q] = __builtin_fma((double)acc[u][q], wgt, gp[q]);
    }
    float* fin = reinterpret_cast<float*>(ring) + wave * (32 * 33 + 32);
    const int kg = lane >> 5, n = lane & 31;
    const int by = 4 * TY + wr, bx = 2 * TX + wc, nb = d >> 5;
    const int row0 = 32 * by, col0 = 32 * bx;
    double* const stats = adv(g.stats, (MODE == I8_A) ? po : ho);
    double* scal = stats + (size_t)kTileStats * (by * nb + bx);
    double v3[3] = {0.0, 0.0, 0.0};
    if constexpr (MODE == I8_A) {
        const double inv = hdr_inv_s12(hA, hB);
        double* A64 = adv(g.A64, po) + (int64_t)(row0 + 4 * kg) * d + col0 + n;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int q = (reg & 3) + 8 * (reg >> 2);
            A64[(int64_t)q * d] = gp[reg] * inv;
            fin[(q + 4 * kg) * 33 + n] = (float)gp[reg];
            v3[0] += gp[reg] * gp[reg];
            if (by == bx && q + 4 * kg == n) v3[1] += gp[reg];
        }
        __builtin_amdgcn_wave_barrier();
        const SplitMat Pm = adv(g.P, po);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) store_tile_planes(fin, Pm, by, bx, d, pass * 64 + lane);
    } else {
        const SplitMat Zm = adv(g.Z[alt ? 1 : 0], po);
        const SplitMat Ym = adv(g.Y[alt ? 1 : 0], po);
        const double inv_c = 1.0 / st_p->c;
        const double* A64 = adv(g.A64in, po) + (int64_t)(row0 + 4 * kg) * d + col0 + n;
        // Z[gc][gr] pairs with R[gr][gc] in tr(Z R): the lane's 16 rows gr = row0 + 8 j + 4 kg + i are four runs of four consecutive k of ROW gc
        // of Z -- 8 bytes of one piece of the A-layout planes each (until round 6: 32 two-byte loads from the transposed planes)
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 zh[4], zl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int half;
            const size_t zi = fa_elem(col0 + n, row0 + 8 * j + 4 * kg, 0, d, half);
            zh[j] = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(Zm.a + zi) + half);
            zl[j] = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(Zm.a + zi + 64) + half);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int q = (reg & 3) + 8 * (reg >> 2), gr = row0 + q + 4 * kg, gc = col0 + n;
            const double R = A64[(int64_t)q * d] * inv_c - gp[reg];
            if constexpr (WITHR) gp[reg] = R;                      // (kept for the verification planes below)
            const double z = (double)used16(zh[reg >> 2][reg & 3], zl[reg >> 2][reg & 3]);
            v3[0] += z * R; v3[1] += R * R;
            if (by == bx && gr == gc) {
                int hy;
                const size_t yi = fa_elem(gr, gr, 0, d, hy);
                v3[2] += (double)used16(reinterpret_cast<const _Float16*>(Ym.a + yi)[hy], reinterpret_cast<const _Float16*>(Ym.a + yi + 64)[hy]);
            }
            fin[(q + 4 * kg) * 33 + n] = fabsf((float)z);           // |Z[col0 + c][row0 + r]| at (r, c)
        }
        __builtin_amdgcn_wave_barrier();
    }
    // largest row / column sum inside the block (nsf_i8's rule: lanes 0..31 one kind, 32..63 the other)
    double m = 0.0;
    if constexpr (MODE == I8_A) {
        if (lane < 32) { for (int c = 0; c < 32; ++c) m += fabsf(fin[lane * 33 + c]); }
        else { for (int q = 0; q < 32; ++q) m += fabsf(fin[q * 33 + lane - 32]); }
    } else {
        if (lane < 32) { for (int q = 0; q < 32; ++q) m += fin[q * 33 + lane]; }
        else { for (int c = 0; c < 32; ++c) m += fin[(lane - 32) * 33 + c]; }
    }
    double mrow = (lane < 32) ? m : 0.0, mcol = (lane >= 32) ? m : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mrow = fmax(mrow, __shfl_xor(mrow, off)); mcol = fmax(mcol, __shfl_xor(mcol, off));
#pragma unroll
        for (int q = 0; q < 3; ++q) v3[q] += __shfl_xor(v3[q], off);
    }
    if (lane == 0) {
        if constexpr (MODE == I8_A) {
            scal[0] = v3[0]; scal[1] = v3[1]; scal[2] = mrow; scal[3] = mcol;
        } else {
            scal[0] = v3[0]; scal[1] = v3[1]; scal[2] = v3[2]; scal[3] = 0.0;
            double* zmax = stats + (size_t)kTileStats * nb * nb + 2 * (size_t)(by * nb + bx);
            zmax[0] = mrow; zmax[1] = mcol;
        }
    }
    if constexpr (MODE == I8_G && WITHR) {
        if (g.Rv.a) {                                              // kVerScale R of this wave's block, both orientations (ns_fast.h: SP_V2 reads them)
            __builtin_amdgcn_wave_barrier();                       // (the |Z| sums above have read the area)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) fin[((reg & 3) + 8 * (reg >> 2) + 4 * kg) * 33 + n] = (float)(gp[reg] * (double)kVerScale);
            __builtin_amdgcn_wave_barrier();
            const SplitMat Rm = adv(g.Rv, po);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) store_tile_planes(fin, Rm, by, bx, d, pass * 64 + lane);
        }
    }
    (void)red;
}

}  // namespace nsf
}  // namespace fad
