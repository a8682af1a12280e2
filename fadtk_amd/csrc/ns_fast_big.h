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
// back) and one ds_read_b128 per lane brings it to the MFMA.  A ring of four stages of one k-step; 12 MFMAs (2 x 2 blocks x hi hi,
// hi lo, lo hi) against 8 ds_read_b128 and 4 LDS-DMA instructions per wave and k-step; operand traffic per tile is a quarter of
// the small-tile kernels'.  64 KiB of LDS: two workgroups per CU.
//
// The epilogue is ns_fast.h's, a 32 x 32 block at a time: each wave parks a block of its 64 x 64 result in its own (now free) part
// of the ring and runs the store_tile tasks over it in a few passes.
#pragma once
#include "ns_fast.h"

namespace fad {
namespace nsf {

constexpr int kBigStages = 4;
constexpr int kBigStage = 1024;                                   // uint4 per stage: A pieces [4 rb][2 planes][64 lanes], then B pieces
constexpr size_t kBigLds = (size_t)kBigStages * kBigStage * 16 + 256;   // + the reduction scratch

template <int MODE>
__global__ __launch_bounds__(256, 2) void nsf_big(SplitArgs g) {
    static_assert(MODE == SP_T || MODE == SP_U, "iteration products only");
    extern __shared__ __attribute__((aligned(16))) uint4 ring[];
    double* red = reinterpret_cast<double*>(ring + kBigStages * kBigStage);
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    // ---- which (song, product, tile): a 1-D grid.  Workgroup L runs on XCD L % 8 and is that XCD's (L / 8)-th: the XCD takes songs
    // xcd, xcd + 8, ... one after the other, product by product, tile by tile -- the t^2 tiles of a product run side by side on ONE
    // XCD and walk the k range in step, so its L2 fetches every operand strip once for the 2 t tiles that read it.  (With z = song the
    // tiles of a song were dealt round the eight XCDs and every L2 fetched everything: 885 MB per T launch at D = 768 x 32 songs through
    // the fabric, 4.9 TB/s, a quarter of the matrix rate -- profiles/r03i_c5_kernel_stats.csv.)  The songs are padded to a multiple of
    // eight (g.nprob_pad); after the product workgroups come the check workgroups of SP_U, one per song.
    constexpr int ZP = (MODE == SP_U) ? 2 : 1;
    const int t = d >> 7, tt = t * t;
    const int L = blockIdx.x;
    const int nprod = tt * ZP * g.nprob_pad;
    if constexpr (MODE == SP_U) {
        if (L >= nprod) { nsf_check<256>(g, (int64_t)(L - nprod) * g.pstride, red); return; }
    }
    const int xcd = L & 7, idx = L >> 3;
    const int unit = idx / tt, tile = idx - unit * tt;
    const int song = 8 * (unit / ZP) + xcd;
    if (song >= g.nprob) return;
    const int zi = unit % ZP;
    const int TY = tile / t, TX = tile - TY * t;
    const int64_t po = (int64_t)song * g.pstride;
    const MatHdr* hB = adv(g.hB, po);
    if (hdr_bad(g.hA, hB, g.gen)) return;
    if (g.skip && *adv(g.skip, po) != 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nks = d >> 4;

    // ---- loads: wave w moves pieces 4 w .. 4 w + 3 of a stage (waves 0, 1: the A side, row blocks 2 (w & 1) + {0, 1}; waves 2, 3: the
    // B side), hi and lo plane of a block being 2 KiB in a row.  Wave-uniform 64-bit base in SGPRs + the lane's 16-byte offset (the
    // form that overlaps with the MFMAs, moments_kernels.h: issue_fast).
    const SplitMat Am = adv(g.A[zi], po), Bm = adv(g.B[zi], po);
    const int side = wave >> 1, blk0 = 2 * (wave & 1);
    const uint4* src_mat = side ? Bm.at : Am.a;
    const int blk_global0 = 4 * (side ? TX : TY) + blk0;
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
    // (all tiles of a product walk k from 0 in step -- letting tile (TY, TX) start (TY t + TX) s k-steps in, s = 1 .. 13, so that they do
    //  not ask for the same pieces at the same moment, measured 1-14 % SLOWER: the six tiles that share a strip want it together)
    auto issue = [&](int ks) {
        const int kk = ks;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int blk = q >> 1, plane = q & 1;
            const uint64_t sb = (uint64_t)(src_mat + fa_idx(blk_global0 + blk, kk, plane, 0, d));
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            const uint32_t dst = ring_lds + (uint32_t)(((ks % kBigStages) * kBigStage + (side * 8 + (blk0 + blk) * 2 + plane) * 64) * 16);
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(dst);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0[i][j][q] = 0.f; acc1[i][j][q] = 0.f; }

    for (int ks = 0; ks < kBigStages - 1 && ks < nks; ++ks) issue(ks);
    for (int ks = 0; ks < nks; ++ks) {
        // stage ks must have landed; up to two younger stages stay in flight (the count has to be an immediate)
        const int ahead = (nks - 1 - ks < kBigStages - 2) ? (nks - 1 - ks) : (kBigStages - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // stage ks is in LDS for every wave; the slot of stage ks - 1 is free
        if (ks + kBigStages - 1 < nks) issue(ks + kBigStages - 1);
        const f16x8* st = reinterpret_cast<const f16x8*>(ring + (ks % kBigStages) * kBigStage) + lane;
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = st[((2 * wr + i) * 2) * 64]; al[i] = st[((2 * wr + i) * 2 + 1) * 64]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { bh[j] = st[(8 + (2 * wc + j) * 2) * 64]; bl[j] = st[(8 + (2 * wc + j) * 2 + 1) * 64]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
            }
    }
    __syncthreads();                                           // every wave has left the ring: it is scratch now

    // ---- epilogue, per wave: its four 32 x 32 blocks one after the other through a [32][33] float area of its own
    float* fin = reinterpret_cast<float*>(ring + wave * kBigStage);
    const int kg = lane >> 5, r = lane & 31;
    const float alpha = (MODE == SP_T) ? g.alpha : 1.f, beta = (MODE == SP_T) ? g.beta_eye : 0.f;
    const SplitMat Cm = adv(g.C[zi], po);
    const bool with_digits = (MODE == SP_U) && zi == 0 && g.Cdig[0];
    uint4* dig = with_digits ? adv(g.Cdig[0], po) : nullptr;
    uint4* dig_t = with_digits ? adv(g.Cdig_t[0], po) : nullptr;
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int by = 4 * TY + 2 * wr + i, bx = 4 * TX + 2 * wc + j;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int q = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                const bool dg = (by == bx) && (q == r);
                const float v = alpha * (acc0[i][j][reg] + acc1[i][j][reg] * kLoInv) + (dg ? beta : 0.f);
                if constexpr (MODE == SP_T) { const double e = (double)v - (dg ? (double)g.gamma : 0.0); ss += e * e; }
                fin[q * 33 + r] = v;
            }
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

}  // namespace nsf
}  // namespace fad
