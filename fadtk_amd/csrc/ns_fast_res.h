// D = 128 (Encodec): the WHOLE Newton-Schulz iteration of one song in one workgroup, iterates resident in LDS and registers.
//
// At D = 128 a matrix in split-float16 form is 64 KiB.  The batched kernels (ns_fast.h on 32 x 32 tiles, ns_fast_big.h on one
// 128 x 128 tile) spend their time outside the products there: per iteration and song three launches' worth of workgroups, each
// reading its operands from memory and writing split planes of the result in two orientations -- 0.45 + 8 x (0.11 + 0.34) ms of a
// 8.2 ms call for 2000 songs of [2250 x 128] (profiles/r03k_c4_kernel_stats.csv), for 8 x 3 products of 12.6 MFLOP per song.  Here one
// workgroup of four waves takes a song from Y0 = A / c to its final iterate without touching memory in between:
//
//   * wave j owns the column block j (32 columns) of every iterate as the B OPERAND of the MFMA, in registers: a 32 x 32 result block
//     in the accumulator layout (lane = column, registers = rows) IS a B operand for the two k-steps of its 32 rows, up to a
//     permutation of k inside a k-step (slot (g, i) <-> k = 4 g + (i & 3) + 8 (i >> 2)) that the A operand is stored with as well;
//   * the left factors (A operands: full matrices) live in LDS, two buffers of 64 KiB, as the 1 KiB pieces the MFMA reads: one
//     ds_read_b128 per lane and piece; a new matrix reaches its buffer through 2-byte scatter writes from the accumulator layout;
//   * per iteration  M_j = Z Y_j -> T_j = 1.5 I_j - 0.5 M_j (P = Z, Q = Y in LDS);  Y'_j = Y T_j;  barrier, T -> P;  Z'_j = T Z_j;
//     barriers, Y' -> Q, Z' -> P.  The coupled iteration in its stable form (Y T and T Z, not the commuted products).
//   * the residual ||I - Z Y||_F and ns_fast.h's convergence rules are evaluated by the workgroup itself: no launches, no state
//     traffic; the kernel ends with the planes of the final Y (both orientations) and Z^T in memory for nsf_digitize / nsf_i8<G>.
//
// Replaces nsf_split<FIRST> + (nsf_split<T> + nsf_split<U>) x iterations for D = 128 batches.  FULL = true also forms
// A = Sigma_b Sigma_s (what nsf_i8<A> does) in front and the exact correction (nsf_digitize + nsf_i8<G>) behind, in the same
// workgroup: the two exact products ran on 32 x 32 tiles built for ONE problem and were 45 % of a call of 2000 songs
// (profiles/r03n_c4_kernel_stats.csv: 0.83 + 0.12 + 1.11 ms) -- here they are 30 + 26 digit-pair products of 128^3 on the int8
// MFMA per song, the wave's column block of digit planes in registers, and nothing of the iterate ever goes to memory:
//   phase 0  A (float64) -> memory (it is needed again at the end), ||A||_F^2, tr A and the EXACT column sums of |A| -> the scale
//   phase 2  digits of the final Y: row blocks through a 24 KiB LDS area (same lane, same 16-byte pieces as the split planes:
//            no permutation), the wave's column block from its B-operand registers; R = A / c - Y Y; tr(Z R), ||R||_F^2, tr Y and
//            the exact ||Z||_1, ||Z||_inf -> the pinned host record nsf_i8<G> writes (frechet.hip: fast_decide_one reads both).
#pragma once
#include "ns_fast.h"

namespace fad {
namespace nsf {

constexpr size_t kResLds = 2 * 65536 + 1024;
constexpr size_t kResLdsFull = kResLds + 6 * 4 * 1024;       // + the digit pieces of one row block of Y

struct ResArgs {
    int gen, max_low;
    double thr_pred;
    const MatHdr* hA; const MatHdr* hB; int64_t pstride;
    double* A64; const double* statsA;   // FULL: A64 is written by this kernel, statsA unused
    NsState* st; Ns32State* s32;
    SplitMat Y[2], Z[2];                 // outputs (not FULL): Y[f & 1].a, Y[f & 1].at, Z[f & 1].at of the final iterate f
    // FULL
    const uint4* Adig; const uint4* Bdig;        // digit planes of s_b Sigma_b (shared) and of (s_s Sigma_s)^T (per song)
    int64_t hstride; double* stats; int* host_words; double* host_vals;      // pinned host: what nsf_i8<G> leaves per problem
    int scaled; double l0_scale;         // scaled steps (ns_check.h), as ns_fast_big.h
};

// position of element (row, col) of a matrix stored as A-operand pieces in LDS: byte offset of its hi half (lo: + 1024)
__device__ __forceinline__ int res_lds_off(int row, int col) {
    const int c = col & 15;
    return (((row >> 5) * 8 + (col >> 4)) * 2) * 1024 + (32 * ((c >> 2) & 1) + (row & 31)) * 16 + 2 * ((c & 3) + 4 * (c >> 3));
}

template <bool FULL>
__global__ __launch_bounds__(256) void nsf_res128(ResArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const P = lds;                       // A-operand pieces of Z, then of T, then of Z'
    char* const Q = lds + 65536;               // ... of Y, then of Y'
    double* const red = reinterpret_cast<double*>(lds + 131072);       // 64 doubles
    constexpr int d = 128, nb = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);            // this wave's column block
    const int n = lane & 31, kg = lane >> 5;
    const int64_t po = (int64_t)blockIdx.x * g.pstride, ho = (int64_t)blockIdx.x * g.hstride;
    const MatHdr* hB = adv(g.hB, po);
    NsState* const st = adv(g.st, po);
    Ns32State* const s32 = adv(g.s32, po);
    auto rowc = [](int reg) { return (reg & 3) + 8 * (reg >> 2); };
    // FULL: what nsf_i8<G> snapshots for the host, written by thread 0 on every way out
    auto snapshot = [&](bool bad, bool skipped) {
        if constexpr (FULL) {
            int* hw = adv(g.host_words, ho); double* hv = adv(g.host_vals, ho);
            hw[0] = bad ? 1 : 0; hw[1] = st->done; hw[2] = st->nonfinite; hw[3] = st->too_few[0]; hw[4] = st->too_few[1];
            hw[5] = s32->ok; hw[6] = s32->failed; hw[7] = s32->final_iter; hw[8] = s32->decided_at; hw[9] = s32->strict; hw[10] = s32->finished;
            hw[11] = skipped ? 1 : 0;
            hv[0] = st->c; hv[1] = st->tr1; hv[2] = st->tr2; hv[3] = st->mean_term;
            for (int q = 0; q < 16; ++q) hv[4 + q] = s32->res[q];
            hw[12] = g.gen;
        }
    };
    if (hdr_bad(g.hA, hB, g.gen)) { if (tid == 0) snapshot(true, true); return; }

    // ---- FULL, phase 0: A = Sigma_b Sigma_s exact (30 digit-pair products on the int8 MFMA), column block j of this wave
    double fro2_l = 0.0, tr_l = 0.0, colsum_max = 0.0;
    if constexpr (FULL) {
        constexpr int kGroups = 2 * (kDigits - 1) - kUminA + 1;
        i32x4 bd[4][kDigits];
        {
            const i32x4* pb = reinterpret_cast<const i32x4*>(adv(g.Bdig, po) + dg_idx(j, 0, 0, lane, d));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int p = 0; p < kDigits; ++p) bd[ks][p] = pb[(ks * kDigits + p) * 64];
        }
        const double inv12 = hdr_inv_s12(g.hA, hB);
        double* A64 = adv(g.A64, po) + (int64_t)(4 * kg) * d + 32 * j + n;
        double cs = 0.0;
#pragma unroll 1
        for (int rbo = 0; rbo < 4; ++rbo) {
            i32x16 acc[kGroups];
#pragma unroll
            for (int u = 0; u < kGroups; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[u][q] = 0;
            const i32x4* pa = reinterpret_cast<const i32x4*>(g.Adig + dg_idx(rbo, 0, 0, lane, d));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                i32x4 a[kDigits];
#pragma unroll
                for (int p = 0; p < kDigits; ++p) a[p] = pa[(ks * kDigits + p) * 64];
#pragma unroll
                for (int p = kDigits - 1; p >= 0; --p)
#pragma unroll
                    for (int q = kDigits - 1; q >= 0; --q)
                        if (p + q >= kUminA) acc[p + q - kUminA] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], bd[ks][q], acc[p + q - kUminA], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                double v = 0.0;
#pragma unroll
                for (int u = 0; u < kGroups; ++u) v = __builtin_fma((double)acc[u][q], __builtin_ldexp(1.0, 7 * (u + kUminA) - 80), v);
                A64[(int64_t)(32 * rbo + rowc(q)) * d] = v * inv12;                 // caller's units, as nsf_i8<A> leaves it
                fro2_l += v * v; cs += fabs(v);
                if (rbo == j && rowc(q) + 4 * kg == n) tr_l += v;
            }
        }
        cs += __shfl_xor(cs, 32);                                                    // the two halves of column 32 j + n
        colsum_max = cs;
    }

    // ---- the scale (what nsf_split<FIRST> does): c = max(u / 2.9, ||A||_F^2 / tr A), u >= the spectral radius
    double c, inv_c;
    double mu = 1.0, l_cur = 1.0;             // step scale of the iteration at hand and the lower bound of its iterate's x (scaled steps)
    {
        double inf_b = 1e300, one_b = 1e300;
        double v2[2] = {0.0, 0.0};
        if constexpr (FULL) {
            // exact ||A||_1 (the largest column sum); the infinity norm would need row sums across the waves: not formed
            one_b = wg8_max<4>(colsum_max, red);
            v2[0] = fro2_l; v2[1] = tr_l;
            __syncthreads();
            wg8_sum<2, 4>(v2, red + 32);
        } else {
            const double* scal = adv(g.statsA, po);
            double* tmax = red;                                              // [2][16]
            if (tid < nb * nb) {
                v2[0] = scal[kTileStats * tid]; v2[1] = scal[kTileStats * tid + 1];
                const double r0 = scal[kTileStats * tid + 2], r1 = scal[kTileStats * tid + 3];
                tmax[tid] = (r0 == r0) ? r0 : 1e300; tmax[16 + tid] = (r1 == r1) ? r1 : 1e300;
            }
            __syncthreads();
            inf_b = 0.0; one_b = 0.0;
            for (int line = 0; line < nb; ++line) {
                double rs = 0.0, cs = 0.0;
                for (int q = 0; q < nb; ++q) { rs += tmax[line * nb + q]; cs += tmax[16 + q * nb + line]; }
                inf_b = fmax(inf_b, rs); one_b = fmax(one_b, cs);
            }
            __syncthreads();
            wg8_sum<2, 4>(v2, red + 32);
        }
        const double fro2 = v2[0], trA = v2[1];
        double u = sqrt(fro2);
        if (inf_b < u) u = inf_b;
        if (one_b < u) u = one_b;
        c = u / 2.9;
        const double wmean = (trA > 0.0) ? fro2 / trA : 0.0;
        if (wmean > c && wmean <= u) c = wmean;
        const double mean_term = st->mean_term, tr1 = g.hA->tr, tr2 = hB->tr;
        const bool bad = !(fro2 == fro2) || isinf(fro2) || !(trA == trA) || isinf(trA) || !(mean_term == mean_term) || isinf(mean_term);
        const bool zero = !bad && !(c > 0.0);
        const bool hopeless = !bad && !zero && (trA * trA < 0.25 * (double)d * fro2 || c < 0.0078125);
        if (g.scaled && !bad && !zero && !hopeless && u > 0.0 && trA * trA < 0.8 * (double)d * fro2) {       // (ns_fast_big.h: SP_FIRST)
            c = u;
            l_cur = ns_l0_from_participation((float)(trA * trA / fro2), d) * g.l0_scale;
            if (l_cur > 0.5) l_cur = 0.5;
            mu = ns_step_scale(l_cur);
        }
        if (tid == 0) {
            st->c = zero ? 1.0 : c * hdr_inv_s12(g.hA, hB);
            st->tr1 = tr1; st->tr2 = tr2;
            st->res_last = 0.0; st->tr_last = 0.0; st->res_min = 1e300; st->tr_safe = 0.0; st->has_safe = 0;
            st->final_iter = zero ? 0 : -1; st->conv = zero ? 1 : 0;
            st->nonfinite = bad ? 1 : 0; st->done = (bad || zero) ? 1 : 0; st->finished = (bad || zero) ? 1 : 0;
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->final_iter = -1; s32->failed = 0;
            s32->upd_skip[0] = 0; s32->upd_skip[1] = 0; s32->skip_corr = 1; s32->decided_at = -1; s32->strict = 0; s32->grew = 0;
            s32->res[0] = 1e300;
            if (bad || zero || hopeless) { s32->done = 1; s32->finished = 1; s32->failed = 1; s32->upd_skip[0] = 1; s32->upd_skip[1] = 1; snapshot(false, true); }
        }
        if (bad || zero || hopeless) return;
        inv_c = 1.0 / (c * hdr_inv_s12(g.hA, hB));       // for A in the caller's units
    }

    // ---- helpers on the accumulator layout: register reg of row block rbo = element (32 rbo + rowc(reg) + 4 kg, 32 j + n);
    // as B-operand registers: k-step 2 rbo + (reg >> 3), slot reg & 7
    // a column block into an LDS buffer as A-operand pieces (2-byte scatter): one lane-dependent base, the rest immediates
    const int scat_base = res_lds_off(4 * kg, 32 * j + n);
    auto scatter = [&](char* buf, const f16x8 (&Bh)[8], const f16x8 (&Bl)[8]) {
        char* p = buf + scat_base;
#pragma unroll
        for (int rbo = 0; rbo < 4; ++rbo)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int off = rbo * 16384 + rowc(reg) * 16;
                *reinterpret_cast<_Float16*>(p + off) = Bh[2 * rbo + (reg >> 3)][reg & 7];
                *reinterpret_cast<_Float16*>(p + off + 1024) = Bl[2 * rbo + (reg >> 3)][reg & 7];
            }
    };
    // (Oh, Ol) = column block j of f((matrix in buf) x (column block in Bh, Bl)), split-float16: hi hi + (hi lo + lo hi) / 2048
    // Software-pipelined over the four row blocks: the conversion of block rbo - 1 (accumulator reads, split, pack: ~650 VALU
    // instructions) stands next to the 24 MFMAs of block rbo and is issued in their shadow -- with one wave per SIMD nothing else
    // would fill the 28 idle issue cycles behind every MFMA.  The fence lets ALU and MFMA instructions cross and holds the LDS
    // reads of the next block back (all 64 of a product hoisted to the top cost 256 registers).
    auto product = [&](const char* buf, const f16x8 (&Bh)[8], const f16x8 (&Bl)[8], f16x8 (&Oh)[8], f16x8 (&Ol)[8], auto f) {
        const f16x8* a = reinterpret_cast<const f16x8*>(buf) + lane;
        f32x16 p0, p1;
        auto convert = [&](int rbo, const f32x16& c0, const f32x16& c1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float t = f(rbo, q, c0[q] + c1[q] * kLoInv);
                _Float16 h, l; split16(t, h, l);
                Oh[2 * rbo + (q >> 3)][q & 7] = h; Ol[2 * rbo + (q >> 3)][q & 7] = l;
            }
        };
#pragma unroll
        for (int rbo = 0; rbo < 4; ++rbo) {
            f32x16 a0, a1;
#pragma unroll
            for (int q = 0; q < 16; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f16x8 ah = a[((rbo * 8 + ks) * 2) * 64], al = a[((rbo * 8 + ks) * 2 + 1) * 64];
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[ks], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[ks], a1, 0, 0, 0);
            }
            if (rbo > 0) convert(rbo - 1, p0, p1);
            p0 = a0; p1 = a1;
            __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x8);
        }
        convert(3, p0, p1);
    };
    auto ident = [](int, int, float x) { return x; };

    // ---- k = 0: Y0 = A / c, Z0 = I:  T0 = 1.5 I - 0.5 Y0,  Y1 = Y0 T0,  Z1 = T0
    f16x8 Yh[8], Yl[8], Zh[8], Zl[8], Th[8], Tl[8];
    {
        const double* A64 = adv(g.A64, po) + (int64_t)(4 * kg) * d + 32 * j + n;       // (FULL: this lane's own stores of phase 0)
        const float a0 = (float)(0.5 * mu * mu * mu), b0 = (float)(1.5 * mu);            // T0 = 1.5 mu_0 I - 0.5 mu_0^3 Y0
#pragma unroll
        for (int rbo = 0; rbo < 4; ++rbo)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float y0 = (float)(A64[(int64_t)(32 * rbo + rowc(reg)) * d] * inv_c);
                asm volatile("" : "+v"(y0));                 // (or the two roundings fuse into an emulated double -> half conversion)
                const float t0 = ((rbo == j && rowc(reg) + 4 * kg == n) ? b0 : 0.f) - a0 * y0;
                _Float16 h, l;
                split16(y0, h, l); Yh[2 * rbo + (reg >> 3)][reg & 7] = h; Yl[2 * rbo + (reg >> 3)][reg & 7] = l;
                split16(t0, h, l); Th[2 * rbo + (reg >> 3)][reg & 7] = h; Tl[2 * rbo + (reg >> 3)][reg & 7] = l;
            }
        scatter(Q, Yh, Yl);                                  // Y0 as the left factor of Y1 = Y0 T0
        scatter(P, Th, Tl);                                  // Z1 = T0
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { Zh[ks] = Th[ks]; Zl[ks] = Tl[ks]; }
        __syncthreads();
        product(Q, Th, Tl, Yh, Yl, ident);                   // Y1 (column block j)
        __syncthreads();
        scatter(Q, Yh, Yl);
        __syncthreads();
    }

    // ---- iterations k = 1 ..: P holds Z_k, Q holds Y_k as left factors; (Yh, Yl), (Zh, Zl) their column blocks j
    int final_iter = -1, decided_at = -1, ok = 0, failed = 0;
    double prev = 1e300;
    bool grew = false;
    for (int k = 1;; ++k) {
        float ss = 0.f;
        {   // this iteration's step scale: mu (1 without scaled steps) from the bound of iterate k
            double lk = l_cur;
            mu = g.scaled ? ns_step_scale(lk) : 1.0;
            if (g.scaled && k > 1) { const double cap = ns_scale_cap(prev, d); if (cap < mu) mu = cap; }     // (prev: the residual of iterate k - 1)
        }
        const float ak = (float)(0.5 * mu * mu * mu), gk = (float)(1.5 * mu - 0.5 * mu * mu * mu);
        product(P, Yh, Yl, Th, Tl, [&](int rbo, int reg, float m) {      // M_j = Z Y_j -> T_j = 1.5 mu I_j - 0.5 mu^3 M_j
            const bool dg = rbo == j && rowc(reg) + 4 * kg == n;
            const float e = (dg ? ak : 0.f) - ak * m;                   // T - (1.5 mu - 0.5 mu^3) I
            ss += e * e;
            return dg ? e + gk : e;
        });
        // r_k = ||I - Z_k Y_k||_F = ||T_k - gamma I||_F / (0.5 mu^3), the same for every thread; then the rules of nsf_check
        double s1[1] = {(double)ss};
        wg8_sum<1, 4>(s1, red);                              // (two barriers: every wave is done reading P as Z)
        const double res = sqrt(s1[0]) / (0.5 * mu * mu * mu);
        if (g.scaled) {                                      // bound of iterate k + 1 (nsf_check)
            if (res == res && res < 1.0) { const double lr = sqrt(1.0 - res); if (lr > l_cur) l_cur = lr; }
            (void)ns_step_scale_with(mu, l_cur);
        }
        if (tid == 0 && k < 16) s32->res[k] = res;
        const bool finite = (res == res) && !isinf(res);
        const bool grows = k >= 4 && res > prev && res > 1e-3;
        const bool give_up = grows && (!g.scaled || grew);          // (nsf_check: one bump after an over-scaled step is not a failure)
        grew = grows;
        if (!finite || k + 1 >= g.max_low || give_up) { failed = 1; final_iter = k; decided_at = k; break; }
        if (res <= 1e-3 && (res > 0.3 * prev || res <= 1e-6)) { ok = 1; final_iter = k; decided_at = k; break; }      // at the floor: Y_k is final
        const double bound = 0.75 * res * res + 0.25 * res * res * res;
        const bool last = bound <= g.thr_pred && mu == 1.0;  // Y_{k+1} is final (the update about to run is a plain step)
        prev = res;
        scatter(P, Th, Tl);                                  // T_k becomes the left factor of Z' = T Z
        product(Q, Th, Tl, Yh, Yl, ident);                   // Y'_j = Y T_j  (the old column block of Y is not needed any more)
        __syncthreads();                                     // T complete in P; every wave is done reading Q as Y
        scatter(Q, Yh, Yl);
        product(P, Zh, Zl, Th, Tl, ident);                   // Z'_j = T Z_j, into the registers T_j has left
        __syncthreads();                                     // every wave is done reading P as T
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { Zh[ks] = Th[ks]; Zl[ks] = Tl[ks]; }
        scatter(P, Zh, Zl);
        __syncthreads();
        if (last) { ok = 1; final_iter = k + 1; decided_at = k; break; }
    }
    if (tid == 0) {
        s32->finished = 1; s32->done = 1; s32->final_iter = final_iter; s32->decided_at = decided_at;
        s32->ok = ok; s32->failed = failed; s32->skip_corr = ok ? 0 : 1;
        s32->upd_skip[0] = 1; s32->upd_skip[1] = 1;
        if (!ok) snapshot(false, true);
    }
    if (!ok) return;
    __syncthreads();

    if constexpr (FULL) {
        // ---- phase 2: the exact correction on the final iterate (what nsf_digitize + nsf_i8<G> do for the other dimensions)
        char* const DG = lds + 131072 + 1024;        // digit pieces [k-step of 32][digit][lane] of ONE row block of Y
        constexpr int kGroups = 2 * (kDigits - 1) - kUminG + 1;
        // from the registers, before they go: tr Y (the lane that holds a diagonal element), column sums of |Z|, digits of Y_j
        double try_l = 0.0, zcol = 0.0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) zcol += (double)fabsf(used16(Zh[ks][i], Zl[ks][i]));
        zcol += __shfl_xor(zcol, 32);
#pragma unroll
        for (int rbo = 0; rbo < 4; ++rbo)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (rbo == j && rowc(reg) + 4 * kg == n) try_l += (double)used16(Yh[2 * rbo + (reg >> 3)][reg & 7], Yl[2 * rbo + (reg >> 3)][reg & 7]);
        i32x4 yd[4][kDigits];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint32_t w[kDigits][4];
#pragma unroll
            for (int p = 0; p < kDigits; ++p) { w[p][0] = 0u; w[p][1] = 0u; w[p][2] = 0u; w[p][3] = 0u; }
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                int dg[kDigits];
                digits_of<float>(used16(Yh[2 * ks + (sl >> 3)][sl & 7], Yl[2 * ks + (sl >> 3)][sl & 7]), dg);
#pragma unroll
                for (int p = 0; p < kDigits; ++p) w[p][sl >> 2] |= ((uint32_t)dg[p] & 0xffu) << (8 * (sl & 3));
            }
#pragma unroll
            for (int p = 0; p < kDigits; ++p) { yd[ks][p][0] = (int)w[p][0]; yd[ks][p][1] = (int)w[p][1]; yd[ks][p][2] = (int)w[p][2]; yd[ks][p][3] = (int)w[p][3]; }
        }
        // ||Z||_inf from the A-operand pieces in P: one thread per row
        double zrow = 0.0;
        if (tid < d) {                                         // row tid = lanes (tid & 31, g = 0 / 1) of the pieces of row block tid >> 5
            const f16x8* zp = reinterpret_cast<const f16x8*>(P) + ((tid >> 5) * 8 * 2) * 64 + (tid & 31);
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const f16x8 h = zp[(2 * ks) * 64 + 32 * gg], l = zp[(2 * ks + 1) * 64 + 32 * gg];
#pragma unroll
                    for (int i = 0; i < 8; ++i) zrow += (double)fabsf(used16(h[i], l[i]));
                }
        }
        const double zinf = wg8_max<4>(zrow, red), zone = wg8_max<4>(zcol, red);
        double corr_l = 0.0, r2_l = 0.0;
        const double* A64 = adv(g.A64, po) + (int64_t)(4 * kg) * d + 32 * j + n;
#pragma unroll 1
        for (int rbo = 0; rbo < 4; ++rbo) {
            {   // digit pieces of rows 32 rbo .. of Y: thread (ks, lane') turns its own two split pieces into six digit pieces
                const int ks = tid >> 6;
                const f16x8* yq = reinterpret_cast<const f16x8*>(Q) + ((rbo * 8 + 2 * ks) * 2) * 64 + lane;
                const f16x8 h0 = yq[0], l0 = yq[64], h1 = yq[128], l1 = yq[192];
                uint32_t w[kDigits][4];
#pragma unroll
                for (int p = 0; p < kDigits; ++p) { w[p][0] = 0u; w[p][1] = 0u; w[p][2] = 0u; w[p][3] = 0u; }
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) {
                    int dg[kDigits];
                    digits_of<float>(sl < 8 ? used16(h0[sl & 7], l0[sl & 7]) : used16(h1[sl & 7], l1[sl & 7]), dg);
#pragma unroll
                    for (int p = 0; p < kDigits; ++p) w[p][sl >> 2] |= ((uint32_t)dg[p] & 0xffu) << (8 * (sl & 3));
                }
#pragma unroll
                for (int p = 0; p < kDigits; ++p)
                    *reinterpret_cast<uint4*>(DG + ((ks * kDigits + p) * 64 + lane) * 16) = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
            }
            __syncthreads();
            i32x16 acc[kGroups];
#pragma unroll
            for (int u = 0; u < kGroups; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[u][q] = 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                i32x4 a[kDigits];
#pragma unroll
                for (int p = 0; p < kDigits; ++p) a[p] = *reinterpret_cast<const i32x4*>(DG + ((ks * kDigits + p) * 64 + lane) * 16);
#pragma unroll
                for (int p = kDigits - 1; p >= 0; --p)
#pragma unroll
                    for (int q = kDigits - 1; q >= 0; --q)
                        if (p + q >= kUminG) acc[p + q - kUminG] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], yd[ks][q], acc[p + q - kUminG], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                double G = 0.0;
#pragma unroll
                for (int u = 0; u < kGroups; ++u) G = __builtin_fma((double)acc[u][q], __builtin_ldexp(1.0, 7 * (u + kUminG) - 80), G);
                const double R = A64[(int64_t)(32 * rbo + rowc(q)) * d] * inv_c - G;
                r2_l += R * R;
                const int o = res_lds_off(32 * j + n, 32 * rbo + rowc(q) + 4 * kg);      // Z[32 j + n][k]: the mirror element of R[k][32 j + n]
                const double z = (double)used16(*reinterpret_cast<const _Float16*>(P + o), *reinterpret_cast<const _Float16*>(P + o + 1024));
                corr_l += z * R;
            }
            __syncthreads();                                 // before the next row block's digits overwrite DG
        }
        double v3[3] = {corr_l, r2_l, try_l};
        wg8_sum<3, 4>(v3, red);
        if (tid == 0) {
            double* stats = adv(g.stats, ho);
            for (int q = 0; q < (kTileStats + 2) * nb * nb; ++q) stats[q] = 0.0;
            stats[0] = v3[0]; stats[1] = v3[1]; stats[2] = v3[2];
            stats[kTileStats * nb * nb] = zinf; stats[kTileStats * nb * nb + 1] = zone;      // exact norms, filed under tile (0, 0)
            snapshot(false, false);
        }
        return;
    }

    // ---- the final iterate to memory in ns_fast.h's fragment-major layout: Y (A layout), Y^T and Z^T (A layouts of the transposes).
    // Task (piece, plane-pair): lane' = (m, g2) of piece (rb, ks): 8 halves at k = 16 ks + 8 g2 + 0..7.
    const int par = final_iter & 1;
    const SplitMat Yo = adv(g.Y[par], po), Zo = adv(g.Z[par], po);
    for (int t = tid; t < 4 * 8 * 64; t += 256) {
        const int ln = t & 63, ks = (t >> 6) & 7, rb = t >> 9;
        const int m = ln & 31, g2 = ln >> 5;
        f16x8 yh, yl, yth, ytl, zth, ztl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 32 * rb + m, k = 16 * ks + 8 * g2 + i;
            const int o = res_lds_off(row, k), ot = res_lds_off(k, row);
            yh[i] = *reinterpret_cast<const _Float16*>(Q + o); yl[i] = *reinterpret_cast<const _Float16*>(Q + o + 1024);
            yth[i] = *reinterpret_cast<const _Float16*>(Q + ot); ytl[i] = *reinterpret_cast<const _Float16*>(Q + ot + 1024);
            zth[i] = *reinterpret_cast<const _Float16*>(P + ot); ztl[i] = *reinterpret_cast<const _Float16*>(P + ot + 1024);
        }
        uint4 u;
        __builtin_memcpy(&u, &yh, 16); Yo.a[fa_idx(rb, ks, 0, ln, d)] = u;
        __builtin_memcpy(&u, &yl, 16); Yo.a[fa_idx(rb, ks, 1, ln, d)] = u;
        __builtin_memcpy(&u, &yth, 16); Yo.at[fa_idx(rb, ks, 0, ln, d)] = u;
        __builtin_memcpy(&u, &ytl, 16); Yo.at[fa_idx(rb, ks, 1, ln, d)] = u;
        __builtin_memcpy(&u, &zth, 16); Zo.at[fa_idx(rb, ks, 0, ln, d)] = u;
        __builtin_memcpy(&u, &ztl, 16); Zo.at[fa_idx(rb, ks, 1, ln, d)] = u;
    }
}

}  // namespace nsf
}  // namespace fad
