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
// Replaces nsf_split<FIRST> + (nsf_split<T> + nsf_split<U>) x iterations for D = 128 batches; A = Sigma_b Sigma_s (nsf_i8<A>) and
// the exact correction (nsf_i8<G>) stay what they are.
#pragma once
#include "ns_fast.h"

namespace fad {
namespace nsf {

constexpr size_t kResLds = 2 * 65536 + 1024;

struct ResArgs {
    int gen, max_low;
    double thr_pred;
    const MatHdr* hA; const MatHdr* hB; int64_t pstride;
    const double* A64; const double* statsA;
    NsState* st; Ns32State* s32;
    SplitMat Y[2], Z[2];                 // outputs: Y[f & 1].a, Y[f & 1].at, Z[f & 1].at of the final iterate f
};

// position of element (row, col) of a matrix stored as A-operand pieces in LDS: byte offset of its hi half (lo: + 1024)
__device__ __forceinline__ int res_lds_off(int row, int col) {
    const int c = col & 15;
    return (((row >> 5) * 8 + (col >> 4)) * 2) * 1024 + (32 * ((c >> 2) & 1) + (row & 31)) * 16 + 2 * ((c & 3) + 4 * (c >> 3));
}

__global__ __launch_bounds__(256) void nsf_res128(ResArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const P = lds;                       // A-operand pieces of Z, then of T, then of Z'
    char* const Q = lds + 65536;               // ... of Y, then of Y'
    double* const red = reinterpret_cast<double*>(lds + 131072);       // 64 doubles
    int* const flag = reinterpret_cast<int*>(lds + 131072 + 512);
    constexpr int d = 128, nb = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);            // this wave's column block
    const int n = lane & 31, kg = lane >> 5;
    const int64_t po = (int64_t)blockIdx.x * g.pstride;
    const MatHdr* hB = adv(g.hB, po);
    if (hdr_bad(g.hA, hB, g.gen)) return;
    NsState* const st = adv(g.st, po);
    Ns32State* const s32 = adv(g.s32, po);

    // ---- the scale, from nsf_i8<A>'s tile statistics (what nsf_split<FIRST> does): c = max(u / 2.9, ||A||_F^2 / tr A)
    double c, inv_c;
    {
        const double* scal = adv(g.statsA, po);
        double* tmax = red;                                              // [2][16]
        double v2[2] = {0.0, 0.0};
        if (tid < nb * nb) {
            v2[0] = scal[kTileStats * tid]; v2[1] = scal[kTileStats * tid + 1];
            const double r0 = scal[kTileStats * tid + 2], r1 = scal[kTileStats * tid + 3];
            tmax[tid] = (r0 == r0) ? r0 : 1e300; tmax[16 + tid] = (r1 == r1) ? r1 : 1e300;
        }
        __syncthreads();
        double inf_b = 0.0, one_b = 0.0;
        for (int line = 0; line < nb; ++line) {
            double rs = 0.0, cs = 0.0;
            for (int q = 0; q < nb; ++q) { rs += tmax[line * nb + q]; cs += tmax[16 + q * nb + line]; }
            inf_b = fmax(inf_b, rs); one_b = fmax(one_b, cs);
        }
        __syncthreads();
        wg8_sum<2, 4>(v2, red + 32);
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
        if (tid == 0) {
            st->c = zero ? 1.0 : c * hdr_inv_s12(g.hA, hB);
            st->tr1 = tr1; st->tr2 = tr2;
            st->res_last = 0.0; st->tr_last = 0.0; st->res_min = 1e300; st->tr_safe = 0.0; st->has_safe = 0;
            st->final_iter = zero ? 0 : -1; st->conv = zero ? 1 : 0;
            st->nonfinite = bad ? 1 : 0; st->done = (bad || zero) ? 1 : 0; st->finished = (bad || zero) ? 1 : 0;
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->final_iter = -1; s32->failed = 0;
            s32->upd_skip[0] = 0; s32->upd_skip[1] = 0; s32->skip_corr = 1; s32->decided_at = -1; s32->strict = 0;
            s32->res[0] = 1e300;
            if (bad || zero || hopeless) { s32->done = 1; s32->finished = 1; s32->failed = 1; s32->upd_skip[0] = 1; s32->upd_skip[1] = 1; }
        }
        if (bad || zero || hopeless) return;
        inv_c = 1.0 / (c * hdr_inv_s12(g.hA, hB));       // for A in the caller's units
    }

    // ---- helpers on the accumulator layout: register reg of row block rbo = element (32 rbo + rowc(reg) + 4 kg, 32 j + n);
    // as B-operand registers: k-step 2 rbo + (reg >> 3), slot reg & 7
    auto rowc = [](int reg) { return (reg & 3) + 8 * (reg >> 2); };
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
    auto product = [&](const char* buf, const f16x8 (&Bh)[8], const f16x8 (&Bl)[8], f16x8 (&Oh)[8], f16x8 (&Ol)[8], auto f) {
        const f16x8* a = reinterpret_cast<const f16x8*>(buf) + lane;
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
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float t = f(rbo, q, a0[q] + a1[q] * kLoInv);
                _Float16 h, l; split16(t, h, l);
                Oh[2 * rbo + (q >> 3)][q & 7] = h; Ol[2 * rbo + (q >> 3)][q & 7] = l;
            }
            __builtin_amdgcn_sched_barrier(0);               // (keeps the 16 operand reads of the next row block out of this one's registers)
        }
    };
    auto ident = [](int, int, float x) { return x; };

    // ---- k = 0: Y0 = A / c, Z0 = I:  T0 = 1.5 I - 0.5 Y0,  Y1 = Y0 T0,  Z1 = T0
    f16x8 Yh[8], Yl[8], Zh[8], Zl[8], Th[8], Tl[8];
    {
        const double* A64 = adv(g.A64, po) + (int64_t)(4 * kg) * d + 32 * j + n;
#pragma unroll
        for (int rbo = 0; rbo < 4; ++rbo)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float y0 = (float)(A64[(int64_t)(32 * rbo + rowc(reg)) * d] * inv_c);
                asm volatile("" : "+v"(y0));                 // (or the two roundings fuse into an emulated double -> half conversion)
                const float t0 = ((rbo == j && rowc(reg) + 4 * kg == n) ? 1.5f : 0.f) - 0.5f * y0;
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
    for (int k = 1;; ++k) {
        float ss = 0.f;
        product(P, Yh, Yl, Th, Tl, [&](int rbo, int reg, float m) {      // M_j = Z Y_j -> T_j = 1.5 I_j - 0.5 M_j
            const bool dg = rbo == j && rowc(reg) + 4 * kg == n;
            const float e = (dg ? 0.5f : 0.f) - 0.5f * m;               // T - I
            ss += e * e;
            return dg ? e + 1.f : e;
        });
        // r_k = ||I - Z_k Y_k||_F = 2 ||T_k - I||_F, the same for every thread; then the rules of nsf_check
        double s1[1] = {(double)ss};
        wg8_sum<1, 4>(s1, red);                              // (two barriers: every wave is done reading P as Z)
        const double res = 2.0 * sqrt(s1[0]);
        if (tid == 0 && k < 16) s32->res[k] = res;
        const bool finite = (res == res) && !isinf(res);
        if (!finite || k + 1 >= g.max_low || (k >= 4 && res > prev && res > 1e-3)) { failed = 1; final_iter = k; decided_at = k; break; }
        if (res <= 1e-3 && (res > 0.3 * prev || res <= 1e-6)) { ok = 1; final_iter = k; decided_at = k; break; }      // at the floor: Y_k is final
        const double bound = 0.75 * res * res + 0.25 * res * res * res;
        const bool last = bound <= g.thr_pred;               // Y_{k+1} is final
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
    (void)flag;
    if (tid == 0) {
        s32->finished = 1; s32->done = 1; s32->final_iter = final_iter; s32->decided_at = decided_at;
        s32->ok = ok; s32->failed = failed; s32->skip_corr = ok ? 0 : 1;
        s32->upd_skip[0] = 1; s32->upd_skip[1] = 1;
    }
    if (!ok) return;
    __syncthreads();

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
