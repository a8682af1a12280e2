// Running (n, sum x, sum x x^T) of frame matrices E[N x D]  -- gfx950 / CDNA4 only.
//
// Replaces np.mean + np.cov of fadtk/fad.py:48 and the per-file np.cov + merge loop of
// fadtk/utils.py:13-45 by ONE pass over E that accumulates raw moments (sum-reducible).
//
// Kernels
//   moments_tile_h16   fp16/bf16 rows -> fp32 partial tiles of E^T E with
//                      v_mfma_f32_32x32x16_{f16,bf16}.  fp16 x fp16 products are exact in fp32;
//                      each workgroup sums a bounded run of rows in fp32 and the partials are
//                      combined in fp64.  Only upper-triangular 128x128 tiles are computed
//                      (E^T E is symmetric).  Column sums ride along on the diagonal tiles.
//   moments_tile_f64   any dtype / any alignment -> fp64 partial tiles with
//                      v_mfma_f64_16x16x4_f64 (products and sums in fp64, like np.cov).
//   moments_reduce     partials (fp32|fp64) -> += packed fp64 accumulator, mirrored.
//   moments_finalize   (n, sum, sumsq) -> mu, cov with ddof.
//
// Data layout in HBM
//   E            row-major [N x ld], one frame per row (the layout of fadtk's .npy files)
//   accumulator  packed fp64 [ n | sum_x[D] | sum_xxT[D*D] ]
//   partials     [split][tile][BT][BT]   (BT = 128 fp32 | 64 fp64)
//   colpart      [split][nt*BT] fp64
#include "fad_common.h"
#include <dlfcn.h>
#include <type_traits>

// Build-time ablation switches for scripts/probe_ablate.py (never set in the product build): bit 0 drops the
// MFMAs, bit 1 the LDS transpose reads, bit 2 the global->LDS loads, bit 3 the per-stage barrier; bit 4 prints
// per-workgroup clocks, bit 5 makes every split of v4 read the same 256 rows (an L2-resident input).
#ifndef FAD_MOM_ABLATE
#define FAD_MOM_ABLATE 0
#endif
#ifndef FAD_MOM_AUX
#define FAD_MOM_AUX 0          // cache-policy bits of the v8 LDS-DMA loads (probe knob)
#endif
#ifndef FAD_MOM_SPREAD
#define FAD_MOM_SPREAD 0       // v8: issue the LDS-DMA loads between the MFMAs instead of in one burst (probe knob)
#endif

namespace fad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kXcd = 8;

// Workgroup id -> work item such that consecutive items land on the SAME XCD (block b runs on
// XCD b % 8): the tiles of one row-split then share that XCD's L2 for their slabs of E.
__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {
    const int xcd = b % kXcd, idx = b / kXcd;
    const int q = nwg / kXcd, r = nwg % kXcd;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void tile_coords(int tile, int nt, int& ta, int& tb) {
    int a = 0, t = tile;
    while (t >= nt - a) { t -= nt - a; ++a; }
    ta = a; tb = a + t;
}

template <int KIND> __device__ __forceinline__ float h16_to_f32(uint32_t bits16) {
    if constexpr (KIND == FAD_F16) {
        _Float16 h; unsigned short s = (unsigned short)bits16; __builtin_memcpy(&h, &s, 2); return (float)h;
    } else {
        return __uint_as_float(bits16 << 16);
    }
}

template <int KIND> __device__ __forceinline__ float sum8(const uint4& v) {
    // sum of the 8 packed halfs/bfloats in fp32: four v_dot2c_f32_{f16,bf16} against (1, 1) -- the column sums
    // ride on the diagonal tiles' waves, whose VALU time is on the kernel's critical path
    float s = 0.f;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (KIND == FAD_F16) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 a; __builtin_memcpy(&a, &w[q], 4);
            const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
            s = __builtin_amdgcn_fdot2(a, one, s, false);
        } else {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            b2 a, one; __builtin_memcpy(&a, &w[q], 4);
            const uint32_t ob = 0x3f803f80u; __builtin_memcpy(&one, &ob, 4);
            s = __builtin_amdgcn_fdot2_f32_bf16(a, one, s, false);
        }
    }
    return s;
}

template <int KIND> __device__ __forceinline__ float sumsq8(const uint4& v) {
    float s = 0.f;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float lo = h16_to_f32<KIND>(w[q] & 0xffffu), hi = h16_to_f32<KIND>(w[q] >> 16);
        s = fmaf(lo, lo, s); s = fmaf(hi, hi, s);
    }
    return s;
}

template <int KIND> __device__ __forceinline__ f32x16 mfma_h16(const uint4& a, const uint4& b, const f32x16& c) {
    if constexpr (KIND == FAD_F16) {
        f16x8 va, vb; __builtin_memcpy(&va, &a, 16); __builtin_memcpy(&vb, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c, 0, 0, 0);
    } else {
        bf16x8 va, vb; __builtin_memcpy(&va, &a, 16); __builtin_memcpy(&vb, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, c, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// fp16 / bf16 tile kernel.  256 threads = 4 waves as 2x2; workgroup tile 128 x 128 of E^T E,
// wave tile 64 x 64 = 2x2 MFMA 32x32 tiles; 32 rows of E per LDS stage (double buffered).
//
// Fragment trick: an MFMA operand wants 8 consecutive k (rows of E) of ONE column per lane, but E
// is row-major.  The sum over k is order-free and the column<->lane assignment is ours to pick,
// so lane i reads the 32-bit word holding columns (2i, 2i+1) of 8 rows and two v_perm_b32 per row
// pair split them into the fragment of the "even" 32x32 tile (columns 2i) and of the "odd" one
// (columns 2i+1).  Output element (fa, reg, fb) of lane l is then
//   a = 64*wr + 2*row(reg, l>>5) + fa,  b = 64*wc + 2*(l&31) + fb,
// i.e. the two fb values are adjacent columns: one 8-byte store.
// ------------------------------------------------------------------------------------------
constexpr int H_BT = 128;     // tile edge
constexpr int H_TS = H_BT * H_BT + 64;   // partial-tile stride (floats): +256 B so that the same element of
                                         // consecutive tiles/splits does not alias onto one memory channel
constexpr int H_KB = 32;      // rows per stage

template <int KIND>
__global__ __launch_bounds__(256) void moments_tile_h16(
    const uint16_t* __restrict__ E, int64_t n, int64_t ld, int d, int nt, int T, int S,
    int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ colpart) {
    __shared__ uint4 smem[2][2][H_KB * 16];     // [buffer][A|B][row*16 + 16B-chunk]  = 32 KiB

    const int w = xcd_contiguous(blockIdx.x, S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const bool diag = (ta == tb);
    const int ca = ta * H_BT, cb = tb * H_BT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kg = lane >> 5;

    const int64_t k_begin = (int64_t)split * rows_per_split;
    const int64_t k_end = (k_begin + rows_per_split < n) ? k_begin + rows_per_split : n;
    const int nkb = (int)((k_end - k_begin + H_KB - 1) / H_KB);

    // staging: thread -> (row r, r+16 ; 16-byte chunk c) of each slab
    const int sr = tid >> 4, sc = tid & 15;
    const bool col_ok_a = (ca + sc * 8) < d;        // d % 8 == 0 on this path: chunk all-in or all-out
    const bool col_ok_b = (cb + sc * 8) < d;
    const uint16_t* ga = E + ca + sc * 8;
    const uint16_t* gb = E + cb + sc * 8;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    uint4 ra[2], rb[2];
    auto fetch = [&](int kb) {
        const int64_t r0 = k_begin + (int64_t)kb * H_KB + sr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t r = r0 + 16 * h;
            const bool ok = r < k_end;
            ra[h] = (ok && col_ok_a) ? *reinterpret_cast<const uint4*>(ga + r * ld) : zero4;
            if (!diag) rb[h] = (ok && col_ok_b) ? *reinterpret_cast<const uint4*>(gb + r * ld) : zero4;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    double csum[2] = {0.0, 0.0};
    const bool do_colsum = diag && (wr == 0);

    if (nkb > 0) fetch(0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        smem[buf][0][sr * 16 + sc] = ra[0];
        smem[buf][0][(sr + 16) * 16 + sc] = ra[1];
        if (!diag) { smem[buf][1][sr * 16 + sc] = rb[0]; smem[buf][1][(sr + 16) * 16 + sc] = rb[1]; }
        __syncthreads();
        if (kb + 1 < nkb) fetch(kb + 1);

        const uint32_t* sA = reinterpret_cast<const uint32_t*>(smem[buf][0]);
        const uint32_t* sB = reinterpret_cast<const uint32_t*>(smem[buf][diag ? 0 : 1]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rbase = ks * 16 + kg * 8;
            uint32_t wa[8], wb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wa[e] = sA[(rbase + e) * 64 + 32 * wr + li];
                wb[e] = sB[(rbase + e) * 64 + 32 * wc + li];
            }
            uint4 a0, a1, b0, b1;
            // even columns: low halves of consecutive rows; odd columns: high halves
            a0.x = __builtin_amdgcn_perm(wa[1], wa[0], 0x05040100u); a1.x = __builtin_amdgcn_perm(wa[1], wa[0], 0x07060302u);
            a0.y = __builtin_amdgcn_perm(wa[3], wa[2], 0x05040100u); a1.y = __builtin_amdgcn_perm(wa[3], wa[2], 0x07060302u);
            a0.z = __builtin_amdgcn_perm(wa[5], wa[4], 0x05040100u); a1.z = __builtin_amdgcn_perm(wa[5], wa[4], 0x07060302u);
            a0.w = __builtin_amdgcn_perm(wa[7], wa[6], 0x05040100u); a1.w = __builtin_amdgcn_perm(wa[7], wa[6], 0x07060302u);
            b0.x = __builtin_amdgcn_perm(wb[1], wb[0], 0x05040100u); b1.x = __builtin_amdgcn_perm(wb[1], wb[0], 0x07060302u);
            b0.y = __builtin_amdgcn_perm(wb[3], wb[2], 0x05040100u); b1.y = __builtin_amdgcn_perm(wb[3], wb[2], 0x07060302u);
            b0.z = __builtin_amdgcn_perm(wb[5], wb[4], 0x05040100u); b1.z = __builtin_amdgcn_perm(wb[5], wb[4], 0x07060302u);
            b0.w = __builtin_amdgcn_perm(wb[7], wb[6], 0x05040100u); b1.w = __builtin_amdgcn_perm(wb[7], wb[6], 0x07060302u);

            acc[0][0] = mfma_h16<KIND>(a0, b0, acc[0][0]);
            acc[0][1] = mfma_h16<KIND>(a0, b1, acc[0][1]);
            acc[1][0] = mfma_h16<KIND>(a1, b0, acc[1][0]);
            acc[1][1] = mfma_h16<KIND>(a1, b1, acc[1][1]);
            if (do_colsum) {     // wave-uniform; 8-term fp32 sums of 16-bit values, then fp64
                csum[0] += (double)sum8<KIND>(b0);
                csum[1] += (double)sum8<KIND>(b1);
            }
        }
    }

    // ---- epilogue: fp32 partial tile, two adjacent columns per store
    float* out = partials + ((int64_t)split * T + tile) * H_TS;
#pragma unroll
    for (int fa = 0; fa < 2; ++fa) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row32 = (reg & 3) + 8 * (reg >> 2) + 4 * kg;       // C/D row of the 32x32 tile
            const int a_local = 64 * wr + 2 * row32 + fa;
            const int b_local = 64 * wc + 2 * li;
            float2 v = make_float2(acc[fa][0][reg], acc[fa][1][reg]);
            *reinterpret_cast<float2*>(out + a_local * H_BT + b_local) = v;
        }
    }
    if (do_colsum) {
        // lanes l and l+32 hold the two k-halves of the same column
        csum[0] += __shfl_xor(csum[0], 32);
        csum[1] += __shfl_xor(csum[1], 32);
        if (kg == 0) {
            double* cp = colpart + (int64_t)split * (nt * H_BT) + cb + 64 * wc + 2 * li;
            cp[0] = csum[0]; cp[1] = csum[1];
        }
    }
}

// ------------------------------------------------------------------------------------------
// v2 of the fp16/bf16 tile kernel: same tiling and fragment trick, but the slabs of E go
// HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip)
// through a ring of NST stages, so each workgroup keeps NST-1 stages (up to 48 KiB) of loads in
// flight instead of one.  v1 was latency-bound: one 16 KiB stage in flight per workgroup gave
// 0.9 TB/s.  Waits are counted (s_waitcnt vmcnt(N), never 0 in steady state) and the barrier is a
// raw s_barrier so that younger stages stay in flight across it.
// Out-of-range rows / columns are redirected per lane to a 16-byte block of zeros.
// ------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KIND, int NST, bool DIAG>
__device__ __forceinline__ void tile_h16_glds_body(
    const uint16_t* __restrict__ E, int64_t k_begin, int64_t k_end, int64_t ld, int d, int nt, int T,
    int split, int tile, int ca, int cb, float* __restrict__ partials, double* __restrict__ colpart,
    uint4* smem, int* __restrict__ shift_flag) {
    constexpr int LPS = DIAG ? 2 : 4;              // glds instructions per wave per stage
    constexpr int STAGE = 2 * H_KB * 16;           // uint4 per stage (A slab + B slab)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kg = lane >> 5;
    const int nkb = (int)((k_end - k_begin + H_KB - 1) / H_KB);

    const int sr = tid >> 4, sc = tid & 15;
    const bool col_ok_a = (ca + sc * 8) < d;
    const bool col_ok_b = (cb + sc * 8) < d;
    const uint16_t* ga = E + ca + sc * 8;
    const uint16_t* gb = E + cb + sc * 8;
    const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(&g_zero16);

    auto issue = [&](int kb) {
        uint4* st = smem + (kb % NST) * STAGE;
        const int64_t r0 = k_begin + (int64_t)kb * H_KB + sr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t r = r0 + 16 * h;
            const bool ok = r < k_end;
            // LDS destination = wave-uniform base + lane*16: rows 16h + 4*wave .. +3, 16 chunks each
            uint4* dstA = st + 256 * h + 64 * wave;
            const uint16_t* srcA = (ok && col_ok_a) ? ga + r * ld : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)srcA, (lptr_t)dstA, 16, 0, 0);
            if (!DIAG) {
                const uint16_t* srcB = (ok && col_ok_b) ? gb + r * ld : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)srcB, (lptr_t)(dstA + H_KB * 16), 16, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    double csum[2] = {0.0, 0.0};
    const bool do_colsum = DIAG && (wr == wc);     // the diagonal waves also hold sum x^2 (diagonal of acc)

    for (int s = 0; s < NST - 1 && s < nkb; ++s) issue(s);

    for (int kb = 0; kb < nkb; ++kb) {
        // stage kb must have landed; up to NST-2 younger stages may stay in flight
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        if (ahead >= 2) wait_vmcnt<2 * LPS>();
        else if (ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // every wave's pieces of stage kb are in LDS; stage kb-1 is free
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);

        const uint32_t* sA = reinterpret_cast<const uint32_t*>(smem + (kb % NST) * STAGE);
        const uint32_t* sB = DIAG ? sA : sA + H_KB * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rbase = ks * 16 + kg * 8;
            uint32_t wa[8], wb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wa[e] = sA[(rbase + e) * 64 + 32 * wr + li];
                wb[e] = sB[(rbase + e) * 64 + 32 * wc + li];
            }
            uint4 a0, a1, b0, b1;
            a0.x = __builtin_amdgcn_perm(wa[1], wa[0], 0x05040100u); a1.x = __builtin_amdgcn_perm(wa[1], wa[0], 0x07060302u);
            a0.y = __builtin_amdgcn_perm(wa[3], wa[2], 0x05040100u); a1.y = __builtin_amdgcn_perm(wa[3], wa[2], 0x07060302u);
            a0.z = __builtin_amdgcn_perm(wa[5], wa[4], 0x05040100u); a1.z = __builtin_amdgcn_perm(wa[5], wa[4], 0x07060302u);
            a0.w = __builtin_amdgcn_perm(wa[7], wa[6], 0x05040100u); a1.w = __builtin_amdgcn_perm(wa[7], wa[6], 0x07060302u);
            b0.x = __builtin_amdgcn_perm(wb[1], wb[0], 0x05040100u); b1.x = __builtin_amdgcn_perm(wb[1], wb[0], 0x07060302u);
            b0.y = __builtin_amdgcn_perm(wb[3], wb[2], 0x05040100u); b1.y = __builtin_amdgcn_perm(wb[3], wb[2], 0x07060302u);
            b0.z = __builtin_amdgcn_perm(wb[5], wb[4], 0x05040100u); b1.z = __builtin_amdgcn_perm(wb[5], wb[4], 0x07060302u);
            b0.w = __builtin_amdgcn_perm(wb[7], wb[6], 0x05040100u); b1.w = __builtin_amdgcn_perm(wb[7], wb[6], 0x07060302u);
            acc[0][0] = mfma_h16<KIND>(a0, b0, acc[0][0]);
            acc[0][1] = mfma_h16<KIND>(a0, b1, acc[0][1]);
            acc[1][0] = mfma_h16<KIND>(a1, b0, acc[1][0]);
            acc[1][1] = mfma_h16<KIND>(a1, b1, acc[1][1]);
            if (do_colsum) {
                csum[0] += (double)sum8<KIND>(b0);
                csum[1] += (double)sum8<KIND>(b1);
            }
        }
    }

    float* out = partials + ((int64_t)split * T + tile) * H_TS;
#pragma unroll
    for (int fa = 0; fa < 2; ++fa) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row32 = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
            const int a_local = 64 * wr + 2 * row32 + fa;
            const int b_local = 64 * wc + 2 * li;
            *reinterpret_cast<float2*>(out + a_local * H_BT + b_local) = make_float2(acc[fa][0][reg], acc[fa][1][reg]);
        }
    }
    if (do_colsum) {
        csum[0] += __shfl_xor(csum[0], 32);
        csum[1] += __shfl_xor(csum[1], 32);
        if (shift_flag) {
            // Shift guard (see moments_tile_f64): within this run of rows, is any column's mean^2 > 64 var?
            // Then fp32 partial sums of x^2 cannot resolve the variance and the block is redone in fp64.
            // sum x^2 of column (2 li + f) is the diagonal element acc[f][f][reg] of the lane whose C/D row
            // (reg&3) + 8 (reg>>2) + 4 kg equals li: kg = (li>>2)&1, reg = (li&3) + 4 (li>>3).
            const double nr = (double)(k_end - k_begin);
            const int myreg = (li & 3) + 4 * (li >> 3);
            const bool own = kg == ((li >> 2) & 1);
            bool hit = false;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float dsel = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dsel = (r == myreg) ? acc[f][f][r] : dsel;
                double s2 = own ? (double)dsel : 0.0;
                s2 += __shfl_xor(s2, 32);
                const double mean = csum[f] / nr, var = s2 / nr - mean * mean;
                const bool col_in = (cb + 64 * wc + 2 * li + f) < d;
                if (col_in && !(mean * mean <= 64.0 * var) && !(csum[f] == 0.0 && s2 == 0.0)) hit = true;
            }
            if (__any(hit) && lane == 0) atomicOr(shift_flag, 1);
        }
        if (kg == 0) {
            double* cp = colpart + (int64_t)split * (nt * H_BT) + cb + 64 * wc + 2 * li;
            cp[0] = csum[0]; cp[1] = csum[1];
        }
    }
}

template <int KIND, int NST>
__global__ __launch_bounds__(256) void moments_tile_h16_glds(
    const uint16_t* __restrict__ E, int64_t n, int64_t ld, int d, int nt, int T, int S,
    int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ colpart,
    int* __restrict__ shift_flag) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];     // the ONLY LDS object: NST x 16 KiB
    const int w = xcd_contiguous(blockIdx.x, S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const int64_t k_begin = (int64_t)split * rows_per_split;
    const int64_t k_end = (k_begin + rows_per_split < n) ? k_begin + rows_per_split : n;
    if (ta == tb)
        tile_h16_glds_body<KIND, NST, true>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT,
                                            partials, colpart, smem_dyn, shift_flag);
    else
        tile_h16_glds_body<KIND, NST, false>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT,
                                             partials, colpart, smem_dyn, nullptr);
}

// ------------------------------------------------------------------------------------------
// v4 = v2 with the operand fragments read by ds_read_b64_tr_b16 (LDS transpose read): in a 16-lane group lane t
// supplies the address of 4 consecutive columns of row t>>2 and receives 4 consecutive ROWS of column t -- exactly
// the k-contiguous fragment an MFMA wants from a row-major slab.  Two such reads per fragment replace four
// ds_read_b32 + four v_perm_b32, at twice the LDS bytes per clock; v2's LDS read port was as busy as its MFMA pipe.
// (Semantics verified on hardware with scripts/probes/tr_probe.hip.)  Columns map naturally: lane i <-> column i.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NST, bool DIAG, bool MULTI>
__device__ __forceinline__ void tile_h16_tr_body(
    const uint16_t* __restrict__ E, int64_t k_begin, int64_t k_end, int64_t ld, int d, int nt, int T,
    int split, int tile, int ca, int cb, float* __restrict__ partials, double* __restrict__ colpart,
    uint4* smem, int* __restrict__ shift_flag) {
    constexpr int LPS = DIAG ? 2 : 4;              // glds instructions per wave per stage
    constexpr int STAGE = 2 * H_KB * 16;           // uint4 per stage (A slab + B slab)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kg = lane >> 5;
    const int nkb = (int)((k_end - k_begin + H_KB - 1) / H_KB);

    // LDS position (row, chunk p) holds global chunk p ^ 4*(row & 3): the transpose reads of four consecutive rows
    // then fall into the four different 64-byte quarters of the bank space (conflict-free).  The swizzle is applied
    // on the SOURCE address because global_load_lds writes lane-linear; rows sr and sr+16 share (row & 3).
    const int sr = tid >> 4, sc = (tid & 15) ^ (((tid >> 4) & 3) << 2);
    const bool col_ok_a = (ca + sc * 8) < d;
    const bool col_ok_b = (cb + sc * 8) < d;
    const uint16_t* ga = E + ca + sc * 8;
    const uint16_t* gb = E + cb + sc * 8;
    const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(&g_zero16);
    // transpose-read addressing: in each 16-lane group lane t points at (row t>>2, columns 4*(t&3)..+3) of a
    // [4 rows][16 cols] block and receives column t of it (4 consecutive k).  Group g of the wave: rows 8*(g>>1),
    // columns 16*(g&1) of the 32-column fragment.
    const int t16 = lane & 15, grp = lane >> 4;
    const int tr_row = 8 * (grp >> 1) + (t16 >> 2);                 // + ks*16 (+4 for the second read)
    const int tr_col = 16 * (grp & 1) + 4 * (t16 & 3);              // + 32*frag + 64*wave-half, in columns

    // part g of the loads of stage kb: off the diagonal (h, side) = (g >> 1, g & 1), on it h = g (A side only)
    auto issue_part = [&](int kb, int g) {
        if (FAD_MOM_ABLATE & 4) return;
        uint4* st = smem + (kb % NST) * STAGE;
        const int h = DIAG ? g : (g >> 1);
        int64_t r = k_begin + (int64_t)kb * H_KB + sr + 16 * h;
        const bool ok = r < k_end;
        if (FAD_MOM_ABLATE & 32) r &= 255;         // probe: every split reads the same 256 rows (L2-resident)
        // LDS destination = wave-uniform base + lane*16: rows 16h + 4*wave .. +3, 16 chunks each
        const bool side_b = !DIAG && (g & 1);
        const uint16_t* src = side_b ? ((ok && col_ok_b) ? gb + r * ld : zsrc) : ((ok && col_ok_a) ? ga + r * ld : zsrc);
        uint4* dstp = st + 256 * h + 64 * wave + (side_b ? H_KB * 16 : 0);
        if (MULTI) {
            // inline asm like the fast form below: ONE LDS-DMA builtin anywhere in the kernel and hipcc's hazard
            // bookkeeping costs the hot loop its gain.  Per-lane 64-bit addresses (lanes may go to the zero block).
            const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)dstp);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dstp, 16, 0, 0);
        }
    };
    auto issue = [&](int kb) {
#pragma unroll
        for (int g = 0; g < LPS; ++g) issue_part(kb, g);
    };
    // The same loads with a wave-uniform 64-bit base in SGPRs + a loop-invariant 32-bit lane offset
    // (global_load_lds_dwordx4 v_off, s[base:base+1]; inline asm -- the builtin always produces 64-bit VGPR addresses).
    // With per-lane 64-bit addresses a CU does not overlap LDS-DMA with MFMAs: independent loader and MFMA waves take
    // the SUM of their times; with an SGPR base they overlap (scripts/probes/dma_mfma_mix.hip: 0.95 -> 0.56 ms where
    // either alone takes 0.47; scripts/probes/stream_pipeline.hip: this kernel's skeleton 50.7 -> 34.6 us).  Only for
    // stages whose 32 rows and 128 + 128 columns are all in range (no zero-source redirection), and kept in a loop of
    // its own: with the builtin form in the same loop body the gain disappears.
    const bool cols_full = (ca + H_BT <= d) && (cb + H_BT <= d) && ld < ((int64_t)1 << 26);
    const uint32_t voff = (uint32_t)(((int64_t)sr * ld + sc * 8) * 2);
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    auto issue_fast = [&](int kb) {
        if (FAD_MOM_ABLATE & 4) return;
#pragma unroll
        for (int g = 0; g < LPS; ++g) {
            const int h = DIAG ? g : (g >> 1);
            const bool side_b = !DIAG && (g & 1);
            const uint64_t sb = (uint64_t)(E + (k_begin + (int64_t)kb * H_KB + 16 * h) * ld + (side_b ? cb : ca));
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            const uint32_t dst = smem_lds + (uint32_t)(((kb % NST) * STAGE + 256 * h + 64 * wave + (side_b ? H_KB * 16 : 0)) * 16);
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(dst);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    // stages [0, nfast) may be loaded the fast way
    // (MULTI = more than one tile.  A single-tile problem, D <= 128, is HBM-bound and measured slower with either asm
    // form -- 63 / 59 vs 53 us for 1M x 128 -- so it keeps the builtin loads throughout.)
    const int nfast = (MULTI && cols_full && !(FAD_MOM_ABLATE & 32)) ? (int)((k_end - k_begin) / H_KB) : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    double csum[2] = {0.0, 0.0};
    const bool do_colsum = DIAG && (wr == wc);     // the diagonal waves also hold sum x^2 (diagonal of acc)

    for (int s = 0; s < NST - 1 && s < nkb; ++s) { if (s < nfast) issue_fast(s); else issue(s); }

    // fragment of one k-step (16 rows) of a slab: two transpose reads (rows r0..r0+3 and r0+4..r0+7 of the lane's
    // 8-row half) give the 8 consecutive k that the 32x32x16 MFMA wants per lane
    auto frag = [&](const char* slab, int ks, int col0, int kb) -> uint4 {
        if (FAD_MOM_ABLATE & 2) return make_uint4(lane + ks, col0 + kb, lane, 0x3c003c00u);
        // byte address of (row, col): row*256 + ((col/8) ^ 4*(row&3))*16 + ((col/4)&1)*8
        const int r0 = ks * 16 + tr_row, r1 = r0 + 4, col = col0 + tr_col;
        const int o0 = r0 * 256 + (((col >> 3) ^ ((r0 & 3) << 2)) << 4) + ((col >> 2) & 1) * 8;
        const int o1 = r1 * 256 + (((col >> 3) ^ ((r1 & 3) << 2)) << 4) + ((col >> 2) & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + o0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + o1));
        uint4 f;
        __builtin_memcpy(&f.x, &lo, 8);
        __builtin_memcpy(&f.z, &hi, 8);
        return f;
    };
    // F[0], F[1] = the wave's two A-side fragments, F[2], F[3] = its two B-side fragments of k-step (kb, ks)
    auto load_frags = [&](int kb, int ks, uint4 (&F)[4]) {
        const char* sA = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        const char* sB = DIAG ? sA : sA + H_KB * 256;
        F[0] = frag(sA, ks, 64 * wr, kb); F[1] = frag(sA, ks, 64 * wr + 32, kb);
        F[2] = frag(sB, ks, 64 * wc, kb); F[3] = frag(sB, ks, 64 * wc + 32, kb);
    };
    auto mma_first = [&](const uint4 (&F)[4]) {
        if (FAD_MOM_ABLATE & 1) {
            acc[0][0][0] += (float)((F[0].x ^ F[0].y ^ F[0].z ^ F[0].w) + (F[1].x ^ F[1].y ^ F[1].z ^ F[1].w) +
                                    (F[2].x ^ F[2].y ^ F[2].z ^ F[2].w) + (F[3].x ^ F[3].y ^ F[3].z ^ F[3].w));
            return;
        }
        acc[0][0] = mfma_h16<KIND>(F[0], F[2], acc[0][0]);
    };
    auto mma_rest = [&](const uint4 (&F)[4]) {
        if (FAD_MOM_ABLATE & 1) return;
        acc[0][1] = mfma_h16<KIND>(F[0], F[3], acc[0][1]);
        acc[1][0] = mfma_h16<KIND>(F[1], F[2], acc[1][0]);
        acc[1][1] = mfma_h16<KIND>(F[1], F[3], acc[1][1]);
        if (do_colsum) {
            csum[0] += (double)sum8<KIND>(F[2]);
            csum[1] += (double)sum8<KIND>(F[3]);
        }
    };

    // one stage: wait for it, workgroup barrier, refill the freed slot, 16 transpose reads, 8 MFMAs
    auto stage = [&](int kb, auto refill_tag) {
        // stage kb must have landed; up to NST-2 younger stages may stay in flight
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        if (ahead >= 2) wait_vmcnt<2 * LPS>();
        else if (ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        if (!(FAD_MOM_ABLATE & 8)) __builtin_amdgcn_s_barrier();   // stage kb is in LDS; stage kb-1 is free
        if (decltype(refill_tag)::value) issue_fast(kb + NST - 1);
        else if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        // all 16 transpose reads of the stage are issued up front (the compiler waits with lgkmcnt(0) before the
        // first MFMA; software-pipelining the reads one k-step or one stage ahead measured no gain -- DESIGN.md)
        uint4 F0[4], F1[4];
        load_frags(kb, 0, F0);
        load_frags(kb, 1, F1);
        mma_first(F0); mma_rest(F0);
        mma_first(F1); mma_rest(F1);
    };
    // hot loop: the stage to refill is a full one -> SGPR-base loads only; then the tail with the general loads
    const int hot = (nfast - (NST - 1) > 0) ? nfast - (NST - 1) : 0;
    int kb = 0;
    for (; kb < hot; ++kb) stage(kb, std::true_type{});
    for (; kb < nkb; ++kb) stage(kb, std::false_type{});

    // partial tile, fragment major (the layout moments_reduce calls 1): float4 index ((fa*4 + fb)*4 + q)*64 + lane holds
    // registers 4q..4q+3 of the 32 x 32 block (fa, fb) = rows 32fa + 8q + 4(lane>>5) + 0..3 of column 32fb + (lane&31);
    // a wave stores 1 KiB per instruction, 16 instructions instead of 64 scattered dword stores
    float4* out = reinterpret_cast<float4*>(partials + ((int64_t)split * T + tile) * H_TS);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x16& a = acc[x][y];
                out[(((2 * wr + x) * 4 + (2 * wc + y)) * 4 + q) * 64 + lane] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
            }
    if (do_colsum) {
        csum[0] += __shfl_xor(csum[0], 32);
        csum[1] += __shfl_xor(csum[1], 32);
        if (shift_flag) {
            // Shift guard (see moments_tile_f64): within this run of rows, is any column's mean^2 > 64 var?
            // Then fp32 partial sums of x^2 cannot resolve the variance and the block is redone in fp64.
            // sum x^2 of column (32 f + li) is the diagonal element acc[f][f][reg] of the lane whose C/D row
            // (reg&3) + 8 (reg>>2) + 4 kg equals li: kg = (li>>2)&1, reg = (li&3) + 4 (li>>3).
            const double nr = (double)(k_end - k_begin);
            const int myreg = (li & 3) + 4 * (li >> 3);
            const bool own = kg == ((li >> 2) & 1);
            bool hit = false;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float dsel = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dsel = (r == myreg) ? acc[f][f][r] : dsel;
                double s2 = own ? (double)dsel : 0.0;
                s2 += __shfl_xor(s2, 32);
                const double mean = csum[f] / nr, var = s2 / nr - mean * mean;
                const bool col_in = (cb + 64 * wc + 32 * f + li) < d;
                if (col_in && !(mean * mean <= 64.0 * var) && !(csum[f] == 0.0 && s2 == 0.0)) hit = true;
            }
            if (__any(hit) && lane == 0) atomicOr(shift_flag, 1);
        }
        if (kg == 0) {
            double* cp = colpart + (int64_t)split * (nt * H_BT) + cb + 64 * wc + li;
            cp[0] = csum[0]; cp[32] = csum[1];
        }
    }
}

template <int KIND, int NST, bool MULTI>
__global__ __launch_bounds__(256) void moments_tile_h16_tr(
    const uint16_t* __restrict__ E, int64_t n, int64_t ld, int d, int nt, int T, int S,
    int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ colpart,
    int* __restrict__ shift_flag) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];     // the ONLY LDS object: NST x 16 KiB
    long long dbg_c0 = 0, dbg_w0 = 0;
    if (FAD_MOM_ABLATE & 16) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }
    const int w = xcd_contiguous(blockIdx.x, S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const int64_t k_begin = (int64_t)split * rows_per_split;
    const int64_t k_end = (k_begin + rows_per_split < n) ? k_begin + rows_per_split : n;
    if (ta == tb)
        tile_h16_tr_body<KIND, NST, true, MULTI>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT,
                                            partials, colpart, smem_dyn, shift_flag);
    else
        tile_h16_tr_body<KIND, NST, false, MULTI>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT,
                                             partials, colpart, smem_dyn, nullptr);
    if ((FAD_MOM_ABLATE & 16) && threadIdx.x == 0 && blockIdx.x % 97 == 0) {
        const long long c = clock64() - dbg_c0, wt = wall_clock64() - dbg_w0;
        printf("ablate %d pipe 4 block %4d tile %d diag %d: %lld shader cycles, %lld wall ticks (100 MHz) -> %.2f GHz, %.1f us\n",
               FAD_MOM_ABLATE, (int)blockIdx.x, tile, (int)(ta == tb), c, wt, (double)c / (10.0 * (double)wt), (double)wt / 100.0);
    }
}

// ------------------------------------------------------------------------------------------
// v8 ("wave tile"): every wave owns a WHOLE 128 x 128 tile in 256 accumulator registers and streams its own
// rows; the four waves of a workgroup take the 16-row k-steps round robin and their tiles are summed once, at
// the end, through LDS.  Why (all measured, scripts/probes + scripts/probe_ablate.py):
//   * LDS transpose reads run at ~120 B/clk per CU.  With 64 x 64 wave tiles an MFMA needs 1 KiB of LDS reads,
//     which makes the LDS port as busy as the matrix pipe (v4: ~590 vs 544 cycles per 32 rows), and the eight
//     waves of a CU queue on it in lockstep.  A 128 x 128 wave tile needs 0.5 KiB per MFMA.
//   * The partial tiles (one per workgroup) are the kernel's other big cost: 510 workgroups x 64 KiB written,
//     then read by the reduce kernel.  One workgroup per CU halves that.
//   * No workgroup barrier and no LDS sharing in the main loop: a wave waits only on its own LDS-DMA counter.
// Per wave: ring of NSL slots of one k-step (16 rows x 128 columns of the A side, + the B side off the diagonal),
// filled by global_load_lds with the same source-side XOR swizzle as v4; per k-step 16 (8) transpose reads feed
// 16 (10 on a diagonal tile: upper blocks only) MFMAs; the reads of step i+1 are issued right behind the first
// MFMA of step i (the compiler only emits lgkmcnt(0) around ds_read_b64_tr_b16, so that is where a full wait is
// harmless).  Partial tile layout is fragment major: float4 index ((fa*4+fb)*4+q)*64+lane holds registers
// 4q..4q+3 of the 32 x 32 block (fa, fb), i.e. rows 32fa + 8q + 4(lane>>5) + 0..3 of column 32fb + (lane&31).
// ------------------------------------------------------------------------------------------
constexpr int W_RING = 32768;                                  // LDS ring bytes per wave
constexpr int W_LDS = 4 * W_RING + 4 * H_BT * 8 + H_BT * 8;    // + per-wave column sums + their total

template <int N> __device__ __forceinline__ void wait_vmcnt_upto(int outstanding_steps) {
    // s_waitcnt vmcnt(outstanding_steps * N) for outstanding_steps in 0..7 (the count must be an immediate)
    switch (outstanding_steps) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<N>(); break;
        case 2: wait_vmcnt<2 * N>(); break;
        case 3: wait_vmcnt<3 * N>(); break;
        case 4: wait_vmcnt<4 * N>(); break;
        case 5: wait_vmcnt<5 * N>(); break;
        case 6: wait_vmcnt<6 * N>(); break;
        default: wait_vmcnt<7 * N>(); break;
    }
}

template <int KIND, bool DIAG>
__device__ __forceinline__ void tile_h16_wave_body(
    const uint16_t* __restrict__ E, int64_t k_begin, int64_t k_end, int64_t ld, int d, int nt, int T,
    int split, int tile, int ca, int cb, float* __restrict__ partials, double* __restrict__ colpart,
    char* smem, int* __restrict__ shift_flag) {
    constexpr int NSL = DIAG ? 8 : 4;              // ring slots (k-steps in flight + the one being read)
    constexpr int SLOTB = DIAG ? 4096 : 8192;      // bytes per slot
    constexpr int LPS = DIAG ? 4 : 8;              // LDS-DMA instructions per k-step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kg = lane >> 5;
    const int nks = (int)((k_end - k_begin + 15) / 16);
    const int nw = (nks > wave) ? (nks - wave + 3) / 4 : 0;          // this wave's k-steps: wave, wave + 4, ...
    char* ring = smem + wave * W_RING;

    // LDS-DMA: one instruction = 4 rows x 256 B; lane -> row lane>>4, 16-byte chunk (lane&15) ^ 4*(row&3)
    const int srow = lane >> 4, sc = (lane & 15) ^ (srow << 2);
    const bool col_ok_a = (ca + sc * 8) < d;
    const bool col_ok_b = (cb + sc * 8) < d;
    const uint16_t* ga = E + ca + sc * 8;
    const uint16_t* gb = E + cb + sc * 8;
    const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(&g_zero16);
    // part g of the loads of own k-step i: rows 4g..4g+3 of the A slab (g < 4) or of the B slab (g >= 4)
    auto issue_part = [&](int i, int g) {
        if (FAD_MOM_ABLATE & 4) return;
        char* slot = ring + (i % NSL) * SLOTB;
        const int h = g & 3;
        const int64_t r = k_begin + (int64_t)(wave + 4 * i) * 16 + srow + 4 * h;
        const bool ok = r < k_end;
        if (g < 4) {
            const uint16_t* srcA = (ok && col_ok_a) ? ga + r * ld : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)srcA, (lptr_t)(slot + h * 1024), 16, 0, FAD_MOM_AUX);
        } else {
            const uint16_t* srcB = (ok && col_ok_b) ? gb + r * ld : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)srcB, (lptr_t)(slot + 4096 + h * 1024), 16, 0, FAD_MOM_AUX);
        }
    };
    auto issue = [&](int i) {
#pragma unroll
        for (int g = 0; g < LPS; ++g) issue_part(i, g);
    };

    // transpose-read addressing (see v4): byte offset of the lane's first read of 32-column fragment f
    const int t16 = lane & 15, grp = lane >> 4;
    const int tr_row = 8 * (grp >> 1) + (t16 >> 2);
    const int tr_col = 16 * (grp & 1) + 4 * (t16 & 3);
    int fo[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int col = 32 * f + tr_col;
        fo[f] = tr_row * 256 + (((col >> 3) ^ ((tr_row & 3) << 2)) << 4) + ((col >> 2) & 1) * 8;
    }
    auto frag = [&](const char* slab, int f) -> uint4 {
        if (FAD_MOM_ABLATE & 2) return make_uint4(lane + f, (uint32_t)(size_t)slab, lane, 0x3c003c00u);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + fo[f]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + fo[f] + 1024));
        uint4 v;
        __builtin_memcpy(&v.x, &lo, 8);
        __builtin_memcpy(&v.z, &hi, 8);
        return v;
    };
    constexpr int NFR = DIAG ? 4 : 8;              // fragments per k-step: A side 0..3 (+ B side 4..7)
    auto load_frags = [&](int i, uint4 (&F)[NFR]) {
        const char* slot = ring + (i % NSL) * SLOTB;
#pragma unroll
        for (int f = 0; f < 4; ++f) F[f] = frag(slot, f);
        if (!DIAG) {
#pragma unroll
            for (int f = 0; f < 4; ++f) F[4 + f] = frag(slot + 4096, f);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    double csum[4] = {0.0, 0.0, 0.0, 0.0};

    auto mma_first = [&](const uint4 (&F)[NFR]) {
        if (FAD_MOM_ABLATE & 1) { acc[0][0][0] += (float)(F[0].x ^ F[1].y ^ F[2].z ^ F[3].w ^ F[NFR - 1].x); return; }
        acc[0][0] = mfma_h16<KIND>(F[0], F[DIAG ? 0 : 4], acc[0][0]);
    };
    auto mma_rest = [&](const uint4 (&F)[NFR], int refill) {      // refill: own k-step whose loads ride along, or -1
        int m = 0;
#pragma unroll
        for (int fa = 0; fa < 4; ++fa)
#pragma unroll
            for (int fb = (DIAG ? fa : 0); fb < 4; ++fb) {
                if (fa == 0 && fb == 0) continue;
                if (!(FAD_MOM_ABLATE & 1)) acc[fa][fb] = mfma_h16<KIND>(F[fa], F[DIAG ? fb : 4 + fb], acc[fa][fb]);
                if (FAD_MOM_SPREAD && m < LPS) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (refill >= 0) issue_part(refill, m);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ++m;
            }
        if (DIAG && !(FAD_MOM_ABLATE & 1)) {
#pragma unroll
            for (int f = 0; f < 4; ++f) csum[f] += (double)sum8<KIND>(F[f]);
        }
    };

    const int n0 = nw < NSL ? nw : NSL;
    for (int s = 0; s < n0; ++s) issue(s);
    uint4 C[NFR], N[NFR];                          // fragments of the current / the next k-step
    if (nw > 0) {
        if (!(FAD_MOM_ABLATE & 4)) wait_vmcnt_upto<LPS>(n0 - 1);
        load_frags(0, C);
    }
    for (int i = 0; i < nw; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        mma_first(C);                              // (lgkmcnt(0) before it: every read of step i has landed)
        __builtin_amdgcn_sched_barrier(0);
        if (!FAD_MOM_SPREAD && i + NSL < nw) issue(i + NSL);          // ... so its slot can be refilled
        if (i + 1 < nw) {
            const int newest = FAD_MOM_SPREAD ? i + NSL - 1 : i + NSL;
            const int youngest = (nw - 1 < newest) ? nw - 1 : newest;
            if (!(FAD_MOM_ABLATE & 4)) wait_vmcnt_upto<LPS>(youngest - (i + 1));      // step i+1 is in LDS
            load_frags(i + 1, N);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_rest(C, (FAD_MOM_SPREAD && i + NSL < nw) ? i + NSL : -1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NFR; ++f) C[f] = N[f];
    }

    // ---- epilogue: sum the four waves' tiles (fp64 sum of four fp32 values, rounded once) and store ------
    __syncthreads();                               // every wave is done with its ring
    float4* xch = reinterpret_cast<float4*>(smem);                   // [wave][2048] per half
    double* colx = reinterpret_cast<double*>(smem + 4 * W_RING);     // [4][128] per-wave column sums
    double* colt = colx + 4 * H_BT;                                  // [128] their total
    if (DIAG) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            csum[f] += __shfl_xor(csum[f], 32);
            if (kg == 0) colx[wave * H_BT + 32 * f + li] = csum[f];
        }
        __syncthreads();
        if (tid < H_BT) {
            const double t = (colx[tid] + colx[H_BT + tid]) + (colx[2 * H_BT + tid] + colx[3 * H_BT + tid]);
            colt[tid] = t;
            colpart[(int64_t)split * (nt * H_BT) + cb + tid] = t;
        }
    }
    float4* out = reinterpret_cast<float4*>(partials + ((int64_t)split * T + tile) * H_TS);
    const double nr = (double)(k_end - k_begin);
    bool hit = false;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int fl = 0; fl < 2; ++fl)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x16& a = acc[2 * half + fl][fb];
                    xch[wave * 2048 + ((fl * 4 + fb) * 4 + q) * 64 + lane] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
                }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = tid + 256 * j;
            const float4 v0 = xch[e], v1 = xch[2048 + e], v2 = xch[4096 + e], v3 = xch[6144 + e];
            const double s0 = ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
            const double s1 = ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
            const double s2 = ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
            const double s3 = ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
            out[half * 2048 + e] = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
            if (DIAG && shift_flag) {
                // Shift guard (see moments_tile_f64): within this run of rows, is any column's mean^2 > 64 var?  The
                // float4 of lane l, quad q of a diagonal block holds rows 8q + 4(l>>5) + 0..3 of column l&31: it
                // contains the diagonal element (sum of x^2 of that column) iff (l&31)>>2 == 2q + (l>>5).
                const int el = e & 63, eq = (e >> 6) & 3, efb = (e >> 8) & 3, efa = 2 * half + (e >> 10);
                const int eli = el & 31;
                if (efa == efb && (eli >> 2) == 2 * eq + (el >> 5)) {
                    const int c = eli & 3, col = 32 * efb + eli;
                    const double sq = (c == 0) ? s0 : (c == 1) ? s1 : (c == 2) ? s2 : s3;
                    const double cs = colt[col];
                    const double mean = cs / nr, var = sq / nr - mean * mean;
                    if ((cb + col) < d && !(mean * mean <= 64.0 * var) && !(cs == 0.0 && sq == 0.0)) hit = true;
                }
            }
        }
        __syncthreads();                           // before the second half overwrites the exchange buffer
    }
    if (DIAG && shift_flag && hit) atomicOr(shift_flag, 1);
}

template <int KIND>
__global__ __launch_bounds__(256) void moments_tile_h16_wave(
    const uint16_t* __restrict__ E, int64_t n, int64_t ld, int d, int nt, int T, int S,
    int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ colpart,
    int* __restrict__ shift_flag) {
    extern __shared__ __attribute__((aligned(16))) char smem_wave[];   // the ONLY LDS object: W_LDS bytes
    long long dbg_c0 = 0, dbg_w0 = 0;
    if (FAD_MOM_ABLATE & 16) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }
    const int w = xcd_contiguous(blockIdx.x, S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const int64_t k_begin = (int64_t)split * rows_per_split;
    const int64_t k_end = (k_begin + rows_per_split < n) ? k_begin + rows_per_split : n;
    if (ta == tb)
        tile_h16_wave_body<KIND, true>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT, partials,
                                       colpart, smem_wave, shift_flag);
    else
        tile_h16_wave_body<KIND, false>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT, partials,
                                        colpart, smem_wave, nullptr);
    if ((FAD_MOM_ABLATE & 16) && threadIdx.x == 0 && blockIdx.x % 47 == 0) {
        const long long c = clock64() - dbg_c0, wt = wall_clock64() - dbg_w0;
        printf("ablate %d pipe 8 block %4d tile %d diag %d: %lld shader cycles, %lld wall ticks (100 MHz) -> %.2f GHz, %.1f us\n",
               FAD_MOM_ABLATE, (int)blockIdx.x, tile, (int)(ta == tb), c, wt, (double)c / (10.0 * (double)wt), (double)wt / 100.0);
    }
}

// ------------------------------------------------------------------------------------------
// v3: same 128 x 128 tile, same LDS ring, but TWO waves per workgroup, each owning 128 (A side) x 64
// (B side) = 4 x 2 MFMA tiles.  The A fragments come from ds_read_b64 (lane i reads columns 4i..4i+3
// of 8 rows -> four fragments), the B fragments from ds_read_b32 as before: 16 LDS reads + 24 v_perm
// feed 8 MFMAs instead of 16 + 16 feeding 4.  v2 saturated the LDS read port (8 waves x 16 reads per
// 128 MFMA cycles); here a CU runs 4 such waves (2 workgroups), one per SIMD.
// ------------------------------------------------------------------------------------------
template <int KIND, int NST, bool DIAG>
__device__ __forceinline__ void tile_h16_w2_body(
    const uint16_t* __restrict__ E, int64_t k_begin, int64_t k_end, int64_t ld, int d, int nt, int T,
    int split, int tile, int ca, int cb, float* __restrict__ partials, double* __restrict__ colpart,
    uint4* smem, int* __restrict__ shift_flag) {
    constexpr int LPS = DIAG ? 4 : 8;              // glds instructions per wave per stage
    constexpr int STAGE = 2 * H_KB * 16;           // uint4 per stage (A slab + B slab)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = B-side half
    const int li = lane & 31, kg = lane >> 5;
    const int nkb = (int)((k_end - k_begin + H_KB - 1) / H_KB);

    const int sr = tid >> 4, sc = tid & 15;        // staging: rows sr + 8h, 16-byte chunk sc
    const bool col_ok_a = (ca + sc * 8) < d;
    const bool col_ok_b = (cb + sc * 8) < d;
    const uint16_t* ga = E + ca + sc * 8;
    const uint16_t* gb = E + cb + sc * 8;
    const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(&g_zero16);

    auto issue = [&](int kb) {
        uint4* st = smem + (kb % NST) * STAGE;
        const int64_t r0 = k_begin + (int64_t)kb * H_KB + sr;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int64_t r = r0 + 8 * h;
            const bool ok = r < k_end;
            uint4* dstA = st + (8 * h + 4 * wc) * 16;            // wave-uniform base; + lane*16 B by the hardware
            const uint16_t* srcA = (ok && col_ok_a) ? ga + r * ld : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)srcA, (lptr_t)dstA, 16, 0, 0);
            if (!DIAG) {
                const uint16_t* srcB = (ok && col_ok_b) ? gb + r * ld : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)srcB, (lptr_t)(dstA + H_KB * 16), 16, 0, 0);
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    double csum[2] = {0.0, 0.0};

    for (int s = 0; s < NST - 1 && s < nkb; ++s) issue(s);

    for (int kb = 0; kb < nkb; ++kb) {
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        if (ahead >= 2) wait_vmcnt<2 * LPS>();
        else if (ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);

        const uint2* sA = reinterpret_cast<const uint2*>(smem + (kb % NST) * STAGE);
        const uint32_t* sB = reinterpret_cast<const uint32_t*>(smem + (kb % NST) * STAGE + (DIAG ? 0 : H_KB * 16));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rbase = ks * 16 + kg * 8;
            uint2 wa[8];
            uint32_t wb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wa[e] = sA[(rbase + e) * 32 + li];               // columns 4 li .. 4 li + 3 of row rbase + e
                wb[e] = sB[(rbase + e) * 64 + 32 * wc + li];     // columns 64 wc + 2 li, + 1
            }
            uint4 a[4], b[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x0 = wa[2 * q].x, x1 = wa[2 * q + 1].x, y0 = wa[2 * q].y, y1 = wa[2 * q + 1].y;
                const uint32_t f0 = __builtin_amdgcn_perm(x1, x0, 0x05040100u), f1 = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                const uint32_t f2 = __builtin_amdgcn_perm(y1, y0, 0x05040100u), f3 = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
                const uint32_t g0 = __builtin_amdgcn_perm(wb[2 * q + 1], wb[2 * q], 0x05040100u);
                const uint32_t g1 = __builtin_amdgcn_perm(wb[2 * q + 1], wb[2 * q], 0x07060302u);
                if (q == 0) { a[0].x = f0; a[1].x = f1; a[2].x = f2; a[3].x = f3; b[0].x = g0; b[1].x = g1; }
                if (q == 1) { a[0].y = f0; a[1].y = f1; a[2].y = f2; a[3].y = f3; b[0].y = g0; b[1].y = g1; }
                if (q == 2) { a[0].z = f0; a[1].z = f1; a[2].z = f2; a[3].z = f3; b[0].z = g0; b[1].z = g1; }
                if (q == 3) { a[0].w = f0; a[1].w = f1; a[2].w = f2; a[3].w = f3; b[0].w = g0; b[1].w = g1; }
            }
#pragma unroll
            for (int fa = 0; fa < 4; ++fa) {
                acc[fa][0] = mfma_h16<KIND>(a[fa], b[0], acc[fa][0]);
                acc[fa][1] = mfma_h16<KIND>(a[fa], b[1], acc[fa][1]);
            }
            if (DIAG) {
                csum[0] += (double)sum8<KIND>(b[0]);
                csum[1] += (double)sum8<KIND>(b[1]);
            }
        }
    }

    float* out = partials + ((int64_t)split * T + tile) * H_TS;
#pragma unroll
    for (int fa = 0; fa < 4; ++fa) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row32 = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
            const int a_local = 4 * row32 + fa;
            const int b_local = 64 * wc + 2 * li;
            *reinterpret_cast<float2*>(out + a_local * H_BT + b_local) = make_float2(acc[fa][0][reg], acc[fa][1][reg]);
        }
    }
    if (DIAG) {
        csum[0] += __shfl_xor(csum[0], 32);
        csum[1] += __shfl_xor(csum[1], 32);
        if (shift_flag) {
            // sum x^2 of column b = 64 wc + 2 li + f is the accumulator element with a_local == b:
            // fa = b & 3, C/D row r = b >> 2 = 16 wc + (li >> 1), held (for C/D column li) by kg = (r>>2)&1, reg = (r&3) + 4 (r>>3)
            const double nr = (double)(k_end - k_begin);
            const int r = 16 * wc + (li >> 1);
            const int myreg = (r & 3) + 4 * (r >> 3);
            const bool own = kg == ((r >> 2) & 1);
            bool hit = false;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int fa_need = 2 * (li & 1) + f;
                float dsel = 0.f;
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int q = 0; q < 16; ++q) dsel = (x == fa_need && q == myreg) ? acc[x][f][q] : dsel;
                double s2 = own ? (double)dsel : 0.0;
                s2 += __shfl_xor(s2, 32);
                const double mean = csum[f] / nr, var = s2 / nr - mean * mean;
                const bool col_in = (cb + 64 * wc + 2 * li + f) < d;
                if (col_in && !(mean * mean <= 64.0 * var) && !(csum[f] == 0.0 && s2 == 0.0)) hit = true;
            }
            if (__any(hit) && lane == 0) atomicOr(shift_flag, 1);
        }
        if (kg == 0) {
            double* cp = colpart + (int64_t)split * (nt * H_BT) + cb + 64 * wc + 2 * li;
            cp[0] = csum[0]; cp[1] = csum[1];
        }
    }
}

template <int KIND, int NST>
__global__ __launch_bounds__(128) void moments_tile_h16_w2(
    const uint16_t* __restrict__ E, int64_t n, int64_t ld, int d, int nt, int T, int S,
    int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ colpart,
    int* __restrict__ shift_flag) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];
    const int w = xcd_contiguous(blockIdx.x, S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const int64_t k_begin = (int64_t)split * rows_per_split;
    const int64_t k_end = (k_begin + rows_per_split < n) ? k_begin + rows_per_split : n;
    if (ta == tb)
        tile_h16_w2_body<KIND, NST, true>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT,
                                          partials, colpart, smem_dyn, shift_flag);
    else
        tile_h16_w2_body<KIND, NST, false>(E, k_begin, k_end, ld, d, nt, T, split, tile, ta * H_BT, tb * H_BT,
                                           partials, colpart, smem_dyn, nullptr);
}

// ------------------------------------------------------------------------------------------
// Shift guard.  The fp16 pass sums exact products in fp32 over bounded runs; that is accurate
// relative to sum x^2, not to the variance.  For a column with |mean| >> std (constant-ish features,
// outlier dimensions of transformer states) the covariance is a small difference of large sums, so
// the update is REDONE exactly (fp64 products and sums, like np.cov's centred dsyrk) when any column
// has mean^2 > 64 var within some workgroup's run of rows.  No host round trip: the fp64 tile kernel
// is launched unconditionally and exits at once when the flag is clear; one reduce launch serves both
// sources.  Two flags alternate between updates so that the reduce of update k can clear the flag of
// update k+1 without a memset.
// ------------------------------------------------------------------------------------------
// (the test itself lives in the epilogue of the diagonal-tile workgroups of moments_tile_h16_glds)

// ------------------------------------------------------------------------------------------
// Generic tile kernel: any input dtype, any pitch/alignment.  Everything in fp64 on
// v_mfma_f64_16x16x4_f64 (A: lane l holds A[i=l&15][k=l>>4]; B[k=l>>4][j=l&15];
// D: col = l&15, row = (l>>4) + 4*reg).  Workgroup tile 64x64, wave tile 32x32, 16 rows/stage.
// ------------------------------------------------------------------------------------------
constexpr int G_BT = 64;
constexpr int G_TS = G_BT * G_BT + 32;   // partial-tile stride (doubles), +256 B as for H_TS
constexpr int G_KB = 16;
constexpr int G_LDS = 80;     // padded row pitch (doubles): consecutive k rows hit the other bank half

template <typename TIn> __device__ __forceinline__ double to_f64(TIn v);
template <> __device__ __forceinline__ double to_f64<double>(double v) { return v; }
template <> __device__ __forceinline__ double to_f64<float>(float v) { return (double)v; }
struct raw_f16 { uint16_t b; };
struct raw_bf16 { uint16_t b; };
template <> __device__ __forceinline__ double to_f64<raw_f16>(raw_f16 v) { return (double)h16_to_f32<FAD_F16>(v.b); }
template <> __device__ __forceinline__ double to_f64<raw_bf16>(raw_bf16 v) { return (double)h16_to_f32<FAD_BF16>(v.b); }

template <typename TIn>
__global__ __launch_bounds__(256) void moments_tile_f64(
    const TIn* __restrict__ E, int64_t n, int64_t ld, int d, int nt, int T, int S,
    int64_t rows_per_split, double* __restrict__ partials, double* __restrict__ colpart,
    const int* __restrict__ gate) {
    __shared__ double smem[2][2][G_KB * G_LDS];      // 40 KiB
    if (gate && *gate == 0) return;                  // shift guard: only runs when the fp16 pass flagged the block

    const int w = xcd_contiguous(blockIdx.x, S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const bool diag = (ta == tb);
    const int ca = ta * G_BT, cb = tb * G_BT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;

    const int64_t k_begin = (int64_t)split * rows_per_split;
    const int64_t k_end = (k_begin + rows_per_split < n) ? k_begin + rows_per_split : n;
    const int nkb = (int)((k_end - k_begin + G_KB - 1) / G_KB);

    const int sr = tid >> 4, sc4 = (tid & 15) * 4;
    double ra[4], rb[4];
    auto fetch = [&](int kb) {
        const int64_t r = k_begin + (int64_t)kb * G_KB + sr;
        const bool ok = r < k_end;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int colA = ca + sc4 + q, colB = cb + sc4 + q;
            ra[q] = (ok && colA < d) ? to_f64<TIn>(E[r * ld + colA]) : 0.0;
            if (!diag) rb[q] = (ok && colB < d) ? to_f64<TIn>(E[r * ld + colB]) : 0.0;
        }
    };

    f64x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double csum[2] = {0.0, 0.0};
    const bool do_colsum = diag && (wr == 0);

    if (nkb > 0) fetch(0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            smem[buf][0][sr * G_LDS + sc4 + q] = ra[q];
            if (!diag) smem[buf][1][sr * G_LDS + sc4 + q] = rb[q];
        }
        __syncthreads();
        if (kb + 1 < nkb) fetch(kb + 1);
        const double* sA = smem[buf][0];
        const double* sB = smem[buf][diag ? 0 : 1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 4 + lk;
            double a[2], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f] = sA[k * G_LDS + 32 * wr + 16 * f + li];
                b[f] = sB[k * G_LDS + 32 * wc + 16 * f + li];
            }
#pragma unroll
            for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
            if (do_colsum) { csum[0] += b[0]; csum[1] += b[1]; }
        }
    }

    double* out = partials + ((int64_t)split * T + tile) * G_TS;
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int a_local = 32 * wr + 16 * fa + lk + 4 * reg;
                const int b_local = 32 * wc + 16 * fb + li;
                out[a_local * G_BT + b_local] = acc[fa][fb][reg];
            }
    if (do_colsum) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            csum[f] += __shfl_xor(csum[f], 16);
            csum[f] += __shfl_xor(csum[f], 32);
        }
        if (lk == 0) {
            double* cp = colpart + (int64_t)split * (nt * G_BT) + cb + 32 * wc + li;
            cp[0] = csum[0]; cp[16] = csum[1];
        }
    }
}

// ------------------------------------------------------------------------------------------
// partials -> packed fp64 accumulator (sum over splits in fp64, fixed order => deterministic).
// One thread per 4 adjacent columns of one tile row.
// ------------------------------------------------------------------------------------------
struct SplitPlan { int nt, T, S; int64_t rows_per_split; };

// One source of partial sums for moments_reduce: `S` row-splits x `T` tiles (+ column partials).
struct ReduceSrc {
    const void* partials; const double* colpart;
    int S, T, nt;
    int tile_blocks;      // workgroups that sum tiles; the following ceil(d/256) sum the columns and the row count
    int layout;           // 0 row major, 1 fragment major (moments_tile_h16_wave)
    int sl;               // "split lanes" (1, 4 or 16), see below
};

// sl "split lanes" share one output group: thread (l, g) sums splits l, l+sl, ... and the sl partial
// sums are combined through LDS in a fixed order.  With hundreds of row-splits (D = 128 uses every
// workgroup slot for one tile) a single thread per output would walk all of them serially.
// overwrite: the accumulator was reset since its last update -- store instead of add (saves the memset).
template <typename PT, int BT>
__device__ __forceinline__ void reduce_body(const ReduceSrc& r, int d, double* __restrict__ acc_packed, double n_add,
                                            bool overwrite, int block, double* red) {
    const PT* __restrict__ partials = static_cast<const PT*>(r.partials);
    const int S = r.S, T = r.T, nt = r.nt, SL = r.sl;
    const int G = 256 / SL;                        // output groups (4 values each) per block
    const int per_tile = BT * BT / 4;
    if (block >= r.tile_blocks) {                  // trailing blocks: column sums and the row count
        const int a = (block - r.tile_blocks) * 256 + threadIdx.x;
        if (a == 0) acc_packed[0] = overwrite ? n_add : acc_packed[0] + n_add;
        if (a >= d) return;
        const int dpad = nt * BT;
        const double* __restrict__ colpart = r.colpart;
        double s0 = 0.0, s1 = 0.0;
        int sp = 0;
        for (; sp + 1 < S; sp += 2) { s0 += colpart[(int64_t)sp * dpad + a]; s1 += colpart[(int64_t)(sp + 1) * dpad + a]; }
        if (sp < S) s0 += colpart[(int64_t)sp * dpad + a];
        acc_packed[1 + a] = overwrite ? s0 + s1 : acc_packed[1 + a] + (s0 + s1);
        return;
    }
    const int sl = threadIdx.x / G, gl = threadIdx.x % G;
    const int64_t g = (int64_t)block * G + gl;
    const bool live = g < (int64_t)T * per_tile;
    int tile = 0, a_local = 0, b_local = 0;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (live) {
        tile = (int)(g / per_tile);
        const int e = (int)(g - (int64_t)tile * per_tile);
        if (r.layout == 0) {               // row major: 4 adjacent columns of one row
            a_local = e / (BT / 4); b_local = (e % (BT / 4)) * 4;
        } else {                           // fragment major (moments_tile_h16_wave): 4 adjacent ROWS of one column
            const int el = e & 63;
            a_local = 32 * (e >> 10) + 8 * ((e >> 6) & 3) + 4 * (el >> 5);
            b_local = 32 * ((e >> 8) & 3) + (el & 31);
        }
        constexpr int TS = (sizeof(PT) == 4) ? BT * BT + 64 : BT * BT + 32;      // H_TS / G_TS
        const PT* p = partials + (int64_t)tile * TS + e * 4;
        const int64_t stride = (int64_t)T * TS;
        int sp = sl;
        if constexpr (sizeof(PT) == 4) {           // four independent loads in flight per thread
            for (; sp + 3 * SL < S; sp += 4 * SL) {
                const float4 v0 = *reinterpret_cast<const float4*>(p + sp * stride);
                const float4 v1 = *reinterpret_cast<const float4*>(p + (sp + SL) * stride);
                const float4 v2 = *reinterpret_cast<const float4*>(p + (sp + 2 * SL) * stride);
                const float4 v3 = *reinterpret_cast<const float4*>(p + (sp + 3 * SL) * stride);
                s[0] += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
                s[1] += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
                s[2] += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
                s[3] += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
            }
        }
        for (; sp < S; sp += SL) {
            if constexpr (sizeof(PT) == 4) {
                const float4 v = *reinterpret_cast<const float4*>(p + sp * stride);
                s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            } else {
                const double2 v0 = *reinterpret_cast<const double2*>(p + sp * stride);
                const double2 v1 = *reinterpret_cast<const double2*>(p + sp * stride + 2);
                s[0] += v0.x; s[1] += v0.y; s[2] += v1.x; s[3] += v1.y;
            }
        }
    }
    if (SL > 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(sl * G + gl) * 4 + q] = s[q];
        __syncthreads();
        if (sl != 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double t = 0.0;
            for (int l = 0; l < SL; ++l) t += red[(l * G + gl) * 4 + q];
            s[q] = t;
        }
    }
    if (!live) return;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    double* M = acc_packed + 1 + d;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int a = ta * BT + a_local + (r.layout ? q : 0), b = tb * BT + b_local + (r.layout ? 0 : q);
        if (a >= d || b >= d) continue;
        const int64_t ab = (int64_t)a * d + b, ba = (int64_t)b * d + a;
        if (ta != tb) {
            M[ab] = overwrite ? s[q] : M[ab] + s[q];
            M[ba] = overwrite ? s[q] : M[ba] + s[q];
        } else if (a <= b) {            // diagonal tile: upper triangle is authoritative
            M[ab] = overwrite ? s[q] : M[ab] + s[q];
            if (a != b) M[ba] = overwrite ? s[q] : M[ba] + s[q];
        }
    }
}

// accumulator += (or =) the sum over splits of ONE of two sources: `prim` (PTA, BTA) when *gate == 0 or there is no
// gate, else `alt` -- the fp64 redo of the block by moments_tile_f64 (shift guard).  One launch for both cases; the
// grid is sized for the larger.
template <typename PTA, int BTA>
__global__ __launch_bounds__(256) void moments_reduce(ReduceSrc prim, ReduceSrc alt, int d, double* __restrict__ acc_packed,
                                                      double n_add, const int* __restrict__ gate,
                                                      int* __restrict__ clear_flag, int overwrite) {
    __shared__ double red[256 * 4];
    if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *clear_flag = 0;     // next update's flag
    if (gate && *gate != 0) reduce_body<double, 64>(alt, d, acc_packed, n_add, overwrite != 0, (int)blockIdx.x, red);
    else reduce_body<PTA, BTA>(prim, d, acc_packed, n_add, overwrite != 0, (int)blockIdx.x, red);
}

// Stage 1 of the two-level reduce used when an update produced hundreds or thousands of partial tiles (long inputs
// at small D: every run of <= 8192 rows is one split).  Block (x, c) sums the splits of chunk c for 256 output
// groups (four loads in flight per thread) into fp64 partials laid out like moments_reduce<double, BT> expects;
// trailing x-blocks do the same for the column partials.
constexpr int PRESUM_CHUNK = 32;
template <int BT>
__global__ __launch_bounds__(256) void moments_presum(
    const float* __restrict__ partials, const double* __restrict__ colpart, int S, int T, int nt, int group_blocks,
    double* __restrict__ partials2, double* __restrict__ colpart2, const int* __restrict__ gate) {
    if (gate && *gate != 0) return;                // the block is being redone in fp64: nothing to pre-sum
    const int c = blockIdx.y;
    const int s0 = c * PRESUM_CHUNK, s1 = (s0 + PRESUM_CHUNK < S) ? s0 + PRESUM_CHUNK : S;
    const int dpad = nt * BT;
    if ((int)blockIdx.x >= group_blocks) {
        const int a = ((int)blockIdx.x - group_blocks) * 256 + threadIdx.x;
        if (a >= dpad) return;
        double t = 0.0;
        for (int sp = s0; sp < s1; ++sp) t += colpart[(int64_t)sp * dpad + a];
        colpart2[(int64_t)c * dpad + a] = t;
        return;
    }
    constexpr int per_tile = BT * BT / 4;
    constexpr int TS32 = BT * BT + 64, TS64 = BT * BT + 32;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)T * per_tile) return;
    const int tile = (int)(g / per_tile), e = (int)(g - (int64_t)tile * per_tile);
    const float* p = partials + (int64_t)tile * TS32 + e * 4;
    const int64_t stride = (int64_t)T * TS32;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int sp = s0;
    for (; sp + 3 < s1; sp += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + sp * stride);
        const float4 v1 = *reinterpret_cast<const float4*>(p + (sp + 1) * stride);
        const float4 v2 = *reinterpret_cast<const float4*>(p + (sp + 2) * stride);
        const float4 v3 = *reinterpret_cast<const float4*>(p + (sp + 3) * stride);
        s[0] += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s[1] += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
        s[2] += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
        s[3] += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
    }
    for (; sp < s1; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(p + sp * stride);
        s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
    }
    double* o = partials2 + ((int64_t)c * T + tile) * TS64 + e * 4;
    *reinterpret_cast<double2*>(o) = make_double2(s[0], s[1]);
    *reinterpret_cast<double2*>(o + 2) = make_double2(s[2], s[3]);
}

static ReduceSrc reduce_src(const void* part, const double* colp, const SplitPlan& p, int bt, int layout) {
    ReduceSrc r;
    r.partials = part; r.colpart = colp; r.S = p.S; r.T = p.T; r.nt = p.nt; r.layout = layout;
    r.sl = (p.S > 64) ? 16 : (p.S > 8) ? 4 : 1;
    r.tile_blocks = (int)cdiv((int64_t)p.T * (bt * bt / 4), 256 / r.sl);
    return r;
}

// prim: the update's own partials; alt (optional, with `gate`): the fp64 redo made by the shift guard
template <typename PT, int BT>
static void launch_reduce(const ReduceSrc& prim, const ReduceSrc* alt, int d, double* acc, double n_add, const int* gate,
                          int* clear_flag, bool overwrite, hipStream_t st) {
    const int col_blocks = (int)cdiv(d, 256);
    int blocks = prim.tile_blocks + col_blocks;
    if (alt && alt->tile_blocks + col_blocks > blocks) blocks = alt->tile_blocks + col_blocks;
    hipLaunchKernelGGL((moments_reduce<PT, BT>), dim3((unsigned)blocks), dim3(256), 0, st, prim, alt ? *alt : prim, d, acc,
                       n_add, alt ? gate : nullptr, clear_flag, overwrite ? 1 : 0);
}

// per-segment column sums: seg_sums[s][a] = sum over rows of segment s of E[r][a]   (fp64)
template <typename TIn>
__global__ __launch_bounds__(128) void segment_colsums(
    const TIn* __restrict__ E, int64_t ld, int d, const int64_t* __restrict__ offsets,
    double* __restrict__ seg_sums) {
    const int64_t seg = blockIdx.x;
    const int a = blockIdx.y * 128 + threadIdx.x;
    if (a >= d) return;
    const int64_t r0 = offsets[seg], r1 = offsets[seg + 1];
    double s0 = 0.0, s1 = 0.0;
    int64_t r = r0;
    for (; r + 1 < r1; r += 2) { s0 += to_f64<TIn>(E[r * ld + a]); s1 += to_f64<TIn>(E[(r + 1) * ld + a]); }
    if (r < r1) s0 += to_f64<TIn>(E[r * ld + a]);
    seg_sums[seg * d + a] = s0 + s1;
}

__global__ __launch_bounds__(256) void packed_axpy(double* __restrict__ dst, const double* __restrict__ src,
                                                   int64_t len) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < len) dst[g] += src[g];
}

// mu = sum/n ; cov = (M - sum sum^T / n) / (n - ddof)
__global__ __launch_bounds__(256) void moments_finalize_kernel(
    const double* __restrict__ acc_packed, int d, int ddof, double* __restrict__ mu,
    double* __restrict__ cov) {
    const double n = acc_packed[0];
    const double* sum = acc_packed + 1;
    const double* M = acc_packed + 1 + d;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < d && mu) mu[g] = sum[g] / n;
    if (g >= (int64_t)d * d) return;
    const int a = (int)(g / d), b = (int)(g - (int64_t)a * d);
    cov[g] = (M[g] - (sum[a] * sum[b]) / n) / (n - (double)ddof);   // commutative: cov == cov^T bit for bit
}

}  // namespace fad

// ==========================================================================================
// handle + host side
// ==========================================================================================
struct fad_moments {
    int d = 0, device = 0;
    double* acc = nullptr;                 // packed [1 + d + d*d]
    bool owns_acc = true;                  // false after fad_moments_bind: the caller's buffer
    fad::DevBuf partials, colpart, stage, seg_off, seg_out, scratch;
    fad::DevBuf partials64, colpart64;     // exact fp64 redo of a block flagged by the shift guard
    fad::DevBuf presum, presum_col;        // stage-1 output of the two-level reduce
    int* shift_flag = nullptr;             // device int[2], ping-pong between updates
    unsigned update_seq = 0;
    int guard = 1;                         // 0 disables the guard (FAD_MOMENTS_SHIFT_GUARD=0)
    bool fresh = false;                    // reset since the last update: the next reduce stores instead of adding
    // opt-in HIP-event timing: a ring of (before tile kernel, after tile kernel, after reduce) triplets,
    // recorded on the caller's stream and only read back by fad_moments_last_timing (no sync in update)
    static constexpr int kRing = 256;
    bool timing = false;
    hipEvent_t* ev = nullptr;              // [kRing][3]
    int ev_count = 0;                      // entries recorded since the last query
    int last_variant = -1;                 // 0: h16 MFMA tile kernel, 1: generic fp64 kernel
    int n_cu = 256;
};

namespace fad {

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return FAD_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&p, want) != hipSuccess) {
        p = nullptr;
        return set_error(FAD_ERR_ALLOC, "hipMalloc of %zu bytes failed", want);
    }
    cap = want;
    return FAD_OK;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }

static int64_t packed_len(int d) { return 1 + (int64_t)d + (int64_t)d * d; }


// max_rows bounds the run of rows one workgroup sums in fp32 (0 = unbounded).  The MFMA accumulate error is
// systematic (it behaves like truncation): measured -3.5e-7 relative at 32768 rows per run, ~1e-8 per 1000 rows,
// so runs are capped at 8192 rows and long inputs simply use more splits than resident slots.
static SplitPlan plan_splits(int64_t n, int d, int bt, int kb, int n_cu, int wg_per_cu, int64_t min_rows,
                             int64_t max_rows = 0) {
    SplitPlan p;
    p.nt = (int)cdiv(d, bt);
    p.T = p.nt * (p.nt + 1) / 2;
    int64_t want = ((int64_t)n_cu * wg_per_cu) / p.T;      // floor: never more workgroups than resident slots
    int64_t max_by_rows = cdiv(n, min_rows);
    int64_t s = want < max_by_rows ? want : max_by_rows;
    if (s < 1) s = 1;
    if (max_rows > 0 && cdiv(n, s) > max_rows) s = cdiv(n, max_rows);
    int64_t rps = cdiv(cdiv(n, s), kb) * kb;
    p.rows_per_split = rps;
    p.S = (int)cdiv(n, rps);
    return p;
}

template <typename TIn>
static void launch_generic(const void* rows, int64_t n, int64_t ld, int d, const SplitPlan& p,
                           double* partials, double* colpart, hipStream_t st, const int* gate = nullptr) {
    hipLaunchKernelGGL((moments_tile_f64<TIn>), dim3(p.S * p.T), dim3(256), 0, st,
                       reinterpret_cast<const TIn*>(rows), n, ld, d, p.nt, p.T, p.S, p.rows_per_split,
                       partials, colpart, gate);
}

// rows must be a DEVICE pointer here.
static int update_device(fad_moments* h, const void* rows, int64_t n, int64_t ld, int dtype, hipStream_t st) {
    const int d = h->d;
    const bool is16 = (dtype == FAD_F16 || dtype == FAD_BF16);
    const bool aligned = is16 && (d % 8 == 0) && (ld % 8 == 0) &&
                         ((reinterpret_cast<uintptr_t>(rows) & 15u) == 0);
    const char* force = getenv("FAD_MOMENTS_FORCE_GENERIC");
    const bool use_h16 = aligned && !(force && force[0] == '1');

    hipEvent_t* ev = nullptr;
    if (h->timing) {
        if (!h->ev) {
            h->ev = new (std::nothrow) hipEvent_t[fad_moments::kRing * 3];
            if (!h->ev) return set_error(FAD_ERR_ALLOC, "out of host memory");
            for (int i = 0; i < fad_moments::kRing * 3; ++i) FAD_HIP_TRY(hipEventCreate(&h->ev[i]));
        }
        ev = h->ev + 3 * (h->ev_count % fad_moments::kRing);
        h->ev_count++;
    }
    if (use_h16) {
        const char* var = getenv("FAD_MOMENTS_VARIANT");
        // default: v4; the one-tile-per-wave kernel (v8) wins on the HBM-bound single-tile shape (16.8M x 128: 0.80 vs
        // 0.91 ms) and loses at D = 512 (62 vs 51 us), see DESIGN.md section 4
        int variant = (var && var[0] >= '1' && var[0] <= '8') ? (var[0] - '0') : ((d <= H_BT && n >= (1 << 22)) ? 8 : 4);
        if (variant >= 5 && variant <= 7) variant = 4;        // 5..7 were experiments, gone
#ifndef FAD_MOM_NST
#define FAD_MOM_NST 4
#endif
#ifndef FAD_MOM_WGPCU
#define FAD_MOM_WGPCU 2        // workgroups per CU the split plan of v4 aims for (probe knob; 3 needs NST = 3)
#endif
        constexpr int NST = FAD_MOM_NST;         // LDS ring depth (stages); a build-time knob for scripts/probe_ablate.py
        // v8: one workgroup per CU, each wave sums at most 8192 rows in fp32 (4 waves per split)
        SplitPlan p = (variant == 8) ? plan_splits(n, d, H_BT, 64, h->n_cu, 1, 256, 4 * 8192)
                                     : plan_splits(n, d, H_BT, H_KB, h->n_cu, FAD_MOM_WGPCU, 256, 8192);
        const int layout = (variant == 8 || variant == 4) ? 1 : 0;      // fragment-major partial tiles
        FAD_TRY(h->partials.reserve((size_t)p.S * p.T * H_TS * sizeof(float)));
        FAD_TRY(h->colpart.reserve((size_t)p.S * p.nt * H_BT * sizeof(double)));
        float* part = static_cast<float*>(h->partials.p);
        double* colp = static_cast<double*>(h->colpart.p);
        if (ev) FAD_HIP_TRY(hipEventRecord(ev[0], st));
        const uint16_t* e16 = static_cast<const uint16_t*>(rows);
        int* flag_now = nullptr; int* flag_next = nullptr;
        if (h->guard && variant != 1) {
            flag_now = h->shift_flag + (h->update_seq & 1u);
            flag_next = h->shift_flag + ((h->update_seq + 1u) & 1u);
            h->update_seq++;
        }
        if (variant == 8) {
            static bool attr8 = false;
            if (!attr8) {
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile_h16_wave<FAD_F16>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile_h16_wave<FAD_BF16>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
                attr8 = true;
            }
            if (dtype == FAD_F16)
                hipLaunchKernelGGL((moments_tile_h16_wave<FAD_F16>), dim3(p.S * p.T), dim3(256), W_LDS, st, e16, n, ld, d, p.nt,
                                   p.T, p.S, p.rows_per_split, part, colp, flag_now);
            else
                hipLaunchKernelGGL((moments_tile_h16_wave<FAD_BF16>), dim3(p.S * p.T), dim3(256), W_LDS, st, e16, n, ld, d, p.nt,
                                   p.T, p.S, p.rows_per_split, part, colp, flag_now);
        } else if (variant == 3) {
            const size_t lds = (size_t)NST * 2 * H_KB * 16 * sizeof(uint4);
            static bool attr3 = false;
            if (!attr3) {
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile_h16_w2<FAD_F16, NST>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile_h16_w2<FAD_BF16, NST>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr3 = true;
            }
            if (dtype == FAD_F16)
                hipLaunchKernelGGL((moments_tile_h16_w2<FAD_F16, NST>), dim3(p.S * p.T), dim3(128), lds, st, e16, n, ld, d,
                                   p.nt, p.T, p.S, p.rows_per_split, part, colp, flag_now);
            else
                hipLaunchKernelGGL((moments_tile_h16_w2<FAD_BF16, NST>), dim3(p.S * p.T), dim3(128), lds, st, e16, n, ld, d,
                                   p.nt, p.T, p.S, p.rows_per_split, part, colp, flag_now);
        } else if (variant == 4) {
            const size_t lds = (size_t)NST * 2 * H_KB * 16 * sizeof(uint4);
            typedef void (*tr_kernel_t)(const uint16_t*, int64_t, int64_t, int, int, int, int, int64_t, float*, double*, int*);
            static const tr_kernel_t kern[2][2] = {
                {&moments_tile_h16_tr<FAD_F16, NST, false>, &moments_tile_h16_tr<FAD_F16, NST, true>},
                {&moments_tile_h16_tr<FAD_BF16, NST, false>, &moments_tile_h16_tr<FAD_BF16, NST, true>}};
            static bool attr4 = false;
            if (!attr4) {
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern[a][b]),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr4 = true;
            }
            hipLaunchKernelGGL(kern[dtype == FAD_F16 ? 0 : 1][p.T > 1 ? 1 : 0], dim3(p.S * p.T), dim3(256), lds, st, e16, n, ld,
                               d, p.nt, p.T, p.S, p.rows_per_split, part, colp, flag_now);
        } else if (variant == 2) {
            const size_t lds = (size_t)NST * 2 * H_KB * 16 * sizeof(uint4);
            static bool attr_set = false;
            if (!attr_set) {
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile_h16_glds<FAD_F16, NST>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile_h16_glds<FAD_BF16, NST>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = true;
            }
            if (dtype == FAD_F16)
                hipLaunchKernelGGL((moments_tile_h16_glds<FAD_F16, NST>), dim3(p.S * p.T), dim3(256), lds, st, e16, n, ld,
                                   d, p.nt, p.T, p.S, p.rows_per_split, part, colp, flag_now);
            else
                hipLaunchKernelGGL((moments_tile_h16_glds<FAD_BF16, NST>), dim3(p.S * p.T), dim3(256), lds, st, e16, n,
                                   ld, d, p.nt, p.T, p.S, p.rows_per_split, part, colp, flag_now);
        } else if (dtype == FAD_F16) {
            hipLaunchKernelGGL((moments_tile_h16<FAD_F16>), dim3(p.S * p.T), dim3(256), 0, st, e16, n, ld, d, p.nt, p.T,
                               p.S, p.rows_per_split, part, colp);
        } else {
            hipLaunchKernelGGL((moments_tile_h16<FAD_BF16>), dim3(p.S * p.T), dim3(256), 0, st, e16, n, ld, d, p.nt, p.T,
                               p.S, p.rows_per_split, part, colp);
        }
        if (ev) FAD_HIP_TRY(hipEventRecord(ev[1], st));
        ReduceSrc alt; const ReduceSrc* altp = nullptr;
        if (flag_now) {
            SplitPlan q = plan_splits(n, d, G_BT, G_KB, h->n_cu, 2, 128);
            FAD_TRY(h->partials64.reserve((size_t)q.S * q.T * G_TS * sizeof(double)));
            FAD_TRY(h->colpart64.reserve((size_t)q.S * q.nt * G_BT * sizeof(double)));
            double* part64 = static_cast<double*>(h->partials64.p);
            double* colp64 = static_cast<double*>(h->colpart64.p);
            if (dtype == FAD_F16) launch_generic<raw_f16>(rows, n, ld, d, q, part64, colp64, st, flag_now);
            else launch_generic<raw_bf16>(rows, n, ld, d, q, part64, colp64, st, flag_now);
            alt = reduce_src(part64, colp64, q, G_BT, 0);
            altp = &alt;
        }
        const bool overwrite = h->fresh;
        if (p.S > 128) {                   // two-level: S -> ceil(S/32) fp64 partials -> accumulator
            SplitPlan p2 = p;
            p2.S = (int)cdiv(p.S, PRESUM_CHUNK);
            FAD_TRY(h->presum.reserve((size_t)p2.S * p.T * (H_BT * H_BT + 32) * sizeof(double)));
            FAD_TRY(h->presum_col.reserve((size_t)p2.S * p.nt * H_BT * sizeof(double)));
            double* ps = static_cast<double*>(h->presum.p);
            double* pc = static_cast<double*>(h->presum_col.p);
            const int gb = (int)cdiv((int64_t)p.T * (H_BT * H_BT / 4), 256);
            hipLaunchKernelGGL((moments_presum<H_BT>), dim3((unsigned)(gb + cdiv(p.nt * H_BT, 256)), (unsigned)p2.S), dim3(256),
                               0, st, part, colp, p.S, p.T, p.nt, gb, ps, pc, (const int*)flag_now);
            launch_reduce<double, H_BT>(reduce_src(ps, pc, p2, H_BT, layout), altp, d, h->acc, (double)n, flag_now, flag_next,
                                        overwrite, st);
        } else {
            launch_reduce<float, H_BT>(reduce_src(part, colp, p, H_BT, layout), altp, d, h->acc, (double)n, flag_now, flag_next,
                                       overwrite, st);
        }
        h->fresh = false;
        if (ev) FAD_HIP_TRY(hipEventRecord(ev[2], st));
        h->last_variant = (variant == 1) ? 2 : 0;
    } else {
        SplitPlan p = plan_splits(n, d, G_BT, G_KB, h->n_cu, 2, 128);
        FAD_TRY(h->partials.reserve((size_t)p.S * p.T * G_TS * sizeof(double)));
        FAD_TRY(h->colpart.reserve((size_t)p.S * p.nt * G_BT * sizeof(double)));
        double* part = static_cast<double*>(h->partials.p);
        double* colp = static_cast<double*>(h->colpart.p);
        if (ev) FAD_HIP_TRY(hipEventRecord(ev[0], st));
        switch (dtype) {
            case FAD_F16: launch_generic<raw_f16>(rows, n, ld, d, p, part, colp, st); break;
            case FAD_BF16: launch_generic<raw_bf16>(rows, n, ld, d, p, part, colp, st); break;
            case FAD_F32: launch_generic<float>(rows, n, ld, d, p, part, colp, st); break;
            case FAD_F64: launch_generic<double>(rows, n, ld, d, p, part, colp, st); break;
            default: return set_error(FAD_ERR_INVALID, "unknown dtype %d", dtype);
        }
        if (ev) FAD_HIP_TRY(hipEventRecord(ev[1], st));
        launch_reduce<double, G_BT>(reduce_src(part, colp, p, G_BT, 0), nullptr, d, h->acc, (double)n, nullptr, nullptr, h->fresh, st);
        h->fresh = false;
        if (ev) FAD_HIP_TRY(hipEventRecord(ev[2], st));
        h->last_variant = 1;
    }
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

// Host rows -> staged through h->stage in bounded chunks (PCIe), then update_device.
static int update_any(fad_moments* h, const void* rows, int64_t n, int64_t ld, int dtype, int on_device,
                      hipStream_t st) {
    if (on_device) return update_device(h, rows, n, ld, dtype, st);
    const size_t es = dtype_size(dtype);
    const int64_t row_bytes = (int64_t)h->d * es;
    int64_t chunk_rows = ((int64_t)1 << 30) / (row_bytes > 0 ? row_bytes : 1);
    if (chunk_rows < 1024) chunk_rows = 1024;
    chunk_rows = (chunk_rows / 32) * 32;
    for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        const int64_t m = (n - r0 < chunk_rows) ? n - r0 : chunk_rows;
        FAD_TRY(h->stage.reserve((size_t)m * row_bytes + 16));
        const char* src = static_cast<const char*>(rows) + r0 * ld * es;
        FAD_HIP_TRY(hipMemcpy2DAsync(h->stage.p, row_bytes, src, ld * es, row_bytes, m,
                                     hipMemcpyHostToDevice, st));
        FAD_TRY(update_device(h, h->stage.p, m, h->d, dtype, st));
        // the staging buffer is reused by the next chunk: wait for the kernels that read it
        if (r0 + m < n) FAD_HIP_TRY(hipStreamSynchronize(st));
    }
    return FAD_OK;
}

}  // namespace fad

using namespace fad;

extern "C" {

int fad_moments_create(int d, int device, fad_moments_t** out) {
    if (!out) return set_error(FAD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || d > 16384) return set_error(FAD_ERR_INVALID, "d=%d out of range [1, 16384]", d);
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return set_error(FAD_ERR_HIP, "cannot select device %d", device);
    fad_moments* h = new (std::nothrow) fad_moments();
    if (!h) return set_error(FAD_ERR_ALLOC, "out of host memory");
    h->d = d; h->device = device; h->n_cu = num_cus(device);
    const size_t bytes = (size_t)packed_len(d) * sizeof(double);
    if (hipMalloc(reinterpret_cast<void**>(&h->acc), bytes) != hipSuccess) {
        delete h;
        return set_error(FAD_ERR_ALLOC, "hipMalloc of %zu bytes failed", bytes);
    }
    if (hipMemset(h->acc, 0, bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&h->shift_flag), 2 * sizeof(int)) != hipSuccess ||
        hipMemset(h->shift_flag, 0, 2 * sizeof(int)) != hipSuccess) {
        (void)hipFree(h->acc); delete h;
        return set_error(FAD_ERR_HIP, "hipMemset / hipMalloc failed");
    }
    const char* gs = getenv("FAD_MOMENTS_SHIFT_GUARD");
    h->guard = !(gs && gs[0] == '0');
    *out = h;
    return FAD_OK;
}

int fad_moments_destroy(fad_moments_t* h) {
    if (!h) return FAD_OK;
    DeviceGuard g(h->device);
    if (h->acc && h->owns_acc) (void)hipFree(h->acc);
    if (h->shift_flag) (void)hipFree(h->shift_flag);
    h->partials64.release(); h->colpart64.release(); h->presum.release(); h->presum_col.release();
    h->partials.release(); h->colpart.release(); h->stage.release();
    h->seg_off.release(); h->seg_out.release(); h->scratch.release();
    if (h->ev) { for (int i = 0; i < fad_moments::kRing * 3; ++i) (void)hipEventDestroy(h->ev[i]); delete[] h->ev; }
    delete h;
    return FAD_OK;
}

// The accumulator's memset is deferred: an update right after a reset overwrites it (saves a launch per set);
// every other reader settles the pending zeroing first.
static int settle(const fad_moments* hc, hipStream_t st) {
    fad_moments* h = const_cast<fad_moments*>(hc);
    if (!h->fresh) return FAD_OK;
    FAD_HIP_TRY(hipMemsetAsync(h->acc, 0, (size_t)packed_len(h->d) * sizeof(double), st));
    h->fresh = false;
    return FAD_OK;
}

int fad_moments_reset(fad_moments_t* h, void* stream) {
    if (!h) return set_error(FAD_ERR_INVALID, "handle is NULL");
    (void)stream;
    h->fresh = true;
    return FAD_OK;
}

int fad_moments_bind(fad_moments_t* h, double* device_packed) {
    if (!h || !device_packed) return set_error(FAD_ERR_INVALID, "NULL argument");
    if (reinterpret_cast<uintptr_t>(device_packed) & 15u) return set_error(FAD_ERR_INVALID, "buffer must be 16-byte aligned");
    DeviceGuard g(h->device);
    if (h->acc && h->owns_acc) FAD_HIP_TRY(hipFree(h->acc));
    h->acc = device_packed;
    h->owns_acc = false;
    h->fresh = true;                               // bind = adopt the buffer and reset
    return FAD_OK;
}

int fad_moments_dim(const fad_moments_t* h) { return h ? h->d : set_error(FAD_ERR_INVALID, "handle is NULL"); }
int64_t fad_moments_packed_len(const fad_moments_t* h) {
    return h ? packed_len(h->d) : (int64_t)set_error(FAD_ERR_INVALID, "handle is NULL");
}

int fad_moments_update(fad_moments_t* h, const void* rows, int64_t n, int64_t ld, int dtype,
                       int on_device, void* stream) {
    if (!h) return set_error(FAD_ERR_INVALID, "handle is NULL");
    if (n < 0 || ld < h->d) return set_error(FAD_ERR_SHAPE, "n=%lld ld=%lld d=%d", (long long)n, (long long)ld, h->d);
    if (dtype_size(dtype) == 0) return set_error(FAD_ERR_INVALID, "unknown dtype %d", dtype);
    if (n == 0) return FAD_OK;
    if (!rows) return set_error(FAD_ERR_INVALID, "rows is NULL");
    DeviceGuard g(h->device);
    return update_any(h, rows, n, ld, dtype, on_device, static_cast<hipStream_t>(stream));
}

int fad_moments_update_segmented(fad_moments_t* h, const void* rows, int64_t n, int64_t ld, int dtype,
                                 const int64_t* offsets, int64_t n_segments, double* seg_sums,
                                 int on_device, void* stream) {
    if (!h) return set_error(FAD_ERR_INVALID, "handle is NULL");
    if (n < 0 || ld < h->d) return set_error(FAD_ERR_SHAPE, "n=%lld ld=%lld d=%d", (long long)n, (long long)ld, h->d);
    if (dtype_size(dtype) == 0) return set_error(FAD_ERR_INVALID, "unknown dtype %d", dtype);
    if (n_segments < 0 || (n_segments > 0 && !offsets)) return set_error(FAD_ERR_INVALID, "bad segment list");
    for (int64_t s = 0; s < n_segments; ++s)
        if (offsets[s] > offsets[s + 1] || offsets[s] < 0 || offsets[s + 1] > n)
            return set_error(FAD_ERR_INVALID, "offsets must be non-decreasing within [0, n]");
    if (n == 0) return FAD_OK;
    if (!rows) return set_error(FAD_ERR_INVALID, "rows is NULL");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t es = dtype_size(dtype);
    const void* drows = rows;
    int64_t dld = ld;
    if (!on_device) {      // one staged copy serves both kernels (bounded by caller: host blocks are per-batch)
        const int64_t row_bytes = (int64_t)h->d * es;
        FAD_TRY(h->stage.reserve((size_t)n * row_bytes + 16));
        FAD_HIP_TRY(hipMemcpy2DAsync(h->stage.p, row_bytes, rows, ld * es, row_bytes, n, hipMemcpyHostToDevice, st));
        drows = h->stage.p; dld = h->d;
    }
    FAD_TRY(update_device(h, drows, n, dld, dtype, st));
    if (seg_sums && n_segments > 0) {
        FAD_TRY(h->seg_off.reserve((size_t)(n_segments + 1) * sizeof(int64_t)));
        FAD_HIP_TRY(hipMemcpyAsync(h->seg_off.p, offsets, (size_t)(n_segments + 1) * sizeof(int64_t),
                                   hipMemcpyHostToDevice, st));
        double* dout = seg_sums;
        if (!on_device) {
            FAD_TRY(h->seg_out.reserve((size_t)n_segments * h->d * sizeof(double)));
            dout = static_cast<double*>(h->seg_out.p);
        }
        const dim3 grid((unsigned)n_segments, (unsigned)cdiv(h->d, 128));
        const int64_t* doff = static_cast<const int64_t*>(h->seg_off.p);
        switch (dtype) {
            case FAD_F16: hipLaunchKernelGGL((segment_colsums<raw_f16>), grid, dim3(128), 0, st, static_cast<const raw_f16*>(drows), dld, h->d, doff, dout); break;
            case FAD_BF16: hipLaunchKernelGGL((segment_colsums<raw_bf16>), grid, dim3(128), 0, st, static_cast<const raw_bf16*>(drows), dld, h->d, doff, dout); break;
            case FAD_F32: hipLaunchKernelGGL((segment_colsums<float>), grid, dim3(128), 0, st, static_cast<const float*>(drows), dld, h->d, doff, dout); break;
            default: hipLaunchKernelGGL((segment_colsums<double>), grid, dim3(128), 0, st, static_cast<const double*>(drows), dld, h->d, doff, dout); break;
        }
        FAD_HIP_TRY(hipGetLastError());
        if (!on_device) {
            FAD_HIP_TRY(hipMemcpyAsync(seg_sums, dout, (size_t)n_segments * h->d * sizeof(double),
                                       hipMemcpyDeviceToHost, st));
        }
    }
    if (!on_device) FAD_HIP_TRY(hipStreamSynchronize(st));
    return FAD_OK;
}

int fad_moments_merge(fad_moments_t* dst, const fad_moments_t* src, void* stream) {
    if (!dst || !src) return set_error(FAD_ERR_INVALID, "handle is NULL");
    if (dst->d != src->d) return set_error(FAD_ERR_SHAPE, "d mismatch %d vs %d", dst->d, src->d);
    if (dst->device != src->device) return set_error(FAD_ERR_INVALID, "handles live on different devices; export/import instead");
    DeviceGuard g(dst->device);
    const int64_t len = packed_len(dst->d);
    FAD_TRY(settle(dst, static_cast<hipStream_t>(stream)));
    FAD_TRY(settle(src, static_cast<hipStream_t>(stream)));
    hipLaunchKernelGGL(packed_axpy, dim3((unsigned)cdiv(len, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dst->acc, src->acc, len);
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

int fad_moments_export(const fad_moments_t* h, double* packed, int on_device, void* stream) {
    if (!h || !packed) return set_error(FAD_ERR_INVALID, "NULL argument");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t bytes = (size_t)packed_len(h->d) * sizeof(double);
    FAD_TRY(settle(h, st));
    FAD_HIP_TRY(hipMemcpyAsync(packed, h->acc, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (!on_device) FAD_HIP_TRY(hipStreamSynchronize(st));
    return FAD_OK;
}

int fad_moments_import(fad_moments_t* h, const double* packed, int on_device, void* stream) {
    if (!h || !packed) return set_error(FAD_ERR_INVALID, "NULL argument");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t bytes = (size_t)packed_len(h->d) * sizeof(double);
    h->fresh = false;                              // the whole accumulator is overwritten
    FAD_HIP_TRY(hipMemcpyAsync(h->acc, packed, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if (!on_device) FAD_HIP_TRY(hipStreamSynchronize(st));
    return FAD_OK;
}

int fad_moments_allreduce(fad_moments_t* h, void* rccl_comm, void* stream) {
    if (!h || !rccl_comm) return set_error(FAD_ERR_INVALID, "NULL argument");
    // ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_fn fn = [] {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");          // the RCCL the host process uses (e.g. torch's)
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            if (sym) break;
            if (void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) sym = dlsym(lib, "ncclAllReduce");
        }
        return reinterpret_cast<allreduce_fn>(sym);
    }();
    if (!fn) return set_error(FAD_ERR_INVALID, "no RCCL in this process (ncclAllReduce not found)");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    FAD_TRY(settle(h, st));
    constexpr int kNcclFloat64 = 8, kNcclSum = 0;                   // rccl.h: ncclDataType_t / ncclRedOp_t
    const int rc = fn(h->acc, h->acc, (size_t)packed_len(h->d), kNcclFloat64, kNcclSum, rccl_comm, st);
    if (rc != 0) return set_error(FAD_ERR_HIP, "ncclAllReduce failed with ncclResult_t %d", rc);
    return FAD_OK;
}

int fad_moments_count(const fad_moments_t* h, int64_t* n, void* stream) {
    if (!h || !n) return set_error(FAD_ERR_INVALID, "NULL argument");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double v = 0.0;
    FAD_TRY(settle(h, st));
    FAD_HIP_TRY(hipMemcpyAsync(&v, h->acc, sizeof(double), hipMemcpyDeviceToHost, st));
    FAD_HIP_TRY(hipStreamSynchronize(st));
    *n = (int64_t)(v + 0.5);
    return FAD_OK;
}

int fad_moments_finalize(const fad_moments_t* h, int ddof, double* mu, double* cov, int64_t* n_out,
                         int on_device, void* stream) {
    if (!h || !cov) return set_error(FAD_ERR_INVALID, "NULL argument");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t n = 0;
    FAD_TRY(fad_moments_count(h, &n, stream));
    if (n_out) *n_out = n;
    if (n < 2) return set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames, you have %lld", (long long)n);
    const int d = h->d;
    fad_moments* hm = const_cast<fad_moments*>(h);
    double* dmu = mu; double* dcov = cov;
    if (!on_device) {
        FAD_TRY(hm->scratch.reserve(((size_t)d * d + d) * sizeof(double)));
        dcov = static_cast<double*>(hm->scratch.p);
        dmu = dcov + (size_t)d * d;
    }
    hipLaunchKernelGGL(moments_finalize_kernel, dim3((unsigned)cdiv((int64_t)d * d, 256)), dim3(256), 0, st,
                       h->acc, d, ddof, dmu, dcov);
    FAD_HIP_TRY(hipGetLastError());
    if (!on_device) {
        FAD_HIP_TRY(hipMemcpyAsync(cov, dcov, (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st));
        if (mu) FAD_HIP_TRY(hipMemcpyAsync(mu, dmu, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, st));
        FAD_HIP_TRY(hipStreamSynchronize(st));
    }
    return FAD_OK;
}

int fad_moments_set_timing(fad_moments_t* h, int enabled) {
    if (!h) return set_error(FAD_ERR_INVALID, "handle is NULL");
    h->timing = enabled != 0;
    h->ev_count = 0;
    return FAD_OK;
}

// Average duration of the tile kernel and of the reduce kernels over every update recorded since
// timing was enabled / last queried (at most the last 256).  Synchronises on the newest event.
int fad_moments_last_timing(fad_moments_t* h, float* ms_main, float* ms_reduce, int* variant) {
    if (!h) return set_error(FAD_ERR_INVALID, "handle is NULL");
    if (!h->timing || !h->ev || h->ev_count == 0)
        return set_error(FAD_ERR_INVALID, "timing was not enabled before the update");
    DeviceGuard g(h->device);
    const int total = h->ev_count;
    const int m = total < fad_moments::kRing ? total : fad_moments::kRing;
    FAD_HIP_TRY(hipEventSynchronize(h->ev[3 * ((total - 1) % fad_moments::kRing) + 2]));
    double a = 0.0, b = 0.0;
    for (int i = total - m; i < total; ++i) {
        hipEvent_t* e = h->ev + 3 * (i % fad_moments::kRing);
        float x = 0.f, y = 0.f;
        FAD_HIP_TRY(hipEventElapsedTime(&x, e[0], e[1]));
        FAD_HIP_TRY(hipEventElapsedTime(&y, e[1], e[2]));
        a += x; b += y;
    }
    if (ms_main) *ms_main = (float)(a / m);
    if (ms_reduce) *ms_reduce = (float)(b / m);
    if (variant) *variant = h->last_variant;
    h->ev_count = 0;
    return FAD_OK;
}

}  // extern "C"

// accessors for the other translation units (frechet.hip)
namespace fad {
const double* moments_packed(const fad_moments* h) { return h->acc; }
int moments_settle(const fad_moments* h, hipStream_t st) { return settle(h, st); }
int moments_device(const fad_moments* h) { return h->device; }
int moments_dim(const fad_moments* h) { return h->d; }
}
