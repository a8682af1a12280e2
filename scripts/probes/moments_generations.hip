// Earlier generations of the fp16/bf16 E^T E tile kernel, kept for reference and A/B probing only.
// NOT part of libfad_hip.so (the product ships moments_tile_h16_tr = "v4" and moments_tile_h16_wave = "v8",
// fadtk_amd/csrc/moments.hip).  Syntax/ISA check:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -c scripts/probes/moments_generations.hip -o /dev/null
//
//   v1  moments_tile_h16        register-staged, one 16 KiB stage in flight            189 us at config 3
//   v2  moments_tile_h16_glds   LDS-DMA ring + ds_read_b32 / v_perm_b32 fragments        58 us
//   v3  moments_tile_h16_w2     two waves x 128x64 wave tiles on ds_read_b64             73 us
// (round-1 measurements).  The ablation switches (FAD_MOM_ABLATE ...) that used to be
// threaded through the product kernels lived here too; see git history of fadtk_amd/csrc/moments.hip at ae632c4.
#include "../../fadtk_amd/csrc/fad_common.h"
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

}  // namespace fad


// ==========================================================================================================================
// v8 (rounds 1-2, shipped until the four-wave kernel's diagonal-tile codegen was fixed): one 128 x 128 tile per WAVE.
// It won the HBM-bound single-tile stream (16.8M x 128: 0.80 vs 0.91 ms) until the role split of the four-wave kernel
// (moments_kernels.h, stage_diag) removed 80 v_accvgpr_mov per stage there; after it: four-wave 760 us (5.65 TB/s) vs wave
// 784 us (5.48 TB/s) at 16.8M x 128, 441 vs 465 us at 9.2M, 48.6 vs 55.7 us at 1M (scripts/probe_variants.py, round 2),
// so the library no longer builds it.  Kept here verbatim for reference; it relies on the declarations of
// fadtk_amd/csrc/moments_kernels.h (TileLaunch, locate, frag helpers) and is not compiled by this file.
// ==========================================================================================================================
#if 0
// ------------------------------------------------------------------------------------------
// moments_tile_h16_wave: every wave owns a WHOLE 128 x 128 tile in 256 accumulator registers and streams its own
// rows; the four waves of a workgroup take the 16-row k-steps round robin and their tiles are summed once, at
// the end, through LDS.  Why (measured in round 1):
//   * LDS transpose reads run at ~120 B/clk per CU.  With 64 x 64 wave tiles an MFMA needs 1 KiB of LDS reads,
//     which makes the LDS port as busy as the matrix pipe, and the eight waves of a CU queue on it in lockstep.
//     A 128 x 128 wave tile needs 0.5 KiB per MFMA.
//   * One workgroup per CU halves the partial tiles (written, then read by the reduce kernel).
//   * No workgroup barrier and no LDS sharing in the main loop: a wave waits only on its own LDS-DMA counter.
// Per wave: ring of NSL slots of one k-step (16 rows x 128 columns of the A side, + the B side off the diagonal),
// filled by global_load_lds with the same source-side XOR swizzle as above; per k-step 16 (8) transpose reads feed
// 16 (10 on a diagonal tile: upper blocks only) MFMAs; the reads of step i+1 are issued right behind the first
// MFMA of step i (the compiler only emits lgkmcnt(0) around ds_read_b64_tr_b16, so that is where a full wait is
// harmless).  It loses to the kernel above at D = 512 (62 vs 51 us: with one wave per SIMD nothing fills the gaps the
// loads leave) and wins on the HBM-bound single-tile stream (16.8M x 128: 0.80 vs 0.91 ms).
// ------------------------------------------------------------------------------------------
constexpr int W_RING = 32768;                                  // LDS ring bytes per wave
constexpr int W_LDS = 4 * W_RING + 4 * H_BT * 8 + H_BT * 8;    // + per-wave column sums + their total

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
        char* slot = ring + (i % NSL) * SLOTB;
        const int h = g & 3;
        const int64_t r = k_begin + (int64_t)(wave + 4 * i) * 16 + srow + 4 * h;
        const bool ok = r < k_end;
        if (g < 4) {
            const uint16_t* srcA = (ok && col_ok_a) ? ga + r * ld : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)srcA, (lptr_t)(slot + h * 1024), 16, 0, 0);
        } else {
            const uint16_t* srcB = (ok && col_ok_b) ? gb + r * ld : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)srcB, (lptr_t)(slot + 4096 + h * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int i) {
#pragma unroll
        for (int g = 0; g < LPS; ++g) issue_part(i, g);
    };

    // transpose-read addressing (see above): byte offset of the lane's first read of 32-column fragment f
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

    auto mma_first = [&](const uint4 (&F)[NFR]) { acc[0][0] = mfma_h16<KIND>(F[0], F[DIAG ? 0 : 4], acc[0][0]); };
    auto mma_rest = [&](const uint4 (&F)[NFR]) {
#pragma unroll
        for (int fa = 0; fa < 4; ++fa)
#pragma unroll
            for (int fb = (DIAG ? fa : 0); fb < 4; ++fb) {
                if (fa == 0 && fb == 0) continue;
                acc[fa][fb] = mfma_h16<KIND>(F[fa], F[DIAG ? fb : 4 + fb], acc[fa][fb]);
            }
        if (DIAG) {
#pragma unroll
            for (int f = 0; f < 4; ++f) csum[f] += (double)sum8<KIND>(F[f]);
        }
    };

    const int n0 = nw < NSL ? nw : NSL;
    for (int s = 0; s < n0; ++s) issue(s);
    uint4 C[NFR], N[NFR];                          // fragments of the current / the next k-step
    if (nw > 0) {
        wait_vmcnt_upto<LPS>(n0 - 1);
        load_frags(0, C);
    }
    for (int i = 0; i < nw; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        mma_first(C);                              // (lgkmcnt(0) before it: every read of step i has landed)
        __builtin_amdgcn_sched_barrier(0);
        if (i + NSL < nw) issue(i + NSL);          // ... so its slot can be refilled
        if (i + 1 < nw) {
            const int newest = i + NSL;
            const int youngest = (nw - 1 < newest) ? nw - 1 : newest;
            wait_vmcnt_upto<LPS>(youngest - (i + 1));      // step i+1 is in LDS
            load_frags(i + 1, N);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_rest(C);
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
__global__ __launch_bounds__(256) void moments_tile_h16_wave(TileLaunch L) {
    extern __shared__ __attribute__((aligned(16))) char smem_wave[];   // the ONLY LDS object: W_LDS bytes
    const int w = xcd_contiguous(blockIdx.x, L.total);
    int split, tile, run_lo, run_hi; int64_t k_begin, k_end;
    const TileSet& s = locate(L, w, split, tile, k_begin, k_end, run_lo, run_hi);      // (runs are a moments_tile_h16_tr feature)
    int ta, tb; tile_coords(tile, L.nt, ta, tb);
    const uint16_t* E = static_cast<const uint16_t*>(s.E);
    float* partials = static_cast<float*>(s.partials);
    if (ta == tb)
        tile_h16_wave_body<KIND, true>(E, k_begin, k_end, s.ld, L.d, L.nt, L.T, split, tile, ta * H_BT, tb * H_BT, partials,
                                       s.colpart, smem_wave, s.flag);
    else
        tile_h16_wave_body<KIND, false>(E, k_begin, k_end, s.ld, L.d, L.nt, L.T, split, tile, ta * H_BT, tb * H_BT, partials,
                                        s.colpart, smem_wave, nullptr);
}

#endif
