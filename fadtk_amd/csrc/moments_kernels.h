// Device side of the moments path (gfx950): the tile kernels, the reduce / presum kernels, the per-segment sums and the
// small finishing kernels, with the launch descriptors they take.  Included by moments.hip only (which holds the overview,
// the handle, the split planner and the C ABI); kept apart so that either file can be read on its own.
#pragma once
#include "fad_common.h"
#include <type_traits>

namespace fad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kXcd = 8;
constexpr int kMaxSets = 16;      // frame matrices per launch
constexpr int kMaxSets256 = 32;   // ... of the 256-column-slab kernel (moments_tile256.h): the statistics of sixteen scores in one launch

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

// One frame matrix of a launch and where its partial sums go.
struct SegRun;
struct TileSet {
    const void* E;             // rows (device)
    int64_t n, ld;             // frames, row pitch in elements
    int64_t rows_per_split;    // split s sums rows [s * rows_per_split, ...)
    int S;                     // row-splits
    int item0;                 // first work item of this set; item = item0 + split * T + tile
    void* partials;            // [S][T][tile stride]
    double* colpart;           // [S][nt * BT] -- [runs][nt * BT] when `runs` is set
    int* flag;                 // shift guard: raised by the fp16 kernels, gate of the second pass / the fp64 redo (or nullptr)
    uint16_t* cvec;            // shift guard, fp16 rows: [S][nt * BT] per-split column shifts (float16 bits), see tile_h16_tr_body
    // Segment-aligned splits (fad_moments_update_segmented on long files): split s sums the runs
    // [split_first_run[s], split_first_run[s+1]), each run = rows of ONE file, and writes every run's column sums to
    // its own colpart row -- the per-file sums fall out of the one pass over E (moments_tile_h16_tr only).
    const SegRun* runs;
    const int* split_first_run;
    // WALK launches (fad_moments_update_segmented_ref on files of one run each): numpy's float32 running column sums of every FILE
    // (utils.py:16: np.mean of a float16 file adds its rows one after the other in float32) leave the same pass -- [n_segments][d]
    float* seg_runsum;
};
struct SegRun { int64_t r0; int32_t rows; int32_t seg; };
struct TileLaunch {
    TileSet set[kMaxSets];
    int nsets, d, nt, T, total;
};

// work item -> (set, split, tile, row range).  `w` is wave-uniform, so the table is read with scalar loads.
__device__ __forceinline__ const TileSet& locate(const TileLaunch& L, int w, int& split, int& tile, int64_t& k_begin,
                                                 int64_t& k_end, int& run_lo, int& run_hi) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < kMaxSets; ++i)
        if (i < L.nsets && w >= L.set[i].item0) si = i;
    const TileSet& s = L.set[si];
    const int local = w - s.item0;
    split = local / L.T; tile = local - split * L.T;
    k_begin = (int64_t)split * s.rows_per_split;
    k_end = (k_begin + s.rows_per_split < s.n) ? k_begin + s.rows_per_split : s.n;
    run_lo = 0; run_hi = 0;
    if (s.runs) { run_lo = s.split_first_run[split]; run_hi = s.split_first_run[split + 1]; }
    return s;
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

template <int KIND> __device__ __forceinline__ f32x16 mfma_h16(const uint4& a, const uint4& b, const f32x16& c) {
    if constexpr (KIND == FAD_F16) {
        f16x8 va, vb; __builtin_memcpy(&va, &a, 16); __builtin_memcpy(&vb, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c, 0, 0, 0);
    } else {
        bf16x8 va, vb; __builtin_memcpy(&va, &a, 16); __builtin_memcpy(&vb, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, c, 0, 0, 0);
    }
}

constexpr int H_BT = 128;     // tile edge
constexpr int H_TS = H_BT * H_BT + 64;   // partial-tile stride (floats): +256 B so that the same element of
                                         // consecutive tiles/splits does not alias onto one memory channel
constexpr int H_KB = 32;      // rows per stage
constexpr int H_NST = 4;      // LDS ring depth (stages): 4 x 16 KiB per workgroup, two workgroups per CU

// Out-of-range rows / columns of an LDS-DMA load are redirected per lane to a 16-byte block of zeros.
__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

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

// ------------------------------------------------------------------------------------------
// moments_tile_h16_tr.  256 threads = 4 waves as 2x2; workgroup tile 128 x 128 of E^T E, wave tile 64 x 64 = 2x2
// MFMA 32x32 tiles; 32 rows of E per LDS stage.
//   * slabs of E go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip)
//     through a ring of NST stages; waits are counted (s_waitcnt vmcnt(N), never 0 in steady state) and the barrier
//     is a raw s_barrier so that younger stages stay in flight across it;
//   * operand fragments by ds_read_b64_tr_b16 (LDS transpose read): in a 16-lane group lane t supplies the address
//     of 4 consecutive columns of row t>>2 and receives 4 consecutive ROWS of column t -- exactly the k-contiguous
//     fragment an MFMA wants from a row-major slab (semantics verified with scripts/probes/tr_probe.hip).
// FAST = LDS-DMA loads issued as inline asm with a wave-uniform SGPR base (see issue_fast); used when the problem has
// more than one tile (MFMA-bound shapes).  Single-tile problems (D <= 128, HBM-bound) measured slower with either
// asm form (63 / 59 vs 53 us for 1M x 128) and keep the builtin loads throughout.
// ------------------------------------------------------------------------------------------
// SHIFT: the second pass of the shift guard (float16 rows).  The first pass found a column whose mean^2 exceeds 64 x its
// variance inside some split -- float32 partial sums of x_i x_j cannot resolve the covariance there -- and left, per split and
// for EVERY column, c_j = float16(mean_j) (0 where mean^2 <= var: nothing to gain) in `cvec`.  This pass streams the same rows
// again and feeds x - c to the MFMAs as an error-free pair of float16 values: x - c = x' + e with x' = fl(x - c) and e the
// rounding error of that subtraction (TwoSum; e = 0 whenever x and c are within a factor of two of each other -- always, on
// the columns that tripped the guard).  Products are exact in float32 as before, three MFMAs per block instead of one
// (x'x' + x'e + ex'; ee is below 2^-22 of it), and what is summed is CENTRED: float32-sum accuracy relative to the variance
// of every column, tripped or not.  moments_reduce restores the raw moments in float64: sum x_i x_j = S'_ij + c_i s'_j +
// s'_i c_j + n c_i c_j with s' the column sums of x - c.  Two passes of this kernel + a few loads per split in the reduce
// instead of the float64 kernel over the whole block: ~10x faster when the guard fires (scripts/probe_guard.py).
template <int KIND, int NST, bool DIAG, bool FAST, bool SHIFT = false, bool WALK = false>
__device__ __forceinline__ void tile_h16_tr_body(
    const uint16_t* __restrict__ E, int64_t k_begin0, int64_t k_end0, int64_t ld, int d, int nt, int T,
    int split, int tile, int ca, int cb, float* __restrict__ partials, double* __restrict__ colpart,
    uint4* smem, int* __restrict__ shift_flag, const SegRun* __restrict__ runs, int run_lo, int run_hi,
    uint16_t* __restrict__ cvec = nullptr, float* __restrict__ seg_runsum = nullptr) {
    static_assert(!WALK || (DIAG && !SHIFT), "the per-file walk rides on the diagonal tiles of the first pass");
    static_assert(!SHIFT || KIND == FAD_F16, "the shifted pass is written for float16 rows");
    constexpr int LPS = DIAG ? 2 : 4;              // glds instructions per wave per stage
    // uint4 per stage: A slab + B slab; a launch whose only tile is the diagonal one (FAST = false: D <= 128, the
    // HBM-bound shape) has no B slab and spends the same 64 KiB on twice as many stages in flight
    constexpr int STAGE = FAST ? 2 * H_KB * 16 : H_KB * 16;
    static_assert(FAST || DIAG, "single-tile launches only have the diagonal tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform (scalar branches around MFMAs)
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kg = lane >> 5;
    // rows of the current run (the whole split unless `runs` is given); the lambdas below see these by reference
    int64_t k_begin = k_begin0, k_end = k_end0;
    int nkb = 0;

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
        uint4* st = smem + (kb % NST) * STAGE;
        const int h = DIAG ? g : (g >> 1);
        const int64_t r = k_begin + (int64_t)kb * H_KB + sr + 16 * h;
        const bool ok = r < k_end;
        // LDS destination = wave-uniform base + lane*16: rows 16h + 4*wave .. +3, 16 chunks each
        const bool side_b = !DIAG && (g & 1);
        const uint16_t* src = side_b ? ((ok && col_ok_b) ? gb + r * ld : zsrc) : ((ok && col_ok_a) ? ga + r * ld : zsrc);
        uint4* dstp = st + 256 * h + 64 * wave + (side_b ? H_KB * 16 : 0);
        if (FAST) {
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
            // (the non-temporal hint `nt` on this load, measured in round 3 on the bench's rotated inputs: 889 instead of 1270 GB/s)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    int nfast = 0;                                 // stages [0, nfast) of the current run may be loaded the fast way

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    double csum[2] = {0.0, 0.0};                   // column sums of the current run
    double ctot[2] = {0.0, 0.0};                   // ... of the whole split (shift guard)
    const bool do_colsum = DIAG && (wr == wc);     // the diagonal waves also hold sum x^2 (diagonal of acc)

    // fragment of one k-step (16 rows) of a slab: two transpose reads (rows r0..r0+3 and r0+4..r0+7 of the lane's
    // 8-row half) give the 8 consecutive k that the 32x32x16 MFMA wants per lane
    auto frag = [&](const char* slab, int ks, int col0) -> uint4 {
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
    // SHIFT: x - c = x' + e on a fragment (8 consecutive rows of ONE column per lane: one scalar c, packed TwoSum);
    // `rows_left` = rows of the run left at the fragment's first row: elements past the end were loaded as zeros and must
    // stay zero (only the last stage of a run can have them; the hot loop passes 8).
    auto split2 = [&](uint4 f, uint32_t c2, int64_t rows_left, uint4& xs, uint4& es) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        uint32_t w[4] = {f.x, f.y, f.z, f.w}, x[4], e[4];
        h2 c; __builtin_memcpy(&c, &c2, 4);
        const h2 b = -c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h2 a; __builtin_memcpy(&a, &w[q], 4);
            const h2 sm = a + b;                       // Knuth's TwoSum: a + b = sm + er exactly
            const h2 bb = sm - a;
            const h2 er = (a - (sm - bb)) + (b - bb);
            __builtin_memcpy(&x[q], &sm, 4);
            __builtin_memcpy(&e[q], &er, 4);
            if (rows_left < 8) {
                const uint32_t m = ((2 * q < rows_left) ? 0xffffu : 0u) | ((2 * q + 1 < rows_left) ? 0xffff0000u : 0u);
                x[q] &= m; e[q] &= m;
            }
        }
        xs = make_uint4(x[0], x[1], x[2], x[3]);
        es = make_uint4(e[0], e[1], e[2], e[3]);
    };
    // this lane's shifts (packed twice) for the fragments it reads: off the diagonal / diagonal waves: columns
    // 64 wr + {0, 32} + li of the A side and 64 wc + {0, 32} + li of the B side; the other two waves of a diagonal tile:
    // columns {0, 32, 64, 96} + li of the (only) slab
    uint32_t cs[4] = {0u, 0u, 0u, 0u};
    if constexpr (SHIFT) {
        const uint16_t* cv = cvec + (int64_t)split * (nt * H_BT);
        const bool four = DIAG && (wr != wc);
        const int col[4] = {four ? ca + li : ca + 64 * wr + li, four ? ca + 32 + li : ca + 64 * wr + 32 + li,
                            four ? ca + 64 + li : cb + 64 * wc + li, four ? ca + 96 + li : cb + 64 * wc + 32 + li};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t h = cv[col[q]]; cs[q] = h | (h << 16); }
    }
    auto rows_left_at = [&](int kb, int ks) -> int64_t { return k_end - (k_begin + (int64_t)kb * H_KB + ks * 16 + 8 * kg); };
    // F[0], F[1] = the wave's two A-side fragments, F[2], F[3] = its two B-side fragments of k-step (kb, ks)
    auto load_frags = [&](int kb, int ks, uint4 (&F)[4], uint4 (&R)[4], bool full) {
        const char* sA = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        const char* sB = DIAG ? sA : sA + H_KB * 256;
        F[0] = frag(sA, ks, 64 * wr); F[1] = frag(sA, ks, 64 * wr + 32);
        F[2] = frag(sB, ks, 64 * wc); F[3] = frag(sB, ks, 64 * wc + 32);
        if constexpr (SHIFT) {
            const int64_t left = full ? 8 : rows_left_at(kb, ks);
#pragma unroll
            for (int q = 0; q < 4; ++q) split2(F[q], cs[q], left, F[q], R[q]);
        }
    };
    auto mma = [&](const uint4 (&F)[4]) {
        acc[0][0] = mfma_h16<KIND>(F[0], F[2], acc[0][0]);
        acc[0][1] = mfma_h16<KIND>(F[0], F[3], acc[0][1]);
        acc[1][0] = mfma_h16<KIND>(F[1], F[2], acc[1][0]);
        acc[1][1] = mfma_h16<KIND>(F[1], F[3], acc[1][1]);
    };
    // SHIFT: + x' e + e x' (R = the rounding errors that go with F)
    auto mma_err = [&](const uint4 (&F)[4], const uint4 (&R)[4]) {
        acc[0][0] = mfma_h16<KIND>(F[0], R[2], acc[0][0]); acc[0][0] = mfma_h16<KIND>(R[0], F[2], acc[0][0]);
        acc[0][1] = mfma_h16<KIND>(F[0], R[3], acc[0][1]); acc[0][1] = mfma_h16<KIND>(R[0], F[3], acc[0][1]);
        acc[1][0] = mfma_h16<KIND>(F[1], R[2], acc[1][0]); acc[1][0] = mfma_h16<KIND>(R[1], F[2], acc[1][0]);
        acc[1][1] = mfma_h16<KIND>(F[1], R[3], acc[1][1]); acc[1][1] = mfma_h16<KIND>(R[1], F[3], acc[1][1]);
    };
    // A DIAGONAL tile is symmetric, so only 20 of the 32 MFMAs of a stage are issued:
    //   * the waves on the tile's diagonal (wr == wc) own a symmetric 64 x 64 block: A and B fragments coincide
    //     (8 transpose reads instead of 16) and the lower 32 x 32 block is skipped -- 3 MFMAs per k-step;
    //   * the block (rows 0..63, columns 64..127) is shared by the two remaining waves: wave (0,1) takes k-step 0 of
    //     every stage, wave (1,0) k-step 1 (4 MFMAs, 8 reads each); wave (1,0) stores its half-sum in the unused
    //     lower-left slots of the partial tile and moments_reduce adds the two (reduce_body, "mirror").
    // The role is a COMPILE-TIME argument and the run loop below is entered once per role: with a (wave-uniform) run-time
    // test inside one loop, hipcc kept the accumulators of the off-diagonal waves in a second register range and copied
    // all 64 of them back and forth around their four MFMAs -- 80 v_accvgpr_mov per stage (found in the ISA, round 2).
    const bool diag_wave = wr == wc;
    // WALK: lane l of a diagonal wave walks column ca + 64 wr + l down the stage's 32 rows IN ORDER, one float32 add per row (what
    // np.mean of the file does): eight transpose reads -- group g of the wave points at columns 16 g .. 16 g + 15 of rows 4 q .. 4 q + 3
    // and every lane receives those four rows of ITS column -- then 32 dependent adds.  Rows past the end of the file are zeros in LDS
    // (s + 0 = s).  ~250 cycles of a wave that waits for HBM at D = 128 anyway; the frames are not read a second time.
    float walk_s = 0.f;
    auto walk_stage = [&](const char* sA) {
        if constexpr (WALK) {
            const int wcol = 64 * wr + 16 * grp + 4 * (t16 & 3);
            s16x4 h[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r0 = 4 * q + (t16 >> 2);
                const int o = r0 * 256 + (((wcol >> 3) ^ ((r0 & 3) << 2)) << 4) + ((wcol >> 2) & 1) * 8;
                h[q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sA + o));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) walk_s = walk_s + h16_to_f32<KIND>((uint32_t)(uint16_t)h[q][j]);
        }
    };
    auto stage_diag = [&](int kb, auto role_tag, bool full) {
        const char* sA = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        if constexpr (decltype(role_tag)::value) {
            walk_stage(sA);
            uint4 A0[2], A1[2];
            A0[0] = frag(sA, 0, 64 * wr); A0[1] = frag(sA, 0, 64 * wr + 32);
            A1[0] = frag(sA, 1, 64 * wr); A1[1] = frag(sA, 1, 64 * wr + 32);
            if constexpr (SHIFT) {
                const int64_t l0 = full ? 8 : rows_left_at(kb, 0), l1 = full ? 8 : rows_left_at(kb, 1);
                uint4 E0[2], E1[2];
                split2(A0[0], cs[0], l0, A0[0], E0[0]); split2(A0[1], cs[1], l0, A0[1], E0[1]);
                split2(A1[0], cs[0], l1, A1[0], E1[0]); split2(A1[1], cs[1], l1, A1[1], E1[1]);
                acc[0][0] = mfma_h16<KIND>(A0[0], E0[0], acc[0][0]); acc[0][0] = mfma_h16<KIND>(E0[0], A0[0], acc[0][0]);
                acc[0][1] = mfma_h16<KIND>(A0[0], E0[1], acc[0][1]); acc[0][1] = mfma_h16<KIND>(E0[0], A0[1], acc[0][1]);
                acc[1][1] = mfma_h16<KIND>(A0[1], E0[1], acc[1][1]); acc[1][1] = mfma_h16<KIND>(E0[1], A0[1], acc[1][1]);
                acc[0][0] = mfma_h16<KIND>(A1[0], E1[0], acc[0][0]); acc[0][0] = mfma_h16<KIND>(E1[0], A1[0], acc[0][0]);
                acc[0][1] = mfma_h16<KIND>(A1[0], E1[1], acc[0][1]); acc[0][1] = mfma_h16<KIND>(E1[0], A1[1], acc[0][1]);
                acc[1][1] = mfma_h16<KIND>(A1[1], E1[1], acc[1][1]); acc[1][1] = mfma_h16<KIND>(E1[1], A1[1], acc[1][1]);
                csum[0] += (double)sum8<KIND>(E0[0]) + (double)sum8<KIND>(E1[0]);
                csum[1] += (double)sum8<KIND>(E0[1]) + (double)sum8<KIND>(E1[1]);
            }
            acc[0][0] = mfma_h16<KIND>(A0[0], A0[0], acc[0][0]);
            acc[0][1] = mfma_h16<KIND>(A0[0], A0[1], acc[0][1]);
            acc[1][1] = mfma_h16<KIND>(A0[1], A0[1], acc[1][1]);
            acc[0][0] = mfma_h16<KIND>(A1[0], A1[0], acc[0][0]);
            acc[0][1] = mfma_h16<KIND>(A1[0], A1[1], acc[0][1]);
            acc[1][1] = mfma_h16<KIND>(A1[1], A1[1], acc[1][1]);
            csum[0] += (double)sum8<KIND>(A0[0]) + (double)sum8<KIND>(A1[0]);
            csum[1] += (double)sum8<KIND>(A0[1]) + (double)sum8<KIND>(A1[1]);
        } else {
            uint4 F[4];
            F[0] = frag(sA, wr, 0); F[1] = frag(sA, wr, 32);       // wave (0,1): k-step 0, wave (1,0): k-step 1
            F[2] = frag(sA, wr, 64); F[3] = frag(sA, wr, 96);
            if constexpr (SHIFT) {
                const int64_t left = full ? 8 : rows_left_at(kb, wr);
                uint4 R[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) split2(F[q], cs[q], left, F[q], R[q]);
                mma_err(F, R);
            }
            mma(F);
        }
    };

    // one stage: wait for it, workgroup barrier, refill the freed slot, 16 transpose reads, 8 MFMAs
    auto stage = [&](int kb, auto refill_tag, auto role_tag) {
        // stage kb must have landed; up to NST-2 younger stages may stay in flight
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        wait_vmcnt_upto<LPS>(ahead);
        __builtin_amdgcn_s_barrier();              // stage kb is in LDS; stage kb-1 is free
        if (decltype(refill_tag)::value) issue_fast(kb + NST - 1);
        else if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        constexpr bool full = decltype(refill_tag)::value;      // hot loop: this stage and the refilled one are whole
        if constexpr (DIAG) {
            stage_diag(kb, role_tag, full);
        } else {
            // all 16 transpose reads of the stage are issued up front (the compiler waits with lgkmcnt(0) before the
            // first MFMA; software-pipelining the reads one k-step or one stage ahead measured no gain -- DESIGN.md)
            uint4 F0[4], F1[4], R0[4], R1[4];
            load_frags(kb, 0, F0, R0, full);
            load_frags(kb, 1, F1, R1, full);
            mma(F0);
            mma(F1);
            if constexpr (SHIFT) { mma_err(F0, R0); mma_err(F1, R1); }
        }
    };
    // One run = a range of rows streamed through the ring: prologue, hot loop (the stage to refill is a full one ->
    // SGPR-base loads only), tail with the general loads.  A split is one run, or -- segment-aligned splits -- the
    // runs of several files back to back: the accumulators carry over, the column sums are flushed per run.
    double total_rows = 0.0;
    const int n_runs = runs ? run_hi - run_lo : 1;
    for (int ri = 0; ri < n_runs; ++ri) {
        int64_t crow = split;                      // colpart row of this run
        if (runs) {
            const SegRun rn = runs[run_lo + ri];
            k_begin = rn.r0; k_end = rn.r0 + rn.rows; crow = run_lo + ri;
            if (ri) __builtin_amdgcn_s_barrier();  // every wave has left the previous run's last stage: its slots are free
        }
        nkb = (int)((k_end - k_begin + H_KB - 1) / H_KB);
        nfast = (FAST && cols_full) ? (int)((k_end - k_begin) / H_KB) : 0;
        total_rows += (double)(k_end - k_begin);
        for (int s0 = 0; s0 < NST - 1 && s0 < nkb; ++s0) { if (s0 < nfast) issue_fast(s0); else issue(s0); }
        const int hot = (nfast - (NST - 1) > 0) ? nfast - (NST - 1) : 0;
        auto run_stages = [&](auto role_tag) {
            int kb = 0;
            for (; kb < hot; ++kb) stage(kb, std::true_type{}, role_tag);
            for (; kb < nkb; ++kb) stage(kb, std::false_type{}, role_tag);
        };
        if (DIAG && diag_wave) run_stages(std::true_type{});
        else run_stages(std::false_type{});
        if constexpr (WALK) {
            if (diag_wave && runs) {               // this file's running sums (one run per file on WALK launches), then the next file's start
                const int col = ca + 64 * wr + lane;
                if (col < d) seg_runsum[(int64_t)runs[run_lo + ri].seg * d + col] = walk_s;
                walk_s = 0.f;
            }
        }
        if (do_colsum) {
            csum[0] += __shfl_xor(csum[0], 32);
            csum[1] += __shfl_xor(csum[1], 32);
            if (kg == 0) {
                double* cp = colpart + crow * (int64_t)(nt * H_BT) + cb + 64 * wc + li;
                cp[0] = csum[0]; cp[32] = csum[1];
            }
            ctot[0] += csum[0]; ctot[1] += csum[1];
            csum[0] = 0.0; csum[1] = 0.0;
        }
    }

    // partial tile, fragment major: float4 index ((fa*4 + fb)*4 + q)*64 + lane holds registers 4q..4q+3 of the
    // 32 x 32 block (fa, fb) = rows 32fa + 8q + 4(lane>>5) + 0..3 of column 32fb + (lane&31);
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
    if (do_colsum && shift_flag) {
        // Shift guard (see moments_tile_f64): within this split's rows, is any column's mean^2 > 64 var?
        // Then fp32 partial sums of x^2 cannot resolve the variance and the block is redone in fp64.
        // sum x^2 of column (32 f + li) is the diagonal element acc[f][f][reg] of the lane whose C/D row
        // (reg&3) + 8 (reg>>2) + 4 kg equals li: kg = (li>>2)&1, reg = (li&3) + 4 (li>>3).
        const double nr = total_rows;
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
            const double mean = ctot[f] / nr, var = s2 / nr - mean * mean;
            const bool col_in = (cb + 64 * wc + 32 * f + li) < d;
            const bool hit_f = col_in && !(mean * mean <= 64.0 * var) && !(ctot[f] == 0.0 && s2 == 0.0);
            hit = hit || hit_f;
            if (cvec && kg == 0) {          // this split's shift for the second pass: the column's mean on the float16 grid, or none
                const bool worth = col_in && (mean * mean > var) && (mean == mean) && !isinf(mean) && fabs(mean) < 65000.0;
                const _Float16 ch = worth ? (_Float16)(float)mean : (_Float16)0.0f;
                uint16_t bits; __builtin_memcpy(&bits, &ch, 2);
                cvec[(int64_t)split * (nt * H_BT) + cb + 64 * wc + 32 * f + li] = bits;
            }
        }
        if (__any(hit) && lane == 0) atomicOr(shift_flag, 1);
    }
}

template <int KIND, int NST, bool FAST, bool SHIFT = false, bool WALK = false>
__global__ __launch_bounds__(256) void moments_tile_h16_tr(TileLaunch L) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];     // the ONLY LDS object: NST x 16 KiB
    const int w = xcd_contiguous(blockIdx.x, L.total);
    int split, tile, run_lo, run_hi; int64_t k_begin, k_end;
    const TileSet& s = locate(L, w, split, tile, k_begin, k_end, run_lo, run_hi);
    if constexpr (SHIFT) {
        if (!s.flag || *s.flag == 0) return;       // second pass of the shift guard: only for sets whose first pass raised the flag
    }
    int ta, tb; tile_coords(tile, L.nt, ta, tb);
    const uint16_t* E = static_cast<const uint16_t*>(s.E);
    float* partials = static_cast<float*>(s.partials);
    // (the second pass neither re-examines the columns nor rewrites the shifts: no flag, but the shifts to read)
    if (ta == tb)
        tile_h16_tr_body<KIND, NST, true, FAST, SHIFT, WALK>(E, k_begin, k_end, s.ld, L.d, L.nt, L.T, split, tile, ta * H_BT, tb * H_BT,
                                                             partials, s.colpart, smem_dyn, SHIFT ? nullptr : s.flag, s.runs, run_lo, run_hi,
                                                             s.cvec, s.seg_runsum);
    else if constexpr (FAST)
        tile_h16_tr_body<KIND, NST, false, FAST, SHIFT>(E, k_begin, k_end, s.ld, L.d, L.nt, L.T, split, tile, ta * H_BT, tb * H_BT,
                                                        partials, s.colpart, smem_dyn, nullptr, s.runs, run_lo, run_hi, s.cvec);
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
//
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
__global__ __launch_bounds__(256) void moments_tile_f64(TileLaunch L) {
    __shared__ double smem[2][2][G_KB * G_LDS];      // 40 KiB
    // one grid for all sets (sized for the longest); a set whose gate is clear has nothing to redo
    const TileSet& s = L.set[blockIdx.y];
    if ((int)blockIdx.x >= s.S * L.T) return;
    if (s.flag && *s.flag == 0) return;              // shift guard: only runs when the fp16 pass flagged the block
    const TIn* __restrict__ E = static_cast<const TIn*>(s.E);
    const int64_t ld = s.ld;
    const int d = L.d, nt = L.nt, T = L.T;

    const int w = xcd_contiguous(blockIdx.x, s.S * T);
    const int split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, nt, ta, tb);
    const bool diag = (ta == tb);
    const int ca = ta * G_BT, cb = tb * G_BT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;

    const int64_t k_begin = (int64_t)split * s.rows_per_split;
    const int64_t k_end = (k_begin + s.rows_per_split < s.n) ? k_begin + s.rows_per_split : s.n;
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

    double* out = static_cast<double*>(s.partials) + ((int64_t)split * T + tile) * G_TS;
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
            double* cp = s.colpart + (int64_t)split * (nt * G_BT) + cb + 32 * wc + li;
            cp[0] = csum[0]; cp[16] = csum[1];
        }
    }
}

// ------------------------------------------------------------------------------------------
// partials -> packed fp64 accumulator (sum over splits in fp64, fixed order => deterministic).
// One thread per 4 adjacent elements of one tile.
// ------------------------------------------------------------------------------------------
struct SplitPlan { int nt, T, S; int64_t rows_per_split; };

// One source of partial sums for moments_reduce: `S` row-splits x `T` tiles (+ column partials).
struct ReduceSrc {
    const void* partials; const double* colpart;
    int S, T, nt;
    int SC;               // rows of colpart (= S, or the number of runs of segment-aligned splits)
    int tile_blocks;      // workgroups that sum tiles; the following ceil(d/256) sum the columns and the row count
    int layout;           // 0 row major, 1 fragment major (the fp16 kernels)
    int sl;               // "split lanes" (1, 4 or 16), see below
    // second pass of the shift guard (tile_h16_tr_body, SHIFT): when the gate is up the partials are sums over x - c and the
    // raw moments come back as S' + c s'^T + s' c^T + n c c^T per split (cvec = the shifts, colpart = s', rows per split below)
    const uint16_t* cvec;
    int64_t rows_per_split, n_rows;
};

// One set of a reduce launch: accumulator += (or =) the sum over splits of ONE of two sources: `prim` when *gate == 0
// or there is no gate, else `alt` -- the fp64 redo of the block by moments_tile_f64 (shift guard) -- or, when prim.cvec is
// set, `prim` again, rewritten by the second pass and un-shifted here.
struct ReduceJob {
    ReduceSrc prim, alt;
    double* acc; double n_add;
    const int* gate; int* clear_flag;
    int overwrite;        // the accumulator was reset since its last update: store instead of add (saves the memset)
};
struct ReduceLaunch { ReduceJob job[kMaxSets]; int d; };

// sl "split lanes" share one output group: thread (l, g) sums splits l, l+sl, ... and the sl partial
// sums are combined through LDS in a fixed order.  With hundreds of row-splits (D = 128 uses every
// workgroup slot for one tile) a single thread per output would walk all of them serially.
__device__ __forceinline__ double f16_bits_to_f64(uint16_t b) { _Float16 h; __builtin_memcpy(&h, &b, 2); return (double)(float)h; }

template <typename PT, int BT>
__device__ __forceinline__ void reduce_body(const ReduceSrc& r, int d, double* __restrict__ acc_packed, double n_add,
                                            bool overwrite, int block, double* red, bool unshift = false) {
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
        const int SC = r.SC;
        int sp = 0;
        for (; sp + 7 < SC; sp += 8) {             // eight loads in flight: this walk is a latency chain (a quarter of a 16 us reduce)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = colpart[(int64_t)(sp + u) * dpad + a];
            s0 += (v[0] + v[1]) + (v[2] + v[3]); s1 += (v[4] + v[5]) + (v[6] + v[7]);
        }
        for (; sp + 1 < SC; sp += 2) { s0 += colpart[(int64_t)sp * dpad + a]; s1 += colpart[(int64_t)(sp + 1) * dpad + a]; }
        if (sp < SC) s0 += colpart[(int64_t)sp * dpad + a];
        if (unshift) {                             // sum x = s' + n c, split by split
            for (int q = 0; q < SC; ++q) {
                const int64_t left = r.n_rows - (int64_t)q * r.rows_per_split;
                const double nq = (double)(left < r.rows_per_split ? left : r.rows_per_split);
                s1 += nq * f16_bits_to_f64(r.cvec[(int64_t)q * dpad + a]);
            }
        }
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
        } else {                           // fragment major: 4 adjacent ROWS of one column
            const int el = e & 63;
            a_local = 32 * (e >> 10) + 8 * ((e >> 6) & 3) + 4 * (el >> 5);
            b_local = 32 * ((e >> 8) & 3) + (el & 31);
        }
        constexpr int TS = (sizeof(PT) == 4) ? BT * BT + 64 : BT * BT + 32;      // H_TS / G_TS
        const PT* p = partials + (int64_t)tile * TS + e * 4;
        const int64_t stride = (int64_t)T * TS;
        // "mirror": on a DIAGONAL tile moments_tile_h16_tr splits the k-steps of the block (rows 0..63, columns 64..127)
        // between two waves; the second half-sum sits in the lower-left slots, 32 x 32 block (fa + 2, fb - 2) =
        // 6 * 256 float4 further on (zeros when a presum wrote the tile)
        int nsrc = 1;
        if (r.layout == 1) {
            int ta0, tb0; tile_coords(tile, nt, ta0, tb0);
            if (ta0 == tb0 && (e >> 10) < 2 && ((e >> 8) & 3) >= 2) nsrc = 2;
        }
        for (int h = 0; h < nsrc; ++h) {
            const PT* ph = p + h * (6 * 256 * 4);
            int sp = sl;
            if constexpr (sizeof(PT) == 4) {           // four independent loads in flight per thread
                for (; sp + 3 * SL < S; sp += 4 * SL) {
                    const float4 v0 = *reinterpret_cast<const float4*>(ph + sp * stride);
                    const float4 v1 = *reinterpret_cast<const float4*>(ph + (sp + SL) * stride);
                    const float4 v2 = *reinterpret_cast<const float4*>(ph + (sp + 2 * SL) * stride);
                    const float4 v3 = *reinterpret_cast<const float4*>(ph + (sp + 3 * SL) * stride);
                    s[0] += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
                    s[1] += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
                    s[2] += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
                    s[3] += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
                }
            }
            if constexpr (sizeof(PT) == 8) {           // (float64 partials: the fp64 kernel's, or the first stage's sums)
                for (; sp + 3 * SL < S; sp += 4 * SL) {
                    double2 v[8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[2 * u] = *reinterpret_cast<const double2*>(ph + (sp + u * SL) * stride);
                        v[2 * u + 1] = *reinterpret_cast<const double2*>(ph + (sp + u * SL) * stride + 2);
                    }
                    s[0] += (v[0].x + v[2].x) + (v[4].x + v[6].x); s[1] += (v[0].y + v[2].y) + (v[4].y + v[6].y);
                    s[2] += (v[1].x + v[3].x) + (v[5].x + v[7].x); s[3] += (v[1].y + v[3].y) + (v[5].y + v[7].y);
                }
            }
            for (; sp < S; sp += SL) {
                if constexpr (sizeof(PT) == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(ph + sp * stride);
                    s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
                } else {
                    const double2 v0 = *reinterpret_cast<const double2*>(ph + sp * stride);
                    const double2 v1 = *reinterpret_cast<const double2*>(ph + sp * stride + 2);
                    s[0] += v0.x; s[1] += v0.y; s[2] += v1.x; s[3] += v1.y;
                }
            }
        }
    }
    if (unshift && live) {
        // + c_a s'_b + s'_a c_b + n c_a c_b for this thread's splits (fp64; the shifts are zero on ordinary columns)
        int ta0, tb0; tile_coords(tile, nt, ta0, tb0);
        const int dpad = nt * BT;
        const int ga = ta0 * BT + a_local, gb = tb0 * BT + b_local;      // layout 1: rows ga..ga+3 of column gb
        for (int sp = sl; sp < S; sp += SL) {
            const int64_t left = r.n_rows - (int64_t)sp * r.rows_per_split;
            const double nq = (double)(left < r.rows_per_split ? left : r.rows_per_split);
            const uint16_t* cv = r.cvec + (int64_t)sp * dpad;
            const double* sv = r.colpart + (int64_t)sp * dpad;
            const double cb_ = f16_bits_to_f64(cv[gb]), sb_ = sv[gb];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double ca_ = f16_bits_to_f64(cv[ga + q]), sa_ = sv[ga + q];
                s[q] += ca_ * sb_ + sa_ * cb_ + nq * ca_ * cb_;
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

// blockIdx.y = set; the grid's x extent is sized for the largest job.
template <typename PTA, int BTA>
__global__ __launch_bounds__(256) void moments_reduce(ReduceLaunch R) {
    __shared__ double red[256 * 4];
    const ReduceJob& j = R.job[blockIdx.y];
    if (j.clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *j.clear_flag = 0;     // next update's flag
    const int col_blocks = (R.d + 255) / 256;
    if (j.gate && *j.gate != 0 && j.prim.cvec) {
        if ((int)blockIdx.x < j.prim.tile_blocks + col_blocks)
            reduce_body<PTA, BTA>(j.prim, R.d, j.acc, j.n_add, j.overwrite != 0, (int)blockIdx.x, red, true);
    } else if (j.gate && *j.gate != 0) {
        if ((int)blockIdx.x < j.alt.tile_blocks + col_blocks)
            reduce_body<double, 64>(j.alt, R.d, j.acc, j.n_add, j.overwrite != 0, (int)blockIdx.x, red);
    } else if ((int)blockIdx.x < j.prim.tile_blocks + col_blocks) {
        reduce_body<PTA, BTA>(j.prim, R.d, j.acc, j.n_add, j.overwrite != 0, (int)blockIdx.x, red);
    }
}

// Stage 1 of the two-level reduce used when an update produced hundreds or thousands of partial tiles (long inputs
// at small D: every run of <= 8192 rows is one split).  Block (x, c) sums the splits of chunk c for 256 output
// groups (four loads in flight per thread) into fp64 partials laid out like moments_reduce<double, BT> expects;
// trailing x-blocks do the same for the column partials.
#ifndef FAD_PRESUM_CHUNK
#define FAD_PRESUM_CHUNK 32
#endif
constexpr int PRESUM_CHUNK = FAD_PRESUM_CHUNK;
template <int BT>
__global__ __launch_bounds__(256) void moments_presum(
    const float* __restrict__ partials, const double* __restrict__ colpart, int S, int SC, int T, int nt, int group_blocks,
    double* __restrict__ partials2, double* __restrict__ colpart2, const int* __restrict__ gate,
    const uint16_t* __restrict__ cvec, int64_t rows_per_split, int64_t n_rows) {
    // gate up: the block is being redone in fp64 (nothing to pre-sum) -- or, with `cvec`, the partials are the second pass's
    // sums over x - c and are un-shifted here, split by split (see reduce_body)
    const bool unshift = gate && *gate != 0 && cvec;
    if (gate && *gate != 0 && !cvec) return;
    auto rows_of = [&](int sp) -> double {
        const int64_t left = n_rows - (int64_t)sp * rows_per_split;
        return (double)(left < rows_per_split ? left : rows_per_split);
    };
    const int c = blockIdx.y;
    const int s0 = c * PRESUM_CHUNK, s1 = (s0 + PRESUM_CHUNK < S) ? s0 + PRESUM_CHUNK : S;
    const int dpad = nt * BT;
    if ((int)blockIdx.x >= group_blocks) {
        // the colpart rows (SC of them: one per split, or one per run -- 4096 files of a config-4 update) are shared out evenly over
        // the chunks; eight loads in flight per thread, and at D = 128 the block's other 128 threads take every second row (the walk
        // of 256 rows, two at a time, was most of this kernel's 35 us)
        __shared__ double colred[256];
        const int lanes = (dpad <= 128) ? 2 : 1, cpb = 256 / lanes;
        const int a = ((int)blockIdx.x - group_blocks) * cpb + (int)threadIdx.x % cpb, l = (int)threadIdx.x / cpb;
        const int per = (SC + (int)gridDim.y - 1) / (int)gridDim.y;
        const int c0 = c * per, c1 = (c0 + per < SC) ? c0 + per : SC;
        double t = 0.0;
        if (a < dpad) {
            int sp = c0 + l;
            for (; sp + 7 * lanes < c1; sp += 8 * lanes) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = colpart[(int64_t)(sp + u * lanes) * dpad + a];
                t += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; sp < c1; sp += lanes) t += colpart[(int64_t)sp * dpad + a];
            if (unshift)
                for (int q = c0 + l; q < c1; q += lanes) t += rows_of(q) * f16_bits_to_f64(cvec[(int64_t)q * dpad + a]);
        }
        colred[threadIdx.x] = t;
        __syncthreads();
        if (l == 0 && a < dpad) colpart2[(int64_t)c * dpad + a] = (lanes == 2) ? colred[threadIdx.x] + colred[threadIdx.x + cpb] : t;
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
    for (; sp + 7 < s1; sp += 8) {                 // eight loads in flight (a chunk of 8 splits: all of them)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (sp + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            s[0] += ((double)v[u].x + (double)v[u + 1].x) + ((double)v[u + 2].x + (double)v[u + 3].x);
            s[1] += ((double)v[u].y + (double)v[u + 1].y) + ((double)v[u + 2].y + (double)v[u + 3].y);
            s[2] += ((double)v[u].z + (double)v[u + 1].z) + ((double)v[u + 2].z + (double)v[u + 3].z);
            s[3] += ((double)v[u].w + (double)v[u + 1].w) + ((double)v[u + 2].w + (double)v[u + 3].w);
        }
    }
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
    if (unshift) {
        int ta0, tb0; tile_coords(tile, nt, ta0, tb0);
        // (the lower-left 64 x 64 block of a DIAGONAL tile holds the second half-sums of its upper-right block -- "mirror",
        // added by the second stage under the upper-right coordinates -- or nothing: no correction there)
        const bool mirror_slot = ta0 == tb0 && (e >> 10) >= 2 && ((e >> 8) & 3) < 2;
        if (!mirror_slot) {
            const int el = e & 63;
            const int ga = ta0 * BT + 32 * (e >> 10) + 8 * ((e >> 6) & 3) + 4 * (el >> 5);
            const int gb = tb0 * BT + 32 * ((e >> 8) & 3) + (el & 31);
            for (int q = s0; q < s1; ++q) {
                const double nq = rows_of(q);
                const uint16_t* cv = cvec + (int64_t)q * dpad;
                const double* sv = colpart + (int64_t)q * dpad;
                const double cb_ = f16_bits_to_f64(cv[gb]), sb_ = sv[gb];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double ca_ = f16_bits_to_f64(cv[ga + k]), sa_ = sv[ga + k];
                    s[k] += ca_ * sb_ + sa_ * cb_ + nq * ca_ * cb_;
                }
            }
        }
    }
    double* o = partials2 + ((int64_t)c * T + tile) * TS64 + e * 4;
    *reinterpret_cast<double2*>(o) = make_double2(s[0], s[1]);
    *reinterpret_cast<double2*>(o + 2) = make_double2(s[2], s[3]);
}

static ReduceSrc reduce_src(const void* part, const double* colp, const SplitPlan& p, int bt, int layout) {
    ReduceSrc r;
    r.cvec = nullptr; r.rows_per_split = 0; r.n_rows = 0;
    r.partials = part; r.colpart = colp; r.S = p.S; r.SC = p.S; r.T = p.T; r.nt = p.nt; r.layout = layout;
    r.sl = (p.S > 64) ? 16 : (p.S > 8) ? 4 : 1;
    r.tile_blocks = (int)cdiv((int64_t)p.T * (bt * bt / 4), 256 / r.sl);
    return r;
}

// ------------------------------------------------------------------------------------------
// Per-segment column sums (files / songs stored back to back): seg_sums[s][a] = sum over the rows of segment s of
// E[r][a], fp64.  Two stages, deterministic: every "piece" (<= SEG_PIECE rows of ONE segment) is summed by one
// workgroup whose threads each own 8 adjacent columns (16-byte loads of fp16 rows) of every (256 / lanes-per-row)-th
// row; then one thread per (segment, column) adds that segment's pieces in order.  Short segments are single
// pieces; a 2250-row file (config 4) is 9 pieces, so a batch of 64 files already fills the chip.
// ------------------------------------------------------------------------------------------
constexpr int SEG_PIECE = 256;
struct SegPiece { int64_t r0; int rows; int seg; };

template <typename TIn>
__global__ __launch_bounds__(256) void segment_piece_sums(
    const TIn* __restrict__ E, int64_t ld, int d, const SegPiece* __restrict__ pieces, double* __restrict__ piece_sums) {
    __shared__ double red[256 * 8];
    const SegPiece pc = pieces[blockIdx.x];
    const int cg = (d + 7) / 8;                    // column groups of 8
    const int tid = threadIdx.x;
    for (int g0 = 0; g0 < cg; g0 += 256) {         // d <= 2048: one pass
        const int lanes = (cg - g0 < 256) ? cg - g0 : 256;          // threads that own a column group
        const int rpi = 256 / lanes;                                // rows handled per iteration
        const int g = g0 + tid % lanes, rl = tid / lanes;
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (rl < rpi) {
            for (int r = rl; r < pc.rows; r += rpi) {
                const TIn* p = E + (pc.r0 + r) * ld + g * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (g * 8 + q < d) s[q] += to_f64<TIn>(p[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) red[tid * 8 + q] = s[q];
        __syncthreads();
        if (tid < lanes) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double t = 0.0;
                for (int l = 0; l < rpi; ++l) t += red[(l * lanes + tid) * 8 + q];
                if (g * 8 + q < d) piece_sums[(int64_t)blockIdx.x * d + g * 8 + q] = t;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(128) void segment_gather_sums(const double* __restrict__ piece_sums, int64_t pitch,
                                                           const int64_t* __restrict__ seg_first_piece, int d,
                                                           double* __restrict__ seg_sums) {
    const int64_t seg = blockIdx.x;
    const int a = blockIdx.y * 128 + threadIdx.x;
    if (a >= d) return;
    double t = 0.0;
    for (int64_t p = seg_first_piece[seg]; p < seg_first_piece[seg + 1]; ++p) t += piece_sums[p * pitch + a];
    seg_sums[seg * d + a] = t;
}

// Per-file mean rows of the online statistics (fadtk/utils.py:16, 36-40), see fad_moments_update_file_means:
// exact[f] = sqrt(n_f) m_f, rounded[f] = sqrt(n_f) m~_f (m~ = the mean as np.mean returns it for `dtype`),
// weighted[f] = n_f m~_f; empty files give zero rows.
template <int DT>
__device__ __forceinline__ double round_mean_like(double v) {
    if constexpr (DT == FAD_F16) return (double)(float)(_Float16)(float)v;
    else if constexpr (DT == FAD_BF16) {
        uint32_t u = __float_as_uint((float)v);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (double)__uint_as_float(u & 0xffff0000u);
    } else if constexpr (DT == FAD_F32) return (double)(float)v;
    else return v;
}
// (seg_runsums: numpy's float32 running column sums of every file -- segment_running_sums below --, or nullptr: then m~ is the rounded
//  EXACT mean, which differs from np.mean's by one ulp in ~0.3 % of the columns of a long file whose frames carry an offset)
template <int DT>
__global__ __launch_bounds__(256) void file_mean_rows(const double* __restrict__ seg_sums, const int64_t* __restrict__ sizes,
                                                      int64_t n_files, int d, double* __restrict__ exact,
                                                      double* __restrict__ rounded, double* __restrict__ weighted,
                                                      const float* __restrict__ seg_runsums = nullptr) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_files * d) return;
    const int64_t f = g / d;
    const double n = (double)sizes[f];
    double m = 0.0, mr = 0.0;
    if (n > 0.0) { m = seg_sums[g] / n; mr = round_mean_like<DT>((seg_runsums && DT != FAD_F64) ? numpy_mean_of_f32_sum(seg_runsums[g], n) : m); }
    const double rt = sqrt(n);
    exact[g] = rt * m; rounded[g] = rt * mr; weighted[g] = n * mr;
}

__global__ __launch_bounds__(256) void packed_axpy(double* __restrict__ dst, const double* __restrict__ src,
                                                   int64_t len) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < len) dst[g] += src[g];
}

// The column sums the way the reference's np.mean(embd_lst, axis=0) forms them (fadtk/fad.py:48): numpy adds the rows one after the
// other in float32 (float16 frames are widened first), so a [100000 x 512] matrix of frames around 0.5 ends 1e-5 off the exact sum and
// the float16-rounded mean differs from the rounded exact mean by one ulp in a few dimensions -- 2e-5 .. 5e-4 of a small Frechet distance
// (measured with the oracle at offsets of 0.5 .. 3 standard deviations x 10).  There is no parallel form of a float32 running sum: one lane
// per column walks the rows in order (16 loads in flight), 0.3 ms per 100 k rows on a few CUs; opt-in per handle
// (fad_moments_set_reference_mean), carried across updates like numpy carries it across the rows of the concatenated matrix.
__device__ __forceinline__ float load_as_float(const raw_f16* p) { return h16_to_f32<FAD_F16>(p->b); }
__device__ __forceinline__ float load_as_float(const raw_bf16* p) { return h16_to_f32<FAD_BF16>(p->b); }
__device__ __forceinline__ float load_as_float(const float* p) { return *p; }
struct RunSumJob { const void* rows; int64_t n, ld; float* run; int start_zero; const int32_t* idx; };      // idx != nullptr: frame r = row idx[r] of `rows` (IDX instantiation)
struct RunSumLaunch { RunSumJob job[kMaxSets]; int d; const RunSumJob* table; };      // table != nullptr: job blockIdx.y is table[blockIdx.y] (device)
// the jobs of a segmented walk (per-file / per-song running sums): segment s = rows [offsets[s], offsets[s + 1]) -> out + s * d
__global__ __launch_bounds__(256) void runsum_segment_jobs(const uint16_t* __restrict__ rows, int64_t ld, int d, const int64_t* __restrict__ offsets,
                                                           int64_t n_segments, float* __restrict__ out, RunSumJob* __restrict__ jobs) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_segments) return;
    RunSumJob j;
    j.rows = rows + offsets[s] * ld; j.n = offsets[s + 1] - offsets[s]; j.ld = ld; j.run = out + s * d; j.start_zero = 1;
    jobs[s] = j;
}
// Workgroup = 16 columns (d / 16 workgroups per matrix: the walk is latency-bound per CU -- ~32 KB in flight against ~2 us -- so it
// is spread over many CUs): waves 1..3 stage tiles of 1024 rows x 16 columns through LDS with 16-byte loads (two tiles: the next one travels
// while this one is walked), lanes 0..15 of wave 0 walk their columns down the tile -- the dependent float32 adds are the critical path.
// Measured for [100000 x 512] float16: one lane per column reading global memory directly 2.4 ms (sixteen two-byte loads of latency at a
// time); 64 columns per workgroup, every wave staging and wave 0 adding 7.9 ms (the adds waited for the wave's own loads); 64 columns with a
// dedicated adding wave 1.8 ms; 16 columns per workgroup with tiles of 256 rows 0.59 ms (one load latency per tile); this form: DESIGN.md 4.1b.  WIDE = rows and pitch 16-byte aligned.
constexpr int kRunRows = 1024, kRunCols = 16;                 // (a tile's loads cover one latency: 64 KiB a tile, two tiles of dynamic LDS)
constexpr size_t kRunLds = (size_t)2 * kRunRows * kRunCols * sizeof(float);
template <typename TIn, bool WIDE>
__global__ __launch_bounds__(256) void moments_running_colsum(RunSumLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float run_lds[];
    float (*tile)[kRunRows][kRunCols] = reinterpret_cast<float (*)[kRunRows][kRunCols]>(run_lds);      // [2]: rows of 16 floats, no padding needed
    const RunSumJob& j = L.job[blockIdx.y];
    const int c0 = blockIdx.x * kRunCols, tid = threadIdx.x;
    if (j.n <= 0) return;
    const TIn* base = static_cast<const TIn*>(j.rows);
    constexpr int EPV = 16 / (int)sizeof(TIn);                     // elements per 16-byte load: 8 (float16 / bfloat16) or 4 (float32)
    constexpr int CPR = kRunCols / EPV;                            // 16-byte chunks per row of the tile
    const int lt = tid - 64;                                       // loader index 0..191 (waves 1..3)
    auto stage = [&](int buf, int64_t r0) {
        if (lt < 0) return;
        if constexpr (WIDE) {
            constexpr int RPP = 192 / CPR;                         // rows per pass
            constexpr int NP = (kRunRows + RPP - 1) / RPP;
            const int q = lt % CPR, ro = lt / CPR;
            uint4 w[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {                         // every load of the tile in flight at once
                const int rr = p * RPP + ro;
                const int64_t r = r0 + rr;
                w[p] = (rr < kRunRows && r < j.n && c0 + q * EPV < L.d) ? *reinterpret_cast<const uint4*>(base + r * j.ld + c0 + q * EPV)
                                                                        : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int rr = p * RPP + ro;
                if (rr < kRunRows) {
                    const TIn* e = reinterpret_cast<const TIn*>(&w[p]);
#pragma unroll
                    for (int u = 0; u < EPV; ++u) tile[buf][rr][q * EPV + u] = load_as_float(e + u);
                }
            }
        } else {
            const int cl = lt % kRunCols, ro = lt / kRunCols;      // 12 rows x 16 columns per pass
            for (int rr = ro; rr < kRunRows; rr += 192 / kRunCols) {
                const int64_t r = r0 + rr;
                tile[buf][rr][cl] = (c0 + cl < L.d && r < j.n) ? load_as_float(base + r * j.ld + c0 + cl) : 0.f;
            }
        }
    };
    const bool col_ok = tid < kRunCols && c0 + tid < L.d;
    float s = (col_ok && !j.start_zero) ? j.run[c0 + tid] : 0.f;
    const int64_t ntiles = (j.n + kRunRows - 1) / kRunRows;
    stage(0, 0);
    __syncthreads();
    for (int64_t t = 0; t < ntiles; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntiles) stage(buf ^ 1, (t + 1) * kRunRows);    // waves 1..3: the next tile, while wave 0 adds
        if (tid < kRunCols) {
            const int64_t left = j.n - t * kRunRows;
            const int rows_here = left < kRunRows ? (int)left : kRunRows;
            if (rows_here == kRunRows) {
                for (int r0 = 0; r0 < kRunRows; r0 += 64) {        // 64 LDS reads ahead of the 64 dependent adds
                    float v[64];
#pragma unroll
                    for (int u = 0; u < 64; ++u) v[u] = tile[buf][r0 + u][tid];
#pragma unroll
                    for (int u = 0; u < 64; ++u) s = s + v[u];
                }
            } else {
                for (int r = 0; r < rows_here; ++r) s = s + tile[buf][r][tid];
            }
        }
        __syncthreads();
    }
    if (col_ok) j.run[c0 + tid] = s;
}

// The same walk for float16 rows (the reference's storage type, model_loader.py:47-48) in a form that runs BESIDE the 256-column tile
// kernel: 25 KiB of LDS and at most 64 VGPRs per wave -- what one CU has left next to a tile workgroup (128 KiB, 2 x 224 registers per
// SIMD) -- so that update_device_multi can put it on a stream of its own (moments.hip: running_sums).  What sets its pace is the
// instruction stream of the ONE wave that adds -- a lone wave issues an instruction every ~5-6 cycles, and every row costs a DEPENDENT add:
//   * the tile in LDS as float32, column-major ([16 columns][192 rows + 4]): one ds_read_b128 brings FOUR consecutive rows of the lane's
//     column (pitch 784 bytes: the lanes' 16-byte pieces of a lane group fall into different bank groups), then four plain v_add_f32,
//     three sets of 16 rows in rotation so that the reads are TWO sets ahead of the adds -- ~1.5 instructions per row, 9 cycles measured
//     (the kernel above: a 4-byte LDS read and an add per row, ~11.5; float16 in LDS with v_fma_mix_f32: 21 -- the compiler separates
//     dependent mix instructions by s_nop; float32 tiles with the reads one set ahead: 13 -- an LDS read takes ~130 cycles, r05a-c);
//   * waves 1..3 feed it: per tile a loader thread owns 16 columns of ONE row (two 16-byte loads, issued four tiles ahead (768 rows) and held in
//     registers meanwhile), widens its 16 values and writes them down the columns (the lanes of a wave write consecutive dwords).
// A walk wave on a CU (64 registers) keeps every workgroup of 256-register waves OUT of that CU: the gated second pass of the shift guard
// (moments_tile256<.., true>) was such a kernel -- with a walk workgroup on every CU even the launch that only reads its gate and exits
// could not be placed, and the caller's stream stood still for 170-300 us per update (r05d) -- and is now held to 224 registers
// (moments_tile256.h).  Workgroups whose columns share the rows' 128-byte lines are dealt to ONE XCD (b % 8).
// Two shapes of the same kernel (FAD_MOMENTS_RUNSUM_COLS, read once): <16 columns, 192-row tiles, four tiles in flight> -- the default:
// 9 cycles per row alone, 0.37 ms per 100 000 rows for up to eight matrices (d / 16 workgroups each: eight matrices of d = 512 put a
// workgroup on every CU) -- and <32 columns, 96-row tiles, five tiles in flight>, which leaves half the chip alone but pays its per-tile
// costs (the barrier, the first LDS reads of a tile) twice as often: 16-21 cycles per row (r05e, r05f).
typedef _Float16 rs_h2 __attribute__((ext_vector_type(2)));
template <int kRsCols, int kRsRows, int RING> struct RsShape {
    static constexpr int pitch = kRsRows + 4;                         // floats
    static constexpr size_t lds = (size_t)2 * kRsCols * pitch * sizeof(float);
};
constexpr size_t kRsLds = 25600;                                      // the larger of the two shapes' LDS (25 088 / 25 600 bytes)
template <int kRsCols, int kRsRows, int RING, bool IDX = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void moments_running_colsum_h16(RunSumLaunch L) {
    constexpr int kRsPitch = RsShape<kRsCols, kRsRows, RING>::pitch;
    static_assert(kRsRows * kRsCols == 192 * 16, "a loader thread owns two 16-byte pieces of a tile");
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];                          // [2][32][kRsPitch]
    // (the job comes through vector loads -- a table in global memory or a dynamically indexed kernel argument; its fields are the same for
    //  every lane and are moved to scalar registers, or four of the kernel's 64 VGPRs hold them and the adding wave spills)
    const RunSumJob jv = L.table ? L.table[blockIdx.y] : L.job[blockIdx.y];
    auto uni64 = [](uint64_t v) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); };
    RunSumJob j;
    j.rows = reinterpret_cast<const void*>(uni64(reinterpret_cast<uint64_t>(jv.rows)));
    j.run = reinterpret_cast<float*>(uni64(reinterpret_cast<uint64_t>(jv.run)));
    j.n = (int64_t)uni64((uint64_t)jv.n); j.ld = (int64_t)uni64((uint64_t)jv.ld);
    j.start_zero = __builtin_amdgcn_readfirstlane(jv.start_zero);
    j.idx = IDX ? reinterpret_cast<const int32_t*>(uni64(reinterpret_cast<uint64_t>(jv.idx))) : nullptr;
    if (j.n <= 0) {                                                  // (an empty segment's sums are zero; an empty set has no buffer to write)
        if (L.table && (int)threadIdx.x < kRsCols && blockIdx.x * kRsCols + threadIdx.x < L.d) j.run[blockIdx.x * kRsCols + threadIdx.x] = 0.f;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int cb = blockIdx.x;
    constexpr int G = 64 / kRsCols;                                  // column blocks per 128-byte line of the rows: dealt to one XCD (b % 8)
    if ((gridDim.x % (8 * G)) == 0) { const int w = blockIdx.x % (8 * G); cb = (blockIdx.x - w) + G * (w & 7) + (w >> 3); }
    const int c0 = cb * kRsCols;
    const uint16_t* base = static_cast<const uint16_t*>(j.rows);
    const int64_t ntiles = (j.n + kRsRows - 1) / kRsRows;
    auto rows_of = [&](int64_t t) { const int64_t left = j.n - t * kRsRows; return left < kRsRows ? (int)left : kRsRows; };
    // Two roles, two loops (wave-uniform branch; every wave passes the same barriers): the register sets of the loaders (five tiles in
    // flight) and of the adding wave (48 rows of operands) never live side by side -- the kernel has to stay within 64 VGPRs
    if (wave == 0) {
        // ---- the adding wave: lanes 0..31 walk their column down the tile, 32 rows of reads ahead of the adds
        const bool adder = lane < kRsCols;
        const bool col_ok = adder && c0 + lane < L.d;
        typedef __attribute__((address_space(1))) const float g_f32;
        float s = (col_ok && !j.start_zero) ? *(g_f32*)(uintptr_t)(j.run + c0 + lane) : 0.f;
        auto walk = [&](int buf, int rows_here) {
            const float4* col = reinterpret_cast<const float4*>(rs_lds + ((size_t)buf * kRsCols + (lane & (kRsCols - 1))) * kRsPitch);
            auto ld4 = [&](float4 (&v)[4], int r) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = col[(r >> 2) + u];
            };
            auto add16 = [&](const float4 (&v)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { s = s + v[u].x; s = s + v[u].y; s = s + v[u].z; s = s + v[u].w; }
            };
            int r = 0;
            if (rows_here == kRsRows) {
                // a whole tile, straight-line: three register sets of 16 rows, read turn and turn about, TWO sets ahead of the adds.  The
                // scheduling fences keep the order: reads of set k + 2, then the adds of set k.
                float4 v[3][4];
                ld4(v[0], 0);
                ld4(v[1], 16);
#pragma unroll
                for (int k = 0; k < kRsRows / 16; ++k) {
                    if (k + 2 < kRsRows / 16) ld4(v[(k + 2) % 3], 16 * (k + 2));
                    __builtin_amdgcn_sched_barrier(0);
                    // ONE wait for the four reads of set k (LDS returns in order: at most the 8 / 4 / 0 reads of the younger sets stay out) instead
                    // of the compiler's one per read: a lone wave pays ~6 cycles for every instruction it issues, waits included
                    // (scripts/probes/dep_add_rate.hip: 5.75 cycles per dependent v_add_f32).  simm16: lgkmcnt in bits 11:8, vmcnt / expcnt at maximum.
                    if (k + 2 < kRsRows / 16) __builtin_amdgcn_s_waitcnt(0xC87F);
                    else if (k + 1 < kRsRows / 16) __builtin_amdgcn_s_waitcnt(0xC47F);
                    else __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);                       // (the adds are no memory operations: nothing else keeps them below the wait)
                    add16(v[k % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            // the last, partial tile: 16 rows at a time, then row by row
            {
                float4 w[4];
#pragma unroll 1
                for (; r + 16 <= rows_here; r += 16) { ld4(w, r); add16(w); }
            }
            const float* colf = reinterpret_cast<const float*>(col);
#pragma unroll 1
            for (; r < rows_here; ++r) s = s + colf[r];
        };
        __syncthreads();
        for (int64_t t = 0; t < ntiles; ++t) {
            if (adder) walk((int)(t & 1), rows_of(t));
            __syncthreads();
        }
        if (col_ok) j.run[c0 + lane] = s;
    } else {
        // ---- loaders (waves 1..3): thread lt owns 16 columns (`half`) of row lt % 96 of a tile.  (Rows past the end and columns past d are
        // never walked / written back: their addresses are clamped, their values do not matter.)
        const int lt = tid - 64;
        const int lrow = lt % kRsRows, half = lt / kRsRows;              // 192 threads = 96 rows x 2 halves (32 columns) / 192 rows (16 columns)
        const int cbase = (c0 + 16 * half < L.d) ? 16 * half : 0;       // (d is a multiple of 8, not necessarily of 32)
        const int q1 = (c0 + cbase + 8 < L.d) ? 8 : 0;
        auto issue = [&](uint4 (&r)[2], int64_t t) {
            const int64_t row = t * kRsRows + lrow;
            // (global, not flat, loads: the pointer comes out of a structure in memory and the compiler cannot tell -- a flat load counts
            //  on the LDS counter as well and forces full waits around the LDS traffic)
            typedef unsigned int rs_u32x4 __attribute__((ext_vector_type(4)));       // (HIP's uint4 is a class: no address-space-qualified copies)
            typedef __attribute__((address_space(1))) const rs_u32x4 g_u4;
            int64_t src = row < j.n ? row : j.n - 1;
            if constexpr (IDX) {                                         // (a dependent load per row; RING tiles in flight cover it)
                typedef __attribute__((address_space(1))) const int32_t g_i32;
                src = *(g_i32*)(uintptr_t)(j.idx + src);
            }
            const uint16_t* p = base + src * j.ld + c0 + cbase;
            // (plain loads: the four workgroups that share a row's 128-byte line meet in their XCD's L2 -- with the streaming hint `nt` on these
            //  loads the walk took 0.69 instead of 0.40 ms and the realistic loop 3 460 instead of 5 020 scores/s, r05y)
            const rs_u32x4 v0 = *(g_u4*)(uintptr_t)p, v1 = *(g_u4*)(uintptr_t)(p + q1);
            r[0] = make_uint4(v0.x, v0.y, v0.z, v0.w);
            r[1] = make_uint4(v1.x, v1.y, v1.z, v1.w);
        };
        auto dump = [&](const uint4 (&r)[2], int buf) {
            float* dst = rs_lds + ((size_t)buf * kRsCols + 16 * half) * kRsPitch + lrow;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t w[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rs_h2 h; __builtin_memcpy(&h, &w[i], 4);
                    dst[(size_t)(8 * q + 2 * i) * kRsPitch] = (float)h[0];
                    dst[(size_t)(8 * q + 2 * i + 1) * kRsPitch] = (float)h[1];
                }
            }
        };
        // RING tiles in flight in registers (tile tau waits in set tau % RING), plus the tile in the other LDS buffer.  32 columns: five
        // (480 rows; six spill; with four -- r05e -- the walk fell to 14 cycles per row: 384 rows are 1.5 us of adds, less than a load takes
        // beside the tile kernel); 16 columns: four (768 rows).
        constexpr int PERIOD = (RING & 1) ? 2 * RING : RING;
        uint4 rg[RING][2];
#pragma unroll
        for (int q = 0; q < RING; ++q)
            if (q < ntiles) issue(rg[q], q);
        dump(rg[0], 0);
        if (RING < ntiles) issue(rg[0], RING);
        __syncthreads();
        // phase t: the adding wave walks tile t; tile t + 1 goes from its register set into the other buffer, tile t + 1 + RING takes the set
        for (int64_t t = 0; t < ntiles; t += PERIOD) {
            bool last = false;
#pragma unroll
            for (int k = 0; k < PERIOD; ++k) {
                if (!last) {
                    const int64_t tt = t + k;
                    // (the scheduling fence keeps a set's new loads behind the last use of its old values: without it the compiler holds both)
                    if (tt + 1 < ntiles) { dump(rg[(k + 1) % RING], (k + 1) & 1); __builtin_amdgcn_sched_barrier(0); if (tt + 1 + RING < ntiles) issue(rg[(k + 1) % RING], tt + 1 + RING); }
                    __syncthreads();
                    last = tt + 1 >= ntiles;
                }
            }
            if (last) break;
        }
    }
}

// numpy's float32 running column sums PER FILE (utils.py:16: np.mean of every file; fad.py:377 per song): the rows of file f are
// [offsets[f], offsets[f + 1]); one thread walks one column (WIDE: the eight columns of a 16-byte piece, 16-bit frames on 16-byte
// aligned rows) down the file's rows in order, 32 rows of loads in flight; consecutive threads take consecutive columns of the same
// file, then the next file.  out: [n_files][d] float32.
template <typename TIn, bool WIDE>
__global__ __launch_bounds__(256) void segment_running_sums(const TIn* __restrict__ rows, int64_t ld, int d, const int64_t* __restrict__ offsets,
                                                            int64_t n_files, float* __restrict__ out) {
    constexpr int CW = WIDE ? 8 : 1;                                // columns per thread
    const int ncg = (d + CW - 1) / CW;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= n_files * ncg) return;
    const int64_t f = item / ncg;
    const int c0 = (int)(item - f * ncg) * CW;
    const int64_t r0 = offsets[f], r1 = offsets[f + 1];
    float s[CW];
#pragma unroll
    for (int q = 0; q < CW; ++q) s[q] = 0.f;
    if constexpr (WIDE) {
        auto add8 = [&](const uint4& w) {
            const uint32_t qd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (std::is_same<TIn, raw_f16>::value) {
                    rs_h2 h; __builtin_memcpy(&h, &qd[i], 4);
                    s[2 * i] = s[2 * i] + (float)h[0]; s[2 * i + 1] = s[2 * i + 1] + (float)h[1];
                } else {
                    s[2 * i] = s[2 * i] + __uint_as_float(qd[i] << 16); s[2 * i + 1] = s[2 * i + 1] + __uint_as_float(qd[i] & 0xffff0000u);
                }
            }
        };
        const uint16_t* base = reinterpret_cast<const uint16_t*>(rows) + c0;
        int64_t r = r0;
        // 32 rows = 512 bytes per thread in flight: the walk is latency-bound per THREAD, and a call of 2000 songs x 16 column groups is
        // 500 waves -- two per CU (eight rows in flight: 0.5 ms for 2000 x [2250 x 128], r05c)
        for (; r + 32 <= r1; r += 32) {
            uint4 v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (r + u) * ld);
#pragma unroll
            for (int u = 0; u < 32; ++u) add8(v[u]);
        }
        for (; r + 4 <= r1; r += 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (r + u) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) add8(v[u]);
        }
        for (; r < r1; ++r) add8(*reinterpret_cast<const uint4*>(base + r * ld));
    } else {
        int64_t r = r0;
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_as_float(rows + (r + u) * ld + c0);
#pragma unroll
            for (int u = 0; u < 8; ++u) s[0] = s[0] + v[u];
        }
        for (; r < r1; ++r) s[0] = s[0] + load_as_float(rows + r * ld + c0);
    }
#pragma unroll
    for (int q = 0; q < CW; ++q)
        if (c0 + q < d) out[f * d + c0 + q] = s[q];
}

// mu = sum/n ; cov = (M - sum sum^T / n) / (n - ddof)
// (`run`: the float32 running column sums above -- then mu = float32(float64(run) / n), numpy's quotient: fad_common.h)
__global__ __launch_bounds__(256) void moments_finalize_kernel(
    const double* __restrict__ acc_packed, int d, int ddof, double* __restrict__ mu,
    double* __restrict__ cov, const float* __restrict__ run = nullptr) {
    const double n = acc_packed[0];
    const double* sum = acc_packed + 1;
    const double* M = acc_packed + 1 + d;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < d && mu) mu[g] = run ? numpy_mean_of_f32_sum(run[g], n) : sum[g] / n;
    if (g >= (int64_t)d * d) return;
    const int a = (int)(g / d), b = (int)(g - (int64_t)a * d);
    cov[g] = (M[g] - (sum[a] * sum[b]) / n) / (n - (double)ddof);   // commutative: cov == cov^T bit for bit
}

// ------------------------------------------------------------------------------------------
// Per-song covariances on the float16 tile body (frechet.hip: the batched per-song chain, songs of at least D + 1 float16 frames).
// Round 2 formed them on the float64 MFMA (song_cov_mfma, frechet_songs.hip): 3.0 ms of a 10.5 ms call for 2000 songs of [2250 x 128],
// 1.0 of 8.5 ms for 32 songs of [1500 x 768] (profiles/r03j_*).  Here a song is what a split is to moments_tile_h16_tr, in its
// SHIFTED form: every column is shifted by c = float16(mean) (the exact mean is known: the statistics kernel ran), the rows enter
// the MFMAs as the error-free pair x - c = x' + e, the products are exact in float32 and what is summed is centred -- float32-sum
// accuracy relative to the variances (~1e-7), which moves tr sqrt(Sigma_b Sigma_s) by ~1e-8 of itself; tr Sigma_s and the mean term
// of the score stay the exact float64 sums of the statistics kernel.  Songs longer than 4096 frames are cut into S runs, each with
// its own float32 partial tile; song_cov_finish sums the runs in float64 and writes
//     Sigma = (S' - s' s'^T / n) / (n - 1),    S' = sum (x - c)(x - c)^T,  s' = sum (x - c)      (both triangles).
struct SongCovLaunch {
    const uint16_t* rows; int64_t ld; int d, nt, T, S;
    const int64_t* offsets; const int64_t* song_ids;       // slot -> song (nullptr: slot == song)
    const double* mean_exact;                              // [song][d]
    const double* var_exact;                               // [song][d] exact variances (float64 one-pass sums), or nullptr
    uint16_t* cvec;                                        // [slot * S + run][nt * H_BT]
    float* partials;                                       // [(slot * S + run) * T + tile][H_TS]
    double* colpart;                                       // [slot * S + run][nt * H_BT]
    double* cov_out;                                       // [slot][d * d]
};

__global__ __launch_bounds__(256) void song_cov_shift(SongCovLaunch L) {
    const int64_t slot = blockIdx.x, s = L.song_ids ? L.song_ids[slot] : slot;
    const int dpad = L.nt * H_BT;
    for (int col = threadIdx.x; col < dpad; col += 256) {
        uint16_t bits = 0;
        if (col < L.d) {
            const double m = L.mean_exact[s * L.d + col];
            const _Float16 ch = ((m == m) && fabs(m) < 65000.0) ? (_Float16)(float)m : (_Float16)0.0f;
            __builtin_memcpy(&bits, &ch, 2);
        }
        for (int run = 0; run < L.S; ++run) L.cvec[(slot * L.S + run) * dpad + col] = bits;
    }
}

template <bool FAST>
__global__ __launch_bounds__(256) void song_cov_tile(SongCovLaunch L) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];
    const int tile = blockIdx.x, run = blockIdx.y;
    const int64_t slot = blockIdx.z, s = L.song_ids ? L.song_ids[slot] : slot;
    const int64_t r0 = L.offsets[s], r1 = L.offsets[s + 1];
    const int64_t per = (((r1 - r0 + L.S - 1) / L.S) + H_KB - 1) / H_KB * H_KB;     // whole stages: only a song's last run ends ragged
    const int64_t k_begin = (r0 + run * per < r1) ? r0 + run * per : r1;
    const int64_t k_end = (k_begin + per < r1) ? k_begin + per : r1;
    const int split = (int)(slot * L.S + run);
    int ta, tb; tile_coords(tile, L.nt, ta, tb);
    if (ta == tb)
        tile_h16_tr_body<FAD_F16, FAST ? H_NST : 2 * H_NST, true, FAST, true>(L.rows, k_begin, k_end, L.ld, L.d, L.nt, L.T, split, tile, ta * H_BT,
                                                                              tb * H_BT, L.partials, L.colpart, smem_dyn, nullptr, nullptr, 0, 0, L.cvec);
    else if constexpr (FAST)
        tile_h16_tr_body<FAD_F16, H_NST, false, FAST, true>(L.rows, k_begin, k_end, L.ld, L.d, L.nt, L.T, split, tile, ta * H_BT, tb * H_BT,
                                                            L.partials, L.colpart, smem_dyn, nullptr, nullptr, 0, 0, L.cvec);
}

// grid (T * 16, songs): a thread owns one float4 of a partial tile = rows ga .. ga + 3 of column gb (the fragment-major layout of
// tile_h16_tr_body); on a diagonal tile only the blocks on and above the diagonal hold sums ("mirror": the block (rows 0..63,
// columns 64..127) is the sum of two half-sums), the rest follows by symmetry
__global__ __launch_bounds__(256) void song_cov_finish(SongCovLaunch L) {
    const int64_t slot = blockIdx.y, s = L.song_ids ? L.song_ids[slot] : slot;
    const int tile = blockIdx.x >> 4, e = (blockIdx.x & 15) * 256 + threadIdx.x;
    int ta, tb; tile_coords(tile, L.nt, ta, tb);
    const int fa = e >> 10, fb = (e >> 8) & 3, q = (e >> 6) & 3, el = e & 63;
    const bool diag = ta == tb;
    if (diag && fa > fb) return;
    const int nsrc = (diag && fa < 2 && fb >= 2) ? 2 : 1;
    const int dpad = L.nt * H_BT;
    const int ga = ta * H_BT + 32 * fa + 8 * q + 4 * (el >> 5), gb = tb * H_BT + 32 * fb + (el & 31);
    double s4[4] = {0.0, 0.0, 0.0, 0.0}, sa[4] = {0.0, 0.0, 0.0, 0.0}, sb = 0.0;
    for (int run = 0; run < L.S; ++run) {
        const int64_t sp = slot * L.S + run;
        const float* p = L.partials + (sp * L.T + tile) * H_TS + e * 4;
        for (int h = 0; h < nsrc; ++h) {
            const float4 v = *reinterpret_cast<const float4*>(p + h * (6 * 256 * 4));
            s4[0] += (double)v.x; s4[1] += (double)v.y; s4[2] += (double)v.z; s4[3] += (double)v.w;
        }
        const double* cp = L.colpart + sp * dpad;
        sb += cp[gb];
#pragma unroll
        for (int k = 0; k < 4; ++k) sa[k] += cp[ga + k];
    }
    const double n = (double)(L.offsets[s + 1] - L.offsets[s]);
    const double inv_n = 1.0 / n, inv = 1.0 / (n - 1.0);
    double* out = L.cov_out + slot * (int64_t)L.d * L.d;
    if (gb >= L.d) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = ga + k;
        if (i >= L.d) continue;
        double v = (s4[k] - (sa[k] * sb) * inv_n) * inv;
        // the diagonal from the statistics kernel's float64 sums: a variance is a sum of POSITIVE terms, the float32 partial sum
        // drifts by ~4e-7 of it (140 accumulations for 2250 frames), and with Sigma_s close to Sigma_b the score moves by half
        // the sum of those drifts (d tr sqrt(Sigma_b Sigma_s) = 1/2 tr dSigma_s there) -- 1.1e-5 of a small score, measured; the
        // off-diagonal sums hover around zero and carry ~1e-8
        if (i == gb && L.var_exact) v = L.var_exact[s * L.d + i];
        // in a diagonal 32 x 32 block both (i, gb) and (gb, i) have a thread, and their float32 partials were summed in a
        // different order: only the upper one writes the pair, so that Sigma_s is symmetric bit for bit and the same on every run
        if (diag && fa == fb && i > gb) continue;
        out[(int64_t)i * L.d + gb] = v;
        out[(int64_t)gb * L.d + i] = v;
    }
}

}  // namespace fad
