// E^T E on 256-column slabs (gfx950): the moments tile kernel for D >= 512 (round 4).  Included by moments.hip after
// moments_kernels.h (whose small device helpers it uses); the block bookkeeping is in tile256_roles.h.
//
// Why: the 128 x 128 kernel (moments_tile_h16_tr) moves 1.5 KB through LDS per MFMA -- 16 KB of LDS-DMA per 32 MFMAs and 1 KB
// of transpose reads per MFMA -- and two of its workgroups per CU sit at the CU's 64 B/clk L1 path; its matrix pipe is busy a
// third of the time.  Here a workgroup of 8 waves owns a CU, streams TWO 256-column slabs per 32 rows (32 KiB per stage, ring
// of four) and issues 136 MFMAs on them (128 for a plain tile): 0.24 KB of LDS-DMA and 0.65 KB of reads per MFMA.
//
//   * slabs: global_load_lds_dwordx4 with an SGPR base (1 KiB per wave instruction, four per wave and stage) into four
//     [32 rows][128 columns] sub-slabs with the XOR swizzle of the 128-kernel; counted s_waitcnt + raw s_barrier;
//   * fragments: ds_read_b64_tr_b16, two per fragment and 16-row k-step, ALL reads of a stage before its MFMAs;
//   * every wave has a compile-time ROLE (tile256_roles.h): the loop it runs updates a fixed set of accumulators
//     (9 blocks = 144 registers, or 8) -- no run-time choice inside the loop (cf. the v_accvgpr_mov story of round 2);
//   * column sums (and sum x^2 for the shift guard) ride on the two waves per superblock that read fragments 0..3 / 4..7
//     anyway (v_dot2c_f32_f16 against (1, 1) resp. against itself);
//   * epilogue: a wave stores its blocks fragment-major (1 KiB per store instruction) at slot 9 wave + b of the item.
//
// SHIFT = the shift guard's second pass (see tile_h16_tr_body): rows enter as the error-free pair x - c = x' + e.
#pragma once
#include "moments_kernels.h"
#include "tile256_roles.h"

namespace fad {

constexpr int T2_KB = 32;                   // rows per slab and stage
#ifndef T2_NST_VALUE
#define T2_NST_VALUE 4
#endif
constexpr int T2_NST = T2_NST_VALUE;        // ring depth (scripts/probes/tile256_bench.hip builds variants)
constexpr int T2_SUB = T2_KB * 16;          // uint4 per [32][128] sub-slab
constexpr int T2_STAGE = 4 * T2_SUB;        // A0 A1 B0 B1: 32 KiB
constexpr size_t kT256Lds = (size_t)T2_NST * T2_STAGE * sizeof(uint4);      // 128 KiB (plan 0)
constexpr size_t kT256LdsCombined = 144 * 1024;                              // plan 1: three 48 KiB stages of an XZ item = the 36 blocks a ZC quartet hands over

struct T256Set {
    const void* E; int64_t n, ld, rows_per_split;
    int S, item0;                           // row-splits; first work item (item = item0 + split * NT + type index)
    float* partials;                        // [S][NT][t256::ITEM_STRIDE]
    double* colpart;                        // [S][2][dpad]   (second row: the second wave quartet of a Z item)
    int* flag;                              // shift guard (or nullptr)
    uint16_t* cvec;                         // [S][dpad] float16 shifts for the second pass (or nullptr)
};
struct T256Launch {
    T256Set set[kMaxSets];
    int nsets, d, nsb, NT, total;
    uint8_t type[t256::MAX_TYPES], sa[t256::MAX_TYPES], sb[t256::MAX_TYPES];
    int plan;                               // 0: P / Q / X / Z items, 1: ZC / XZ (tile256_roles.h)
};

// Workgroup barrier that the instruction scheduler may not move anything across (MFMAs have no memory effects: without the
// fences hipcc slides them over a bare s_barrier and the load / MFMA halves of the ping-pong loop dissolve).
__device__ __forceinline__ void t2_phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ uint4 t2_frag(uint32_t lds_byte) {
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)lds_byte);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(lds_byte + 1024));      // rows + 4: same swizzle
    uint4 f;
    __builtin_memcpy(&f.x, &lo, 8);
    __builtin_memcpy(&f.z, &hi, 8);
    return f;
}

// x - c = x' + e (packed TwoSum, float16); rows past the end of the run (loaded as zeros) stay zero
__device__ __forceinline__ void t2_split2(const uint4& f, uint32_t c2, int64_t rows_left, uint4& xs, uint4& es) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    uint32_t w[4] = {f.x, f.y, f.z, f.w}, x[4], e[4];
    h2 c; __builtin_memcpy(&c, &c2, 4);
    const h2 b = -c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h2 a; __builtin_memcpy(&a, &w[q], 4);
        const h2 sm = a + b;
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
}

template <int KIND> __device__ __forceinline__ float t2_sumsq8(const uint4& v, float s) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (KIND == FAD_F16) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 a; __builtin_memcpy(&a, &w[q], 4);
            s = __builtin_amdgcn_fdot2(a, a, s, false);
        } else {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            b2 a; __builtin_memcpy(&a, &w[q], 4);
            s = __builtin_amdgcn_fdot2_f32_bf16(a, a, s, false);
        }
    }
    return s;
}

// One wave's share of a work item.  Everything role-dependent is a compile-time constant.  XZT = the XZ items' stage layout (XR
// role only): per quartet three 8 KiB sub-slabs -- the 128-column A half, then the 256 B-side columns -- of ITS 32 rows; six
// sub-slabs = 48 KiB per 64-row stage, ring of three, six LDS-DMA pieces per wave and stage.  Otherwise: four sub-slabs (slab A,
// slab B) = 32 KiB per stage, ring of T2_NST, four pieces per wave.
template <int KIND, int ROLE, bool SHIFT, bool XZT>
__device__ __forceinline__ void tile256_wave(
    const T256Launch& L, const T256Set& s, int split, int ti, int type, int sa, int sb, const t256::WaveJob job, uint4* smem) {
    using RD = t256::RoleDef<ROLE>;
    constexpr int NF = RD::NF, NB = RD::NB;
    static_assert(!SHIFT || KIND == FAD_F16, "the shifted pass is written for float16 rows");
    static_assert(!XZT || ROLE == t256::XR, "the XZ layout carries XR waves only");
    constexpr int NSTG = XZT ? 3 : T2_NST;              // ring depth
    constexpr int NSUB = XZT ? 6 : 4;                   // sub-slabs per stage
    constexpr int STG = NSUB * T2_SUB;                  // uint4 per stage
    constexpr int LPS = XZT ? 6 : 4;                    // LDS-DMA pieces per wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quartet = wave >> 2;
    const int li = lane & 31, kg = lane >> 5;
    const uint16_t* __restrict__ E = static_cast<const uint16_t*>(s.E);
    const int64_t ld = s.ld;
    const int d = L.d;
    // Z, ZC, XZ: the quartets take alternate 32-row stages (a "stage" of the loop below covers 64 rows)
    const bool two_rows = type >= t256::TYPE_Z;
    const bool zlike = type == t256::TYPE_Z || type == t256::TYPE_ZC;
    const int xh = XZT ? (type - t256::TYPE_XZ0) : 0;                            // XZ: which 128-row half of the tile
    const int64_t k_begin = (int64_t)split * s.rows_per_split;
    const int64_t k_end = (k_begin + s.rows_per_split < s.n) ? k_begin + s.rows_per_split : s.n;
    const int rows_per_stage = two_rows ? 2 * T2_KB : T2_KB;
    const int nkb = (int)((k_end - k_begin + rows_per_stage - 1) / rows_per_stage);
    const int colA = t256::SB * sa, colB = t256::SB * (zlike ? sa : sb);        // first column behind slab A / slab B
    const int my_rowoff = two_rows ? T2_KB * quartet : 0;                       // rows of a stage this wave's fragments come from

    // ---- loads.  Piece p of a wave: (sub-slab, 4-row group) -> LDS uint4 offset inside the stage, source column and row offset
    const int lrow = lane >> 4, lchunk = (lane & 15) ^ (lrow << 2);             // swizzle on the SOURCE side (LDS-DMA writes lane-linear)
    const uint32_t voff = (uint32_t)(((int64_t)lrow * ld + lchunk * 8) * 2);
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(&g_zero16);
    auto piece_geom = [&](int p, int& lds_u4, int& col0, int& rowoff) {
        if constexpr (XZT) {
            const int idx = LPS * wave + p;                                     // 0..47: sub-slab idx / 8, row group idx % 8
            const int ss = idx >> 3, rg = idx & 7, q = ss / 3, kind = ss - 3 * q;
            lds_u4 = ss * T2_SUB + 64 * rg;
            col0 = kind == 0 ? colA + 128 * xh : colB + 128 * (kind - 1);
            rowoff = T2_KB * q + 4 * rg;
        } else {
            const int sub = wave >> 1;                                          // 0, 1: slab A; 2, 3: slab B
            lds_u4 = sub * T2_SUB + 256 * (wave & 1) + 64 * p;
            col0 = (sub < 2 ? colA : colB) + 128 * (sub & 1);
            rowoff = 16 * (wave & 1) + 4 * p + ((zlike && sub >= 2) ? T2_KB : 0);
        }
    };
    const bool cols_full = (colA + t256::SB <= d) && (colB + t256::SB <= d) && ld < ((int64_t)1 << 26);
    auto piece_fast = [&](int kb, int p) {     // SGPR base + 32-bit lane offset
#ifdef T2_ABL_NODMA                          // ablation: the ring is filled once and never refilled
        if (kb >= NSTG - 1) return;
#endif
        int lds_u4, col0, rowoff;
        piece_geom(p, lds_u4, col0, rowoff);
        const uint64_t sbq = (uint64_t)(E + (k_begin + (int64_t)kb * rows_per_stage + rowoff) * ld + col0);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sbq);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sbq >> 32));
        const uint64_t ub = ((uint64_t)hi << 32) | lo;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(smem_lds + (uint32_t)(((kb % NSTG) * STG + lds_u4) * 16));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
    };
    auto issue_fast = [&](int kb) {
#pragma unroll
        for (int p = 0; p < LPS; ++p) piece_fast(kb, p);
    };
    auto issue_slow = [&](int kb) {         // edge stages: per-lane 64-bit addresses, rows / columns out of range read the zero block
#pragma unroll
        for (int p = 0; p < LPS; ++p) {
            int lds_u4, col0, rowoff;
            piece_geom(p, lds_u4, col0, rowoff);
            const int64_t r = k_begin + (int64_t)kb * rows_per_stage + rowoff + lrow;
            const bool col_ok = (col0 + lchunk * 8) < d;                        // d % 8 == 0: a chunk is in or out as a whole
            const uint16_t* src = (r < k_end && col_ok) ? E + r * ld + col0 + lchunk * 8 : zsrc;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(smem_lds + (uint32_t)(((kb % NSTG) * STG + lds_u4) * 16));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
        }
    };
    const int nfast = cols_full ? (int)((k_end - k_begin) / rows_per_stage) : 0;      // stages [0, nfast) are whole
    auto issue = [&](int kb) { if (kb < nfast) issue_fast(kb); else issue_slow(kb); };

    // ---- fragments: byte offset inside a stage of F[i] for k-step 0 (k-step 1: + 4096)
    const int t16 = lane & 15, grp = lane >> 4;
    const int tr_row = 8 * (grp >> 1) + (t16 >> 2), tr_col = 16 * (grp & 1) + 4 * (t16 & 3);
    uint32_t foff[NF];
    int fcol[NF];                            // global column of this lane's element of F[i] (column sums, shifts)
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        int subslab, f;                      // which 8 KiB sub-slab of the stage, and the fragment's index in its superblock
        bool from_a;
        if constexpr (XZT) {
            from_a = i < 4;
            f = from_a ? 4 * xh + i : job.b0 + (i - 4);
            subslab = 3 * quartet + (from_a ? 0 : 1 + (f >> 2));
        } else if constexpr (ROLE == t256::XR) {
            from_a = i < 4;
            f = from_a ? job.a0 + i : job.b0 + (i - 4);
            subslab = (from_a ? 0 : 2) + (f >> 2);
        } else {
            from_a = job.slab == 0;
            f = RD::frag[i];
            subslab = 2 * job.slab + (f >> 2);
        }
        const int col = 32 * (f & 3) + tr_col;
        foff[i] = (uint32_t)(subslab * (T2_SUB * 16) + tr_row * 256 + (((col >> 3) ^ ((tr_row & 3) << 2)) << 4) + ((col >> 2) & 1) * 8);
        fcol[i] = (from_a ? colA : colB) + 32 * f + li;
    }
    uint32_t cs[NF];                         // SHIFT: this lane's shift per fragment, packed twice
#pragma unroll
    for (int i = 0; i < NF; ++i) cs[i] = 0u;
    if constexpr (SHIFT) {
        const uint16_t* cv = s.cvec + (int64_t)split * (L.nsb * t256::SB);
#pragma unroll
        for (int i = 0; i < NF; ++i) { const uint32_t h = cv[fcol[i]]; cs[i] = h | (h << 16); }
    }
    auto rows_left_at = [&](int kb, int ks) -> int64_t {
        return k_end - (k_begin + (int64_t)kb * rows_per_stage + my_rowoff + ks * 16 + 8 * kg);
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
    // column sums (and sum x^2 for the guard): every triangle-family wave takes TWO of its fragments -- superblock fragments
    // 0,1 (TRI_LO), 2,3 (RECT_D), 4,5 (RECT_C), 6,7 (TRI_HI) -- so that no wave carries more than 2 x 2 x 9 VALU ops per stage
    constexpr bool CSUM = ROLE != t256::XR;
    constexpr int CS0 = (ROLE == t256::TRI_HI || ROLE == t256::RECT_C) ? 2 : 0;          // F[CS0], F[CS0 + 1]
    double csum[2] = {0.0, 0.0};
    float csq[2] = {0.f, 0.f};

    uint4 F0[NF], F1[NF];                    // the fragments of the stage in hand: k-step 0 / 1
    // LOAD half of a stage: every transpose read of the stage
    auto load_frags = [&](int kb) {
#ifdef T2_ABL_NOREAD                         // ablation (scripts/probes/tile256_bench.hip): no transpose reads
        if (kb > 0) return;
#endif
        const uint32_t base = smem_lds + (uint32_t)((kb % NSTG) * STG * 16);
#pragma unroll
        for (int i = 0; i < NF; ++i) F0[i] = t2_frag(base + foff[i]);
#pragma unroll
        for (int i = 0; i < NF; ++i) F1[i] = t2_frag(base + foff[i] + 4096);
    };
    // MFMA half of a stage; `refill`: this wave's LDS-DMA pieces of stage kb + NSTG - 1 go BETWEEN the MFMAs (a piece holds the wave
    // ~60 cycles among bare MFMAs, 100-185 among the transpose reads of the load half -- MI355X_MICROARCH.md)
    auto mma = [&](int kb, auto refill) {
        if constexpr (SHIFT) {              // x - c = x' + e, one k-step at a time (register budget)
            if (decltype(refill)::value) issue_fast(kb + NSTG - 1);
            const bool full = kb < nfast;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 X[NF], R[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) t2_split2(ks ? F1[i] : F0[i], cs[i], full ? 8 : rows_left_at(kb, ks), X[i], R[i]);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[b] = mfma_h16<KIND>(X[RD::fa[b]], X[RD::fb[b]], acc[b]);
                    acc[b] = mfma_h16<KIND>(X[RD::fa[b]], R[RD::fb[b]], acc[b]);
                    acc[b] = mfma_h16<KIND>(R[RD::fa[b]], X[RD::fb[b]], acc[b]);
                }
                if constexpr (CSUM) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) csum[i] += (double)sum8<KIND>(X[CS0 + i]) + (double)sum8<KIND>(R[CS0 + i]);
                }
            }
            return;
        }
#ifdef T2_ABL_NOMMA                          // ablation: no MFMAs (the fragments stay live through a cheap VALU use)
#pragma unroll
        for (int i = 0; i < NF; ++i) acc[0][i] += __uint_as_float((F0[i].x ^ F0[i].y ^ F0[i].z ^ F0[i].w ^ F1[i].x ^ F1[i].y ^ F1[i].z ^ F1[i].w) & 0x007fffffu);
        if (decltype(refill)::value) issue_fast(kb + NSTG - 1);
        return;
#endif
        constexpr int PH = LPS / 2;          // pieces per k-step: behind MFMA 1, 3 (, 5)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[b] = ks ? mfma_h16<KIND>(F1[RD::fa[b]], F1[RD::fb[b]], acc[b]) : mfma_h16<KIND>(F0[RD::fa[b]], F0[RD::fb[b]], acc[b]);
                if (decltype(refill)::value && (b & 1) && (b >> 1) < PH) {
                    __builtin_amdgcn_sched_barrier(0); piece_fast(kb + NSTG - 1, PH * ks + (b >> 1)); __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#ifndef T2_OPT_NOCOLSUM
        if constexpr (CSUM) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                csum[i] += (double)sum8<KIND>(F0[CS0 + i]) + (double)sum8<KIND>(F1[CS0 + i]);
                csq[i] = t2_sumsq8<KIND>(F1[CS0 + i], t2_sumsq8<KIND>(F0[CS0 + i], csq[i]));
            }
        }
#endif
    };
    // this wave's pieces of stage `kb + 1` have landed once at most (stages issued beyond it) x LPS of its loads are outstanding
    auto wait_next = [&](int kb) {
        if (kb + 1 >= nkb) return;
        const int last = (kb + NSTG - 1 < nkb - 1) ? kb + NSTG - 1 : nkb - 1;         // youngest stage issued so far
        wait_vmcnt_upto<LPS>(last - (kb + 1));
    };
    static_assert(NSTG <= 8, "wait_vmcnt_upto counts at most seven stages");

    // ---- PING-PONG.  The two waves of a SIMD are wave w and wave w + 4: the quartets run HALF A STAGE APART, so that while one
    // wave of every SIMD is in its load half (transpose reads) the other one keeps the matrix pipe busy with its 16-18 MFMAs and
    // issues its LDS-DMA pieces between them.  (With both in the same phase -- the first version -- a stage took ~2600 cycles: DMA
    // issue, reads and 1088 cycles of MFMAs one after the other.)  Half-steps h, a barrier b_h after each:
    //     quartet 0:  L(0) b0 M(0) b1 L(1) b2 M(1) b3 ...          quartet 1:  --  b0 L(0) b1 M(0) b2 L(1) b3 ...
    //   * a wave waits for its OWN pieces of stage k + 1 at the end of L(k): both quartets have done so before b_{2k+1}, the
    //     barrier in front of the first L(k + 1);
    //   * the slot of stage k - 1 is refilled in M(k): its last readers (quartet 1 in L(k - 1), reads drained by lgkmcnt(0))
    //     are behind b_{2k-1}.
    for (int s0 = 0; s0 < NSTG - 1 && s0 < nkb; ++s0) issue(s0);
    wait_vmcnt_upto<LPS>(((nkb < NSTG - 1) ? nkb : NSTG - 1) - 1);                    // stages issued beyond stage 0
    t2_phase_barrier();
    if (quartet == 1) t2_phase_barrier();                                               // b0
    const int hot = nfast - (NSTG - 1) > 0 ? nfast - (NSTG - 1) : 0;                  // stages whose refill is a whole stage
    int kb = 0;
    for (; kb < hot; ++kb) {                 // no branches: the refill (SGPR-base pieces) rides in the MFMA half
        load_frags(kb);
        wait_vmcnt<LPS * (NSTG - 3)>();      // in flight at this point: stages kb + 1 .. kb + NSTG - 2; kb + 1 must have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t2_phase_barrier();
        mma(kb, std::true_type{});
        t2_phase_barrier();
    }
    for (; kb < nkb; ++kb) {
        load_frags(kb);
        if (kb + NSTG - 1 < nkb) issue(kb + NSTG - 1);
        wait_next(kb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t2_phase_barrier();
        mma(kb, std::false_type{});
        if (!(quartet == 1 && kb == nkb - 1)) t2_phase_barrier();
    }

    // ---- combined items (ZC, XZ): quartet 1 hands its blocks to quartet 0 through LDS (the ring is idle: every piece issued
    // has landed and been read), which adds them to its own -- 36 / 32 partial blocks per workgroup instead of 72 / 64
    const bool combined = t256::combined_type(type);
    if (combined) {
        float4* box = reinterpret_cast<float4*>(smem) + (size_t)(wave & 3) * (9 * 256);
        if (quartet == 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    box[(b * 4 + q) * 64 + lane] = make_float4(acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]);
        }
        __syncthreads();
        if (quartet == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = box[(b * 4 + q) * 64 + lane];
                    acc[b][4 * q] += v.x; acc[b][4 * q + 1] += v.y; acc[b][4 * q + 2] += v.z; acc[b][4 * q + 3] += v.w;
                }
        }
    }

    // ---- epilogue: blocks, fragment major -- float4 (q, lane) of block b = registers 4q..4q+3 = rows 8q + 4 (lane >> 5) + 0..3
    // of column lane & 31
    if (!combined || quartet == 0) {
        float4* out = reinterpret_cast<float4*>(s.partials + ((int64_t)split * L.NT + ti) * t256::ITEM_STRIDE) + (size_t)(9 * wave) * 256;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(b * 4 + q) * 64 + lane] = make_float4(acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]);
    }

    if constexpr (CSUM) {
        // column sums of this wave's two fragments over the rows it saw; the second quartet of a Z / ZC item writes the second row
        const int dpad = L.nsb * t256::SB;
        const int half = (zlike && job.slab) ? 1 : 0;
        double* cp = s.colpart + ((int64_t)split * 2 + half) * dpad;
        int64_t my_rows = k_end - k_begin;
        if (zlike) {    // rows of the 64-row stages that fall into this quartet's half
            const int64_t whole = my_rows / 64, rem = my_rows - whole * 64;
            my_rows = whole * 32 + (half ? (rem > 32 ? rem - 32 : 0) : (rem < 32 ? rem : 32));
        }
        bool hit = false;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = fcol[CS0 + i];
            double sx = csum[i];
            sx += __shfl_xor(sx, 32);
            if (kg == 0) cp[col] = sx;
            if constexpr (!SHIFT) {
                if (s.flag) {
                    double s2 = (double)csq[i];
                    s2 += __shfl_xor(s2, 32);
                    const double nr = (double)my_rows;
                    const double mean = sx / nr, var = s2 / nr - mean * mean;
                    const bool col_in = col < d;
                    // the same rule as the 128-kernel; sum x^2 is a float32 sum here (relative error ~1e-6: immaterial against 64 x)
                    hit = hit || (col_in && nr > 0.0 && !(mean * mean <= 64.0 * var) && !(sx == 0.0 && s2 == 0.0));
                    if (s.cvec && kg == 0 && half == 0) {
                        const bool worth = col_in && nr > 0.0 && (mean * mean > var) && (mean == mean) && !isinf(mean) && fabs(mean) < 65000.0;
                        const _Float16 ch = worth ? (_Float16)(float)mean : (_Float16)0.0f;
                        uint16_t bits; __builtin_memcpy(&bits, &ch, 2);
                        s.cvec[(int64_t)split * dpad + col] = bits;
                    }
                }
            }
        }
        if constexpr (!SHIFT) {
            if (s.flag && __any(hit) && lane == 0) atomicOr(s.flag, 1);
        }
    }
}

// Both passes are held to 224 registers (amdgpu_num_vgpr counts HALF of the unified file: 112 -> 224, found on a toy kernel).  The
// first pass needs 222 anyway; the SHIFT pass -- the gated second pass of the shift guard -- took 256 (+ 52 B/lane of scratch), and a
// workgroup of 256-register waves fills its SIMDs: even the launch that only reads the gate and exits could not be PLACED on a CU while a
// wave of the running-sum walk (moments_kernels.h: moments_running_colsum_h16, 64 registers, on a stream of its own) sat there -- r05d: the
// caller's stream stood still for 170-300 us per update.  The pass itself runs for heavily shifted frames only; there it now spills more.
template <int KIND, bool SHIFT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(112))) void moments_tile256(T256Launch L) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];
    const int w = xcd_contiguous(blockIdx.x, L.total);
    int si = 0;
#pragma unroll
    for (int i = 1; i < kMaxSets; ++i)
        if (i < L.nsets && w >= L.set[i].item0) si = i;
    const T256Set& s = L.set[si];
    if constexpr (SHIFT) {
        if (!s.flag || *s.flag == 0) return;
    }
    const int local = w - s.item0;
    const int split = local / L.NT, ti = local - split * L.NT;
    const int type = L.type[ti], sa = L.sa[ti], sb = L.sb[ti];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const t256::WaveJob job = t256::wave_job(type, wave);
    if (type == t256::TYPE_XZ0 || type == t256::TYPE_XZ1) {      // (workgroup-uniform)
        tile256_wave<KIND, t256::XR, SHIFT, true>(L, s, split, ti, type, sa, sb, job, smem_dyn);
        return;
    }
    switch (job.role) {              // wave-uniform: every wave runs ONE of these loops, all with the same stage count and barriers
        case t256::TRI_LO: tile256_wave<KIND, t256::TRI_LO, SHIFT, false>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        case t256::TRI_HI: tile256_wave<KIND, t256::TRI_HI, SHIFT, false>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        case t256::RECT_C: tile256_wave<KIND, t256::RECT_C, SHIFT, false>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        case t256::RECT_D: tile256_wave<KIND, t256::RECT_D, SHIFT, false>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        default: tile256_wave<KIND, t256::XR, SHIFT, false>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
    }
}

// ------------------------------------------------------------------------------------------
// partial blocks -> packed float64 accumulator.  One thread = one float4 (4 adjacent ROWS of one column) of one 32 x 32
// output block, for every SL-th row-split; the SL partial sums meet in LDS in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
struct R256Job {
    const float* partials; const double* colpart; const uint16_t* cvec;
    double* acc; double n_add;
    const int* gate; int* clear_flag;
    int S, overwrite;
    int64_t rows_per_split, n_rows;
};
struct R256Launch {
    R256Job job[kMaxSets];
    const t256::BlockSrc* table;             // device: n_blocks(8 nsb) entries
    int d, nsb, NT, nblk, sl;                // sl = split lanes per output group (1, 4 or 16)
    uint32_t two_mask;                       // bit a: superblock a's column sums have a second row (its triangle came from a Z / ZC item)
};

__global__ __launch_bounds__(256) void moments_reduce256(R256Launch R) {
    __shared__ double red[256 * 4];
    const R256Job& j = R.job[blockIdx.y];
    if (j.clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *j.clear_flag = 0;
    const bool unshift = j.gate && *j.gate != 0 && j.cvec;
    const int SL = R.sl, G = 256 / SL;
    const int dpad = R.nsb * t256::SB;
    const int tile_blocks = (R.nblk * 256 + G - 1) / G;
    const int block = blockIdx.x;
    const int S = j.S;
    auto rows_of = [&](int sp) -> double {
        const int64_t left = j.n_rows - (int64_t)sp * j.rows_per_split;
        return (double)(left < j.rows_per_split ? left : j.rows_per_split);
    };
    if (block >= tile_blocks) {              // trailing blocks: column sums and the row count
        // 64 columns x 4 split lanes per block, eight independent loads per thread and round (a first version walked the S
        // splits with one dependent load after the other in 2 blocks: 20 of the kernel's 28 us)
        const int a = (block - tile_blocks) * 64 + (threadIdx.x & 63), l = threadIdx.x >> 6;
        if (block == tile_blocks && threadIdx.x == 0) j.acc[0] = j.overwrite ? j.n_add : j.acc[0] + j.n_add;
        double t = 0.0;
        if (a < R.d) {
            const bool two = (R.two_mask >> (a / t256::SB)) & 1u;
            const double* cp = j.colpart + a;
            int sp = l;
            for (; sp + 28 < S; sp += 32) {
                double v[8], w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = cp[((int64_t)(sp + 4 * u) * 2) * dpad]; w[u] = two ? cp[((int64_t)(sp + 4 * u) * 2 + 1) * dpad] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u] + w[u];
            }
            for (; sp < S; sp += 4) t += cp[((int64_t)sp * 2) * dpad] + (two ? cp[((int64_t)sp * 2 + 1) * dpad] : 0.0);
            if (unshift)
                for (int q = l; q < S; q += 4) t += rows_of(q) * f16_bits_to_f64(j.cvec[(int64_t)q * dpad + a]);
        }
        red[threadIdx.x] = t;
        __syncthreads();
        if (l == 0 && a < R.d) {
            const double tot = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
            j.acc[1 + a] = j.overwrite ? tot : j.acc[1 + a] + tot;
        }
        return;
    }
    const int sl = threadIdx.x / G, gl = threadIdx.x % G;
    const int64_t g = (int64_t)block * G + gl;
    const bool live = g < (int64_t)R.nblk * 256;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int bi = 0, bj = 0, e = 0;
    if (live) {
        const int ob = (int)(g >> 8);
        e = (int)(g & 255);
        // block index -> (bi, bj) of the upper triangle, row major
        const int nb = t256::NFR * R.nsb;
        int rem = ob;
        while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
        bj = bi + rem;
        const t256::BlockSrc src = R.table[ob];
        const int64_t stride = (int64_t)R.NT * t256::ITEM_STRIDE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (src.src[h] < 0) continue;
            const int ti = src.src[h] / t256::SLOTS, slot = src.src[h] - ti * t256::SLOTS;
            const float* p = j.partials + (int64_t)ti * t256::ITEM_STRIDE + (int64_t)slot * t256::BLK + e * 4;
            int sp = sl;
#ifdef R256_ABL_NOLOAD                       // ablation: no partial tiles read
            sp = S;
#endif
            for (; sp + 7 * SL < S; sp += 8 * SL) {        // eight loads in flight per thread: the reduce is a latency chain otherwise
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (sp + u * SL) * stride);
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    s[0] += (double)v[u].x + (double)v[u + 1].x; s[1] += (double)v[u].y + (double)v[u + 1].y;
                    s[2] += (double)v[u].z + (double)v[u + 1].z; s[3] += (double)v[u].w + (double)v[u + 1].w;
                }
            }
            for (; sp < S; sp += SL) {
                const float4 v = *reinterpret_cast<const float4*>(p + sp * stride);
                s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            }
        }
        if (unshift) {                        // + c_a s'_b + s'_a c_b + n c_a c_b, split by split (see reduce_body)
            const int ga = 32 * bi + 8 * (e >> 6) + 4 * ((e & 63) >> 5), gb = 32 * bj + (e & 31);
            const bool two_a = (R.two_mask >> (ga / t256::SB)) & 1u, two_b = (R.two_mask >> (gb / t256::SB)) & 1u;
            for (int sp = sl; sp < S; sp += SL) {
                const double nq = rows_of(sp);
                const uint16_t* cv = j.cvec + (int64_t)sp * dpad;
                const double* sv = j.colpart + ((int64_t)sp * 2) * dpad;
                const double cb_ = f16_bits_to_f64(cv[gb]), sb_ = sv[gb] + (two_b ? sv[dpad + gb] : 0.0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double ca_ = f16_bits_to_f64(cv[ga + q]), sa_ = sv[ga + q] + (two_a ? sv[dpad + ga + q] : 0.0);
                    s[q] += ca_ * sb_ + sa_ * cb_ + nq * ca_ * cb_;
                }
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
#ifdef R256_ABL_NOWRITE                      // ablation: one store per thread instead of the block and its mirror image
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) j.acc[1] = 0.0;
    return;
#endif
    double* M = j.acc + 1 + R.d;
    const int d = R.d;
    const int a0 = 32 * bi + 8 * (e >> 6) + 4 * ((e & 63) >> 5), b = 32 * bj + (e & 31);
    if (b >= d) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int a = a0 + q;
        if (a >= d) continue;
        const int64_t ab = (int64_t)a * d + b, ba = (int64_t)b * d + a;
        if (bi != bj) {
            M[ab] = j.overwrite ? s[q] : M[ab] + s[q];
            M[ba] = j.overwrite ? s[q] : M[ba] + s[q];
        } else if (a <= b) {                  // diagonal block: the upper triangle is authoritative
            M[ab] = j.overwrite ? s[q] : M[ab] + s[q];
            if (a != b) M[ba] = j.overwrite ? s[q] : M[ba] + s[q];
        }
    }
}

}  // namespace fad
