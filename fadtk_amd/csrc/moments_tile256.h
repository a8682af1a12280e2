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
constexpr int T2_NST = 4;                   // ring depth
constexpr int T2_SUB = T2_KB * 16;          // uint4 per [32][128] sub-slab
constexpr int T2_STAGE = 4 * T2_SUB;        // A0 A1 B0 B1: 32 KiB
constexpr size_t kT256Lds = (size_t)T2_NST * T2_STAGE * sizeof(uint4);      // 128 KiB

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
};

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

// One wave's share of a work item.  Everything role-dependent is a compile-time constant; `slabsel[i]` (0 = slab A, 1 = slab B)
// and `fragid[i]` (0..7 within the superblock) say where fragment F[i] comes from.
template <int KIND, int ROLE, bool SHIFT>
__device__ __forceinline__ void tile256_wave(
    const T256Launch& L, const T256Set& s, int split, int ti, int type, int sa, int sb, const t256::WaveJob job, uint4* smem) {
    using RD = t256::RoleDef<ROLE>;
    constexpr int NF = RD::NF, NB = RD::NB;
    constexpr bool TRI = (ROLE == t256::TRI_LO || ROLE == t256::TRI_HI);      // carries the column sums of its four fragments
    static_assert(!SHIFT || KIND == FAD_F16, "the shifted pass is written for float16 rows");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kg = lane >> 5;
    const uint16_t* __restrict__ E = static_cast<const uint16_t*>(s.E);
    const int64_t ld = s.ld;
    const int d = L.d;
    const bool zt = type == t256::TYPE_Z;
    const int64_t k_begin = (int64_t)split * s.rows_per_split;
    const int64_t k_end = (k_begin + s.rows_per_split < s.n) ? k_begin + s.rows_per_split : s.n;
    const int rows_per_stage = zt ? 2 * T2_KB : T2_KB;                          // Z: slab B = the NEXT 32 rows of slab A's columns
    const int nkb = (int)((k_end - k_begin + rows_per_stage - 1) / rows_per_stage);
    const int colA = t256::SB * sa, colB = t256::SB * (zt ? sa : sb);           // first column behind slab A / slab B

    // ---- loads: wave w fills rows 16 (w & 1) + 4 q + (lane >> 4), q = 0..3, of sub-slab w >> 1 (0, 1: slab A; 2, 3: slab B)
    const int sub = wave >> 1;
    const int ld_col0 = (sub < 2 ? colA : colB) + 128 * (sub & 1);
    const int ld_rowoff = 16 * (wave & 1) + ((zt && sub >= 2) ? T2_KB : 0);
    const int lrow = lane >> 4, lchunk = (lane & 15) ^ (lrow << 2);             // swizzle on the SOURCE side (LDS-DMA writes lane-linear)
    const bool col_ok = (ld_col0 + lchunk * 8) < d;                             // d % 8 == 0: a chunk is in or out as a whole
    const uint32_t voff = (uint32_t)(((int64_t)lrow * ld + lchunk * 8) * 2);
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(&g_zero16);
    const bool cols_full = (colA + t256::SB <= d) && (colB + t256::SB <= d) && ld < ((int64_t)1 << 26);
    auto issue_fast = [&](int kb) {
        const uint32_t dst0 = smem_lds + (uint32_t)(((kb % T2_NST) * T2_STAGE + sub * T2_SUB + 256 * (wave & 1)) * 16);
        const uint16_t* src0 = E + (k_begin + (int64_t)kb * rows_per_stage + ld_rowoff) * ld + ld_col0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t sbq = (uint64_t)(src0 + (int64_t)(4 * q) * ld);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sbq);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sbq >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(64 * q * 16));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    auto issue_slow = [&](int kb) {         // edge stages: per-lane 64-bit addresses, rows / columns out of range read the zero block
        const uint32_t dst0 = smem_lds + (uint32_t)(((kb % T2_NST) * T2_STAGE + sub * T2_SUB + 256 * (wave & 1)) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = k_begin + (int64_t)kb * rows_per_stage + ld_rowoff + 4 * q + lrow;
            const uint16_t* src = (r < k_end && col_ok) ? E + r * ld + ld_col0 + lchunk * 8 : zsrc;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(64 * q * 16));
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
    int fslab[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        int slab, f;
        if constexpr (ROLE == t256::XR) { slab = (i < 4) ? 0 : 1; f = (i < 4) ? job.a0 + i : job.b0 + (i - 4); }
        else { slab = job.slab; f = RD::frag[i]; }
        const int col = 32 * (f & 3) + tr_col;
        foff[i] = (uint32_t)((2 * slab + (f >> 2)) * (T2_SUB * 16) + tr_row * 256 + (((col >> 3) ^ ((tr_row & 3) << 2)) << 4) + ((col >> 2) & 1) * 8);
        fcol[i] = (slab == 0 ? colA : colB) + 32 * f + li;
        fslab[i] = slab;
    }
    uint32_t cs[NF];                         // SHIFT: this lane's shift per fragment, packed twice
#pragma unroll
    for (int i = 0; i < NF; ++i) cs[i] = 0u;
    if constexpr (SHIFT) {
        const uint16_t* cv = s.cvec + (int64_t)split * (L.nsb * t256::SB);
#pragma unroll
        for (int i = 0; i < NF; ++i) { const uint32_t h = cv[fcol[i]]; cs[i] = h | (h << 16); }
    }
    auto rows_left_at = [&](int kb, int ks, int slab) -> int64_t {
        return k_end - (k_begin + (int64_t)kb * rows_per_stage + ((zt && slab) ? T2_KB : 0) + ks * 16 + 8 * kg);
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
    double csum[4] = {0.0, 0.0, 0.0, 0.0};
    float csq[4] = {0.f, 0.f, 0.f, 0.f};

    // one stage's arithmetic: all transpose reads of both k-steps, then the MFMAs (the compiler interleaves them by lgkmcnt)
    auto compute = [&](int kb) {
        const uint32_t base = smem_lds + (uint32_t)((kb % T2_NST) * T2_STAGE * 16);
        if constexpr (SHIFT) {              // one k-step at a time (x', e and the raw fragment of both k-steps do not fit the registers)
            const bool full = kb < nfast;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 X[NF], R[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) X[i] = t2_frag(base + foff[i] + 4096 * ks);
#pragma unroll
                for (int i = 0; i < NF; ++i) t2_split2(X[i], cs[i], full ? 8 : rows_left_at(kb, ks, fslab[i]), X[i], R[i]);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[b] = mfma_h16<KIND>(X[RD::fa[b]], X[RD::fb[b]], acc[b]);
                    acc[b] = mfma_h16<KIND>(X[RD::fa[b]], R[RD::fb[b]], acc[b]);
                    acc[b] = mfma_h16<KIND>(R[RD::fa[b]], X[RD::fb[b]], acc[b]);
                }
                if constexpr (TRI) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) csum[i] += (double)sum8<KIND>(X[i]) + (double)sum8<KIND>(R[i]);
                }
            }
            return;
        }
        uint4 F0[NF], F1[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) F0[i] = t2_frag(base + foff[i]);
#pragma unroll
        for (int i = 0; i < NF; ++i) F1[i] = t2_frag(base + foff[i] + 4096);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = mfma_h16<KIND>(F0[RD::fa[b]], F0[RD::fb[b]], acc[b]);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = mfma_h16<KIND>(F1[RD::fa[b]], F1[RD::fb[b]], acc[b]);
        if constexpr (TRI) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                csum[i] += (double)sum8<KIND>(F0[i]) + (double)sum8<KIND>(F1[i]);
                csq[i] = t2_sumsq8<KIND>(F1[i], t2_sumsq8<KIND>(F0[i], csq[i]));
            }
        }
    };

    for (int s0 = 0; s0 < T2_NST - 1 && s0 < nkb; ++s0) issue(s0);
    // hot loop: the stage to refill is a whole one (SGPR-base loads only) and two younger stages stay in flight -- no branches
    const int hot = nfast - (T2_NST - 1) > 0 ? nfast - (T2_NST - 1) : 0;
    int kb = 0;
    for (; kb < hot; ++kb) {
        wait_vmcnt<4 * (T2_NST - 2)>();
        __builtin_amdgcn_s_barrier();               // stage kb is in LDS, stage kb - 1 is free
        issue_fast(kb + T2_NST - 1);
        compute(kb);
    }
    for (; kb < nkb; ++kb) {
        const int ahead = (nkb - 1 - kb < T2_NST - 2) ? (nkb - 1 - kb) : (T2_NST - 2);
        if (ahead >= 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + T2_NST - 1 < nkb) issue(kb + T2_NST - 1);
        compute(kb);
    }

    // ---- epilogue: blocks, fragment major -- float4 (q, lane) of block b = registers 4q..4q+3 = rows 8q + 4 (lane >> 5) + 0..3
    // of column lane & 31
    float4* out = reinterpret_cast<float4*>(s.partials + ((int64_t)split * L.NT + ti) * t256::ITEM_STRIDE) + (size_t)(9 * wave) * 256;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[(b * 4 + q) * 64 + lane] = make_float4(acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]);

    if constexpr (TRI) {
        // column sums of this wave's four fragments over the rows it saw; a Z item's second quartet (slab B) writes the second row
        const int dpad = L.nsb * t256::SB;
        const int half = (zt && job.slab) ? 1 : 0;
        double* cp = s.colpart + ((int64_t)split * 2 + half) * dpad;
        int64_t my_rows = k_end - k_begin;
        if (zt) {       // rows of the 64-row stages that fall into this quartet's half
            const int64_t whole = my_rows / 64, rem = my_rows - whole * 64;
            my_rows = whole * 32 + (half ? (rem > 32 ? rem - 32 : 0) : (rem < 32 ? rem : 32));
        }
        bool hit = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sx = csum[i];
            sx += __shfl_xor(sx, 32);
            if (kg == 0) cp[fcol[i]] = sx;
            if constexpr (!SHIFT) {
                if (s.flag) {
                    double s2 = (double)csq[i];
                    s2 += __shfl_xor(s2, 32);
                    const double nr = (double)my_rows;
                    const double mean = sx / nr, var = s2 / nr - mean * mean;
                    const bool col_in = fcol[i] < d;
                    // the same rule as the 128-kernel; sum x^2 is a float32 sum here (relative error ~1e-6: immaterial against 64 x)
                    hit = hit || (col_in && nr > 0.0 && !(mean * mean <= 64.0 * var) && !(sx == 0.0 && s2 == 0.0));
                    if (s.cvec && kg == 0 && half == 0) {
                        const bool worth = col_in && nr > 0.0 && (mean * mean > var) && (mean == mean) && !isinf(mean) && fabs(mean) < 65000.0;
                        const _Float16 ch = worth ? (_Float16)(float)mean : (_Float16)0.0f;
                        uint16_t bits; __builtin_memcpy(&bits, &ch, 2);
                        s.cvec[(int64_t)split * dpad + fcol[i]] = bits;
                    }
                }
            }
        }
        if constexpr (!SHIFT) {
            if (s.flag && __any(hit) && lane == 0) atomicOr(s.flag, 1);
        }
    }
}

template <int KIND, bool SHIFT>
__global__ __launch_bounds__(512) void moments_tile256(T256Launch L) {
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
    switch (job.role) {              // wave-uniform: every wave runs ONE of these loops, all with the same stage count and barriers
        case t256::TRI_LO: tile256_wave<KIND, t256::TRI_LO, SHIFT>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        case t256::TRI_HI: tile256_wave<KIND, t256::TRI_HI, SHIFT>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        case t256::RECT_C: tile256_wave<KIND, t256::RECT_C, SHIFT>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        case t256::RECT_D: tile256_wave<KIND, t256::RECT_D, SHIFT>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
        default: tile256_wave<KIND, t256::XR, SHIFT>(L, s, split, ti, type, sa, sb, job, smem_dyn); break;
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
    uint8_t z_sb;                            // superblock whose column sums have a second row (Z item), or 255
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
        const int a = (block - tile_blocks) * 256 + threadIdx.x;
        if (a == 0) j.acc[0] = j.overwrite ? j.n_add : j.acc[0] + j.n_add;
        if (a >= R.d) return;
        const bool two = (a / t256::SB) == (int)R.z_sb;
        double s0 = 0.0, s1 = 0.0;
        for (int sp = 0; sp < S; ++sp) {
            s0 += j.colpart[((int64_t)sp * 2) * dpad + a];
            if (two) s1 += j.colpart[((int64_t)sp * 2 + 1) * dpad + a];
            if (unshift) s1 += rows_of(sp) * f16_bits_to_f64(j.cvec[(int64_t)sp * dpad + a]);
        }
        j.acc[1 + a] = j.overwrite ? s0 + s1 : j.acc[1 + a] + (s0 + s1);
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
            for (; sp < S; sp += SL) {
                const float4 v = *reinterpret_cast<const float4*>(p + sp * stride);
                s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            }
        }
        if (unshift) {                        // + c_a s'_b + s'_a c_b + n c_a c_b, split by split (see reduce_body)
            const int ga = 32 * bi + 8 * (e >> 6) + 4 * ((e & 63) >> 5), gb = 32 * bj + (e & 31);
            const bool two_a = (ga / t256::SB) == (int)R.z_sb, two_b = (gb / t256::SB) == (int)R.z_sb;
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
