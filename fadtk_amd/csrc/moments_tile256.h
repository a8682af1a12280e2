// E^T E on 256-column slabs (gfx950): the moments tile kernel for D >= 512.  Included by moments.hip after moments_kernels.h
// (whose small device helpers it uses); the block bookkeeping is in tile256_roles.h.
//
// A workgroup of 8 waves owns a CU, streams TWO 256-column slabs of 32 rows per stage (32 KiB, ring of T2_NST stages) and issues
// 136 MFMAs on them (128 for a plain tile).  Round 6 rewrote the stage loop and the stage layout (rounds 4-5: HISTORY.md):
//
//   * stage layout [32 rows][slab A 256 columns | slab B 256 columns]: one LDS row = 1 KiB = ONE LDS-DMA piece
//     (global_load_lds_dwordx4, SGPR base + lane offset).  For a P / Q item the two slabs are adjacent in memory, so a piece is
//     1 KiB of one frame row, contiguous (scripts/probes/r6_limits.hip: the P + Q pair of an XCD streams 5.45 TB/s with such pieces
//     against 4.56 with pieces of 4 rows x 256 B).  The XOR swizzle of the 16-byte chunks -- chunk ^ 4 (row & 3) -- is applied on
//     the SOURCE side (the DMA writes lane-linear): four lane-offset registers, one per row residue;
//   * fragments: ds_read_b64_tr_b16, two per fragment and 16-row k-step;
//   * FREE-RUNNING waves, ONE barrier per stage.  Every wave runs the same software pipeline -- while it multiplies the fragments of
//     one k-step it reads the next k-step's and issues its share of the refill between the MFMAs -- and the two waves of a SIMD are
//     NOT phase-locked any more (rounds 4-5 ran the two wave quartets half a stage apart, two barriers per stage).  The barrier sits
//     between the two k-steps of a stage: behind it the next stage is in LDS for everybody and the slot of the stage before may be
//     refilled (see the loop).  What bounds the kernel is the package's power budget: it runs at the 1400 W cap;
//   * every wave has a compile-time ROLE (tile256_roles.h): the loop it runs updates a fixed set of accumulators (9 blocks = 144
//     registers, or 8);
//   * column sums (and sum x^2 for the shift guard) ride on the two waves per superblock that read fragments 0..3 / 4..7 anyway
//     (v_dot2c_f32_f16 against (1, 1) resp. against itself);
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
constexpr int T2_ROW = 1024;                // bytes per LDS row of a stage: 256 columns of slab A, then 256 of slab B
constexpr int T2_STAGE = T2_KB * T2_ROW;    // 32 KiB
constexpr size_t kT256Lds = (size_t)T2_NST * T2_STAGE;      // 128 KiB

struct T256Set {
    const void* E; int64_t n, ld, rows_per_split;
    int S, item0;                           // row-splits; first work item (item = item0 + split * NT + type index)
    float* partials;                        // [S][NT][t256::ITEM_STRIDE]
    double* colpart;                        // [S][2][dpad]   (second row: the second wave quartet of a Z item)
    int* flag;                              // shift guard (or nullptr)
    uint16_t* cvec;                         // [S][dpad] float16 shifts for the second pass (or nullptr)
    // IDX launches (fad_moments_update_multi_indexed: the resamples of FAD-inf, fad.py:333-337): frame r of this set is row idx[r] of E,
    // a matrix of n_src rows -- the gather rides on the row offsets of the LDS-DMA pieces, nothing is materialised
    const int32_t* idx; int64_t n_src;
};
struct T256Launch {
    T256Set set[kMaxSets256];
    int nsets, d, nsb, NT, total;
    uint8_t type[t256::MAX_TYPES], sa[t256::MAX_TYPES], sb[t256::MAX_TYPES];
};

// Workgroup barrier that the instruction scheduler may not move anything across (MFMAs have no memory effects: without the
// fences hipcc slides them over a bare s_barrier).
__device__ __forceinline__ void t2_phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ uint4 t2_frag(uint32_t lds_byte) {
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)lds_byte);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(lds_byte + 4 * T2_ROW));      // rows + 4: same swizzle
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

// One wave's share of a work item.  Everything role-dependent is a compile-time constant.
template <int KIND, int ROLE, bool SHIFT, bool IDX>
__device__ __forceinline__ void tile256_wave(
    const T256Launch& L, const T256Set& s, int split, int ti, int type, int sa, int sb, const t256::WaveJob job, char* smem) {
    using RD = t256::RoleDef<ROLE>;
    constexpr int NF = RD::NF, NB = RD::NB;
    static_assert(!SHIFT || KIND == FAD_F16, "the shifted pass is written for float16 rows");
    constexpr int NSTG = T2_NST;                        // ring depth
    constexpr int LPS = 4;                              // LDS-DMA pieces (= LDS rows) per wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quartet = wave >> 2;
    const int li = lane & 31, kg = lane >> 5;
    const uint16_t* __restrict__ E = static_cast<const uint16_t*>(s.E);
    const int64_t ld = s.ld;
    const int d = L.d;
    // Z: slab B = the NEXT 32 rows of slab A's columns; the quartets take one half each (a "stage" of the loop covers 64 rows)
    const bool two_rows = type == t256::TYPE_Z;
    const int64_t k_begin = (int64_t)split * s.rows_per_split;
    const int64_t k_end = (k_begin + s.rows_per_split < s.n) ? k_begin + s.rows_per_split : s.n;
    const int rows_per_stage = two_rows ? 2 * T2_KB : T2_KB;
    const int nkb = (int)((k_end - k_begin + rows_per_stage - 1) / rows_per_stage);
    const int colA = t256::SB * sa, colB = t256::SB * (two_rows ? sa : sb);      // first column behind slab A / slab B
    const int my_rowoff = two_rows ? T2_KB * quartet : 0;                        // rows of a stage this wave's fragments come from

    // ---- loads.  Piece p of wave w = LDS row 4 w + p of the stage, fetched by ONE buffer_load_dwordx4 ... lds.  The DMA writes
    // lane-linear (lane l -> bytes 16 l .. 16 l + 15 of the row), so lane l FETCHES chunk l ^ 4 p of the row (p = row & 3: the
    // swizzle): chunks 0..31 are slab A's columns, 32..63 slab B's (its frame row 32 further down for a Z item).
    // The buffer resource covers exactly this split's rows [k_begin, k_end) -- base = row k_begin, num_records = the bytes up to the
    // end of row k_end - 1 -- and the hardware's range check does the edges: a lane whose offset falls behind the last row reads
    // ZEROS (rows of a last, partial stage; whole stages a short split never had), and a lane whose COLUMN is not below d (ragged
    // last superblock) carries the offset 2^31, out of range for good.  No second form of the load, no per-lane addresses, and the
    // offsets are 32-bit: voff[p] = this lane's offset in stage 0, the stage's offset is added per piece (one v_add_u32; the SGPR
    // offset field of the instruction is NOT range-checked, so it cannot carry it).
    typedef int srd_t __attribute__((ext_vector_type(4)));
    srd_t srd;
    {
        // IDX: the resource covers the whole source matrix (the rows of a split come from anywhere in it)
        const uint64_t b = (uint64_t)(IDX ? E : E + k_begin * ld);
        const uint64_t nbytes = (uint64_t)(((IDX ? s.n_src : k_end - k_begin) - 1) * ld + d) * 2;   // (update_tile256 keeps it below 2^31 bytes)
        srd[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        srd[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));     // stride 0: a raw buffer
        srd[2] = __builtin_amdgcn_readfirstlane((int)(uint32_t)nbytes);
        srd[3] = 0x00020000;                                                              // gfx9 raw buffer, 32-bit data format
    }
    uint32_t voff[LPS];
    bool lane_in_b[LPS];
#pragma unroll
    for (int p = 0; p < LPS; ++p) {
        const int c = lane ^ (p << 2);
        const bool in_b = c >= 32;
        lane_in_b[p] = in_b;
        const int col = in_b ? colB + (c - 32) * 8 : colA + c * 8;                         // d % 8 == 0: a chunk is in or out as a whole
        const int64_t row = IDX ? 0 : LPS * wave + p + ((in_b && two_rows) ? T2_KB : 0);   // (IDX: the row's offset comes from the index)
        voff[p] = col < d ? (uint32_t)((row * ld + col) * 2) : 0x80000000u;
    }
    const uint32_t stage_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((int64_t)rows_per_stage * ld * 2));
    const uint32_t row_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ld * 2));
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(smem_lds + (uint32_t)(LPS * wave * T2_ROW));
    // IDX: the byte offsets of the source rows behind this wave's pieces of one stage (scalar loads; a frame past the end of the split
    // gets 2^31: out of range, zeros).  `ro2`: the second row of a Z item's pieces (slab B = the frame 32 further down).
    struct StageRows { uint32_t ro[LPS], ro2[LPS]; };
    auto rows_of_stage = [&](int kb) -> StageRows {
        StageRows r;
        if constexpr (IDX) {
            const int64_t p0 = k_begin + (int64_t)kb * rows_per_stage + LPS * wave;
#pragma unroll
            for (int p = 0; p < LPS; ++p) {
                const int64_t q = p0 + p, q2 = q + T2_KB;
                r.ro[p] = q < k_end ? (uint32_t)s.idx[q] * row_bytes : 0x80000000u;
                r.ro2[p] = (two_rows && q2 < k_end) ? (uint32_t)s.idx[q2 < k_end ? q2 : q] * row_bytes : 0x80000000u;
            }
        } else {
#pragma unroll
            for (int p = 0; p < LPS; ++p) { r.ro[p] = (uint32_t)kb * stage_bytes; r.ro2[p] = r.ro[p]; }
        }
        return r;
    };
    auto piece = [&](int kb, int p, const StageRows& r) {
#ifdef T2_ABL_NODMA                          // ablation: the ring is filled once and never refilled
        if (kb >= NSTG - 1) return;
#endif
        uint32_t vo;
        if constexpr (IDX) vo = voff[p] + ((two_rows && lane_in_b[p]) ? r.ro2[p] : r.ro[p]);
        else vo = voff[p] + r.ro[p];
        const uint32_t m0v = wave_lds + (uint32_t)((kb % NSTG) * T2_STAGE + p * T2_ROW);
        // (`nt`, the streaming hint, on these loads: 214 against 218 us per 8-matrix launch in the back-to-back harness at the power cap, nothing in
        //  bench.py's loop -- 11 350-11 670 scores/s against 11 650-11 850 -- FETCH_SIZE + 0.2 %: not used.  `lds` has to be the LAST modifier: the
        //  assembler rejects "offen lds nt".)
#ifdef T2_NT
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds" ::"v"(vo), "s"(srd), "s"(m0v) : "memory");
#else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(srd), "s"(m0v) : "memory");
#endif
    };
    auto issue = [&](int kb) {
        const StageRows r = rows_of_stage(kb);
#pragma unroll
        for (int p = 0; p < LPS; ++p) piece(kb, p, r);
    };
    const int nwhole = (int)((k_end - k_begin) / rows_per_stage);                        // stages [0, nwhole) have all their rows

    // ---- fragments: byte offset inside a stage of F[i] for k-step 0 (k-step 1: + 16 rows)
    const int t16 = lane & 15, grp = lane >> 4;
    const int tr_row = 8 * (grp >> 1) + (t16 >> 2), tr_col = 16 * (grp & 1) + 4 * (t16 & 3);
    auto frag_of = [&](int i, bool& from_a) -> int {        // F[i]: which slab, and the fragment's index in its superblock
        if constexpr (ROLE == t256::XR) {
            from_a = i < 4;
            return from_a ? job.a0 + i : job.b0 + (i - 4);
        } else {
            from_a = job.slab == 0;
            return RD::frag[i];
        }
    };
    auto fcol = [&](int i) -> int {                          // global column of this lane's element of F[i] (column sums, shifts)
        bool from_a; const int f = frag_of(i, from_a);
        return (from_a ? colA : colB) + 32 * f + li;
    };
    uint32_t foff[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        bool from_a; const int f = frag_of(i, from_a);
        const int col = (from_a ? 0 : t256::SB) + 32 * f + tr_col;               // column inside the 512-column LDS row
        foff[i] = (uint32_t)(tr_row * T2_ROW + (((col >> 3) ^ ((tr_row & 3) << 2)) << 4) + ((col >> 2) & 1) * 8);
    }
    uint32_t cs[NF];                         // SHIFT: this lane's shift per fragment, packed twice
#pragma unroll
    for (int i = 0; i < NF; ++i) cs[i] = 0u;
    if constexpr (SHIFT) {
        const uint16_t* cv = s.cvec + (int64_t)split * (L.nsb * t256::SB);
#pragma unroll
        for (int i = 0; i < NF; ++i) { const uint32_t h = cv[fcol(i)]; cs[i] = h | (h << 16); }
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

    uint4 F0[NF], F1[NF];                    // the fragments of k-step 0 / 1 (F1 of stage k is read while F0 is multiplied, F0 of stage k + 1 while F1 is)
    auto load_half = [&](int kb, int ks, uint4 (&F)[NF]) {
#ifdef T2_ABL_NOREAD                         // ablation (scripts/probes/tile256_bench.hip): no transpose reads
        if (kb > 0) return;
#endif
        const uint32_t base = smem_lds + (uint32_t)((kb % NSTG) * T2_STAGE + ks * 16 * T2_ROW);
#pragma unroll
        for (int i = 0; i < NF; ++i) F[i] = t2_frag(base + foff[i]);
    };
    // The MFMAs of one k-step; `refill`: this wave's LDS-DMA pieces 2 ks, 2 ks + 1 of stage kb + NSTG - 1 go BETWEEN them.
    auto half = [&](int kb, int ks, const uint4 (&F)[NF], auto refill, const StageRows& rr) {
        constexpr int PH = LPS / 2;          // pieces per k-step: behind MFMA 1, 3
        if constexpr (SHIFT) {               // x - c = x' + e
            if (decltype(refill)::value) { piece(kb + NSTG - 1, PH * ks, rr); piece(kb + NSTG - 1, PH * ks + 1, rr); }
            const bool full = kb < nwhole;
            uint4 X[NF], R[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) t2_split2(F[i], cs[i], full ? 8 : rows_left_at(kb, ks), X[i], R[i]);
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
            return;
        }
#ifdef T2_ABL_NOMMA                          // ablation: no MFMAs (the fragments stay live through a cheap VALU use)
#pragma unroll
        for (int i = 0; i < NF; ++i) acc[0][i] += __uint_as_float((F[i].x ^ F[i].y ^ F[i].z ^ F[i].w) & 0x007fffffu);
        if (decltype(refill)::value) { piece(kb + NSTG - 1, PH * ks, rr); piece(kb + NSTG - 1, PH * ks + 1, rr); }
        return;
#endif
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc[b] = mfma_h16<KIND>(F[RD::fa[b]], F[RD::fb[b]], acc[b]);
            if (decltype(refill)::value && (b & 1) && (b >> 1) < PH) {
                __builtin_amdgcn_sched_barrier(0); piece(kb + NSTG - 1, PH * ks + (b >> 1), rr); __builtin_amdgcn_sched_barrier(0);
            }
#ifndef T2_OPT_NOCOLSUM
            if constexpr (CSUM) {
                if (b == 4 || b == 6) {      // the column sums of one fragment each, among the later MFMAs
                    const int i = (b - 4) >> 1;
                    csum[i] += (double)sum8<KIND>(F[CS0 + i]);
                }
            }
#endif
        }
    };
    // this wave's pieces of stage `kb + 1` have landed once at most (stages issued beyond it) x LPS of its loads are outstanding
    auto wait_next = [&](int kb) {
        const int last = (kb + NSTG - 1 < nkb - 1) ? kb + NSTG - 1 : nkb - 1;         // youngest stage issued so far
        wait_vmcnt_upto<LPS>(last - (kb + 1));
    };
    static_assert(NSTG <= 8 && NSTG >= 3, "wait_vmcnt_upto counts at most seven stages");

    // ---- the loop.  Barrier B(k + 1) sits between the two k-steps of stage k.  A wave arrives there with (a) its reads of stage k
    // complete (F1 came back: lgkmcnt(0)) and (b) its own pieces of stage k + 1 landed (counted vmcnt).  Hence behind B(k + 1): stage
    // k + 1 may be read by everybody; and the slot of stage k may be refilled -- which is what iteration k + 1 does with the pieces of
    // stage k + NSTG (same slot), all of them issued behind B(k + 1).
    for (int s0 = 0; s0 < NSTG - 1 && s0 < nkb; ++s0) issue(s0);
    wait_vmcnt_upto<LPS>(((nkb < NSTG - 1) ? nkb : NSTG - 1) - 1);                    // stages issued beyond stage 0
    t2_phase_barrier();
    load_half(0, 0, F0);
    const int hot = nkb - (NSTG - 1) > 0 ? nkb - (NSTG - 1) : 0;                      // stages behind which a refill is due
    int kb = 0;
    // (static priority for waves 4..7 and the next k-step's reads spread one per MFMA were measured -- within the +-3 % of the harness,
    //  profiles/r06b_tile256_variants.txt -- and are not in the code)
    for (; kb < hot; ++kb) {                 // no branches: the refill rides between the MFMAs
        const StageRows rr = rows_of_stage(kb + NSTG - 1);     // (IDX: four scalar loads, in flight while the k-step's reads are issued)
        load_half(kb, 1, F1);
        __builtin_amdgcn_sched_barrier(0);
        half(kb, 0, F0, std::true_type{}, rr);
        __builtin_amdgcn_sched_barrier(0);
        // in flight at this point: stages kb + 1 .. kb + NSTG - 2 and half of kb + NSTG - 1; kb + 1 must have landed
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS * (NSTG - 3) + LPS / 2) : "memory");
        t2_phase_barrier();
        load_half(kb + 1, 0, F0);
        __builtin_amdgcn_sched_barrier(0);
        half(kb, 1, F1, std::true_type{}, rr);
        __builtin_amdgcn_sched_barrier(0);
    }
    const StageRows none = rows_of_stage(0);
    for (; kb < nkb; ++kb) {                 // the last NSTG - 1 stages: nothing left to fetch
        load_half(kb, 1, F1);
        __builtin_amdgcn_sched_barrier(0);
        half(kb, 0, F0, std::false_type{}, none);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool more = kb + 1 < nkb;
        if (more) {
            wait_next(kb);
            t2_phase_barrier();
            load_half(kb + 1, 0, F0);
            __builtin_amdgcn_sched_barrier(0);
        }
        half(kb, 1, F1, std::false_type{}, none);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: blocks, fragment major -- float4 (q, lane) of block b = registers 4q..4q+3 = rows 8q + 4 (lane >> 5) + 0..3
    // of column lane & 31
    {
        float4* out = reinterpret_cast<float4*>(s.partials + ((int64_t)split * L.NT + ti) * t256::ITEM_STRIDE) + (size_t)(9 * wave) * 256;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(b * 4 + q) * 64 + lane] = make_float4(acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]);
    }

    // ---- shift guard, first pass: sum x^2 of a column is the DIAGONAL of its 32 x 32 diagonal block -- it sits in the accumulators
    // of whichever triangle-family wave owns that block (rounds 4-5 summed it beside the column sums: 16 more v_dot2c per wave and
    // stage, 9 % of the launch).  The owners leave the diagonals in LDS (the ring is idle: every wave is behind its last read), the
    // waves that hold the column sums pick theirs up.  All eight waves pass the two barriers (XR waves do nothing else).
    float s2col[2] = {0.f, 0.f};
    if constexpr (!SHIFT) {
        if (s.flag) {                        // (uniform over the workgroup: a property of the set)
            float* dsq = reinterpret_cast<float*>(smem) + quartet * t256::SB;
            __syncthreads();
            if constexpr (CSUM) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (RD::fa[b] != RD::fb[b]) continue;                  // (a diagonal block: element (c, c) is register 4 (c >> 3) + (c & 3) of lane 32 ((c >> 2) & 1) + c)
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((li >> 3) == q && (li & 3) == j) v = acc[b][4 * q + j];
                    if (kg == ((li >> 2) & 1)) dsq[32 * RD::frag[RD::fa[b]] + li] = v;
                }
            }
            __syncthreads();
            if constexpr (CSUM) {
#pragma unroll
                for (int i = 0; i < 2; ++i) s2col[i] = dsq[32 * RD::frag[CS0 + i] + li];
            }
        }
    }

    if constexpr (CSUM) {
        // column sums of this wave's two fragments over the rows it saw; the second quartet of a Z item writes the second row
        const int dpad = L.nsb * t256::SB;
        const int half_ = (two_rows && job.slab) ? 1 : 0;
        double* cp = s.colpart + ((int64_t)split * 2 + half_) * dpad;
        int64_t my_rows = k_end - k_begin;
        if (two_rows) {    // rows of the 64-row stages that fall into this quartet's half
            const int64_t whole = my_rows / 64, rem = my_rows - whole * 64;
            my_rows = whole * 32 + (half_ ? (rem > 32 ? rem - 32 : 0) : (rem < 32 ? rem : 32));
        }
        bool hit = false;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = fcol(CS0 + i);
            double sx = csum[i];
            sx += __shfl_xor(sx, 32);
            if (kg == 0) cp[col] = sx;
            if constexpr (!SHIFT) {
                if (s.flag) {
                    const double s2 = (double)s2col[i];
                    const double nr = (double)my_rows;
                    const double mean = sx / nr, var = s2 / nr - mean * mean;
                    const bool col_in = col < d;
                    // the same rule as the 128-kernel; sum x^2 is a float32 sum here (relative error ~1e-6: immaterial against 64 x)
                    hit = hit || (col_in && nr > 0.0 && !(mean * mean <= 64.0 * var) && !(sx == 0.0 && s2 == 0.0));
                    if (s.cvec && kg == 0 && half_ == 0) {
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

// Both passes are held to 224 registers (amdgpu_num_vgpr counts HALF of the unified file: 112 -> 224, found on a toy kernel): a
// workgroup of 256-register waves fills its SIMDs, and even the SHIFT launch that only reads the gate and exits could not be PLACED on a
// CU while a wave of the running-sum walk (moments_kernels.h: moments_running_colsum_h16, 64 registers, on a stream of its own) sat
// there -- r05d: the caller's stream stood still for 170-300 us per update.
template <int KIND, bool SHIFT, bool IDX = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(112))) void moments_tile256(T256Launch L) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_dyn[];
    char* const smem_bytes = reinterpret_cast<char*>(smem_dyn);
    const int w = xcd_contiguous(blockIdx.x, L.total);
    int si = 0;
#pragma unroll
    for (int i = 1; i < kMaxSets256; ++i)
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
        case t256::TRI_LO: tile256_wave<KIND, t256::TRI_LO, SHIFT, IDX>(L, s, split, ti, type, sa, sb, job, smem_bytes); break;
        case t256::TRI_HI: tile256_wave<KIND, t256::TRI_HI, SHIFT, IDX>(L, s, split, ti, type, sa, sb, job, smem_bytes); break;
        case t256::RECT_C: tile256_wave<KIND, t256::RECT_C, SHIFT, IDX>(L, s, split, ti, type, sa, sb, job, smem_bytes); break;
        case t256::RECT_D: tile256_wave<KIND, t256::RECT_D, SHIFT, IDX>(L, s, split, ti, type, sa, sb, job, smem_bytes); break;
        default: tile256_wave<KIND, t256::XR, SHIFT, IDX>(L, s, split, ti, type, sa, sb, job, smem_bytes); break;
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
    R256Job job[kMaxSets256];
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
