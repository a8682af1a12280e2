// Batched fp64 GEMM  C = alpha * A * B + beta_eye * I  on v_mfma_f64_16x16x4_f64 (gfx950).
//
// The Newton-Schulz square-root iteration of the Frechet distance is three of these per step
// (replaces scipy.linalg.sqrtm / eig of fadtk/fad.py:88-92).  fp64 is required: the distance is a
// cancellation and fp32 iterations miss the 1e-4 bar at N=100k, D=512.
//
// 256 threads = 4 waves as 2x2; workgroup tile BT x BT (32 or 64), 64-deep k stages, register
// prefetch of the next stage, A stored [row][k] with pitch 66 and B [k][col] with pitch BT+16 --
// both conflict-free for ds_read_b64.  MFMA fragment maps (f64 differs from the f32 maps):
//   A: lane l holds A[i = l&15][k = l>>4]     B: lane l holds B[k = l>>4][j = l&15]
//   D: col = l&15, row = (l>>4) + 4*reg       (reg = 0..3)
// Optional epilogue: per-workgroup sum of (C - gamma I)^2 written to a partial slot
// (deterministic two-level reduction; no atomics).
#include "fad_common.h"
#include "ns_check.h"
#include "ns32.h"
#include "ns_mean.h"

#include <cstdlib>

namespace fad {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// blockIdx.z = batch_index * ntypes + type.  Each "type" is one GEMM shape of the iteration (e.g.
// Y' = Y T and Z' = T Z share a launch); each batch index is one independent problem (one song).
struct GemmArgs {
    const double* A[2]; const double* B[2]; double* C[2];
    int64_t sa[2], sb[2], sc[2];         // per-batch-index strides (elements); 0 = shared operand
    double alpha[2], beta_eye[2], gamma[2];
    const double* mu[2]; int64_t mu_stride[2];      // GemmType::mu (device-side step scale of the T product), or nullptr
    double* partials[2];                 // [batch][slots] or nullptr
    const int* skip;                     // skip[batch_index * skip_stride] != 0 -> problem is finished
    int skip_stride;
    int ntypes;
    int remap;                           // XCD-aware tile map on/off
    int pstride;                         // partial slots reserved per problem
    int gemm_z;                          // blockIdx.z >= gemm_z: checker blocks (problem = blockIdx.z - gemm_z)
    int sym;                             // GemmType::sym (all types of the launch)
    int tri;                             // B upper triangular: the k loop of column tile tx ends with the tile's last column
    // fp32 operands (TIn = float instantiation: the fp64 correction product of the mixed-precision Newton-Schulz):
    // A32/B32 replace A/B; when `sel` is given and *sel is odd the *_alt pointers are used (the final iterate of
    // the low-precision iteration lives in one of two ping-pong buffers, known only on the device)
    const float* A32; const float* B32; const float* A32_alt; const float* B32_alt;
    const int* sel;
    NsCheckArgs chk;
    NsProductExt ext;                    // MODE 1 / 2 epilogues (ns32.h)
};

constexpr int KB = 64;                 // k depth of one LDS stage
constexpr int PA = KB + 2;             // A pitch (doubles): rows i..i+15 land on distinct bank pairs

// The operands of a D=512 iteration are 2 MB matrices that the PREVIOUS kernel wrote from all 8 XCDs, so
// every first touch is an L2 miss served by the Infinity Cache / HBM (~1-2 us), and each panel is wanted by
// 16 workgroups.  Two measures (the first version ran at ~20 % of the fp64 MFMA rate, latency-bound):
//   * prefetch DEPTH stages ahead into registers (16-byte loads), so a thread has DEPTH x 64 k of both
//     operands in flight while the MFMAs chew on the LDS-resident stage;
//   * XCD-aware tile map: workgroup b runs on XCD b % 8; XCD (ex, ey) of a 2 x 4 grid owns a contiguous
//     block of output tiles, so each private L2 fetches 1/2 of A's rows and 1/4 of B's columns once.
// KSPLIT (used for the 32 x 32 tile): every wave owns the WHOLE tile as 2 x 2 MFMA tiles but only every
// 4th k-step; the four partial tiles are summed through LDS at the end.  A wave then carries four
// independent accumulators instead of one 128-long dependent chain: measured, the dependent
// v_mfma_f64_16x16x4_f64 chain (not the loads) was what held the first versions at ~20 % of peak.
// (Tried and dropped: splitting k over TWO workgroups per tile that meet through a ticket in global memory.  The
// device-scope release/acquire it needs writes back / invalidates the XCD's whole L2 per workgroup: 0.18 -> 0.53 ms.)
// NW = waves per workgroup (4; 8 only with KSPLIT): a single D = 512 GEMM is 256 workgroups = one per CU, and four
// waves per CU run the fp64 MFMA at 34 TFLOP/s where eight reach 45 (scripts/probes/mfma_rate.hip).
// MODE (KSPLIT + FULL, one problem): what the workgroup does with its finished 32 x 32 tile besides / instead of storing it
//   1  C = C1 C2 of the Frechet distance: also the tile's statistics for the scale of the iteration (what ns_tilestats
//      computes in a launch of its own: row / column sums of |a|, sum a^2, shares of tr A, tr C1, tr C2); one spare
//      workgroup (blockIdx.z == 1) forms the mean term ||mu1 - mu2||^2 meanwhile
//   2  G = Y Y of the fp64 correction: the tile is NOT stored; R = A/c - G stays in the workgroup, which writes the
//      tile's share of tr(Z R), ||R||_F^2, tr Y and the row / column sums of |Z| (what ns32_corr_partials did in a
//      launch of its own, re-reading A and G)
template <int BT, int DEPTH, bool KSPLIT, bool FULL, int NW = 4, typename TIn = double, int MODE = 0>
__global__ __launch_bounds__(NW * 64) void gemm_f64_kernel(int d, GemmArgs g) {
    constexpr int NT = NW * 64;              // threads
    constexpr int MT = KSPLIT ? 2 : BT / 32; // MFMA tiles per wave per side
    constexpr int PB = BT + 16;
    constexpr int NV = BT * KB / 2 / NT;     // double2 loads per thread per operand per stage
    constexpr int BV = BT / 2;               // double2 per B row
    constexpr int STAGE_D = BT * PA + KB * PB;                  // doubles of one LDS stage
    constexpr int SMEM_D = (KSPLIT && NW * 32 * 33 > STAGE_D) ? NW * 32 * 33 : STAGE_D;
    __shared__ __attribute__((aligned(16))) double smem[SMEM_D + 8];
    double* sA = smem;
    double* sB = smem + BT * PA;
    double* red = smem + SMEM_D;

    if constexpr (MODE == 1) {
        if (blockIdx.z == 1) {                 // the spare workgroup: mean term -> state (256 threads, the other waves leave)
            __shared__ float gaps[1024];
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 256) {
                const double mt = mean_term_block(g.ext.mu1, g.ext.mu2, d, g.ext.mean_dtype, gaps, red);
                if (threadIdx.x == 0) g.ext.st->mean_term = mt;
            }
            return;
        }
    }
    if ((int)blockIdx.z >= g.gemm_z) {         // checker blocks: one live workgroup per problem
        // (the check is written for 256 threads; surplus waves leave, a finished wave no longer counts at s_barrier)
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 256) ns_check_block(g.chk, (int64_t)blockIdx.z - g.gemm_z, red);
        return;
    }
    const int zi = (g.ntypes == 2) ? (blockIdx.z & 1) : 0;
    int64_t zb = (g.ntypes == 2) ? (blockIdx.z >> 1) : blockIdx.z;
    int ty = blockIdx.y, tx = blockIdx.x;
    if (g.tri) {
        // column tile j of a triangular B costs j + 1 stages: hand the tiles out longest first (workgroups start in the order of
        // their linear index), so that the tail of the launch is made of the one-stage tiles
        const int64_t L = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int64_t per = (int64_t)gridDim.y * g.gemm_z;
        tx = (int)(gridDim.x - 1 - L / per);
        const int64_t rem = L % per;
        zb = rem / gridDim.y; ty = (int)(rem % gridDim.y);
    }
    if (g.skip && g.skip[zb * g.skip_stride] != 0) return;
    const double* A = g.A[zi] + zb * g.sa[zi];
    const double* B = g.B[zi] + zb * g.sb[zi];
    const float* A32 = g.A32; const float* B32 = g.B32;
    if constexpr (sizeof(TIn) == 4) {
        if (g.sel && (*g.sel & 1)) { A32 = g.A32_alt; B32 = g.B32_alt; }
    }
    double* C = g.C[zi] + zb * g.sc[zi];
    double alpha = g.alpha[zi], beta_eye = g.beta_eye[zi], gamma = g.gamma[zi];
    if (g.mu[zi]) {                                 // scaled Newton-Schulz step: T = 1.5 m I - 0.5 m^3 Z Y
        const double m = g.mu[zi][zb * g.mu_stride[zi]];
        alpha = -0.5 * m * m * m; beta_eye = 1.5 * m; gamma = beta_eye + alpha;
    }

    // ---- tile coordinates (XCD-aware when the tile grid splits evenly into 2 x 4 blocks)
    const int t = gridDim.x;
    if (g.remap && (t & 3) == 0) {
        const int b = blockIdx.y * t + blockIdx.x;
        // (symmetric products: the blocks below the diagonal have no work, so problem zb gives block (h + zb) % 8 to XCD h --
        //  over a batch every XCD gets every block; measured without the rotation: no gain at all from skipping 28 of 64 tiles)
        const int xcd = (b + (g.sym ? (int)(zb & 7) : 0)) & 7, idx = b >> 3;
        const int R = t >> 1, Cc = t >> 2;
        ty = (xcd >> 2) * R + idx / Cc;
        tx = (xcd & 3) * Cc + idx % Cc;
    }
    const int slot = ty * t + tx;
    if constexpr (!KSPLIT && MODE == 0) {
        if (g.sym && tx < ty) {              // mirror image of a tile another workgroup computes; its residual slot reads as zero
            if (g.partials[zi] && threadIdx.x == 0) g.partials[zi][zb * g.pstride + slot] = 0.0;
            return;
        }
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: plain k-loop, no exec masking
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int row0 = ty * BT, col0 = tx * BT;
    const bool vec = ((d & 1) == 0);         // 16-byte loads need even d (row starts stay 16-B aligned)
    int nkb = (d + KB - 1) / KB;
    if (g.tri) nkb = min(nkb, (min(col0 + BT, d) + KB - 1) / KB);        // rows k > the tile's last column of B are zeros

    // three named register sets (an array of arrays indexed in a loop ends up in scratch memory, which
    // forces a wait on every prefetch right after it is issued)
    d2 ra0[NV], rb0[NV], ra1[NV], rb1[NV], ra2[NV], rb2[NV];
    auto fetch = [&](d2 (&pa)[NV], d2 (&pb)[NV], int kb) {
        if (kb >= nkb) return;
        const int k0 = kb * KB;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = tid + q * NT;
            const int ai = e / (KB / 2), ak = (e % (KB / 2)) * 2;        // A tile [BT][64], pairs along k
            const int r = row0 + ai, k = k0 + ak;
            const int bk = e / BV, bj = (e % BV) * 2;                    // B tile [64][BT], pairs along j
            const int kk = k0 + bk, c = col0 + bj;
            if constexpr (FULL && sizeof(TIn) == 4) {       // fp32 operands, converted on the way in
                const float2 fa = *reinterpret_cast<const float2*>(A32 + (int64_t)r * d + k);
                const float2 fb = *reinterpret_cast<const float2*>(B32 + (int64_t)kk * d + c);
                pa[q] = (d2){(double)fa.x, (double)fa.y};
                pb[q] = (d2){(double)fb.x, (double)fb.y};
            } else if constexpr (FULL) {      // d % 64 == 0: no bounds, no branches -- hipcc otherwise wraps EVERY guarded load in an exec
                             // branch with its own s_waitcnt vmcnt(0), which serialises the whole prefetch
                pa[q] = *reinterpret_cast<const d2*>(A + (int64_t)r * d + k);
                pb[q] = *reinterpret_cast<const d2*>(B + (int64_t)kk * d + c);
            } else if (vec) {
                pa[q] = (r < d && k < d) ? *reinterpret_cast<const d2*>(A + (int64_t)r * d + k) : (d2){0.0, 0.0};
                pb[q] = (kk < d && c < d) ? *reinterpret_cast<const d2*>(B + (int64_t)kk * d + c) : (d2){0.0, 0.0};
            } else {
                pa[q].x = (r < d && k < d) ? A[(int64_t)r * d + k] : 0.0;
                pa[q].y = (r < d && k + 1 < d) ? A[(int64_t)r * d + k + 1] : 0.0;
                pb[q].x = (kk < d && c < d) ? B[(int64_t)kk * d + c] : 0.0;
                pb[q].y = (kk < d && c + 1 < d) ? B[(int64_t)kk * d + c + 1] : 0.0;
            }
        }
    };

    f64x4 acc[MT][MT];
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < MT; ++y) acc[x][y] = (f64x4){0.0, 0.0, 0.0, 0.0};

    // one stage: registers -> LDS, refill the same registers DEPTH stages ahead, MFMAs from LDS
    auto stage = [&](d2 (&pa)[NV], d2 (&pb)[NV], int kb) {
        if (kb >= nkb) return;
        if (kb) __syncthreads();                 // everyone is done reading the previous stage
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = tid + q * NT;
            *reinterpret_cast<d2*>(sA + (e / (KB / 2)) * PA + (e % (KB / 2)) * 2) = pa[q];
            *reinterpret_cast<d2*>(sB + (e / BV) * PB + (e % BV) * 2) = pb[q];
        }
        __syncthreads();
        fetch(pa, pb, kb + DEPTH);
#pragma unroll 4
        for (int ks = (KSPLIT ? wave : 0); ks < KB / 4; ks += (KSPLIT ? NW : 1)) {
            const int k = ks * 4 + lk;
            double a[MT], b[MT];
#pragma unroll
            for (int f = 0; f < MT; ++f) {
                a[f] = sA[((KSPLIT ? 0 : wr * (BT / 2)) + 16 * f + li) * PA + k];
                b[f] = sB[k * PB + (KSPLIT ? 0 : wc * (BT / 2)) + 16 * f + li];
            }
#pragma unroll
            for (int fa = 0; fa < MT; ++fa)
#pragma unroll
                for (int fb = 0; fb < MT; ++fb)
                    acc[fa][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
        }
    };

    fetch(ra0, rb0, 0);
    if (DEPTH > 1) fetch(ra1, rb1, 1);
    if (DEPTH > 2) fetch(ra2, rb2, 2);
    for (int kb = 0; kb < nkb; kb += DEPTH) {
        stage(ra0, rb0, kb);
        if (DEPTH > 1) stage(ra1, rb1, kb + 1);
        if (DEPTH > 2) stage(ra2, rb2, kb + 2);
    }

    double ss = 0.0;
    if constexpr (KSPLIT && MODE != 0) {
        static_assert(MODE == 0 || (NW == 8 && FULL), "statistics epilogues: 512 threads, d % 64 == 0");
        __shared__ double sred[8 * 5];
        // operands of the epilogue, requested before the partial tiles are summed: element (rr, cc) of this tile, and for
        // MODE 2 the element (cc, rr) of Z's mirror tile (tr(Z R) pairs R_ij with Z_ji)
        double a_in[2]; float z_in[2]; double diag_in[2][2];
        const float* Zf = nullptr;
        if constexpr (MODE == 2) Zf = (g.sel && (*g.sel & 1)) ? g.ext.Z32_alt : g.ext.Z32;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * NT, rr = e >> 5, cc = e & 31;
            const int r = row0 + rr, c = col0 + cc;
            a_in[q] = 0.0; z_in[q] = 0.f; diag_in[q][0] = 0.0; diag_in[q][1] = 0.0;
            if constexpr (MODE == 2) {
                a_in[q] = g.ext.A64[(int64_t)r * d + c];
                z_in[q] = Zf[(int64_t)c * d + r];
                if (r == c) diag_in[q][0] = (double)A32[(int64_t)r * d + r];                 // Y_rr
            } else {
                if (r == c) { diag_in[q][0] = A[(int64_t)r * d + r]; diag_in[q][1] = B[(int64_t)r * d + r]; }   // C1_rr, C2_rr
            }
        }
        __syncthreads();
        double* part = smem;                                   // [NW][32][33]
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    part[wave * (32 * 33) + (16 * fa + lk + 4 * reg) * 33 + 16 * fb + li] = acc[fa][fb][reg];
        __syncthreads();
        double vq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * NT, rr = e >> 5, cc = e & 31;
            const double sum = ((part[rr * 33 + cc] + part[(32 * 33) + rr * 33 + cc]) +
                                (part[2 * (32 * 33) + rr * 33 + cc] + part[3 * (32 * 33) + rr * 33 + cc])) +
                               ((part[4 * (32 * 33) + rr * 33 + cc] + part[5 * (32 * 33) + rr * 33 + cc]) +
                                (part[6 * (32 * 33) + rr * 33 + cc] + part[7 * (32 * 33) + rr * 33 + cc]));
            vq[q] = alpha * sum;
        }
        __syncthreads();                                       // every partial has been read: the buffer is free
        double* P = smem;                                      // [32][33]: the values whose |.| row / column sums are wanted
        const int nb = gridDim.x;
        double* rowabs = g.ext.stats;                          // [bj][d]
        double* colabs = g.ext.stats + (int64_t)nb * d;        // [bi][d]
        double v5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if constexpr (MODE == 1) {
            double* scal = g.ext.stats + 2 * (int64_t)nb * d + (int64_t)8 * (ty * nb + tx);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + q * NT, rr = e >> 5, cc = e & 31;
                const int r = row0 + rr, c = col0 + cc;
                const double v = vq[q];
                C[(int64_t)r * d + c] = v;
                P[rr * 33 + cc] = v;
                v5[0] += v * v;
                if (r == c) { v5[2] += v; v5[3] += diag_in[q][0]; v5[4] += diag_in[q][1]; }
            }
            v5[1] = v5[0];      // slot of sum a_ij a_ji (tr A^2, needs the mirror tile): ||A||_F^2 >= tr A^2 stands in -- the two
                                // differ by the non-normal part of A only and either is merely a guess of where the bulk sits
            __syncthreads();
            if (tid < 32) {
                double t = 0.0;
                for (int c = 0; c < 32; ++c) t += fabs(P[tid * 33 + c]);
                rowabs[(int64_t)tx * d + row0 + tid] = t;
            } else if (tid < 64) {
                const int c = tid - 32;
                double t = 0.0;
                for (int rr = 0; rr < 32; ++rr) t += fabs(P[rr * 33 + c]);
                colabs[(int64_t)ty * d + col0 + c] = t;
            }
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v5[q] += __shfl_xor(v5[q], off);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 5; ++q) sred[wave * 5 + q] = v5[q];
            }
            __syncthreads();
            if (tid < 5) {
                double t = 0.0;
                for (int w = 0; w < 8; ++w) t += sred[w * 5 + tid];
                scal[tid] = t;
            }
        } else {
            // R = A/c - G for this tile; the mirror tile of Z sits in z_in.  P holds Z's mirror tile (tx, ty) TRANSPOSED:
            // P[rr][cc] = Z[col0 + cc][row0 + rr], so its row sums are column sums of that tile and vice versa.
            double* scal = g.ext.stats + 2 * (int64_t)nb * d + (int64_t)8 * (ty * nb + tx);
            const double inv = 1.0 / g.ext.st->c;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + q * NT, rr = e >> 5, cc = e & 31;
                const int r = row0 + rr, c = col0 + cc;
                const double R = a_in[q] * inv - vq[q];
                const double z = (double)z_in[q];
                P[rr * 33 + cc] = z;
                v5[0] += z * R;                               // Z_ji R_ij
                v5[1] += R * R;
                if (r == c) v5[2] += diag_in[q][0];
            }
            __syncthreads();
            if (tid < 32) {                                   // column tid of P = row (col0 + tid) of Z, its entries in columns row0..row0+31
                double t = 0.0;
                for (int rr = 0; rr < 32; ++rr) t += fabs(P[rr * 33 + tid]);
                rowabs[(int64_t)ty * d + col0 + tid] = t;     // Z tile (bi = tx, bj = ty): rowabs[bj][row]
            } else if (tid < 64) {                            // row (tid - 32) of P = column (row0 + tid - 32) of Z
                const int rr = tid - 32;
                double t = 0.0;
                for (int c = 0; c < 32; ++c) t += fabs(P[rr * 33 + c]);
                colabs[(int64_t)tx * d + row0 + rr] = t;      // colabs[bi][col]
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v5[q] += __shfl_xor(v5[q], off);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) sred[wave * 5 + q] = v5[q];
            }
            __syncthreads();
            if (tid < 3) {
                double t = 0.0;
                for (int w = 0; w < 8; ++w) t += sred[w * 5 + tid];
                scal[tid] = t;
            }
        }
        return;
    } else if constexpr (KSPLIT) {
        // sum the waves' partial tiles through LDS, then a row-major (coalesced) store of C
        __syncthreads();
        double* part = smem;                                   // [NW][32][33]
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    part[wave * (32 * 33) + (16 * fa + lk + 4 * reg) * 33 + 16 * fb + li] = acc[fa][fb][reg];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 1024 / NT; ++q) {
            const int e = tid + q * NT, rr = e >> 5, cc = e & 31;
            const int r = row0 + rr, c = col0 + cc;
            double sum = (part[rr * 33 + cc] + part[(32 * 33) + rr * 33 + cc]) +
                         (part[2 * (32 * 33) + rr * 33 + cc] + part[3 * (32 * 33) + rr * 33 + cc]);
            if (NW == 8)
                sum += (part[4 * (32 * 33) + rr * 33 + cc] + part[5 * (32 * 33) + rr * 33 + cc]) +
                       (part[6 * (32 * 33) + rr * 33 + cc] + part[7 * (32 * 33) + rr * 33 + cc]);
            if (r < d && c < d) {
                const double v = alpha * sum + (r == c ? beta_eye : 0.0);
                C[(int64_t)r * d + c] = v;
                const double e2 = v - (r == c ? gamma : 0.0);
                ss += e2 * e2;
            }
        }
    } else {
#pragma unroll
        for (int fa = 0; fa < MT; ++fa)
#pragma unroll
            for (int fb = 0; fb < MT; ++fb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = row0 + wr * (BT / 2) + 16 * fa + lk + 4 * reg;
                    const int c = col0 + wc * (BT / 2) + 16 * fb + li;
                    if (r < d && c < d) {
                        const double v = alpha * acc[fa][fb][reg] + (r == c ? beta_eye : 0.0);
                        C[(int64_t)r * d + c] = v;
                        if (g.sym && tx > ty) C[(int64_t)c * d + r] = v;
                        const double e = v - (r == c ? gamma : 0.0);
                        ss += e * e;
                    }
                }
        if (g.sym && tx > ty) ss *= 2.0;         // the mirrored tile's share of the residual
    }
    double* partials = g.partials[zi];
    if (partials) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        __syncthreads();
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (tid == 0) {
            double t = (red[0] + red[1]) + (red[2] + red[3]);
            if (NW == 8) t += (red[4] + red[5]) + (red[6] + red[7]);
            partials[zb * g.pstride + slot] = t;
        }
    }
}

static int pick_bt(int d, int64_t count, int device, bool sym = false) {
    const int64_t t = cdiv(d, 64);
    const int64_t t64 = (sym ? t * (t + 1) / 2 : t * t) * count;            // tiles that do work
    return (t64 >= num_cus(device)) ? 64 : 32;
}

int gemm_f64_slots(int d, int ntypes, int64_t batch, int device) {
    const int64_t t = cdiv(d, pick_bt(d, (int64_t)ntypes * batch, device));
    return (int)(t * t);
}

int gemm_f64_slots_max(int d) { const int64_t t = cdiv(d, 32); return (int)(t * t); }

int gemm_f64_launch(int d, const GemmType* types, int ntypes, int64_t batch, const int* skip, int skip_stride,
                    hipStream_t stream, int device, int partial_stride, const NsCheckArgs* check) {
    if (ntypes < 1 || ntypes > 2 || batch < 1) return set_error(FAD_ERR_INVALID, "gemm: ntypes=%d batch=%lld", ntypes, (long long)batch);
    const bool sym = types[0].sym && (ntypes == 1 || types[1].sym);
    const int bt = pick_bt(d, (int64_t)ntypes * batch, device, sym);
    const int64_t t = cdiv(d, bt);
    const int64_t slots = t * t;
    const int64_t max_b = 65535 / (ntypes + (check ? 1 : 0));
    for (int64_t done = 0; done < batch; done += max_b) {
        const int64_t m = (batch - done < max_b) ? batch - done : max_b;
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        for (int i = 0; i < ntypes; ++i) {
            g.A[i] = types[i].A + done * types[i].sa; g.B[i] = types[i].B + done * types[i].sb;
            g.C[i] = types[i].C + done * types[i].sc;
            g.sa[i] = types[i].sa; g.sb[i] = types[i].sb; g.sc[i] = types[i].sc;
            g.alpha[i] = types[i].alpha; g.beta_eye[i] = types[i].beta_eye; g.gamma[i] = types[i].gamma;
            g.mu[i] = types[i].mu ? types[i].mu + done * types[i].mu_stride : nullptr; g.mu_stride[i] = types[i].mu_stride;
            g.partials[i] = types[i].partials ? types[i].partials + done * (partial_stride > 0 ? partial_stride : slots) : nullptr;
        }
        g.skip = skip ? skip + done * skip_stride : nullptr;
        g.skip_stride = skip_stride; g.ntypes = ntypes;
        static const int env_remap = [] { const char* e = getenv("FAD_GEMM_REMAP"); return e ? atoi(e) : 1; }();
        static const int env_depth = [] { const char* e = getenv("FAD_GEMM_DEPTH"); return e ? atoi(e) : 1; }();
        g.sym = (sym && bt == 64) ? 1 : 0;
        g.tri = (ntypes == 1) ? types[0].b_upper : 0;
        g.remap = g.tri ? 0 : env_remap;            // (the XCD map hands whole column blocks to an XCD: with a triangular B
                                                    //  those blocks cost 1x .. 5x -- plain order mixes them)
        g.pstride = partial_stride > 0 ? partial_stride : (int)slots;
        g.gemm_z = (int)(m * ntypes);
        if (check) {
            g.chk = *check;
            g.chk.st_all += done; g.chk.partials_all += done * check->pstride; g.chk.Yall += done * check->stride;
        }
        dim3 grid((unsigned)t, (unsigned)t, (unsigned)(m * ntypes + (check ? m : 0)));
        const bool full = (d % KB) == 0;
        if (bt == 64) {
            if (full) hipLaunchKernelGGL((gemm_f64_kernel<64, 2, false, true>), grid, dim3(256), 0, stream, d, g);
            else hipLaunchKernelGGL((gemm_f64_kernel<64, 2, false, false>), grid, dim3(256), 0, stream, d, g);
        } else {
            static const int env_w8 = [] { const char* e = getenv("FAD_GEMM_WAVES8"); return e ? atoi(e) : 1; }();
            const bool w8 = env_w8 && full && (int64_t)grid.x * grid.y * (unsigned)(m * ntypes) <= (int64_t)num_cus(device);
            if (w8) hipLaunchKernelGGL((gemm_f64_kernel<32, 1, true, true, 8>), grid, dim3(512), 0, stream, d, g);
            else if (full && env_depth == 1) hipLaunchKernelGGL((gemm_f64_kernel<32, 1, true, true>), grid, dim3(256), 0, stream, d, g);
            else if (full && env_depth == 2) hipLaunchKernelGGL((gemm_f64_kernel<32, 2, true, true>), grid, dim3(256), 0, stream, d, g);
            else if (full) hipLaunchKernelGGL((gemm_f64_kernel<32, 3, true, true>), grid, dim3(256), 0, stream, d, g);
            else hipLaunchKernelGGL((gemm_f64_kernel<32, 3, true, false>), grid, dim3(256), 0, stream, d, g);
        }
    }
    FAD_HIP_TRY(hipGetLastError());
    return (int)slots;
}

int gemm_f64_product_stats_launch(int d, const double* C1, const double* C2, double* A, const int* skip, const NsProductExt& ext,
                                  hipStream_t stream) {
    if (d % KB != 0) return set_error(FAD_ERR_INVALID, "gemm_f64_product_stats: d=%d is not a multiple of %d", d, KB);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = C1; g.B[0] = C2; g.C[0] = A; g.alpha[0] = 1.0; g.ntypes = 1; g.remap = 1; g.gemm_z = 1;
    g.skip = skip; g.skip_stride = 0;
    g.ext = ext;
    const unsigned t = (unsigned)(d / 32);
    hipLaunchKernelGGL((gemm_f64_kernel<32, 1, true, true, 8, double, 1>), dim3(t, t, 2), dim3(512), 0, stream, d, g);
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

int gemm_f64_correction_launch(int d, const float* Y, const float* Y_alt, const int* sel, const int* skip, const NsProductExt& ext,
                               hipStream_t stream) {
    if (d % KB != 0) return set_error(FAD_ERR_INVALID, "gemm_f64_correction: d=%d is not a multiple of %d", d, KB);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.alpha[0] = 1.0; g.ntypes = 1; g.remap = 1; g.gemm_z = 1;
    g.skip = skip; g.skip_stride = 0;
    g.A32 = Y; g.B32 = Y; g.A32_alt = Y_alt; g.B32_alt = Y_alt; g.sel = sel;
    g.ext = ext;
    const unsigned t = (unsigned)(d / 32);
    hipLaunchKernelGGL((gemm_f64_kernel<32, 1, true, true, 8, float, 2>), dim3(t, t, 1), dim3(512), 0, stream, d, g);
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

// (first-use warm-up, common.cpp: warm_code_objects -- loading this translation unit's code object costs ~75 ms at the first launch)
__global__ void code_object_anchor_kernel_gemm_f64() {}
const void* code_object_anchor_gemm_f64() { return reinterpret_cast<const void*>(&code_object_anchor_kernel_gemm_f64); }
}  // namespace fad
