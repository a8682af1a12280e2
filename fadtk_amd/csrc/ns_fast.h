// The square-root chain of ONE Frechet score in eight launches (gfx950): replaces scipy.linalg.sqrtm / eig of fadtk/fad.py:88-92 for
// well-conditioned pairs of dimension 256 / 384 / 512 / 768 / 1024 (frechet.hip: fast_begin; everything else keeps the routes there).
// Batches of problems (per-song scores) run the same arithmetic through ns_fast_big.h (D >= 256: 128 x 128 tiles staged through LDS) and
// ns_fast_res.h (D = 128: a song resident in one workgroup); this file's kernels serve ONE problem, or a handful.
//
//   tr sqrt(A), A = C1 C2:   Newton-Schulz  Y0 = A/c, Z0 = I;  T = (3I - ZY)/2;  Y <- Y T;  Z <- T Z   on LOW-precision operands,
//   then ONE float64-accurate correction   tr sqrt(A/c) = tr Y + 1/2 tr(Z (A/c - Y Y)) + O(err^2)     (SURVEY.md H1).
//
// Round 2 ran the iteration on the f32-input MFMA (157 TFLOP/s peak) and the two products that need float64 accuracy (A = C1 C2 and
// G = Y Y) on the f64 MFMA: twelve launches, ~100 us.  What the profile of that chain and of this file's first version showed
// (profiles/r03b_*): a kernel of this chain costs ~4.2 us before it does anything (an empty 500-workgroup launch takes that long
// between two kernels that leave megabytes dirty in eight non-coherent L2s), and a 512^3 product on 32 x 32 tiles is bound by the
// CU's vector-memory path -- 128-192 KB of operands per workgroup, one cache line per clock -- not by the matrix pipe.  Hence:
//
//   * FEWER launches: the scale of the iteration is derived inside iteration 0 (no ns_prepare), the closing decision is the
//     host's, from partials the correction kernel writes straight into pinned host memory (no finish kernel): eight launches.
//   * FRAGMENT-MAJOR operands: every matrix is stored by its producer the way the MFMA wants it -- for a 32-row block and a k-step,
//     the 64 lanes' 16-byte operand pieces back to back (1 KiB per wave instruction, 8 full cache lines instead of 32-64 partial
//     ones) -- and in BOTH orientations (X as an A operand, X^T as a B operand), so every product is plain coalesced loads
//     straight into registers: no LDS staging, no transpose reads; LDS only sums the k-split partial tiles of the eight waves.
//   * iteration products on split float16: x = hi + lo / 2048 (hi = fp16(x), lo = fp16((x - hi) 2048): 22 significant bits), a
//     product is three v_mfma_f32_32x32x16_f16 terms, hi hi + (hi lo + lo hi) / 2048, accumulated in float32 -- float32-class
//     accuracy at 16x the f32-MFMA rate (scripts/ns_emulate_split.py: iterates and corrected trace indistinguishable from float32).
//   * exact products on the int8 MFMA: both operands on a fixed-point grid of 2^-40 (|x| < 2: covariances are normalised by a
//     power of two, the iterates are O(1) by construction), cut into six balanced base-128 digits; the 30 digit-pair products
//     whose weight matters run on v_mfma_i32_32x32x32_i8 with EXACT int32 accumulation, pairs of equal weight share an
//     accumulator, the eight accumulators are combined in float64: the product of the (2^-41-rounded) operands to ~1e-15.
//
// Launches (D = 512: grid 16 x 16 of 32 x 32 tiles, 512 threads, one workgroup per CU):
//   K1 nsf_prepare        packed moments -> mu x2, tr Sigma, power-of-two scales, digit planes of both Sigma
//   K2 nsf_i8<A>          A = C1 C2 exact: A (float64), its split planes, per-tile statistics (sum a^2, trace, largest row / column
//                         sum of |a|); a spare workgroup forms the mean term ||mu1 - mu2||^2 (the reference's dtype quirk included)
//   K3 nsf_split<FIRST>   every workgroup derives the scale c from K2's statistics, then Y1 = Y0 T0 = 1.5 Y0 - 0.5 Y0^2 from the
//                         product A A, and Z1 = T0 (Z0 = I)
//   K4/K6 nsf_split<T>    T = (3I - Z Y)/2 + the residual partials
//   K5/K7 nsf_split<U>    Y <- Y T, Z <- T Z (two products, one launch) + the convergence check as an extra workgroup + the digit
//                         planes of the new Y
//   K8 nsf_i8<G>          G = Y Y exact on the final iterate; R = A/c - G never leaves the workgroup: tr(Z R), ||R||^2, tr Y and the
//                         |Z| bounds go to pinned host memory, where frechet.hip (fast_decide) accepts or rejects the result
#pragma once
#include "fad_common.h"
#include "ns_check.h"
#include "ns32.h"
#include "ns_mean.h"

namespace fad {
namespace nsf {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kDigits = 6;             // balanced base-128 digits per value
// digit pairs (p, q) with p + q >= UMIN are multiplied (weight 2^(7 (p + q) - 80)); pairs of equal p + q share an accumulator.
// UMIN = 3 (30 pairs) reproduces the product of the grid values to ~1e-15 (A = C1 C2, whose scale c may be small); UMIN = 4 (26
// pairs) to ~1e-12 -- enough for G = Y Y, whose entries are O(1) (scripts/ns_emulate_split.py)
constexpr int kUminA = 3, kUminG = 4;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;
constexpr int kTileStats = 4;          // doubles per tile in the statistics K2 / K8 leave

// ---- fragment-major layouts (indices in units of 16 bytes).  lane = 32 g + r.
// split planes of a matrix W (A layout): piece (rb, ks, plane, lane) = the 8 halves W[32 rb + r][16 ks + 8 g + 0..7] of plane
// (0 = hi, 1 = scaled lo); a wave's operand of k-step ks is the 64 consecutive pieces of (rb, ks, plane)
__device__ __host__ __forceinline__ size_t fa_idx(int rb, int ks, int plane, int lane, int d) {
    return (((size_t)rb * (size_t)(d >> 4) + (size_t)ks) * 2 + (size_t)plane) * 64 + (size_t)lane;
}
// digit planes of W: piece (rb, ks, p, lane) = digit p of the 16 values W[32 rb + r][32 ks + 16 g + 0..15]
__device__ __host__ __forceinline__ size_t dg_idx(int rb, int ks, int p, int lane, int d) {
    return (((size_t)rb * (size_t)(d >> 5) + (size_t)ks) * kDigits + (size_t)p) * 64 + (size_t)lane;
}
// where element (row, k) of W lives: 16-byte piece and the half / byte inside it
__device__ __host__ __forceinline__ size_t fa_elem(int row, int k, int plane, int d, int& half) {
    half = k & 7;
    return fa_idx(row >> 5, k >> 4, plane, 32 * ((k >> 3) & 1) + (row & 31), d);
}
__device__ __host__ __forceinline__ size_t dg_elem(int row, int k, int p, int d, int& byte) {
    byte = k & 15;
    return dg_idx(row >> 5, k >> 5, p, 32 * ((k >> 4) & 1) + (row & 31), d);
}

// value = sum_p dg[p] 128^p 2^-40 (+ less than 2^-41), |value| <= 2; every step is exact in the type of v
template <typename F> __device__ __forceinline__ void digits_of(F v, int (&dg)[kDigits]) {
    F t = v * (F)32;
#pragma unroll
    for (int p = kDigits - 1; p >= 0; --p) {
        const F r = (sizeof(F) == 4) ? (F)__builtin_rintf((float)t) : (F)__builtin_rint((double)t);
        dg[p] = (int)r;
        t = (t - r) * (F)128;
    }
}

__device__ __forceinline__ void split16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * kLoScale);
}
__device__ __forceinline__ float used16(_Float16 hi, _Float16 lo) { return (float)hi + (float)lo * kLoInv; }

// sums / maxima over a workgroup of NW waves (8: 512 threads); red = NW doubles of LDS per value
template <int N, int NW = 8> __device__ __forceinline__ void wg8_sum(double (&v)[N], double* red) {
#pragma unroll
    for (int q = 0; q < N; ++q)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) red[(threadIdx.x >> 6) * N + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w * N + q];
        v[q] = t;
    }
}
template <int NW = 8> __device__ __forceinline__ double wg8_max(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = fmax(t, red[w]);
    return t;
}

// XCD-aware tile map (as gemm_f32.hip / gemm_f64.hip): workgroup b runs on XCD b % 8; XCD (ex, ey) of a 2 x 4 grid owns a
// contiguous block of output tiles, so each private L2 fetches 1/2 of the A rows and 1/4 of the B columns once.
__device__ __forceinline__ void tile_of_block(int& ty, int& tx) {
    ty = blockIdx.y; tx = blockIdx.x;
    const int t = gridDim.x;
    if ((t & 3) == 0) {
        const int b = blockIdx.y * t + blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        const int R = t >> 1, Cc = t >> 2;
        ty = (xcd >> 2) * R + idx / Cc;
        tx = (xcd & 3) * Cc + idx % Cc;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Per-matrix header (device): what K1 finds out about one covariance.  `gen` is the caller's per-call token: a flag word that
// EQUALS it was raised during this call (no reset pass between calls).
struct MatHdr {
    double s;           // power-of-two scale: |s Sigma| <= 1
    double tr;          // tr Sigma (caller's units)
    int bad;            // non-finite / non-positive diagonal, or its set has < 2 rows
    int flag_gen;       // == gen: some element is not finite or does not fit the fixed-point grid
    int pad[2];
};
__device__ __forceinline__ bool hdr_bad(const MatHdr* a, const MatHdr* b, int gen) {
    return (a->bad | b->bad) != 0 || a->flag_gen == gen || b->flag_gen == gen;
}
__device__ __forceinline__ double hdr_inv_s12(const MatHdr* a, const MatHdr* b) { return 1.0 / (a->s * b->s); }      // powers of two: exact

// BATCHES (per-song scores against one baseline, frechet.hip: fast_songs): problem b = blockIdx.z / (z-slices of the kernel); every
// per-problem buffer of problem b lives `pstride` bytes behind problem 0's (host-side buffers: `hstride`); the baseline's digit
// planes and header (the A side of A = Sigma_b Sigma_s) are shared.  A single pair is a batch of one with pstride = 0.
template <class T> __device__ __forceinline__ T* adv(T* p, int64_t bytes) {
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes);
}

// One matrix in split-float16 form: fragment-major planes of X (as an A operand) and of X^T (as a B operand), d * d / 4 pieces each.
struct SplitMat { uint4* a; uint4* at; };
__device__ __forceinline__ SplitMat adv(const SplitMat& m, int64_t bytes) { return SplitMat{adv(m.a, bytes), adv(m.at, bytes)}; }

// ------------------------------------------------------------------------------------------------------------------------------
// K1: means, mean term, traces, scales and digit planes of both covariances from the packed moments (or the caller's matrices).
// grid (d * d / 2048 + 1, 2), 512 threads: a thread owns 4 consecutive columns of one row = one dword of every digit plane;
// the extra workgroup of set 0 forms both means and the mean term ||mu1 - mu2||^2 (the reference's dtype quirk included).
constexpr int kPrepMaxSets = 64;      // two sets for each of the (at most 32) pairs of a batch: frechet_internal.h kMaxMultiPairs
struct PrepArgs {
    const double* acc[2];            // packed moments [n | sum | sum xxT], or nullptr: the caller's (mu, Sigma) are used as they are
    const double* cov_in[2];         // ... then these (device)
    const double* mu_in[2];
    int d, ddof, gen, mean_dtype;
    double* mus;                     // [2][d]          (written only with acc)
    double* covs;                    // [2][d * d]      (written only with acc: what the float64 route reads if it has to take over)
    uint4* dig[2];                   // digit planes of s_1 Sigma_1 (A operand) and of (s_2 Sigma_2)^T (B operand)
    NsState* st; MatHdr* hdr[2];
    // batch = 1 (per-song scores): blockIdx.y = 0 is the baseline (cov_in[0], dig[0], hdr[0]); blockIdx.y = 1 + b is song b:
    // cov_in[1] + b d^2, and dig[1], hdr[1], st advanced by b * pstride bytes.  No means, no mean term, no spare workgroup.
    // batch = 2 (B independent PAIRS from packed moments, frechet.hip: pairs_begin): blockIdx.y = 2 b + set, packed moments
    // accs[2 b + set]; EVERYTHING (mus, covs, dig, hdr, st) of pair b lives b * pstride bytes behind pair 0's; means, mean term and
    // the spare workgroup as for a single pair.
    int batch; int64_t pstride;
    const double* accs[kPrepMaxSets];
    // numpy's float32 running column sums of a set (fad_moments_set_reference_mean), or nullptr: then its mean is float32(float64(run) / n)
    // -- what np.mean returns before its final cast (fad.py:48) -- instead of the exact sum / n
    const float* run[2]; const float* runs[kPrepMaxSets];
    // batch = 2: leave `covs` unwritten (a third of the launch's bytes; the float64 route of a batch fills them with nsf_pairs_covs if
    // it has to take pairs over -- frechet.hip: fad_frechet_multi_end)
    int no_covs;
    // runs of 2048 elements per workgroup (0 / 1: one; the grid's x is d * d / (2048 per) [+ 1 for the spare workgroup])
    int per;
};

// Sigma[row][k] from the packed moments: the expression of moments_finalize_kernel up to the reciprocals (symmetric bit for bit)
__device__ __forceinline__ double prep_cov(double m, double sr, double sk, double inv_n, double inv_nd) { return (m - (sr * sk) * inv_n) * inv_nd; }

__global__ __launch_bounds__(512) void nsf_prepare(PrepArgs a) {
    __shared__ double red[8 * 2];
    __shared__ double mu_lds[2 * 1024];
    __shared__ float gaps[1024];
    const int tid = threadIdx.x, d = a.d;
    const bool pairs = a.batch == 2, songs = a.batch == 1;
    const int set = songs ? (blockIdx.y ? 1 : 0) : (pairs ? (int)(blockIdx.y & 1) : (int)blockIdx.y);
    // byte offset of this problem's buffers
    const int64_t po = songs ? (blockIdx.y ? (int64_t)(blockIdx.y - 1) * a.pstride : 0) : (pairs ? (int64_t)(blockIdx.y >> 1) * a.pstride : 0);
    if (!songs && blockIdx.x == gridDim.x - 1) {
        // the spare workgroup: means of both sets -> global (with acc) and LDS, then the mean term -> state
        if (set != 0 || tid >= 256) return;
        double* mus = adv(a.mus, po);
        for (int q = 0; q < 2; ++q) {
            const double* acq = pairs ? a.accs[(blockIdx.y & ~1u) + q] : a.acc[q];
            const float* rq = pairs ? a.runs[(blockIdx.y & ~1u) + q] : a.run[q];
            for (int i = tid; i < d; i += 256) {
                const double m = acq ? ((rq ? numpy_mean_of_f32_sum(rq[i], acq[0]) : acq[1 + i] / acq[0])) : a.mu_in[q][i];   // (as finalize_for_frechet)
                mu_lds[q * 1024 + i] = m;
                if (acq) mus[(int64_t)q * d + i] = m;
            }
        }
        __syncthreads();
        const double mt = mean_term_block(mu_lds, mu_lds + 1024, d, a.mean_dtype, gaps, red);
        if (tid == 0) adv(a.st, po)->mean_term = mt;
        return;
    }
    const double* acc = pairs ? a.accs[blockIdx.y] : a.acc[set];
    const double n = acc ? acc[0] : 2.0;
    const double* sum = acc ? acc + 1 : nullptr;
    const double* M = acc ? acc + 1 + d : a.cov_in[set] + ((songs && blockIdx.y) ? (int64_t)(blockIdx.y - 1) * d * d : 0);
    MatHdr* hdr = adv(a.hdr[set], po);
    NsState* st = adv(a.st, po);
    const double inv_n = 1.0 / n, inv_nd = 1.0 / (n - (double)a.ddof);
    // this thread's dword: quarter qd of the piece of lane (r, g), k-step ks, row block rb -- consecutive threads write
    // consecutive dwords of consecutive pieces.  A workgroup takes `per` runs of 2048 elements (a.per; the pairs of a batch: each
    // workgroup finds the scale for itself -- 512 strided diagonal reads and two reductions -- which for ONE run is more L2 traffic than the run).
    const int per = a.per > 1 ? a.per : 1;
    int T = blockIdx.x * per * 512 + tid;
    int qd = T & 3, piece = T >> 2;
    int r = piece & 31, g = (piece >> 5) & 1, ksrb = piece >> 6, ks = ksrb % (d >> 5), rb = ksrb / (d >> 5);
    int row = 32 * rb + r, k0 = 32 * ks + 16 * g + 4 * qd;
    // its (first run's) elements are requested first; the scale is found while they travel.  (The packed moments start at an odd
    // double: 8-byte loads.)
    double m[4], sk[4], sr = 0.0;
    // the B operand is Sigma_2^T: the moments give a bit-for-bit symmetric matrix (read it row-wise), a CALLER's Sigma_2 need
    // not be (fad_frechet on host arrays): its planes are filled from the transposed read, so that the product is Sigma_1 Sigma_2
    // on every route (the float64 routes and the reference form sigma1.dot(sigma2), fad.py:88)
    const bool transposed = !acc && set == 1;
    auto fetch = [&]() {
        const double* Mrow = M + (int64_t)row * d + k0;
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = transposed ? M[(int64_t)(k0 + q) * d + row] : Mrow[q];
        if (acc) {
            sr = sum[row];
#pragma unroll
            for (int q = 0; q < 4; ++q) sk[q] = sum[k0 + q];
        }
    };
    fetch();
    // every workgroup finds the scale itself: trace and largest diagonal entry (a NaN / Inf on the diagonal shows in the trace)
    double tr1[1] = {0.0}, mx = 0.0;
    for (int i = tid; i < d; i += 512) {
        const double md = M[(int64_t)i * d + i];
        const double v = acc ? (md - (sum[i] * sum[i]) * inv_n) * inv_nd : md;
        tr1[0] += v; mx = fmax(mx, v);
    }
    mx = wg8_max(mx, red);
    wg8_sum<1>(tr1, red);
    const double tr = tr1[0];
    const bool few = acc && n < 2.0;
    const bool bad = few || !(tr == tr) || isinf(tr) || !(mx > 0.0) || isinf(mx);
    int ex = 0;
    if (!bad) (void)frexp(mx, &ex);                      // mx = m 2^ex, m in [0.5, 1)
    const double s = bad ? 1.0 : ldexp(1.0, -ex);
    if (blockIdx.x == 0 && tid == 0) {
        hdr->s = s; hdr->tr = tr; hdr->bad = bad ? 1 : 0;
        if (!songs) st->too_few[set] = few ? 1 : 0;
        if (songs ? set == 1 : set == 0) {               // the per-call reset of the iteration state (finalize_for_frechet did this)
            st->done = 0; st->finished = 0; st->nonfinite = 0; st->conv = 0; st->final_iter = -1;
            st->upd_skip[0] = 0; st->upd_skip[1] = 0;
            if (songs) { st->too_few[0] = 0; st->too_few[1] = 0; st->mean_term = 0.0; }
        }
    }
    bool off_grid = false;
    uint4* out = adv(a.dig[set], po);
    for (int it = 0; it < per; ++it) {
        if (it > 0) {
            T += 512;
            qd = T & 3; piece = T >> 2;
            r = piece & 31; g = (piece >> 5) & 1; ksrb = piece >> 6; ks = ksrb % (d >> 5); rb = ksrb / (d >> 5);
            row = 32 * rb + r; k0 = 32 * ks + 16 * g + 4 * qd;
            fetch();
        }
        uint32_t w[kDigits];
#pragma unroll
        for (int p = 0; p < kDigits; ++p) w[p] = 0u;
        double* cov_out = acc ? adv(a.covs, po) + (int64_t)set * d * d + (int64_t)row * d + k0 : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double c0 = acc ? prep_cov(m[q], sr, sk[q], inv_n, inv_nd) : m[q];
            if (acc && !a.no_covs) cov_out[q] = c0;
            const double v = c0 * s;
            const bool fits = fabs(v) <= 1.9375;             // false for NaN / Inf, or |Sigma_ij| > max diagonal (not a covariance)
            off_grid = off_grid || !fits;
            int dg[kDigits];
            digits_of<double>((bad || !fits) ? 0.0 : v, dg);
#pragma unroll
            for (int p = 0; p < kDigits; ++p) w[p] |= ((uint32_t)dg[p] & 0xffu) << (8 * q);
        }
#pragma unroll
        for (int p = 0; p < kDigits; ++p) reinterpret_cast<uint32_t*>(out + dg_idx(rb, ks, p, 32 * g + r, d))[qd] = w[p];
    }
    if (off_grid) hdr->flag_gen = a.gen;                 // (every raiser writes the same value)
}

// The covariances nsf_prepare left out (no_covs) for the pairs of a batch, element for element what it would have written.
// grid (d * d / 2048, 2 B), 512 threads: a thread owns 4 consecutive columns of one row.
__global__ __launch_bounds__(512) void nsf_pairs_covs(PrepArgs a) {
    const int d = a.d, set = (int)(blockIdx.y & 1);
    const int64_t po = (int64_t)(blockIdx.y >> 1) * a.pstride;
    const double* acc = a.accs[blockIdx.y];
    const double n = acc[0];
    const double* sum = acc + 1;
    const double* M = acc + 1 + d;
    const double inv_n = 1.0 / n, inv_nd = 1.0 / (n - (double)a.ddof);
    const int64_t e = ((int64_t)blockIdx.x * 512 + threadIdx.x) * 4;
    const int row = (int)(e / d), k0 = (int)(e - (int64_t)row * d);
    const double sr = sum[row];
    double* cov_out = adv(a.covs, po) + (int64_t)set * d * d + (int64_t)row * d + k0;
#pragma unroll
    for (int q = 0; q < 4; ++q) cov_out[q] = prep_cov(M[(int64_t)row * d + k0 + q], sr, sum[k0 + q], inv_n, inv_nd);
}

// ------------------------------------------------------------------------------------------------------------------------------
// the tile in `fin` (float [32][33]) -> split planes of X and X^T (16-byte pieces), and optionally the digit planes of both.
// 512 threads; (ty, tx) = the tile's block row / block column.
// task t in [0, 256) of the split planes, task t in [0, 512) of the digit planes: one thread each in the 512-thread kernels, a few
// passes of one wave in the batched kernel (ns_fast_big.h)
__device__ __forceinline__ void store_tile_planes(const float* fin, const SplitMat& X, int ty, int tx, int d, int tid) {
    {
        // piece q (8 k) of line `ln`: tid < 128: X, line = row, k = columns; else X^T, line = column, k = rows
        const bool tr = tid >= 128;
        const int ln = tid & 31, q = (tid >> 5) & 3;
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tr ? fin[(8 * q + j) * 33 + ln] : fin[ln * 33 + 8 * q + j];
            _Float16 h, l; split16(v, h, l); hi[j] = h; lo[j] = l;
        }
        uint4* base = tr ? X.at : X.a;
        const int rb = tr ? tx : ty, ks = 2 * (tr ? ty : tx) + (q >> 1), lane = 32 * (q & 1) + ln;
        uint4 uh, ul; __builtin_memcpy(&uh, &hi, 16); __builtin_memcpy(&ul, &lo, 16);
        base[fa_idx(rb, ks, 0, lane, d)] = uh;
        base[fa_idx(rb, ks, 1, lane, d)] = ul;
    }
}
__device__ __forceinline__ void store_tile_digits(const float* fin, uint4* dig, uint4* dig_t, int ty, int tx, int d, int tid) {
    {
        // four consecutive k per thread = one dword per digit.  threads 256..511: X; threads 0..255: X^T
        const bool tr = tid < 256;
        const int u = tid & 255, ln = u & 31, q = u >> 5;            // dword q (0..7) of line ln: k = 4 q .. 4 q + 3
        uint32_t w[kDigits];
#pragma unroll
        for (int p = 0; p < kDigits; ++p) w[p] = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = tr ? fin[(4 * q + j) * 33 + ln] : fin[ln * 33 + 4 * q + j];
            _Float16 h, l; split16(v, h, l);
            int dg[kDigits];
            digits_of<float>(used16(h, l), dg);              // the value the MFMAs of the iteration see
#pragma unroll
            for (int p = 0; p < kDigits; ++p) w[p] |= ((uint32_t)dg[p] & 0xffu) << (8 * j);
        }
        uint4* base = tr ? dig_t : dig;
        const int rb = tr ? tx : ty, ks = tr ? ty : tx, lane = 32 * (q >> 2) + ln;
#pragma unroll
        for (int p = 0; p < kDigits; ++p)
            reinterpret_cast<uint32_t*>(base + dg_idx(rb, ks, p, lane, d))[q & 3] = w[p];
    }
}
__device__ __forceinline__ void store_tile(const float* fin, const SplitMat& X, uint4* dig, uint4* dig_t, int ty, int tx, int d, int tid) {
    if (tid < 256) store_tile_planes(fin, X, ty, tx, d, tid);
    if (dig) store_tile_digits(fin, dig, dig_t, ty, tx, d, tid);
}

// per-tile statistics record (K2: of A; K8: of the correction) -- kTileStats doubles per tile, tile index ty * nb + tx
//   K2: sum a^2 | trace share | largest row sum of |a| in the tile | largest column sum
//   K8: tr(Z R) share | sum R^2 | tr Y share | (unused);  then [nb * nb][2]: largest partial row / column sums of |Z| per tile

// ------------------------------------------------------------------------------------------------------------------------------
// K2 / K8: exact product of two digit-plane matrices.  Workgroup tile 32 x 32, 512 threads; wave w owns the k-steps
// [w NS8, (w + 1) NS8) of 32 k each (d = 256 NS8).
enum { I8_A = 0, I8_G = 1 };
constexpr int kHostWords = 16, kHostVals = 24;   // what K8 snapshots for the host besides the per-tile partials
struct I8Args {
    const uint4* Adig; const uint4* Bdig;        // A operand rows / B operand COLUMNS (= rows of B^T), digit planes
    const uint4* Adig_alt; const uint4* Bdig_alt; const int* sel;      // I8_G: used instead when *sel is odd (ping-pong iterates)
    int d, gen;
    const MatHdr* hA; const MatHdr* hB;          // headers of the two covariances (hA: the shared baseline in a batch of songs)
    int64_t pstride, hstride;                    // bytes between consecutive problems of a batch: device buffers / host buffers
    int64_t astride;                             // ... of the A SIDE (hA; I8_A: Adig): 0 = shared baseline (songs), pstride for a batch of pairs
    const int* skip;                             // *skip != 0: nothing to do
    double* stats;                               // I8_A: [nb * nb][4] (device);  I8_G: [nb * nb][4] then [nb * nb][2] (PINNED HOST memory)
    // I8_A
    double* A64; SplitMat P;
    NsState* st;
    // I8_G
    const double* A64in;
    SplitMat Y[2], Z[2];
    const Ns32State* s32;
    int* host_words;                             // pinned host: snapshot of the state words the host decides from (kHostWords ints)
    double* host_vals;                           // ... and doubles (kHostVals)
    SplitMat Rv;                                 // I8_G, or {nullptr}: split planes of kVerScale R (both orientations) for the verification products
    int scaled;                                  // I8_G: the chain ran with SplitArgs::scaled (host word 14 then tells whether THIS problem took scaled steps)
};
// The verification of a correction the norm bound cannot vouch for (frechet.hip: fast_decide_one; ill-conditioned products, where
// ||Z||^3 ||R||^2 overestimates the neglected terms 10^5..10^8 times): with P = Z R and E = I - Z Y,
//     tr sqrt(A/c) = tr Y + 1/2 tr(Z R) + 1/2 tr(E P) - [second order]  + O(||E||^2 ||P||),   [second order] ~ 1/8 tr(Z P P)
// (Y^-1 = (I - E)^-1 Z; the second-order term of the root at Y^2 in the direction R is sum_ij R_ij R_ji / (2 y_i y_j (y_i + y_j)) in Y's
// eigenbasis, which 1/8 tr(Z P P) = sum_ij R_ij R_ji (y_i + y_j) / (8 y_i^2 y_j^2) overestimates by (y_i + y_j)^2 / (4 y_i y_j) >= 1
// for the symmetrisable products at hand).  Two launches on the split-float16 kernels: SP_V2 forms P' = Z R' (R' = kVerScale R, so
// that float16 holds it) and E' = kVerScale (I - Z Y) side by side, SP_V3 forms Q' = Z P' and leaves per tile
//     sum Q'_ij P'_ji | sum E'_ij P'_ji | sum P'_ij^2 | sum E'_ij^2        (kVerStats doubles, pinned host memory)
// -- estimates and a small correction: float32-class products are ample.  scripts/ns_emulate_verify.py emulates the whole rule.
constexpr float kVerScale = 4096.f;
constexpr int kVerStats = 4;

template <int NS8, int MODE>
__global__ __launch_bounds__(512) void nsf_i8(I8Args g) {
    // LDS: 21 KB.  (Round 3's first version kept all eight waves' partial tiles side by side: 72 KB -- and a workgroup of this
    // chain could not share a CU with TWO workgroups of the moments tile kernel (2 x 64 KB of the 160): with one HIP stream per
    // score in flight every launch of the chain took one of that kernel's two slots on every CU.  Now the partial tiles meet in a
    // tree through two slots.)
    __shared__ __attribute__((aligned(16))) double part[2 * 32 * 33];
    __shared__ float fin[32 * 33];
    __shared__ float finR[(MODE == I8_G) ? 32 * 33 : 1];      // I8_G: kVerScale R of the tile, for the verification products (g.Rv)
    __shared__ double red[8 * 4];
    constexpr int kUmin = (MODE == I8_A) ? kUminA : kUminG;
    constexpr int kGroups = 2 * (kDigits - 1) - kUmin + 1;
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    const int64_t po = (int64_t)blockIdx.z * g.pstride, ho = (int64_t)blockIdx.z * g.hstride;      // this problem's buffers
    const MatHdr* hB = adv(g.hB, po);
    const MatHdr* hA = adv(g.hA, (int64_t)blockIdx.z * g.astride);
    NsState* const st_p = adv(g.st, po);
    const bool bad = hdr_bad(hA, hB, g.gen);
    const bool skipped = bad || (g.skip && *adv(g.skip, po) != 0);
    if constexpr (MODE == I8_G) {
        // whatever happens, the host finds the state of the iteration next to the partials
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            const NsState* st = st_p; const Ns32State* s = adv(g.s32, po);
            int* hw = adv(g.host_words, ho); double* hv = adv(g.host_vals, ho);
            hw[0] = bad ? 1 : 0; hw[1] = st->done; hw[2] = st->nonfinite; hw[3] = st->too_few[0]; hw[4] = st->too_few[1];
            hw[5] = s->ok; hw[6] = s->failed; hw[7] = s->final_iter; hw[8] = s->decided_at; hw[9] = s->strict; hw[10] = s->finished;
            hw[11] = skipped ? 1 : 0;
            hv[0] = st->c; hv[1] = st->tr1; hv[2] = st->tr2; hv[3] = st->mean_term;
#pragma unroll
            for (int q = 0; q < 16; ++q) hv[4 + q] = s->res[q];
            hw[14] = (g.scaled && st->mu[0] != 1.0) ? 1 : 0;
            hw[12] = g.gen;                                          // the snapshot belongs to this score
        }
    }
    if (skipped) return;
    int ty, tx; tile_of_block(ty, tx);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, r = lane & 31;
    const int row0 = ty * 32, col0 = tx * 32;
    const bool alt = (MODE == I8_G) && g.sel && (*adv(g.sel, po) & 1);
    // (A = Sigma_b Sigma_s: the A operand is the batch's shared baseline; G = Y Y: both operands are the problem's own)
    const uint4* Ad = (MODE == I8_A) ? adv(g.Adig, (int64_t)blockIdx.z * g.astride) : adv(alt ? g.Adig_alt : g.Adig, po);
    const uint4* Bd = adv(alt ? g.Bdig_alt : g.Bdig, po);
    const bool idle = wave * NS8 >= (d >> 5);            // d = 128: four k-steps, waves 4..7 contribute zeros

    // Operand pieces are requested most significant digit first, A and B alternating, and the MFMAs follow in the order the
    // pieces land: as soon as digit m of both operands is there, every pair with min(p, q) = m can go (the compiler places the
    // counted waits) -- the matrix pipe starts after two pieces instead of twelve.  D <= 512: all pieces of the wave's k-steps are
    // requested up front; above, the next k-step is requested while the current one is multiplied.
    constexpr int NBUF = (NS8 <= 2) ? NS8 : 2;
    i32x4 a[NBUF][kDigits], b[NBUF][kDigits];
    auto fetch = [&](int buf, int s) {
        const i32x4* pa = reinterpret_cast<const i32x4*>(Ad + dg_idx(ty, wave * NS8 + s, 0, lane, d));
        const i32x4* pb = reinterpret_cast<const i32x4*>(Bd + dg_idx(tx, wave * NS8 + s, 0, lane, d));
#pragma unroll
        for (int m = kDigits - 1; m >= 0; --m) { a[buf][m] = pa[64 * m]; b[buf][m] = pb[64 * m]; }
    };
    i32x16 acc[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][q] = 0;
    auto mma = [&](int buf) {
#pragma unroll
        for (int m = kDigits - 1; m >= 0; --m) {
#pragma unroll
            for (int q = kDigits - 1; q >= m; --q)
                if (m + q >= kUmin) acc[m + q - kUmin] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[buf][m], b[buf][q], acc[m + q - kUmin], 0, 0, 0);
#pragma unroll
            for (int p = kDigits - 1; p > m; --p)
                if (p + m >= kUmin) acc[p + m - kUmin] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[buf][p], b[buf][m], acc[p + m - kUmin], 0, 0, 0);
        }
    };
    if (idle) {
        // nothing to multiply: this wave's share of the tile is zero
    } else if constexpr (NS8 <= 2) {
#pragma unroll
        for (int s = 0; s < NS8; ++s) fetch(s, s);
#pragma unroll
        for (int s = 0; s < NS8; ++s) mma(s);
    } else {
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < NS8; ++s) {
            if (s + 1 < NS8) fetch((s + 1) & 1, s + 1);
            mma(s & 1);
        }
    }
    // this wave's share of the tile in float64: sum_u acc_u 2^(7 u - 80)
    double gp[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gp[q] = 0.0;
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const double wgt = __builtin_ldexp(1.0, 7 * (u + kUmin) - 80);
#pragma unroll
        for (int q = 0; q < 16; ++q) gp[q] = __builtin_fma((double)acc[u][q], wgt, gp[q]);
    }
    // the eight shares meet in a tree: waves 6, 7 hand theirs to 4, 5; those to 2, 3; those to 0, 1; whose two sums are added below
    {
        auto put = [&](int slot) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) part[slot * (32 * 33) + ((reg & 3) + 8 * (reg >> 2) + 4 * kg) * 33 + r] = gp[reg];
        };
        auto add = [&](int slot) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gp[reg] += part[slot * (32 * 33) + ((reg & 3) + 8 * (reg >> 2) + 4 * kg) * 33 + r];
        };
        if (wave >= 6) put(wave - 6);
        __syncthreads();
        if (wave == 4 || wave == 5) add(wave - 4);
        __syncthreads();
        if (wave == 4 || wave == 5) put(wave - 4);
        __syncthreads();
        if (wave == 2 || wave == 3) add(wave - 2);
        __syncthreads();
        if (wave == 2 || wave == 3) put(wave - 2);
        __syncthreads();
        if (wave < 2) add(wave);
        __syncthreads();
        if (wave < 2) put(wave);
        __syncthreads();
    }
    // thread -> elements (rr, 2 cp), (rr, 2 cp + 1)
    const int rr = tid >> 4, cp = tid & 15;
    double G2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = rr * 33 + 2 * cp + q;
        G2[q] = part[o] + part[(32 * 33) + o];
    }
    const int nb = gridDim.x;
    const int gr = row0 + rr, gc = col0 + 2 * cp;
    double* const stats = adv(g.stats, (MODE == I8_A) ? po : ho);
    double* scal = stats + (size_t)kTileStats * (ty * nb + tx);
    double m = 0.0, v3[3] = {0.0, 0.0, 0.0};
    if constexpr (MODE == I8_A) {
        // A in the caller's units (float64), the split planes of its normalised image P = s1 s2 A, the tile's statistics
        const double inv = hdr_inv_s12(hA, hB);
        *reinterpret_cast<double2*>(adv(g.A64, po) + (int64_t)gr * d + gc) = make_double2(G2[0] * inv, G2[1] * inv);
        fin[rr * 33 + 2 * cp] = (float)G2[0]; fin[rr * 33 + 2 * cp + 1] = (float)G2[1];
        v3[0] = G2[0] * G2[0] + G2[1] * G2[1];
        v3[1] = (gr == gc ? G2[0] : 0.0) + (gr == gc + 1 ? G2[1] : 0.0);
        __syncthreads();
        store_tile(fin, adv(g.P, po), nullptr, nullptr, ty, tx, d, tid);
        if (tid < 32) { for (int c = 0; c < 32; ++c) m += fabsf(fin[tid * 33 + c]); }                  // row sum of |a|
        else if (tid < 64) { for (int q = 0; q < 32; ++q) m += fabsf(fin[q * 33 + tid - 32]); }        // column sum
    } else {
        // R = A/c - G for this tile; Z enters through its mirror tile: Z^T[gr][gc] = Z[gc][gr] pairs with R[gr][gc] in tr(Z R)
        const SplitMat Zm = adv(g.Z[alt ? 1 : 0], po);
        const SplitMat Ym = adv(g.Y[alt ? 1 : 0], po);
        const double inv_c = 1.0 / st_p->c;
        const double2 a2 = *reinterpret_cast<const double2*>(adv(g.A64in, po) + (int64_t)gr * d + gc);
        int half;
        const size_t zi0 = fa_elem(gr, gc, 0, d, half);               // Z^T in the A layout: row gr, k = gc (even); gc + 1 sits in the same piece
        const f16x2 zh = *reinterpret_cast<const f16x2*>(reinterpret_cast<const _Float16*>(Zm.at + zi0) + half);
        const f16x2 zl = *reinterpret_cast<const f16x2*>(reinterpret_cast<const _Float16*>(Zm.at + zi0 + 64) + half);
        const double z0 = (double)used16(zh[0], zl[0]), z1 = (double)used16(zh[1], zl[1]);
        const double R0 = a2.x * inv_c - G2[0], R1 = a2.y * inv_c - G2[1];
        v3[0] = z0 * R0 + z1 * R1; v3[1] = R0 * R0 + R1 * R1;
        if (gr == gc || gr == gc + 1) {
            int hy;
            const size_t yi = fa_elem(gr, gr, 0, d, hy);
            v3[2] = (double)used16(reinterpret_cast<const _Float16*>(Ym.a + yi)[hy], reinterpret_cast<const _Float16*>(Ym.a + yi + 64)[hy]);
        }
        fin[rr * 33 + 2 * cp] = fabsf((float)z0); fin[rr * 33 + 2 * cp + 1] = fabsf((float)z1);      // |Z[col0 + c][row0 + r]| at (r, c)
        finR[rr * 33 + 2 * cp] = (float)(R0 * (double)kVerScale); finR[rr * 33 + 2 * cp + 1] = (float)(R1 * (double)kVerScale);
        __syncthreads();
        if (g.Rv.a && tid < 256) store_tile_planes(finR, adv(g.Rv, po), ty, tx, d, tid);
        if (tid < 32) { for (int q = 0; q < 32; ++q) m += fin[q * 33 + tid]; }                         // fixed c: part of ROW col0 + c of Z
        else if (tid < 64) { for (int c = 0; c < 32; ++c) m += fin[(tid - 32) * 33 + c]; }             // fixed r: part of COLUMN row0 + r of Z
    }
    // lanes 0..31 of wave 0 hold 32 row sums, lanes 32..63 the 32 column sums: the tile's largest of each
    double mrow = (tid < 32) ? m : 0.0, mcol = (tid >= 32 && tid < 64) ? m : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mrow = fmax(mrow, __shfl_xor(mrow, off)); mcol = fmax(mcol, __shfl_xor(mcol, off)); }
    wg8_sum<3>(v3, red);
    if (tid == 0) {
        if constexpr (MODE == I8_A) {
            scal[0] = v3[0]; scal[1] = v3[1]; scal[2] = mrow; scal[3] = mcol;
        } else {
            scal[0] = v3[0]; scal[1] = v3[1]; scal[2] = v3[2]; scal[3] = 0.0;
            // this tile holds |Z| of rows col0.. (row block tx) x columns row0.. (column block ty)
            double* zmax = stats + (size_t)kTileStats * nb * nb + 2 * (size_t)(ty * nb + tx);
            zmax[0] = mrow; zmax[1] = mcol;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// K3 .. K7: products on split-float16 operands.  Workgroup tile 32 x 32, 512 threads; wave w owns the k-steps [w NS, (w + 1) NS)
// of 16 k each (d = 128 NS).
enum { SP_FIRST = 0, SP_T = 1, SP_U = 2, SP_V2 = 3, SP_V3 = 4 };
struct SplitArgs {
    int d, gen;
    const MatHdr* hA; const MatHdr* hB;  // headers of the two covariances (hA: the shared baseline in a batch of songs)
    int64_t pstride;                     // bytes between consecutive problems of a batch (everything below but hA is per problem)
    int64_t astride;                     // ... of hA: 0 = shared (songs), pstride for a batch of pairs
    const int* skip;
    SplitMat A[2], B[2], C[2];           // per product of the launch: A operand, B operand (its ^T planes are read), output
    uint4* Cdig[2]; uint4* Cdig_t[2];    // digit planes of C[0] and of C[0]^T (SP_FIRST, SP_U)
    float alpha, beta_eye, gamma;        // SP_T: C = alpha A B + beta_eye I, residual partials of (C - gamma I)
    double* partials;                    // [slots] (SP_T)
    // SP_FIRST: A[0] = B[0] = P (split planes of the normalised product), A64 = the product in the caller's units
    const double* A64; const double* statsA;
    NsState* st; Ns32State* s32;
    // SP_U: the check of iteration k rides on the launch (blockIdx.z == 2)
    int k, max_low, nslots;
    int nprob;                           // ns_fast_big.h: problems of the batch
    double l0_scale;                     // ... multiplier on the x_min estimate of ns_l0_from_participation
    int scaled;                          // ns_fast_big.h: scaled steps -- T_k = 1.5 mu I - 0.5 mu^3 Z Y with mu = st->mu[k] per problem (ns_check.h)
    double thr_pred;
    const double* chk_partials;
    // pairs (frechet.hip: fast_enqueue / pairs_enqueue): lp_wide = 1 lets the chain take decaying spectra on scaled steps -- the
    // participation-ratio rule only applies without it -- as long as the x_min estimate stays above l0_min (below, float16 cannot hold Z
    // and the float32-class products cannot resolve the small eigenvalues: the float64 route)
    int lp_wide; double l0_min;
    // SP_V2 / SP_V3: the final iterate's planes are (Zf, Yf)[*sel & 1]; B[0] = R' (V2) / P' (V3); C[0] = P', C[1] = E' (V2); E' read back as
    // A[1] (V3); vstats: [tiles][kVerStats] per problem (pinned host, hstride apart); vwords: the problem's host words (slot 13 = gen)
    const int* sel; SplitMat Zf[2], Yf[2];
    double* vstats; int* vwords; int64_t hstride;
};

// Decision of iteration k from r_k = ||I - Z_k Y_k||_F = 2 ||T_k - I||_F (one workgroup; the rules are those of round 2's
// float32 leg, gemm_f32.hip: ns32_check).
template <int NT = 512> __device__ __forceinline__ void nsf_check(const SplitArgs& g, int64_t po, double* red) {
    Ns32State* st = adv(g.s32, po);
    NsState* st64 = adv(g.st, po);
    const double* chk_partials = adv(g.chk_partials, po);
    const int k = g.k;
    if (st->finished || st64->done) {
        if (threadIdx.x == 0) { st->upd_skip[(k + 1) & 1] = 1; if (st64->done && !st->finished) { st->finished = 1; st->failed = 1; st->done = 1; } }
        return;
    }
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < g.nslots; i += NT) s[0] += chk_partials[i];
    wg8_sum<1, NT / 64>(s, red);
    if (threadIdx.x != 0) return;
    // (scaled step: the partials hold (T - (1.5 mu - 0.5 mu^3) I)^2 = (0.5 mu^3)^2 (I - Z Y)^2)
    const double mu_k = g.scaled ? st64->mu[k] : 1.0;
    const double res = 2.0 * sqrt(s[0]) / (mu_k * mu_k * mu_k);
    st->res[ns32_slot(k)] = res;
    if (g.scaled && k + 1 < kMaxIter) {
        // scale of iteration k + 1: every x of iterate k lies above the schedule's bound -- and, once the residual is below 1, above
        // sqrt(1 - res) (|1 - x^2| <= ||I - Z Y||) -- so a bound that was too careful stops over-scaling as soon as the iterate shows it
        double l = st64->l_cur;
        if (res == res && res < 1.0) { const double lr = sqrt(1.0 - res); if (lr > l) l = lr; }
        (void)ns_step_scale_with(mu_k, l);                        // iterate k + 1's bound under the step this launch is taking
        double ln = l, mn = ns_step_scale(ln);
        const double cap = ns_scale_cap(res, g.d);                 // (iterate k + 1's own residual is not known yet: iterate k's)
        st64->mu[k + 1] = mn < cap ? mn : cap;
        st64->l_cur = l;
    }
    const double prev = (k > 0) ? st->res[ns32_slot(k - 1)] : 1e300;
    const bool finite = (res == res) && !isinf(res);
    // (round 3 ended "k >= 8 and still above 1" here: a song of 2 D frames -- condition number of a few hundred, residual ~1 at
    //  iteration 8, at the float32 floor by 11 -- never got through; only a residual that GROWS above the floor is hopeless)
    // (scaled steps: a lower bound far too small over-scales an iterate once -- the residual bumps by 5-20 % -- before the refined bound
    //  and the cap take over: scripts/ns_emulate_adaptive.py; only two growing residuals in a row give up there)
    const bool grows = k >= 4 && res > prev && res > 1e-3;
    const bool give_up = grows && (!g.scaled || st->grew);
    st->grew = grows ? 1 : 0;
    if (!finite || k + 1 >= g.max_low || give_up) {
        st->failed = 1; st->finished = 1; st->done = 1; st->final_iter = k; st->decided_at = k; st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    if (k >= 1 && res <= 1e-3 && (res > 0.3 * prev || res <= 1e-6)) {       // at the floor: Y_k is final
        st->ok = 1; st->skip_corr = 0; st->finished = 1; st->done = 1; st->final_iter = k; st->decided_at = k; st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    const double bound = 0.75 * res * res + 0.25 * res * res * res;
    if (bound <= (st->strict ? 2e-6 : g.thr_pred) && mu_k == 1.0) {      // Y_{k+1} (this launch's update, a plain step) is final
        st->ok = 1; st->skip_corr = 0; st->finished = 1; st->done = 1; st->final_iter = k + 1; st->decided_at = k; st->upd_skip[(k + 1) & 1] = 1;
    }
}

template <int NS, int MODE>
__global__ __launch_bounds__(512) void nsf_split(SplitArgs g) {
    __shared__ __attribute__((aligned(16))) float part[4 * 32 * 33];      // (four slots: waves 4..7 hand their partial tiles to waves 0..3; see nsf_i8)
    __shared__ float fin[32 * 33];
    __shared__ float fin2[32 * 33];
    __shared__ double red[8 * 4];
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    constexpr int ZPER = (MODE == SP_U) ? 3 : ((MODE == SP_V2) ? 2 : 1);         // z-slices per problem
    const int64_t po = (int64_t)(blockIdx.z / ZPER) * g.pstride;
    const int zs = (int)(blockIdx.z % ZPER);
    if constexpr (MODE == SP_U) {
        if (zs == 2) {
            if (blockIdx.x == 0 && blockIdx.y == 0) nsf_check<512>(g, po, red);
            return;
        }
    }
    const MatHdr* hB = adv(g.hB, po);
    const MatHdr* hA = adv(g.hA, (int64_t)(blockIdx.z / ZPER) * g.astride);
    if (hdr_bad(hA, hB, g.gen)) return;
    const int zi = (MODE == SP_U || MODE == SP_V2) ? zs : 0;
    int ty, tx; tile_of_block(ty, tx);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, r = lane & 31;
    const int row0 = ty * 32, col0 = tx * 32;
    constexpr bool VER = (MODE == SP_V2 || MODE == SP_V3);
    if constexpr (VER) {
        if (g.skip && *adv(g.skip, po) != 0) return;                 // (no final iterate: nothing to verify)
        if (MODE == SP_V3 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
            adv(g.vwords, (int64_t)blockIdx.z * g.hstride)[13] = g.gen;         // the verification record belongs to this score
    }
    const int par = VER ? (*adv(g.sel, po) & 1) : 0;                 // which of the ping-pong iterates is final

    // every load of the product is issued before anything else: the operands were written by the previous kernel from all
    // eight XCDs, the first touch of a panel is an L2 miss, and one exposed latency is all this kernel should pay
    f16x8 ah[NS], al[NS], bh[NS], bl[NS];
    {
        // V2: P' = Z R' (slice 0), Z Y (slice 1); V3: Q' = Z P'
        const SplitMat Am = adv(VER ? g.Zf[par] : g.A[zi], po);
        const SplitMat Bm = adv((MODE == SP_V2 && zs == 1) ? g.Yf[par] : g.B[VER ? 0 : zi], po);
        const f16x8* pa = reinterpret_cast<const f16x8*>(Am.a + fa_idx(ty, wave * NS, 0, lane, d));
        const f16x8* pb = reinterpret_cast<const f16x8*>(Bm.at + fa_idx(tx, wave * NS, 0, lane, d));
#pragma unroll
        for (int s = 0; s < NS; ++s) { ah[s] = pa[128 * s]; al[s] = pa[128 * s + 64]; bh[s] = pb[128 * s]; bl[s] = pb[128 * s + 64]; }
    }
    const int rr = tid >> 4, cp = tid & 15;
    double inv_c = 0.0, inv_cn = 0.0;
    float m1 = 1.5f, m3 = 0.5f;                                      // SP_FIRST: 1.5 mu_0, 0.5 mu_0^3
    double2 a2 = make_double2(0.0, 0.0);
    if constexpr (MODE == SP_FIRST) {
        a2 = *reinterpret_cast<const double2*>(adv(g.A64, po) + (int64_t)(row0 + rr) * d + col0 + 2 * cp);    // this thread's elements of A
        // ---- the scale (what ns_prepare did in a launch of its own): every workgroup, identically, from K2's tile statistics.
        // One record per thread goes through LDS (a per-thread loop over a row of records is a chain of dependent cache misses).
        const int nb = gridDim.x;
        const double* scal = adv(g.statsA, po);
        double* tmax = reinterpret_cast<double*>(part);              // [2][nb * nb]: largest row / column sum of |a| per tile
        double v2[2] = {0.0, 0.0};
        for (int k = tid; k < nb * nb; k += 512) {
            const double2 r0 = *reinterpret_cast<const double2*>(scal + kTileStats * k), r1 = *reinterpret_cast<const double2*>(scal + kTileStats * k + 2);
            v2[0] += r0.x; v2[1] += r0.y;
            tmax[k] = (r1.x == r1.x) ? r1.x : 1e300; tmax[nb * nb + k] = (r1.y == r1.y) ? r1.y : 1e300;
        }
        __syncthreads();
        // ||A||_inf <= max over row blocks of (sum over column blocks of the tile's largest row sum); ||A||_1 likewise
        double bnd = 0.0;
        if (tid < 2 * nb) {
            const int line = tid % nb; const bool col = tid >= nb;
            for (int q = 0; q < nb; ++q) bnd += col ? tmax[nb * nb + q * nb + line] : tmax[line * nb + q];
        }
        const double inf_b = wg8_max((tid < nb) ? bnd : 0.0, red);
        const double one_b = wg8_max((tid >= nb && tid < 2 * nb) ? bnd : 0.0, red);
        wg8_sum<2>(v2, red);
        const double fro2 = v2[0], trA = v2[1];
        double u = sqrt(fro2);
        if (inf_b < u) u = inf_b;
        if (one_b < u) u = one_b;
        // every eigenvalue of A/c must stay below 3 (above, Y converges to a NEGATIVE root): c = u / 2.9 is safe because u bounds
        // the spectral radius; the tile bounds are ~15 % above the true norms of a noise-like matrix, which round 2 divided by 2.5
        double c = u / 2.9;
        const double wmean = (trA > 0.0) ? fro2 / trA : 0.0;      // where the bulk of a flat spectrum sits (||A||_F^2 stands in for tr A^2)
        if (wmean > c && wmean <= u) c = wmean;
        NsState* const st_p = adv(g.st, po);
        const double mean_term = st_p->mean_term, tr1 = hA->tr, tr2 = hB->tr;
        const bool bad = !(fro2 == fro2) || isinf(fro2) || !(trA == trA) || isinf(trA) || !(mean_term == mean_term) || isinf(mean_term);
        const bool zero = !bad && !(c > 0.0);
        // Which products the float32-class iteration serves.  Its bulk must not lie far below the largest covariance entries (A lives
        // on a fixed-point grid of 2^-41 relative to those).  Without scaled steps (g.lp_wide = 0): spectra flat within a few hundred
        // only -- participation ratio (tr A)^2 / ||A||_F^2 >= d/4.  With them: whatever the x_min estimate puts above g.l0_min
        // (ns_fast_big.h: nsf_big<SP_FIRST> is the batch's form of this block).
        const bool flat = trA * trA >= 0.25 * (double)d * fro2;
        const bool scaled = g.scaled && !bad && !zero && u > 0.0 && trA * trA < 0.8 * (double)d * fro2;
        double l0 = 1.0;
        if (scaled) { l0 = ns_l0_from_participation((float)(trA * trA / fro2), d) * g.l0_scale; if (l0 > 0.5) l0 = 0.5; }
        const bool hopeless = !bad && !zero && (c < 0.0078125 || (g.lp_wide ? (scaled && l0 < g.l0_min) || (!scaled && !flat) : !flat));
        double mu0 = 1.0;
        if (scaled && !hopeless) {
            c = u;                                               // every x = sqrt(lambda / c) in (0, 1]: the steps lift the lower end
            double l = l0;
            mu0 = ns_step_scale(l);
            if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { double ln = l; st_p->mu[0] = mu0; st_p->mu[1] = ns_step_scale(ln); st_p->l_cur = l; }
        } else if (g.scaled && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            st_p->mu[0] = 1.0; st_p->mu[1] = 1.0; st_p->l_cur = 1.0;
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            NsState* st = st_p; Ns32State* s32 = adv(g.s32, po);
            st->c = zero ? 1.0 : c * hdr_inv_s12(hA, hB);     // in the caller's units: A / st->c = (s1 s2 A) / c
            st->tr1 = tr1; st->tr2 = tr2;
            st->res_last = 0.0; st->tr_last = 0.0; st->res_min = 1e300; st->tr_safe = 0.0; st->has_safe = 0;
            st->final_iter = zero ? 0 : -1; st->conv = zero ? 1 : 0;
            st->nonfinite = bad ? 1 : 0; st->done = (bad || zero) ? 1 : 0; st->finished = (bad || zero) ? 1 : 0;
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->final_iter = -1; s32->failed = 0;
            s32->upd_skip[0] = 0; s32->upd_skip[1] = 0; s32->skip_corr = 1; s32->decided_at = -1; s32->strict = 0; s32->grew = 0;
            s32->res[0] = 1e300;
            if (bad || zero || hopeless) { s32->done = 1; s32->finished = 1; s32->failed = 1; s32->upd_skip[0] = 1; s32->upd_skip[1] = 1; }
        }
        if (bad || zero || hopeless) return;
        inv_cn = 1.0 / c;                                // for the normalised product P P
        inv_c = inv_cn / hdr_inv_s12(hA, hB);          // for A in the caller's units: 1 / st->c
        m1 = (float)(1.5 * mu0); m3 = (float)(0.5 * mu0 * mu0 * mu0);
    } else if constexpr (!VER) {
        if (g.skip && *adv(g.skip, po) != 0) return;
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc1, 0, 0, 0);
    }
    if (wave >= 4) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) part[(wave - 4) * (32 * 33) + ((reg & 3) + 8 * (reg >> 2) + 4 * kg) * 33 + r] = acc0[reg] + acc1[reg] * kLoInv;
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) part[wave * (32 * 33) + ((reg & 3) + 8 * (reg >> 2) + 4 * kg) * 33 + r] += acc0[reg] + acc1[reg] * kLoInv;
    }
    __syncthreads();
    float alpha = (MODE == SP_T) ? g.alpha : 1.f, beta = (MODE == SP_T) ? g.beta_eye : 0.f, gamma = g.gamma;
    if (MODE == SP_T && g.scaled) {                                  // this problem's step scale (set by FIRST / the previous check)
        const double m = adv(g.st, po)->mu[g.k];
        alpha = (float)(-0.5 * m * m * m); beta = (float)(1.5 * m); gamma = beta + alpha;
    }
    double ss = 0.0;
    double v4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = rr * 33 + 2 * cp + q;
        const float sum = (part[o] + part[(32 * 33) + o]) + (part[2 * (32 * 33) + o] + part[3 * (32 * 33) + o]);
        const bool dg = (row0 + rr) == (col0 + 2 * cp + q);
        if constexpr (MODE == SP_FIRST) {
            // Y0 = A/c; the product is P P = (c Y0)^2 in normalised units: Y1 = mu0 Y0 T0 = 1.5 mu0 Y0 - 0.5 mu0^3 Y0^2, Z1 = T0 = 1.5 mu0 I - 0.5 mu0^3 Y0
            const float y0 = (float)((q ? a2.y : a2.x) * inv_c);
            const float y2 = (float)((double)sum * (inv_cn * inv_cn));
            fin[o] = m1 * y0 - m3 * y2;
            fin2[o] = (dg ? m1 : 0.f) - m3 * y0;
        } else if constexpr (MODE == SP_V2) {
            fin[o] = zs ? ((dg ? 1.f : 0.f) - sum) * kVerScale : sum;        // E' = kVerScale (I - Z Y)  |  P' = Z R'
        } else if constexpr (MODE == SP_V3) {
            // Q'_ij (this thread's) pairs with P'_ji -- the mirror element, read from the ^T planes at (i, j) -- and so does E'_ij
            const int gr = row0 + rr, gc = col0 + 2 * cp;             // (gc even: both elements of the pair sit in one piece)
            int half;
            const size_t pi = fa_elem(gr, gc, 0, d, half);
            const SplitMat Pm = adv(g.B[0], po), Em = adv(g.A[1], po);
            const float pt = used16(reinterpret_cast<const _Float16*>(Pm.at + pi)[half + q], reinterpret_cast<const _Float16*>(Pm.at + pi + 64)[half + q]);
            const float pij = used16(reinterpret_cast<const _Float16*>(Pm.a + pi)[half + q], reinterpret_cast<const _Float16*>(Pm.a + pi + 64)[half + q]);
            const float eij = used16(reinterpret_cast<const _Float16*>(Em.a + pi)[half + q], reinterpret_cast<const _Float16*>(Em.a + pi + 64)[half + q]);
            v4[0] += (double)sum * (double)pt; v4[1] += (double)eij * (double)pt; v4[2] += (double)pij * (double)pij; v4[3] += (double)eij * (double)eij;
        } else {
            const float v = alpha * sum + (dg ? beta : 0.f);
            fin[o] = v;
            if constexpr (MODE == SP_T) { const double e = (double)v - (dg ? (double)gamma : 0.0); ss += e * e; }
        }
    }
    if constexpr (MODE == SP_V3) {
        wg8_sum<4>(v4, red);
        if (tid == 0) {
            double* o = adv(g.vstats, (int64_t)blockIdx.z * g.hstride) + (size_t)kVerStats * (ty * gridDim.x + tx);
            o[0] = v4[0]; o[1] = v4[1]; o[2] = v4[2]; o[3] = v4[3];
        }
        return;
    }
    __syncthreads();
    const bool with_digits = ((MODE == SP_FIRST) || (MODE == SP_U && zi == 0)) && g.Cdig[0] != nullptr;     // (batches digitise the final Y only)
    store_tile(fin, adv(g.C[zi], po), with_digits ? adv(g.Cdig[0], po) : nullptr, with_digits ? adv(g.Cdig_t[0], po) : nullptr, ty, tx, d, tid);
    if constexpr (MODE == SP_FIRST) store_tile(fin2, adv(g.C[1], po), nullptr, nullptr, ty, tx, d, tid);       // Z1 = T0
    if constexpr (MODE == SP_T) {
        double s1[1] = {ss};
        wg8_sum<1>(s1, red);
        if (tid == 0) adv(g.partials, po)[ty * gridDim.x + tx] = s1[0];
    }
}

// Batches: the digit planes of the FINAL iterate only (the single-problem chain writes them for every new Y, because its K8 must follow
// K7 without a launch in between; in a batch that was half of the update kernel's epilogue, ten times per song).
// grid (d * d / 16 / 256, 2, problems): a thread forms one 16-byte digit piece of every plane, y = 0: of Y, y = 1: of Y^T.
struct DigArgs {
    int d, gen;
    const MatHdr* hA; const MatHdr* hB; int64_t pstride;
    const Ns32State* s32;
    SplitMat Y[2]; uint4* dig[2]; uint4* dig_t[2];
    int64_t astride;                     // bytes between consecutive problems' hA (0 = shared)
};
__global__ __launch_bounds__(256) void nsf_digitize(DigArgs g) {
    const int64_t po = (int64_t)blockIdx.z * g.pstride;
    if (hdr_bad(adv(g.hA, (int64_t)blockIdx.z * g.astride), adv(g.hB, po), g.gen)) return;
    const Ns32State* s = adv(g.s32, po);
    if (s->skip_corr) return;
    const int d = g.d, par = s->final_iter & 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (d * d) >> 4) return;
    const SplitMat Ym = adv(g.Y[par], po);
    const uint4* src = blockIdx.y ? Ym.at : Ym.a;
    uint4* dst = adv(blockIdx.y ? g.dig_t[par] : g.dig[par], po);
    const int lane = t & 63, ksrb = t >> 6, ks = ksrb % (d >> 5), rb = ksrb / (d >> 5);
    const int m = lane & 31, g2 = lane >> 5;
    uint32_t w[kDigits][4];
#pragma unroll
    for (int p = 0; p < kDigits; ++p) { w[p][0] = 0u; w[p][1] = 0u; w[p][2] = 0u; w[p][3] = 0u; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint4 uh = src[fa_idx(rb, 2 * ks + g2, 0, 32 * h + m, d)], ul = src[fa_idx(rb, 2 * ks + g2, 1, 32 * h + m, d)];
        f16x8 hi, lo; __builtin_memcpy(&hi, &uh, 16); __builtin_memcpy(&lo, &ul, 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int dg[kDigits];
            digits_of<float>(used16(hi[j], lo[j]), dg);
            const int byte = 8 * h + j;
#pragma unroll
            for (int p = 0; p < kDigits; ++p) w[p][byte >> 2] |= ((uint32_t)dg[p] & 0xffu) << (8 * (byte & 3));
        }
    }
#pragma unroll
    for (int p = 0; p < kDigits; ++p) dst[dg_idx(rb, ks, p, lane, d)] = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
}

// re-arm the low-precision iteration after the host rejected a PREDICTED final iterate (rare): it goes on from that iterate
// and only the float32 floor ends it now
__global__ void nsf_rearm(Ns32State* s32) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        s32->done = 0; s32->finished = 0; s32->ok = 0; s32->skip_corr = 1; s32->upd_skip[0] = 0; s32->upd_skip[1] = 0;
        s32->strict = 1;
    }
}

}  // namespace nsf
}  // namespace fad
