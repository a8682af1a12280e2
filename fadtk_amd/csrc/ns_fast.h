// The square-root chain of ONE Frechet score in nine launches (gfx950): replaces scipy.linalg.sqrtm / eig of fadtk/fad.py:88-92 for
// well-conditioned pairs of dimension 256 / 512 / 768 / 1024 (frechet.hip: fast_begin; everything else keeps the routes there).
//
//   tr sqrt(A), A = C1 C2:   Newton-Schulz  Y0 = A/c, Z0 = I;  T = (3I - ZY)/2;  Y <- Y T;  Z <- T Z   on LOW-precision operands,
//   then ONE float64-accurate correction   tr sqrt(A/c) = tr Y + 1/2 tr(Z (A/c - Y Y)) + O(err^2)     (SURVEY.md H1).
//
// Round 2 ran the iteration on the f32-input MFMA (157 TFLOP/s peak, 7-12 us per 512^3 product) and the two products that need
// float64 accuracy (A = C1 C2 and G = Y Y) on the f64 MFMA (13 and 16 us).  Here every product runs on the 16x-faster matrix pipes:
//
//   * iteration products: operands are stored as SPLIT float16 planes, x = hi + lo / 2048 (hi = fp16(x), lo = fp16((x - hi) 2048):
//     22 significant bits), and a product is three v_mfma_f32_32x32x16_f16 terms, hi hi + (hi lo + lo hi) / 2048, accumulated in
//     float32 -- float32-class accuracy (the dropped lo lo term is 2^-22 of the result), emulated in scripts/ns_emulate_split.py:
//     the iterates and the corrected trace are indistinguishable from the float32 route's.
//   * exact products: both operands are put on a fixed-point grid of 2^-40 (|x| < 2: covariances are normalised by a power of two,
//     the iterates are O(1) by construction) and cut into six balanced base-128 digits (int8); the 30 digit-pair products whose
//     weight matters run on v_mfma_i32_32x32x32_i8 with EXACT int32 accumulation (|sum| <= 6 * 512..1024 * 64^2 < 2^26), pairs of
//     equal weight share an accumulator, and the eight accumulators are combined in float64.  Result: the product of the
//     (2^-41-rounded) operands to ~1e-15 -- measured in the emulation 5e-13 for C1 C2 and 2.5e-15 for Y Y against float64 BLAS.
//
// Every matrix is written by its producer in BOTH orientations (X and X^T planes): an MFMA operand wants 8 (16 for int8) consecutive
// k per lane, so both operands of every product are plain 16-byte row loads straight into registers -- no LDS staging, no
// transpose reads; LDS only sums the k-split partial tiles of the eight waves.  A D = 512 product is 256 workgroups = one per CU.
//
// Launches (D = 512: t = 16, grid 16 x 16 of 32 x 32 tiles, 512 threads):
//   K1 nsf_prepare        packed moments -> (mu, Sigma) x2, tr Sigma, power-of-two scales, digit planes of both Sigma
//   K2 nsf_i8<A>          A = C1 C2 exact: A (float64), its float32 image in both orientations, row sums of |A|, sum a^2, tr A;
//                         a spare workgroup forms the mean term ||mu1 - mu2||^2 (the reference's dtype quirk included)
//   K3 nsf_split<FIRST>   every workgroup derives the scale c from K2's statistics (what ns_prepare did in a launch of its own),
//                         Y1 = Y0 T0, Z1 = T0 (Z0 = I)
//   K4/K6 nsf_split<T>    T = (3I - Z Y)/2 + the residual partials
//   K5/K7 nsf_split<U>    Y <- Y T, Z <- T Z (two products, one launch) + the convergence check as an extra workgroup + the digit
//                         planes of the new Y
//   K8 nsf_i8<G>          G = Y Y exact on the final iterate; R = A/c - G never leaves the workgroup: tr(Z R), ||R||^2, tr Y, |Z| sums
//   K9 nsf_finish         decide (error estimate), result -> pinned host memory
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
constexpr int kDigBlock = kDigits * 16; // bytes of one (row, 16 k) block: digit p of the 16 values at + 16 p
constexpr int kUmin = 3;               // digit pairs (p, q) with p + q >= kUmin are multiplied (weight 2^(7 (p + q) - 80))
constexpr int kGroups = 2 * (kDigits - 1) - kUmin + 1;      // accumulators: p + q = 3 .. 10
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

// byte offset of the digit block of (row, k / 16) in a digit-plane matrix of dimension d
__device__ __host__ __forceinline__ size_t dig_off(int row, int kb16, int d) { return ((size_t)row * (size_t)(d >> 4) + (size_t)kb16) * kDigBlock; }

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

// sums / maxima over a 512-thread workgroup (8 waves); red = 8 doubles of LDS per value
template <int N> __device__ __forceinline__ void wg8_sum(double (&v)[N], double* red) {
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
        for (int w = 0; w < 8; ++w) t += red[w * N + q];
        v[q] = t;
    }
}
__device__ __forceinline__ double wg8_max(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) t = fmax(t, red[w]);
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
// Per-score header (device): what K1 finds out about the two covariances
struct FastHdr {
    double s[2];        // power-of-two scales: |s_i Sigma_i| <= 1
    double tr[2];       // tr Sigma_i (caller's units)
    int bad[2];         // covariance i is not finite / has no positive diagonal / its set has < 2 rows: the float64 route decides
};
__device__ __forceinline__ bool hdr_bad(const FastHdr* h) { return (h->bad[0] | h->bad[1]) != 0; }
__device__ __forceinline__ double hdr_inv_s12(const FastHdr* h) { return 1.0 / (h->s[0] * h->s[1]); }      // powers of two: exact

// One matrix in split-float16 form, both orientations (each plane d x d, row-major).
struct SplitMat { _Float16* h; _Float16* l; _Float16* th; _Float16* tl; };

// ------------------------------------------------------------------------------------------------------------------------------
// K1: (mu, Sigma) of both sets from the packed moments (or the caller's matrices), traces, scales, digit planes.
// grid (d * d / 4096, 2), 256 threads: a thread owns one (row, 16 k) block of one covariance.
struct PrepArgs {
    const double* acc[2];            // packed moments [n | sum | sum xxT], or nullptr: the caller's (mu, Sigma) are used as they are
    const double* cov_in[2];         // ... then these (device)
    int d, ddof;
    double* mus;                     // [2][d]          (written only with acc)
    double* covs;                    // [2][d * d]      (written only with acc)
    int8_t* dig[2];                  // digit planes of s_1 Sigma_1 (A operand) and of (s_2 Sigma_2)^T (B operand)
    NsState* st; Ns32State* s32; FastHdr* hdr;
};

__global__ __launch_bounds__(256) void nsf_prepare(PrepArgs a) {
    __shared__ double red[8];
    const int set = blockIdx.y, tid = threadIdx.x, d = a.d;
    const double* acc = a.acc[set];
    const double n = acc ? acc[0] : 2.0;
    const double* sum = acc ? acc + 1 : nullptr;
    const double* M = acc ? acc + 1 + d : a.cov_in[set];
    const double inv_n = 1.0 / n, inv_nd = 1.0 / (n - (double)a.ddof);
    auto elem = [&](int r, int c) -> double {            // Sigma[r][c]; same expression as moments_finalize_kernel (symmetric bit for bit)
        const double m = M[(int64_t)r * d + c];
        return acc ? (m - (sum[r] * sum[c]) * inv_n) * inv_nd : m;
    };
    // every workgroup finds the scale itself: trace and largest diagonal entry (512 loads; a NaN / Inf shows in the trace)
    double tr = 0.0, mx = 0.0;
    for (int i = tid; i < d; i += 256) { const double v = elem(i, i); tr += v; mx = fmax(mx, v); }
    tr = block_sum(tr, red);
    mx = block_max(mx, red);
    const bool few = acc && n < 2.0;
    const bool bad = few || !(tr == tr) || isinf(tr) || !(mx > 0.0) || isinf(mx);
    int ex = 0;
    if (!bad) (void)frexp(mx, &ex);                      // mx = m 2^ex, m in [0.5, 1)
    const double s = bad ? 1.0 : ldexp(1.0, -ex);
    if (blockIdx.x == 0 && tid == 0) {
        a.hdr->s[set] = s; a.hdr->tr[set] = tr; a.hdr->bad[set] = bad ? 1 : 0;
        a.st->too_few[set] = few ? 1 : 0;
        if (set == 0) {                                  // the per-call reset of the iteration state (finalize_for_frechet did this)
            a.st->done = 0; a.st->finished = 0; a.st->nonfinite = 0; a.st->conv = 0; a.st->final_iter = -1;
            a.st->upd_skip[0] = 0; a.st->upd_skip[1] = 0;
        }
    }
    if (acc && blockIdx.x == 0)
        for (int i = tid; i < d; i += 256) a.mus[(int64_t)set * d + i] = sum[i] / n;      // (bit for bit what finalize_for_frechet writes)
    // this thread's block: row r, columns 16 kb .. 16 kb + 15
    const int blocks_per_row = d >> 4;
    const int g = blockIdx.x * 256 + tid;
    const int r = g / blocks_per_row, kb = g - r * blocks_per_row;
    if (r >= d) return;
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = elem(r, 16 * kb + q);
    if (acc) {
        double* cov = a.covs + (int64_t)set * d * d + (int64_t)r * d + 16 * kb;
#pragma unroll
        for (int q = 0; q < 16; q += 2) *reinterpret_cast<double2*>(cov + q) = make_double2(v[q], v[q + 1]);
    }
    // (Sigma_2 is used as its own transpose: the moments give a bit-for-bit symmetric matrix; for caller-given matrices the
    //  product formed is Sigma_1 Sigma_2^T, which differs from Sigma_1 Sigma_2 by the asymmetry of the caller's Sigma_2 only)
    uint32_t w[kDigits][4];
#pragma unroll
    for (int p = 0; p < kDigits; ++p) { w[p][0] = 0u; w[p][1] = 0u; w[p][2] = 0u; w[p][3] = 0u; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int dg[kDigits];
        digits_of<double>(bad ? 0.0 : v[q] * s, dg);
#pragma unroll
        for (int p = 0; p < kDigits; ++p) w[p][q >> 2] |= ((uint32_t)dg[p] & 0xffu) << (8 * (q & 3));
    }
    uint4* out = reinterpret_cast<uint4*>(a.dig[set] + dig_off(r, kb, d));
#pragma unroll
    for (int p = 0; p < kDigits; ++p) out[p] = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// K2 / K8: exact product of two digit-plane matrices.  Workgroup tile 32 x 32, 512 threads; wave w owns the k-steps
// [w NS8, (w + 1) NS8) of 32 k each (d = 256 NS8); operands straight from global memory (16 bytes per lane, digit and k-step).
enum { I8_A = 0, I8_G = 1 };
struct I8Args {
    const int8_t* Adig; const int8_t* Bdig;      // A operand rows / B operand COLUMNS (= rows of B^T), digit planes
    const int8_t* Adig_alt; const int8_t* Bdig_alt; const int* sel;      // I8_G: used instead when *sel is odd (ping-pong iterates)
    int d;
    const FastHdr* hdr;
    const int* skip;                             // *skip != 0: nothing to do
    double* stats;                               // I8_A: rowabs [nb][d] | scal [nb * nb][2] = (sum a^2, tr share)
                                                 // I8_G: zrow [nb][d] | zcol [nb][d] | scal [nb * nb][4] = (tr(Z R), sum R^2, tr Y share)
    // I8_A
    double* A64; float* P; float* Pt;
    NsState* st; const double* mu1; const double* mu2; int mean_dtype;
    // I8_G
    const double* A64in;
    SplitMat Y[2], Z[2];
};

template <int NS8, int MODE>
__global__ __launch_bounds__(512) void nsf_i8(I8Args g) {
    __shared__ __attribute__((aligned(16))) double part[8 * 32 * 33];
    __shared__ double fin[32 * 33];
    __shared__ double red[8 * 4];
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    if constexpr (MODE == I8_A) {
        if (blockIdx.z == 1) {                   // the spare workgroup: mean term -> state
            __shared__ float gaps[1024];
            if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 256) {
                const double mt = mean_term_block(g.mu1, g.mu2, d, g.mean_dtype, gaps, red);
                if (tid == 0) g.st->mean_term = mt;
            }
            return;
        }
    }
    if (hdr_bad(g.hdr)) return;
    if (g.skip && *g.skip != 0) return;
    int ty, tx; tile_of_block(ty, tx);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kg = lane >> 5;
    const int row0 = ty * 32, col0 = tx * 32;
    const bool alt = (MODE == I8_G) && g.sel && (*g.sel & 1);
    const int8_t* Ad = alt ? g.Adig_alt : g.Adig;
    const int8_t* Bd = alt ? g.Bdig_alt : g.Bdig;

    i32x4 a[2][kDigits], b[2][kDigits];
    auto fetch = [&](int buf, int s) {
        const int kb16 = 2 * (wave * NS8 + s) + kg;
        const i32x4* pa = reinterpret_cast<const i32x4*>(Ad + dig_off(row0 + r, kb16, d));
        const i32x4* pb = reinterpret_cast<const i32x4*>(Bd + dig_off(col0 + r, kb16, d));
#pragma unroll
        for (int p = 0; p < kDigits; ++p) { a[buf][p] = pa[p]; b[buf][p] = pb[p]; }
    };
    i32x16 acc[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][q] = 0;
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < NS8; ++s) {
        if (s + 1 < NS8) fetch((s + 1) & 1, s + 1);
        // pairs (p, q), p + q = u >= kUmin; the j-th pair of every u in turn, so that consecutive MFMAs use different accumulators
#pragma unroll
        for (int j = 0; j < kDigits; ++j)
#pragma unroll
            for (int u = kUmin; u <= 2 * (kDigits - 1); ++u) {
                const int p_lo = (u > kDigits - 1) ? u - (kDigits - 1) : 0;
                const int n_u = (u <= kDigits - 1) ? u + 1 : 2 * (kDigits - 1) - u + 1;
                if (j < n_u) {
                    const int p = p_lo + j, q = u - p;
                    acc[u - kUmin] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s & 1][p], b[s & 1][q], acc[u - kUmin], 0, 0, 0);
                }
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
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
        part[wave * (32 * 33) + rr * 33 + r] = gp[reg];
    }
    __syncthreads();
    // thread -> elements (rr, 2 cp), (rr, 2 cp + 1)
    const int rr = tid >> 4, cp = tid & 15;
    double G2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = rr * 33 + 2 * cp + q;
        G2[q] = ((part[o] + part[(32 * 33) + o]) + (part[2 * (32 * 33) + o] + part[3 * (32 * 33) + o])) +
                ((part[4 * (32 * 33) + o] + part[5 * (32 * 33) + o]) + (part[6 * (32 * 33) + o] + part[7 * (32 * 33) + o]));
    }
    const int nb = gridDim.x;
    const int gr = row0 + rr, gc = col0 + 2 * cp;
    if constexpr (MODE == I8_A) {
        // A in the caller's units (float64), its normalised float32 image in both orientations, the tile's statistics
        const double inv = hdr_inv_s12(g.hdr);
        *reinterpret_cast<double2*>(g.A64 + (int64_t)gr * d + gc) = make_double2(G2[0] * inv, G2[1] * inv);
        *reinterpret_cast<float2*>(g.P + (int64_t)gr * d + gc) = make_float2((float)G2[0], (float)G2[1]);
        fin[rr * 33 + 2 * cp] = G2[0]; fin[rr * 33 + 2 * cp + 1] = G2[1];
        double v2[2] = {G2[0] * G2[0] + G2[1] * G2[1], (gr == gc ? G2[0] : 0.0) + (gr == gc + 1 ? G2[1] : 0.0)};
        __syncthreads();
        {   // transposed image: thread -> (column cc, rows 2 rp, 2 rp + 1)
            const int cc = tid >> 4, rp = tid & 15;
            *reinterpret_cast<float2*>(g.Pt + (int64_t)(col0 + cc) * d + row0 + 2 * rp) =
                make_float2((float)fin[(2 * rp) * 33 + cc], (float)fin[(2 * rp + 1) * 33 + cc]);
        }
        if (tid < 32) {
            double t = 0.0;
            for (int c = 0; c < 32; ++c) t += fabs(fin[tid * 33 + c]);
            g.stats[(int64_t)tx * d + row0 + tid] = t;                       // rowabs[tx][row]
        }
        wg8_sum<2>(v2, red);
        if (tid == 0) {
            double* scal = g.stats + (int64_t)nb * d + 2 * (ty * nb + tx);
            scal[0] = v2[0]; scal[1] = v2[1];
        }
    } else {
        // R = A/c - G for this tile; Z enters through its mirror tile: Z^T[gr][gc] = Z[gc][gr] pairs with R[gr][gc] in tr(Z R)
        const SplitMat& Zm = g.Z[alt ? 1 : 0];
        const SplitMat& Ym = g.Y[alt ? 1 : 0];
        const double inv_c = 1.0 / g.st->c;
        const double2 a2 = *reinterpret_cast<const double2*>(g.A64in + (int64_t)gr * d + gc);
        const f16x2 zh = *reinterpret_cast<const f16x2*>(Zm.th + (int64_t)gr * d + gc);
        const f16x2 zl = *reinterpret_cast<const f16x2*>(Zm.tl + (int64_t)gr * d + gc);
        const double z0 = (double)used16(zh[0], zl[0]), z1 = (double)used16(zh[1], zl[1]);
        const double R0 = a2.x * inv_c - G2[0], R1 = a2.y * inv_c - G2[1];
        double v3[3] = {z0 * R0 + z1 * R1, R0 * R0 + R1 * R1, 0.0};
        if (gr == gc) v3[2] = (double)used16(Ym.h[(int64_t)gr * d + gr], Ym.l[(int64_t)gr * d + gr]);
        if (gr == gc + 1) v3[2] = (double)used16(Ym.h[(int64_t)gr * d + gr], Ym.l[(int64_t)gr * d + gr]);
        fin[rr * 33 + 2 * cp] = fabs(z0); fin[rr * 33 + 2 * cp + 1] = fabs(z1);      // |Z[col0 + c][row0 + r]| at (r, c)
        __syncthreads();
        double* zrow = g.stats;                                  // [nb][d]: partial sums over a 32-column block of a ROW of Z
        double* zcol = g.stats + (int64_t)nb * d;                // [nb][d]: partial sums over a 32-row block of a COLUMN of Z
        if (tid < 32) {                                          // fixed c = tid: sum over r of |Z[col0 + c][row0 + r]| -> row col0 + c, column block ty
            double t = 0.0;
            for (int q = 0; q < 32; ++q) t += fin[q * 33 + tid];
            zrow[(int64_t)ty * d + col0 + tid] = t;
        } else if (tid < 64) {                                   // fixed r: sum over c of |Z[col0 + c][row0 + r]| -> column row0 + r, row block tx
            const int q = tid - 32;
            double t = 0.0;
            for (int c = 0; c < 32; ++c) t += fin[q * 33 + c];
            zcol[(int64_t)tx * d + row0 + q] = t;
        }
        wg8_sum<3>(v3, red);
        if (tid == 0) {
            double* scal = g.stats + 2 * (int64_t)nb * d + 4 * (ty * nb + tx);
            scal[0] = v3[0]; scal[1] = v3[1]; scal[2] = v3[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// K3 .. K7: products on split-float16 operands.  Workgroup tile 32 x 32, 512 threads; wave w owns the k-steps [w NS, (w + 1) NS)
// of 16 k each (d = 128 NS).
enum { SP_FIRST = 0, SP_T = 1, SP_U = 2 };
struct SplitArgs {
    int d;
    const FastHdr* hdr;
    const int* skip;
    SplitMat A[2], B[2], C[2];           // per product of the launch: A operand, B operand (its ^T planes are read), output
    int8_t* Cdig[2]; int8_t* Cdig_t[2];  // digit planes of C[0] and of C[0]^T (SP_FIRST, SP_U)
    float alpha, beta_eye, gamma;        // SP_T: C = alpha A B + beta_eye I, residual partials of (C - gamma I)
    double* partials;                    // [slots] (SP_T)
    // SP_FIRST
    const float* P; const float* Pt; const double* statsA;
    NsState* st; Ns32State* s32;
    // SP_U: the check of iteration k rides on the launch (blockIdx.z == 2)
    int k, max_low, nslots;
    double thr_pred;
    const double* chk_partials;
};

// Decision of iteration k from r_k = ||I - Z_k Y_k||_F = 2 ||T_k - I||_F (one workgroup; the rules are those of round 2's
// float32 leg, gemm_f32.hip: ns32_check).
__device__ __forceinline__ void nsf_check(const SplitArgs& g, double* red) {
    Ns32State* st = g.s32;
    const int k = g.k;
    if (st->finished || g.st->done) {
        if (threadIdx.x == 0) { st->upd_skip[(k + 1) & 1] = 1; if (g.st->done && !st->finished) { st->finished = 1; st->failed = 1; st->done = 1; } }
        return;
    }
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < g.nslots; i += 512) s[0] += g.chk_partials[i];
    wg8_sum<1>(s, red);
    if (threadIdx.x != 0) return;
    const double res = 2.0 * sqrt(s[0]);
    if (k < 16) st->res[k] = res;
    const double prev = (k > 0 && k <= 16) ? st->res[k - 1] : 1e300;
    const bool finite = (res == res) && !isinf(res);
    if (!finite || k + 1 >= g.max_low || (k >= 8 && res > 1.0)) {
        st->failed = 1; st->finished = 1; st->done = 1; st->final_iter = k; st->decided_at = k; st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    if (k >= 1 && res <= 1e-3 && (res > 0.3 * prev || res <= 1e-6)) {       // at the floor: Y_k is final
        st->ok = 1; st->skip_corr = 0; st->finished = 1; st->done = 1; st->final_iter = k; st->decided_at = k; st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    const double bound = 0.75 * res * res + 0.25 * res * res * res;
    if (bound <= (st->strict ? 2e-6 : g.thr_pred)) {              // Y_{k+1} (this launch's update) is final
        st->ok = 1; st->skip_corr = 0; st->finished = 1; st->done = 1; st->final_iter = k + 1; st->decided_at = k; st->upd_skip[(k + 1) & 1] = 1;
    }
}

// 8 halves of a fragment from 8 floats: hi and (scaled) lo parts
__device__ __forceinline__ void split_frag(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { _Float16 h, l; split16(v[q], h, l); hi[q] = h; lo[q] = l; }
}

// the tile in `fin` (float [32][33]) -> split planes of X (row-major) and X^T, and optionally the digit planes of both
__device__ __forceinline__ void store_tile(const float* fin, const SplitMat& X, int8_t* dig, int8_t* dig_t, int row0, int col0, int d, int tid) {
    {   // row-major: (rr, 2 cp), (rr, 2 cp + 1)
        const int rr = tid >> 4, cp = tid & 15;
        _Float16 h0, l0, h1, l1;
        split16(fin[rr * 33 + 2 * cp], h0, l0); split16(fin[rr * 33 + 2 * cp + 1], h1, l1);
        const int64_t o = (int64_t)(row0 + rr) * d + col0 + 2 * cp;
        *reinterpret_cast<f16x2*>(X.h + o) = (f16x2){h0, h1};
        *reinterpret_cast<f16x2*>(X.l + o) = (f16x2){l0, l1};
    }
    {   // transposed: column cc of the tile is a row of X^T; rows 2 rp, 2 rp + 1 of the tile are adjacent there
        const int cc = tid >> 4, rp = tid & 15;
        _Float16 h0, l0, h1, l1;
        split16(fin[(2 * rp) * 33 + cc], h0, l0); split16(fin[(2 * rp + 1) * 33 + cc], h1, l1);
        const int64_t o = (int64_t)(col0 + cc) * d + row0 + 2 * rp;
        *reinterpret_cast<f16x2*>(X.th + o) = (f16x2){h0, h1};
        *reinterpret_cast<f16x2*>(X.tl + o) = (f16x2){l0, l1};
    }
    if (dig) {
        // four consecutive k per thread = one dword per digit.  tid < 256: X (k = columns); tid >= 256: X^T (k = rows)
        const bool tr = tid >= 256;
        const int u = tid & 255, line = u >> 3, k4 = (u & 7) * 4;
        int8_t* base = tr ? dig_t + dig_off(col0 + line, (row0 + k4) >> 4, d) + ((row0 + k4) & 15)
                          : dig + dig_off(row0 + line, (col0 + k4) >> 4, d) + ((col0 + k4) & 15);
        uint32_t w[kDigits];
#pragma unroll
        for (int p = 0; p < kDigits; ++p) w[p] = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = tr ? fin[(k4 + q) * 33 + line] : fin[line * 33 + k4 + q];
            _Float16 h, l; split16(v, h, l);
            int dg[kDigits];
            digits_of<float>(used16(h, l), dg);              // the value the MFMAs of the iteration see
#pragma unroll
            for (int p = 0; p < kDigits; ++p) w[p] |= ((uint32_t)dg[p] & 0xffu) << (8 * q);
        }
#pragma unroll
        for (int p = 0; p < kDigits; ++p) *reinterpret_cast<uint32_t*>(base + 16 * p) = w[p];
    }
}

template <int NS, int MODE>
__global__ __launch_bounds__(512) void nsf_split(SplitArgs g) {
    __shared__ __attribute__((aligned(16))) float part[8 * 32 * 33];
    __shared__ float fin[32 * 33];
    __shared__ float fin2[32 * 33];
    __shared__ double red[8 * 4];
    const int d = g.d, tid = threadIdx.x, lane = tid & 63;
    if constexpr (MODE == SP_U) {
        if (blockIdx.z == 2) {
            if (blockIdx.x == 0 && blockIdx.y == 0) nsf_check(g, red);
            return;
        }
    }
    if (hdr_bad(g.hdr)) return;
    const int zi = (MODE == SP_U) ? (int)blockIdx.z : 0;
    int ty, tx; tile_of_block(ty, tx);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kg = lane >> 5;
    const int row0 = ty * 32, col0 = tx * 32;
    const int k0 = wave * (16 * NS) + 8 * kg;            // this lane's first k; step s adds 16 s

    f16x8 ah[NS], al[NS], bh[NS], bl[NS];
    if constexpr (MODE == SP_FIRST) {
        // operands do not exist as split matrices yet: Y0 = P / c and T0 = (3I - Y0)/2 are formed on the way in.  The float32
        // rows are requested first, the scale is derived while they travel.
        float4 pa[NS][2], pb[NS][2];
        const float* rowA = g.P + (int64_t)(row0 + r) * d + k0;
        const float* rowB = g.Pt + (int64_t)(col0 + r) * d + k0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pa[s][0] = *reinterpret_cast<const float4*>(rowA + 16 * s); pa[s][1] = *reinterpret_cast<const float4*>(rowA + 16 * s + 4);
            pb[s][0] = *reinterpret_cast<const float4*>(rowB + 16 * s); pb[s][1] = *reinterpret_cast<const float4*>(rowB + 16 * s + 4);
        }
        const int rr = tid >> 4, cp = tid & 15;
        const float2 pz = *reinterpret_cast<const float2*>(g.P + (int64_t)(row0 + rr) * d + col0 + 2 * cp);     // this thread's elements of Z1 = T0
        // ---- the scale (what ns_prepare did in a launch of its own): every workgroup, identically
        const int nb = gridDim.x;
        const double* rowabs = g.statsA;
        const double* scal = g.statsA + (int64_t)nb * d;
        double mr = 0.0;
        for (int i = tid; i < d; i += 512) {
            double rs = 0.0;
            for (int k = 0; k < nb; k += 8) {
                double q8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) q8[q] = (k + q < nb) ? rowabs[(int64_t)(k + q) * d + i] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) rs += q8[q];
            }
            mr = fmax(mr, (rs == rs) ? rs : 1e300);
        }
        double v2[2] = {0.0, 0.0};
        for (int k = tid; k < nb * nb; k += 512) { v2[0] += scal[2 * k]; v2[1] += scal[2 * k + 1]; }
        const double inf_norm = wg8_max(mr, red);
        wg8_sum<2>(v2, red);
        const double fro2 = v2[0], trA = v2[1];
        double u = sqrt(fro2);
        if (inf_norm < u) u = inf_norm;
        double c = u / 2.5;                              // every eigenvalue of A/c below 3 (above, Y converges to a NEGATIVE root)
        const double wmean = (trA > 0.0) ? fro2 / trA : 0.0;      // where the bulk of a flat spectrum sits (||A||_F^2 stands in for tr A^2)
        if (wmean > c && wmean <= u) c = wmean;
        const double mean_term = g.st->mean_term, tr1 = g.hdr->tr[0], tr2 = g.hdr->tr[1];
        const bool bad = !(fro2 == fro2) || isinf(fro2) || !(trA == trA) || isinf(trA) || !(mean_term == mean_term) || isinf(mean_term);
        const bool zero = !bad && !(c > 0.0);
        // the float32-class iteration only serves spectra that are flat within a few hundred: participation ratio (tr A)^2 / ||A||_F^2 >= d/4
        const bool hopeless = !bad && !zero && (trA * trA < 0.25 * (double)d * fro2);
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            NsState* st = g.st; Ns32State* s32 = g.s32;
            st->c = zero ? 1.0 : c * hdr_inv_s12(g.hdr);     // in the caller's units: A / st->c = (s1 s2 A) / c
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
        const float inv = (float)(1.0 / c);
        // ---- operands
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float va[8] = {pa[s][0].x * inv, pa[s][0].y * inv, pa[s][0].z * inv, pa[s][0].w * inv,
                                 pa[s][1].x * inv, pa[s][1].y * inv, pa[s][1].z * inv, pa[s][1].w * inv};
            split_frag(va, ah[s], al[s]);
            const int kk = k0 + 16 * s, j = col0 + r;    // B^T row j, elements k = kk .. kk + 7: T0[k][j] = 1.5 [k == j] - 0.5 Y0[k][j]
            const float pv[8] = {pb[s][0].x, pb[s][0].y, pb[s][0].z, pb[s][0].w, pb[s][1].x, pb[s][1].y, pb[s][1].z, pb[s][1].w};
            float vb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) vb[q] = ((kk + q == j) ? 1.5f : 0.f) - 0.5f * (pv[q] * inv);
            split_frag(vb, bh[s], bl[s]);
        }
        fin2[rr * 33 + 2 * cp] = ((row0 + rr == col0 + 2 * cp) ? 1.5f : 0.f) - 0.5f * (pz.x * inv);
        fin2[rr * 33 + 2 * cp + 1] = ((row0 + rr == col0 + 2 * cp + 1) ? 1.5f : 0.f) - 0.5f * (pz.y * inv);
    } else {
        const SplitMat& Am = g.A[zi];
        const SplitMat& Bm = g.B[zi];
        const int64_t oa = (int64_t)(row0 + r) * d + k0, ob = (int64_t)(col0 + r) * d + k0;
        // every load of the product is issued before anything else: the operands were written by the previous kernel from all
        // eight XCDs, the first touch of a panel is an L2 miss, and one exposed latency is all this kernel should pay
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ah[s] = *reinterpret_cast<const f16x8*>(Am.h + oa + 16 * s);
            al[s] = *reinterpret_cast<const f16x8*>(Am.l + oa + 16 * s);
            bh[s] = *reinterpret_cast<const f16x8*>(Bm.th + ob + 16 * s);
            bl[s] = *reinterpret_cast<const f16x8*>(Bm.tl + ob + 16 * s);
        }
        if (g.skip && *g.skip != 0) return;
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
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
        part[wave * (32 * 33) + rr * 33 + r] = acc0[reg] + acc1[reg] * kLoInv;
    }
    __syncthreads();
    const int rr = tid >> 4, cp = tid & 15;
    const float alpha = (MODE == SP_T) ? g.alpha : 1.f, beta = (MODE == SP_T) ? g.beta_eye : 0.f;
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = rr * 33 + 2 * cp + q;
        const float sum = ((part[o] + part[(32 * 33) + o]) + (part[2 * (32 * 33) + o] + part[3 * (32 * 33) + o])) +
                          ((part[4 * (32 * 33) + o] + part[5 * (32 * 33) + o]) + (part[6 * (32 * 33) + o] + part[7 * (32 * 33) + o]));
        const bool dg = (row0 + rr) == (col0 + 2 * cp + q);
        const float v = alpha * sum + (dg ? beta : 0.f);
        fin[o] = v;
        if constexpr (MODE == SP_T) { const double e = (double)v - (dg ? (double)g.gamma : 0.0); ss += e * e; }
    }
    __syncthreads();
    const bool with_digits = (MODE == SP_FIRST) || (MODE == SP_U && zi == 0);
    store_tile(fin, g.C[zi], with_digits ? g.Cdig[0] : nullptr, with_digits ? g.Cdig_t[0] : nullptr, row0, col0, d, tid);
    if constexpr (MODE == SP_FIRST) store_tile(fin2, g.C[1], nullptr, nullptr, row0, col0, d, tid);       // Z1 = T0
    if constexpr (MODE == SP_T) {
        double s1[1] = {ss};
        wg8_sum<1>(s1, red);
        if (tid == 0) g.partials[ty * gridDim.x + tx] = s1[0];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// K9: reduce the correction's partials, decide, write the result where the host reads it (pinned host memory).  One block.
struct FastResult {          // = MixedResult of frechet.hip + `prepared`
    int status;              // 1 accepted, 2 rejected -> float64 iteration, 4 predicted final iterate rejected -> iterate on, 0 not finished
    int iters, decided_at, nonfinite, too_few0, too_few1;
    double tr_scaled, c, tr1, tr2, mean_term, res, est;
    int prepared;            // A = C1 C2 (float64) and the armed state are valid: the float64 route may start from them
    int pad;
};

__global__ __launch_bounds__(256) void nsf_finish(const double* __restrict__ stats, int d, int nb, const FastHdr* __restrict__ hdr,
                                                  const NsState* __restrict__ st, Ns32State* __restrict__ s32,
                                                  FastResult* __restrict__ out, int max_low) {
    __shared__ double red[4];
    __shared__ double red3[12];
    const int tid = threadIdx.x;
    const bool live = !hdr_bad(hdr) && s32->ok != 0;
    double mr = 0.0, mc = 0.0, corr = 0.0, r2 = 0.0, tr = 0.0;
    if (live) {
        const double* zrow = stats;
        const double* zcol = stats + (int64_t)nb * d;
        const double* scal = stats + 2 * (int64_t)nb * d;
        for (int i = tid; i < d; i += 256) {
            double rs = 0.0, cs = 0.0;
            for (int k = 0; k < nb; k += 8) {
                double a8[8], b8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { a8[q] = (k + q < nb) ? zrow[(int64_t)(k + q) * d + i] : 0.0; b8[q] = (k + q < nb) ? zcol[(int64_t)(k + q) * d + i] : 0.0; }
#pragma unroll
                for (int q = 0; q < 8; ++q) { rs += a8[q]; cs += b8[q]; }
            }
            mr = fmax(mr, rs); mc = fmax(mc, cs);
        }
        for (int k = tid; k < nb * nb; k += 256) { corr += scal[4 * k]; r2 += scal[4 * k + 1]; tr += scal[4 * k + 2]; }
    }
    const double zinf = block_max(mr, red), zone = block_max(mc, red);
    double v3[3] = {corr, r2, tr};
    block_sum_n<3>(v3, red3);
    corr = v3[0]; r2 = v3[1]; tr = v3[2];
    if (tid != 0) return;
    FastResult o;
    o.status = 0; o.iters = s32->final_iter; o.decided_at = s32->decided_at; o.nonfinite = st->nonfinite;
    o.too_few0 = st->too_few[0]; o.too_few1 = st->too_few[1];
    o.c = st->c; o.tr1 = st->tr1; o.tr2 = st->tr2; o.mean_term = st->mean_term;
    o.tr_scaled = 0.0; o.res = 0.0; o.est = 0.0; o.prepared = hdr_bad(hdr) ? 0 : 1; o.pad = 0;
    if (hdr_bad(hdr) || st->done || s32->failed) {
        o.status = 2;
    } else if (live) {
        const int f = s32->final_iter;
        const int fm = f < 16 ? f : 15;
        double res = s32->res[fm];
        if (s32->decided_at == f - 1) { const double rp = s32->res[f - 1 < 16 ? f - 1 : 15]; res = 0.75 * rp * rp + 0.25 * rp * rp * rp; if (res < 2e-6) res = 2e-6; }
        const double zn = sqrt(zinf * zone), rn = sqrt(r2);
        const double trs = tr + 0.5 * corr;
        const double est = zn * zn * zn * rn * rn / 8.0 + zn * res * rn / 2.0;
        const bool finite = (trs == trs) && !isinf(trs) && (est == est) && !isinf(est);
        o.tr_scaled = trs; o.res = res; o.est = est;
        o.status = (finite && est <= 1e-9 * fabs(trs)) ? 1 : 2;
        if (o.status == 2 && finite && !s32->strict && s32->decided_at == f - 1 && f + 1 < max_low) {
            o.status = 4;        // a PREDICTED final iterate the correction cannot absorb: go on from it, the float32 floor ends it now
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->skip_corr = 1; s32->upd_skip[0] = 0; s32->upd_skip[1] = 0;
            s32->strict = 1;
        }
    }
    *out = o;
}

}  // namespace nsf
}  // namespace fad
