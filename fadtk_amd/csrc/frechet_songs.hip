// Per-song Frechet distances against one baseline on the GPU (gfx950): fad_frechet_batched_vs_baseline.  Replaces the per-song loop of
// score_individual, fadtk/fad.py:373-387 (np.cov + calc_frechet_distance per song on a thread pool).  This file: the per-song kernels
// (statistics, covariances, the rank-one and Gram-matrix shortcuts) and the dispatcher that sorts the songs of a call by frame count
// into routes; the square roots themselves run in frechet.hip (low-precision chain batched over songs) and frechet_f64.hip (float64).
//
// Two-frame songs (Whisper) never need a matrix root: with d = x1 - x2,
// Sigma_s = d d^T / 2 is rank one and tr sqrt(Sigma_b Sigma_s) = sqrt(d^T Sigma_b d / 2).
#include "frechet_internal.h"

#include <algorithm>
#include <cmath>
#include <type_traits>

namespace fad {

typedef double f64x4 __attribute__((ext_vector_type(4)));


// ==========================================================================================
// per-song kernels
// ==========================================================================================
template <typename TIn> __device__ __forceinline__ double ld_f64(const TIn* p, int64_t i);
struct r_f16 { uint16_t b; };
struct r_bf16 { uint16_t b; };
template <> __device__ __forceinline__ double ld_f64<double>(const double* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ double ld_f64<float>(const float* p, int64_t i) { return (double)p[i]; }
template <> __device__ __forceinline__ double ld_f64<r_f16>(const r_f16* p, int64_t i) {
    _Float16 h; uint16_t s = p[i].b; __builtin_memcpy(&h, &s, 2); return (double)(float)h;
}
template <> __device__ __forceinline__ double ld_f64<r_bf16>(const r_bf16* p, int64_t i) {
    return (double)__uint_as_float(((uint32_t)p[i].b) << 16);
}

// numpy's mean of an fp16 / bf16 / fp32 matrix is rounded to that dtype
template <typename TIn> __device__ __forceinline__ double round_like_input(double v) { return v; }
template <> __device__ __forceinline__ double round_like_input<float>(double v) { return (double)(float)v; }
template <> __device__ __forceinline__ double round_like_input<r_f16>(double v) { return (double)(float)(_Float16)(float)v; }
template <> __device__ __forceinline__ double round_like_input<r_bf16>(double v) {
    uint32_t u = __float_as_uint((float)v);
    u += 0x7fffu + ((u >> 16) & 1u);              // round to nearest even
    return (double)__uint_as_float(u & 0xffff0000u);
}

// The song's mean of column `a` as the reference's np.mean(embd, axis=0) returns it (fad.py:48, mean_mode = 1): numpy widens float16 /
// bfloat16 frames to float32, adds the rows ONE AFTER THE OTHER in float32, divides (fad_common.h: numpy_mean_of_f32_sum) and rounds the
// quotient to the frames' type.  For a few thousand frames the float32 running sum ends ~1e-6 off the exact one: the float32 mean differs
// in its last bits (up to 5e-5 of a small score, tests/test_gpu_fuzz.py), the float16 mean by one ulp in ~0.3 % of the dimensions when the
// frames carry an offset (rounds 1-4 returned the rounded EXACT mean for 16-bit frames; fixture g4.shifted holds the walk to the
// reference's own scores).  One thread walks the column in that order, eight loads in flight; float64 frames: numpy's sum is the exact one.
template <typename TIn> __device__ __forceinline__ float ld_f32(const TIn* p, int64_t i) { return (float)ld_f64<TIn>(p, i); }
// (`run` != nullptr: the song's running sum of this column, walked by segment_running_sums_launch -- songs of many frames: one thread
//  per column walking 2250 rows inside the statistics kernel cost 0.5 ms of a 2.6 ms call.  kMeanPlaceholder: the rounded exact mean for
//  now; the walk runs on a stream of its own beside the covariances and the square roots, and batched_impl swaps the mean term at the end)
#define kMeanPlaceholder (reinterpret_cast<const float*>(uintptr_t(1)))
template <typename TIn>
__device__ __forceinline__ double mean_like_reference(const TIn* __restrict__ rows, int64_t ld, int a, int64_t r0, int64_t r1, double m_exact,
                                                      const float* __restrict__ run = nullptr) {
    if constexpr (std::is_same<TIn, double>::value) {
        return m_exact;
    } else {
        if (r1 <= r0) return 0.0;
        if (run == kMeanPlaceholder) return round_like_input<TIn>(m_exact);      // (the caller replaces the mean term afterwards: batched_impl)
        if (run) return round_like_input<TIn>(numpy_mean_of_f32_sum(*run, (double)(r1 - r0)));
        float acc = 0.f;
        int64_t r = r0;
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld_f32<TIn>(rows, (r + u) * ld + a);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + v[u];
        }
        for (; r < r1; ++r) acc = acc + ld_f32<TIn>(rows, r * ld + a);
        return round_like_input<TIn>(numpy_mean_of_f32_sum(acc, (double)(r1 - r0)));
    }
}

// One workgroup per song: exact fp64 mean, mean as the reference sees it, ||mu_b - mean||^2,
// tr Sigma_s = sum ||x - mean||^2 / (n - 1), and for two-frame songs the difference row d = x1 - x2.
template <typename TIn>
__global__ __launch_bounds__(256) void song_stats(const TIn* __restrict__ rows, int64_t ld, int d,
                                                  const int64_t* __restrict__ offsets, const double* __restrict__ mu_b,
                                                  int mean_mode, double* __restrict__ mean_exact,
                                                  double* __restrict__ scal /*[S][2]*/) {
    __shared__ double red[4];
    const int64_t s = blockIdx.x;
    const int64_t r0 = offsets[s], r1 = offsets[s + 1];
    const int64_t n = r1 - r0;
    double mt = 0.0, ts = 0.0;
    for (int a = threadIdx.x; a < d; a += 256) {
        double sum = 0.0;
        for (int64_t r = r0; r < r1; ++r) sum += ld_f64<TIn>(rows, r * ld + a);
        const double m = (n > 0) ? sum / (double)n : 0.0;
        const double mr = mean_mode ? mean_like_reference<TIn>(rows, ld, a, r0, r1, m) : m;
        if (mean_exact) mean_exact[s * d + a] = m;
        double sq = 0.0;
        for (int64_t r = r0; r < r1; ++r) { const double c = ld_f64<TIn>(rows, r * ld + a) - m; sq += c * c; }
        ts += sq;
        const double df = mu_b[a] - mr;
        mt += df * df;
    }
    mt = block_sum(mt, red);
    ts = block_sum(ts, red);
    if (threadIdx.x == 0) { scal[2 * s] = mt; scal[2 * s + 1] = (n > 1) ? ts / (double)(n - 1) : 0.0; }
}

// The same for songs of many frames: one workgroup per (song, 64 columns); its four waves take every fourth row each and their
// partial sums meet in LDS -- one thread per column walking 2250 rows twice made this kernel 15 % of the per-song route at the
// Encodec shape, and one workgroup per song left a call of 64 long songs with 64 workgroups.  The chunks' shares of the two
// scalars are summed by song_scal_sum.  (Songs of a few frames keep the kernel above: there the columns are the parallelism.)
template <typename TIn>
__global__ __launch_bounds__(256) void song_stats_long(const TIn* __restrict__ rows, int64_t ld, int d,
                                                       const int64_t* __restrict__ offsets, const double* __restrict__ mu_b,
                                                       int mean_mode, double* __restrict__ mean_exact,
                                                       double* __restrict__ part /*[S][chunks][2]*/, const float* __restrict__ runs = nullptr) {
    __shared__ double psum[4][64];
    __shared__ double red[4];
    const int64_t s = blockIdx.x;
    const int64_t r0 = offsets[s], r1 = offsets[s + 1];
    const int64_t n = r1 - r0;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int a = blockIdx.y * 64 + cl;
    const bool ok = a < d;
    double mt = 0.0, ts = 0.0;
    double sum = 0.0;
    if (ok) for (int64_t r = r0 + rl; r < r1; r += 4) sum += ld_f64<TIn>(rows, r * ld + a);
    psum[rl][cl] = sum;
    __syncthreads();
    const double tot = (psum[0][cl] + psum[1][cl]) + (psum[2][cl] + psum[3][cl]);
    const double m = (n > 0) ? tot / (double)n : 0.0;
    __syncthreads();
    double sq = 0.0;
    if (ok) for (int64_t r = r0 + rl; r < r1; r += 4) { const double c = ld_f64<TIn>(rows, r * ld + a) - m; sq += c * c; }
    psum[rl][cl] = sq;
    __syncthreads();
    if (rl == 0 && ok) {
        const double mr = mean_mode ? mean_like_reference<TIn>(rows, ld, a, r0, r1, m, runs == kMeanPlaceholder ? runs : (runs ? runs + s * d + a : nullptr)) : m;
        if (mean_exact) mean_exact[s * d + a] = m;
        ts = (psum[0][cl] + psum[1][cl]) + (psum[2][cl] + psum[3][cl]);
        const double df = mu_b[a] - mr;
        mt = df * df;
    }
    mt = block_sum(mt, red);
    ts = block_sum(ts, red);
    if (threadIdx.x == 0) {
        double* o = part + 2 * (s * gridDim.y + blockIdx.y);
        o[0] = mt; o[1] = (n > 1) ? ts / (double)(n - 1) : 0.0;
    }
}

// float16 frames, 16-byte aligned rows (D a multiple of 8): ONE pass, eight columns (one 16-byte load) per thread and row, the
// workgroup's other threads on other rows.  Sums of x - x0 and of (x - x0)^2 in float64, x0 = the song's first frame (the
// differences and their squares are exact in float64; sum q - s^2 / n loses a factor (1 + (mean - x0)^2 / var) of 1e-16).  One
// workgroup per song writes the mean, the mean term and tr Sigma_s.  The two-pass kernel above read the frames twice, two bytes per
// lane: 1.14 ms for 2000 songs of [2250 x 128] = 1.0 TB/s (profiles/r03m_c4_kernel_stats.csv).
__global__ __launch_bounds__(256) void song_stats_f16(const uint16_t* __restrict__ rows, int64_t ld, int d,
                                                      const int64_t* __restrict__ offsets, const double* __restrict__ mu_b,
                                                      int mean_mode, double* __restrict__ mean_exact, double* __restrict__ var_exact,
                                                      double* __restrict__ out /*[S][chunks][2]: scal itself when there is one chunk*/,
                                                      const float* __restrict__ runs = nullptr) {
    // grid (songs, chunks of 128 columns): 16 column groups of 8 side by side, 16 row lanes
    __shared__ double sm[16 * 16 * 8 * 2];               // [row lane][group][column][sum | sum of squares]
    __shared__ double red[4];
    const int64_t s = blockIdx.x;
    const int64_t r0 = offsets[s], r1 = offsets[s + 1], n = r1 - r0;
    const int gl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int grp = blockIdx.y * 16 + gl;
    const bool live = grp * 8 < d;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    double sx[8], sq[8];
    float x0[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { sx[q] = 0.0; sq[q] = 0.0; x0[q] = 0.f; }
    if (live && n > 0) {
        const uint4 u0 = *reinterpret_cast<const uint4*>(rows + r0 * ld + grp * 8);
        h8 h; __builtin_memcpy(&h, &u0, 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) x0[q] = (float)h[q];
        for (int64_t r = r0 + rl; r < r1; r += 16) {
            const uint4 u = *reinterpret_cast<const uint4*>(rows + r * ld + grp * 8);
            h8 x; __builtin_memcpy(&x, &u, 16);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double e = (double)((float)x[q] - x0[q]);       // exact: two float16 values
                sx[q] += e; sq[q] = __builtin_fma(e, e, sq[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { sm[((rl * 16 + gl) * 8 + q) * 2] = sx[q]; sm[((rl * 16 + gl) * 8 + q) * 2 + 1] = sq[q]; }
    __syncthreads();
    double mt = 0.0, ts = 0.0;
    if (threadIdx.x < 128) {                             // one thread per column of the chunk
        const int cl = threadIdx.x, a = blockIdx.y * 128 + cl;
        if (a < d) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int l = 0; l < 16; ++l) { s1 += sm[((l * 16 + (cl >> 3)) * 8 + (cl & 7)) * 2]; s2 += sm[((l * 16 + (cl >> 3)) * 8 + (cl & 7)) * 2 + 1]; }
            const double first = (n > 0) ? ld_f64<r_f16>(reinterpret_cast<const r_f16*>(rows), r0 * ld + a) : 0.0;
            const double m = (n > 0) ? first + s1 / (double)n : 0.0;
            const double mr = mean_mode ? mean_like_reference<r_f16>(reinterpret_cast<const r_f16*>(rows), ld, a, r0, r1, m, runs == kMeanPlaceholder ? runs : (runs ? runs + s * d + a : nullptr)) : m;
            if (mean_exact) mean_exact[s * d + a] = m;
            const double df = mu_b[a] - mr;
            mt = df * df;
            ts = (n > 0) ? s2 - (s1 * s1) / (double)n : 0.0;
            if (var_exact) var_exact[s * d + a] = (n > 1) ? ts / (double)(n - 1) : 0.0;      // the diagonal of Sigma_s, exact
        }
    }
    mt = block_sum(mt, red);
    ts = block_sum(ts, red);
    if (threadIdx.x == 0) {
        double* o = out + 2 * (s * gridDim.y + blockIdx.y);
        o[0] = mt; o[1] = (n > 1) ? ts / (double)(n - 1) : 0.0;
    }
}

// ||mu_b - mean||^2 per song with the mean as np.mean forms it, from the song's float32 running column sums (segment_running_sums_launch)
template <typename TIn>
__global__ __launch_bounds__(256) void song_mean_terms_from_runs(const float* __restrict__ runs, const int64_t* __restrict__ offsets, int d,
                                                                 const double* __restrict__ mu_b, double* __restrict__ mt_out) {
    __shared__ double red[4];
    const int64_t s = blockIdx.x;
    const double n = (double)(offsets[s + 1] - offsets[s]);
    double mt = 0.0;
    if (n > 0.0)
        for (int a = threadIdx.x; a < d; a += 256) {
            const double df = mu_b[a] - round_like_input<TIn>(numpy_mean_of_f32_sum(runs[s * d + a], n));
            mt += df * df;
        }
    mt = block_sum(mt, red);
    if (threadIdx.x == 0) mt_out[s] = mt;
}

__global__ __launch_bounds__(256) void song_scal_sum(const double* __restrict__ part, int chunks, int64_t n_songs,
                                                     double* __restrict__ scal /*[S][2]*/) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (song, which scalar)
    if (e >= 2 * n_songs) return;
    const int64_t s = e >> 1; const int w = (int)(e & 1);
    double t = 0.0;
    for (int c = 0; c < chunks; ++c) t += part[2 * (s * chunks + c) + w];
    scal[e] = t;
}

// Sigma_s = Xc^T Xc / (n-1) with the exact fp64 mean (np.cov), 16x16 threads per 16x16 tile; grid (t, t, songs)
template <typename TIn>
__global__ __launch_bounds__(256) void song_cov(const TIn* __restrict__ rows, int64_t ld, int d,
                                                const int64_t* __restrict__ offsets, const int64_t* __restrict__ song_ids,
                                                const double* __restrict__ mean_exact, double* __restrict__ cov_out) {
    const int64_t slot = blockIdx.z;
    const int64_t s = song_ids[slot];
    const int a = blockIdx.y * 16 + (threadIdx.x >> 4), b = blockIdx.x * 16 + (threadIdx.x & 15);
    if (a >= d || b >= d) return;
    const int64_t r0 = offsets[s], r1 = offsets[s + 1];
    const double ma = mean_exact[s * d + a], mb = mean_exact[s * d + b];
    double acc = 0.0;
    for (int64_t r = r0; r < r1; ++r)
        acc += (ld_f64<TIn>(rows, r * ld + a) - ma) * (ld_f64<TIn>(rows, r * ld + b) - mb);
    cov_out[slot * (int64_t)d * d + (int64_t)a * d + b] = acc / (double)(r1 - r0 - 1);
}

// The same on the fp64 MFMA for songs of many frames (Encodec: [2250 x 128] per song, CLAP: hundreds x 512): one workgroup per
// upper-triangular 64 x 64 tile of one song's covariance, 4 waves as 2 x 2 each owning 2 x 2 v_mfma_f64_16x16x4_f64 tiles,
// 16-row stages of (x - mean) staged through LDS as doubles (the layout of moments_tile_f64, moments_kernels.h), both
// triangles written.  The scalar kernel above ran at ~4 TFLOP/s and was half of the D x D route's time at those shapes
// (scripts/probe_songs_general.py).  grid (tiles, 1, songs).
constexpr int SC_LDS = 80;                 // padded row pitch (doubles), as G_LDS of the moments kernels
template <typename TIn>
__global__ __launch_bounds__(256) void song_cov_mfma(const TIn* __restrict__ rows, int64_t ld, int d, int nt,
                                                     const int64_t* __restrict__ offsets, const int64_t* __restrict__ song_ids,
                                                     const double* __restrict__ mean_exact, double* __restrict__ cov_out) {
    __shared__ double smem[2][2][16 * SC_LDS];      // [buffer][A | B][row][col]
    const int64_t slot = blockIdx.z;
    const int64_t s = song_ids ? song_ids[slot] : slot;
    int ta = 0, t = blockIdx.x;
    while (t >= nt - ta) { t -= nt - ta; ++ta; }
    const int tb = ta + t;
    const bool diag = ta == tb;
    const int ca = ta * 64, cb = tb * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t r0 = offsets[s], r1 = offsets[s + 1];
    const int nkb = (int)((r1 - r0 + 15) / 16);
    const double* mean = mean_exact ? mean_exact + s * d : nullptr;          // nullptr: the rows are centred already
    const int sr = tid >> 4, sc4 = (tid & 15) * 4;
    double ma[4], mb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ma[q] = (mean && ca + sc4 + q < d) ? mean[ca + sc4 + q] : 0.0;
        mb[q] = (mean && cb + sc4 + q < d) ? mean[cb + sc4 + q] : 0.0;
    }
    double ra[4], rb[4];
    auto fetch = [&](int kb) {
        const int64_t r = r0 + (int64_t)kb * 16 + sr;
        const bool ok = r < r1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int colA = ca + sc4 + q, colB = cb + sc4 + q;
            ra[q] = (ok && colA < d) ? ld_f64<TIn>(rows, r * ld + colA) - ma[q] : 0.0;
            if (!diag) rb[q] = (ok && colB < d) ? ld_f64<TIn>(rows, r * ld + colB) - mb[q] : 0.0;
        }
    };
    f64x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f64x4){0.0, 0.0, 0.0, 0.0};
    if (nkb > 0) fetch(0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            smem[buf][0][sr * SC_LDS + sc4 + q] = ra[q];
            if (!diag) smem[buf][1][sr * SC_LDS + sc4 + q] = rb[q];
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
                a[f] = sA[k * SC_LDS + 32 * wr + 16 * f + li];
                b[f] = sB[k * SC_LDS + 32 * wc + 16 * f + li];
            }
#pragma unroll
            for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
                    acc[fa][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
        }
    }
    const double inv = 1.0 / (double)(r1 - r0 - 1);
    double* out = cov_out + slot * (int64_t)d * d;
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int a_ = ca + 32 * wr + 16 * fa + lk + 4 * reg;
                const int b_ = cb + 32 * wc + 16 * fb + li;
                if (a_ < d && b_ < d) {
                    const double v = acc[fa][fb][reg] * inv;
                    out[(int64_t)a_ * d + b_] = v;
                    if (!diag) out[(int64_t)b_ * d + a_] = v;
                }
            }
}

// Two-frame songs through the batched GEMM: Dm[r] = x1 - x2 (fp64, exact), W = Dm Sigma_b (rows packed D at a
// time, Sigma_b shared), q[r] = W[r] . Dm[r].  (Round 1's 16-songs-per-workgroup kernel re-read all of Sigma_b per
// workgroup and ran at ~4 TFLOP/s; this product runs at the fp64 MFMA ceiling: 12.7 GFLOP in 250 us at config 5.)
// pair_stats_diff reads the two frames ONCE: the difference row for the product and the song's scalars (||mu_b - mean||^2 with
// the mean as the reference sees it, tr Sigma_s) -- as a separate statistics kernel plus a difference kernel the rows were read
// twice and two S x D float64 mean arrays nobody needed were written (63 us of a 340 us chain at config 5).
template <typename TIn>
__global__ __launch_bounds__(256) void pair_stats_diff(const TIn* __restrict__ rows, int64_t ld, int d,
                                                       const int64_t* __restrict__ offsets, const int64_t* __restrict__ song_ids,
                                                       int64_t n_pairs, const double* __restrict__ mu_b, int mean_mode,
                                                       double* __restrict__ dm, double* __restrict__ scal /*[S][2]*/) {
    __shared__ double red[4];
    const int64_t r = blockIdx.x;
    if (r >= n_pairs) {                                             // padding rows of the last D-row problem
        for (int a = threadIdx.x; a < d; a += 256) dm[r * d + a] = 0.0;
        return;
    }
    const int64_t s = song_ids ? song_ids[r] : r;
    const int64_t r0 = offsets[s];
    double mt = 0.0, ts = 0.0;
    for (int a = threadIdx.x; a < d; a += 256) {
        const double x1 = ld_f64<TIn>(rows, r0 * ld + a), x2 = ld_f64<TIn>(rows, (r0 + 1) * ld + a);
        dm[r * d + a] = x1 - x2;
        const double m = (x1 + x2) / 2.0;
        const double mr = mean_mode ? mean_like_reference<TIn>(rows, ld, a, r0, r0 + 2, m) : m;      // (two frames: float32(x1 + x2) / 2, rounded)
        const double c1 = x1 - m, c2 = x2 - m;
        ts += c1 * c1 + c2 * c2;
        const double df = mu_b[a] - mr;
        mt += df * df;
    }
    mt = block_sum(mt, red);
    ts = block_sum(ts, red);
    if (threadIdx.x == 0) { scal[2 * s] = mt; scal[2 * s + 1] = ts; }
}

// t[r] = W[r] . Dm[r] (W = Dm U, so t = q / 2), and with it the song's score: mean term + tr Sigma_b + tr Sigma_s - 2 sqrt(q / 2)
__global__ __launch_bounds__(256) void pair_rowdot_score(const double* __restrict__ w, const double* __restrict__ dm, int d,
                                                         const int64_t* __restrict__ song_ids, const double* __restrict__ scal,
                                                         const double* __restrict__ tr_b, double* __restrict__ score) {
    __shared__ double red[4];
    const int64_t r = blockIdx.x;
    double t = 0.0;
    for (int a = threadIdx.x; a < d; a += 256) t += w[r * d + a] * dm[r * d + a];
    t = block_sum(t, red);
    if (threadIdx.x == 0) {
        const int64_t s = song_ids ? song_ids[r] : r;
        const double root = t > 0.0 ? sqrt(t) : 0.0;             // t = d^T U d = (d^T Sigma_b d) / 2
        score[s] = (t == t) ? scal[2 * s] + *tr_b + scal[2 * s + 1] - 2.0 * root : t;
    }
}

// d^T S d = 2 d^T U d  with  U = strict upper triangle of (S + S^T)/2 plus half its diagonal: the product W = Dm U then skips
// the zero half of U (gemm b_upper) -- 13/24 of the flops of Dm S at D = 768, and it is the flops that bound this route.
__global__ __launch_bounds__(256) void upper_half(const double* __restrict__ m, int d, double* __restrict__ u) {
    const int64_t i = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += 256) {
        const double v = 0.5 * (m[i * d + j] + m[(int64_t)j * d + i]);
        u[i * d + j] = (j > i) ? v : (j == i ? 0.5 * v : 0.0);
    }
}

__global__ __launch_bounds__(256) void diag_trace(const double* __restrict__ m, int d, double* __restrict__ out) {
    __shared__ double red[4];
    double t = 0.0;
    for (int a = threadIdx.x; a < d; a += 256) t += m[(size_t)a * d + a];
    t = block_sum(t, red);
    if (threadIdx.x == 0) *out = t;
}

// ------------------------------------------------------------------------------------------
// Songs with 3 <= n <= 64 frames (n - 1 < D): the non-zero eigenvalues of Sigma_b Sigma_s equal those of the
// n x n Gram matrix  G = Xc Sigma_b Xc^T / (n - 1)  (Xc = centred frames), so
//     tr sqrt(Sigma_b Sigma_s) = sum_i sqrt(lambda_i(G)).
// W = Xc Sigma_b for ALL such songs is one batched fp64 MFMA GEMM (rows packed D at a time against the shared,
// L2-resident Sigma_b); one workgroup per song then forms G = W Xc^T in LDS and diagonalises it with a parallel
// cyclic Jacobi (round-robin pairs; eigenvalues only).  Replaces a D x D matrix root per song.
// ------------------------------------------------------------------------------------------
constexpr int GRAM_MAX = 64;

template <typename TIn>
__global__ __launch_bounds__(256) void gram_center_rows(const TIn* __restrict__ rows, int64_t ld, int d,
                                                        const int64_t* __restrict__ src_row, const int64_t* __restrict__ row_song,
                                                        const double* __restrict__ mean_exact, int64_t n_rows,
                                                        double* __restrict__ xc) {
    const int64_t r = blockIdx.x;
    const bool live = r < n_rows;
    const int64_t sr = live ? src_row[r] : 0, song = live ? row_song[r] : 0;
    for (int a = threadIdx.x; a < d; a += 256)
        xc[r * d + a] = live ? ld_f64<TIn>(rows, sr * ld + a) - mean_exact[song * d + a] : 0.0;      // pad rows are zero
}

__global__ __launch_bounds__(256) void gram_eig(const double* __restrict__ xc, const double* __restrict__ w, int d,
                                                const int64_t* __restrict__ first_row, const int* __restrict__ n_rows,
                                                double* __restrict__ tr_sqrt_out) {
    __shared__ double G[GRAM_MAX][GRAM_MAX + 1];
    __shared__ double cs[GRAM_MAX / 2][2];
    __shared__ int pq[GRAM_MAX / 2][2];
    __shared__ int perm[GRAM_MAX];
    __shared__ double red[4];
    const int song = blockIdx.x, tid = threadIdx.x;
    const int n = n_rows[song];
    const int m = (n + 1) & ~1;                            // even size; an odd n gets one zero row/column
    const int64_t r0 = first_row[song];
    const double inv = 1.0 / (double)(n - 1);

    for (int e = tid; e < m * m; e += 256) {               // G = W Xc^T / (n-1), symmetrised
        const int i = e / m, j = e % m;
        double acc = 0.0;
        if (i < n && j < n) {
            const double* wi = w + (r0 + i) * d;
            const double* xj = xc + (r0 + j) * d;
            const double* wj = w + (r0 + j) * d;
            const double* xi = xc + (r0 + i) * d;
            double a0 = 0.0, a1 = 0.0;
            for (int k = 0; k < d; ++k) { a0 += wi[k] * xj[k]; a1 += wj[k] * xi[k]; }
            acc = 0.5 * (a0 + a1) * inv;
        }
        G[i][j] = acc;
    }
    if (tid < m) perm[tid] = tid;
    __syncthreads();

    const int half = m / 2;
    for (int sweep = 0; sweep < 30; ++sweep) {
        // off-diagonal mass relative to the diagonal decides convergence
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < m * m; e += 256) {
            const int i = e / m, j = e % m;
            const double v = G[i][j];
            if (i == j) dia += v * v; else off += v * v;
        }
        off = block_sum(off, red);
        dia = block_sum(dia, red);
        if (off <= 1e-30 * dia || off == 0.0) break;
        for (int round = 0; round < m - 1; ++round) {
            if (tid < half) {                              // rotation for pair (p, q) of this round
                int p = perm[tid], q = perm[m - 1 - tid];
                if (p > q) { const int t = p; p = q; q = t; }
                const double app = G[p][p], aqq = G[q][q], apq = G[p][q];
                double c = 1.0, sn = 0.0;
                if (fabs(apq) > 1e-300) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t); sn = t * c;
                }
                pq[tid][0] = p; pq[tid][1] = q; cs[tid][0] = c; cs[tid][1] = sn;
            }
            __syncthreads();
            for (int e = tid; e < half * m; e += 256) {    // columns p, q of every row:  G <- G J
                const int k = e / m, i = e % m;
                const int p = pq[k][0], q = pq[k][1];
                const double c = cs[k][0], sn = cs[k][1];
                const double gip = G[i][p], giq = G[i][q];
                G[i][p] = c * gip - sn * giq;
                G[i][q] = sn * gip + c * giq;
            }
            __syncthreads();
            for (int e = tid; e < half * m; e += 256) {    // rows p, q of every column:  G <- J^T G
                const int k = e / m, j = e % m;
                const int p = pq[k][0], q = pq[k][1];
                const double c = cs[k][0], sn = cs[k][1];
                const double gpj = G[p][j], gqj = G[q][j];
                G[p][j] = c * gpj - sn * gqj;
                G[q][j] = sn * gpj + c * gqj;
            }
            if (tid == 0) {                                // round-robin: position 0 stays, the rest rotate
                const int last = perm[m - 1];
                for (int k = m - 1; k > 1; --k) perm[k] = perm[k - 1];
                perm[1] = last;
            }
            __syncthreads();
        }
    }
    double t = 0.0;
    for (int i = tid; i < m; i += 256) { const double lam = G[i][i]; t += lam > 0.0 ? sqrt(lam) : 0.0; }
    t = block_sum(t, red);
    if (tid == 0) tr_sqrt_out[song] = t;
}
}  // namespace fad

using namespace fad;


// ------------------------------------------------------------------------------------------
// Songs with 64 < n <= D frames (n - 1 < D): the same Gram identity, but an n x n matrix no longer fits one workgroup's LDS.
// G = W Xc^T / (n - 1) is formed by a batched MFMA kernel and its root trace comes from the batched Newton-Schulz iteration
// on n_pad x n_pad problems (n_pad = the sub-batch's longest song, rounded up to 64) instead of on the rank-deficient D x D
// product Sigma_b Sigma_s (10-second clips of a 50-frames-per-second D = 768 model: 499 frames).  G itself is singular -- the
// centred frames sum to zero, G 1 = 0 -- so the iteration runs on
//     G' = diag(G + (alpha / n) 1 1^T,  alpha I_pad),     alpha = tr G / n,
// whose extra eigenvalues are exactly alpha (1 is an exact null vector of G):  tr sqrt(G) = tr sqrt(G') - (1 + pad) sqrt(alpha).
// ------------------------------------------------------------------------------------------
constexpr int GRAM_TR_PARTS = 16;
__global__ __launch_bounds__(256) void gram_trace(const double* __restrict__ xc, const double* __restrict__ w, int d,
                                                  const int64_t* __restrict__ first_row, const int* __restrict__ n_rows,
                                                  double* __restrict__ tr_part /*[songs][GRAM_TR_PARTS]*/) {
    __shared__ double red[4];
    const int64_t k = blockIdx.x;
    const int64_t base = first_row[k] * d, len = (int64_t)n_rows[k] * d;
    const int64_t per = (len + GRAM_TR_PARTS - 1) / GRAM_TR_PARTS, e0 = blockIdx.y * per, e1 = (e0 + per < len) ? e0 + per : len;
    double t = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) t += w[base + e] * xc[base + e];
    t = block_sum(t, red);
    if (threadIdx.x == 0) tr_part[k * GRAM_TR_PARTS + blockIdx.y] = t;
}
// tr G of one song from its partials -- the same sum, in the same order, on the device (gram_big) and on the host
__host__ __device__ inline double gram_trace_total(const double* part, int n) {
    double t = 0.0;
    for (int q = 0; q < GRAM_TR_PARTS; ++q) t += part[q];
    return t / (double)(n - 1);
}

// grid (np/64, np/64, songs): one 64 x 64 tile of G' per workgroup, four waves of 32 x 32 on v_mfma_f64_16x16x4_f64, 16-deep k stages
__global__ __launch_bounds__(256) void gram_big(const double* __restrict__ xc, const double* __restrict__ w, int d, int np,
                                                const int64_t* __restrict__ first_row, const int* __restrict__ n_rows,
                                                const double* __restrict__ tr_part, double* __restrict__ gout) {
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    constexpr int P = 18;                                   // LDS pitch (doubles)
    __shared__ double sA[64 * P], sB[64 * P];
    const int64_t k = blockIdx.z;
    const int n = n_rows[k];
    const int64_t f = first_row[k];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;
    const double trg = gram_trace_total(tr_part + k * GRAM_TR_PARTS, n);
    const bool dead = !(trg > 0.0);                         // no spread at all (or not finite): the host scores it without a root
    const double alpha = dead ? 1.0 : trg / (double)n;
    double* G = gout + k * (int64_t)np * np;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    if (!dead && row0 < n && col0 < n) {
        const int lr = tid >> 2, lc = (tid & 3) * 4;        // this thread stages 4 consecutive k of one row of each operand
        const bool okA = row0 + lr < n, okB = col0 + lr < n;
        const double* pa = w + (f + row0 + lr) * d;
        const double* pb = xc + (f + col0 + lr) * d;
        for (int k0 = 0; k0 < d; k0 += 16) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = k0 + lc + q;
                sA[lr * P + lc + q] = (okA && kk < d) ? pa[kk] : 0.0;
                sB[lr * P + lc + q] = (okB && kk < d) ? pb[kk] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                double a[2], b[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a[q] = sA[(wr * 32 + 16 * q + li) * P + ks * 4 + lk];
                    b[q] = sB[(wc * 32 + 16 * q + li) * P + ks * 4 + lk];
                }
#pragma unroll
                for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb)
                        acc[fa][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
            }
        }
    }
    const double inv = 1.0 / (double)(n - 1), shift = alpha / (double)n;
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = row0 + wr * 32 + 16 * fa + lk + 4 * reg, c = col0 + wc * 32 + 16 * fb + li;
                double v;
                if (dead) v = (r == c) ? 1.0 : 0.0;
                else if (r < n && c < n) v = acc[fa][fb][reg] * inv + shift;
                else v = (r == c) ? alpha : 0.0;
                G[(int64_t)r * np + c] = v;
            }
}

// sqrt(A) of problem 0 of a finished iteration: sqrt(c) Y[final_iter & 1], symmetrised
__global__ __launch_bounds__(256) void root_from_state(const NsState* __restrict__ st, const double* __restrict__ y0,
                                                       const double* __restrict__ y1, int d, double* __restrict__ out) {
    const double* y = (st->final_iter & 1) ? y1 : y0;
    const double sc = sqrt(st->c);
    const int64_t i = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += 256) out[i * d + j] = 0.5 * sc * (y[i * d + j] + y[(int64_t)j * d + i]);
}

__global__ __launch_bounds__(256) void identity_and_zeros(double* __restrict__ eye, int np, double* __restrict__ zeros) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < (int64_t)np * np) eye[e] = (e / np == e % np) ? 1.0 : 0.0;
    if (e < np) zeros[e] = 0.0;
}

// ------------------------------------------------------------------------------------------
namespace fad {

template <typename TIn>
static int batched_core(int d, const double* dmu_b, const double* dcov_b, const TIn* drows, int64_t ld,
                        const int64_t* h_off, const int64_t* d_off, int64_t n_songs, int mean_mode, int device,
                        hipStream_t st, Workspace& ws, const SongKnobs& knobs, double* out_scores, int32_t* out_status, bool defer_means) {
    const int64_t dd = (int64_t)d * d;
    std::vector<int64_t> pairs, gram, gram_ns, general;
    const bool gram_on = knobs.gram;
    for (int64_t s = 0; s < n_songs; ++s) {
        const int64_t n = h_off[s + 1] - h_off[s];
        if (n < 2) { out_status[s] = FAD_ERR_TOO_FEW_ROWS; out_scores[s] = __builtin_nan(""); }
        else if (n == 2) { out_status[s] = FAD_OK; pairs.push_back(s); }
        else if (gram_on && n <= GRAM_MAX && n - 1 < d) { out_status[s] = FAD_OK; gram.push_back(s); }
        else if (gram_on && n - 1 < d) { out_status[s] = FAD_OK; gram_ns.push_back(s); }
        else { out_status[s] = FAD_OK; general.push_back(s); }
    }
    const bool others = !gram.empty() || !gram_ns.empty() || !general.empty();

    // ---- per-song scalars and means
    // songbuf: scal [S*2] | score [S] | tr_b [1] | ids (int64) [S] | mean_exact [S*d] (only when a song has more than two frames)
    FAD_TRY(ws.songbuf.reserve(((size_t)n_songs * 4 + 2 + (others ? (size_t)2 * n_songs * d : 0)) * sizeof(double) + 64));
    double* scal = static_cast<double*>(ws.songbuf.p);
    double* score_dev = scal + 2 * (size_t)n_songs;
    double* trb_dev = score_dev + n_songs;
    int64_t* ids_dev = reinterpret_cast<int64_t*>(trb_dev + 1);
    double* mean_exact = others ? reinterpret_cast<double*>(ids_dev + n_songs) : nullptr;
    double* var_exact = nullptr;                         // [S * d] the exact variances: only the one-pass float16 statistics kernel leaves them
    hipLaunchKernelGGL(diag_trace, dim3(1), dim3(256), 0, st, dcov_b, d, trb_dev);
    std::vector<double> h_scal;
    double tr_b = 0.0;
    if (others) {                       // (two-frame songs get their scalars from pair_stats_diff)
        const bool stats16_on = knobs.stats16;
        // songs of many frames: numpy's float32 running sums per song in a kernel of their own (the reference's per-song mean, fad.py:377)
        // -- on a stream of its own when the caller (batched_impl) defers the mean terms
        const float* runs = defer_means ? kMeanPlaceholder : nullptr;
        if (!defer_means && mean_mode && !std::is_same<TIn, double>::value && (h_off[n_songs] - h_off[0]) / n_songs >= 64) {
            FAD_TRY(ws.songrun.reserve((size_t)n_songs * d * sizeof(float)));
            const int code = std::is_same<TIn, r_f16>::value ? FAD_F16 : (std::is_same<TIn, r_bf16>::value ? FAD_BF16 : FAD_F32);
            FAD_TRY(segment_running_sums_launch(drows, ld, d, code, d_off, n_songs, static_cast<float*>(ws.songrun.p), st, &ws.songjobs,
                                                (h_off[n_songs] - h_off[0]) / n_songs, device));
            runs = static_cast<const float*>(ws.songrun.p);
        }
        if (std::is_same<TIn, r_f16>::value && stats16_on && (h_off[n_songs] - h_off[0]) / n_songs >= 64 && song_cov_f16_ok(drows, ld, d)) {
            const int chunks = (int)cdiv(d, 128);
            double* part = scal;
            if (chunks > 1) { FAD_TRY(ws.rows2.reserve((size_t)2 * n_songs * chunks * sizeof(double))); part = static_cast<double*>(ws.rows2.p); }
            hipLaunchKernelGGL(song_stats_f16, dim3((unsigned)n_songs, (unsigned)chunks), dim3(256), 0, st,
                               reinterpret_cast<const uint16_t*>(drows), ld, d, d_off, dmu_b, mean_mode, mean_exact,
                               mean_exact + (size_t)n_songs * d, part, runs);
            var_exact = mean_exact + (size_t)n_songs * d;
            if (chunks > 1) hipLaunchKernelGGL(song_scal_sum, dim3((unsigned)cdiv(2 * n_songs, 256)), dim3(256), 0, st, part, chunks, n_songs, scal);
        } else if ((h_off[n_songs] - h_off[0]) / n_songs >= 64) {
            const int chunks = (int)cdiv(d, 64);
            FAD_TRY(ws.rows2.reserve((size_t)2 * n_songs * chunks * sizeof(double)));
            double* part = static_cast<double*>(ws.rows2.p);
            hipLaunchKernelGGL((song_stats_long<TIn>), dim3((unsigned)n_songs, (unsigned)chunks), dim3(256), 0, st, drows, ld, d, d_off,
                               dmu_b, mean_mode, mean_exact, part, runs);
            hipLaunchKernelGGL(song_scal_sum, dim3((unsigned)cdiv(2 * n_songs, 256)), dim3(256), 0, st, part, chunks, n_songs, scal);
        } else
            hipLaunchKernelGGL((song_stats<TIn>), dim3((unsigned)n_songs), dim3(256), 0, st, drows, ld, d, d_off, dmu_b,
                               mean_mode, mean_exact, scal);
        h_scal.resize((size_t)2 * n_songs);
        FAD_HIP_TRY(hipMemcpyAsync(h_scal.data(), scal, h_scal.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        FAD_HIP_TRY(hipMemcpyAsync(&tr_b, trb_dev, sizeof(double), hipMemcpyDeviceToHost, st));
        FAD_HIP_TRY(hipStreamSynchronize(st));       // here, not later: an error return below must not leave a copy into this frame in flight
    }

    // ---- two-frame songs: closed form  tr sqrt = sqrt(d^T Sigma_b d / 2); the score is finished on the device and comes
    // back as ONE copy into pinned memory (the scalars, the row dots and the diagonal used to travel separately)
    if (!pairs.empty()) {
        const bool identity = (int64_t)pairs.size() == n_songs;            // every song has two frames: row r IS song r
        const int64_t budget_rows = std::max<int64_t>(d, ((int64_t)1 << 30) / ((int64_t)d * 16));
        for (size_t p0 = 0; p0 < pairs.size(); p0 += (size_t)budget_rows) {
            const int64_t P = (int64_t)std::min<size_t>((size_t)budget_rows, pairs.size() - p0);
            const int64_t nb = cdiv(P, d), Ppad = nb * d;
            FAD_TRY(ws.songmat.reserve(((size_t)2 * Ppad * d + dd) * sizeof(double)));
            double* dm = static_cast<double*>(ws.songmat.p);
            double* wmat = dm + (size_t)Ppad * d;
            double* uhalf = wmat + (size_t)Ppad * d;
            hipLaunchKernelGGL(upper_half, dim3((unsigned)d), dim3(256), 0, st, dcov_b, d, uhalf);
            const int64_t* d_ids = nullptr;
            if (!identity) {
                FAD_HIP_TRY(hipMemcpyAsync(ids_dev + p0, pairs.data() + p0, P * sizeof(int64_t), hipMemcpyHostToDevice, st));
                d_ids = ids_dev + p0;
            }
            const int64_t* d_off_chunk = identity ? d_off + p0 : d_off;     // identity: chunk row r is song p0 + r
            double* scal_chunk = identity ? scal + 2 * p0 : scal;
            double* score_chunk = identity ? score_dev + p0 : score_dev;
            hipLaunchKernelGGL((pair_stats_diff<TIn>), dim3((unsigned)Ppad), dim3(256), 0, st, drows, ld, d, d_off_chunk, d_ids, P,
                               dmu_b, mean_mode, dm, scal_chunk);
            GemmType gt{dm, dd, uhalf, 0, wmat, dd, 1.0, 0.0, 0.0, nullptr, 1};
            const int rc = gemm_f64_launch(d, &gt, 1, nb, nullptr, 0, st, device);
            if (rc < 0) return rc;
            hipLaunchKernelGGL(pair_rowdot_score, dim3((unsigned)P), dim3(256), 0, st, wmat, dm, d, d_ids, scal_chunk, trb_dev,
                               score_chunk);
        }
        double* h_score = static_cast<double*>(ws.song_pin) + (n_songs + 1);       // behind the offsets (reserved by the caller)
        FAD_HIP_TRY(hipMemcpyAsync(h_score, score_dev, (size_t)n_songs * sizeof(double), hipMemcpyDeviceToHost, st));
        FAD_HIP_TRY(hipStreamSynchronize(st));
        for (const int64_t s : pairs) {
            const double v = h_score[s];
            out_scores[s] = v;
            if (!(v == v)) out_status[s] = FAD_ERR_NOT_FINITE;
        }
    }
    // ---- songs with 3..64 frames: n x n Gram matrix + Jacobi eigenvalues
    if (!gram.empty()) {
        const int64_t budget_rows = std::max<int64_t>(d, ((int64_t)1 << 30) / ((int64_t)d * 16));     // ~1 GiB of Xc + W
        size_t g0 = 0;
        while (g0 < gram.size()) {
            std::vector<int64_t> src_row, row_song, first_row;
            std::vector<int> nrows;
            size_t g1 = g0;
            while (g1 < gram.size()) {
                const int64_t sg = gram[g1], n = h_off[sg + 1] - h_off[sg];
                if (!src_row.empty() && (int64_t)src_row.size() + n > budget_rows) break;
                first_row.push_back((int64_t)src_row.size()); nrows.push_back((int)n);
                for (int64_t r = 0; r < n; ++r) { src_row.push_back(h_off[sg] + r); row_song.push_back(sg); }
                ++g1;
            }
            const int64_t R = (int64_t)src_row.size(), nb = cdiv(R, d), Rpad = nb * d, ns = (int64_t)(g1 - g0);
            // device scratch: xc [Rpad*d] | w [Rpad*d] | tr [ns]   and index arrays
            FAD_TRY(ws.songmat.reserve(((size_t)2 * Rpad * d + ns) * sizeof(double)));
            double* xc = static_cast<double*>(ws.songmat.p);
            double* wmat = xc + (size_t)Rpad * d;
            double* trs = wmat + (size_t)Rpad * d;
            FAD_TRY(ws.rows2.reserve(((size_t)2 * R + ns) * sizeof(int64_t) + (size_t)ns * sizeof(int) + 64));
            int64_t* d_src = static_cast<int64_t*>(ws.rows2.p);
            int64_t* d_song = d_src + R;
            int64_t* d_first = d_song + R;
            int* d_n = reinterpret_cast<int*>(d_first + ns);
            FAD_HIP_TRY(hipMemcpyAsync(d_src, src_row.data(), R * sizeof(int64_t), hipMemcpyHostToDevice, st));
            FAD_HIP_TRY(hipMemcpyAsync(d_song, row_song.data(), R * sizeof(int64_t), hipMemcpyHostToDevice, st));
            FAD_HIP_TRY(hipMemcpyAsync(d_first, first_row.data(), ns * sizeof(int64_t), hipMemcpyHostToDevice, st));
            FAD_HIP_TRY(hipMemcpyAsync(d_n, nrows.data(), ns * sizeof(int), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((gram_center_rows<TIn>), dim3((unsigned)Rpad), dim3(256), 0, st, drows, ld, d, d_src, d_song,
                               mean_exact, R, xc);
            GemmType gt{xc, dd, dcov_b, 0, wmat, dd, 1.0, 0.0, 0.0, nullptr};          // W = Xc Sigma_b, D rows per problem
            const int rc = gemm_f64_launch(d, &gt, 1, nb, nullptr, 0, st, device);
            if (rc < 0) return rc;
            hipLaunchKernelGGL(gram_eig, dim3((unsigned)ns), dim3(256), 0, st, xc, wmat, d, d_first, d_n, trs);
            std::vector<double> h_tr((size_t)ns);
            FAD_HIP_TRY(hipMemcpyAsync(h_tr.data(), trs, ns * sizeof(double), hipMemcpyDeviceToHost, st));
            FAD_HIP_TRY(hipStreamSynchronize(st));         // also keeps the host index vectors alive until the copies ran
            for (int64_t k = 0; k < ns; ++k) {
                const int64_t sg = gram[g0 + k];
                const double t = h_tr[k];
                if (!(t == t) || !(tr_b == tr_b) || t > 1e300) { out_status[sg] = FAD_ERR_NOT_FINITE; out_scores[sg] = __builtin_nan(""); continue; }
                out_scores[sg] = h_scal[2 * sg] + tr_b + h_scal[2 * sg + 1] - 2.0 * t;
            }
            g0 = g1;
        }
    }

    // ---- songs with 65..D frames: n x n Gram matrix + batched Newton-Schulz on it (see gram_big)
    if (!gram_ns.empty()) {
        const int64_t budget_rows = std::max<int64_t>(d, ((int64_t)1 << 30) / ((int64_t)d * 16));     // ~1 GiB of Xc + W
        const size_t budget_mats = (size_t)3 << 30;                                                   // Newton-Schulz matrices
        size_t g0 = 0;
        while (g0 < gram_ns.size()) {
            std::vector<int64_t> src_row, row_song, first_row;
            std::vector<int> nrows;
            size_t g1 = g0;
            int n_max = 0;
            while (g1 < gram_ns.size()) {
                const int64_t sg = gram_ns[g1], n = h_off[sg + 1] - h_off[sg];
                const int64_t np_try = cdiv(std::max<int64_t>(n_max, n), 64) * 64;
                if (!src_row.empty() && ((int64_t)src_row.size() + n > budget_rows ||
                                         (size_t)(g1 - g0 + 1) * 7 * np_try * np_try * sizeof(double) > budget_mats)) break;
                first_row.push_back((int64_t)src_row.size()); nrows.push_back((int)n);
                if ((int)n > n_max) n_max = (int)n;
                for (int64_t r = 0; r < n; ++r) { src_row.push_back(h_off[sg] + r); row_song.push_back(sg); }
                ++g1;
            }
            const int64_t R = (int64_t)src_row.size(), nb = cdiv(R, d), Rpad = nb * d, ns = (int64_t)(g1 - g0);
            const int np = (int)(cdiv(n_max, 64) * 64);
            const int64_t npp = (int64_t)np * np;
            // device scratch: xc [Rpad*d] | w [Rpad*d] | G' [ns*np*np] | I [np*np] | zeros [np] | tr G partials [ns*16]
            FAD_TRY(ws.songmat.reserve(((size_t)2 * Rpad * d + (size_t)(ns + 1) * npp + np + ns * GRAM_TR_PARTS) * sizeof(double)));
            double* xc = static_cast<double*>(ws.songmat.p);
            double* wmat = xc + (size_t)Rpad * d;
            double* gmat = wmat + (size_t)Rpad * d;
            double* eye = gmat + (size_t)ns * npp;
            double* zeros = eye + npp;
            double* trg = zeros + np;
            FAD_TRY(ws.rows2.reserve(((size_t)2 * R + ns) * sizeof(int64_t) + (size_t)ns * sizeof(int) + 64));
            int64_t* d_src = static_cast<int64_t*>(ws.rows2.p);
            int64_t* d_song = d_src + R;
            int64_t* d_first = d_song + R;
            int* d_n = reinterpret_cast<int*>(d_first + ns);
            FAD_HIP_TRY(hipMemcpyAsync(d_src, src_row.data(), R * sizeof(int64_t), hipMemcpyHostToDevice, st));
            FAD_HIP_TRY(hipMemcpyAsync(d_song, row_song.data(), R * sizeof(int64_t), hipMemcpyHostToDevice, st));
            FAD_HIP_TRY(hipMemcpyAsync(d_first, first_row.data(), ns * sizeof(int64_t), hipMemcpyHostToDevice, st));
            FAD_HIP_TRY(hipMemcpyAsync(d_n, nrows.data(), ns * sizeof(int), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((gram_center_rows<TIn>), dim3((unsigned)Rpad), dim3(256), 0, st, drows, ld, d, d_src, d_song,
                               mean_exact, R, xc);
            GemmType gt{xc, dd, dcov_b, 0, wmat, dd, 1.0, 0.0, 0.0, nullptr};          // W = Xc Sigma_b, D rows per problem
            const int rc = gemm_f64_launch(d, &gt, 1, nb, nullptr, 0, st, device);
            if (rc < 0) return rc;
            hipLaunchKernelGGL(gram_trace, dim3((unsigned)ns, GRAM_TR_PARTS), dim3(256), 0, st, xc, wmat, d, d_first, d_n, trg);
            hipLaunchKernelGGL(gram_big, dim3((unsigned)(np / 64), (unsigned)(np / 64), (unsigned)ns), dim3(256), 0, st, xc, wmat, d, np,
                               d_first, d_n, trg, gmat);
            hipLaunchKernelGGL(identity_and_zeros, dim3((unsigned)cdiv(npp, 256)), dim3(256), 0, st, eye, np, zeros);
            FAD_TRY(ws.small.reserve(ns_small_bytes(np, ns)));
            NsState* dstates = static_cast<NsState*>(ws.small.p);
            enqueue_clear_states(dstates, ns, st);
            NsState* hs = nullptr;
            const int sym_on = knobs.sym ? 1 : 0;
            NsProblem pb{np, ns, gmat, npp, eye, 0, zeros, 0, zeros, 0, -1, sym_on};    // A = G' I, symmetric like every iterate
            FAD_TRY(run_ns(pb, 0, 0.0, device, st, ws, &hs));                          // (synchronises: the index vectors may go)
            std::vector<double> h_trg((size_t)ns * GRAM_TR_PARTS);
            FAD_HIP_TRY(hipMemcpy(h_trg.data(), trg, h_trg.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < ns; ++k) {
                const int64_t sg = gram_ns[g0 + k];
                const double tg = gram_trace_total(h_trg.data() + k * GRAM_TR_PARTS, nrows[k]);
                if (!(tg == tg) || !(tr_b == tr_b) || tg > 1e300) { out_status[sg] = FAD_ERR_NOT_FINITE; out_scores[sg] = __builtin_nan(""); continue; }
                double tr_sqrt = 0.0;
                if (tg > 0.0) {
                    if (hs[k].nonfinite) { out_status[sg] = FAD_ERR_NOT_FINITE; out_scores[sg] = __builtin_nan(""); continue; }
                    const double alpha = tg / (double)nrows[k];
                    tr_sqrt = sqrt(hs[k].c) * hs[k].tr_last - (double)(1 + np - nrows[k]) * sqrt(alpha);
                    if (tr_sqrt < 0.0) tr_sqrt = 0.0;
                    if (hs[k].conv == 0) out_status[sg] = FAD_ERR_NOT_CONVERGED;
                }
                out_scores[sg] = h_scal[2 * sg] + tr_b + h_scal[2 * sg + 1] - 2.0 * tr_sqrt;
            }
            g0 = g1;
        }
    }

    // ---- songs of at least D + 1 frames, D in {128, 256, 384, 512, 768, 1024}: the eight-launch chain, batched over the songs (fast_songs);
    // whatever it does not accept falls through to the float64 routes below.  FAD_SONG_FAST=0 switches it off.
    // (read per call -- a batched call is milliseconds; 2 = strict: an error when the chain accepts NO song of the call, for tests)
    const int fastsongs_on = knobs.fast;
    if (fastsongs_on && fast_song_dim(d) && !general.empty() && tr_b == tr_b) {
        ws.pool = &thread_pool(device);
        std::vector<int64_t> rest;
        const int64_t sub = std::min<int64_t>(fast_songs_capacity(d, (size_t)3 << 30), (int64_t)general.size());
        FAD_TRY(ws.songmat.reserve((size_t)sub * dd * sizeof(double)));
        double* covs = static_cast<double*>(ws.songmat.p);
        const int nt64 = (int)cdiv(d, 64);
        std::vector<double> trs; std::vector<char> okv;
        for (size_t g0 = 0; g0 < general.size(); g0 += (size_t)sub) {
            const int64_t B = (int64_t)std::min<size_t>((size_t)sub, general.size() - g0);
            FAD_HIP_TRY(hipMemcpyAsync(ids_dev, general.data() + g0, B * sizeof(int64_t), hipMemcpyHostToDevice, st));
            // float16 frames: the covariances on the float16 matrix pipe, shifted by the song's mean (moments_kernels.h: song_cov_*;
            // FAD_SONG_COV16=0: the float64 MFMA kernel, as for every other dtype)
            const bool cov16_on = knobs.cov16;
            if (std::is_same<TIn, r_f16>::value && cov16_on && var_exact && song_cov_f16_ok(drows, ld, d)) {      // (only with the exact diagonal)
                int64_t max_frames = 0;
                for (int64_t b = 0; b < B; ++b) { const int64_t sg = general[g0 + b]; max_frames = std::max(max_frames, h_off[sg + 1] - h_off[sg]); }
                FAD_TRY(song_cov_f16_launch(drows, ld, d, d_off, ids_dev, B, max_frames, mean_exact, var_exact, covs, ws.songcov, device, st));
            } else {
                hipLaunchKernelGGL((song_cov_mfma<TIn>), dim3((unsigned)(nt64 * (nt64 + 1) / 2), 1, (unsigned)B), dim3(256), 0, st, drows,
                                   ld, d, nt64, d_off, ids_dev, mean_exact, covs);
            }
            FAD_TRY(fast_songs(d, B, dcov_b, covs, st, ws, trs, okv, device, knobs));       // (synchronises: `general` may be read again)
            for (int64_t b = 0; b < B; ++b) {
                const int64_t sg = general[g0 + b];
                if (okv[b]) out_scores[sg] = h_scal[2 * sg] + tr_b + h_scal[2 * sg + 1] - 2.0 * trs[b];
                else rest.push_back(sg);
            }
        }
        if (fastsongs_on == 2 && rest.size() == general.size())
            return set_error(FAD_ERR_INVALID, "FAD_SONG_FAST=2: the batched fast chain accepted none of %zu songs", general.size());
        general.swap(rest);
    }

    // ---- songs of D + 1 .. 8 D frames (D >= 64): the symmetric form of the D x D problem.  With B = sqrt(Sigma_b) (ONE
    // Newton-Schulz problem per call) the product Sigma_b Sigma_s is similar to B Sigma_s B = cov(Xc B), so the song's matrix is
    // the covariance of its transformed frames -- symmetric, like every iterate of its root, and the iteration's products skip
    // the mirrored tiles (GemmType::sym); the D x D product Sigma_b Sigma_s is never formed.  Costs one [n x D][D x D] product
    // per song: longer songs (Encodec: 2250 frames at D = 128) and baselines whose root does not converge keep the route below.
    const bool symroute_on = knobs.sym;
    if (symroute_on && d >= 64 && !general.empty()) {
        std::vector<int64_t> sym_songs, rest;
        const int64_t max_mult = knobs.sym_max_mult;
        for (const int64_t sg : general) ((h_off[sg + 1] - h_off[sg] <= max_mult * d) ? sym_songs : rest).push_back(sg);
        bool have_root = false;
        double *broot = nullptr, *eye = nullptr, *zeros = nullptr;
        if (!sym_songs.empty()) {
            FAD_TRY(ws.base_root.reserve((size_t)(2 * dd + d) * sizeof(double)));
            broot = static_cast<double*>(ws.base_root.p); eye = broot + dd; zeros = eye + dd;
            hipLaunchKernelGGL(identity_and_zeros, dim3((unsigned)cdiv(dd, 256)), dim3(256), 0, st, eye, d, zeros);
            FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
            NsState* dstates = static_cast<NsState*>(ws.small.p);
            enqueue_clear_states(dstates, 1, st);
            NsState* hsb = nullptr;
            double* yb[2] = {nullptr, nullptr};
            NsProblem pbb{d, 1, dcov_b, 0, eye, 0, zeros, 0, zeros, 0, -1, 0};
            FAD_TRY(run_ns(pbb, 0, 0.0, device, st, ws, &hsb, false, yb));
            have_root = hsb[0].conv == 1 && !hsb[0].nonfinite && hsb[0].final_iter >= 0;
            if (have_root) hipLaunchKernelGGL(root_from_state, dim3((unsigned)d), dim3(256), 0, st, dstates, yb[0], yb[1], d, broot);
        }
        if (have_root) {
            const int64_t budget_rows = std::max<int64_t>(d, ((int64_t)1 << 30) / ((int64_t)d * 16));     // ~1 GiB of Xc + Xc B
            const size_t budget_mats = (size_t)3 << 30;
            const int nt64 = (int)cdiv(d, 64);
            size_t g0 = 0;
            while (g0 < sym_songs.size()) {
                std::vector<int64_t> src_row, row_song, first_row;
                size_t g1 = g0;
                while (g1 < sym_songs.size()) {
                    const int64_t sg = sym_songs[g1], n = h_off[sg + 1] - h_off[sg];
                    if (!src_row.empty() && ((int64_t)src_row.size() + n > budget_rows ||
                                             (size_t)(g1 - g0 + 1) * 7 * dd * sizeof(double) > budget_mats || g1 - g0 >= 4096)) break;
                    first_row.push_back((int64_t)src_row.size());
                    for (int64_t r = 0; r < n; ++r) { src_row.push_back(h_off[sg] + r); row_song.push_back(sg); }
                    ++g1;
                }
                first_row.push_back((int64_t)src_row.size());
                const int64_t R = (int64_t)src_row.size(), nb = cdiv(R, d), Rpad = nb * d, B = (int64_t)(g1 - g0);
                // device scratch: xc [Rpad*d] | xc B [Rpad*d] | covariances [B*d*d]
                FAD_TRY(ws.songmat.reserve(((size_t)2 * Rpad * d + (size_t)B * dd) * sizeof(double)));
                double* xc = static_cast<double*>(ws.songmat.p);
                double* xp = xc + (size_t)Rpad * d;
                double* covs = xp + (size_t)Rpad * d;
                FAD_TRY(ws.rows2.reserve(((size_t)2 * R + B + 1) * sizeof(int64_t) + 64));
                int64_t* d_src = static_cast<int64_t*>(ws.rows2.p);
                int64_t* d_song = d_src + R;
                int64_t* d_first = d_song + R;
                FAD_HIP_TRY(hipMemcpyAsync(d_src, src_row.data(), R * sizeof(int64_t), hipMemcpyHostToDevice, st));
                FAD_HIP_TRY(hipMemcpyAsync(d_song, row_song.data(), R * sizeof(int64_t), hipMemcpyHostToDevice, st));
                FAD_HIP_TRY(hipMemcpyAsync(d_first, first_row.data(), (B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL((gram_center_rows<TIn>), dim3((unsigned)Rpad), dim3(256), 0, st, drows, ld, d, d_src, d_song,
                                   mean_exact, R, xc);
                GemmType gt{xc, dd, broot, 0, xp, dd, 1.0, 0.0, 0.0, nullptr};              // Xc B, D rows per problem
                const int rc = gemm_f64_launch(d, &gt, 1, nb, nullptr, 0, st, device);
                if (rc < 0) return rc;
                hipLaunchKernelGGL((song_cov_mfma<double>), dim3((unsigned)(nt64 * (nt64 + 1) / 2), 1, (unsigned)B), dim3(256), 0, st, xp,
                                   (int64_t)d, d, nt64, d_first, (const int64_t*)nullptr, (const double*)nullptr, covs);
                FAD_TRY(ws.small.reserve(ns_small_bytes(d, B)));
                NsState* dstates = static_cast<NsState*>(ws.small.p);
                enqueue_clear_states(dstates, B, st);
                NsState* hs = nullptr;
                NsProblem pb{d, B, covs, dd, eye, 0, zeros, 0, zeros, 0, -1, 1};            // A = cov(Xc B) I
                FAD_TRY(run_ns(pb, 0, 0.0, device, st, ws, &hs));                          // (synchronises: the index vectors may go)
                for (int64_t b = 0; b < B; ++b) {
                    const int64_t sg = sym_songs[g0 + b];
                    if (!(tr_b == tr_b)) { out_status[sg] = FAD_ERR_NOT_FINITE; out_scores[sg] = __builtin_nan(""); continue; }
                    // This route sees the song through sqrt(Sigma_b), which Newton-Schulz delivers to ~1e-10, and an eigenvalue of the
                    // transformed covariance moves with that error divided by its own square root: only songs whose iteration shows
                    // a moderate spread keep the result (a start value (1.5)^-13 below 1 is lambda_min / c ~ 3e-5); the others --
                    // and whatever did not converge or overflowed here -- go on to the product route below, which forms
                    // Sigma_b Sigma_s itself.  (Round 3: a k^-3 spectrum at D = 768 came back 4e-5 off with status 0, at D = 1024
                    // as NaN; tests/test_gpu_parity.py: test_songs_full_rank_route_on_the_matrix_pipes.)
                    if (hs[b].nonfinite || hs[b].conv == 0 || hs[b].final_iter < 0 || hs[b].final_iter > kSymMaxIter) { rest.push_back(sg); continue; }
                    const double tr_sqrt = sqrt(hs[b].c) * hs[b].tr_last;
                    out_scores[sg] = h_scal[2 * sg] + tr_b + h_scal[2 * sg + 1] - 2.0 * tr_sqrt;
                }
                g0 = g1;
            }
            general.swap(rest);
        }
    }

    // ---- remaining songs: batched D x D Newton-Schulz against the shared baseline
    if (!general.empty()) {
        size_t budget = (size_t)3 << 30;                         // bytes of matrices per sub-batch
        int64_t sub = (int64_t)(budget / ((size_t)7 * dd * sizeof(double)));
        if (sub < 1) sub = 1;
        if (sub > 4096) sub = 4096;
        if (sub > (int64_t)general.size()) sub = (int64_t)general.size();
        FAD_TRY(ws.songmat.reserve((size_t)sub * dd * sizeof(double)));
        FAD_TRY(ws.small.reserve(ns_small_bytes(d, sub)));
        double* covs = static_cast<double*>(ws.songmat.p);
        const unsigned t16 = (unsigned)cdiv(d, 16);
        for (size_t g0 = 0; g0 < general.size(); g0 += (size_t)sub) {
            const int64_t B = (int64_t)std::min<size_t>((size_t)sub, general.size() - g0);
            FAD_HIP_TRY(hipMemcpyAsync(ids_dev, general.data() + g0, B * sizeof(int64_t), hipMemcpyHostToDevice, st));
            if (d >= 64) {
                const int nt64 = (int)cdiv(d, 64);
                hipLaunchKernelGGL((song_cov_mfma<TIn>), dim3((unsigned)(nt64 * (nt64 + 1) / 2), 1, (unsigned)B), dim3(256), 0, st, drows,
                                   ld, d, nt64, d_off, ids_dev, mean_exact, covs);
            } else {
                hipLaunchKernelGGL((song_cov<TIn>), dim3(t16, t16, (unsigned)B), dim3(256), 0, st, drows, ld, d, d_off,
                                   ids_dev, mean_exact, covs);
            }
            // gather the reference-rounded means of this sub-batch contiguously: reuse q area? keep simple:
            // mean_ref rows of the sub-batch are not contiguous, so run NS with mu2 = mu_b (mean term = 0)
            // and take the mean term from song_stats instead.
            NsState* dstates = static_cast<NsState*>(ws.small.p);
            enqueue_clear_states(dstates, B, st);
            NsState* hs = nullptr;
            NsProblem pb{d, B, dcov_b, 0, covs, dd, dmu_b, 0, dmu_b, 0, -1};
            FAD_TRY(run_ns(pb, 0, 0.0, device, st, ws, &hs));
            for (int64_t b = 0; b < B; ++b) {
                const int64_t s = general[g0 + b];
                if (hs[b].nonfinite) { out_status[s] = FAD_ERR_NOT_FINITE; out_scores[s] = __builtin_nan(""); continue; }
                const double tr_sqrt = sqrt(hs[b].c) * hs[b].tr_last;
                out_scores[s] = h_scal[2 * s] + hs[b].tr1 + hs[b].tr2 - 2.0 * tr_sqrt;
                if (hs[b].conv == 0) out_status[s] = FAD_ERR_NOT_CONVERGED;
            }
        }
    }
    return FAD_OK;
}

// Songs of many frames with the reference's own means (mean_mode = 1, 16-bit / float32 frames): the walk of numpy's running sums is a second
// pass over all frames (0.4 ms for 2000 x [2250 x 128]) that nothing but the MEAN TERM of a score waits for -- and a score is linear in it.
// So the walk and the mean terms run on the library's side stream beside the covariances and the square roots (MFMA work, the HBM idle), the
// statistics kernels take the rounded exact mean as a placeholder, and the difference of the two mean terms is added to the scores at the end.
template <typename TIn>
static int batched_impl(int d, const double* dmu_b, const double* dcov_b, const TIn* drows, int64_t ld,
                        const int64_t* h_off, const int64_t* d_off, int64_t n_songs, int mean_mode, int device,
                        hipStream_t st, Workspace& ws, const SongKnobs& knobs, double* out_scores, int32_t* out_status) {
    int64_t long_songs = 0;
    for (int64_t s = 0; s < n_songs; ++s) long_songs += (h_off[s + 1] - h_off[s]) > 2;
    hipStream_t side = moments_side_stream(device);
    const bool defer = mean_mode == 1 && !std::is_same<TIn, double>::value && long_songs > 0 && side != nullptr &&
                       (h_off[n_songs] - h_off[0]) / n_songs >= 64;
    if (!defer) return batched_core<TIn>(d, dmu_b, dcov_b, drows, ld, h_off, d_off, n_songs, mean_mode, device, st, ws, knobs, out_scores, out_status, false);
    FAD_TRY(ws.songrun.reserve((size_t)n_songs * d * sizeof(float) + (size_t)n_songs * sizeof(double) + 64));
    float* runs = static_cast<float*>(ws.songrun.p);
    double* mt_ref = reinterpret_cast<double*>(static_cast<char*>(ws.songrun.p) + (((size_t)n_songs * d * sizeof(float) + 63) & ~(size_t)63));
    hipEvent_t fork = nullptr, join = nullptr;
    FAD_HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    FAD_HIP_TRY(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    int rc = FAD_OK;
    do {
        if (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(side, fork, 0) != hipSuccess) { rc = set_error(FAD_ERR_HIP, "event fork failed"); break; }
        const int code = std::is_same<TIn, r_f16>::value ? FAD_F16 : (std::is_same<TIn, r_bf16>::value ? FAD_BF16 : FAD_F32);
        rc = segment_running_sums_launch(drows, ld, d, code, d_off, n_songs, runs, side, &ws.songjobs, (h_off[n_songs] - h_off[0]) / n_songs, device);
        if (rc != FAD_OK) break;
        hipLaunchKernelGGL((song_mean_terms_from_runs<TIn>), dim3((unsigned)n_songs), dim3(256), 0, side, runs, d_off, d, dmu_b, mt_ref);
        if (hipEventRecord(join, side) != hipSuccess) { rc = set_error(FAD_ERR_HIP, "event join failed"); break; }
        rc = batched_core<TIn>(d, dmu_b, dcov_b, drows, ld, h_off, d_off, n_songs, mean_mode, device, st, ws, knobs, out_scores, out_status, true);
        // the swap: score += ||mu_b - numpy's mean||^2 - ||mu_b - rounded exact mean||^2   (songs of more than two frames; the placeholders are the
        // statistics kernels' scal[2 s], still in the workspace)
        if (hipStreamWaitEvent(st, join, 0) != hipSuccess) { if (rc == FAD_OK) rc = set_error(FAD_ERR_HIP, "event wait failed"); break; }
        std::vector<double> h_ref((size_t)n_songs), h_scal((size_t)2 * n_songs);
        if (hipMemcpyAsync(h_ref.data(), mt_ref, h_ref.size() * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(h_scal.data(), ws.songbuf.p, h_scal.size() * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { if (rc == FAD_OK) rc = set_error(FAD_ERR_HIP, "copy of the mean terms failed"); break; }
        for (int64_t s = 0; s < n_songs; ++s) {
            if (h_off[s + 1] - h_off[s] <= 2) continue;                 // (two-frame songs took numpy's mean inside pair_stats_diff)
            if (!(out_scores[s] == out_scores[s])) continue;             // (not scored)
            out_scores[s] += h_ref[s] - h_scal[2 * s];
        }
    } while (false);
    if (rc != FAD_OK) (void)hipStreamSynchronize(side);                  // (nothing of this call may still be reading the frames when it returns an error)
    (void)hipEventDestroy(fork); (void)hipEventDestroy(join);
    return rc;
}

// (first-use warm-up, common.cpp: warm_code_objects -- loading this translation unit's code object costs ~75 ms at the first launch)
const void* code_object_anchor_frechet_songs() { return reinterpret_cast<const void*>(&song_stats_f16); }
}  // namespace fad

extern "C" int fad_frechet_batched_vs_baseline(int d, const double* mu_b, const double* cov_b,
                                               const void* rows, int64_t n_rows, int64_t ld, int dtype,
                                               const int64_t* offsets, int64_t n_songs, int mean_mode,
                                               int on_device, int device, void* stream,
                                               double* out_scores, int32_t* out_status) {
    if (d < 1 || d > 16384) return set_error(FAD_ERR_INVALID, "d=%d out of range", d);
    if (!mu_b || !cov_b || !offsets || !out_scores || !out_status || n_songs < 0 || n_rows < 0)
        return set_error(FAD_ERR_INVALID, "NULL or negative argument");
    if (ld < d) return set_error(FAD_ERR_SHAPE, "ld=%lld < d=%d", (long long)ld, d);
    if (dtype_size(dtype) == 0) return set_error(FAD_ERR_INVALID, "unknown dtype %d", dtype);
    for (int64_t s = 0; s < n_songs; ++s)
        if (offsets[s] > offsets[s + 1] || offsets[s] < 0 || offsets[s + 1] > n_rows)
            return set_error(FAD_ERR_INVALID, "offsets must be non-decreasing within [0, n_rows]");
    if (n_songs == 0) return FAD_OK;
    if (!rows && n_rows > 0) return set_error(FAD_ERR_INVALID, "rows is NULL");
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return set_error(FAD_ERR_HIP, "cannot select device %d", device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a slot that no score in flight owns: the kernels of fad_frechet_from_moments_begin jobs still read and write their slots' buffers
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    const int64_t dd = (int64_t)d * d;
    const size_t es = dtype_size(dtype);

    const double* dmu = mu_b; const double* dcov = cov_b; const void* drows = rows; int64_t dld = ld;
    if (!on_device) {
        FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));
        double* s = static_cast<double*>(ws.stage.p);
        FAD_HIP_TRY(hipMemcpyAsync(s, cov_b, dd * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + dd, mu_b, d * sizeof(double), hipMemcpyHostToDevice, st));
        dcov = s; dmu = s + dd;
        const int64_t row_bytes = (int64_t)d * es;
        FAD_TRY(ws.rows.reserve((size_t)(n_rows > 0 ? n_rows : 1) * row_bytes + 16));
        if (n_rows > 0)
            FAD_TRY(host_to_device_2d(ws.rows.p, (size_t)row_bytes, rows, (size_t)(ld * es), (size_t)row_bytes, (size_t)n_rows, device, st));
        drows = ws.rows.p; dld = d;
    }
    FAD_TRY(ws.offs.reserve((size_t)(n_songs + 1) * sizeof(int64_t)));
    FAD_TRY(ws.reserve_song_pin((size_t)(2 * n_songs + 2) * sizeof(double)));        // offsets up | scores down
    memcpy(ws.song_pin, offsets, (size_t)(n_songs + 1) * sizeof(int64_t));
    FAD_HIP_TRY(hipMemcpyAsync(ws.offs.p, ws.song_pin, (size_t)(n_songs + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    const int64_t* d_off = static_cast<const int64_t*>(ws.offs.p);
    const SongKnobs knobs = SongKnobs::from_env();               // every environment switch of the per-song routes, once per call

    switch (dtype) {
        case FAD_F16: return batched_impl<r_f16>(d, dmu, dcov, static_cast<const r_f16*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, knobs, out_scores, out_status);
        case FAD_BF16: return batched_impl<r_bf16>(d, dmu, dcov, static_cast<const r_bf16*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, knobs, out_scores, out_status);
        case FAD_F32: return batched_impl<float>(d, dmu, dcov, static_cast<const float*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, knobs, out_scores, out_status);
        default: return batched_impl<double>(d, dmu, dcov, static_cast<const double*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, knobs, out_scores, out_status);
    }
}

