// Frechet distance between Gaussians on the GPU (gfx950): single pair and batched per-song.
//
// Replaces calc_frechet_distance, fadtk/fad.py:51-120:
//     FAD = ||mu1 - mu2||^2 + tr C1 + tr C2 - 2 tr sqrt(C1 C2)
// and the per-song loop of score_individual, fadtk/fad.py:373-387.
//
// The reference returns tr sqrt through scipy.linalg.eig (sum of sqrt of the eigenvalues of
// C1 C2, fad.py:91-92) and also runs scipy.linalg.sqrtm for a diagnostic (fad.py:88).  Here
// tr sqrt(A), A = C1 C2, comes from the coupled Newton-Schulz iteration
//     Y0 = A / c, Z0 = I;   T = (3 I - Z Y) / 2;   Y <- Y T;   Z <- T Z;     Y -> sqrt(A / c)
// entirely in fp64 on MFMA tiles (gemm_f64.hip).  Scale c = max(tr(A^2)/tr(A), U/2.5) with
// U = min(||A||_F, ||A||_1, ||A||_inf) >= rho(A): every eigenvalue of A/c stays below 3 and the bulk of a
// flat spectrum starts near 1 (ns_prepare).  Iteration 0 needs no T/Z GEMM (Z0 = I, ns_first).  Stopping is
// decided ON DEVICE per problem by a checker workgroup that rides on the update-GEMM launch (ns_check.h), so
// the host enqueues iterations blindly and syncs once per chunk:
//   1  ||I - Z Y||_F <= tol, or the bound 3/4 r^2 + 1/4 r^3 on the NEXT residual is (one more Y update,
//      no further T GEMM)                          (full-rank product)
//   2  trace(Y) AND the residual stand still       (rank-deficient product: null directions never
//                                                   converge but add nothing to the trace; stopping
//                                                   here also keeps Z from blowing up)
//   0  max_iter
// A non-finite residual triggers the reference's eps fallback (fad.py:94-99) once (single-pair API).
//
// Two-frame songs (Whisper) never need a matrix root: with d = x1 - x2,
// Sigma_s = d d^T / 2 is rank one and tr sqrt(Sigma_b Sigma_s) = sqrt(d^T Sigma_b d / 2).
#include "fad_common.h"
#include "ns_check.h"
#include "ns32.h"
#include "ns_mean.h"
#include "ns_fast.h"
#include "ns_fast_big.h"
#include "ns_fast_res.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <type_traits>
#include <vector>

struct fad_moments;
namespace fad {
const double* moments_packed(const fad_moments* h);
int moments_settle(const fad_moments* h, hipStream_t st);      // pending reset -> zeros
int moments_device(const fad_moments* h);
int moments_dim(const fad_moments* h);

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- statistics of A = C1 C2 for the scale of the iteration, by 32 x 32 tile pairs -----------------------------
// Workgroup (bi, bj, problem) loads tile (bi, bj) of A and its mirror (bj, bi) -- both as coalesced 256-byte row
// segments -- and writes: sum |a| of the tile's 32 rows / 32 columns (partial infinity / one norms), and the scalars
// sum a^2 (Frobenius), sum a_ij a_ji (adds up to tr A^2), the tile's share of tr A, tr C1, tr C2.  ns_prepare adds the
// partials in a fixed order (deterministic).  (The first version gave every ROW its own workgroup, which read the
// matching column with a 4 KiB stride: 5.5 us at D = 512 for 2 MB of data.)
constexpr int kStatScal = 8;                         // doubles per tile: sumsq, cross, trA, tr1, tr2, (3 spare)
static int64_t stat_blocks(int d) { return cdiv(d, 32); }
static size_t stat_doubles(int d) { const int64_t nb = stat_blocks(d); return (size_t)(2 * nb * d + kStatScal * nb * nb); }

__global__ __launch_bounds__(256) void ns_tilestats(const double* __restrict__ Aall, int d,
                                                    const double* __restrict__ cov1, int64_t s1,
                                                    const double* __restrict__ cov2, int64_t s2,
                                                    double* __restrict__ stats_all, const NsState* __restrict__ st) {
    __shared__ double P[32][33], Q[32][33];
    __shared__ double red[20];
    const int b = blockIdx.z;
    if (st[b].done) return;
    const int nb = gridDim.x, bi = blockIdx.y, bj = blockIdx.x;
    const double* A = Aall + (int64_t)b * d * d;
    double* stats = stats_all + (int64_t)b * (2 * (int64_t)nb * d + (int64_t)kStatScal * nb * nb);
    double* rowabs = stats;                            // [bj][d]
    double* colabs = stats + (int64_t)nb * d;          // [bi][d]
    double* scal = stats + 2 * (int64_t)nb * d + (int64_t)kStatScal * (bi * nb + bj);
    const int tid = threadIdx.x, r = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gi = bi * 32 + r, gj = bj * 32 + c0 + q;          // element (r, c0+q) of tile (bi, bj)
        P[r][c0 + q] = (gi < d && gj < d) ? A[(int64_t)gi * d + gj] : 0.0;
        const int hi = bj * 32 + r, hj = bi * 32 + c0 + q;          // element (r, c0+q) of tile (bj, bi)
        Q[r][c0 + q] = (hi < d && hj < d) ? A[(int64_t)hi * d + hj] : 0.0;
    }
    __syncthreads();
    if (tid < 32) {
        double t = 0.0;
        for (int c = 0; c < 32; ++c) t += fabs(P[tid][c]);
        if (bi * 32 + tid < d) rowabs[(int64_t)bj * d + bi * 32 + tid] = t;
    } else if (tid < 64) {
        const int c = tid - 32;
        double t = 0.0;
        for (int rr = 0; rr < 32; ++rr) t += fabs(P[rr][c]);
        if (bj * 32 + c < d) colabs[(int64_t)bi * d + bj * 32 + c] = t;
    }
    double sq = 0.0, cr = 0.0, tr = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double v = P[r][c0 + q];
        sq += v * v;
        cr += v * Q[c0 + q][r];
        if (bi == bj && r == c0 + q) {
            tr += v;
            const int64_t i = bi * 32 + r;
            if (i < d) { t1 += cov1[b * s1 + i * d + i]; t2 += cov2[b * s2 + i * d + i]; }
        }
    }
    double v[5] = {sq, cr, tr, t1, t2};
    block_sum_n<5>(v, red);
    if (tid == 0) { scal[0] = v[0]; scal[1] = v[1]; scal[2] = v[2]; scal[3] = v[3]; scal[4] = v[4]; }
}

// rs = sum_k rowabs[k][i], cs = sum_k colabs[k][i] in a fixed order, with the loads of eight partials in flight at once
// (a plain loop issued them one dependent round trip after the other: 17 us for D = 512)
__device__ __forceinline__ void sum_partials(const double* __restrict__ rowabs, const double* __restrict__ colabs, int nb,
                                             int d, int i, double& rs, double& cs) {
    rs = 0.0; cs = 0.0;
    int k = 0;
    for (; k + 8 <= nb; k += 8) {
        double r[8], c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { r[q] = rowabs[(int64_t)(k + q) * d + i]; c[q] = colabs[(int64_t)(k + q) * d + i]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { rs += r[q]; cs += c[q]; }
    }
    for (; k < nb; ++k) { rs += rowabs[(int64_t)k * d + i]; cs += colabs[(int64_t)k * d + i]; }
}

// one block per problem: scale c, traces, mean term; arms the iteration state.
// mean_dtype: FAD_F16 / FAD_BF16 / FAD_F32 = the reference's mean term for embeddings of that dtype, else float64.
__global__ __launch_bounds__(256) void ns_prepare(const double* __restrict__ stats_all, int d, int nb,
                                                  const double* __restrict__ mu1, int64_t m1,
                                                  const double* __restrict__ mu2, int64_t m2, int mean_dtype,
                                                  NsState* __restrict__ st_all, int mean_given = 0,
                                                  Ns32State* __restrict__ s32 = nullptr, int allow_scaled = 1) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    NsState* st = st_all + b;
    if (st->done) return;
    const double* stats = stats_all + (int64_t)b * (2 * (int64_t)nb * d + (int64_t)kStatScal * nb * nb);
    const double* rowabs = stats;
    const double* colabs = stats + (int64_t)nb * d;
    const double* scal = stats + 2 * (int64_t)nb * d;
    mu1 += b * m1; mu2 += b * m2;
    double mr = 0.0, mc = 0.0;
    for (int i = tid; i < d; i += 256) {
        double rs, cs;
        sum_partials(rowabs, colabs, nb, d, i, rs, cs);
        mr = fmax(mr, rs); mc = fmax(mc, cs);
    }
    double sq = 0.0, ta2 = 0.0, ta = 0.0, t1 = 0.0, t2 = 0.0;
    for (int k = tid; k < nb * nb; k += 256) {
        const double* sc = scal + (int64_t)kStatScal * k;
        sq += sc[0]; ta2 += sc[1]; ta += sc[2]; t1 += sc[3]; t2 += sc[4];
    }
    __shared__ double red5[20];
    __shared__ float gaps[1024];
    const double inf_norm = block_max(mr, red);
    const double one_norm = block_max(mc, red);
    double v5[5] = {sq, t1, t2, ta2, ta};              // NaNs/Infs propagate through the sums
    block_sum_n<5>(v5, red5);
    const double fro2 = v5[0], tr1 = v5[1], tr2 = v5[2], trA2 = v5[3], trA = v5[4];
    // mean_given: a spare workgroup of the C1 C2 launch has put the mean term into the state already (gemm_f64.hip)
    double mean_term = mean_given ? st->mean_term : mean_term_block(mu1, mu2, d, mean_dtype, gaps, red);
    if (tid == 0) {
        if (s32) {                                   // the low-precision leg starts from a clean state as well
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->final_iter = -1; s32->failed = 0;
            s32->upd_skip[0] = 0; s32->upd_skip[1] = 0; s32->skip_corr = 1; s32->decided_at = -1; s32->strict = 0;
            s32->res[0] = 1e300;
        }
        // Scale: the iteration needs every eigenvalue of A/c below 3 (above, Y converges to a NEGATIVE root).
        // U = min(||A||_F, ||A||_1, ||A||_inf) >= rho(A) makes c = U/2.5 always safe; the lambda-weighted mean
        // tr(A^2)/tr(A) <= lambda_max is where the bulk of the spectrum sits, and starting the bulk near 1 saves
        // 1-3 iterations when U is loose (flat spectra: U ~ 2.5-3x lambda_max).  c = max of the two.
        double u = sqrt(fro2);
        if (inf_norm < u) u = inf_norm;
        if (one_norm < u) u = one_norm;
        double c = u / 2.5;
        const double wmean = (trA > 0.0) ? trA2 / trA : 0.0;
        if (wmean > c && wmean <= u) c = wmean;
        // SCALED STEPS for decaying spectra (round 4).  The participation ratio (tr A)^2 / tr(A^2) = (sum lambda)^2 / sum lambda^2
        // (exact for a non-normal A as well) says how many eigenvalues matter; below d/4 -- the products the low-precision legs
        // give up on -- the start is c = u (every eigenvalue x^2 of A/c in (0, 1], which the scaled cubic needs) and the lower
        // end l_0 of the spectrum of sqrt(A/c) is ESTIMATED from a power-law model: the exponent p with PR(p) = (sum k^-p)^2 /
        // sum k^-2p (sums by the trapezoid rule), x_min = d^(-p/2), a third of that as l_0.  The schedule stays a valid
        // Newton-Schulz iteration whatever l_0 is: too small only pushes the top of the spectrum further down before it comes back
        // (at worst the optimal rate for that l_0), too large leaves the eigenvalues below it to the plain growth.
        // (not on the symmetric per-song route: it hands a song on by the number of PLAIN iterations it needed -- its proxy for a
        //  spread that sqrt(Sigma_b) at ~1e-10 cannot carry, kSymMaxIter)
        const bool scaled = allow_scaled && (trA > 0.0) && (trA2 > 0.0) && (trA * trA < 0.25 * (double)d * trA2) && (u > 0.0);
        double l = 1.0;
        if (scaled) {
            c = u;
            // (float arithmetic on the hardware's exp2 / log2: the double pow() of a first version made this one thread take 90 us)
            const float pr = (float)(trA * trA / trA2), lg = __log2f((float)d);
            auto S = [&](float p) {                 // sum_{k=1..d} k^-p, trapezoid rule on the integral
                if (fabsf(p - 1.0f) < 1e-4f) return 0.5f * (1.0f + exp2f(-lg)) + lg * 0.69314718f;
                return 0.5f * (1.0f + exp2f(-p * lg)) + (exp2f((1.0f - p) * lg) - 1.0f) / (1.0f - p);
            };
            float lo = 0.0f, hi = 8.0f, pf = 4.0f;
            for (int it = 0; it < 20; ++it) {
                pf = 0.5f * (lo + hi);
                const float s1 = S(pf), val = s1 * s1 / S(2.0f * pf);
                if (val > pr) lo = pf; else hi = pf;
            }
            l = (double)(exp2f(-0.5f * pf * lg) * (1.0f / 3.0f));
            if (l > 0.5) l = 0.5;
            if (l < 1e-5) l = 1e-5;
        }
        {
            int k = 0;
            for (; k < kMaxIter && scaled && l < 0.9; ++k) {          // (a dozen steps at most: 1e-5 -> 0.9)
                const double m = sqrt(3.0 / (1.0 + l + l * l));
                l = m * l * (3.0 - m * m * l * l) / 2.0;
                st->mu[k] = m;
            }
            for (; k < kMaxIter; ++k) st->mu[k] = 1.0;
        }
        const bool bad = !(fro2 == fro2) || isinf(fro2) || !(tr1 == tr1) || !(tr2 == tr2) || isinf(tr1) ||
                         isinf(tr2) || !(mean_term == mean_term) || isinf(mean_term);
        st->c = c; st->tr1 = tr1; st->tr2 = tr2; st->mean_term = mean_term;
        st->res_last = 0.0; st->tr_last = 0.0;
        st->res_min = 1e300; st->tr_safe = 0.0; st->has_safe = 0;
        st->final_iter = -1; st->conv = 0;
        st->nonfinite = bad ? 1 : 0;
        st->done = bad ? 1 : 0;
        st->finished = bad ? 1 : 0;
        if (!bad && !(c > 0.0)) {            // A == 0: its root is 0, nothing to iterate
            st->done = 1; st->finished = 1; st->conv = 1; st->final_iter = 0; st->c = 1.0;
        }
        if (s32) {
            // Is the float32 leg worth starting?  Its result is only accepted while ||Z|| ~ (lambda_min / c)^-1/2 stays below ~20
            // (est is cubic in it), i.e. for spectra that are flat within a factor of a few hundred.  The participation
            // ratio (tr A)^2 / ||A||_F^2 <= rank counts the eigenvalues that matter: d for a flat spectrum, 28 of 512 for
            // covariances decaying like k^-1/2 (already rejected, after 12 iterations), a handful for real embeddings.
            // Below d/4 the leg is switched off here -- every one of its launches skips -- and the host goes straight
            // to the float64 iteration, which reuses this product and this state.  A rule on the inputs alone.
            const bool hopeless = !bad && (c > 0.0) && (trA * trA < 0.25 * (double)d * fro2);
            if (bad || !(c > 0.0) || hopeless) {
                s32->done = 1; s32->finished = 1; s32->failed = 1; s32->upd_skip[0] = 1; s32->upd_skip[1] = 1;
            }
        }
    }
}

// Iteration 0 needs no GEMM for T and Z: with Z0 = I,  T0 = (3I - Y0)/2 and Z1 = T0.  This kernel writes
// Y0 = A/c, T0 (twice: as T and as Z1) and the per-block partial sums of (T0 - I)^2, i.e. the residual of
// iteration 0 in the same form the T GEMM produces it.  grid (ceil(d*d/256), B).
__global__ __launch_bounds__(256) void ns_first(const double* __restrict__ Aall, int d, const NsState* __restrict__ st,
                                                double* __restrict__ Y0, double* __restrict__ T, double* __restrict__ Z1,
                                                int64_t stride, double* __restrict__ partials_all, int nslots) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    if (st[b].done) return;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double e2 = 0.0;
    if (g < (int64_t)d * d) {
        const double inv = 1.0 / st[b].c, m = st[b].mu[0], m3 = m * m * m;       // (a scaled first step: T0 = 1.5 mu I - 0.5 mu^3 Y0)
        const int r = (int)(g / d), c = (int)(g - (int64_t)r * d);
        const double y = Aall[(int64_t)b * d * d + g] * inv;
        const double t = (r == c ? 1.5 * m : 0.0) - 0.5 * m3 * y;
        Y0[b * stride + g] = y;
        T[b * stride + g] = t;
        Z1[b * stride + g] = t;
        const double e = t - (r == c ? 1.5 * m - 0.5 * m3 : 0.0);
        e2 = e * e;
    }
    const double s = block_sum(e2, red);
    if (threadIdx.x == 0) partials_all[(int64_t)b * nslots + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void add_diag(double* __restrict__ M, int d, double eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d) M[(int64_t)i * d + i] += eps;
}

// packed moments -> mu, cov (same formula as moments_finalize_kernel) + the n >= 2 check
__global__ __launch_bounds__(256) void finalize_for_frechet(const double* __restrict__ acc1,
                                                            const double* __restrict__ acc2, int d, int ddof,
                                                            double* __restrict__ mus, double* __restrict__ covs,
                                                            NsState* __restrict__ st) {
    const double* acc = blockIdx.y ? acc2 : acc1;
    double* mu = mus + (int64_t)blockIdx.y * d;
    double* cov = covs + (int64_t)blockIdx.y * d * d;
    const double n = acc[0];
    const double* sum = acc + 1;
    const double* M = acc + 1 + d;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) {
        // also the per-call reset of the iteration state (clear_states), folded in to save a launch: nothing else
        // reads these words before this kernel has finished
        st->too_few[blockIdx.y] = (n < 2.0) ? 1 : 0;
        if (blockIdx.y == 0) {
            st->done = 0; st->finished = 0; st->nonfinite = 0; st->conv = 0; st->final_iter = -1;
            st->upd_skip[0] = 0; st->upd_skip[1] = 0;
        }
    }
    if (g < d) mu[g] = sum[g] / n;
    if (g >= (int64_t)d * d) return;
    const int a = (int)(g / d), b = (int)(g - (int64_t)a * d);
    cov[g] = (M[g] - (sum[a] * sum[b]) / n) / (n - (double)ddof);   // commutative: cov == cov^T bit for bit
}

__global__ void clear_states(NsState* st, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b < B) {
        st[b].too_few[0] = 0; st[b].too_few[1] = 0; st[b].done = 0; st[b].finished = 0; st[b].nonfinite = 0; st[b].conv = 0; st[b].final_iter = -1;
        st[b].upd_skip[0] = 0; st[b].upd_skip[1] = 0;
    }
}

// ------------------------------------------------------------------------------------------
struct Workspace : NsWorkspace {
    DevBuf rows, offs, songbuf, songmat, rows2;     // per-song path
    DevBuf base_root;                               // ... sqrt(Sigma_b) | I | zeros of the symmetric D x D route
    void* song_pin = nullptr; size_t song_pin_cap = 0;      // ... and its pinned staging: offsets going up, scores coming down
    DevBuf mats32;                                  // low-precision leg: Y32[2], Z32[2], T32 (floats)
    DevBuf fast;                                    // the eight-launch chain (ns_fast.h): header, digit planes, split planes
    int fast_gen = 0;                               // per-call token of that chain (MatHdr::flag_gen)
    DevBuf songcov;                                 // ... scratch of the float16 per-song covariances (partial tiles, column sums, shifts)
    DevBuf fast_songs;                              // ... its batched form for songs: baseline digits + one block per song
    void* fast_songs_pin = nullptr; size_t fast_songs_pin_cap = 0;      // ... and what its correction kernel leaves for the host
    DevBuf fast_pairs;                              // ... and for B independent pairs (fad_frechet_from_moments_multi_begin): one block per pair
    void* fast_pairs_pin = nullptr; size_t fast_pairs_pin_cap = 0;
    struct Multi {                                  // an in-flight batch of pairs
        int count = 0, gen = 0;
        bool enqueued = false;                      // the batched chain is on the stream (else: end() scores the pairs one by one)
        const fad_moments_t* h1[8] = {nullptr}; const fad_moments_t* h2[8] = {nullptr};
    } multi;
    // an in-flight score (fad_frechet_from_moments_begin .. fad_frechet_end): everything the collecting side needs
    bool busy = false;
    struct Job {
        int d = 0, device = 0, k = 0, mean_dtype = -1, ddof = 1;
        bool mixed = false;                         // the low-precision chain was enqueued (else: end() runs the synchronous path)
        bool fast = false;                          // ... in its eight-launch form (ns_fast.h); nsf_prepare has staged (mu, Sigma)
        int gen = 0;                                // ... and this is its token
        double eps = 0.0;
        hipStream_t stream = nullptr;
        const double *cov1 = nullptr, *cov2 = nullptr, *mu1 = nullptr, *mu2 = nullptr;
    } job;
    hipEvent_t done_ev = nullptr;
    struct Pool* pool = nullptr;
    void release_all() {
        release(); rows.release(); offs.release(); songbuf.release(); songmat.release(); rows2.release(); mats32.release(); base_root.release(); fast.release();
        fast_songs.release(); songcov.release(); fast_pairs.release();
        if (fast_pairs_pin) { (void)hipHostFree(fast_pairs_pin); fast_pairs_pin = nullptr; fast_pairs_pin_cap = 0; }
        if (fast_songs_pin) { (void)hipHostFree(fast_songs_pin); fast_songs_pin = nullptr; fast_songs_pin_cap = 0; }
        if (done_ev) { (void)hipEventDestroy(done_ev); done_ev = nullptr; }
        if (song_pin) { (void)hipHostFree(song_pin); song_pin = nullptr; song_pin_cap = 0; }
    }
    int reserve_song_pin(size_t bytes) {
        if (song_pin && song_pin_cap >= bytes) return FAD_OK;
        if (song_pin) (void)hipHostFree(song_pin);
        song_pin = nullptr; song_pin_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&song_pin, bytes + bytes / 2 + 4096, hipHostMallocDefault));
        song_pin_cap = bytes + bytes / 2 + 4096;
        return FAD_OK;
    }
};

// One pool of workspaces per (host thread, device), returned to the device when the thread ends (PerThreadDevice):
// calls from a thread pool (fad.py:229, 387 use tmap) never share scratch memory, and one thread can keep up to
// kSlots scores in flight (fad_frechet_from_moments_begin) -- each owns a slot until fad_frechet_end collects it.
struct Pool {
    static constexpr int kSlots = 8;
    Workspace slot[kSlots];
    int lp_iters = 5;                               // iterations the low-precision leg needed last time on this thread
    bool lp_hopeless = false;                       // ... or gave up on at once (a decaying spectrum): the next score enqueues iteration 0
                                                    // and the closing kernel only -- the LAUNCH count follows the history, never the value
    int f64_iters = 0;                              // ... and the float64 iteration (single pair), 0 = not known yet
    int mixed = -1;                                 // FAD_FRECHET_MIXED (read once): 0 = always the fp64 iteration
    int fast = -1;                                  // FAD_FRECHET_FAST (read once): 0 = round 2's twelve-launch float32 chain
    double pred_thr = 0.0;                          // FAD_FRECHET_PRED_THR (read once; -1 = the built-in rule), see pred_threshold
    void release_all() { for (Workspace& w : slot) w.release_all(); }
};
static Pool& thread_pool(int device) {
    static thread_local PerThreadDevice<Pool> set;
    return set.get(device);
}
static Workspace* free_slot(int device) {
    Pool& p = thread_pool(device);
    for (Workspace& w : p.slot)
        if (!w.busy) { w.pool = &p; return &w; }
    return nullptr;
}

struct NsProblem {                  // B problems of dimension d; strides in elements (0 = shared)
    int d; int64_t B;
    const double* cov1; int64_t s_cov1;
    const double* cov2; int64_t s_cov2;
    const double* mu1; int64_t s_mu1;
    const double* mu2; int64_t s_mu2;
    int mean_dtype;                 // ns_prepare: dtype whose rounding the mean term reproduces, or -1 (float64)
    int sym = 0;                    // cov1 cov2 is symmetric (then so is every iterate): the products may skip the mirrored tiles
};

static int ns_pstride(int d) {                       // partial slots per problem: GEMM tiles or ns_first blocks
    const int64_t a = gemm_f64_slots_max(d), b = cdiv((int64_t)d * d, 256);
    return (int)(a > b ? a : b);
}
static size_t ns_small_bytes(int d, int64_t B) {
    return (size_t)B * (sizeof(NsState) + sizeof(Ns32State) + ((size_t)ns_pstride(d) + stat_doubles(d)) * sizeof(double)) + 256;
}

// Enqueue + run the batched iteration.  On return host_states (pinned, B entries) holds the final
// per-problem state; the caller turns it into scores.  States must have been cleared by the caller
// (so that pre-kernels like finalize_for_frechet can raise too_few).
// reuse_prepared: A = C1 C2 (first matrix of ws.mats) and the armed state are those of a float32 attempt on the same problem
// that just gave up (mixed_begin: same buffer, same ns_prepare) -- product, statistics and scale are not formed again.
static int run_ns(const NsProblem& pb, int max_iter, double tol, int device, hipStream_t stream, Workspace& ws,
                  NsState** host_states, bool reuse_prepared = false, double** y_bufs = nullptr) {
    const int d = pb.d;
    const int64_t B = pb.B, dd = (int64_t)d * d;
    if (max_iter <= 0) max_iter = 64;
    if (max_iter > kMaxIter) max_iter = kMaxIter;
    const double tol_res = (tol > 0.0) ? tol : 1e-13 * d;
    const double tol_tr = 1e-13;

    FAD_TRY(ws.mats.reserve((size_t)(6 * dd * B) * sizeof(double)));
    double* A = static_cast<double*>(ws.mats.p);
    double* Y[2] = {A + dd * B, A + 2 * dd * B};
    double* Z[2] = {A + 3 * dd * B, A + 4 * dd * B};
    double* T = A + 5 * dd * B;
    if (y_bufs) { y_bufs[0] = Y[0]; y_bufs[1] = Y[1]; }      // the answer of problem b is sqrt(c) Y[final_iter & 1] (ns_check.h)
    NsState* dstates = static_cast<NsState*>(ws.small.p);
    double* partials = reinterpret_cast<double*>(dstates + B);
    const int pstride = ns_pstride(d);
    double* tilestats = partials + (size_t)B * pstride;
    const int* skip_t = &dstates[0].done;            // T GEMMs stop once convergence is known or predicted

    const size_t hbytes = (size_t)B * sizeof(NsState);
    if (!ws.pinned || ws.pinned_cap < hbytes) {
        if (ws.pinned) (void)hipHostFree(ws.pinned);
        ws.pinned = nullptr; ws.pinned_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.pinned, hbytes + 4096, hipHostMallocDefault));
        ws.pinned_cap = hbytes + 4096;
    }
    NsState* hs = static_cast<NsState*>(ws.pinned);
    *host_states = hs;

    GemmType g[2];
    int rc;
    if (!reuse_prepared) {
        g[0] = {pb.cov1, pb.s_cov1, pb.cov2, pb.s_cov2, A, dd, 1.0, 0.0, 0.0, nullptr, 0, pb.sym};
        rc = gemm_f64_launch(d, g, 1, B, skip_t, kStateInts, stream, device);
        if (rc < 0) return rc;
        const unsigned nb = (unsigned)stat_blocks(d);
        hipLaunchKernelGGL(ns_tilestats, dim3(nb, nb, (unsigned)B), dim3(256), 0, stream, A, d, pb.cov1, pb.s_cov1, pb.cov2,
                           pb.s_cov2, tilestats, dstates);
        hipLaunchKernelGGL(ns_prepare, dim3((unsigned)B), dim3(256), 0, stream, tilestats, d, (int)nb, pb.mu1, pb.s_mu1, pb.mu2,
                           pb.s_mu2, pb.mean_dtype, dstates, 0, (Ns32State*)nullptr, pb.sym ? 0 : 1);
    }
    // iteration 0 without GEMMs for T and Z (Z0 = I): Y0, T0, Z1 = T0, residual partials
    const int nslots0 = (int)cdiv(dd, 256);
    hipLaunchKernelGGL(ns_first, dim3((unsigned)nslots0, (unsigned)B), dim3(256), 0, stream, A, d, dstates, Y[0], T, Z[1],
                       dd, partials, pstride);

    // launches are enqueued blind, `chunk` iterations at a time: first what the previous single-pair call on this thread needed
    // (+1 for the check that closes a predicted finish; 6 = what well-conditioned D=512 products take), then four at a time.
    // Every host round trip in between costs the chain ~20-30 us; the decisions are the device's, so the count only sets how
    // many launches end up skipped.
    int cur = 0, k = 0, chunk = 6;
    if (B == 1 && ws.pool && ws.pool->f64_iters > 0) chunk = ws.pool->f64_iters + 1;
    bool all_done = false;
    NsCheckArgs chk;
    chk.max_iter = max_iter; chk.st_all = dstates; chk.partials_all = partials; chk.pstride = pstride; chk.stride = dd;
    chk.d = d; chk.tol_res = tol_res; chk.tol_tr = tol_tr;
    while (!all_done && k < max_iter) {
        const int stop = (k + chunk < max_iter) ? k + chunk : max_iter;
        for (; k < stop; ++k) {
            int nslots = nslots0;
            if (k > 0) {
                g[0] = {Z[cur], dd, Y[cur], dd, T, dd, -0.5, 1.5, 1.0, partials, 0, pb.sym};
                g[0].mu = &dstates[0].mu[k]; g[0].mu_stride = (int64_t)(sizeof(NsState) / sizeof(double));      // the step's scale: on the device
                nslots = gemm_f64_launch(d, g, 1, B, skip_t, kStateInts, stream, device, pstride);
                if (nslots < 0) return nslots;
            }
            // update GEMMs of iteration k + its convergence check as one extra workgroup per problem
            chk.k = k; chk.nslots = nslots; chk.Yall = Y[cur];
            g[0] = {Y[cur], dd, T, dd, Y[cur ^ 1], dd, 1.0, 0.0, 0.0, nullptr, 0, pb.sym};
            g[1] = {T, dd, Z[cur], dd, Z[cur ^ 1], dd, 1.0, 0.0, 0.0, nullptr, 0, pb.sym};
            rc = gemm_f64_launch(d, g, k == 0 ? 1 : 2, B, &dstates[0].upd_skip[k & 1], kStateInts, stream, device, 0,
                                 &chk);                                                               // Z1 = T0 is in place
            if (rc < 0) return rc;
            cur ^= 1;
        }
        // (a problem whose convergence was PREDICTED by the last check of this chunk is closed by the first check
        // of the next chunk -- its GEMMs are already switched off -- rather than by a launch of its own)
        FAD_HIP_TRY(hipMemcpyAsync(hs, dstates, hbytes, hipMemcpyDeviceToHost, stream));
        FAD_HIP_TRY(hipStreamSynchronize(stream));
        all_done = true;
        for (int64_t b = 0; b < B; ++b) if (!hs[b].finished) { all_done = false; break; }
        chunk = 4;
    }
    FAD_HIP_TRY(hipGetLastError());
    if (B == 1 && ws.pool && hs[0].finished && hs[0].final_iter >= 0) ws.pool->f64_iters = hs[0].final_iter + 1;
    return FAD_OK;
}

// ==========================================================================================
// Mixed-precision leg (single pair, d % 64 == 0): Newton-Schulz in fp32 on the f32-input MFMA down to the fp32
// floor, then ONE fp64 correction
//     tr sqrt(A) = tr Y + 1/2 tr(Z (A - Y Y)) + O(err^2),      A, Y Y and the traces in fp64,
// (first-order Newton step of X -> X^2 = A around Y with Z ~ Y^-1; SURVEY.md section 7 H1 measured 2e-9).  With
// S = sqrt(A), D = S - Y and Z = S^-1 + G the neglected terms are 1/2 tr(S^-1 D^2) and 1/2 tr(G R), bounded by
//     est = ||Z||^3 ||R||_F^2 / 8 + ||Z|| r ||R||_F / 2,    ||Z|| <= sqrt(||Z||_1 ||Z||_inf),  r = last residual;
// the result is accepted when est <= 1e-9 |tr| (measured: est overestimates the true error 10-1000x; config 3:
// est 2e-12, error 2e-13), otherwise -- ill-conditioned or rank-deficient products, fp32 not converging -- the
// all-fp64 iteration above runs from scratch.  Per call: A (fp64 GEMM), statistics, 4-5 fp32 iterations at
// ~11 us instead of ~25, Y Y (fp64 GEMM on fp32 operands), two small reduction kernels, ONE host sync; the result
// is written straight into pinned host memory.  The number of blind iterations is the count the previous call on
// this thread needed (scores of one run need the same count; a short batch is topped up two at a time).
// ==========================================================================================
struct MixedResult {       // status / pieces of the result, in pinned host memory (written by ns32_finish, or by fast_decide on the host)
    int status;            // 0: low-precision iteration not finished yet, 1: accepted, 2: rejected -> fp64 iteration,
                           // 4: a PREDICTED final iterate was rejected -> iterate on from `iters` with the strict threshold
    int iters, decided_at, nonfinite, too_few0, too_few1;
    double tr_scaled, c, tr1, tr2, mean_term, res, est;
    int prepared;          // A = C1 C2 (float64) and the armed state are valid: the float64 route may start from them
    int pad;
};

// One block: reduce the partials, decide, write the result where the host reads it (pinned host memory).
__global__ __launch_bounds__(256) void ns32_finish(const double* __restrict__ stats, int d, int nb,
                                                   const NsState* __restrict__ st, Ns32State* __restrict__ s32,
                                                   MixedResult* __restrict__ out, int max_low) {
    __shared__ double red[4];
    __shared__ double red3[12];
    const int tid = threadIdx.x;
    const bool live = s32->ok != 0;
    double mr = 0.0, mc = 0.0, corr = 0.0, r2 = 0.0, tr = 0.0;
    if (live) {
        const double* rowabs = stats;
        const double* colabs = stats + (int64_t)nb * d;
        const double* scal = stats + 2 * (int64_t)nb * d;
        for (int i = tid; i < d; i += 256) {
            double rs, cs;
            sum_partials(rowabs, colabs, nb, d, i, rs, cs);
            mr = fmax(mr, rs); mc = fmax(mc, cs);
        }
        for (int k = tid; k < nb * nb; k += 256) {
            const double* sc = scal + (int64_t)kStatScal * k;
            corr += sc[0]; r2 += sc[1]; tr += sc[2];
        }
    }
    const double zinf = block_max(mr, red), zone = block_max(mc, red);
    double v3[3] = {corr, r2, tr};
    block_sum_n<3>(v3, red3);
    corr = v3[0]; r2 = v3[1]; tr = v3[2];
    if (tid != 0) return;
    MixedResult o;
    o.status = 0; o.iters = s32->final_iter; o.decided_at = s32->decided_at; o.nonfinite = st->nonfinite;
    o.too_few0 = st->too_few[0]; o.too_few1 = st->too_few[1];
    o.c = st->c; o.tr1 = st->tr1; o.tr2 = st->tr2; o.mean_term = st->mean_term;
    o.tr_scaled = 0.0; o.res = 0.0; o.est = 0.0; o.prepared = 1; o.pad = 0;
    if (st->done || s32->failed) {
        o.status = 2;                                 // bad / zero product or fp32 gave up: the fp64 path decides
    } else if (live) {
        const int f = s32->final_iter;
        const int fm = f < 16 ? f : 15;
        // residual of the final iterate: measured when the check stopped AT it, else the bound from the one before
        double res = s32->res[fm];
        if (s32->decided_at == f - 1) { const double rp = s32->res[f - 1 < 16 ? f - 1 : 15]; res = 0.75 * rp * rp + 0.25 * rp * rp * rp; if (res < 2e-6) res = 2e-6; }
        const double zn = sqrt(zinf * zone), rn = sqrt(r2);
        const double trs = tr + 0.5 * corr;
        const double est = zn * zn * zn * rn * rn / 8.0 + zn * res * rn / 2.0;
        const bool finite = (trs == trs) && !isinf(trs) && (est == est) && !isinf(est);
        o.tr_scaled = trs; o.res = res; o.est = est;
        o.status = (finite && est <= 1e-9 * fabs(trs)) ? 1 : 2;
        if (o.status == 2 && finite && !s32->strict && s32->decided_at == f - 1 && f + 1 < max_low) {
            // the iterate was taken as final on a PREDICTED residual and the correction cannot absorb it: nothing is
            // lost -- (Y_f, Z_f) are intact, the iteration goes on from there and only the fp32 floor ends it now
            o.status = 4;
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->skip_corr = 1; s32->upd_skip[0] = 0; s32->upd_skip[1] = 0;
            s32->strict = 1;
        }
    }
    *out = o;            // pinned host memory: visible to the host once the stream has been synchronised
}

constexpr int kMaxLow = 14;
constexpr int kSymMaxIter = 16;   // symmetric per-song route: iterates beyond this mean a spread the route's sqrt(Sigma_b) cannot carry

// When may the check of iteration k declare Y_{k+1} final from the bound b = 3/4 r_k^2 + 1/4 r_k^3 on its residual?
// The fp64 correction leaves an error of about (||Z||^3/8 + ||Z||/2) b^2 (ns32_finish: est, with ||R|| <~ b), which has
// to stay below 1e-9 |tr sqrt| ~ 1e-9 d for a flat spectrum: b <~ 2.5e-3 sqrt-ish of d/512 for ||Z|| ~ 2-3.  The
// Frobenius bound b itself overestimates the residual it predicts ~10x (measured, config 3: b = 1.4e-3, next residual
// 1.4e-4, est 3e-10 |tr|), so 2.5e-3 d/512 is taken as is.  Waiting for the fp32 floor instead (b <= 2e-6, round 1)
// costs one more iteration -- two launches of ~10 us -- on every well-conditioned score.  A rejected prediction costs
// one correction and one more trip to the host; the rest of THAT score then runs strict (Ns32State::strict), so the
// result is a function of the inputs alone, never of what the thread scored before.  FAD_FRECHET_PRED_THR (read once
// per thread) overrides the rule (tests use it to force a rejection).
static double pred_threshold(Pool* p, int d) {
    if (p && p->pred_thr == 0.0) { const char* e = getenv("FAD_FRECHET_PRED_THR"); p->pred_thr = (e && atof(e) > 0.0) ? atof(e) : -1.0; }
    if (p && p->pred_thr > 0.0) return p->pred_thr;
    return 2.5e-3 * (double)d / 512.0;
}

struct MixedBufs {
    double* A; float *Y[2], *Z[2], *T;
    NsState* dstate; double* partials; double* tilestats; Ns32State* s32; MixedResult* hres;
    unsigned nb;
};
static MixedBufs mixed_bufs(Workspace& ws, int d) {
    const int64_t dd = (int64_t)d * d;
    MixedBufs m;
    m.A = static_cast<double*>(ws.mats.p);
    m.Y[0] = static_cast<float*>(ws.mats32.p); m.Y[1] = m.Y[0] + dd;
    m.Z[0] = m.Y[1] + dd; m.Z[1] = m.Y[1] + 2 * dd;
    m.T = m.Y[1] + 3 * dd;
    m.dstate = static_cast<NsState*>(ws.small.p);
    m.partials = reinterpret_cast<double*>(m.dstate + 1);
    m.tilestats = m.partials + ns_pstride(d);
    m.s32 = reinterpret_cast<Ns32State*>(m.tilestats + stat_doubles(d));
    m.hres = reinterpret_cast<MixedResult*>(static_cast<char*>(ws.pinned) + sizeof(NsState));
    m.nb = (unsigned)stat_blocks(d);
    return m;
}

// iterations [ws.job.k, upto) of the low-precision leg, then the closing kernels (fp64 correction, result -> pinned host)
static int mixed_enqueue(Workspace& ws, int upto) {
    const int d = ws.job.d;
    hipStream_t stream = ws.job.stream;
    MixedBufs m = mixed_bufs(ws, d);
    int rc;
    for (int& k = ws.job.k; k < upto; ++k) {
        const int cur = k & 1;
        Gemm32Args g;
        memset(&g, 0, sizeof(g));
        if (k == 0) {
            // iteration 0 in one launch: Y0 = A/c and T0 = (3I - Y0)/2 are formed while A is staged, Y1 = Y0 T0, Z1 = T0
            // (Z0 = I needs no product).  No check rides on it -- its residual ||I - Y0|| decides nothing a well-posed
            // problem cares about (a non-finite product was caught by ns_prepare); the first check is iteration 1's.
            g.C[0] = m.Y[1]; g.C[1] = m.Z[1]; g.alpha[0] = 1.0f; g.ntypes = 1; g.A64 = m.A; g.st64 = m.dstate;
            g.skip = &m.s32->done;               // (set by ns_prepare: bad / zero product, or a spectrum the float32 leg cannot serve)
            FAD_TRY(gemm_f32_first_launch(d, g, stream));
            continue;
        }
        // T = (3I - Z Y)/2 and the residual partials of iteration k
        g.A[0] = m.Z[cur]; g.B[0] = m.Y[cur]; g.C[0] = m.T; g.alpha[0] = -0.5f; g.beta_eye[0] = 1.5f; g.gamma[0] = 1.0f;
        g.partials[0] = m.partials; g.skip = &m.s32->done; g.ntypes = 1;
        const int nslots = gemm_f32_launch(d, g, stream);
        if (nslots < 0) return nslots;
        memset(&g, 0, sizeof(g));
        // Y <- Y T, Z <- T Z + the check of iteration k as an extra workgroup
        g.A[0] = m.Y[cur]; g.B[0] = m.T; g.C[0] = m.Y[cur ^ 1]; g.alpha[0] = 1.0f;
        g.A[1] = m.T; g.B[1] = m.Z[cur]; g.C[1] = m.Z[cur ^ 1]; g.alpha[1] = 1.0f;
        g.ntypes = 2;
        g.skip = &m.s32->upd_skip[k & 1];
        g.check = 1; g.k = k; g.max_low = kMaxLow; g.nslots = nslots; g.chk_partials = m.partials; g.st = m.s32; g.st64 = m.dstate;
        g.thr_pred = pred_threshold(ws.pool, d);
        rc = gemm_f32_launch(d, g, stream);
        if (rc < 0) return rc;
    }
    // fp64 correction on the final iterate (which of the ping-pong buffers: known on the device only): Y Y in fp64 with the
    // statistics of R = A/c - Y Y and Z formed in the epilogue (one launch instead of product + ns32_corr_partials)
    NsProductExt ext;
    memset(&ext, 0, sizeof(ext));
    ext.stats = m.tilestats; ext.st = m.dstate; ext.A64 = m.A; ext.Z32 = m.Z[0]; ext.Z32_alt = m.Z[1];
    FAD_TRY(gemm_f64_correction_launch(d, m.Y[0], m.Y[1], &m.s32->final_iter, &m.s32->skip_corr, ext, stream));
    hipLaunchKernelGGL(ns32_finish, dim3(1), dim3(256), 0, stream, m.tilestats, d, (int)m.nb, m.dstate, m.s32, m.hres, kMaxLow);
    FAD_HIP_TRY(hipGetLastError());
    if (!ws.done_ev) FAD_HIP_TRY(hipEventCreateWithFlags(&ws.done_ev, hipEventDisableTiming));
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, stream));
    return FAD_OK;
}

// ==========================================================================================
// The eight-launch form of the chain (ns_fast.h): exact products on the int8 MFMA, iteration on split-float16 operands.
// D in {256, 512, 768, 1024}; FAD_FRECHET_FAST=0 keeps round 2's float32 chain (the two are compared in the tests).
// ==========================================================================================
static bool mixed_eligible(Workspace& ws, int d, int max_iter, double tol);
static bool fast_dim(int d) { return d == 256 || d == 384 || d == 512 || d == 768 || d == 1024; }
static bool fast_eligible(Workspace& ws, int d, int max_iter, double tol) {
    Pool* p = ws.pool;
    if (p && p->fast < 0) { const char* e = getenv("FAD_FRECHET_FAST"); p->fast = (e && e[0] == '0') ? 0 : 1; }
    return (!p || p->fast) && fast_dim(d) && mixed_eligible(ws, d, max_iter, tol);
}

struct FastBufs {
    nsf::MatHdr* hdr;                               // [2]
    uint4* digC[2];
    nsf::SplitMat P, Y[2], Z[2], T;
    uint4 *digY[2], *digYt[2];
    // pinned host memory behind NsState + MixedResult: what the correction kernel leaves for fast_decide
    int* host_words; double* host_vals; double* host_stats;
};
static size_t fast_bytes(int d) {
    const size_t dd = (size_t)d * d;
    return 256 + 2 * 6 * dd + 6 * 8 * dd + 4 * 6 * dd + 256;
}
static size_t fast_pinned_bytes(int d) {
    const size_t nb = (size_t)d / 32;
    return sizeof(NsState) + sizeof(MixedResult) + 64 + nsf::kHostWords * sizeof(int) + nsf::kHostVals * sizeof(double) +
           (nsf::kTileStats + 2) * nb * nb * sizeof(double) + 64;
}
static FastBufs fast_bufs(Workspace& ws, int d) {
    const size_t dd = (size_t)d * d;
    char* p = static_cast<char*>(ws.fast.p);
    FastBufs f;
    f.hdr = reinterpret_cast<nsf::MatHdr*>(p); p += 256;
    for (int i = 0; i < 2; ++i) { f.digC[i] = reinterpret_cast<uint4*>(p); p += 6 * dd; }
    nsf::SplitMat* mats[6] = {&f.P, &f.Y[0], &f.Y[1], &f.Z[0], &f.Z[1], &f.T};
    for (nsf::SplitMat* m : mats) {
        m->a = reinterpret_cast<uint4*>(p); p += 4 * dd;
        m->at = reinterpret_cast<uint4*>(p); p += 4 * dd;
    }
    for (int i = 0; i < 2; ++i) { f.digY[i] = reinterpret_cast<uint4*>(p); p += 6 * dd; f.digYt[i] = reinterpret_cast<uint4*>(p); p += 6 * dd; }
    char* h = static_cast<char*>(ws.pinned);
    f.host_words = nullptr; f.host_vals = nullptr; f.host_stats = nullptr;
    if (h && ws.pinned_cap >= fast_pinned_bytes(d)) {
        h += ((sizeof(NsState) + sizeof(MixedResult) + 63) / 64) * 64;
        f.host_vals = reinterpret_cast<double*>(h); h += nsf::kHostVals * sizeof(double);
        f.host_stats = reinterpret_cast<double*>(h); h += (nsf::kTileStats + 2) * ((size_t)d / 32) * ((size_t)d / 32) * sizeof(double);
        f.host_words = reinterpret_cast<int*>(h);
    }
    return f;
}

// K1 on `stream`: mu of both sets and (from packed moments) their covariances into the slot's staging area, state reset, scales,
// digit planes.  acc1 == nullptr: the caller's device matrices cov1 / cov2 are used as they are.
static int fast_prepare(Workspace& ws, int d, int ddof, const double* acc1, const double* acc2, const double* cov1, const double* cov2,
                        const double* mu1, const double* mu2, int mean_dtype, double* mus, double* covs, hipStream_t st) {
    void* const before = ws.fast.p;
    FAD_TRY(ws.fast.reserve(fast_bytes(d)));
    FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
    FastBufs f = fast_bufs(ws, d);
    if (ws.fast.p != before) FAD_HIP_TRY(hipMemsetAsync(f.hdr, 0, 256, st));       // a fresh header: no stale token in its flag words
    ws.job.gen = ++ws.fast_gen;
    if (ws.fast_gen > (1 << 30)) ws.fast_gen = 1;
    nsf::PrepArgs a;
    memset(&a, 0, sizeof(a));
    a.acc[0] = acc1; a.acc[1] = acc2; a.cov_in[0] = cov1; a.cov_in[1] = cov2; a.mu_in[0] = mu1; a.mu_in[1] = mu2;
    a.d = d; a.ddof = ddof; a.gen = ws.job.gen; a.mean_dtype = mean_dtype;
    a.mus = mus; a.covs = covs;
    a.dig[0] = f.digC[0]; a.dig[1] = f.digC[1];
    a.st = static_cast<NsState*>(ws.small.p);
    a.hdr[0] = f.hdr; a.hdr[1] = f.hdr + 1;
    hipLaunchKernelGGL(nsf::nsf_prepare, dim3((unsigned)((int64_t)d * d / 2048 + 1), 2), dim3(512), 0, st, a);
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

template <int NS> static void fast_launch_split(int mode, unsigned t, unsigned B, const nsf::SplitArgs& g, hipStream_t st) {
    if (mode == nsf::SP_FIRST) hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_FIRST>), dim3(t, t, B), dim3(512), 0, st, g);
    else if (mode == nsf::SP_T) hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_T>), dim3(t, t, B), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_U>), dim3(t, t, 3 * B), dim3(512), 0, st, g);
}
static void fast_split(int d, int mode, const nsf::SplitArgs& g, hipStream_t st, unsigned B = 1) {
    const unsigned t = (unsigned)(d / 32);
    switch (d) {
        case 128: fast_launch_split<1>(mode, t, B, g, st); break;
        case 256: fast_launch_split<2>(mode, t, B, g, st); break;
        case 384: fast_launch_split<3>(mode, t, B, g, st); break;
        case 512: fast_launch_split<4>(mode, t, B, g, st); break;
        case 768: fast_launch_split<6>(mode, t, B, g, st); break;
        default: fast_launch_split<8>(mode, t, B, g, st); break;
    }
}
// the batched form of SP_T / SP_U on 128 x 128 tiles (ns_fast_big.h); 64 KiB + of dynamic LDS: the attribute is set once per device
static int fast_split_big(int d, int mode, nsf::SplitArgs g, hipStream_t st, unsigned B, int device) {
    static std::atomic<unsigned> ready{0};
    if (device >= 0 && device < 32 && !(ready.load(std::memory_order_acquire) & (1u << device))) {
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_big<nsf::SP_T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kBigLds));
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_big<nsf::SP_U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kBigLds));
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_big<nsf::SP_FIRST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kBigLds));
        ready.fetch_or(1u << device, std::memory_order_release);
    }
    const unsigned t = (unsigned)(d / 128), Bp = (B + 7u) & ~7u;
    g.nprob = (int)B; g.nprob_pad = (int)Bp;
    if (mode == nsf::SP_FIRST) hipLaunchKernelGGL((nsf::nsf_big<nsf::SP_FIRST>), dim3(t * t * Bp), dim3(256), nsf::kBigLds, st, g);
    else if (mode == nsf::SP_T) hipLaunchKernelGGL((nsf::nsf_big<nsf::SP_T>), dim3(t * t * Bp), dim3(256), nsf::kBigLds, st, g);
    else hipLaunchKernelGGL((nsf::nsf_big<nsf::SP_U>), dim3(2 * t * t * Bp + B), dim3(256), nsf::kBigLds, st, g);
    return FAD_OK;
}
static int fast_i8_big(int d, int mode, const nsf::I8Args& g, hipStream_t st, unsigned B, int device) {
    static std::atomic<unsigned> ready{0};
    if (device >= 0 && device < 32 && !(ready.load(std::memory_order_acquire) & (1u << device))) {
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_i8_big<nsf::I8_A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kI8BigLds));
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_i8_big<nsf::I8_G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kI8BigLds));
        ready.fetch_or(1u << device, std::memory_order_release);
    }
    const unsigned tiles = (unsigned)((d / 128) * (d / 64)), Bp = (B + 7u) & ~7u;
    if (mode == nsf::I8_A) hipLaunchKernelGGL((nsf::nsf_i8_big<nsf::I8_A>), dim3(tiles * Bp), dim3(512), nsf::kI8BigLds, st, g, (int)B, (int)Bp);
    else hipLaunchKernelGGL((nsf::nsf_i8_big<nsf::I8_G>), dim3(tiles * Bp), dim3(512), nsf::kI8BigLds, st, g, (int)B, (int)Bp);
    return FAD_OK;
}
template <int NS8> static void fast_launch_i8(int mode, unsigned t, unsigned B, const nsf::I8Args& g, hipStream_t st) {
    if (mode == nsf::I8_A) hipLaunchKernelGGL((nsf::nsf_i8<NS8, nsf::I8_A>), dim3(t, t, B), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((nsf::nsf_i8<NS8, nsf::I8_G>), dim3(t, t, B), dim3(512), 0, st, g);
}
static void fast_i8(int d, int mode, const nsf::I8Args& g, hipStream_t st, unsigned B = 1) {
    const unsigned t = (unsigned)(d / 32);
    switch (d) {
        case 128: case 256: fast_launch_i8<1>(mode, t, B, g, st); break;       // (d = 128: four k-steps, half the waves idle)
        case 384: case 512: fast_launch_i8<2>(mode, t, B, g, st); break;                  // (d = 384: twelve k-steps, waves 6, 7 idle)
        case 768: fast_launch_i8<3>(mode, t, B, g, st); break;
        default: fast_launch_i8<4>(mode, t, B, g, st); break;
    }
}

// iterations [ws.job.k, upto) of the chain, then the exact correction, whose partials land in pinned host memory
static int fast_enqueue(Workspace& ws, int upto) {
    const int d = ws.job.d;
    hipStream_t stream = ws.job.stream;
    MixedBufs m = mixed_bufs(ws, d);
    FastBufs f = fast_bufs(ws, d);
    if (!f.host_words) return set_error(FAD_ERR_ALLOC, "pinned result area of the fast Frechet chain is missing");
    const int nslots = (d / 32) * (d / 32);
    for (int& k = ws.job.k; k < upto; ++k) {
        nsf::SplitArgs g;
        memset(&g, 0, sizeof(g));
        g.d = d; g.gen = ws.job.gen; g.hA = f.hdr; g.hB = f.hdr + 1; g.st = m.dstate; g.s32 = m.s32;
        if (k == 0) {
            // A = C1 C2 (exact) + its statistics + the mean term, then iteration 0: Y1 = Y0 T0, Z1 = T0
            nsf::I8Args a;
            memset(&a, 0, sizeof(a));
            a.Adig = f.digC[0]; a.Bdig = f.digC[1]; a.d = d; a.gen = ws.job.gen; a.hA = f.hdr; a.hB = f.hdr + 1; a.stats = m.tilestats;
            a.A64 = m.A; a.P = f.P; a.st = m.dstate;
            fast_i8(d, nsf::I8_A, a, stream);
            g.A[0] = f.P; g.B[0] = f.P; g.C[0] = f.Y[1]; g.C[1] = f.Z[1]; g.Cdig[0] = f.digY[1]; g.Cdig_t[0] = f.digYt[1];
            g.A64 = m.A; g.statsA = m.tilestats;
            fast_split(d, nsf::SP_FIRST, g, stream);
            continue;
        }
        const int cur = k & 1;
        // T = (3I - Z Y)/2 and the residual partials of iteration k
        g.A[0] = f.Z[cur]; g.B[0] = f.Y[cur]; g.C[0] = f.T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f;
        g.partials = m.partials; g.skip = &m.s32->done;
        fast_split(d, nsf::SP_T, g, stream);
        // Y <- Y T, Z <- T Z + the check of iteration k as an extra workgroup
        memset(&g, 0, sizeof(g));
        g.d = d; g.gen = ws.job.gen; g.hA = f.hdr; g.hB = f.hdr + 1; g.st = m.dstate; g.s32 = m.s32;
        g.A[0] = f.Y[cur]; g.B[0] = f.T; g.C[0] = f.Y[cur ^ 1];
        g.A[1] = f.T; g.B[1] = f.Z[cur]; g.C[1] = f.Z[cur ^ 1];
        g.Cdig[0] = f.digY[cur ^ 1]; g.Cdig_t[0] = f.digYt[cur ^ 1];
        g.skip = &m.s32->upd_skip[k & 1];
        g.k = k; g.max_low = kMaxLow; g.nslots = nslots; g.chk_partials = m.partials;
        g.thr_pred = pred_threshold(ws.pool, d);
        fast_split(d, nsf::SP_U, g, stream);
    }
    // exact correction on the final iterate (which of the ping-pong buffers: known on the device only)
    nsf::I8Args a;
    memset(&a, 0, sizeof(a));
    a.Adig = f.digY[0]; a.Bdig = f.digYt[0]; a.Adig_alt = f.digY[1]; a.Bdig_alt = f.digYt[1]; a.sel = &m.s32->final_iter;
    a.d = d; a.gen = ws.job.gen; a.hA = f.hdr; a.hB = f.hdr + 1; a.skip = &m.s32->skip_corr; a.stats = f.host_stats; a.st = m.dstate; a.A64in = m.A;
    a.Y[0] = f.Y[0]; a.Y[1] = f.Y[1]; a.Z[0] = f.Z[0]; a.Z[1] = f.Z[1];
    a.s32 = m.s32; a.host_words = f.host_words; a.host_vals = f.host_vals;
    f.host_words[12] = 0;                          // (the kernel stamps the snapshot with this score's token)
    fast_i8(d, nsf::I8_G, a, stream);
    FAD_HIP_TRY(hipGetLastError());
    if (!ws.done_ev) FAD_HIP_TRY(hipEventCreateWithFlags(&ws.done_ev, hipEventDisableTiming));
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, stream));
    return FAD_OK;
}

// The closing decision, on the host (the enqueued chain has been waited for): reduce the correction's per-tile partials, bound the
// neglected terms, accept / reject -- what ns32_finish does on the device for the float32 chain, minus a launch.
// hw / hv / hsx: what nsf_i8<G> left for ONE problem.  -> status 1 accepted, 2 rejected, 4 a predicted final iterate was rejected
// (the iteration may go on from it), 0 not finished yet.
static void fast_decide_one(const int* hw, const double* hv, const double* hsx, int nb, MixedResult* out) {
    MixedResult o;
    memset(&o, 0, sizeof(o));
    const bool bad = hw[0] != 0;
    o.status = 0; o.iters = hw[7]; o.decided_at = hw[8]; o.nonfinite = hw[2]; o.too_few0 = hw[3]; o.too_few1 = hw[4];
    o.c = hv[0]; o.tr1 = hv[1]; o.tr2 = hv[2]; o.mean_term = hv[3];
    // (the float64 route never starts from THIS chain's product: A was formed from covariances on a fixed-point grid of 2^-41,
    //  fine for the flat spectra the chain accepts, but the small eigenvalues of the spectra it gives up on move with a perturbation
    //  divided by their own square root -- measured 4e-3 of a FAD on a k^-3 spectrum)
    o.prepared = 0;
    const bool ok = hw[5] != 0, failed = hw[6] != 0, strict = hw[9] != 0;
    if (bad || hw[1] || failed) {
        o.status = 2;                                // bad / zero product or the low-precision leg gave up: the float64 route decides
    } else if (ok && !hw[11]) {
        double corr = 0.0, r2 = 0.0, tr = 0.0, zinf = 0.0, zone = 0.0;
        for (int t = 0; t < nb * nb; ++t) { corr += hsx[nsf::kTileStats * t]; r2 += hsx[nsf::kTileStats * t + 1]; tr += hsx[nsf::kTileStats * t + 2]; }
        const double* zmax = hsx + (size_t)nsf::kTileStats * nb * nb;     // [ty * nb + tx]: (row part, column part) of |Z|, rows block tx, columns block ty
        for (int x = 0; x < nb; ++x) {
            double rs = 0.0, cs = 0.0;
            for (int y = 0; y < nb; ++y) { rs += zmax[2 * (y * nb + x)]; cs += zmax[2 * (x * nb + y) + 1]; }
            if (rs > zinf) zinf = rs;                // >= the largest row sum of |Z| over row block x
            if (cs > zone) zone = cs;                // >= the largest column sum over column block x
        }
        const int fi = hw[7], fm = fi < 16 ? (fi < 0 ? 0 : fi) : 15;
        double res = hv[4 + fm];
        if (hw[8] == fi - 1) { const double rp = hv[4 + (fi - 1 < 16 ? (fi - 1 < 0 ? 0 : fi - 1) : 15)]; res = 0.75 * rp * rp + 0.25 * rp * rp * rp; if (res < 2e-6) res = 2e-6; }
        const double zn = std::sqrt(zinf * zone), rn = std::sqrt(r2);
        const double trs = tr + 0.5 * corr;
        const double est = zn * zn * zn * rn * rn / 8.0 + zn * res * rn / 2.0;
        const bool finite = std::isfinite(trs) && std::isfinite(est);
        o.tr_scaled = trs; o.res = res; o.est = est;
        // accepted when the bound on the neglected terms is below 1e-9 of the trace, or moves the DISTANCE by less than 1e-5 of
        // itself (10x inside the 1e-4 bar AS A BOUND: it overestimates the true error 10..10^5 times, most for spread spectra
        // where the norm bound of Z is 3-4x its 2-norm and enters cubed -- a song of 2 D frames, condition 400: bound 2e-8 of the
        // trace, true error 2e-13, scripts/ns_emulate_split.py)
        const double fad = o.mean_term + o.tr1 + o.tr2 - 2.0 * std::sqrt(o.c) * trs;
        const bool accept = finite && (est <= 1e-9 * std::fabs(trs) || 2.0 * std::sqrt(o.c) * est <= 1e-5 * std::fabs(fad));
        o.status = accept ? 1 : 2;
        if (!accept && finite && !strict && hw[8] == fi - 1 && fi + 1 < kMaxLow) o.status = 4;
    }
    *out = o;
}

static int fast_decide(Workspace& ws) {
    const int d = ws.job.d, nb = d / 32;
    MixedBufs m = mixed_bufs(ws, d);
    FastBufs f = fast_bufs(ws, d);
    if (f.host_words[12] != ws.job.gen) return set_error(FAD_ERR_HIP, "the correction kernel of the fast Frechet chain left no result");
    fast_decide_one(f.host_words, f.host_vals, f.host_stats, nb, m.hres);
    if (m.hres->status == 4) {
        // the iterate was taken as final on a PREDICTED residual and the correction cannot absorb it: nothing is lost --
        // (Y_f, Z_f) are intact, the iteration goes on from there and only the float32 floor ends it now
        hipLaunchKernelGGL(nsf::nsf_rearm, dim3(1), dim3(64), 0, ws.job.stream, m.s32);
        FAD_HIP_TRY(hipGetLastError());
    }
    return FAD_OK;
}

// ==========================================================================================
// The same chain for a BATCH of songs against one baseline (fad_frechet_batched_vs_baseline, songs with at least D + 1 frames:
// every song is a full D x D problem): tr sqrt(Sigma_b Sigma_s) for B songs in eight launches of B times the workgroups.
// Replaces the per-song scipy.linalg.sqrtm / eig of fadtk/fad.py:373-378 for those songs; songs whose product the chain does not
// accept (spread spectra, non-finite input) are handed back to the float64 routes.  D in {128, 256, 384, 512, 768, 1024}.
// ==========================================================================================
static bool fast_song_dim(int d) { return d == 128 || d == 256 || d == 384 || d == 512 || d == 768 || d == 1024; }

struct SongBlock {                               // byte offsets inside one song's device block, and its size
    size_t hdr, st, s32, partials, stats, A64, P, Y[2], Z[2], T, digS, digY[2], digYt[2], stride;
};
static SongBlock song_block(int d) {
    const size_t dd = (size_t)d * d, nb = (size_t)d / 32;
    SongBlock b; size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    b.hdr = take(sizeof(nsf::MatHdr)); b.st = take(sizeof(NsState)); b.s32 = take(sizeof(Ns32State));
    b.partials = take(nb * nb * sizeof(double)); b.stats = take(nsf::kTileStats * nb * nb * sizeof(double));
    b.A64 = take(8 * dd); b.P = take(8 * dd);
    b.Y[0] = take(8 * dd); b.Y[1] = take(8 * dd); b.Z[0] = take(8 * dd); b.Z[1] = take(8 * dd); b.T = take(8 * dd);
    b.digS = take(6 * dd);
    b.digY[0] = take(6 * dd); b.digYt[0] = take(6 * dd); b.digY[1] = take(6 * dd); b.digYt[1] = take(6 * dd);
    b.stride = o;
    return b;
}
static size_t song_host_stride(int d) {
    const size_t nb = (size_t)d / 32;
    return ((nsf::kHostVals + (nsf::kTileStats + 2) * nb * nb) * sizeof(double) + nsf::kHostWords * sizeof(int) + 63) & ~(size_t)63;
}
static int64_t fast_songs_capacity(int d, size_t budget_bytes) {
    const int64_t n = (int64_t)(budget_bytes / song_block(d).stride);
    return n < 1 ? 1 : (n > 16384 ? 16384 : n);
}

// covs: B covariances [d x d] float64 on the device; -> tr_sqrt[b] and ok[b] (1: accepted, 0: hand the song to the float64 routes)
static int fast_songs(int d, int64_t B, const double* dcov_b, const double* covs, hipStream_t st, Workspace& ws,
                      std::vector<double>& tr_sqrt, std::vector<char>& ok, int device) {
    // FAD_SONG_BIG = smallest batch that iterates on the 128 x 128 tiles of ns_fast_big.h (default 8: a handful of songs fills the chip
    // only on 32 x 32 tiles; 0 = never; read per call -- tests force either kernel family on the same songs)
    const char* big_env = getenv("FAD_SONG_BIG");
    const long big_min = big_env ? atol(big_env) : 8;
    const bool big = big_min > 0 && B >= big_min;
    // FAD_SONG_RES=0: D = 128 iterates through the batched kernels like the other dimensions (tests compare)
    const char* res_env = getenv("FAD_SONG_RES");
    const bool resident = d == 128 && !(res_env && res_env[0] == '0');
    const bool res_full = resident && !(res_env && res_env[0] == '1');      // 1: only the iteration resident; default: the exact products too
    const size_t dd = (size_t)d * d;
    const int nb = d / 32;
    const SongBlock L = song_block(d);
    const size_t hs = song_host_stride(d);
    FAD_TRY(ws.fast_songs.reserve(512 + 6 * dd + (size_t)B * L.stride));
    if (!ws.fast_songs_pin || ws.fast_songs_pin_cap < (size_t)B * hs) {
        if (ws.fast_songs_pin) (void)hipHostFree(ws.fast_songs_pin);
        ws.fast_songs_pin = nullptr; ws.fast_songs_pin_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.fast_songs_pin, (size_t)B * hs + 4096, hipHostMallocDefault));
        ws.fast_songs_pin_cap = (size_t)B * hs + 4096;
    }
    char* base = static_cast<char*>(ws.fast_songs.p);
    nsf::MatHdr* hdr_b = reinterpret_cast<nsf::MatHdr*>(base);
    uint4* dig_b = reinterpret_cast<uint4*>(base + 512);
    char* blk = base + 512 + ((6 * dd + 255) & ~(size_t)255);
    char* hpin = static_cast<char*>(ws.fast_songs_pin);
    double* h_vals = reinterpret_cast<double*>(hpin);
    double* h_stats = h_vals + nsf::kHostVals;
    int* h_words = reinterpret_cast<int*>(h_stats + (size_t)(nsf::kTileStats + 2) * nb * nb);
    auto at = [&](size_t off) { return blk + off; };
    auto mat = [&](size_t off) { nsf::SplitMat m; m.a = reinterpret_cast<uint4*>(at(off)); m.at = reinterpret_cast<uint4*>(at(off + 4 * dd)); return m; };
    const int gen = ++ws.fast_gen;
    if (ws.fast_gen > (1 << 30)) ws.fast_gen = 1;
    // fresh headers carry no stale token
    FAD_HIP_TRY(hipMemsetAsync(hdr_b, 0, sizeof(nsf::MatHdr), st));
    FAD_HIP_TRY(hipMemset2DAsync(at(L.hdr), L.stride, 0, sizeof(nsf::MatHdr), (size_t)B, st));

    nsf::PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.cov_in[0] = dcov_b; pa.cov_in[1] = covs; pa.d = d; pa.ddof = 1; pa.gen = gen; pa.mean_dtype = -1;
    pa.dig[0] = dig_b; pa.dig[1] = reinterpret_cast<uint4*>(at(L.digS));
    pa.st = reinterpret_cast<NsState*>(at(L.st)); pa.hdr[0] = hdr_b; pa.hdr[1] = reinterpret_cast<nsf::MatHdr*>(at(L.hdr));
    pa.batch = 1; pa.pstride = (int64_t)L.stride;
    hipLaunchKernelGGL(nsf::nsf_prepare, dim3((unsigned)(dd / 2048), (unsigned)(1 + B)), dim3(512), 0, st, pa);

    NsState* st0 = reinterpret_cast<NsState*>(at(L.st));
    Ns32State* s32_0 = reinterpret_cast<Ns32State*>(at(L.s32));
    double* partials = reinterpret_cast<double*>(at(L.partials));
    double* stats = reinterpret_cast<double*>(at(L.stats));
    double* A64 = reinterpret_cast<double*>(at(L.A64));
    const nsf::SplitMat P = mat(L.P), Y[2] = {mat(L.Y[0]), mat(L.Y[1])}, Z[2] = {mat(L.Z[0]), mat(L.Z[1])}, T = mat(L.T);
    uint4* digY[2] = {reinterpret_cast<uint4*>(at(L.digY[0])), reinterpret_cast<uint4*>(at(L.digY[1]))};
    uint4* digYt[2] = {reinterpret_cast<uint4*>(at(L.digYt[0])), reinterpret_cast<uint4*>(at(L.digYt[1]))};
    auto split_args = [&]() {
        nsf::SplitArgs g;
        memset(&g, 0, sizeof(g));
        g.d = d; g.gen = gen; g.hA = hdr_b; g.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); g.pstride = (int64_t)L.stride;
        g.st = st0; g.s32 = s32_0;
        return g;
    };
    {
        nsf::I8Args a;
        memset(&a, 0, sizeof(a));
        a.Adig = dig_b; a.Bdig = reinterpret_cast<uint4*>(at(L.digS)); a.d = d; a.gen = gen; a.hA = hdr_b;
        a.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); a.pstride = (int64_t)L.stride; a.stats = stats; a.A64 = A64; a.P = P; a.st = st0;
        if (res_full) { /* nsf_res128<FULL> forms the product itself */ }
        else if (big && d >= 256) FAD_TRY(fast_i8_big(d, nsf::I8_A, a, st, (unsigned)B, device));
        else fast_i8(d, nsf::I8_A, a, st, (unsigned)B);
        if (resident) {
            // D = 128: the whole iteration of a song in one workgroup (ns_fast_res.h)
            static std::atomic<unsigned> ready{0};
            if (device >= 0 && device < 32 && !(ready.load(std::memory_order_acquire) & (1u << device))) {
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_res128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kResLds));
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_res128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kResLdsFull));
                ready.fetch_or(1u << device, std::memory_order_release);
            }
            nsf::ResArgs r;
            memset(&r, 0, sizeof(r));
            r.gen = gen; r.max_low = kMaxLow; r.thr_pred = pred_threshold(ws.pool, d); r.hA = hdr_b; r.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr));
            r.pstride = (int64_t)L.stride; r.A64 = A64; r.statsA = stats; r.st = st0; r.s32 = s32_0;
            r.Y[0] = Y[0]; r.Y[1] = Y[1]; r.Z[0] = Z[0]; r.Z[1] = Z[1];
            if (res_full) {
                // ... and the two exact products with it: A = Sigma_b Sigma_s in front, the correction behind; the host record is this kernel's
                r.Adig = dig_b; r.Bdig = reinterpret_cast<uint4*>(at(L.digS)); r.hstride = (int64_t)hs;
                r.stats = h_stats; r.host_words = h_words; r.host_vals = h_vals;
                for (int64_t b = 0; b < B; ++b) reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + b * hs)[12] = 0;
                hipLaunchKernelGGL(nsf::nsf_res128<true>, dim3((unsigned)B), dim3(256), nsf::kResLdsFull, st, r);
            } else {
                hipLaunchKernelGGL(nsf::nsf_res128<false>, dim3((unsigned)B), dim3(256), nsf::kResLds, st, r);
            }
        } else {
            nsf::SplitArgs g = split_args();
            g.A[0] = P; g.B[0] = P; g.C[0] = Y[1]; g.C[1] = Z[1]; g.A64 = A64; g.statsA = stats;      // (no digit planes: nsf_digitize, below)
            if (big && d >= 256) FAD_TRY(fast_split_big(d, nsf::SP_FIRST, g, st, (unsigned)B, device));
            else fast_split(d, nsf::SP_FIRST, g, st, (unsigned)B);
        }
    }
    tr_sqrt.assign((size_t)B, 0.0); ok.assign((size_t)B, 0);
    std::vector<char> settled((size_t)B, 0);
    // iterations 1..8 blind (a song of 2 D .. 20 D frames needs 7-11: its product has a condition number of a few hundred), then the
    // correction for the songs whose check finished them; if any is still iterating, the rest of the budget in one go (every song
    // stops itself: the launches of a finished song exit at once)
    static const bool trace = [] { const char* e = getenv("FAD_FAST_TRACE"); return e && e[0] == '1'; }();
    int k = 1, upto = resident ? 1 : 9;                 // (resident: nothing left to launch but the correction)
    for (;;) {
        for (; k < upto; ++k) {
            const int cur = k & 1;
            nsf::SplitArgs g = split_args();
            g.A[0] = Z[cur]; g.B[0] = Y[cur]; g.C[0] = T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f;
            g.partials = partials; g.skip = &s32_0->done;
            if (big) FAD_TRY(fast_split_big(d, nsf::SP_T, g, st, (unsigned)B, device));
            else fast_split(d, nsf::SP_T, g, st, (unsigned)B);
            g = split_args();
            g.A[0] = Y[cur]; g.B[0] = T; g.C[0] = Y[cur ^ 1]; g.A[1] = T; g.B[1] = Z[cur]; g.C[1] = Z[cur ^ 1];
            g.skip = &s32_0->upd_skip[k & 1];
            g.k = k; g.max_low = kMaxLow; g.nslots = big ? (d / 128) * (d / 128) : nb * nb; g.chk_partials = partials; g.thr_pred = pred_threshold(ws.pool, d);
            if (big) FAD_TRY(fast_split_big(d, nsf::SP_U, g, st, (unsigned)B, device));
            else fast_split(d, nsf::SP_U, g, st, (unsigned)B);
        }
        if (!res_full) {
            nsf::DigArgs dg;
            memset(&dg, 0, sizeof(dg));
            dg.d = d; dg.gen = gen; dg.hA = hdr_b; dg.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); dg.pstride = (int64_t)L.stride; dg.s32 = s32_0;
            dg.Y[0] = Y[0]; dg.Y[1] = Y[1]; dg.dig[0] = digY[0]; dg.dig[1] = digY[1]; dg.dig_t[0] = digYt[0]; dg.dig_t[1] = digYt[1];
            hipLaunchKernelGGL(nsf::nsf_digitize, dim3((unsigned)((dd / 16 + 255) / 256), 2, (unsigned)B), dim3(256), 0, st, dg);
        }
        if (!res_full) {
            nsf::I8Args a;
            memset(&a, 0, sizeof(a));
            a.Adig = digY[0]; a.Bdig = digYt[0]; a.Adig_alt = digY[1]; a.Bdig_alt = digYt[1]; a.sel = &s32_0->final_iter;
            a.d = d; a.gen = gen; a.hA = hdr_b; a.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); a.pstride = (int64_t)L.stride; a.hstride = (int64_t)hs;
            a.skip = &s32_0->skip_corr; a.stats = h_stats; a.st = st0; a.A64in = A64;
            a.Y[0] = Y[0]; a.Y[1] = Y[1]; a.Z[0] = Z[0]; a.Z[1] = Z[1]; a.s32 = s32_0; a.host_words = h_words; a.host_vals = h_vals;
            for (int64_t b = 0; b < B; ++b) reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + b * hs)[12] = 0;
            if (big && d >= 256) FAD_TRY(fast_i8_big(d, nsf::I8_G, a, st, (unsigned)B, device));
            else fast_i8(d, nsf::I8_G, a, st, (unsigned)B);
        }
        FAD_HIP_TRY(hipGetLastError());
        FAD_HIP_TRY(hipStreamSynchronize(st));
        bool pending = false;
        for (int64_t b = 0; b < B; ++b) {
            if (settled[b]) continue;
            const int* hw = reinterpret_cast<const int*>(reinterpret_cast<const char*>(h_words) + b * hs);
            const double* hv = reinterpret_cast<const double*>(reinterpret_cast<const char*>(h_vals) + b * hs);
            const double* hx = reinterpret_cast<const double*>(reinterpret_cast<const char*>(h_stats) + b * hs);
            if (hw[12] != gen) return set_error(FAD_ERR_HIP, "the correction kernel of the batched fast chain left no result for song %lld", (long long)b);
            MixedResult r;
            fast_decide_one(hw, hv, hx, nb, &r);
            if (r.status == 0 && k < kMaxLow && !resident) { pending = true; continue; }      // not finished yet: more iterations for this song
            settled[b] = 1;
            if (r.status == 1) { ok[b] = 1; tr_sqrt[b] = std::sqrt(r.c) * r.tr_scaled; }
            if (trace)
                fprintf(stderr, "[fad fast songs] song %lld: status %d iters %d decided_at %d res %.3e est %.3e tr %.6e c %.3e (words bad %d done %d ok %d failed %d skipped %d)\n",
                        (long long)b, r.status, r.iters, r.decided_at, r.res, r.est, r.tr_scaled, r.c, hw[0], hw[1], hw[5], hw[6], hw[11]);
        }
        if (!pending) break;
        upto = kMaxLow;
    }
    return FAD_OK;
}


// ==========================================================================================
// The chain for B INDEPENDENT PAIRS of packed moments in one sequence of launches (fad_frechet_from_moments_multi_begin): the
// same eight kernels as a single pair with B times the workgroups -- a launch of this chain costs ~4 us before it does
// anything and its workgroups are latency-bound, so B scores cost little more than one (bench.py keeps several scores in
// flight: their chains are enqueued as ONE batch).  Every buffer of pair b lives b * stride bytes behind pair 0's.
// ==========================================================================================
struct PairBlock {
    size_t hdrA, hdrB, st, s32, partials, stats, A64, P, Y[2], Z[2], T, digA, digB, digY[2], digYt[2], mus, covs, stride;
};
static PairBlock pair_block(int d) {
    const size_t dd = (size_t)d * d, nb = (size_t)d / 32;
    PairBlock b; size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    b.hdrA = take(sizeof(nsf::MatHdr)); b.hdrB = take(sizeof(nsf::MatHdr)); b.st = take(sizeof(NsState)); b.s32 = take(sizeof(Ns32State));
    b.partials = take(nb * nb * sizeof(double)); b.stats = take(nsf::kTileStats * nb * nb * sizeof(double));
    b.A64 = take(8 * dd); b.P = take(8 * dd);
    b.Y[0] = take(8 * dd); b.Y[1] = take(8 * dd); b.Z[0] = take(8 * dd); b.Z[1] = take(8 * dd); b.T = take(8 * dd);
    b.digA = take(6 * dd); b.digB = take(6 * dd);
    b.digY[0] = take(6 * dd); b.digYt[0] = take(6 * dd); b.digY[1] = take(6 * dd); b.digYt[1] = take(6 * dd);
    b.mus = take(2 * (size_t)d * sizeof(double)); b.covs = take(2 * dd * sizeof(double));
    b.stride = o;
    return b;
}

// enqueue: K1 (pairs mode) .. K8 for B pairs; nothing is waited for
static int pairs_enqueue(Workspace& ws, int d, int B, const fad_moments_t* const* h1, const fad_moments_t* const* h2, int ddof,
                         int mean_dtype, hipStream_t st) {
    const size_t dd = (size_t)d * d;
    const int nb = d / 32;
    // FAD_PAIRS_BIG = smallest batch whose products run on the 128 x 128 / 128 x 64 tiles of ns_fast_big.h (their operand traffic per
    // block is a tenth of the 32 x 32 kernels'; below, too few workgroups to fill the chip); read per call, 0 = never
    const char* big_env = getenv("FAD_PAIRS_BIG");
    const long big_min = big_env ? atol(big_env) : 3;
    const bool big = big_min > 0 && B >= big_min && d >= 256;
    const int device = ws.job.device;
    const PairBlock L = pair_block(d);
    const size_t hs = song_host_stride(d);
    void* const before = ws.fast_pairs.p;
    FAD_TRY(ws.fast_pairs.reserve((size_t)(B > 4 ? 8 : 4) * L.stride + 256));      // (room for a full batch at once: no regrowth between calls)
    if (!ws.fast_pairs_pin || ws.fast_pairs_pin_cap < (size_t)B * hs) {
        if (ws.fast_pairs_pin) (void)hipHostFree(ws.fast_pairs_pin);
        ws.fast_pairs_pin = nullptr; ws.fast_pairs_pin_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.fast_pairs_pin, (size_t)8 * hs + 4096, hipHostMallocDefault));
        ws.fast_pairs_pin_cap = (size_t)8 * hs + 4096;
    }
    char* blk = static_cast<char*>(ws.fast_pairs.p);
    char* hpin = static_cast<char*>(ws.fast_pairs_pin);
    double* h_vals = reinterpret_cast<double*>(hpin);
    double* h_stats = h_vals + nsf::kHostVals;
    int* h_words = reinterpret_cast<int*>(h_stats + (size_t)(nsf::kTileStats + 2) * nb * nb);
    auto at = [&](size_t off) { return blk + off; };
    auto mat = [&](size_t off) { nsf::SplitMat m; m.a = reinterpret_cast<uint4*>(at(off)); m.at = reinterpret_cast<uint4*>(at(off + 4 * dd)); return m; };
    const int gen = ++ws.fast_gen;
    if (ws.fast_gen > (1 << 30)) ws.fast_gen = 1;
    ws.multi.gen = gen;
    if (ws.fast_pairs.p != before) {               // fresh headers carry no stale token
        FAD_HIP_TRY(hipMemset2DAsync(at(L.hdrA), L.stride, 0, sizeof(nsf::MatHdr), (size_t)B, st));
        FAD_HIP_TRY(hipMemset2DAsync(at(L.hdrB), L.stride, 0, sizeof(nsf::MatHdr), (size_t)B, st));
    }
    nsf::MatHdr* hA = reinterpret_cast<nsf::MatHdr*>(at(L.hdrA));
    nsf::MatHdr* hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdrB));
    NsState* st0 = reinterpret_cast<NsState*>(at(L.st));
    Ns32State* s32_0 = reinterpret_cast<Ns32State*>(at(L.s32));
    double* partials = reinterpret_cast<double*>(at(L.partials));
    double* stats = reinterpret_cast<double*>(at(L.stats));
    double* A64 = reinterpret_cast<double*>(at(L.A64));
    const nsf::SplitMat P = mat(L.P), Y[2] = {mat(L.Y[0]), mat(L.Y[1])}, Z[2] = {mat(L.Z[0]), mat(L.Z[1])}, T = mat(L.T);
    uint4* digY[2] = {reinterpret_cast<uint4*>(at(L.digY[0])), reinterpret_cast<uint4*>(at(L.digY[1]))};
    uint4* digYt[2] = {reinterpret_cast<uint4*>(at(L.digYt[0])), reinterpret_cast<uint4*>(at(L.digYt[1]))};

    nsf::PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    for (int b = 0; b < B; ++b) { pa.accs[2 * b] = moments_packed(h1[b]); pa.accs[2 * b + 1] = moments_packed(h2[b]); }
    pa.acc[0] = pa.accs[0]; pa.acc[1] = pa.accs[1];
    pa.d = d; pa.ddof = ddof; pa.gen = gen; pa.mean_dtype = mean_dtype;
    pa.mus = reinterpret_cast<double*>(at(L.mus)); pa.covs = reinterpret_cast<double*>(at(L.covs));
    pa.dig[0] = reinterpret_cast<uint4*>(at(L.digA)); pa.dig[1] = reinterpret_cast<uint4*>(at(L.digB));
    pa.st = st0; pa.hdr[0] = hA; pa.hdr[1] = hB;
    pa.batch = 2; pa.pstride = (int64_t)L.stride;
    hipLaunchKernelGGL(nsf::nsf_prepare, dim3((unsigned)(dd / 2048 + 1), (unsigned)(2 * B)), dim3(512), 0, st, pa);

    auto split_args = [&]() {
        nsf::SplitArgs g;
        memset(&g, 0, sizeof(g));
        g.d = d; g.gen = gen; g.hA = hA; g.hB = hB; g.pstride = (int64_t)L.stride; g.astride = (int64_t)L.stride;
        g.st = st0; g.s32 = s32_0;
        return g;
    };
    {
        nsf::I8Args a;
        memset(&a, 0, sizeof(a));
        a.Adig = reinterpret_cast<uint4*>(at(L.digA)); a.Bdig = reinterpret_cast<uint4*>(at(L.digB)); a.d = d; a.gen = gen; a.hA = hA; a.hB = hB;
        a.pstride = (int64_t)L.stride; a.astride = (int64_t)L.stride; a.stats = stats; a.A64 = A64; a.P = P; a.st = st0;
        if (big) FAD_TRY(fast_i8_big(d, nsf::I8_A, a, st, (unsigned)B, device));
        else fast_i8(d, nsf::I8_A, a, st, (unsigned)B);
        nsf::SplitArgs g = split_args();
        g.A[0] = P; g.B[0] = P; g.C[0] = Y[1]; g.C[1] = Z[1]; g.A64 = A64; g.statsA = stats;
        if (big) FAD_TRY(fast_split_big(d, nsf::SP_FIRST, g, st, (unsigned)B, device));      // (digit planes of the final Y only: nsf_digitize below)
        else { g.Cdig[0] = digY[1]; g.Cdig_t[0] = digYt[1]; fast_split(d, nsf::SP_FIRST, g, st, (unsigned)B); }
    }
    int want = ws.pool ? ws.pool->lp_iters : 5;
    if (want < 2) want = 2;                          // (a pair that is not through after the blind batch goes the single way)
    if (want > kMaxLow) want = kMaxLow;
    for (int k = 1; k < want; ++k) {
        const int cur = k & 1;
        nsf::SplitArgs g = split_args();
        g.A[0] = Z[cur]; g.B[0] = Y[cur]; g.C[0] = T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f;
        g.partials = partials; g.skip = &s32_0->done;
        if (big) FAD_TRY(fast_split_big(d, nsf::SP_T, g, st, (unsigned)B, device));
        else fast_split(d, nsf::SP_T, g, st, (unsigned)B);
        g = split_args();
        g.A[0] = Y[cur]; g.B[0] = T; g.C[0] = Y[cur ^ 1]; g.A[1] = T; g.B[1] = Z[cur]; g.C[1] = Z[cur ^ 1];
        if (!big) { g.Cdig[0] = digY[cur ^ 1]; g.Cdig_t[0] = digYt[cur ^ 1]; }
        g.skip = &s32_0->upd_skip[k & 1];
        g.k = k; g.max_low = kMaxLow; g.nslots = big ? (d / 128) * (d / 128) : nb * nb; g.chk_partials = partials; g.thr_pred = pred_threshold(ws.pool, d);
        if (big) FAD_TRY(fast_split_big(d, nsf::SP_U, g, st, (unsigned)B, device));
        else fast_split(d, nsf::SP_U, g, st, (unsigned)B);
    }
    if (big) {
        nsf::DigArgs dg;
        memset(&dg, 0, sizeof(dg));
        dg.d = d; dg.gen = gen; dg.hA = hA; dg.hB = hB; dg.pstride = (int64_t)L.stride; dg.astride = (int64_t)L.stride; dg.s32 = s32_0;
        dg.Y[0] = Y[0]; dg.Y[1] = Y[1]; dg.dig[0] = digY[0]; dg.dig[1] = digY[1]; dg.dig_t[0] = digYt[0]; dg.dig_t[1] = digYt[1];
        hipLaunchKernelGGL(nsf::nsf_digitize, dim3((unsigned)((dd / 16 + 255) / 256), 2, (unsigned)B), dim3(256), 0, st, dg);
    }
    {
        nsf::I8Args a;
        memset(&a, 0, sizeof(a));
        a.Adig = digY[0]; a.Bdig = digYt[0]; a.Adig_alt = digY[1]; a.Bdig_alt = digYt[1]; a.sel = &s32_0->final_iter;
        a.d = d; a.gen = gen; a.hA = hA; a.hB = hB; a.pstride = (int64_t)L.stride; a.astride = (int64_t)L.stride; a.hstride = (int64_t)hs;
        a.skip = &s32_0->skip_corr; a.stats = h_stats; a.st = st0; a.A64in = A64;
        a.Y[0] = Y[0]; a.Y[1] = Y[1]; a.Z[0] = Z[0]; a.Z[1] = Z[1]; a.s32 = s32_0; a.host_words = h_words; a.host_vals = h_vals;
        for (int b = 0; b < B; ++b) reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + (size_t)b * hs)[12] = 0;
        if (big) FAD_TRY(fast_i8_big(d, nsf::I8_G, a, st, (unsigned)B, device));
        else fast_i8(d, nsf::I8_G, a, st, (unsigned)B);
    }
    FAD_HIP_TRY(hipGetLastError());
    if (!ws.done_ev) FAD_HIP_TRY(hipEventCreateWithFlags(&ws.done_ev, hipEventDisableTiming));
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, st));
    return FAD_OK;
}

// Enqueue the whole low-precision chain of ONE problem on `stream` (nothing is waited for): C1 C2, statistics, scale,
// iteration 0, the blind batch of iterations, the closing kernels.  The state words must have been cleared.
static int mixed_begin(const NsProblem& pb, int device, hipStream_t stream, Workspace& ws) {
    const int d = pb.d;
    const int64_t dd = (int64_t)d * d;
    FAD_TRY(ws.mats.reserve((size_t)(6 * dd) * sizeof(double)));
    FAD_TRY(ws.mats32.reserve((size_t)(5 * dd) * sizeof(float)));
    const size_t hbytes = ws.job.fast ? fast_pinned_bytes(d) : sizeof(NsState) + sizeof(MixedResult);
    if (!ws.pinned || ws.pinned_cap < hbytes) {
        if (ws.pinned) (void)hipHostFree(ws.pinned);
        ws.pinned = nullptr; ws.pinned_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.pinned, hbytes + 4096, hipHostMallocDefault));
        ws.pinned_cap = hbytes + 4096;
    }
    MixedBufs m = mixed_bufs(ws, d);
    m.hres->status = -1;
    ws.job.d = d; ws.job.device = device; ws.job.stream = stream; ws.job.k = 0;
    if (ws.job.fast) {                           // (nsf_prepare is on the stream already: fast_prepare)
        ws.job.mu1 = pb.mu1; ws.job.mu2 = pb.mu2; ws.job.mean_dtype = pb.mean_dtype;
        int want = ws.pool ? ws.pool->lp_iters : 5;
        if (want < 2) want = 2;
        if (want > kMaxLow) want = kMaxLow;
        if (ws.pool && ws.pool->lp_hopeless) want = 1;       // (a pair that does iterate is topped up by mixed_finish)
        return fast_enqueue(ws, want);
    }
    // A = C1 C2 with its tile statistics from the epilogue and the mean term from a spare workgroup (one launch instead of
    // product + ns_tilestats), then the scale
    NsProductExt ext;
    memset(&ext, 0, sizeof(ext));
    ext.stats = m.tilestats; ext.mu1 = pb.mu1; ext.mu2 = pb.mu2; ext.mean_dtype = pb.mean_dtype; ext.st = m.dstate;
    FAD_TRY(gemm_f64_product_stats_launch(d, pb.cov1, pb.cov2, m.A, &m.dstate->done, ext, stream));
    hipLaunchKernelGGL(ns_prepare, dim3(1), dim3(256), 0, stream, m.tilestats, d, (int)m.nb, pb.mu1, (int64_t)0, pb.mu2,
                       (int64_t)0, pb.mean_dtype, m.dstate, 1, m.s32);
    int want = ws.pool ? ws.pool->lp_iters : 5;
    if (want < 2) want = 2;
    if (want > kMaxLow) want = kMaxLow;
    return mixed_enqueue(ws, want);
}

// Wait for the chain, top it up two iterations at a time while the device says "not finished yet".
// -> FAD_OK with res->status 1 (accepted: res holds the pieces) or 2 (run the fp64 iteration).
static int mixed_finish(Workspace& ws, MixedResult* res) {
    MixedBufs m = mixed_bufs(ws, ws.job.d);
    for (;;) {
        FAD_HIP_TRY(hipEventSynchronize(ws.done_ev));
        if (ws.job.fast) FAD_TRY(fast_decide(ws));
        if (m.hres->status == 4) {                 // predicted final iterate rejected: go on from it (state re-armed on the device)
            ws.job.k = m.hres->iters;
            m.hres->status = 0;
        } else if (m.hres->status != 0 || ws.job.k >= kMaxLow) {
            break;
        }
        if (ws.job.k >= kMaxLow) break;
        const int upto = (ws.job.k + 2 < kMaxLow) ? ws.job.k + 2 : kMaxLow;
        FAD_TRY(ws.job.fast ? fast_enqueue(ws, upto) : mixed_enqueue(ws, upto));
    }
    *res = *m.hres;
    if (res->status == 0) res->status = 2;
    if (res->status == 1 && res->decided_at >= 0 && ws.pool) ws.pool->lp_iters = res->decided_at + 1;
    if (ws.pool && ws.job.fast) ws.pool->lp_hopeless = res->status == 2 && res->iters < 0;      // given up before the first check
    return FAD_OK;
}

static bool mixed_eligible(Workspace& ws, int d, int max_iter, double tol) {
    Pool* p = ws.pool;
    if (p && p->mixed < 0) { const char* e = getenv("FAD_FRECHET_MIXED"); p->mixed = (e && e[0] == '0') ? 0 : 1; }
    return (!p || p->mixed) && d % 64 == 0 && max_iter <= 0 && tol <= 0.0;
}

// single pair, with the reference's eps fallback; cov/mu are DEVICE pointers
static int frechet_single(int d, const double* cov1, const double* cov2, const double* mu1, const double* mu2,
                          double eps, int max_iter, double tol, int mean_dtype, int device, hipStream_t stream,
                          Workspace& ws, double* out_fad, fad_diag_t* diag, bool check_few) {
    const int64_t dd = (int64_t)d * d;
    NsState* hs = nullptr;
    bool reuse = false;
    NsProblem pb{d, 1, cov1, 0, cov2, 0, mu1, 0, mu2, 0, mean_dtype};
    if (mixed_eligible(ws, d, max_iter, tol)) {
        MixedResult r;
        if (!ws.job.mixed) FAD_TRY(mixed_begin(pb, device, stream, ws));       // (an async job enqueued it already)
        ws.job.mixed = false;
        FAD_TRY(mixed_finish(ws, &r));
        if (check_few && (r.too_few0 || r.too_few1))
            return set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames in each set");
        if (r.status == 1) {
            const double tr_sqrt = sqrt(r.c) * r.tr_scaled;
            if (out_fad) *out_fad = r.mean_term + r.tr1 + r.tr2 - 2.0 * tr_sqrt;
            if (diag) {
                diag->iters = r.iters + 1; diag->converged = 3; diag->used_eps = 0; diag->route = ws.job.fast ? 2 : 1;
                diag->residual = r.res; diag->scale = r.c; diag->mean_term = r.mean_term; diag->tr1 = r.tr1; diag->tr2 = r.tr2;
                diag->tr_sqrt = tr_sqrt;
            }
            return FAD_OK;
        }
        // rejected: the float64 iteration takes over from the product C1 C2 and the state that was armed for this very
        // problem (nothing in the low-precision leg writes to either) -- unless the chain never got that far (prepared = 0)
        reuse = r.prepared != 0;
    }
    FAD_TRY(run_ns(pb, max_iter, tol, device, stream, ws, &hs, reuse));
    if (check_few && (hs->too_few[0] || hs->too_few[1]))
        return set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames in each set");
    bool used_eps = false;
    if (hs->nonfinite && eps > 0.0) {
        // fad.py:94-99: add eps to both diagonals and take the root again
        FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));     // keeps the first 2dd+2d intact
        double* E1 = static_cast<double*>(ws.stage.p) + 2 * dd + 2 * d;
        double* E2 = E1 + dd;
        FAD_HIP_TRY(hipMemcpyAsync(E1, cov1, dd * sizeof(double), hipMemcpyDeviceToDevice, stream));
        FAD_HIP_TRY(hipMemcpyAsync(E2, cov2, dd * sizeof(double), hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(add_diag, dim3((unsigned)cdiv(d, 256)), dim3(256), 0, stream, E1, d, eps);
        hipLaunchKernelGGL(add_diag, dim3((unsigned)cdiv(d, 256)), dim3(256), 0, stream, E2, d, eps);
        hipLaunchKernelGGL(clear_states, dim3(1), dim3(64), 0, stream, static_cast<NsState*>(ws.small.p), (int64_t)1);
        NsProblem pe{d, 1, E1, 0, E2, 0, mu1, 0, mu2, 0, mean_dtype};
        FAD_TRY(run_ns(pe, max_iter, tol, device, stream, ws, &hs));
        used_eps = true;
    }
    if (hs->nonfinite) {
        if (diag) { memset(diag, 0, sizeof(*diag)); diag->used_eps = used_eps; diag->residual = hs->res_last; }
        return set_error(FAD_ERR_NOT_FINITE,
                         "sqrt(C1 C2) did not stay finite (NaN/Inf input or a product with negative eigenvalues)");
    }
    const double tr_sqrt = sqrt(hs->c) * hs->tr_last;
    double tr1 = hs->tr1, tr2 = hs->tr2;          // traces refer to the caller's inputs (fad.py:119-120)
    if (used_eps) { tr1 -= eps * d; tr2 -= eps * d; }
    if (out_fad) *out_fad = hs->mean_term + tr1 + tr2 - 2.0 * tr_sqrt;
    if (diag) {
        diag->iters = hs->final_iter + 1; diag->converged = hs->conv; diag->used_eps = used_eps ? 1 : 0;
        diag->route = 0; diag->residual = hs->res_last; diag->scale = hs->c;
        diag->mean_term = hs->mean_term; diag->tr1 = tr1; diag->tr2 = tr2; diag->tr_sqrt = tr_sqrt;
    }
    if (hs->conv == 0)
        return set_error(FAD_ERR_NOT_CONVERGED, "Newton-Schulz stopped at max_iter with residual %.3e", hs->res_last);
    return FAD_OK;
}

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
        const double mr = mean_mode ? round_like_input<TIn>(m) : m;
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
                                                       double* __restrict__ part /*[S][chunks][2]*/) {
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
        const double mr = mean_mode ? round_like_input<TIn>(m) : m;
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
                                                      double* __restrict__ out /*[S][chunks][2]: scal itself when there is one chunk*/) {
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
            const double mr = mean_mode ? round_like_input<r_f16>(m) : m;
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
        const double mr = mean_mode ? round_like_input<TIn>(m) : m;
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

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int fad_frechet(int d, const double* mu1, const double* cov1, const double* mu2, const double* cov2,
                double eps, int max_iter, double tol, int on_device, int device, void* stream,
                double* out_fad, fad_diag_t* diag) {
    if (d < 1 || d > 16384) return set_error(FAD_ERR_INVALID, "d=%d out of range", d);
    if (!mu1 || !mu2 || !cov1 || !cov2 || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return set_error(FAD_ERR_HIP, "cannot select device %d", device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.job = Workspace::Job();
    FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
    const bool fast = fast_eligible(ws, d, max_iter, tol);
    if (!fast) hipLaunchKernelGGL(clear_states, dim3(1), dim3(64), 0, st, static_cast<NsState*>(ws.small.p), (int64_t)1);
    const int64_t dd = (int64_t)d * d;
    const double *dc1 = cov1, *dc2 = cov2, *dm1 = mu1, *dm2 = mu2;
    if (!on_device) {
        FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));
        double* s = static_cast<double*>(ws.stage.p);
        FAD_HIP_TRY(hipMemcpyAsync(s, cov1, dd * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + dd, cov2, dd * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + 2 * dd, mu1, d * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + 2 * dd + d, mu2, d * sizeof(double), hipMemcpyHostToDevice, st));
        dc1 = s; dc2 = s + dd; dm1 = s + 2 * dd; dm2 = s + 2 * dd + d;
    }
    if (fast) {                                    // state reset, scales and digit planes from the caller's matrices
        ws.job.fast = true;
        FAD_TRY(fast_prepare(ws, d, 1, nullptr, nullptr, dc1, dc2, dm1, dm2, -1, nullptr, nullptr, st));
    }
    return frechet_single(d, dc1, dc2, dm1, dm2, eps, max_iter, tol, -1, device, st, ws, out_fad, diag, false);
}

// A score from two moments handles: checks, scratch, the job record; (mu, Sigma) of both handles go to the slot's staging
// area and the iteration state is cleared.  (Forming Sigma inside the first product instead -- from the packed statistics,
// between registers and LDS -- was measured: 18.7 us against 12.4 + 4.8 for product + this launch; dropped.)
static int stage_from_moments(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps, int mean_dtype,
                              hipStream_t st, Workspace& ws, int max_iter = 0, double tol = 0.0) {
    if (!h1 || !h2) return set_error(FAD_ERR_INVALID, "NULL argument");
    const int d = moments_dim(h1), device = moments_device(h1);
    if (moments_dim(h2) != d)
        return set_error(FAD_ERR_SHAPE, "Training and test covariances have different dimensions (%d vs %d)", d, moments_dim(h2));
    if (moments_device(h2) != device) return set_error(FAD_ERR_INVALID, "handles live on different devices");
    FAD_TRY(moments_settle(h1, st));
    FAD_TRY(moments_settle(h2, st));
    FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
    const int64_t dd = (int64_t)d * d;
    FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));
    double* s = static_cast<double*>(ws.stage.p);
    ws.job = Workspace::Job();
    ws.job.d = d; ws.job.device = device; ws.job.stream = st; ws.job.eps = eps; ws.job.mean_dtype = mean_dtype; ws.job.ddof = ddof;
    ws.job.cov1 = s; ws.job.cov2 = s + dd; ws.job.mu1 = s + 2 * dd; ws.job.mu2 = s + 2 * dd + d;
    if (fast_eligible(ws, d, max_iter, tol)) {     // the eight-launch chain: its first kernel does this staging as well
        ws.job.fast = true;
        return fast_prepare(ws, d, ddof, moments_packed(h1), moments_packed(h2), nullptr, nullptr, nullptr, nullptr, mean_dtype, s + 2 * dd, s, st);
    }
    hipLaunchKernelGGL(finalize_for_frechet, dim3((unsigned)cdiv(dd, 256), 2), dim3(256), 0, st, moments_packed(h1), moments_packed(h2),
                       d, ddof, s + 2 * dd, s, static_cast<NsState*>(ws.small.p));
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

int fad_frechet_from_moments(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps,
                             int max_iter, double tol, int mean_dtype, void* stream, double* out_fad, fad_diag_t* diag) {
    if (!h1 || !h2 || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    const int device = moments_device(h1);
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.pool = &thread_pool(device);
    FAD_TRY(stage_from_moments(h1, h2, ddof, eps, mean_dtype, st, ws, max_iter, tol));
    const Workspace::Job j = ws.job;
    return frechet_single(j.d, j.cov1, j.cov2, j.mu1, j.mu2, eps, max_iter, tol, mean_dtype, device, st, ws, out_fad, diag, true);
}

int fad_frechet_from_moments_begin(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps, int mean_dtype,
                                   void* stream, fad_frechet_job_t** job) {
    if (!h1 || !h2 || !job) return set_error(FAD_ERR_INVALID, "NULL argument");
    *job = nullptr;
    const int device = moments_device(h1);
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.pool = &thread_pool(device);
    const bool mixed = mixed_eligible(ws, moments_dim(h1), 0, 0.0);
    FAD_TRY(stage_from_moments(h1, h2, ddof, eps, mean_dtype, st, ws));
    if (mixed) {
        const Workspace::Job& j = ws.job;
        NsProblem pb{j.d, 1, j.cov1, 0, j.cov2, 0, j.mu1, 0, j.mu2, 0, mean_dtype};
        FAD_TRY(mixed_begin(pb, device, st, ws));
        ws.job.mixed = true;
    }
    ws.busy = true;
    *job = reinterpret_cast<fad_frechet_job_t*>(wsp);
    return FAD_OK;
}

int fad_frechet_end(fad_frechet_job_t* job, double* out_fad, fad_diag_t* diag) {
    if (!job || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    Workspace& ws = *reinterpret_cast<Workspace*>(job);
    if (!ws.busy) return set_error(FAD_ERR_INVALID, "this job was collected already");
    DeviceGuard g(ws.job.device);
    ws.busy = false;                               // the slot is free again whatever happens below
    const Workspace::Job j = ws.job;
    // the low-precision chain is in flight (or nothing is: the synchronous path runs now); frechet_single collects /
    // tops up / falls back to the fp64 iteration exactly as the blocking entry point does
    return frechet_single(j.d, j.cov1, j.cov2, j.mu1, j.mu2, j.eps, 0, 0.0, j.mean_dtype, j.device, j.stream, ws, out_fad, diag, true);
}

int fad_frechet_from_moments_multi_begin(int count, const fad_moments_t* const* h1, const fad_moments_t* const* h2, int ddof, double eps,
                                         int mean_dtype, void* stream, fad_frechet_job_t** job) {
    if (!h1 || !h2 || !job) return set_error(FAD_ERR_INVALID, "NULL argument");
    *job = nullptr;
    if (count < 1 || count > 8) return set_error(FAD_ERR_INVALID, "count=%d out of range [1, 8]", count);
    for (int b = 0; b < count; ++b) if (!h1[b] || !h2[b]) return set_error(FAD_ERR_INVALID, "pair %d: NULL handle", b);
    const int device = moments_device(h1[0]), d = moments_dim(h1[0]);
    for (int b = 0; b < count; ++b) {
        if (moments_dim(h1[b]) != d || moments_dim(h2[b]) != d)
            return set_error(FAD_ERR_SHAPE, "Training and test covariances have different dimensions (pair %d: %d vs %d, pair 0: %d)", b,
                             moments_dim(h1[b]), moments_dim(h2[b]), d);
        if (moments_device(h1[b]) != device || moments_device(h2[b]) != device) return set_error(FAD_ERR_INVALID, "handles live on different devices");
    }
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.pool = &thread_pool(device);
    ws.job = Workspace::Job();
    ws.job.d = d; ws.job.device = device; ws.job.stream = st; ws.job.eps = eps; ws.job.mean_dtype = mean_dtype; ws.job.ddof = ddof;
    ws.multi = Workspace::Multi();
    ws.multi.count = count;
    for (int b = 0; b < count; ++b) { ws.multi.h1[b] = h1[b]; ws.multi.h2[b] = h2[b]; }
    if (fast_eligible(ws, d, 0, 0.0)) {
        for (int b = 0; b < count; ++b) { FAD_TRY(moments_settle(h1[b], st)); FAD_TRY(moments_settle(h2[b], st)); }
        FAD_TRY(pairs_enqueue(ws, d, count, h1, h2, ddof, mean_dtype, st));
        ws.multi.enqueued = true;
    }
    ws.busy = true;
    *job = reinterpret_cast<fad_frechet_job_t*>(wsp);
    return FAD_OK;
}

int fad_frechet_multi_end(fad_frechet_job_t* job, int count, double* out_fad, fad_diag_t* diag) {
    if (!job || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    Workspace& ws = *reinterpret_cast<Workspace*>(job);
    if (!ws.busy) return set_error(FAD_ERR_INVALID, "this job was collected already");
    if (count != ws.multi.count) return set_error(FAD_ERR_INVALID, "this job holds %d pairs, not %d", ws.multi.count, count);
    const Workspace::Job j = ws.job;
    const Workspace::Multi m = ws.multi;
    DeviceGuard g(j.device);
    const int d = j.d, nb = d / 32;
    bool done[8] = {false};
    int rc = FAD_OK;
    if (m.enqueued) {
        const hipError_t e = hipEventSynchronize(ws.done_ev);
        if (e != hipSuccess) { ws.busy = false; return set_error(FAD_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e)); }
        const size_t hs = song_host_stride(d);
        const char* hpin = static_cast<const char*>(ws.fast_pairs_pin);
        const size_t off_stats = nsf::kHostVals * sizeof(double), off_words = off_stats + (size_t)(nsf::kTileStats + 2) * nb * nb * sizeof(double);
        int learnt = -1;
        for (int b = 0; b < count; ++b) {
            const double* hv = reinterpret_cast<const double*>(hpin + (size_t)b * hs);
            const double* hx = reinterpret_cast<const double*>(hpin + (size_t)b * hs + off_stats);
            const int* hw = reinterpret_cast<const int*>(hpin + (size_t)b * hs + off_words);
            if (hw[12] != m.gen) continue;              // (no record: the single route decides)
            MixedResult r;
            fast_decide_one(hw, hv, hx, nb, &r);
            if (r.too_few0 || r.too_few1) {
                if (rc == FAD_OK) rc = set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames in each set");
                out_fad[b] = NAN; done[b] = true;
                continue;
            }
            if (r.status != 1) continue;
            const double tr_sqrt = sqrt(r.c) * r.tr_scaled;
            out_fad[b] = r.mean_term + r.tr1 + r.tr2 - 2.0 * tr_sqrt;
            if (diag) {
                fad_diag_t& q = diag[b];
                memset(&q, 0, sizeof(q));
                q.iters = r.iters + 1; q.converged = 3; q.used_eps = 0; q.route = 2;
                q.residual = r.res; q.scale = r.c; q.mean_term = r.mean_term; q.tr1 = r.tr1; q.tr2 = r.tr2; q.tr_sqrt = tr_sqrt;
            }
            if (r.decided_at + 1 > learnt) learnt = r.decided_at + 1;
            done[b] = true;
        }
        if (learnt > 0 && ws.pool) ws.pool->lp_iters = learnt;
    }
    ws.busy = false;                               // the slot is free again: the single route below takes any free one
    ws.multi = Workspace::Multi();
    for (int b = 0; b < count; ++b) {
        if (done[b]) continue;
        // not eligible, not finished within the blind batch, or rejected: exactly what the single entry point does for this pair
        const int r1 = fad_frechet_from_moments(m.h1[b], m.h2[b], j.ddof, j.eps, 0, 0.0, j.mean_dtype, j.stream, &out_fad[b], diag ? &diag[b] : nullptr);
        if (r1 != FAD_OK && rc == FAD_OK) rc = r1;
    }
    return rc;
}

int fad_frechet_cancel(fad_frechet_job_t* job) {
    if (!job) return set_error(FAD_ERR_INVALID, "NULL argument");
    Workspace& ws = *reinterpret_cast<Workspace*>(job);
    if (!ws.busy) return FAD_OK;
    DeviceGuard g(ws.job.device);
    // the enqueued kernels still write into the slot's buffers: it may only be handed out again once they are through
    if ((ws.job.mixed || ws.multi.enqueued) && ws.done_ev) FAD_HIP_TRY(hipEventSynchronize(ws.done_ev));
    else FAD_HIP_TRY(hipStreamSynchronize(ws.job.stream));
    ws.busy = false;
    ws.job = Workspace::Job();
    ws.multi = Workspace::Multi();
    return FAD_OK;
}

}  // extern "C"

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
static int batched_impl(int d, const double* dmu_b, const double* dcov_b, const TIn* drows, int64_t ld,
                        const int64_t* h_off, const int64_t* d_off, int64_t n_songs, int mean_mode, int device,
                        hipStream_t st, Workspace& ws, double* out_scores, int32_t* out_status) {
    const int64_t dd = (int64_t)d * d;
    std::vector<int64_t> pairs, gram, gram_ns, general;
    static const bool gram_on = [] { const char* e = getenv("FAD_SONG_GRAM"); return !(e && e[0] == '0'); }();
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
        static const bool stats16_on = [] { const char* e = getenv("FAD_SONG_STATS16"); return !(e && e[0] == '0'); }();
        if (std::is_same<TIn, r_f16>::value && stats16_on && (h_off[n_songs] - h_off[0]) / n_songs >= 64 && song_cov_f16_ok(drows, ld, d)) {
            const int chunks = (int)cdiv(d, 128);
            double* part = scal;
            if (chunks > 1) { FAD_TRY(ws.rows2.reserve((size_t)2 * n_songs * chunks * sizeof(double))); part = static_cast<double*>(ws.rows2.p); }
            hipLaunchKernelGGL(song_stats_f16, dim3((unsigned)n_songs, (unsigned)chunks), dim3(256), 0, st,
                               reinterpret_cast<const uint16_t*>(drows), ld, d, d_off, dmu_b, mean_mode, mean_exact,
                               mean_exact + (size_t)n_songs * d, part);
            var_exact = mean_exact + (size_t)n_songs * d;
            if (chunks > 1) hipLaunchKernelGGL(song_scal_sum, dim3((unsigned)cdiv(2 * n_songs, 256)), dim3(256), 0, st, part, chunks, n_songs, scal);
        } else if ((h_off[n_songs] - h_off[0]) / n_songs >= 64) {
            const int chunks = (int)cdiv(d, 64);
            FAD_TRY(ws.rows2.reserve((size_t)2 * n_songs * chunks * sizeof(double)));
            double* part = static_cast<double*>(ws.rows2.p);
            hipLaunchKernelGGL((song_stats_long<TIn>), dim3((unsigned)n_songs, (unsigned)chunks), dim3(256), 0, st, drows, ld, d, d_off,
                               dmu_b, mean_mode, mean_exact, part);
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
            hipLaunchKernelGGL(clear_states, dim3((unsigned)cdiv(ns, 64)), dim3(64), 0, st, dstates, ns);
            NsState* hs = nullptr;
            static const int sym_on = [] { const char* e = getenv("FAD_SONG_SYM"); return (e && e[0] == '0') ? 0 : 1; }();
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
    const char* fs_env = getenv("FAD_SONG_FAST");
    const int fastsongs_on = fs_env ? atoi(fs_env) : 1;
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
            static const bool cov16_on = [] { const char* e = getenv("FAD_SONG_COV16"); return !(e && e[0] == '0'); }();
            if (std::is_same<TIn, r_f16>::value && cov16_on && var_exact && song_cov_f16_ok(drows, ld, d)) {      // (only with the exact diagonal)
                int64_t max_frames = 0;
                for (int64_t b = 0; b < B; ++b) { const int64_t sg = general[g0 + b]; max_frames = std::max(max_frames, h_off[sg + 1] - h_off[sg]); }
                FAD_TRY(song_cov_f16_launch(drows, ld, d, d_off, ids_dev, B, max_frames, mean_exact, var_exact, covs, ws.songcov, device, st));
            } else {
                hipLaunchKernelGGL((song_cov_mfma<TIn>), dim3((unsigned)(nt64 * (nt64 + 1) / 2), 1, (unsigned)B), dim3(256), 0, st, drows,
                                   ld, d, nt64, d_off, ids_dev, mean_exact, covs);
            }
            FAD_TRY(fast_songs(d, B, dcov_b, covs, st, ws, trs, okv, device));       // (synchronises: `general` may be read again)
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
    static const bool symroute_on = [] { const char* e = getenv("FAD_SONG_SYM"); return !(e && e[0] == '0'); }();
    if (symroute_on && d >= 64 && !general.empty()) {
        std::vector<int64_t> sym_songs, rest;
        static const int64_t max_mult = [] { const char* e = getenv("FAD_SONG_SYM_MAX_FRAMES_PER_DIM"); return e ? (int64_t)atoll(e) : (int64_t)8; }();
        for (const int64_t sg : general) ((h_off[sg + 1] - h_off[sg] <= max_mult * d) ? sym_songs : rest).push_back(sg);
        bool have_root = false;
        double *broot = nullptr, *eye = nullptr, *zeros = nullptr;
        if (!sym_songs.empty()) {
            FAD_TRY(ws.base_root.reserve((size_t)(2 * dd + d) * sizeof(double)));
            broot = static_cast<double*>(ws.base_root.p); eye = broot + dd; zeros = eye + dd;
            hipLaunchKernelGGL(identity_and_zeros, dim3((unsigned)cdiv(dd, 256)), dim3(256), 0, st, eye, d, zeros);
            FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
            NsState* dstates = static_cast<NsState*>(ws.small.p);
            hipLaunchKernelGGL(clear_states, dim3(1), dim3(64), 0, st, dstates, (int64_t)1);
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
                hipLaunchKernelGGL(clear_states, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, dstates, B);
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
            hipLaunchKernelGGL(clear_states, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, dstates, B);
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

    switch (dtype) {
        case FAD_F16: return batched_impl<r_f16>(d, dmu, dcov, static_cast<const r_f16*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, out_scores, out_status);
        case FAD_BF16: return batched_impl<r_bf16>(d, dmu, dcov, static_cast<const r_bf16*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, out_scores, out_status);
        case FAD_F32: return batched_impl<float>(d, dmu, dcov, static_cast<const float*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, out_scores, out_status);
        default: return batched_impl<double>(d, dmu, dcov, static_cast<const double*>(drows), dld, offsets, d_off, n_songs, mean_mode, device, st, ws, out_scores, out_status);
    }
}
