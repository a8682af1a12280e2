// Frechet distance between Gaussians on the GPU (gfx950): the all-float64 Newton-Schulz iteration (frechet_internal.h lists the files).
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
#include "frechet_internal.h"
#include "ns_mean.h"

#include <cmath>

namespace fad {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- statistics of A = C1 C2 for the scale of the iteration, by 32 x 32 tile pairs -----------------------------
// Workgroup (bi, bj, problem) loads tile (bi, bj) of A and its mirror (bj, bi) -- both as coalesced 256-byte row
// segments -- and writes: sum |a| of the tile's 32 rows / 32 columns (partial infinity / one norms), and the scalars
// sum a^2 (Frobenius), sum a_ij a_ji (adds up to tr A^2), the tile's share of tr A, tr C1, tr C2.  ns_prepare adds the
// partials in a fixed order (deterministic).  (The first version gave every ROW its own workgroup, which read the
// matching column with a 4 KiB stride: 5.5 us at D = 512 for 2 MB of data.)

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
            s32->upd_skip[0] = 0; s32->upd_skip[1] = 0; s32->skip_corr = 1; s32->decided_at = -1; s32->strict = 0; s32->grew = 0;
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
            l = ns_l0_from_participation((float)(trA * trA / trA2), d);
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
                                                            NsState* __restrict__ st, const float* __restrict__ run1 = nullptr,
                                                            const float* __restrict__ run2 = nullptr) {
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
    const float* run = blockIdx.y ? run2 : run1;                    // (numpy's running sums: ns_fast.h, PrepArgs::run)
    if (g < d) mu[g] = run ? numpy_mean_of_f32_sum(run[g], n) : sum[g] / n;
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



// Enqueue + run the batched iteration.  On return host_states (pinned, B entries) holds the final
// per-problem state; the caller turns it into scores.  States must have been cleared by the caller
// (so that pre-kernels like finalize_for_frechet can raise too_few).
// reuse_prepared: A = C1 C2 (first matrix of ws.mats) and the armed state are those of a float32 attempt on the same problem
// that just gave up (mixed_begin: same buffer, same ns_prepare) -- product, statistics and scale are not formed again.
int run_ns(const NsProblem& pb, int max_iter, double tol, int device, hipStream_t stream, Workspace& ws,
           NsState** host_states, bool reuse_prepared, double** y_bufs, int first_chunk) {
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
    if (first_chunk > 0) chunk = first_chunk;          // (a batch of pairs: what the thread's last such batch needed, frechet.hip)
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

// problems of a batch that need no iteration (bit b of `mask`): closed before the first launch looks at them
__global__ void mark_states_done(NsState* st, uint32_t mask, int B) {
    const int b = threadIdx.x;
    if (b < B && ((mask >> b) & 1u)) { st[b].done = 1; st[b].finished = 1; st[b].upd_skip[0] = 1; st[b].upd_skip[1] = 1; st[b].conv = 1; st[b].final_iter = 0; }
}
void enqueue_mark_states_done(NsState* st, uint32_t mask, int B, hipStream_t stream) {
    hipLaunchKernelGGL(mark_states_done, dim3(1), dim3(64), 0, stream, st, mask, B);
}
void enqueue_clear_states(NsState* st, int64_t B, hipStream_t stream) {
    hipLaunchKernelGGL(clear_states, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, stream, st, B);
}
void enqueue_add_diag(double* M, int d, double eps, hipStream_t stream) {
    hipLaunchKernelGGL(add_diag, dim3((unsigned)cdiv(d, 256)), dim3(256), 0, stream, M, d, eps);
}
void enqueue_finalize_for_frechet(const double* acc1, const double* acc2, int d, int ddof, double* mus, double* covs, NsState* st,
                                  hipStream_t stream, const float* run1, const float* run2) {
    hipLaunchKernelGGL(finalize_for_frechet, dim3((unsigned)cdiv((int64_t)d * d, 256), 2), dim3(256), 0, stream, acc1, acc2, d, ddof, mus, covs, st, run1, run2);
}
void enqueue_ns_prepare(const double* stats_all, int d, int nb, const double* mu1, int64_t m1, const double* mu2, int64_t m2,
                        int mean_dtype, NsState* st_all, int mean_given, Ns32State* s32, int64_t B, hipStream_t stream) {
    hipLaunchKernelGGL(ns_prepare, dim3((unsigned)B), dim3(256), 0, stream, stats_all, d, nb, mu1, m1, mu2, m2, mean_dtype, st_all, mean_given, s32);
}

// (first-use warm-up, common.cpp: warm_code_objects -- loading this translation unit's code object costs ~75 ms at the first launch)
const void* code_object_anchor_frechet_f64() { return reinterpret_cast<const void*>(&ns_first); }
}  // namespace fad
