// Per-problem state of the Newton-Schulz trace-sqrt iteration and its convergence check.
//
// The check is a one-workgroup job (reduce the residual partials of the T GEMM, take trace(Y), decide).
// As a kernel of its own it cost ~5 us per iteration on the critical path of a ~30 us iteration, so the
// update-GEMM launch carries it as one extra workgroup per problem ("checker block", gemm_f64.hip): the
// decision of iteration k is taken WHILE Y_{k+1}, Z_{k+1} are being computed and first matters to the
// launches of iteration k+1.  Replaces the convergence logic hidden inside scipy.linalg.sqrtm
// (fadtk/fad.py:88) -- there is no reference code for it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fad {

constexpr int kMaxIter = 64;

struct NsState {
    double c, tr1, tr2, mean_term;
    double res_last, tr_last;
    double res_min, tr_safe;   // divergence guard: smallest residual so far, trace(Y) of the last iterate within 1.5x of it
    int has_safe, pad_;
    int done;          // no more T GEMMs / residual checks for this problem
    int final_iter, conv, nonfinite;
    int too_few[2];    // set by finalize_for_frechet: set i has fewer than two rows
    int finished;      // tr_last is final (also the host's "all done" test)
    // Update GEMMs of iteration k skip when upd_skip[k & 1] != 0.  Two words because the check of iteration k
    // runs concurrently with the update GEMMs of iteration k: it only ever WRITES the word of iteration k+1, never
    // the one the launch it shares the grid with is reading (the two workgroups of a split-K tile must agree).
    int upd_skip[2];
    double res[kMaxIter];
    double tr[kMaxIter];
    // Scaled steps (round 4): (Y, Z) <- mu_k (Y, Z) before iteration k, folded into T_k = 1.5 mu I - 0.5 mu^3 Z Y.  With every
    // eigenvalue x^2 of Z Y in [l_k^2, 1], mu_k^2 = 3 / (1 + l_k + l_k^2) is the scaling of the cubic x (3 - x^2) / 2 that lifts the
    // lower end fastest (Chen & Chow); l_{k+1} = mu l (3 - mu^2 l^2) / 2.  Small eigenvalues then grow ~6.75x per step instead
    // of 2.25x: a k^-2 product takes 13 iterations instead of 20, k^-4 20 instead of 36 (scripts/ns_emulate_scaled.py).
    // mu[k] = 1 once l_k >= 0.9 (and always for flat spectra): the plain iteration, whose rules close the problem.  Written by
    // ns_prepare, read by ns_first, the T product (gemm_f64.hip) and the check.
    double mu[kMaxIter];
    // lower bound of the eigenvalues x of the CURRENT iterate's sqrt(Z Y) (batched low-precision chain, ns_fast_big.h: the step scale
    // of iteration k + 1 is set by the check of iteration k from this bound and the residual it has just measured)
    double l_cur;
};

// x_min estimate for the scaled steps: invert PR(p) = (sum k^-p)^2 / sum k^-2p, k = 1..d, for the exponent p of a power-law spectrum
// with participation ratio pr = (tr A)^2 / tr(A^2), then x_min = d^(-p/2) / 3 (a third: safety).  Float arithmetic on the hardware's
// exp2 / log2 (the double pow() of a first version took 90 us on one thread).
__device__ __forceinline__ double ns_l0_from_participation(float pr, int d) {
    const float lg = __log2f((float)d);
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
    double l = (double)(exp2f(-0.5f * pf * lg) * (1.0f / 3.0f));
    if (l > 0.5) l = 0.5;
    if (l < 1e-5) l = 1e-5;
    return l;
}
// the bound after a step of GIVEN scale mu (x -> mu x (3 - mu^2 x^2) / 2 is increasing on [0, 1 / mu])
__device__ __forceinline__ double ns_step_scale_with(double mu, double& l) {
    l = mu * l * (3.0 - mu * mu * l * l) / 2.0;
    if (l > 1.0) l = 1.0;
    return mu;
}
// Cap on a step scale from the residual r = ||I - Z Y||_F of the iterate it is applied to: were the spectrum concentrated at
// x^2 = 1 - r / sqrt(d) (the root mean square of 1 - x^2), mu^2 = 1 / x^2 would put it on the peak of the cubic; a larger scale -- from
// a lower bound l that lags far behind a nearly converged iterate whose r is still above 1 -- throws the bulk back (measured: a start
// at a tenth of the x_min estimate ended in a growing residual and the float64 fallback; with the cap it costs one or two iterations).
__device__ __forceinline__ double ns_scale_cap(double res, int d) {
    double rr = res / sqrt((double)d);
    if (!(rr < 0.66)) rr = 0.66;
    return sqrt(1.0 / (1.0 - rr));
}
// One scaled step: with every x in [l, 1], mu^2 = 3 / (1 + l + l^2) (1 once l >= 0.9); l <- mu l (3 - mu^2 l^2) / 2.
__device__ __forceinline__ double ns_step_scale(double& l) {
    if (!(l < 0.9)) { l = l * (3.0 - l * l) / 2.0; return 1.0; }
    const double m = sqrt(3.0 / (1.0 + l + l * l));
    l = m * l * (3.0 - m * m * l * l) / 2.0;
    return m;
}
constexpr int kStateInts = sizeof(NsState) / sizeof(int);

struct NsCheckArgs {
    int k, max_iter;
    NsState* st_all;
    const double* partials_all;      // [problem][pstride] sums of (T - I)^2 per GEMM workgroup
    int nslots, pstride;
    const double* Yall;              // Y_k of problem b at Yall + b * stride
    int64_t stride;
    int d;
    double tol_res, tol_tr;
};

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// N sums at once for a 256-thread block (one barrier pair instead of N): red = 4 * N doubles of LDS
template <int N> __device__ __forceinline__ void block_sum_n(double (&v)[N], double* red) {
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
    for (int q = 0; q < N; ++q) v[q] = (red[q] + red[N + q]) + (red[2 * N + q] + red[3 * N + q]);
}
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// One workgroup (256 threads) per problem b; `red` = 4 doubles of LDS.
__device__ __forceinline__ void ns_check_block(const NsCheckArgs& a, int64_t b, double* red) {
    NsState* st = a.st_all + b;
    const int k = a.k, d = a.d;
    if (st->finished) {                            // closed earlier: keep the following update launches off
        if (threadIdx.x == 0) st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    const double* Y = a.Yall + b * a.stride;
    const int tid = threadIdx.x;
    double t = 0.0;
    for (int i = tid; i < d; i += 256) t += Y[(int64_t)i * d + i];
    const double tr = block_sum(t, red);
    if (st->done) {
        // the previous check predicted convergence after one more update: Y is that final iterate
        if (tid == 0) {
            st->tr[k] = tr; st->res[k] = st->res_last;
            const bool finite = (tr == tr) && !isinf(tr);
            if (!finite) st->nonfinite = 1;
            st->tr_last = tr; st->final_iter = k; st->finished = 1;
            st->upd_skip[(k + 1) & 1] = 1;         // (the word of iteration k was set by the predicting check)
        }
        return;
    }
    const double* partials = a.partials_all + b * a.pstride;
    double s = 0.0;
    for (int i = tid; i < a.nslots; i += 256) s += partials[i];
    const double sumsq = block_sum(s, red);
    if (tid != 0) return;
    // ||I - ZY||_F = 2 ||T - I||_F; with a scaled step T = 1.5 mu I - 0.5 mu^3 ZY and the partials hold (T - (1.5 mu - 0.5 mu^3) I)^2
    const double mu_k = st->mu[k];
    const double res = 2.0 * sqrt(sumsq) / (mu_k * mu_k * mu_k);
    st->res[k] = res; st->tr[k] = tr;
    const bool finite = (res == res) && !isinf(res) && (tr == tr) && !isinf(tr);
    const double tr_prev = st->tr_last;
    const double res_prev = (k > 0) ? st->res[k - 1] : 0.0;
    // Divergence guard.  Eigenvalues of the product at roundoff level (slightly negative ones of a near-singular or
    // rank-deficient C1 C2 with a smoothly decaying spectrum) grow ~2.25x per iteration and blow Y, Z up around
    // iteration 50, long after trace(Y) has converged -- they add ~0 to the trace until then.  Two exits keep the
    // converged trace instead of a NaN:
    //   runaway  trace quiet (two increments <= 1e-9 |tr|) while the residual has GROWN twice in a row;
    //   explode  the residual quadruples (or leaves the floats) after iteration 3: return the trace of the last
    //            iterate whose residual was within 1.5x of the smallest seen -- provided the trace of the previous
    //            iterate still agrees with it to 1e-6 (a product with genuinely negative eigenvalues diverges in
    //            the trace as well: that stays an error, fad.py:102-106).
    if (finite && res <= 1.5 * st->res_min) { st->tr_safe = tr; st->has_safe = 1; }
    if (finite && res < st->res_min) st->res_min = res;
    const bool explode = k >= 4 && st->has_safe && (!finite || res > 4.0 * res_prev) &&
                         fabs(tr_prev - st->tr_safe) <= 1e-6 * fabs(st->tr_safe);
    const bool runaway = finite && k >= 3 && fabs(tr - tr_prev) <= 1e-9 * fabs(tr) &&
                         fabs(tr_prev - st->tr[k - 2]) <= 1e-9 * fabs(tr) && res > res_prev && res_prev > st->res[k - 2];
    if (explode || runaway) {
        st->tr_last = explode ? st->tr_safe : tr_prev;
        st->conv = 2; st->final_iter = k;
        st->done = 1; st->finished = 1; st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    if (finite) { st->res_last = res; st->tr_last = tr; st->final_iter = k; }
    // Stagnation = a rank-deficient product: the null directions keep the residual frozen while the trace has
    // converged.  Both must stand still (the trace alone can pause by coincidence: with c = tr(A^2)/tr(A),
    // tr(Y1) == tr(Y0) exactly), and not before the second iteration.
    const bool stalled = k >= 2 && fabs(tr - tr_prev) <= a.tol_tr * fabs(tr) && fabs(res - res_prev) <= 1e-9 * res;
    int finish = 0;
    if (!finite) { st->nonfinite = 1; finish = 1; }
    else if (res <= a.tol_res) { st->conv = 1; finish = 1; }
    else if (stalled) { st->conv = 2; finish = 1; }
    else if (k + 1 >= a.max_iter) { st->conv = 0; finish = 1; }
    if (finish) {
        // Y_k is the answer.  The update GEMMs of iteration k are running right now and finish undisturbed (their
        // result is simply not used); the next launches are switched off, this word by the next check.
        st->done = 1; st->finished = 1; st->upd_skip[(k + 1) & 1] = 1;
        return;
    }
    // E_{k+1} = (3 E_k^2 + E_k^3) / 4 for E = I - ZY, hence ||E_{k+1}||_F <= 3/4 res^2 + 1/4 res^3: when that
    // bound is already below the tolerance the NEXT iterate is converged -- let this iteration's update finish,
    // skip the next T GEMM and the next update; the next check closes the problem with trace(Y_{k+1}).
    const double bound = 0.75 * res * res + 0.25 * res * res * res;
    if (bound <= a.tol_res) { st->done = 1; st->conv = 1; st->res_last = bound; st->upd_skip[(k + 1) & 1] = 1; }
}

}  // namespace fad
