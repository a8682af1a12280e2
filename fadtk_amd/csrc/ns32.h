// Shared declarations of the low-precision Newton-Schulz leg (gemm_f32.hip, frechet.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "ns_check.h"

namespace fad {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Low-precision iteration state of one problem (lives next to its NsState).
struct Ns32State {
    int done;            // no more T GEMMs
    int finished;        // final iterate known (ok or failed)
    int ok;              // converged to the fp32 floor: (Y, Z)[final_iter & 1] are usable for the correction
    int final_iter;      // index f of the final iterate Y_f
    int failed;          // not finite / too slow: take the fp64 path
    int upd_skip[2];     // as NsState::upd_skip
    int skip_corr;       // 1 until `ok`: keeps the fp64 correction GEMM off
    int decided_at;      // iteration whose check closed the problem (the host sizes the next call's batch by it)
    int strict;          // 1: predict the final iterate only from the fp32 floor (set for a retry, see ns32_finish)
    int grew;            // scaled steps: the previous check saw the residual grow (one bump after an over-scaled step is not a failure)
    double res[16];      // residual of iteration k at slot k & 15 (ns32_slot): the chains of more than 16 iterations (scaled steps on
                         // decaying spectra) keep the last sixteen
};
__host__ __device__ __forceinline__ int ns32_slot(int k) { return k < 0 ? 0 : (k & 15); }

struct Gemm32Args {
    const float* A[2]; const float* B[2]; float* C[2];
    float alpha[2], beta_eye[2], gamma[2];
    double* partials[2];                 // [slots] sums of (C - gamma I)^2 per workgroup, or nullptr
    const int* skip;                     // *skip != 0 -> nothing to do
    int ntypes;                          // blockIdx.z < ntypes: GEMM z; blockIdx.z == ntypes: checker block
    // checker (rides on the update launch, see ns_check.h for the idea)
    int check;                           // 1: blockIdx.z == ntypes runs the convergence check of iteration k
    int k, max_low, nslots;
    double thr_pred;                     // bound on the NEXT residual below which the next iterate is taken as final
    const double* chk_partials;
    Ns32State* st;
    const NsState* st64;                 // problem-level state (done = A was bad / zero; c = the scale)
    const double* A64;                   // gemm_f32_first_launch: the fp64 product A = C1 C2 that iteration 0 starts from
};


// Extras of the two fp64 products of the mixed-precision route whose epilogues also produce statistics (gemm_f64.hip,
// gemm_f64_kernel MODE 1 / 2).  stats: tile statistics in the layout of ns_tilestats (frechet_f64.hip).
struct NsProductExt {
    double* stats;
    // MODE 1 (A = C1 C2): the spare workgroup's mean term
    const double* mu1; const double* mu2; int mean_dtype; NsState* st;
    // MODE 2 (G = Y Y): R = A64 / st->c - G, paired with Z (Z32_alt when *sel is odd)
    const double* A64; const float* Z32; const float* Z32_alt;
};

// A = C1 C2 (fp64, d % 64 == 0, one problem) + tile statistics + mean term -> ext.st->mean_term.  skip as gemm_f64_launch.
int gemm_f64_product_stats_launch(int d, const double* C1, const double* C2, double* A, const int* skip, const NsProductExt& ext,
                                  hipStream_t stream);
// The correction product Y Y on fp32 operands in fp64 with the statistics of R = A/c - Y Y and Z (nothing is stored but those).
int gemm_f64_correction_launch(int d, const float* Y, const float* Y_alt, const int* sel, const int* skip, const NsProductExt& ext,
                               hipStream_t stream);

// launches the GEMM(s) of `g` (+ the checker block when g.check); returns the partial slots per GEMM or < 0
int gemm_f32_launch(int d, const Gemm32Args& g, hipStream_t stream);
// Iteration 0 in one launch: with Y0 = A64 / st64->c and T0 = (3 I - Y0) / 2 formed on the way into LDS,
// C[0] = Y1 = Y0 T0 and C[1] = Z1 = T0 (Z0 = I needs no product).  skip as above.
int gemm_f32_first_launch(int d, const Gemm32Args& g, hipStream_t stream);

}  // namespace fad
