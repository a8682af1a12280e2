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
    int pad;
    double res[16];
};

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
    const NsState* st64;                 // problem-level state (done = A was bad / zero)
};


int gemm_f64_from_f32_launch(int d, const float* A, const float* B, const float* A_alt, const float* B_alt, const int* sel,
                             double* C, double alpha, const int* skip, hipStream_t stream);

// launches the GEMM(s) of `g` (+ the checker block when g.check); returns the partial slots per GEMM or < 0
int gemm_f32_launch(int d, const Gemm32Args& g, hipStream_t stream);

}  // namespace fad
