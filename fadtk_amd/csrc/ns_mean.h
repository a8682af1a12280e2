// The mean term ||mu1 - mu2||^2 of the Frechet distance as the reference forms it (fad.py:83, 119), one workgroup.
#pragma once
#include "fad_common.h"
#include "ns_check.h"

namespace fad {

// np.mean of a float16 / bfloat16 / float32 matrix is rounded to that dtype; fadtk then forms
// diff = mu1 - mu2 and diff.dot(diff) IN that dtype (fad.py:83, 119): for float16 numpy accumulates the dot product
// sequentially in float32 and rounds the result to float16 -- reproduced bit for bit by one lane.
__device__ __forceinline__ double round_f16(double v) { return (double)(float)(_Float16)(float)v; }
__device__ __forceinline__ double round_bf16(double v) {
    uint32_t u = __float_as_uint((float)v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (double)__uint_as_float(u & 0xffff0000u);
}

// Called by the first 256 threads of a workgroup (all of them, uniformly).  mean_dtype: FAD_F16 / FAD_BF16 / FAD_F32 =
// the reference's mean term for embeddings of that dtype, anything else = float64.  gaps: 1024 floats of LDS, red: 4
// doubles of LDS.  The value is returned to thread 0 (other threads: unspecified).
__device__ __forceinline__ double mean_term_block(const double* __restrict__ mu1, const double* __restrict__ mu2, int d,
                                                  int mean_dtype, float* gaps, double* red) {
    const int tid = threadIdx.x;
    double mt = 0.0;
    if (mean_dtype >= 16) {
        // FAD_MEAN_SECOND_ONLY | dtype: only the SECOND mean carries the dtype's rounding, the arithmetic stays float64 -- what
        // score_inf does (fad.py:333-341): mu_base comes from the statistics cache in float64, mu_eval = np.mean of the resampled
        // float16 frames is float16, and numpy promotes their difference to float64.
        const int dt = mean_dtype & 3;
        for (int i = tid; i < d; i += 256) {
            const double m2 = (dt == FAD_F16) ? round_f16(mu2[i]) : (dt == FAD_BF16) ? round_bf16(mu2[i])
                              : (dt == FAD_F32) ? (double)(float)mu2[i] : mu2[i];
            const double df = mu1[i] - m2;
            mt += df * df;
        }
        return block_sum(mt, red);
    }
    for (int i = tid; i < d; i += 256) { const double df = mu1[i] - mu2[i]; mt += df * df; }
    double mean_term = block_sum(mt, red);             // NaNs/Infs propagate through the sum
    if (mean_dtype == FAD_F16 || mean_dtype == FAD_BF16) {
        // the gaps are formed by all threads (through LDS, 1024 at a time); lane 0 only runs the ordered float32 sum
        float acc = 0.f;
        for (int i0 = 0; i0 < d; i0 += 1024) {
            __syncthreads();
            for (int i = i0 + tid; i < d && i < i0 + 1024; i += 256) {
                const double a1 = (mean_dtype == FAD_F16) ? round_f16(mu1[i]) : round_bf16(mu1[i]);
                const double a2 = (mean_dtype == FAD_F16) ? round_f16(mu2[i]) : round_bf16(mu2[i]);
                gaps[i - i0] = (float)((mean_dtype == FAD_F16) ? round_f16(a1 - a2) : round_bf16(a1 - a2));
            }
            __syncthreads();
            if (tid == 0) {
                const int m = (d - i0 < 1024) ? d - i0 : 1024;
                int i = 0;
                for (; i + 16 <= m; i += 16) {       // 16 LDS reads in flight, then the ordered chain of 16 fmas
                    float gq[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) gq[q] = gaps[i + q];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc = __fmaf_rn(gq[q], gq[q], acc);   // product of two halfs is exact in float
                }
                for (; i < m; ++i) acc = __fmaf_rn(gaps[i], gaps[i], acc);
            }
        }
        mean_term = (mean_dtype == FAD_F16) ? round_f16((double)acc) : round_bf16((double)acc);
    }
    if (tid == 0 && mean_dtype == FAD_F32) {
        double acc = 0.0;
        for (int i = 0; i < d; ++i) { const double g = (double)(float)((double)(float)mu1[i] - (double)(float)mu2[i]); acc += g * g; }
        mean_term = (double)(float)acc;
    }
    return mean_term;
}

}  // namespace fad
