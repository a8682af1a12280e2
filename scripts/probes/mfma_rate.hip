// Micro-probe: issue rate / dependent latency of MFMA instructions on gfx950, and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate scripts/probes/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void probe(int iters, double* out, long long* cyc) {
    const long long t0 = __builtin_readcyclecounter();       // s_memtime: shader clock
    if (KIND == 0) {
        f64x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if (KIND == 1) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0;
        f16x8 a, b; for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 1e-3f); b[q] = (_Float16)1.0f; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}

template <int KIND, int NACC>
void run(const char* name, int wgs, int iters, double flop_per_mfma) {
    double* out; long long* cyc; hipMalloc(&out, (size_t)wgs * 256 * 8); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, NACC><<<wgs, 256>>>(iters, out, cyc);               // warm
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<KIND, NACC><<<wgs, 256>>>(iters, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma_wave = (double)iters * NACC;
    const double tf = n_mfma_wave * 4 * wgs * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-28s wgs=%4d acc=%d: %8.3f ms  %7.1f cycles/MFMA(wave clk)  clk=%.2f GHz  %8.1f TFLOP/s\n", name, wgs, NACC, ms,
           (double)c / n_mfma_wave, (double)c / (ms * 1e6), tf);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    run<0, 1>("f64 16x16x4 dependent", 256, it, 2048);
    run<0, 4>("f64 16x16x4 4 chains", 256, it, 2048);
    run<0, 4>("f64 16x16x4 4 chains 2w/SIMD", 512, it, 2048);
    run<0, 4>("f64 16x16x4 4 chains 4w/SIMD", 1024, it, 2048);
    run<1, 1>("f32 16x16x4 dependent", 256, it, 2048);
    run<1, 4>("f32 16x16x4 4 chains", 256, it, 2048);
    run<2, 1>("f16 32x32x16 dependent", 256, it, 32768);
    run<2, 4>("f16 32x32x16 4 chains", 256, it, 32768);
    run<2, 4>("f16 32x32x16 4 chains 2w", 512, it, 32768);
    return 0;
}
