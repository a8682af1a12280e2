// Micro-probe: do global_load_lds (16 B/lane LDS-DMA), ds_read_b64_tr_b16 and MFMA overlap inside a CU?
// Every wave loops over: NG LDS-DMA loads (L2-hot window) + NL LDS transpose reads + NM MFMAs, two
// workgroups of 4 waves per CU (the moments kernel's occupancy).  Compare the mixes with the single-op runs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix scripts/probes/dma_mfma_mix.hip && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NG, int NL, int NM, int REGSTAGE>
__global__ __launch_bounds__(256) void mix(const char* __restrict__ buf, int iters, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];      // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = buf + (size_t)(blockIdx.x % 32) * 65536 + wave * 1024 + lane * 16;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    f16x8 a, b; for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 1e-3f); b[q] = (_Float16)1.0f; }
    uint32_t x = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (REGSTAGE == 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint4* dst = smem + (((it * NG + g) & 15) * 4 + wave) * 64;
                __builtin_amdgcn_global_load_lds((gptr_t)(base + ((it * NG + g) & 15) * 4096), (lptr_t)dst, 16, 0, 0);
            }
        } else {
            uint4 r[NG > 0 ? NG : 1];
#pragma unroll
            for (int g = 0; g < NG; ++g) r[g] = *reinterpret_cast<const uint4*>(base + ((it * NG + g) & 15) * 4096);
#pragma unroll
            for (int g = 0; g < NG; ++g) smem[(((it * NG + g) & 15) * 4 + wave) * 64 + lane] = r[g];
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)((char*)smem + ((it + l) & 63) * 1024 + lane * 8));
            x ^= (uint32_t)v[0] + (uint32_t)v[3];
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
        if (NG > 0 && REGSTAGE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NG) : "memory");
    }
    const long long t1 = clock64();
    __syncthreads();
    float s = (float)x + reinterpret_cast<float*>(smem)[tid * 37 & 16383];      // keeps the LDS writes (and their loads) alive
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NG, int NL, int NM, int REGSTAGE>
void run(const char* buf, float* out, long long* cyc, int wgs = 512) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mix<NG, NL, NM, REGSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    mix<NG, NL, NM, REGSTAGE><<<wgs, 256, 65536>>>(buf, iters, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    mix<NG, NL, NM, REGSTAGE><<<wgs, 256, 65536>>>(buf, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("wgs=%d %s glds=%d ldsrd=%d mfma=%d : %8.3f ms  %7.1f cycles/iter (wave clock)  %.2f GHz\n", wgs, REGSTAGE ? "reg-staged" : "LDS-DMA   ", NG,
           NL, NM, ms, (double)c / iters, (double)c / (ms * 1e6));
}

int main() {
    char* buf; float* out; long long* cyc;
    hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
    run<4, 0, 0, 0>(buf, out, cyc);
    run<0, 16, 0, 0>(buf, out, cyc);
    run<0, 0, 8, 0>(buf, out, cyc);
    run<4, 16, 0, 0>(buf, out, cyc);
    run<4, 0, 8, 0>(buf, out, cyc);
    run<0, 16, 8, 0>(buf, out, cyc);
    run<4, 16, 8, 0>(buf, out, cyc);
    run<2, 16, 8, 0>(buf, out, cyc);
    run<4, 0, 0, 1>(buf, out, cyc);
    run<4, 0, 8, 1>(buf, out, cyc);
    run<4, 16, 8, 1>(buf, out, cyc);
    run<2, 16, 8, 1>(buf, out, cyc);
    for (int wgs : {512, 256}) {
        run<0, 0, 16, 0>(buf, out, cyc, wgs);
        run<8, 0, 16, 0>(buf, out, cyc, wgs);
        run<8, 0, 0, 0>(buf, out, cyc, wgs);
        run<8, 16, 16, 0>(buf, out, cyc, wgs);
    }
    return 0;
}
