// Micro-probe: does a workgroup barrier per 8 KiB stage (what the single-tile moments kernel has) cost the LDS-DMA stream
// bandwidth?  Same loop as lds_dma_rate.hip (HBM stream, each workgroup walks its own chunk), two 1 KiB loads per wave and
// stage, ring of DEPTH stages, with and without s_barrier after the wait, with and without some MFMA work per stage.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_barrier scripts/probes/lds_dma_barrier.hip && /tmp/lds_dma_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DEPTH, bool BARRIER, int MFMAS>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ buf, size_t chunk, int stages, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = buf + (size_t)blockIdx.x * chunk + (size_t)wave * 2048 + (size_t)lane * 16;
    f32x16 acc; for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    f16x8 a, b; for (int q = 0; q < 8; ++q) { a[q] = (_Float16)1.0f; b[q] = (_Float16)(lane * 0.01f); }
    auto issue = [&](int s) {
        uint4* dst = smem + ((s % DEPTH) * 8 + wave * 2) * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)s * 8192), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)s * 8192 + 1024), (lptr_t)(dst + 64), 16, 0, 0);
    };
    for (int s = 0; s < DEPTH - 1 && s < stages; ++s) issue(s);
    for (int s = 0; s < stages; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 2)) : "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (s + DEPTH - 1 < stages) issue(s + DEPTH - 1);
        else { issue(s); }                                   // keep the count per stage constant at the tail
#pragma unroll
        for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(smem)[blockIdx.x & 63] + acc[0];
}

template <int DEPTH, bool BARRIER, int MFMAS>
void run(char* buf, size_t bytes, int wgs, float* out) {
    const size_t lds = (size_t)DEPTH * 8192;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<DEPTH, BARRIER, MFMAS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t chunk = bytes / wgs; const int stages = (int)(chunk / 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        stream<DEPTH, BARRIER, MFMAS><<<wgs, 256, lds>>>(buf, chunk, stages, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("wgs=%5d depth=%d barrier=%d mfma/stage/wave=%d: %7.1f us  %5.2f TB/s\n", wgs, DEPTH, (int)BARRIER, MFMAS, best * 1e3,
           (double)wgs * stages * 8192 / (best * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)2359296000;
    char* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 1 << 16); hipMemset(buf, 1, bytes);
    for (int wgs : {512, 2048}) {
        run<8, false, 0>(buf, bytes, wgs, out);
        run<8, true, 0>(buf, bytes, wgs, out);
        run<8, true, 5>(buf, bytes, wgs, out);
        run<4, true, 5>(buf, bytes, wgs, out);
        run<8, false, 5>(buf, bytes, wgs, out);
    }
    return 0;
}
