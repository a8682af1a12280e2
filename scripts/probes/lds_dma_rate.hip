// Micro-probe: how fast can gfx950 move bytes L2/MALL/HBM -> LDS with global_load_lds (16 B per lane),
// as a function of the bytes kept in flight per CU?  Answers "is the moments tile kernel bound by a
// bandwidth or by latency x bytes-in-flight".
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_rate scripts/probes/lds_dma_rate.hip && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Every wave instruction moves 1 KiB: either 1 KiB contiguous (SEG = 1024) or four 256-byte row segments one
// `ld` apart (SEG = 256, the access shape of a [rows x 512] fp16 panel of 128 columns).
template <int DEPTH, int SEG>
__global__ __launch_bounds__(256) void stream_lds(const char* __restrict__ buf, size_t window, size_t wg_stride, size_t span,
                                                  size_t ld, int iters, float* out, int group) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // group > 1: `group` consecutive workgroups OF ONE XCD (block b runs on XCD b % 8) share a window.
    const int per_xcd = gridDim.x / 8;
    const int w = (group > 1) ? ((int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8)) / group : (int)blockIdx.x;
    const char* base = buf + ((size_t)w * wg_stride) % span;
    size_t lane_off;
    if (SEG == 1024) lane_off = (size_t)lane * 16;
    else lane_off = (size_t)(lane >> 4) * ld + (size_t)(lane & 15) * 16;
    const size_t step = (SEG == 1024) ? 4096 : 16 * ld;            // bytes of address space one WG instruction-group covers
    const size_t wave_off = (SEG == 1024) ? (size_t)wave * 1024 : (size_t)wave * 4 * ld;
    size_t off = 0;
    for (int it = 0; it < iters; ++it) {
        uint4* dst = smem + ((it % DEPTH) * 4 + wave) * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off + wave_off + lane_off), (lptr_t)dst, 16, 0, 0);
        off += step; if (off >= window) off = 0;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(smem)[blockIdx.x & 63];
}

template <int DEPTH, int SEG>
void run(const char* what, char* buf, int wgs, size_t window, size_t wg_stride, size_t span, int iters, float* out, int group = 1) {
    const size_t lds = (size_t)DEPTH * 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_lds<DEPTH, SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream_lds<DEPTH, SEG><<<wgs, 256, lds>>>(buf, window, wg_stride, span, 1024, iters, out, group);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    stream_lds<DEPTH, SEG><<<wgs, 256, lds>>>(buf, window, wg_stride, span, 1024, iters, out, group);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * iters * 4096.0;
    printf("%-8s seg=%4d wgs=%5d depth=%2d (%5.1f KiB in flight/WG): %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU\n", what, SEG, wgs,
           DEPTH, DEPTH * 4.0, ms, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256);
}

template <int SEG>
void sweep(const char* what, char* buf, size_t window, size_t wg_stride, size_t span, int iters, float* out, int group = 1) {
    for (int wgs : {256, 512, 1024, 2048}) {
        run<2, SEG>(what, buf, wgs, window, wg_stride, span, iters, out, group);
        run<4, SEG>(what, buf, wgs, window, wg_stride, span, iters, out, group);
        run<8, SEG>(what, buf, wgs, window, wg_stride, span, iters, out, group);
        if (wgs <= 1024) run<16, SEG>(what, buf, wgs, window, wg_stride, span, iters, out, group);
    }
}

int main() {
    const size_t total = (size_t)4 << 30;
    char* buf; float* out;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, total); hipMalloc(&out, 1 << 16);
    // L2-hot: 32 windows of 64 KiB (2 MiB in all), every workgroup cycles inside its own window.
    sweep<1024>("L2-hot", buf, 64 << 10, 64 << 10, 2 << 20, 4096, out);
    sweep<256>("L2-hot", buf, 64 << 10, 64 << 10, 2 << 20, 4096, out);
    // MALL-resident: 128 MiB span, 512 KiB windows.
    sweep<1024>("MALL", buf, 512 << 10, 512 << 10, 128 << 20, 2048, out);
    // HBM stream: every workgroup walks its own 1 MiB once (<= 2 GiB touched).
    sweep<1024>("HBM", buf, 1 << 20, 1 << 20, total / 2, 256, out);
    sweep<256>("HBM", buf, 1 << 20, 1 << 20, total / 2, 256, out);
    // Shared stream: groups of 10 workgroups read the SAME window (the moments kernel: 10 tiles per split).
    printf("-- shared stream, 10 workgroups per window --\n");
    sweep<256>("HBMx10", buf, 1 << 20, 1 << 20, total / 2, 256, out, 10);
    return 0;
}
