// Micro-probe: what read bandwidth does this MI355X deliver to a plain streaming kernel?  (The ceiling the HBM-bound
// D = 128 moments stream should be priced against besides the nominal 8 TB/s.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_read scripts/probes/hbm_read.hip && /tmp/hbm_read
// Every workgroup streams a contiguous chunk (as a row-split of the moments kernel does) or the grid strides through
// the buffer; UNROLL independent 16-byte loads per thread are in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool CHUNKED>
__global__ __launch_bounds__(256) void reader(const u32x4* __restrict__ buf, size_t n16, uint32_t* out) {
    const size_t per_wg = n16 / gridDim.x;
    size_t i = CHUNKED ? (size_t)blockIdx.x * per_wg + threadIdx.x : (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t end = CHUNKED ? (size_t)(blockIdx.x + 1) * per_wg : n16;
    const size_t step = CHUNKED ? 256 : (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i + (UNROLL - 1) * step < end; i += UNROLL * step) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(buf + i + u * step);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;           // never true for the random fill; keeps the loads
}

template <int UNROLL, bool CHUNKED>
void run(const u32x4* buf, size_t bytes, int wgs, uint32_t* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((reader<UNROLL, CHUNKED>), dim3(wgs), dim3(256), 0, 0, buf, bytes / 16, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-8s unroll=%d wgs=%5d: %8.1f us  %5.2f TB/s (%4.1f %% of 8 TB/s)\n", CHUNKED ? "chunked" : "strided", UNROLL, wgs,
           best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12 / 8 * 100);
}

int main() {
    const size_t bytes = (size_t)2359296000;        // 4096 files x 2250 frames x 128 x 2 B: one config-4 group
    u32x4* buf; uint32_t* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 0x5a, bytes);
    for (int wgs : {512, 1024, 2048, 4096, 8192}) {
        run<4, true>(buf, bytes, wgs, out);
        run<8, true>(buf, bytes, wgs, out);
        run<4, false>(buf, bytes, wgs, out);
        run<8, false>(buf, bytes, wgs, out);
    }
    return 0;
}
