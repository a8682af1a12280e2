// Micro-probe: do plain global loads (to registers, written to LDS one iteration later -- the classic register-
// staged GEMM pipeline) overlap with MFMAs on a gfx950 CU, where LDS-DMA loads do not (dma_mfma_mix.hip)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vmo scripts/probes/vmem_mfma_overlap.hip && /tmp/vmo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every wave: NG 16-byte loads per lane per iteration (1 KiB per wave instruction, L2-resident 64 KiB window per
// workgroup, offsets from a runtime stride so nothing can be hoisted), the PREVIOUS iteration's data goes to LDS,
// NM independent-accumulator MFMAs.
template <int NG, int NM>
__global__ __launch_bounds__(256) void mix(const char* __restrict__ buf, int iters, int stride, float* out) {
    __shared__ u32x4 smem[4096];             // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = buf + (size_t)(blockIdx.x % 32) * 65536 + wave * 1024 + lane * 16;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    f16x8 a, b; for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 1e-3f); b[q] = (_Float16)1.0f; }
    u32x4 cur[NG > 0 ? NG : 1], nxt[NG > 0 ? NG : 1];
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) cur[g] = (u32x4){(uint32_t)lane, 0u, 0u, 0u};
    int off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            nxt[g] = *reinterpret_cast<const u32x4*>(base + off);
            off = (off + stride) & 0xf000;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) smem[((it * NG + g) & 15) * 256 + tid] = cur[g];      // last iteration's data
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) cur[g] = nxt[g];
    }
    __syncthreads();
    float s = reinterpret_cast<float*>(smem)[(tid * 37) & 16383];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) s += (float)cur[g].x;
    out[blockIdx.x * 256 + tid] = s;
}

template <int NG, int NM>
void run(const char* buf, float* out, int wgs) {
    const int iters = 2000;
    mix<NG, NM><<<wgs, 256>>>(buf, iters, 4096, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    mix<NG, NM><<<wgs, 256>>>(buf, iters, 4096, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("wgs=%d loads=%d mfma=%d : %8.3f ms  (%.1f GB/s per CU)\n", wgs, NG, NM, ms,
           NG ? (double)wgs * iters * NG * 4096.0 / (ms * 1e-3) / 1e9 / 256 : 0.0);
}

int main() {
    char* buf; float* out;
    hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 1024 * 256 * 4);
    for (int wgs : {256, 512}) {
        run<4, 0>(buf, out, wgs); run<0, 8>(buf, out, wgs); run<4, 8>(buf, out, wgs);
        run<8, 0>(buf, out, wgs); run<0, 16>(buf, out, wgs); run<8, 16>(buf, out, wgs);
    }
    return 0;
}
