// Pipeline probe: would a 256 x 256 workgroup tile (8 waves, wave tile 128 x 64, ONE workgroup per CU, 4 x 32 KiB LDS ring) feed the
// matrix pipe better than the shipped 128 x 128 tile (4 waves, two workgroups per CU)?  Same skeleton as stream_pipeline.hip
// variant B (SGPR-base LDS-DMA, 32-row stages, transpose reads, no epilogue, no diagonal economy); the wide tile moves half the
// bytes through LDS-DMA per MFMA and issues 16 instead of 8 MFMAs per wave between two barriers.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/t256 scripts/probes/tile256_pipeline.hip && /tmp/t256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef void __attribute__((address_space(3)))* lptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int D = 512, KB = 32, NST = 4;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one LDS-DMA instruction: 4 rows x 128 columns (1 KiB) into a [32 rows][128 cols] sub-slab image (v4's XOR swizzle)
__device__ __forceinline__ void dma_1k(const uint16_t* E, int64_t row0, int col0, int lane, uint32_t lds_byte) {
    const uint64_t sb = (uint64_t)(E + row0 * D + col0);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
    const uint64_t ub = ((uint64_t)hi << 32) | lo;
    const uint32_t voff = (uint32_t)(((lane >> 4) * D + ((lane & 15) ^ ((lane >> 4) << 2)) * 8) * 2);
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
}
__device__ __forceinline__ uint4 frag(const char* slab, int ks, int col0, int lane) {
    const int row = ks * 16 + 8 * (lane >> 5) + ((lane & 15) >> 2), col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int o = row * 256 + (((col >> 3) ^ ((row & 3) << 2)) << 4) + ((col >> 2) & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + o));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + o + 1024));
    uint4 f; __builtin_memcpy(&f.x, &lo, 8); __builtin_memcpy(&f.z, &hi, 8);
    return f;
}
__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, const f32x16& c) {
    f16x8 va, vb; __builtin_memcpy(&va, &a, 16); __builtin_memcpy(&vb, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c, 0, 0, 0);
}

// ---- 128 x 128 tile, 4 waves (the shipped structure); grid = S * 16 (all 16 tiles of the 4 x 4 grid: no triangle here)
__global__ __launch_bounds__(256) void pipe128(const uint16_t* __restrict__ E, int64_t n, int S, int64_t rps, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int STAGE = 2 * KB * 16;
    const int w = blockIdx.x, split = w / 16, tile = w % 16, ca = (tile >> 2) * 128, cb = (tile & 3) * 128;
    const int64_t k0 = (int64_t)split * rps, k1 = (k0 + rps < n) ? k0 + rps : n;
    const int nkb = (int)((k1 - k0) / KB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    auto issue = [&](int kb) {
        const int slot = kb % NST;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row0 = k0 + (int64_t)kb * KB + 16 * h + 4 * wave;
            const int o = slot * STAGE + 256 * h + 64 * wave;
            dma_1k(E, row0, ca, lane, lds0 + o * 16);
            dma_1k(E, row0, cb, lane, lds0 + (o + KB * 16) * 16);
        }
    };
    f32x16 acc[2][2];
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    for (int s = 0; s < NST - 1 && s < nkb; ++s) issue(s);
    for (int kb = 0; kb < nkb; ++kb) {
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        if (ahead >= 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        const char* sA = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        const char* sB = sA + KB * 256;
        uint4 A[2][2], B[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f) { A[ks][f] = frag(sA, ks, 64 * wr + 32 * f, lane); B[ks][f] = frag(sB, ks, 64 * wc + 32 * f, lane); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mma(A[ks][x], B[ks][y], acc[x][y]);
    }
    float s = 0.f;
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) s += acc[x][y][q];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---- 256 x 256 tile, 8 waves as 2 x 4, wave tile 128 x 64; grid = S * 4 (the 2 x 2 grid of 256-tiles)
__global__ __launch_bounds__(512) void pipe256(const uint16_t* __restrict__ E, int64_t n, int S, int64_t rps, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int SUB = KB * 16;                 // uint4 per [32][128] sub-slab
    constexpr int STAGE = 4 * SUB;               // A0 A1 B0 B1
    const int w = blockIdx.x, split = w / 4, tile = w % 4, ca = (tile >> 1) * 256, cb = (tile & 1) * 256;
    const int64_t k0 = (int64_t)split * rps, k1 = (k0 + rps < n) ? k0 + rps : n;
    const int nkb = (int)((k1 - k0) / KB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    auto issue = [&](int kb) {                   // wave w: sub-slab w >> 1, rows 16 (w & 1) + 4 q
        const int slot = kb % NST, sub = wave >> 1;
        const int col0 = (sub < 2 ? ca : cb) + 128 * (sub & 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row0 = k0 + (int64_t)kb * KB + 16 * (wave & 1) + 4 * q;
            const int o = slot * STAGE + sub * SUB + 256 * (wave & 1) + 64 * q;
            dma_1k(E, row0, col0, lane, lds0 + o * 16);
        }
    };
    f32x16 acc[4][2];
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    for (int s = 0; s < NST - 1 && s < nkb; ++s) issue(s);
    for (int kb = 0; kb < nkb; ++kb) {
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        if (ahead >= 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        const char* st = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        const char* sA = st + wr * (SUB * 16);                       // this wave's 128 A-side columns = sub-slab wr
        const char* sB = st + (2 + (wc >> 1)) * (SUB * 16);          // its 64 B-side columns: half of sub-slab 2 + (wc >> 1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 A[4], B[2];
#pragma unroll
            for (int f = 0; f < 4; ++f) A[f] = frag(sA, ks, 32 * f, lane);
#pragma unroll
            for (int g = 0; g < 2; ++g) B[g] = frag(sB, ks, 64 * (wc & 1) + 32 * g, lane);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mma(A[x], B[y], acc[x][y]);
        }
    }
    float s = 0.f;
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) s += acc[x][y][q];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

__global__ void fill(uint16_t* E, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = (uint32_t)(i * 2654435761u) >> 20;
        _Float16 v = (_Float16)(((int)(h & 255) - 128) * (1.0f / 256.0f));
        __builtin_memcpy(&E[i], &v, 2);
    }
}

int main() {
    const int64_t n = 200000;                    // the two sets of config 3 as one stream of rows
    uint16_t* E; float* out;
    hipMalloc(&E, (size_t)n * D * 2 + (1 << 20)); hipMalloc(&out, (size_t)4096 * 512 * 4);
    fill<<<2048, 256>>>(E, (size_t)n * D);
    const size_t lds128 = (size_t)NST * 2 * KB * 16 * 16, lds256 = (size_t)NST * 4 * KB * 16 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe256), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 2.0 * n * D * D;        // the full 512 x 512 product, both variants
    for (int variant = 0; variant < 2; ++variant) {
        for (int S : (variant == 0 ? std::vector<int>{32, 64} : std::vector<int>{64, 128})) {
            const int64_t rps = ((n + S - 1) / S + KB - 1) / KB * KB;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (variant == 0) pipe128<<<S * 16, 256, lds128>>>(E, n, S, rps, out);
                else pipe256<<<S * 4, 512, lds256>>>(E, n, S, rps, out);
                hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
                float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
            }
            std::vector<float> h(1024); hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost);
            printf("%s S=%3d (%4d workgroups, %5ld rows each): %7.1f us = %6.0f TFLOP/s issued (%4.1f %% of the fp16 MFMA peak)  [%g]\n",
                   variant == 0 ? "128 x 128, 4 waves" : "256 x 256, 8 waves", S, variant == 0 ? S * 16 : S * 4, (long)rps, best * 1e3,
                   flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12 / 2500 * 100, (double)h[5]);
        }
    }
    return 0;
}
