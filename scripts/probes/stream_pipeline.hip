// Pipeline-structure probe for the moments tile kernel: the skeleton of v4 (config 3: N = 100000 rows of 512 fp16, 51
// row-splits x 10 tiles, 128 x 128 tile, 32-row stages through a 4-slot LDS ring, 16 transpose reads + 8 MFMAs per wave
// and stage) with three ways of moving the data:
//   A  every wave loads AND computes, one workgroup barrier per stage, 64-bit VGPR addresses          (= v4)
//   B  like A with SGPR-base + 32-bit-offset addressing
//   C  two extra LOADER waves per workgroup roll ahead through LDS flags (no barrier), SGPR-base addressing; the four
//      compute waves only read LDS and run MFMAs
// Prints the kernel time of each and a checksum of what the MFMAs accumulated (must agree across variants).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sp scripts/probes/stream_pipeline.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int D = 512, BT = 128, KB = 32, NST = 4, NT = D / BT, T = NT * (NT + 1) / 2;
constexpr int STAGE = 2 * KB * 16;            // uint4 per ring slot (A slab + B slab)
constexpr int SPIN_LIMIT = 1 << 22;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {
    const int xcd = b % 8, idx = b / 8, q = nwg / 8, r = nwg % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void tile_coords(int tile, int& ta, int& tb) {
    int a = 0, t = tile;
    while (t >= NT - a) { t -= NT - a; ++a; }
    ta = a; tb = a + t;
}

// one LDS-DMA instruction: rows row0 + (lane >> 4) [+ 4 * part], 256 B of columns starting at col0
template <bool SADDR>
__device__ __forceinline__ void dma_1k(const uint16_t* E, int64_t row0, int col0, int lane, uint32_t lds_byte, uint4* lds_ptr) {
    if (SADDR) {
        const uint64_t sb = (uint64_t)(E + row0 * D + col0);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
        const uint64_t ub = ((uint64_t)hi << 32) | lo;
        const uint32_t voff = (uint32_t)(((lane >> 4) * D + ((lane & 15) ^ ((lane >> 4) << 2)) * 8) * 2);     // v4's XOR swizzle
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
    } else {
        const uint16_t* src = E + (row0 + (lane >> 4)) * D + col0 + ((lane & 15) ^ ((lane >> 4) << 2)) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_ptr, 16, 0, 0);
    }
}

// the compute of one stage for one of the four compute waves: 16 transpose reads, 8 MFMAs
__device__ __forceinline__ void compute_stage(const char* sA, const char* sB, int lane, int wr, int wc, f32x16 (&acc)[2][2]) {
    uint4 F[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const char* slab = (f < 2) ? sA : sB;
            const int col0 = ((f < 2) ? 64 * wr : 64 * wc) + 32 * (f & 1);
            const int row = ks * 16 + 8 * (lane >> 5) + ((lane & 15) >> 2), col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const int o = row * 256 + (((col >> 3) ^ ((row & 3) << 2)) << 4) + ((col >> 2) & 1) * 8;          // conflict-free (v4)
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + o));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(slab + o + 1024));
            __builtin_memcpy(&F[ks][f].x, &lo, 8); __builtin_memcpy(&F[ks][f].z, &hi, 8);
        }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                f16x8 a, b; __builtin_memcpy(&a, &F[ks][x], 16); __builtin_memcpy(&b, &F[ks][2 + y], 16);
                acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[x][y], 0, 0, 0);
            }
}

// ---------------------------------------------------------------------------- variants A / B
template <bool SADDR>
__global__ __launch_bounds__(256) void pipe_ab(const uint16_t* __restrict__ E, int64_t n, int S, int64_t rows_per_split, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int w = xcd_contiguous(blockIdx.x, S * T), split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, ta, tb);
    const bool diag = ta == tb;
    const int ca = ta * BT, cb = tb * BT;
    const int64_t k0 = (int64_t)split * rows_per_split, k1 = (k0 + rows_per_split < n) ? k0 + rows_per_split : n;
    const int nkb = (int)((k1 - k0) / KB);       // whole stages only (probe)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    auto issue = [&](int kb) {
        const int slot = kb % NST;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row0 = k0 + (int64_t)kb * KB + 16 * h + 4 * wave;
            const int o = slot * STAGE + 256 * h + 64 * wave;
            dma_1k<SADDR>(E, row0, ca, lane, lds0 + o * 16, smem + o);
            if (!diag) dma_1k<SADDR>(E, row0, cb, lane, lds0 + (o + KB * 16) * 16, smem + o + KB * 16);
        }
    };
    f32x16 acc[2][2];
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    for (int s = 0; s < NST - 1 && s < nkb; ++s) issue(s);
    for (int kb = 0; kb < nkb; ++kb) {
        const int ahead = (nkb - 1 - kb < NST - 2) ? (nkb - 1 - kb) : (NST - 2);
        if (diag) { if (ahead >= 2) wait_vmcnt<4>(); else if (ahead == 1) wait_vmcnt<2>(); else wait_vmcnt<0>(); }
        else { if (ahead >= 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
        __builtin_amdgcn_s_barrier();
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        const char* sA = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        compute_stage(sA, diag ? sA : sA + KB * 256, lane, wr, wc, acc);
    }
    float s = 0.f;
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) s += acc[x][y][q];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// ---------------------------------------------------------------------------- variant C
// waves 0-3 compute, waves 4-5 load.  LDS words: landed = loader-wave stage completions (2 per stage), freed = compute-
// wave stage completions (4 per stage), err = bail-out flag.
__global__ __launch_bounds__(384) void pipe_c(const uint16_t* __restrict__ E, int64_t n, int S, int64_t rows_per_split, float* out,
                                              int* fail) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(smem + NST * STAGE);       // [0] landed, [1] freed, [2] err
    const int w = xcd_contiguous(blockIdx.x, S * T), split = w / T, tile = w - split * T;
    int ta, tb; tile_coords(tile, ta, tb);
    const bool diag = ta == tb;
    const int ca = ta * BT, cb = tb * BT;
    const int64_t k0 = (int64_t)split * rows_per_split, k1 = (k0 + rows_per_split < n) ? k0 + rows_per_split : n;
    const int nkb = (int)((k1 - k0) / KB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    if (tid < 4) flags[tid] = 0;
    __syncthreads();

    auto spin_until = [&](int which, uint32_t need) -> bool {       // false = gave up (sets err so everybody leaves)
        for (int i = 0; i < SPIN_LIMIT; ++i) {
            if (flags[which] >= need) return true;
            if (flags[2]) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        flags[2] = 1;
        return false;
    };

    if (wave >= 4) {                                   // ---- loader waves: half of every stage each
        const int lw = wave - 4;
        // loader lw takes the 16-row half h = lw of the stage: 4 instructions (rows 4q..) per slab
        auto issue_half = [&](int kb) {
            const int slot = kb % NST;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row0 = k0 + (int64_t)kb * KB + 16 * lw + 4 * q;
                const int o = slot * STAGE + 256 * lw + 64 * q;
                dma_1k<true>(E, row0, ca, lane, lds0 + o * 16, smem + o);
                if (!diag) dma_1k<true>(E, row0, cb, lane, lds0 + (o + KB * 16) * 16, smem + o + KB * 16);
            }
        };
        constexpr int K = 2;                           // stages this wave keeps in flight
        for (int kb = 0; kb < nkb; ++kb) {
            if (kb >= NST && !spin_until(1, 4u * (uint32_t)(kb - NST + 1))) break;      // the slot's previous stage was consumed
            issue_half(kb);
            if (kb >= K) {                              // stage kb-K of this wave has landed
                if (diag) wait_vmcnt<4 * K>(); else wait_vmcnt<8 * K>();
                if (lane == 0) atomicAdd(const_cast<uint32_t*>(&flags[0]), 1u);
            }
        }
        // drain: the last K stages
        if (diag) wait_vmcnt<4>(); else wait_vmcnt<8>();
        if (nkb >= 2 && lane == 0) atomicAdd(const_cast<uint32_t*>(&flags[0]), 1u);
        wait_vmcnt<0>();
        if (nkb >= 1 && lane == 0) atomicAdd(const_cast<uint32_t*>(&flags[0]), 1u);
        return;
    }
    // ---- compute waves
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.f;
    bool ok = true;
    for (int kb = 0; kb < nkb && ok; ++kb) {
        ok = spin_until(0, 2u * (uint32_t)(kb + 1));
        if (!ok) break;
        const char* sA = reinterpret_cast<const char*>(smem + (kb % NST) * STAGE);
        compute_stage(sA, diag ? sA : sA + KB * 256, lane, wr, wc, acc);
        // the MFMAs above were issued after their operands landed (lgkmcnt(0)), so the slot can be given back
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) atomicAdd(const_cast<uint32_t*>(&flags[1]), 1u);
    }
    if (!ok && lane == 0) atomicAdd(fail, 1);
    float s = 0.f;
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int q = 0; q < 16; ++q) s += acc[x][y][q];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

__global__ void fill(uint16_t* E, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = (uint32_t)(i * 2654435761u) >> 20;
        _Float16 v = (_Float16)(((int)(h & 255) - 128) * (1.0f / 256.0f));
        __builtin_memcpy(&E[i], &v, 2);
    }
}

int main() {
    const int64_t n = 100000;
    const int S = 51;
    const int64_t rps = ((n + S - 1) / S + KB - 1) / KB * KB;
    uint16_t* E; float* out; int* fail;
    hipMalloc(&E, (size_t)n * D * 2 + (1 << 20)); hipMalloc(&out, (size_t)S * T * 256 * 4); hipMalloc(&fail, 4);
    fill<<<2048, 256>>>(E, (size_t)n * D);
    hipMemset(fail, 0, 4);
    const size_t lds_ab = (size_t)NST * STAGE * 16, lds_c = lds_ab + 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe_ab<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ab);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe_ab<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ab);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe_c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
    std::vector<float> h((size_t)S * T * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 3; ++variant) {
        float best = 1e9f; double sum = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(out, 0, h.size() * 4);
            hipEventRecord(e0);
            if (variant == 0) pipe_ab<false><<<S * T, 256, lds_ab>>>(E, n, S, rps, out);
            else if (variant == 1) pipe_ab<true><<<S * T, 256, lds_ab>>>(E, n, S, rps, out);
            else pipe_c<<<S * T, 384, lds_c>>>(E, n, S, rps, out, fail);
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("variant %d failed: %s\n", variant, hipGetErrorString(hipGetLastError())); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        for (float v : h) sum += v;
        int f = 0; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("variant %c: %7.1f us   checksum %.6e   bail-outs %d\n", "ABC"[variant], best * 1e3, sum, f);
    }
    return 0;
}
