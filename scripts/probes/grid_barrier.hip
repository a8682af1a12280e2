// Micro-probe: cost of a hand-written grid-wide barrier on gfx950 (one 512-thread workgroup per CU) and whether data
// written before it by other XCDs is visible after it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier scripts/probes/grid_barrier.hip && /tmp/grid_barrier
// Kernel: NB rounds of { every workgroup writes a 4 KiB tile of a matrix; barrier; every workgroup reads the tiles of
// 16 OTHER workgroups (as one phase of a D = 512 product would) and checks them }.  Reported: us per round with the
// exchange, us per bare barrier, and the same chain as NB separate launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <bool EXCHANGE>
__global__ __launch_bounds__(512) void persistent(float* buf0, float* buf1, unsigned* bar, int rounds, int* bad) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    int errors = 0;
    for (int r = 0; r < rounds; ++r) {
        float* w = (r & 1) ? buf1 : buf0;
        if (EXCHANGE) {
            w[(size_t)b * 1024 + tid] = (float)(r * 1000 + b);
            w[(size_t)b * 1024 + 512 + tid] = (float)(r * 1000 + b);
        }
        grid_barrier(bar, (unsigned)(r + 1) * G);
        if (EXCHANGE) {
            for (int j = 1; j <= 16; ++j) {
                const int o = (b + j * 37) % G;
                const float v = w[(size_t)o * 1024 + tid] + w[(size_t)o * 1024 + 512 + tid];
                if (v != 2.0f * (float)(r * 1000 + o)) ++errors;
            }
        }
    }
    if (errors) atomicAdd(bad, errors);
}

__global__ __launch_bounds__(512) void one_round(float* w, const float* prev, int r, int* bad) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    int errors = 0;
    if (prev) {
        for (int j = 1; j <= 16; ++j) {
            const int o = (b + j * 37) % G;
            const float v = prev[(size_t)o * 1024 + tid] + prev[(size_t)o * 1024 + 512 + tid];
            if (v != 2.0f * (float)((r - 1) * 1000 + o)) ++errors;
        }
    }
    w[(size_t)b * 1024 + tid] = (float)(r * 1000 + b);
    w[(size_t)b * 1024 + 512 + tid] = (float)(r * 1000 + b);
    if (errors) atomicAdd(bad, errors);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int G = p.multiProcessorCount;
    float *b0, *b1; unsigned* bar; int* bad;
    CK(hipMalloc(&b0, (size_t)G * 4096)); CK(hipMalloc(&b1, (size_t)G * 4096));
    CK(hipMalloc(&bar, 4)); CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 16;
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipMemsetAsync(bar, 0, 4, 0));
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(persistent<true>, dim3(G), dim3(512), 0, 0, b0, b1, bar, rounds, bad);
            else hipLaunchKernelGGL(persistent<false>, dim3(G), dim3(512), 0, 0, b0, b1, bar, rounds, bad);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s: %d workgroups, %d rounds: %.2f us total, %.2f us per round\n", mode == 0 ? "persistent + exchange" : "persistent, bare barrier",
               G, rounds, best * 1e3f, best * 1e3f / rounds);
    }
    {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < rounds; ++r)
                hipLaunchKernelGGL(one_round, dim3(G), dim3(512), 0, 0, (r & 1) ? b1 : b0, r ? ((r & 1) ? b0 : b1) : nullptr, r, bad);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("separate launches: %d rounds: %.2f us total, %.2f us per round\n", rounds, best * 1e3f, best * 1e3f / rounds);
    }
    int hbad = -1;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("stale reads: %d\n", hbad);
    return hbad != 0;
}
