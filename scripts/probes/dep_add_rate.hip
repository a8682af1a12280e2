// Micro-probe: what a LONE wave pays per dependent v_add_f32 on gfx950 -- the floor of np.mean's running-sum walk (one dependent add per row
// and column, moments_kernels.h: moments_running_colsum_h16).  One wave per workgroup, one workgroup per CU; chains of float adds (not re-associated without fast-math; the ISA was checked):   CH independent chains (1, 2, 4: the compiler packs pairs of chains into v_pk_add_f32), with LANES active lanes (64 or 16), optionally with an
// LDS read + counted wait per four adds (the walk's instruction mix).  Prints shader cycles per add INSTRUCTION and per chain step.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dep_add_rate scripts/probes/dep_add_rate.hip && /tmp/dep_add_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH, bool LDS>
__global__ __launch_bounds__(64) void probe(int iters, int lanes, float* out, long long* cyc) {
    __shared__ float4 tile[256];
    for (int i = threadIdx.x; i < 256; i += 64) tile[i] = make_float4(1e-3f * i, 2e-3f, 3e-3f, 4e-3f);
    __syncthreads();
    float s0 = threadIdx.x, s1 = 1.f, s2 = 2.f, s3 = 3.f;
    const float x = 1.0f + 1e-3f * threadIdx.x;
    if ((int)threadIdx.x >= lanes) return;
    const long long t0 = __builtin_readcyclecounter();
    float4 nx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nx[k] = LDS ? tile[(4 * k + threadIdx.x) & 255] : make_float4(x, x, x, x);
    for (int it = 0; it < iters; it += 4) {                                   // 16 steps per trip; the next trip's reads are issued first
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = nx[k];
        if (LDS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) nx[k] = tile[(it + 4 + 4 * k + threadIdx.x) & 255];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = (q == 0) ? v[k].x : (q == 1) ? v[k].y : (q == 2) ? v[k].z : v[k].w;
                s0 = s0 + a;                          // (plain C: float adds are not re-associated; inline asm adds get an s_nop each)
                if (CH > 1) s1 = s1 + a;
                if (CH > 2) { s2 = s2 + a; s3 = s3 + a; }
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CH, bool LDS>
void run(const char* name, int lanes) {
    float* out; long long* cyc; hipMalloc(&out, 256 * 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 200000;
    probe<CH, LDS><<<256, 64>>>(1000, lanes, out, cyc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); probe<CH, LDS><<<256, 64>>>(iters, lanes, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double steps = 4.0 * iters;                    // steps of each chain
    printf("%-44s lanes=%2d: %7.3f ms, clock %.2f GHz: %5.2f cycles per add instruction, %5.2f cycles per step of a chain\n", name, lanes, ms,
           (double)c / (ms * 1e6), (double)c / (steps * CH), (double)c / steps);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int lanes : {64, 16}) {
        run<1, false>("1 chain, registers only", lanes);
        run<2, false>("2 chains (one v_pk_add_f32 per step), registers", lanes);
        run<4, false>("4 chains (two v_pk_add_f32 per step), registers", lanes);
        run<1, true>("1 chain + ds_read_b128 per 4 adds", lanes);
        run<2, true>("2 chains (v_pk_add_f32) + ds_read_b128 / 4 steps", lanes);
    }
    return 0;
}
