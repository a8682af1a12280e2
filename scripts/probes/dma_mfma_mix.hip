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

// Specialised waves: a 512-thread workgroup whose waves 0-3 only run MFMAs and whose waves 4-7 only issue LDS-DMA
// loads (one of each kind per SIMD).  If a CU can move data into LDS while its matrix pipes run, this takes
// max(loads, MFMA); if not, their sum.
template <int NG, int NM, int WHO, int MODE, int CK = 0>      // CK 1: the compute waves run VALU FMAs instead of MFMAs; WHO: 1 = MFMA waves only active, 2 = loader waves only, 3 = both
__global__ __launch_bounds__(512) void spec(const char* __restrict__ buf, int iters, int stride, float* out) {  // MODE 1: register-staged loaders
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];      // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float s = 0.f;
    if (wave < 4) {
        if (WHO & 1) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
            f16x8 a, b; for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 1e-3f); b[q] = (_Float16)1.0f; }
            if (CK == 0) {
                for (int it = 0; it < iters; ++it)
#pragma unroll
                    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            } else if (CK == 2) {
                typedef double f64x4 __attribute__((ext_vector_type(4)));
                f64x4 dacc[4];
                for (int i = 0; i < 4; ++i) dacc[i] = (f64x4){0, 0, 0, 0};
                const double da = lane * 1e-3, db = 1.0 + lane * 1e-4;
                for (int it = 0; it < iters; ++it)
#pragma unroll
                    for (int m = 0; m < NM / 4; ++m) dacc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, dacc[m & 3], 0, 0, 0);
                for (int i = 0; i < 4; ++i) s += (float)(dacc[i][0] + dacc[i][3]);
            } else {
                float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
                for (int it = 0; it < iters; ++it)
#pragma unroll
                    for (int m = 0; m < NM * 8; ++m) {       // ~ the same wall time as NM MFMAs: 4 dependent chains of v_fma_f32
                        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f);
                        x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
                    }
                s += x0 + x1 + x2 + x3;
            }
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        }
    } else if (WHO & 2) {
        const int lw = wave - 4;
        const char* base = buf + (size_t)(blockIdx.x % 32) * 65536 + lw * 1024 + lane * 16;
        if (MODE == 0) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    uint4* dst = smem + (((it * NG + g) & 15) * 4 + lw) * 64;
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + ((it * NG + g) & 15) * 4096), (lptr_t)dst, 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NG) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 3) {
            // LDS-DMA again, but the address is a wave-uniform 64-bit base (SGPR pair) + a loop-invariant 32-bit lane
            // offset: global_load_lds_dwordx4 v_off, s[base:base+1] -- 4 instead of 8 address bytes per lane
            const uint32_t voff = (uint32_t)(lane * 16);
            const char* wbase = buf + (size_t)(blockIdx.x % 32) * 65536 + lw * 1024;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    uint4* dst = smem + (((it * NG + g) & 15) * 4 + lw) * 64;
                    const uint64_t sb = (uint64_t)(wbase + ((it * NG + g) & 15) * 4096);
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
                    const uint64_t ub = ((uint64_t)hi << 32) | lo;
                    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)dst);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NG) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 4 || MODE == 5) {
            // plain 16-byte loads into registers, no LDS: MODE 4 = 64-bit VGPR addresses, MODE 5 = SGPR base + 32-bit offset
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 r[NG];
            uint32_t x = 0;
            const uint32_t voff = (uint32_t)(lane * 16);
            const char* wbase = buf + (size_t)(blockIdx.x % 32) * 65536 + lw * 1024;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const uint64_t sb = (uint64_t)(wbase + ((it * NG + g) & 15) * 4096);
                    if (MODE == 5) {
                        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
                        const uint64_t ub = ((uint64_t)hi << 32) | lo;
                        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r[g]) : "v"(voff), "s"(ub) : "memory");
                    } else {
                        const uint64_t va = sb + voff;
                        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[g]) : "v"(va) : "memory");
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int g = 0; g < NG; ++g) x ^= r[g].x + r[g].w;
            }
            s += (float)x;
        } else if (MODE == 2) {
            uint32_t x = 0;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int g = 0; g < NG * 2; ++g) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)((char*)smem + ((it + g) & 63) * 1024 + lane * 8));
                    x ^= (uint32_t)v[0] + (uint32_t)v[3];
                }
            s += (float)x;
        } else {
            // plain loads into registers; the PREVIOUS iteration's registers go to LDS (ds_write_b128)
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 cur[NG], nxt[NG];
            for (int g = 0; g < NG; ++g) cur[g] = (u4){(uint32_t)lane, 0u, 0u, 0u};
            int off = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    nxt[g] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(base + off));   // bypass L1 reuse
                    off = (off + stride) & 0xf000;
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    *reinterpret_cast<u4*>(smem + (((it * NG + g) & 15) * 4 + lw) * 64 + lane) = cur[g];
#pragma unroll
                for (int g = 0; g < NG; ++g) cur[g] = nxt[g];
            }
            for (int g = 0; g < NG; ++g) s += (float)cur[g].x;
        }
    }
    __syncthreads();
    s += reinterpret_cast<float*>(smem)[tid * 37 & 16383];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NG, int NM, int WHO, int MODE = 0, int CK = 0>
void run_spec(const char* buf, float* out) {
    const int iters = 2000, wgs = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spec<NG, NM, WHO, MODE, CK>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    spec<NG, NM, WHO, MODE, CK><<<wgs, 512, 65536>>>(buf, iters, 4096, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    spec<NG, NM, WHO, MODE, CK><<<wgs, 512, 65536>>>(buf, iters, 4096, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("specialised waves (4 %s + 4 %s loader per CU): %s loads=%d mfma=%d : %8.3f ms\n", CK == 2 ? "MFMA-f64" : CK ? "VALU" : "MFMA", MODE == 5 ? "load(saddr)" : MODE == 4 ? "load(vaddr)" : MODE == 3 ? "LDS-DMA(saddr)" : MODE == 2 ? "LDS-read" : MODE ? "register-staged" : "LDS-DMA",
           WHO == 1 ? "MFMA waves only  " : WHO == 2 ? "loader waves only" : "both             ", NG, NM, ms);
}

int main() {
    char* buf; float* out; long long* cyc;
    hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 8);
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
    run_spec<8, 16, 1>(buf, out); run_spec<8, 16, 2>(buf, out); run_spec<8, 16, 3>(buf, out);
    run_spec<4, 16, 2>(buf, out); run_spec<4, 16, 3>(buf, out);
    run_spec<8, 16, 2, 1>(buf, out); run_spec<8, 16, 3, 1>(buf, out);
    run_spec<4, 16, 2, 1>(buf, out); run_spec<4, 16, 3, 1>(buf, out);
    run_spec<8, 16, 1, 0, 1>(buf, out); run_spec<8, 16, 3, 0, 1>(buf, out);      // VALU compute + LDS-DMA
    run_spec<8, 16, 2, 2>(buf, out); run_spec<8, 16, 3, 2>(buf, out);            // MFMA + LDS transpose reads
    run_spec<8, 16, 2, 3>(buf, out); run_spec<8, 16, 3, 3>(buf, out);            // MFMA + LDS-DMA with saddr addressing
    run_spec<8, 16, 1, 4, 2>(buf, out);                                          // fp64 MFMA alone
    run_spec<8, 16, 2, 4, 2>(buf, out); run_spec<8, 16, 3, 4, 2>(buf, out);      // fp64 MFMA + plain loads, 64-bit VGPR addresses
    run_spec<8, 16, 2, 5, 2>(buf, out); run_spec<8, 16, 3, 5, 2>(buf, out);      // fp64 MFMA + plain loads, SGPR base
    return 0;
}
