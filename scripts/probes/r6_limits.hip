// Round-6 limit probes for the moments slab kernel (gfx950).  Two questions, each answered by a kernel that does ONE thing:
//   A. what does the chip sustain on dense v_mfma_f32_32x32x16_f16 with Gaussian operands (power!), one or two waves per SIMD,
//      a 4 x 4 block of accumulators per wave, no LDS, no loads -- in long runs and in 200-us bursts;
//   B. what does the LDS-DMA path sustain on the kernel's own stream shape ([rows x 512] float16, 32-row stages of 32 KiB through a
//      ring of four, one workgroup per CU): every workgroup its own rows / two workgroups of one XCD the same rows (the P + Q
//      items), pieces of 4 rows x 256 B or 1 row x 1 KiB, default or nt cache policy, 4 / 2 / 1 issuing waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/r6_limits scripts/probes/r6_limits.hip && /tmp/r6_limits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void fill(uint16_t* E, size_t n, uint32_t seed, int zero) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.0f / 256.0f) - 1.9921875f;
        _Float16 v = zero ? (_Float16)0.f : (_Float16)(u * 1.7320508f);
        __builtin_memcpy(&E[i], &v, 2);
    }
}

// ---- A
template <int NA, int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_pure(const uint4* __restrict__ src, int iters, float* out, long long* cyc) {
    const long long t0 = __builtin_readcyclecounter();
    f16x8 F[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint4 v = src[((size_t)blockIdx.x * 8 + i) * blockDim.x + threadIdx.x]; __builtin_memcpy(&F[i], &v, 16); }
    f32x16 acc[NA];
#pragma unroll
    for (int b = 0; b < NA; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < NA; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[(b >> 2) & 3], F[4 + (b & 3)], acc[b], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NA; ++b) s += acc[b][0] + acc[b][7] + acc[b][15];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}

template <int NA, int threads>
void run_mfma(const char* what, const uint4* src, int iters, int bursts, float* out, long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_pure<NA, threads><<<256, threads>>>(src, iters, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int b = 0; b < bursts; ++b) mfma_pure<NA, threads><<<256, threads>>>(src, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NA * (threads / 64) * 256 * bursts;
    printf("A %-10s %d waves/SIMD acc=%2d iters=%6d x %3d launches: %8.3f ms  %7.1f TFLOP/s  (one launch %.1f us, %.1f clk/MFMA per SIMD, clk %.2f GHz)\n", what,
           threads / 256, NA, iters, bursts, ms, n * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e3 / bursts, (double)c / ((double)iters * NA * (threads / 256)),
           (double)c / (ms / bursts * 1e6));
}

// ---- B
struct DmaArgs { const uint16_t* E; int64_t rows_per_wg, ld; int share, shape, nissue, stages; };
template <int NT_POLICY>
__global__ __launch_bounds__(256) void dma_stream(DmaArgs a, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block b runs on XCD b % 8: consecutive w on one XCD
    const int per = gridDim.x / 8, w = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int region = a.share ? w / a.share : w;
    const uint16_t* base = a.E + (int64_t)region * a.rows_per_wg * a.ld;
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    // shape 0: piece = 4 rows x 256 B (sub-slab of 128 columns), shape 1: piece = 1 row x 1 KiB
    const int lrow = lane >> 4, lchunk = lane & 15;
    const uint32_t voff = a.shape == 0 ? (uint32_t)(((int64_t)lrow * a.ld + lchunk * 8) * 2) : (uint32_t)(lane * 16);
    const int nst = a.stages;
    const int ppw = 32 / a.nissue;                // pieces per issuing wave and stage
    auto issue = [&](int kb) {
        if (wave >= a.nissue) return;
        for (int p = 0; p < ppw; ++p) {
            const int idx = wave * ppw + p;       // 0..31
            const uint16_t* src = a.shape == 0 ? base + ((int64_t)kb * 32 + 4 * (idx & 7)) * a.ld + 128 * (idx >> 3) : base + ((int64_t)kb * 32 + idx) * a.ld;
            const uint64_t sb = (uint64_t)src;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(smem_lds + (uint32_t)(((kb % nst) * 32 + idx) * 1024));
            if (NT_POLICY) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    const int nkb = (int)(a.rows_per_wg / 32);
    for (int s = 0; s < nst - 1; ++s) issue(s);
    for (int kb = 0; kb < nkb; ++kb) {
        // stage kb landed: at most (nst - 2) younger stages of this wave outstanding
        if (wave < a.nissue) {
            const int young = (nkb - 1 - kb < nst - 2 ? nkb - 1 - kb : nst - 2) * ppw;
            if (young >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            else if (young >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (young >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (young >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (young >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kb + nst - 1 < nkb) issue(kb + nst - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(smem)[blockIdx.x & 63];
}

void run_dma(const char* what, const uint16_t* E, int64_t total_rows, int share, int shape, int nissue, int stages, int nt, float* out) {
    DmaArgs a; a.E = E; a.ld = 512; a.share = share; a.shape = shape; a.nissue = nissue; a.stages = stages;
    const int regions = share ? 256 / share : 256;
    a.rows_per_wg = total_rows / regions / 32 * 32;
    const size_t lds = (size_t)stages * 32768;
    auto k = nt ? dma_stream<1> : dma_stream<0>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 256, lds>>>(a, out); hipDeviceSynchronize();
    float best = 1e9, sum = 0; const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); k<<<256, 256, lds>>>(a, out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    const double uniq = (double)regions * a.rows_per_wg * 1024.0, moved = 256.0 * a.rows_per_wg * 1024.0;
    printf("B %-26s share=%d shape=%d issuers=%d ring=%d nt=%d: avg %7.1f us best %7.1f  unique %5.2f TB/s  into LDS %5.2f TB/s (%5.1f GB/s per CU)\n", what, share, shape,
           nissue, stages, nt, sum / reps * 1e3, best * 1e3, uniq / (sum / reps * 1e-3) / 1e12, moved / (sum / reps * 1e-3) / 1e12, moved / (sum / reps * 1e-3) / 1e9 / 256);
}


// ---- C: the tile kernel's own refill pattern without anything else: 8 waves, piece p of wave w = row 4 w + p of the 32-row stage
// (1 KiB), ring of four, ONE barrier per stage, pieces 0, 1 issued before it and 2, 3 behind it (split = 1) or all four behind it.
struct DmaC { const uint16_t* E; int64_t rows_per_wg, ld; int share, swz, split, buf, lag; };
__global__ __launch_bounds__(512) void dma_c(DmaC a, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = gridDim.x / 8, w = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int region = a.share ? w / a.share : w;
    const uint16_t* base = a.E + (int64_t)region * a.rows_per_wg * a.ld;
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    typedef int srd_t __attribute__((ext_vector_type(4)));
    srd_t srd;
    const uint64_t b = (uint64_t)base;
    srd[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b); srd[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
    srd[2] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a.rows_per_wg * a.ld * 2)); srd[3] = 0x00020000;
    uint32_t voff[4];
    for (int p = 0; p < 4; ++p) voff[p] = (uint32_t)(((4 * wave + p) * a.ld) * 2 + ((a.swz ? (lane ^ (p << 2)) : lane) * 16));
    const uint32_t stage_bytes = (uint32_t)(32 * a.ld * 2);
    auto piece = [&](int kb, int p) {
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(smem_lds + (uint32_t)(((kb & 3) * 32 + 4 * wave + p) * 1024));
        if (a.buf) {
            const uint32_t vo = voff[p] + (uint32_t)kb * stage_bytes;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(srd), "s"(m0v) : "memory");
        } else {
            const uint64_t sb = (uint64_t)base + (uint64_t)kb * stage_bytes;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[p]), "s"(ub), "s"(m0v) : "memory");
        }
    };
    const int nkb = (int)(a.rows_per_wg / 32);
    if (a.lag && (w & 1)) __builtin_amdgcn_s_sleep(127);       // the second partner of a pair starts late
    for (int s = 0; s < 3; ++s) for (int p = 0; p < 4; ++p) piece(s, p);
    for (int kb = 0; kb < nkb; ++kb) {
        const bool more = kb + 3 < nkb;
        if (a.split && more) { piece(kb + 3, 0); piece(kb + 3, 1); }
        // stage kb + 1 landed (stage kb at kb = 0 as well: conservative)
        if (more) { if (a.split) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) { if (!a.split) { piece(kb + 3, 0); piece(kb + 3, 1); } piece(kb + 3, 2); piece(kb + 3, 3); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(smem)[blockIdx.x & 63];
}
void run_c(const char* what, const uint16_t* E, int64_t total_rows, int share, int swz, int split, int buf, int lag, float* out) {
    DmaC a; a.E = E; a.ld = 512; a.share = share; a.swz = swz; a.split = split; a.buf = buf; a.lag = lag;
    const int regions = share ? 256 / share : 256;
    a.rows_per_wg = total_rows / regions / 32 * 32;
    const size_t lds = 4 * 32768;
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dma_c<<<256, 512, lds>>>(a, out); hipDeviceSynchronize();
    float best = 1e9, sum = 0; const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); dma_c<<<256, 512, lds>>>(a, out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    const double uniq = (double)regions * a.rows_per_wg * 1024.0;
    printf("C %-30s share=%d swz=%d split=%d buf=%d lag=%d: avg %7.1f us best %7.1f  unique %5.2f TB/s\n", what, share, swz, split, buf, lag,
           sum / reps * 1e3, best * 1e3, uniq / (sum / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : "AB";
    float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    if (strchr(only, 'A')) {
        uint16_t* src; const size_t n = (size_t)256 * 8 * 512 * 8;
        hipMalloc(&src, n * 2);
        for (int zero = 0; zero < 2; ++zero) {
            fill<<<1024, 256>>>(src, n, 17u, zero);
            const char* w = zero ? "zeros" : "gaussian";
            run_mfma<16, 256>(w, (const uint4*)src, 400, 1, out, cyc);          // ~ one 8-set launch's MFMAs per SIMD: 195 stages x 34
            run_mfma<16, 256>(w, (const uint4*)src, 400, 40, out, cyc);
            run_mfma<16, 256>(w, (const uint4*)src, 40000, 1, out, cyc);
            run_mfma<12, 512>(w, (const uint4*)src, 266, 40, out, cyc);
            run_mfma<12, 512>(w, (const uint4*)src, 26600, 1, out, cyc);
            run_mfma<8, 512>(w, (const uint4*)src, 40000, 1, out, cyc);
        }
    }
    if (strchr(only, 'B')) {
        const int64_t rows = 1600000;           // 1.64 GB: past the 256 MB Infinity Cache
        uint16_t* E; hipMalloc(&E, (size_t)rows * 1024 + 4096);
        fill<<<4096, 256>>>(E, (size_t)rows * 512, 3u, 0);
        hipDeviceSynchronize();
        const int64_t tr = 800000;              // = 8 sets of 100 000 rows
        for (int nt = 0; nt < 2; ++nt) {
            run_dma("own rows", E, 2 * tr, 0, 0, 4, 4, nt, out);
            run_dma("own rows, 1 KiB pieces", E, 2 * tr, 0, 1, 4, 4, nt, out);
            run_dma("pairs share (P+Q)", E, tr, 2, 0, 4, 4, nt, out);
            run_dma("pairs share, 1 KiB pieces", E, tr, 2, 1, 4, 4, nt, out);
        }
        run_dma("pairs share, ring 5", E, tr, 2, 0, 4, 5, 0, out);
        run_dma("pairs share, ring 3", E, tr, 2, 0, 4, 3, 0, out);
        run_dma("pairs share, 2 issuers", E, tr, 2, 0, 2, 4, 0, out);
        run_dma("pairs share, 1 issuer", E, tr, 2, 0, 1, 4, 0, out);
        run_dma("own rows, 1 issuer", E, 2 * tr, 0, 0, 1, 4, 0, out);
        run_dma("own rows, ring 5", E, 2 * tr, 0, 0, 4, 5, 0, out);
        run_dma("quads share", E, tr / 2, 4, 0, 4, 4, 0, out);
    }
    if (strchr(only, 'C')) {
        const int64_t rows = 1600000;
        uint16_t* E; hipMalloc(&E, (size_t)rows * 1024 + 4096);
        fill<<<4096, 256>>>(E, (size_t)rows * 512, 3u, 0);
        hipDeviceSynchronize();
        const int64_t tr = 800000;
        run_dma("B: pairs share, 1 KiB pieces", E, tr, 2, 1, 4, 4, 0, out);
        for (int buf = 0; buf < 2; ++buf)
            for (int swz = 0; swz < 2; ++swz)
                for (int split = 0; split < 2; ++split) run_c("pairs share", E, tr, 2, swz, split, buf, 0, out);
        run_c("pairs share, Q late", E, tr, 2, 1, 1, 1, 1, out);
        run_c("pairs share, Q late", E, tr, 2, 0, 0, 0, 1, out);
        run_c("own rows", E, 2 * tr, 0, 1, 1, 1, 0, out);
        run_c("own rows", E, 2 * tr, 0, 0, 0, 0, 0, out);
    }
    return 0;
}
