// Round-6 probe for the batched chain's k loop (ns_fast_big.h): how many operand bytes per second does ONE CU take from its XCD's L2
//   0. through the LDS-DMA path alone (what nsf_big does: 16 pieces of 1 KiB per k-step and workgroup, ring of three, one barrier a step),
//   1. through plain global_load_dwordx4 into registers alone (every wave its own 8 fragments: 32 KiB per k-step and workgroup),
//   2. through both at once (A operands by LDS-DMA: 8 KiB; B operands into registers: 16 KiB),
// with 1 / 2 / 3 workgroups of 256 threads per CU and, optionally, the k-step's LDS reads (8 ds_read_b128 per lane) beside them?
// The source is L2-resident (2 MiB per XCD, walked again and again).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/r6_l2_paths scripts/probes/r6_l2_paths.hip && /tmp/r6_l2_paths
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRegion = 2 << 20;             // bytes per XCD

struct Args { const char* src; int steps, reads; };

template <int MODE>
__global__ __launch_bounds__(256, 3) void loop(Args a, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, nth = blockIdx.x >> 3;
    const char* base = a.src + (size_t)xcd * kRegion;
    const uint32_t smem_lds = (uint32_t)(size_t)(lptr_t)smem;
    const uint32_t voff = (uint32_t)(lane * 16);
    constexpr int kDma = (MODE == 0) ? 4 : ((MODE == 2) ? 2 : 0);      // LDS-DMA pieces per wave and step
    constexpr int kReg = (MODE == 1) ? 8 : ((MODE == 2) ? 4 : 0);      // register loads per wave and step
    constexpr int kPieces = 4 * kDma;                                   // per workgroup and step, through LDS
    uint32_t pos = (uint32_t)(nth * 37 * 1024) & (kRegion - 1);         // where this workgroup is in the region
    auto dma = [&](int s) {
#pragma unroll
        for (int p = 0; p < kDma; ++p) {
            const uint64_t sb = (uint64_t)(base + ((pos + (uint32_t)(s * 32 + wave * kDma + p) * 1024u) & (kRegion - 1)));
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(smem_lds + (uint32_t)(((s % 3) * 16 + wave * kDma + p) * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ub), "s"(m0v) : "memory", "m0");
        }
    };
    // register loads as asm as well: the compiler's own s_waitcnt for a C++ load counts only the loads it knows of and would wait for
    // the LDS-DMA pieces issued before them too (vmcnt is in order); the waits below name the registers they make valid
    u32x4 r[2][8];
    auto regs = [&](int s, int b) {
#pragma unroll
        for (int p = 0; p < kReg; ++p) {
            const uint64_t sb = (uint64_t)(base + ((pos + (uint32_t)(s * 32 + 16 + wave * kReg + p) * 1024u) & (kRegion - 1)));
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            const uint64_t ub = ((uint64_t)hi << 32) | lo;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[b][p]) : "v"(voff), "s"(ub) : "memory");
        }
    };
    u32x4 acc = {0, 0, 0, 0};
    auto fold = [&](const u32x4 v) { acc ^= v; };
    auto foldq = [&](const uint4 v) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; };
    if (kDma) { dma(0); dma(1); }
    if (kReg) regs(0, 0);
    for (int s = 0; s < a.steps; s += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ss = s + h;
            if (kReg) regs(ss + 1, h ^ 1);
            // stage ss has landed: one younger stage of DMA pieces + the register loads just issued may still be in flight
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[h][0]), "+v"(r[h][1]), "+v"(r[h][2]), "+v"(r[h][3]), "+v"(r[h][4]), "+v"(r[h][5]), "+v"(r[h][6]), "+v"(r[h][7]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" : "+v"(r[h][0]), "+v"(r[h][1]), "+v"(r[h][2]), "+v"(r[h][3]) :: "memory");
            if (kDma) {
                __builtin_amdgcn_s_barrier();
                dma(ss + 2);
                if (a.reads) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) foldq(smem[((ss % 3) * 16 + ((q * 2 + (wave & 1)) % (kPieces ? kPieces : 1))) * 64 + lane]);
                }
            }
#pragma unroll
            for (int p = 0; p < kReg; ++p) fold(r[h][p]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}

template <int MODE>
static void run(const char* what, const char* src, int wgs_per_cu, int reads, uint32_t* out) {
    Args a; a.src = src; a.steps = 4096; a.reads = reads;
    const size_t lds = 3 * 16 * 1024 + 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop<MODE><<<grid, 256, lds>>>(a, out); hipDeviceSynchronize();
    float best = 1e9f, sum = 0; const int reps = 5;
    for (int q = 0; q < reps; ++q) {
        hipEventRecord(e0); loop<MODE><<<grid, 256, lds>>>(a, out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    const double lds_bytes = (MODE == 0 ? 16384.0 : (MODE == 2 ? 8192.0 : 0.0)), reg_bytes = (MODE == 1 ? 32768.0 : (MODE == 2 ? 16384.0 : 0.0));
    const double t = sum / reps * 1e-3, per_cu = (lds_bytes + reg_bytes) * a.steps * wgs_per_cu / t / 1e9;
    printf("%-34s %d WG/CU reads=%d: %7.1f us (best %7.1f)  %6.1f ns per k-step and workgroup   %6.1f GB/s per CU (LDS-DMA %5.1f + registers %5.1f)\n", what, wgs_per_cu, reads,
           t * 1e6, best * 1e3, t * 1e9 / a.steps, per_cu, lds_bytes * a.steps * wgs_per_cu / t / 1e9, reg_bytes * a.steps * wgs_per_cu / t / 1e9);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  error: %s\n", hipGetErrorString(e));
}

int main() {
    char* src; uint32_t* out;
    hipMalloc(&src, (size_t)8 * kRegion + 65536); hipMemset(src, 1, (size_t)8 * kRegion + 65536);
    hipMalloc(&out, 4096 * 4);
    for (int reads = 0; reads < 2; ++reads)
        for (int w = 1; w <= 3; ++w) {
            run<0>("LDS-DMA alone (16 KiB a step)", src, w, reads, out);
            if (!reads) run<1>("registers alone (32 KiB a step)", src, w, reads, out);
            run<2>("A by LDS-DMA, B into registers", src, w, reads, out);
        }
    hipDeviceSynchronize();
    return 0;
}
