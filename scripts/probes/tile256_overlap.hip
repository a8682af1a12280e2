// Can the reduce of update i run BESIDE the tile kernel of update i + 1?  (fadtk_amd/csrc/moments_tile256.h)
// The tile kernel holds every CU with one 8-wave workgroup (2 x 224 VGPRs per SIMD, 128 KiB of LDS); a reduce workgroup (4 waves,
// 48 VGPRs, 8 KiB) fits beside it.  Main stream (high priority): tile kernels back to back on double-buffered partials; side stream
// (low priority): the reduces, each behind its tile kernel's event, with T2_PAD bytes of dynamic LDS to bound how many of its
// workgroups a CU takes.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/t256o scripts/probes/tile256_overlap.hip
#include "../../fadtk_amd/csrc/moments_tile256.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fad;
__global__ void fill(uint16_t* E, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.0f / 256.0f) - 1.9921875f;
        _Float16 v = (_Float16)(u * 1.7320508f);
        __builtin_memcpy(&E[i], &v, 2);
    }
}
int main(int argc, char** argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 512;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 100000;
    const int sets = 2, npairs = 3, n_cu = 256, NBUF = 2;
    const int nsb = (d + 255) / 256, dpad = nsb * 256;
    std::vector<uint16_t*> E(npairs * sets);
    for (auto& p : E) { hipMalloc(&p, (size_t)n * d * 2 + 4096); fill<<<2048, 256>>>(p, (size_t)n * d, (uint32_t)(&p - E.data()) * 7919u); }
    T256Launch L[NBUF]; R256Launch R[NBUF];
    t256::BlockSrc* dtab = nullptr;
    int blocks = 0;
    for (int q = 0; q < NBUF; ++q) {
        memset(&L[q], 0, sizeof(L[q])); memset(&R[q], 0, sizeof(R[q]));
        L[q].nsets = sets; L[q].d = d; L[q].nsb = nsb; L[q].plan = 0; L[q].NT = t256::item_types(nsb, L[q].type, L[q].sa, L[q].sb, 0);
        const int kb = (nsb & 1) ? 64 : 32;
        int64_t r = ((n * sets * L[q].NT + n_cu - 1) / n_cu + kb - 1) / kb * kb;
        while (sets * ((n + r - 1) / r) * L[q].NT > n_cu && r < 8192) r += kb;
        const int S = (int)((n + r - 1) / r);
        if (!dtab) {
            std::vector<t256::BlockSrc> tab(t256::n_blocks(8 * nsb));
            if (!t256::build_block_table(nsb, tab.data(), 0)) return 1;
            hipMalloc(&dtab, tab.size() * sizeof(tab[0])); hipMemcpy(dtab, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice);
        }
        R[q].table = dtab; R[q].d = d; R[q].nsb = nsb; R[q].NT = L[q].NT; R[q].nblk = t256::n_blocks(8 * nsb); R[q].two_mask = (nsb & 1) ? (1u << (nsb - 1)) : 0u;
        int* flags; hipMalloc(&flags, 64); hipMemset(flags, 0, 64);
        int item = 0;
        for (int i = 0; i < sets; ++i) {
            T256Set& s = L[q].set[i];
            s.n = n; s.ld = d; s.rows_per_split = r; s.S = S; s.item0 = item; item += S * L[q].NT;
            hipMalloc(&s.partials, (size_t)S * L[q].NT * t256::ITEM_STRIDE * 4);
            hipMalloc(&s.colpart, (size_t)S * 2 * dpad * 8);
            hipMalloc(&s.cvec, (size_t)S * dpad * 2);
            s.flag = flags + 2 * i;
            R256Job& j = R[q].job[i];
            j.partials = s.partials; j.colpart = s.colpart; j.cvec = s.cvec; j.gate = s.flag; j.clear_flag = flags + 2 * i + 1;
            hipMalloc(&j.acc, (size_t)(1 + d + (size_t)d * d) * 8);
            j.n_add = (double)n; j.S = S; j.overwrite = 1; j.rows_per_split = r; j.n_rows = n;
        }
        L[q].total = item;
        R[q].sl = (S > 32) ? 16 : (S > 8) ? 4 : 1;
        const int G = 256 / R[q].sl;
        blocks = (R[q].nblk * 256 + G - 1) / G + (d + 63) / 64;
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile256<FAD_F16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kT256LdsCombined);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_reduce256), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    hipStream_t sm, ss;
    const bool prio = !getenv("T2_NO_PRIO");
    hipStreamCreateWithPriority(&sm, hipStreamNonBlocking, prio ? hi : 0);
    hipStreamCreateWithPriority(&ss, hipStreamNonBlocking, prio ? lo : 0);
    hipEvent_t te[NBUF], re[NBUF];
    for (int q = 0; q < NBUF; ++q) { hipEventCreateWithFlags(&te[q], hipEventDisableTiming); hipEventCreateWithFlags(&re[q], hipEventDisableTiming); }
    const int reps = 60;
    for (int pad : {0, 2048, 4096, 6144, 8192, 10240, 12288, 16384, 20480}) {
        for (int mode = 0; mode < (getenv("T2_SIDE") ? 2 : 1); ++mode) {        // 0: everything on the main stream; 1: reduces on the side stream
            double best = 1e9;
            for (int trial = 0; trial < 3; ++trial) {
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                for (int it = 0; it < reps; ++it) {
                    const int q = it % NBUF;
                    for (int i = 0; i < sets; ++i) L[q].set[i].E = E[(it % npairs) * sets + i];
                    if (mode == 1 && it >= NBUF) hipStreamWaitEvent(sm, re[q], 0);      // the partials of buffer q are free again
                    moments_tile256<FAD_F16, false><<<L[q].total, 512, kT256Lds, sm>>>(L[q]);
                    if (mode == 1) { hipEventRecord(te[q], sm); hipStreamWaitEvent(ss, te[q], 0); }
                    moments_reduce256<<<dim3(blocks, sets), 256, pad, mode == 1 ? ss : sm>>>(R[q]);
                    if (mode == 1) hipEventRecord(re[q], ss);
                }
                hipStreamSynchronize(sm); hipStreamSynchronize(ss);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
                if (us < best) best = us;
            }
            std::vector<double> h(4);
            hipMemcpy(h.data(), R[1].job[1].acc, h.size() * 8, hipMemcpyDeviceToHost);
            printf("  reduce pad %5d B, %s: %7.1f us per update (tile + reduce)   [n=%g sum0=%.6g]\n", pad, mode ? "reduce on a side stream" : "one stream             ", best, h[0], h[1]);
        }
    }
    return hipGetLastError() != hipSuccess;
}
