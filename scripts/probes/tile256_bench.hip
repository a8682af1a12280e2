// Timing harness for the shipped 256-column-slab kernel (fadtk_amd/csrc/moments_tile256.h) and its reduce at the config-3 pair
// (2 x [100000 x 512] float16, three pairs rotated: every launch streams from HBM), built in several variants by -D flags:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DT2_OPT_KS_READS] [-DT2_OPT_NOCOLSUM] [-DT2_NST_VALUE=3] -o /tmp/t256b scripts/probes/tile256_bench.hip
// Prints the average kernel / reduce duration (HIP events) and a checksum of the accumulators.
#include "../../fadtk_amd/csrc/moments_tile256.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fad;
__global__ void fill(uint16_t* E, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // ~N(0, 1) with full float16 mantissas (sum of four uniforms; data with a few significant bits lets the chip clock higher)
        uint32_t h = (uint32_t)(i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.0f / 256.0f) - 1.9921875f;
        _Float16 v = (_Float16)(u * 1.7320508f);
        __builtin_memcpy(&E[i], &v, 2);
    }
}
int main(int argc, char** argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 512;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 100000;
    const int sets = getenv("T2_SETS") ? atoi(getenv("T2_SETS")) : 2, npairs = (getenv("T2_SAME_PAIR") ? 1 : 3), n_cu = 256;
    const int nsb = (d + 255) / 256, dpad = nsb * 256;
    std::vector<uint16_t*> E(npairs * sets);
    for (auto& p : E) {
        hipMalloc(&p, (size_t)n * d * 2 + 4096); fill<<<2048, 256>>>(p, (size_t)n * d, (uint32_t)(&p - E.data()) * 7919u);
        if (getenv("T2_ZERO")) hipMemset(p, 0, (size_t)n * d * 2);        // all-zero frames: the same instruction stream at the chip's lowest switching power
    }
    T256Launch L; memset(&L, 0, sizeof(L));
    L.nsets = sets; L.d = d; L.nsb = nsb; L.NT = t256::item_types(nsb, L.type, L.sa, L.sb);
    const int kb = (nsb & 1) ? 64 : 32;
    const size_t lds = kT256Lds;
    int64_t r = ((n * sets * L.NT + n_cu - 1) / n_cu + kb - 1) / kb * kb;
    while (sets * ((n + r - 1) / r) * L.NT > n_cu && r < 8192) r += kb;
    if (getenv("T2_ROWS")) r = atoll(getenv("T2_ROWS"));             // probe: rows per split that are NOT a multiple of the stage
    const int S = (int)((n + r - 1) / r);
    printf("d=%d n=%ld: %d item types, %d splits of %ld rows per set, %d workgroups\n", d, (long)n, L.NT, S, (long)r, sets * S * L.NT);
    std::vector<t256::BlockSrc> tab(t256::n_blocks(8 * nsb));
    if (!t256::build_block_table(nsb, tab.data())) return 1;
    t256::BlockSrc* dtab; hipMalloc(&dtab, tab.size() * sizeof(tab[0])); hipMemcpy(dtab, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice);
    R256Launch R; memset(&R, 0, sizeof(R));
    R.table = dtab; R.d = d; R.nsb = nsb; R.NT = L.NT; R.nblk = t256::n_blocks(8 * nsb); R.two_mask = (nsb & 1) ? (1u << (nsb - 1)) : 0u;
    int* flags; hipMalloc(&flags, 64); hipMemset(flags, 0, 64);
    int item = 0;
    for (int i = 0; i < sets; ++i) {
        T256Set& s = L.set[i];
        s.n = n; s.ld = d; s.rows_per_split = r; s.S = S; s.item0 = item; item += S * L.NT;
        hipMalloc(&s.partials, (size_t)S * L.NT * t256::ITEM_STRIDE * 4);
        hipMalloc(&s.colpart, (size_t)S * 2 * dpad * 8);
        hipMalloc(&s.cvec, (size_t)S * dpad * 2);
        s.flag = flags + 2 * i;
        R256Job& j = R.job[i];
        j.partials = s.partials; j.colpart = s.colpart; j.cvec = s.cvec; j.gate = s.flag; j.clear_flag = flags + 2 * i + 1;
        hipMalloc(&j.acc, (size_t)(1 + d + (size_t)d * d) * 8);
        j.n_add = (double)n; j.S = S; j.overwrite = 1; j.rows_per_split = r; j.n_rows = n;
    }
    L.total = item;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_tile256<FAD_F16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kT256Lds);
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    for (int sl : {8, 4, 2, 1}) {
        R.sl = sl;
        const int G = 256 / sl;
        const int blocks = (R.nblk * 256 + G - 1) / G + (d + 63) / 64;
        double tk = 0, tr = 0, tr2 = 0; const int reps = getenv("T2_REPS") ? atoi(getenv("T2_REPS")) : 12;      // (T2_REPS: long runs for clock / power sampling)
        for (int it = -2; it < reps; ++it) {
            for (int i = 0; i < sets; ++i) L.set[i].E = E[((it + 2) % npairs) * sets + i];
            hipEventRecord(e[0]);
            moments_tile256<FAD_F16, false><<<L.total, 512, lds>>>(L);
            hipEventRecord(e[1]);
            moments_reduce256<<<dim3(blocks, sets), 256>>>(R);
            hipEventRecord(e[2]);
            if (getenv("T2_REDUCE_TWICE")) {       // the same reduce again, right behind the first: partials clean, as resident as they get
                moments_reduce256<<<dim3(blocks, sets), 256>>>(R);
                hipEventRecord(e[3]);
                hipEventSynchronize(e[3]);
                float c; hipEventElapsedTime(&c, e[2], e[3]); if (it >= 0) tr2 += c;
            }
            if (hipEventSynchronize(e[2]) != hipSuccess) { printf("failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            float a, b; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]);
            if (it >= 0) { tk += a; tr += b; }
        }
        std::vector<double> h(1 + d + (size_t)d * d);
        hipMemcpy(h.data(), R.job[sets - 1].acc, h.size() * 8, hipMemcpyDeviceToHost);
        double ck1 = 0, ck2 = 0;            // checksums over the whole accumulator of the last set (compare builds of the kernel with each other)
        for (size_t i = 0; i < h.size(); ++i) { ck1 += h[i]; ck2 += h[i] * (double)((i * 2654435761u) % 1021); }
        const double fl = sets * 2.0 * n * d * d;
        if (tr2 > 0) printf("  (second reduce right behind the first: %6.1f us)\n", tr2 / reps * 1e3);
        printf("  sl=%2d: tile %7.1f us (%5.1f %% of 2.5 PF algorithmic)   reduce %6.1f us (%d workgroups)   [n=%g sum0=%.6g M00=%.9g ck=%.12g %.12g]\n", sl, tk / reps * 1e3,
               fl / (tk / reps * 1e-3) / 2.5e15 * 100, tr / reps * 1e3, blocks * sets, h[0], h[1], h[1 + d], ck1, ck2);
    }
    return 0;
}
