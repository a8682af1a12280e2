// Which (problem, item) a workgroup of the batched chain kernels (ns_fast_big.h) takes -- plain C++, shared by the device code, the host's
// grid sizes (frechet.hip) and the CPU test of the coverage (tests/native_cpu/big_slots_cover.cpp).
//
// A 1-D grid, `per_song` items (products x tiles) to a problem.  Workgroup L runs on XCD L % 8 and is that XCD's (L / 8)-th: the XCD takes
// problems xcd, xcd + 8, ... one after the other, item by item -- the tiles of a product run side by side on ONE XCD and walk the k range in
// step, so its L2 fetches every operand strip once for the 2 t tiles that read it.  (With z = problem the tiles were dealt round the eight
// XCDs and every L2 fetched everything: 885 MB per T launch at D = 768 x 32 songs through the fabric, 4.9 TB/s, a quarter of the matrix
// rate -- profiles/r03i_c5_kernel_stats.csv.)  The nprob % 8 problems left over after the groups of eight are cut into eight runs of
// consecutive items, one per XCD: 20 pairs cost every XCD 2.5 problems -- not three on the XCDs that would hold a whole third one beside
// four idle ones.
#pragma once

namespace fad {
namespace nsf {

struct BigSlot { int song, item; bool live; };

// items per XCD of the leftover problems
constexpr int big_tail_run(int nprob, int per_song) { return ((nprob & 7) * per_song + 7) >> 3; }
// workgroups of a launch (a multiple of eight: the XCD of a workgroup is its index modulo eight)
constexpr int big_grid(int nprob, int per_song) { return 8 * ((nprob >> 3) * per_song + big_tail_run(nprob, per_song)); }

constexpr BigSlot big_slot(int L, int nprob, int per_song) {
    const int xcd = L & 7, idx = L >> 3, whole = (nprob >> 3) * per_song;
    if (idx < whole) return BigSlot{8 * (idx / per_song) + xcd, idx % per_song, true};
    const int run = big_tail_run(nprob, per_song), j = idx - whole, w = xcd * run + j;
    if (j >= run || w >= (nprob & 7) * per_song) return BigSlot{0, 0, false};
    return BigSlot{(nprob & ~7) + w / per_song, w % per_song, true};
}

// The iteration products' tile: 128 x 64 NJ.  The narrow one (NJ = 1) when the wide one would leave CUs with a lone workgroup
// (ns_fast_big.h: nsf_big) -- `products` = 1 (SP_T, SP_FIRST) or 2 (SP_U) per problem.
constexpr int big_nj(int d, int products, int nprob) { return (products * (d / 128) * (d / 128) * nprob < 512) ? 1 : 2; }
constexpr int big_tiles(int d, int nj) { return (d / 128) * (d / (64 * nj)); }

}  // namespace nsf
}  // namespace fad
