// CPU check of fadtk_amd/csrc/tile256_roles.h (g++, no GPU): for every superblock count the work items of a row-split cover
// every 32 x 32 block on or above the diagonal exactly once (twice on the Z item's triangle: its two wave quartets take
// alternate stages), and the matrix-pipe load is balanced over the four SIMDs (wave w and w + 4 share one).
#include "../../fadtk_amd/csrc/tile256_roles.h"
#include <cstdio>
#include <vector>
using namespace fad::t256;
int main() {
    for (int nsb = 1; nsb <= MAX_SB; ++nsb) {
        const int nb = NFR * nsb;
        std::vector<BlockSrc> tab((size_t)n_blocks(nb));
        if (!build_block_table(nsb, tab.data())) { printf("nsb=%d: table inconsistent\n", nsb); return 1; }
        uint8_t type[MAX_TYPES], sa[MAX_TYPES], sb[MAX_TYPES];
        const int nt = item_types(nsb, type, sa, sb);
        if (nt > MAX_TYPES) { printf("nsb=%d: %d item types\n", nsb, nt); return 1; }
        std::vector<int> count((size_t)nb * nb, 0);
        long blocks = 0;
        for (int t = 0; t < nt; ++t) {
            int per_simd[4] = {0, 0, 0, 0};
            for (int w = 0; w < 8; ++w)
                for (int b = 0; b < 9; ++b) {
                    int bi, bj;
                    if (!slot_block(type[t], sa[t], sb[t], w, b, &bi, &bj)) continue;
                    count[(size_t)bi * nb + bj]++; per_simd[w & 3]++; ++blocks;
                }
            for (int q = 1; q < 4; ++q)
                if (per_simd[q] != per_simd[0]) { printf("nsb=%d item %d: SIMD loads %d %d %d %d\n", nsb, t, per_simd[0], per_simd[1], per_simd[2], per_simd[3]); return 1; }
        }
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j < nb; ++j) {
                const bool z = (nsb & 1) && i / NFR == nsb - 1 && j / NFR == nsb - 1;
                const int want = (i <= j) ? (z ? 2 : 1) : 0;
                if (count[(size_t)i * nb + j] != want) { printf("nsb=%d block (%d,%d): %d, want %d\n", nsb, i, j, count[(size_t)i * nb + j], want); return 1; }
            }
        // the table's entries point back at the right slots
        for (int i = 0; i < nb; ++i)
            for (int j = i; j < nb; ++j) {
                const BlockSrc e = tab[(size_t)block_index(i, j, nb)];
                for (int h = 0; h < 2; ++h) {
                    if (e.src[h] < 0) continue;
                    const int t = e.src[h] / SLOTS, slot = e.src[h] % SLOTS;
                    int bi, bj;
                    if (!slot_block(type[t], sa[t], sb[t], slot / 9, slot % 9, &bi, &bj) || bi != i || bj != j) { printf("nsb=%d: table entry (%d,%d) wrong\n", nsb, i, j); return 1; }
                }
            }
        printf("nsb=%d: %d item types, %ld blocks per split, %d output blocks ok\n", nsb, nt, blocks, n_blocks(nb));
    }
    return 0;
}
