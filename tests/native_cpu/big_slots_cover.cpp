// CPU check of fadtk_amd/csrc/big_slots.h (g++, no GPU): for every batch size and items-per-problem count the workgroups of a launch take
// every (problem, item) exactly once, a whole group of eight problems keeps each problem on ONE XCD, and no XCD gets more than one item
// above its share.
#include "../../fadtk_amd/csrc/big_slots.h"
#include <cstdio>
#include <vector>
using namespace fad::nsf;
int main() {
    const int per[] = {1, 2, 4, 8, 9, 16, 18, 32, 36, 64, 72, 128};
    for (int nprob = 1; nprob <= 70; ++nprob)
        for (int per_song : per) {
            const int grid = big_grid(nprob, per_song);
            if (grid % 8) { printf("nprob=%d per=%d: grid %d\n", nprob, per_song, grid); return 1; }
            std::vector<int> seen((size_t)nprob * per_song, 0), home((size_t)nprob, -1);
            int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int L = 0; L < grid; ++L) {
                const BigSlot s = big_slot(L, nprob, per_song);
                if (!s.live) continue;
                if (s.song < 0 || s.song >= nprob || s.item < 0 || s.item >= per_song) { printf("nprob=%d per=%d L=%d: (%d, %d)\n", nprob, per_song, L, s.song, s.item); return 1; }
                seen[(size_t)s.song * per_song + s.item]++; load[L & 7]++;
                if (s.song < (nprob & ~7)) {
                    if (home[s.song] >= 0 && home[s.song] != (L & 7)) { printf("nprob=%d per=%d: problem %d on two XCDs\n", nprob, per_song, s.song); return 1; }
                    home[s.song] = L & 7;
                }
            }
            for (size_t k = 0; k < seen.size(); ++k)
                if (seen[k] != 1) { printf("nprob=%d per=%d: item %zu taken %d times\n", nprob, per_song, k, seen[k]); return 1; }
            int lo = load[0], hi = load[0];
            for (int x = 1; x < 8; ++x) { if (load[x] < lo) lo = load[x]; if (load[x] > hi) hi = load[x]; }
            const int share = (nprob * per_song + 7) / 8;
            if (hi > share) { printf("nprob=%d per=%d: an XCD takes %d items, share %d\n", nprob, per_song, hi, share); return 1; }
        }
    printf("big_slots ok: 20 problems x 16 items -> grid %d, %d per XCD\n", big_grid(20, 16), big_grid(20, 16) / 8);
    return 0;
}
