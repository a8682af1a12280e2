// Who computes what in the 256-column-slab E^T E kernel (moments_tile256.h) -- plain C++, shared by the device code, the host
// planner / reduce tables (moments.hip) and the CPU test that checks the coverage (tests/native_cpu/tile256_cover.cpp).
//
// The D columns of E are cut into "superblocks" of 256 columns = 8 "fragments" of 32 columns (one MFMA operand each).  A
// workgroup (8 waves, ONE per CU) streams TWO superblock slabs A and B of its row range through LDS -- 32 rows x 256 columns
// x 2 = 32 KiB per stage -- and its waves own 32 x 32 blocks of E^T E by ROLE:
//
//   TRI_LO / TRI_HI / RECT_C / RECT_D   four waves that cover the 36 blocks on and above the diagonal of ONE superblock's
//                                       256 x 256 diagonal tile, 9 blocks per wave (fragments read per 16-row k-step: 4, 4, 6, 6)
//   XR                                  a 128 x 64 rectangle of an off-diagonal tile: 4 A-side x 2 B-side fragments, 8 blocks
//
// and a work item (one workgroup) is of one of four TYPES:
//
//   P(a, b)   waves 0-3: triangle of superblock a;   waves 4-7: rows 0..127 of the full tile (a, b)      68 blocks
//   Q(a, b)   waves 0-3: triangle of superblock b;   waves 4-7: rows 128..255 of the full tile (a, b)    68 blocks
//   X(a, b)   all eight waves XR: the full tile (a, b), a < b                                            64 blocks
//   Z(a)      both wave quartets the triangle of a, on ALTERNATE 32-row stages (slab B = the next 32 rows of slab A's
//             columns): the leftover diagonal tile of an odd superblock count                             2 x 36 blocks
//
// P and Q of a pair split the 136 blocks of "two triangles + the tile between them" evenly and read the SAME 64 KB of every
// 32 rows, so that with both on one XCD the frames cross HBM once.  Every SIMD runs one triangle wave (18 MFMAs per stage) and
// one XR wave (16).
#pragma once
#include <cstdint>

namespace fad {
namespace t256 {

constexpr int SB = 256;            // columns per superblock
constexpr int FR = 32;             // columns per fragment
constexpr int NFR = SB / FR;       // fragments per superblock (8)
constexpr int SLOTS = 72;          // partial-tile slots per work item: wave w, local block b -> slot 9 w + b (XR waves use 8 of their 9)
constexpr int BLK = FR * FR;       // floats per 32 x 32 block
constexpr int ITEM_STRIDE = SLOTS * BLK + 64;     // floats per work item's partials (+256 B: consecutive items do not alias one channel)
constexpr int MAX_SB = 8;          // D <= 2048
constexpr int MAX_TYPES = 32;      // work items per row-split at MAX_SB: 4 pairs x 2 + 24 full tiles

enum Role : int { TRI_LO = 0, TRI_HI = 1, RECT_C = 2, RECT_D = 3, XR = 4 };
enum Type : int { TYPE_P = 0, TYPE_Q = 1, TYPE_X = 2, TYPE_Z = 3 };

// fragments a role loads per k-step (index into the wave's F[] array) and the blocks it owns: block b = F[fa[b]]^T F[fb[b]]
template <int ROLE> struct RoleDef;
template <> struct RoleDef<TRI_LO> {       // fragments 0..3 of the superblock: T(0..3) without (3, 3)
    static constexpr int NF = 4, NB = 9;
    static constexpr int frag[6] = {0, 1, 2, 3, -1, -1};
    static constexpr int fa[9] = {0, 0, 0, 0, 1, 1, 1, 2, 2};
    static constexpr int fb[9] = {0, 1, 2, 3, 1, 2, 3, 2, 3};
};
template <> struct RoleDef<TRI_HI> {       // fragments 4..7: T(4..7) without (4, 4)
    static constexpr int NF = 4, NB = 9;
    static constexpr int frag[6] = {4, 5, 6, 7, -1, -1};
    static constexpr int fa[9] = {0, 0, 0, 1, 1, 1, 2, 2, 3};
    static constexpr int fb[9] = {1, 2, 3, 1, 2, 3, 2, 3, 3};
};
template <> struct RoleDef<RECT_C> {       // rows {0, 1} x columns {4..7}, + (4, 4)
    static constexpr int NF = 6, NB = 9;
    static constexpr int frag[6] = {0, 1, 4, 5, 6, 7};
    static constexpr int fa[9] = {0, 0, 0, 0, 1, 1, 1, 1, 2};
    static constexpr int fb[9] = {2, 3, 4, 5, 2, 3, 4, 5, 2};
};
template <> struct RoleDef<RECT_D> {       // rows {2, 3} x columns {4..7}, + (3, 3)
    static constexpr int NF = 6, NB = 9;
    static constexpr int frag[6] = {2, 3, 4, 5, 6, 7};
    static constexpr int fa[9] = {0, 0, 0, 0, 1, 1, 1, 1, 1};
    static constexpr int fb[9] = {2, 3, 4, 5, 2, 3, 4, 5, 1};
};
template <> struct RoleDef<XR> {           // F[0..3] = A-side fragments a0..a0+3 (slab A), F[4..5] = B-side fragments b0, b0+1 (slab B)
    static constexpr int NF = 6, NB = 8;
    static constexpr int frag[6] = {0, 1, 2, 3, 0, 1};      // relative to a0 / b0
    static constexpr int fa[9] = {0, 0, 1, 1, 2, 2, 3, 3, -1};
    static constexpr int fb[9] = {4, 5, 4, 5, 4, 5, 4, 5, -1};
};

// What wave `wave` of a work item of `type` does: its role; for triangle roles the slab (0 = A, 1 = B) whose superblock it works
// on; for XR the first A-side fragment (slab A) and the first B-side fragment (slab B).
struct WaveJob { int role, slab, a0, b0; };
inline constexpr WaveJob wave_job(int type, int wave) {
    if (type == TYPE_X) return WaveJob{XR, 0, 4 * (wave >> 2), 2 * (wave & 3)};
    if (type == TYPE_Z) return WaveJob{wave & 3, wave >> 2, 0, 0};
    if (wave < 4) return WaveJob{wave, type == TYPE_Q ? 1 : 0, 0, 0};
    return WaveJob{XR, 0, type == TYPE_Q ? 4 : 0, 2 * (wave - 4)};
}

// The work items of one row-split for `nsb` superblocks: pairs (2p, 2p+1) as P + Q, every other tile a < b as X, the last
// superblock of an odd count as Z.  Returns the count; sa/sb = the superblocks behind slab A / slab B (Z: sb = sa).
inline int item_types(int nsb, uint8_t* type, uint8_t* sa, uint8_t* sb) {
    int n = 0;
    for (int p = 0; 2 * p + 1 < nsb; ++p) {
        type[n] = TYPE_P; sa[n] = (uint8_t)(2 * p); sb[n] = (uint8_t)(2 * p + 1); ++n;
        type[n] = TYPE_Q; sa[n] = (uint8_t)(2 * p); sb[n] = (uint8_t)(2 * p + 1); ++n;
    }
    for (int a = 0; a < nsb; ++a)
        for (int b = a + 1; b < nsb; ++b) {
            if ((a & 1) == 0 && b == a + 1) continue;          // inside a P/Q pair
            type[n] = TYPE_X; sa[n] = (uint8_t)a; sb[n] = (uint8_t)b; ++n;
        }
    if (nsb & 1) { type[n] = TYPE_Z; sa[n] = sb[n] = (uint8_t)(nsb - 1); ++n; }
    return n;
}

// Global block coordinates (in fragments of 32 columns: row block bi <= column block bj) of local block `b` of wave `wave` of
// an item (type, sa, sb); false for the unused ninth slot of an XR wave.
inline bool slot_block(int type, int sa, int sb, int wave, int b, int* bi, int* bj) {
    const WaveJob j = wave_job(type, wave);
    auto tri = [&](const int* frag, const int* fa, const int* fb, int nb) {
        if (b >= nb) return false;
        const int base = NFR * (j.slab == 0 ? sa : sb);
        *bi = base + frag[fa[b]]; *bj = base + frag[fb[b]];
        return true;
    };
    switch (j.role) {
        case TRI_LO: return tri(RoleDef<TRI_LO>::frag, RoleDef<TRI_LO>::fa, RoleDef<TRI_LO>::fb, 9);
        case TRI_HI: return tri(RoleDef<TRI_HI>::frag, RoleDef<TRI_HI>::fa, RoleDef<TRI_HI>::fb, 9);
        case RECT_C: return tri(RoleDef<RECT_C>::frag, RoleDef<RECT_C>::fa, RoleDef<RECT_C>::fb, 9);
        case RECT_D: return tri(RoleDef<RECT_D>::frag, RoleDef<RECT_D>::fa, RoleDef<RECT_D>::fb, 9);
        default:
            if (b >= 8) return false;
            *bi = NFR * sa + j.a0 + RoleDef<XR>::fa[b];
            *bj = NFR * sb + j.b0 + (RoleDef<XR>::fb[b] - 4);
            return true;
    }
}

// Reduce table: for every block (bi <= bj) of the nb x nb block grid (nb = 8 nsb), where its partial sums sit:
// src[k] = item_type_index * SLOTS + slot, k = 0 (always) and 1 (the second wave quartet of a Z item, else -1).
struct BlockSrc { int16_t src[2]; };
inline int block_index(int bi, int bj, int nb) { return bi * nb - bi * (bi - 1) / 2 + (bj - bi); }      // upper triangle, row major
inline int n_blocks(int nb) { return nb * (nb + 1) / 2; }
inline bool build_block_table(int nsb, BlockSrc* table /* n_blocks(8 nsb) */) {
    uint8_t type[MAX_TYPES], sa[MAX_TYPES], sb[MAX_TYPES];
    const int nt = item_types(nsb, type, sa, sb), nb = NFR * nsb;
    for (int i = 0; i < n_blocks(nb); ++i) table[i].src[0] = table[i].src[1] = -1;
    for (int t = 0; t < nt; ++t)
        for (int w = 0; w < 8; ++w)
            for (int b = 0; b < 9; ++b) {
                int bi, bj;
                if (!slot_block(type[t], sa[t], sb[t], w, b, &bi, &bj)) continue;
                if (bi > bj || bj >= nb) return false;
                BlockSrc& e = table[block_index(bi, bj, nb)];
                const int16_t s = (int16_t)(t * SLOTS + 9 * w + b);
                if (e.src[0] < 0) e.src[0] = s;
                else if (e.src[1] < 0 && type[t] == TYPE_Z) e.src[1] = s;
                else return false;                                   // covered twice
            }
    for (int i = 0; i < n_blocks(nb); ++i) if (table[i].src[0] < 0) return false;                          // not covered
    return true;
}

}  // namespace t256
}  // namespace fad
