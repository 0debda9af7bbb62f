// geom.hpp -- sizes and capacities that more than one translation unit must agree on: the kernels, their
// launchers (tu_*.hip) and the workspace / buffer-size queries of the C ABI (abi.hip) all read them here.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace mrphy {

constexpr int WAVE = 64;
constexpr int HIST_STEP = 3 * WAVE;               // history: elements per time step of one 64-spin tile
constexpr int AB_HIST_STEP = 12 * WAVE;           // beff2ab history: the 3x4 state per step and tile
constexpr int SEG = 16;                           // K2 / K2b: steps per checkpoint segment
// Generic in SEG: K2 (k_fused_fwd.hpp: only the checkpoint stride), the checkpoint / workspace size queries.
// NOT generic: k_bloch_rfgr_bwd_mc (step = lane >> 2, needs SEG * 4 == WAVE) and the reduction tile of both fused
// adjoints (red_idx).  Both carry a static_assert; change SEG only together with them.
constexpr int64_t K2B_MAX_WAVES = 256 * 8;        // K2b: resident waves, 8 per CU
constexpr int K2B_MAXC = 8;                       // fused adjoint: largest coil capacity
constexpr int64_t K2B_MC_MAX_WAVES = 256 * 8;     // 18 KB of LDS per wave -> 8 per CU = 2 per SIMD
constexpr int BWD_MAXC = 32;                      // K0 adjoint: largest coil capacity of the one-pass kernels

// broadcastable per-spin constant (see mrphy_hip.h): element (n, s) at p[n * sn + s * sm]
struct Bc {
    const void* p;
    int64_t sn, sm;
};

inline int64_t hist_tiles(int64_t N, int64_t nM) { return (N * nM + WAVE - 1) / WAVE; }
inline int64_t hist_elems(int64_t N, int64_t nM, int64_t nT)
{
    return hist_tiles(N, nM) * nT * HIST_STEP;
}

// The history of K1h / K3 (ABI 5) as 1..HIST_MAX_PARTS separately allocated parts: the 64-spin tiles are dealt to the
// parts in blocks (tile -> part tile / tiles_per_part) or round-robin (tile -> part tile % n_parts), a tile's
// nT * HIST_STEP elements are contiguous inside its part.  One launch then writes (K1h) / reads (K3) every part at the
// same time: a write stream spread over two separately allocated blocks runs in the fast placement mode where ONE
// block of the same total size usually does not.  Passed to the kernels by value (kernarg).
constexpr int HIST_MAX_PARTS = 8;
struct HistParts {
    void* p[HIST_MAX_PARTS];
    uint32_t tiles_per_part;          // every part holds this many tiles (the last one may use fewer)
    int32_t n_parts;                  // 0: no history wanted
    int32_t interleaved;              // 0: blocks of tiles_per_part consecutive tiles; 1: round-robin
};
inline int64_t hist_tiles_per_part(int64_t N, int64_t nM, int64_t n_parts)
{
    return n_parts > 0 ? (hist_tiles(N, nM) + n_parts - 1) / n_parts : 0;
}
inline HistParts hist_one_part(const void* p, int64_t N, int64_t nM)
{
    HistParts h = {};
    h.p[0] = const_cast<void*>(p);
    h.n_parts = p ? 1 : 0;
    h.tiles_per_part = (uint32_t)hist_tiles(N, nM);
    return h;
}

inline int64_t k2b_waves(int64_t nM)
{
    const int64_t tiles = (nM + WAVE - 1) / WAVE;
    return tiles < K2B_MAX_WAVES ? tiles : K2B_MAX_WAVES;
}

inline int64_t k2b_mc_waves(int64_t nM)
{
    const int64_t tiles = (nM + WAVE - 1) / WAVE;
    return tiles < K2B_MC_MAX_WAVES ? tiles : K2B_MC_MAX_WAVES;
}

// coil capacity of the one-pass K0 adjoint for nC coils: 8 / 16 / 32, or 0 = the generic passes
// (no b1 map, or more than BWD_MAXC coils).  Used by the launcher AND the workspace query.
// Round 4: more than BWD_MAXC coils run the same pass over blocks of BWD_MAXC coils (every coil's sums are
// independent of the others', so nothing is carried between blocks: grad_Beff is read once per block).
inline int bwd_capacity(int64_t nC, bool has_b1)
{
    if (nC < 2 || !has_b1) return 0;
    return nC <= 8 ? 8 : (nC <= 16 ? 16 : 32);
}
// padded coil count of the SGPR pass (k_rfgr2beff_bwd_sgpr) for nC coils, 0 as above
inline int bwd_padded_coils(int64_t nC, bool has_b1)
{
    if (!bwd_capacity(nC, has_b1)) return 0;
    if (nC > BWD_MAXC) return BWD_MAXC;               // blocks of 32 (the last one padded to its own count)
    return nC <= 4 ? 4 : (nC <= 8 ? 8 : (nC <= 12 ? 12 : (nC <= 16 ? 16 : (nC <= 24 ? 24 : 32))));
}

inline int64_t bwd_spin_groups(int64_t nM);
// Workspace of the multi-coil K0 adjoint (2..BWD_MAXC coils): [partial sums (nSG, N, 3 + 2 nC, nT) |
// packed coefficient rows (N nM, 2 MC + 4)], the second part on a 256-byte boundary.
inline size_t bwd_pack_offset(size_t ts, int64_t N, int64_t nM, int64_t nT, int64_t nC)
{
    const size_t sums = (size_t)(bwd_spin_groups(nM) * N * (3 + 2 * nC) * nT) * ts;
    return (sums + 255) / 256 * 256;
}
inline int64_t bwd_spin_groups(int64_t nM)
{
    int64_t g = (nM + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256) g = 256;
    return g;
}

}  // namespace mrphy
