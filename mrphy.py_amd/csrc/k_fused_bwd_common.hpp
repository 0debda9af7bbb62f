// k_fused_bwd_common.hpp -- what the one-coil and the multi-coil fused adjoints share
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

// =============================================================================================
// K2b: adjoint of the fused kernel -- grad_Mo -> grad_Mi, grad_rf, grad_gr without Beff, history
// or grad_Beff in HBM (single-coil rf).  K2 leaves a checkpoint of M every SEG = 16 steps.  A wave
// walks the segments of its 64 spins backwards; per segment it
//   1. recomputes the 16 pre-step states from the checkpoint into registers (the very states the
//      forward pass went through, so no inversion error),
//   2. sweeps the adjoint over the 16 steps, re-assembling the field on the fly,
//   3. reduces the five per-step contributions
//        gr_x,y,z += loc_{x,y,z} * gBz     rf_re += b1r*gBx + b1i*gBy     rf_im += b1r*gBy - b1i*gBx
//      over its 64 spins with an LDS transpose-sum (80 rows x 64 lanes, slot-swizzled: conflict-free
//      ds_read_b128), and adds the 80 sums into ITS OWN row of the workspace.
// Waves are persistent (grid.x = min(tiles, 2048)) and take tiles w, w+P, ... in order, so every
// workspace row is accumulated in a fixed order; a second pass sums the rows in fixed order:
// deterministic, no float atomics.
// =============================================================================================
// (SEG = 16 steps per checkpoint segment: geom.hpp)
// Reduction tile: 80 rows x 64 lanes, NO padding (20480 B = exactly 1/8 of a CU's LDS, so 8 waves
// = 2 per SIMD are resident; with a padded pitch of 68 it was 21760 B -> 7 per CU, SIMD load
// 2:2:2:1).  Conflict-free row reads come from an XOR swizzle of the 16-B slot index instead:
// element (row, lane) lives in slot (lane/4) ^ (row & 15).
constexpr int RED_PITCH = WAVE;
// (K2B_MAX_WAVES = 256 * 8 resident waves, 8 per CU: geom.hpp)
// The swizzle is a bijection of the 16 slots of a row for ANY row count, so red_idx is correct for every SEG; it
// is conflict-free for the 16-row groups of SEG = 16 it was laid out for.  The single-coil kernel needs
// 5 * SEG <= 2 * WAVE rows (two passes of row sums) and batches of 4 steps; the multi-coil kernel needs SEG == 16
// outright (k_fused_mc_bwd.hpp).
static_assert(SEG % 4 == 0 && 5 * SEG <= 2 * WAVE, "K2b: 4-step batches, 5 * SEG reduction rows in two passes");
static_assert(SEG == 16, "K2b's reduction tile (red_idx: row & 15, 20480 B = 1/8 of a CU's LDS) is laid out for SEG = 16");
__device__ __forceinline__ int red_idx(int row, int l)
{
    return row * RED_PITCH + ((((l >> 2) ^ (row & 15)) << 2) | (l & 3));
}

template <typename T>
struct FusedBwdArgs {
    const T* Mck;                    // (nT/SEG, N*nM, 3)
    const T* rf;  int64_t rf_sn;
    const T* gr;  int64_t gr_sn;
    const T* loc;
    Bc df, gam;
    const T* b1;                     // (N, nM, 2) or null
    Bc g, E1, E2;
    const void* E1m1;
    const T* gMo;
    T* gMi;                          // may be null
    T* work;                         // (P, N, 5, nT)
    int64_t N, nM, nT, P;
};

// Pass 2: sum the P workspace rows per (n, quantity, t) in a fixed order.  Block = 32 time points
// x 8 row groups (group g takes rows g, g+8, ...: 128-B coalesced reads per row), then the eight
// partial sums are combined through LDS in group order -- deterministic, and nT/32 * 5 blocks
// instead of nT/256 * 5 (40 blocks at nT = 2048 took 0.45 ms for 73 MB).
constexpr int P2_T = 32, P2_G = 8;
