// mrphy_hip.hip -- gfx950 kernels + C ABI (include/mrphy_hip.h) for the Bloch-simulation hot path.
//
// Execution model shared by the time-stepping kernels (K1 fwd, K3 bwd, K2 fused):
//   * one workgroup = ONE 64-lane wavefront = 64 consecutive spins (rows of the compact layout);
//     lane = spin, the magnetisation lives in 3 VGPRs for the whole pulse, time loop in-kernel;
//   * Beff is (rows, nT, 3): a spin's samples are contiguous in time, lanes are nT*12 B apart.
//     A chunk of TC steps x 64 spins is fetched with lanes running ALONG TIME (16 B per lane,
//     whole 128-B lines per row), parked in VGPRs while the previous chunk is being integrated
//     (register prefetch: the bytes in flight live in registers, not in LDS), then transposed
//     through a padded LDS tile so that each lane reads its own spin's samples with
//     conflict-free ds_read_b128;
//   * history / gradient tiles go back the same way (in place in the LDS tile, coalesced store).
// No MFMA anywhere: the only contraction (loc . gr) has K = 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/mrphy_hip.h"
#include "bloch_math.hpp"

using namespace mrphy;

namespace {
#include "k_common.hpp"
#include "k_blochsim.hpp"
#include "k_rfgr2beff.hpp"
#include "k_fused.hpp"
#include "k_aux.hpp"

// ---------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------
inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return (int)e;
}

inline size_t tsize(int dtype) { return dtype == MRPHY_F64 ? 8 : 4; }
inline size_t csize(int dtype) { return (dtype == MRPHY_F32 || dtype == MRPHY_F32P) ? 4 : 8; }

// steps per chunk of the chunked kernels.  fp32: 16 steps = 192 B per row.  fp64 forward: 8 steps =
// 192 B per row as well -- with 16 (384 B, 24 staging vectors = 96 VGPRs per lane) the fp64 forward
// builds need 354 VGPRs = ONE wave per SIMD; with 8, 244 = two (same-box A/B at 64^3 x 1024, round 3:
// K1 1.55 -> 1.42 ms, K1h 2.49 -> 2.40 ms).  The fp64 adjoint stays at 16: with 8 it got slower
// (3.92 -> 4.51 ms; 356 VGPRs either way is one wave per SIMD, and the smaller chunk doubles the
// barriers).  Putting the fp64 large-angle path (ocml sincos) behind a real call to shrink the
// kernels was tried too: the call's register convention spilled the HOT path (fused K2 0.78 -> 2.58 ms).
template <typename T> constexpr int TC_FWD = sizeof(T) == 8 ? 8 : 16;
template <typename T> constexpr int TC_BWD = 16;

// Development knobs exist only in the -DMRPHY_DEV_KNOBS build (tools/build_dev.py ->
// tools/libmrphy_hip_dev.so): environment variables that select alternative builds / block orders
// for A/B measurements (re-read at every launch, so one process can sweep them), and a
// per-workgroup time-stamp buffer.  The shipped library reads no
// environment variable and instantiates none of the alternatives.
#ifdef MRPHY_DEV_KNOBS
inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// MRPHY_K0_VARIANT = order*1000 + rows_per_block/8*10 + nt
inline int k0_variant() { return env_int("MRPHY_K0_VARIANT", 0); }
// MRPHY_BWD_VARIANT = waves per SIMD the K3 build is bounded for (2, 3, 4)
inline int bwd_variant() { return env_int("MRPHY_BWD_VARIANT", 0); }
// MRPHY_XCD_SWEEP=0 turns the XCD-contiguous tile order of the line kernels off
inline bool xcd_sweep() { return env_int("MRPHY_XCD_SWEEP", 1) != 0; }
// MRPHY_FWD_VARIANT = OCC*100 + SPLIT*10 + NT selects an alternative K1 build
inline int fwd_variant() { return env_int("MRPHY_FWD_VARIANT", 0); }
// MRPHY_K0_STEPS=0: multi-coil rfgr2beff on the element-per-thread builds instead of k_rfgr2beff_steps
inline bool k0_steps() { return env_int("MRPHY_K0_STEPS", 1) != 0; }
// MRPHY_K0_PK=0: exact coil counts on k_rfgr2beff_steps instead of the packed-scalar kernel k_rfgr2beff_pk
inline bool k0_pk() { return env_int("MRPHY_K0_PK", 1) != 0; }
// MRPHY_K0ADJ_TP: alternative multi-coil K0 adjoints (1|2|4: DPP pass, 0: element-per-thread pass, 12: SGPR pass, TP = 2)
inline int k0adj_tp() { return env_int("MRPHY_K0ADJ_TP", -1); }
// MRPHY_PRIO_ROT=N (re-read at every launch): rotate s_setprio with progress in the line kernels
inline int prio_rot() { return env_int("MRPHY_PRIO_ROT", 0); }
// MRPHY_LDS_PAD=bytes of dynamic LDS added to the line kernels' launches: caps the workgroups per CU
// (160 KB / (9216 + pad)) without touching the code -- occupancy experiments
inline unsigned lds_pad() { return (unsigned)env_int("MRPHY_LDS_PAD", 0); }
unsigned long long* g_dev_stamps = nullptr;        // 4 x uint64 per workgroup, or null
int64_t g_dev_stamps_cap = 0;                      // workgroups the buffer holds
#else
constexpr int k0_variant() { return 0; }
constexpr int bwd_variant() { return 0; }
constexpr bool xcd_sweep() { return true; }
constexpr int fwd_variant() { return 0; }
constexpr bool k0_steps() { return true; }
constexpr bool k0_pk() { return true; }
constexpr unsigned lds_pad() { return 0; }
#endif

inline int64_t hist_elems(int64_t N, int64_t nM, int64_t nT)
{
    return ((N * nM + WAVE - 1) / WAVE) * nT * HIST_STEP;
}

// rows on 128-B lines and whole 32-step periods: what the line-granular kernels need
inline bool lines_shape_ok(const void* Beff, int64_t nT)
{
    return aligned_to(Beff, 128) && nT > 0 && (nT % 32 == 0) && (768 * nT < (int64_t)4294967295);
}

template <typename T, typename CT>
int run_fwd(const void* Mi, const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo,
            void* Mpre, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    FwdArgs<T> a;
    a.Mi = (const T*)Mi; a.Beff = (const T*)Beff; a.Mo = (T*)Mo; a.Mpre = (T*)Mpre;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    // vector path of the chunked kernel (16-B global accesses need element alignment only)
    a.vec_ok = aligned_to(Beff, sizeof(T));      // element alignment is enough (V16::utype)
    a.per_xcd = 0;
    if (a.rows == 0) return 0;
    dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
#ifdef MRPHY_DEV_KNOBS
    a.stamps = (int64_t)grid.x <= g_dev_stamps_cap ? g_dev_stamps : nullptr;
    a.prio_rot = prio_rot();
#endif
    if constexpr (sizeof(T) == 4) {
        const int v = fwd_variant();
        if (lines_shape_ok(Beff, nT) && v != 16 && v != 32) {
            // XCD-contiguous tile order pays where the kernel writes (history: 10.07 -> 8.75 ms at
            // 128^3 x 1024); for the read-only forward it is neutral (15.70 vs 15.60 ms), left off.
            if (xcd_sweep() && Mpre) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; }
            // development knob MRPHY_FWD_VARIANT = OCC*100 + SPLIT*10 + NT selects a build.
            // measured on MI355X, 128^3 x 4096, no history (ms): 320 16.88 | 321 15.82 |
            // 330 17.14 | 331 15.72
#define MRPHY_L(OCC_, SP_, NT_, SV_)                                                             \
    do {                                                                                         \
        if (E1.p) hipLaunchKernelGGL((k_bloch_fwd_lines<CT, true, OCC_, SP_, NT_, SV_>), grid, \
                                     dim3(WAVE), lds_pad(), st, a);                              \
        else      hipLaunchKernelGGL((k_bloch_fwd_lines<CT, false, OCC_, SP_, NT_, SV_>), grid, \
                                     dim3(WAVE), lds_pad(), st, a);                              \
    } while (0)
            if (Mpre) {
                // with history: 3 waves/SIMD (the 4-wave build: 10.06 vs 8.72 ms at 128^3 x 1024)
                MRPHY_L(3, 3, true, true);
            } else {
                // The 3-wave build with 3-/4-step batches (139-150 VGPRs, no scratch) for every mode.
                // Rounds 1-2 ran the fast step on a 4-wave build (2-/3-step batches, 128 VGPRs, 12 B/lane
                // of scratch outside the loop) because a 64^3 grid is then one generation of waves
                // (measured then: 64^3 x 4096 2.29 vs 2.48 ms, 128^3 x 4096 equal).  On three boxes in
                // round 3 the 3-wave build won everywhere (ms, fast step, 3-wave | 4-wave build):
                // 64^3 x 1024 0.53 | 0.56, 64^3 x 2048 0.98 | 1.19, 64^3 x 4096 1.97 | 2.24,
                // 128^3 x 1024 3.85 | 4.29, 128^3 x 4096 14.96-15.09 | 16.8-17.7 (0.85 vs 0.73-0.77 of peak)
                // -- and capping the 4-wave BUILD at 3 or 2 waves/SIMD (dynamic LDS padding, dev knob
                // MRPHY_LDS_PAD) leaves it where it is (0.72-0.75): it is the code of the small batches
                // (more LDS round trips and barriers per piece), not the occupancy; the 3-wave build is
                // indifferent to caps of 8...16 waves per CU (profiles/r03_occupancy_cap_*.json).
                // Precise step: the 4-wave build spills (44 B/lane, 6 scratch accesses per 32 steps):
                // 18.7-20.2 vs 15.5-15.8 ms.
#ifdef MRPHY_DEV_KNOBS
                switch (v) {
                case 321: MRPHY_L(3, 2, true, false); return launch_status();
                case 331: MRPHY_L(3, 3, true, false); return launch_status();
                case 341: MRPHY_L(3, 4, true, false); return launch_status();
                case 441: MRPHY_L(4, 4, true, false); return launch_status();
                default: break;
                }
#endif
                MRPHY_L(3, 3, true, false);
            }
#undef MRPHY_L
            return launch_status();
        }
    }
    if (Mpre)
        hipLaunchKernelGGL((k_bloch_fwd<T, CT, TC_FWD<T>, true>), grid, dim3(WAVE), 0, st, a);
#ifdef MRPHY_DEV_KNOBS
    else if (fwd_variant() == 32)
        hipLaunchKernelGGL((k_bloch_fwd<T, CT, 32, false>), grid, dim3(WAVE), 0, st, a);
#endif
    else
        hipLaunchKernelGGL((k_bloch_fwd<T, CT, TC_FWD<T>, false>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

template <typename T, typename CT>
int run_bwd(const void* Mpre, const void* Beff, Bc g, Bc E1, Bc E2, const void* gMo, void* gMi,
            void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    BwdArgs<T> a;
    a.Mpre = (const T*)Mpre; a.Beff = (const T*)Beff; a.gMo = (const T*)gMo;
    a.gMi = (T*)gMi; a.gBeff = (T*)gBeff; a.gC = (T*)gC;
    a.g = g; a.E1 = E1; a.E2 = E2;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, sizeof(T)) && (!gBeff || aligned_to(gBeff, sizeof(T)));
    a.per_xcd = 0;
    if (a.rows == 0) return 0;
    dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
#ifdef MRPHY_DEV_KNOBS
    a.stamps = (int64_t)grid.x <= g_dev_stamps_cap ? g_dev_stamps : nullptr;
    a.prio_rot = prio_rot();
#endif
    if (gC) {      // gradients w.r.t. the constants as well: the chunked kernel's GC build (any shape)
        hipLaunchKernelGGL((k_bloch_bwd<T, CT, TC_BWD<T>, true>), grid, dim3(WAVE), 0, st, a);
        return launch_status();
    }
    if constexpr (sizeof(T) == 4) {
        if (lines_shape_ok(Beff, nT) && (!gBeff || aligned_to(gBeff, 128)) &&
            fwd_variant() != 16) {
            if (xcd_sweep()) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; }
            const int occ = bwd_variant();
            // development knob MRPHY_BWD_VARIANT = waves per SIMD the build is bounded for (2, 3)
#define MRPHY_LB(OCC_)                                                                           \
    do {                                                                                         \
        if (E1.p) hipLaunchKernelGGL((k_bloch_bwd_lines<CT, true, OCC_, true>), grid,             \
                                     dim3(WAVE), lds_pad(), st, a);                              \
        else      hipLaunchKernelGGL((k_bloch_bwd_lines<CT, false, OCC_, true>), grid,            \
                                     dim3(WAVE), lds_pad(), st, a);                              \
    } while (0)
            // same-box A/B at 128^3 x 1024 (ms), round 2: history fetched in-batch 13.28 | one batch
            // ahead: 2 waves/SIMD 12.83, 3 waves/SIMD 13.04 with 36 B/lane of spills; without forming
            // w, v in the adjoint step the 3-wave build has 134-136 VGPRs and no spills: 12.6-12.8, the
            // default (round 3, 64^3 x 2048: 2 | 3 | 4 waves 3.33 | 3.36 | 3.32 ms: no occupancy effect)
#ifdef MRPHY_DEV_KNOBS
            if (occ == 2) { MRPHY_LB(2); return launch_status(); }
            if (occ == 4) { MRPHY_LB(4); return launch_status(); }
#endif
            (void)occ;
            MRPHY_LB(3);
#undef MRPHY_LB
            return launch_status();
        }
    }
    hipLaunchKernelGGL((k_bloch_bwd<T, CT, TC_BWD<T>, false>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

template <typename T>
int run_rfgr2beff(const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc,
                  Bc df, Bc gam, const void* b1, void* beff, int64_t N, int64_t nM, int64_t nT,
                  int64_t nC, hipStream_t st)
{
    BeffArgs<T> a;
    a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr; a.gr_sn = gr_sn;
    a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1; a.beff = (T*)beff;
    a.nM = nM; a.nT = nT; a.nC = nC;
    if (N * nM * nT == 0) return 0;
    // Block order matters more than anything else here.  Blocks are dealt round-robin to the 8 XCDs,
    // so block b works on tile (b % 8) * per_xcd + b / 8: every XCD (and its L2) sweeps its own
    // contiguous eighth of Beff, time tiles fastest, i.e. 8 linear write streams.
    // measured (128^3 x 4096, ms; v = order*1000 + rows/8*10 + nt):
    //   order 0 (spin tile fastest, grid y = time tile): 128 rows+nt 16.1-17.2 | 64 rows 16.6
    //   order 1 (time tile fastest, no XCD split):       128 rows+nt 19.3
    //   order 2 (XCD sweep): 8 rows+nt 16.4 | 16+nt 14.56 | 16 14.77 | 32+nt 14.90 | 48+nt 14.92
    //                        64 15.09 | 64+nt 17.45 | 128 15.20 | 128+nt 17.88
    a.rows_per_block = 16;
    a.nt = 1;
    int order = 2;
    if (k0_variant() > 0) {
        a.nt = (k0_variant() % 10) != 0; a.rows_per_block = (k0_variant() % 1000 / 10) * 8;
        order = k0_variant() / 1000;
    }
    if (a.rows_per_block < 8) a.rows_per_block = 64;
    if (a.rows_per_block > K0_MAX_ROWS) a.rows_per_block = K0_MAX_ROWS;
    // The build: smallest register/LDS coil capacity that holds nC; everything that depends on it
    // -- elements per thread (hence the time-tile count gy), the cap on rows per block (the
    // kernel's LDS array of b1 rows) -- is read from K0Geom, the table the kernel itself uses.
    const int ncm = (nC == 1) ? 1 : (!b1 ? 0 : (nC <= 8 ? 8 : (nC <= 16 ? 16 : (nC <= K0_MAXC ? 32 : 0))));
    const bool vec = aligned_to(beff, sizeof(T));
    const int64_t L = 3 * nT;
    const dim3 block(K0_THREADS);
    // 2..32 coils with a map: the step-per-thread kernel (a thread owns whole time points)
    auto launch_steps = [&](auto mc_tag) -> int {
        constexpr int MC = decltype(mc_tag)::value;
        using G = K0StepGeom<T, MC>;
        if (k0_variant() <= 0 || a.rows_per_block > G::ROWS) a.rows_per_block = G::ROWS;
        const int64_t gy = (nT + (int64_t)K0_THREADS * G::TP - 1) / ((int64_t)K0_THREADS * G::TP);
        if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
        const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
        dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
        a.gy = 0; a.nblk = 0; a.per_xcd = 0;
        if (order >= 1 && gx * gy < (int64_t(1) << 31) - 8) {
            a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
            grid = dim3(a.nblk, 1, (unsigned)N);
            if (order == 2) { a.per_xcd = (a.nblk + 7) / 8; grid.x = a.per_xcd * 8; }
        }
        hipLaunchKernelGGL((k_rfgr2beff_steps<T, MC>), grid, block, 0, st, a);
        return launch_status();
    };
    // exact coil counts 4 / 8 / 12 / 16 with a map: b1 rows as scalar operands of packed FMAs
    auto launch_pk = [&](auto nc_tag) -> int {
        constexpr int NC = decltype(nc_tag)::value;
        a.rows_per_block = 128;
        const int64_t gy = (nT + (int64_t)K0_THREADS * 2 - 1) / ((int64_t)K0_THREADS * 2);
        if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
        const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
        dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
        a.gy = 0; a.nblk = 0; a.per_xcd = 0;
        if (gx * gy < (int64_t(1) << 31) - 8) {
            a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
            a.per_xcd = (a.nblk + 7) / 8;
            grid = dim3(a.per_xcd * 8, 1, (unsigned)N);
        }
        hipLaunchKernelGGL((k_rfgr2beff_pk<T, NC>), grid, block, 0, st, a);
        return launch_status();
    };
    // (measured, 64^3 x 1024, ms, steps kernel | this one: 4 coils 0.80 | 0.68, 8 coils 0.85 | 0.71, 12 coils
    // 0.92 | 0.90, 16 coils 0.99 | 0.86 -- and 24 coils 1.16 | 1.30, 32 coils 1.32 | 1.60: there the scalar
    // loads of the rows, 3.3-3.6 B/ns per CU when waves walk their own rows, are the wall; up to 16 only)
    if (vec && b1 && k0_pk()) {
        switch (nC) {
        case 4:  return launch_pk(std::integral_constant<int, 4>{});
        case 8:  return launch_pk(std::integral_constant<int, 8>{});
        case 12: return launch_pk(std::integral_constant<int, 12>{});
        case 16: return launch_pk(std::integral_constant<int, 16>{});
#ifdef MRPHY_DEV_KNOBS
        case 24: if (sizeof(T) == 4) return launch_pk(std::integral_constant<int, 24>{}); break;
        case 32: if (sizeof(T) == 4) return launch_pk(std::integral_constant<int, 32>{}); break;
#endif
        default: break;
        }
    }
    if (vec && k0_steps()) {                         // (dev knob MRPHY_K0_STEPS=0: the element-per-thread builds)
        if (ncm == 8)  return launch_steps(std::integral_constant<int, 8>{});
        if (ncm == 16) return launch_steps(std::integral_constant<int, 16>{});
        if (ncm == 32) return launch_steps(std::integral_constant<int, 32>{});
    }
    auto launch = [&](auto ncm_tag) -> int {
        constexpr int NCM = decltype(ncm_tag)::value;
        using G = K0Geom<T, NCM>;
        if (a.rows_per_block > G::ROWS) a.rows_per_block = G::ROWS;
        const int vw = vec ? G::VW : 1;
        const int64_t gy = (L + (int64_t)K0_THREADS * vw - 1) / ((int64_t)K0_THREADS * vw);
        if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
        const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
        dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
        a.gy = 0; a.nblk = 0; a.per_xcd = 0;
        if (order >= 1 && gx * gy < (int64_t(1) << 31) - 8) {
            a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
            grid = dim3(a.nblk, 1, (unsigned)N);
            if (order == 2) { a.per_xcd = (a.nblk + 7) / 8; grid.x = a.per_xcd * 8; }
        }
        if (vec) hipLaunchKernelGGL((k_rfgr2beff<T, G::VW, NCM>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_rfgr2beff<T, 1, NCM>), grid, block, 0, st, a);
        return launch_status();
    };
    switch (ncm) {
    case 1:  return launch(std::integral_constant<int, 1>{});
    case 8:  return launch(std::integral_constant<int, 8>{});
    case 16: return launch(std::integral_constant<int, 16>{});
    case 32: return launch(std::integral_constant<int, 32>{});
    default: return launch(std::integral_constant<int, 0>{});
    }
}

// coil capacity of the one-pass K0 adjoint for nC coils: 8 / 16 / 32, or 0 = the generic passes
// (no b1 map, or more than BWD_MAXC coils).  Used by the launcher AND the workspace query.
inline int bwd_capacity(int64_t nC, bool has_b1)
{
    if (nC < 2 || !has_b1 || nC > BWD_MAXC) return 0;
    return nC <= 8 ? 8 : (nC <= 16 ? 16 : 32);
}
// padded coil count of the SGPR pass (k_rfgr2beff_bwd_sgpr) for nC coils, 0 as above
inline int bwd_padded_coils(int64_t nC, bool has_b1)
{
    if (!bwd_capacity(nC, has_b1)) return 0;
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

template <typename T>
int run_rfgr2beff_bwd(const void* gB, const void* loc, const void* b1, void* grf, void* ggr,
                      void* work, int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st)
{
    BeffBwdArgs<T> a;
    a.gB = (const T*)gB; a.loc = (const T*)loc; a.b1 = (const T*)b1; a.work = (T*)work;
    a.grf = (T*)grf; a.ggr = (T*)ggr;
    a.N = N; a.nM = nM; a.nT = nT; a.nC = nC;
    a.nSG = bwd_spin_groups(nM);
    a.spins_per_group = (nM + a.nSG - 1) / a.nSG;
    if (N * nT == 0) return 0;
    if (N * (nC + 1) > 65535 || 3 + 2 * nC > 65535) return MRPHY_EINVAL;
    const unsigned tx = (unsigned)((nT + 255) / 256);
    if (nC == 1) {                                       // vector-load path
        const int64_t L = 3 * nT;
        constexpr int VWV = V16<T>::N;
        const bool vec = aligned_to(gB, sizeof(T));
        const int vw = vec ? VWV : 1;
        const dim3 g1((unsigned)((L + 256 * (int64_t)vw - 1) / (256 * (int64_t)vw)), (unsigned)a.nSG,
                      (unsigned)N);
        if (vec) hipLaunchKernelGGL((k_rfgr2beff_bwd_p1v<T, VWV>), g1, dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((k_rfgr2beff_bwd_p1v<T, 1>), g1, dim3(256), 0, st, a);
        int e = launch_status();
        if (e) return e;
        hipLaunchKernelGGL((k_rfgr2beff_bwd_p2v<T>), dim3(tx, 1, (unsigned)N), dim3(256), 0, st, a);
        return launch_status();
    }
    if (const int cap = bwd_capacity(nC, b1 != nullptr)) {   // 2..32 coils: one pass over gB
        using std::integral_constant;
        (void)cap;
#ifdef MRPHY_DEV_KNOBS
        // A/B baselines of the dev build (MRPHY_K0ADJ_TP): 1 / 2 / 4 = the DPP pass with that many time points
        // per thread, 0 = round 2's element-per-thread pass, 12 = the SGPR pass with two time points per thread
        int tp = k0adj_tp();
        if (tp > nT) tp = 1;                           // the DPP kernel reads TP whole time points per row
        auto launch_steps = [&](auto mc_tag, auto tp_tag) -> int {
            constexpr int MC = decltype(mc_tag)::value, TP = decltype(tp_tag)::value;
            const int64_t per_block = 256 * (int64_t)TP;
            const dim3 g1((unsigned)((nT + per_block - 1) / per_block), (unsigned)a.nSG, (unsigned)N);
            hipLaunchKernelGGL((k_rfgr2beff_bwd_steps<T, MC, TP>), g1, dim3(256), 0, st, a);
            int e = launch_status();
            if (e) return e;
            hipLaunchKernelGGL((k_rfgr2beff_bwd_p2<T>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                               dim3(256), 0, st, a);
            return launch_status();
        };
        switch (cap * 10 + tp) {
        case 81:  return launch_steps(integral_constant<int, 8>{}, integral_constant<int, 1>{});
        case 82:  return launch_steps(integral_constant<int, 8>{}, integral_constant<int, 2>{});
        case 84:  return launch_steps(integral_constant<int, 8>{}, integral_constant<int, 4>{});
        case 161: return launch_steps(integral_constant<int, 16>{}, integral_constant<int, 1>{});
        case 162: return launch_steps(integral_constant<int, 16>{}, integral_constant<int, 2>{});
        case 164: return launch_steps(integral_constant<int, 16>{}, integral_constant<int, 4>{});
        case 321: return launch_steps(integral_constant<int, 32>{}, integral_constant<int, 1>{});
        case 322: return launch_steps(integral_constant<int, 32>{}, integral_constant<int, 2>{});
        default: break;
        }
        if (tp == 0) {
            const int64_t L = 3 * nT;
            const bool vec = aligned_to(gB, sizeof(T));
            auto launch = [&](auto mc_tag) -> int {
                constexpr int MC = decltype(mc_tag)::value;
                using G = BwdGeom<T, MC>;
                const int vw = vec ? G::VW : 1;
                const dim3 g1((unsigned)((L + 256 * (int64_t)vw - 1) / (256 * (int64_t)vw)), (unsigned)a.nSG,
                              (unsigned)N);
                if (vec) hipLaunchKernelGGL((k_rfgr2beff_bwd_p1mc<T, G::VW, MC>), g1, dim3(256), 0, st, a);
                else     hipLaunchKernelGGL((k_rfgr2beff_bwd_p1mc<T, 1, MC>), g1, dim3(256), 0, st, a);
                int e = launch_status();
                if (e) return e;
                hipLaunchKernelGGL((k_rfgr2beff_bwd_p2mc<T, MC>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                                   dim3(256), 0, st, a);
                return launch_status();
            };
            switch (cap) {
            case 8:  return launch(integral_constant<int, 8>{});
            case 16: return launch(integral_constant<int, 16>{});
            default: return launch(integral_constant<int, 32>{});
            }
        }
#endif
        // The step-per-thread pass with the spins' coefficients in SGPRs: a pre-pass packs b1 and loc,
        // zero-padded to the padded coil count, behind the partial sums in the workspace (bwd_pack_offset:
        // launcher and query agree by construction); the partial sums have the layout of the generic pass 2.
        auto launch_sgpr = [&](auto mc_tag) -> int {
            constexpr int MC = decltype(mc_tag)::value;
            T* pk = reinterpret_cast<T*>(static_cast<char*>(work) + bwd_pack_offset(sizeof(T), N, nM, nT, nC));
            PackArgs<T> pa;
            pa.b1 = (const T*)b1; pa.loc = (const T*)loc; pa.pk = pk; pa.rows = N * nM; pa.nC = nC; pa.MC = MC;
            const int64_t words = N * nM * (2 * MC + 4);
            if ((words + 255) / 256 > 2147483647) return MRPHY_EINVAL;
            hipLaunchKernelGGL((k_pack_coefs<T>), dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, pa);
            int e = launch_status();
            if (e) return e;
            BeffBwdPkArgs<T> b;
            b.gB = a.gB; b.pk = pk; b.work = a.work; b.N = N; b.nM = nM; b.nT = nT; b.nC = nC;
            b.spins_per_group = a.spins_per_group;
#ifdef MRPHY_DEV_KNOBS
            if (k0adj_tp() == 12 && nT >= 2) {
                const dim3 g2((unsigned)((nT + 511) / 512), (unsigned)a.nSG, (unsigned)N);
                hipLaunchKernelGGL((k_rfgr2beff_bwd_sgpr<T, MC, 2>), g2, dim3(256), 0, st, b);
            } else
#endif
            {
                const dim3 g1((unsigned)((nT + 255) / 256), (unsigned)a.nSG, (unsigned)N);
                hipLaunchKernelGGL((k_rfgr2beff_bwd_sgpr<T, MC, 1>), g1, dim3(256), 0, st, b);
            }
            e = launch_status();
            if (e) return e;
            hipLaunchKernelGGL((k_rfgr2beff_bwd_p2<T>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                               dim3(256), 0, st, a);
            return launch_status();
        };
        switch (bwd_padded_coils(nC, true)) {
        case 4:  return launch_sgpr(integral_constant<int, 4>{});
        case 8:  return launch_sgpr(integral_constant<int, 8>{});
        case 12: return launch_sgpr(integral_constant<int, 12>{});
        case 16: return launch_sgpr(integral_constant<int, 16>{});
        case 24: return launch_sgpr(integral_constant<int, 24>{});
        default: return launch_sgpr(integral_constant<int, 32>{});
        }
    }
    hipLaunchKernelGGL((k_rfgr2beff_bwd_p1<T>), dim3(tx, (unsigned)a.nSG, (unsigned)(N * (nC + 1))),
                       dim3(256), 0, st, a);
    int e = launch_status();
    if (e) return e;
    hipLaunchKernelGGL((k_rfgr2beff_bwd_p2<T>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                       dim3(256), 0, st, a);
    return launch_status();
}

template <typename T, typename CT>
int run_rfgr_fwd(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                 int64_t nT, int64_t nC, hipStream_t st)
{
    FusedArgs<T> a;
    a.Mi = (const T*)Mi; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
    a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.Mo = (T*)Mo; a.Mck = (T*)Mck;
    a.ck_every = ck_every > 0 ? ck_every : 1;
    a.N = N; a.nM = nM; a.nT = nT; a.nC = nC;
    if (N * nM == 0) return 0;
    if (N > 65535) return MRPHY_EINVAL;
    const dim3 grid((unsigned)((nM + WAVE - 1) / WAVE), (unsigned)N);
#define MRPHY_K2(NCM_, CK_, RX_, HB_) \
    hipLaunchKernelGGL((k_bloch_rfgr_fwd<T, CT, NCM_, CK_, RX_, HB_>), grid, dim3(WAVE), 0, st, a)
#define MRPHY_K2H(NCM_, HB_)                                                                     \
    do {                                                                                         \
        if (ck) { if (rx) MRPHY_K2(NCM_, true, true, HB_); else MRPHY_K2(NCM_, true, false, HB_); }   \
        else    { if (rx) MRPHY_K2(NCM_, false, true, HB_); else MRPHY_K2(NCM_, false, false, HB_); } \
    } while (0)
#define MRPHY_K2C(NCM_) MRPHY_K2H(NCM_, true)
    const bool ck = (Mck != nullptr), rx = (E1.p != nullptr);
    // the smallest register/LDS coil capacity that holds nC (each build sizes its b1 registers and
    // its LDS rf buffer for exactly that capacity: never launch one with more coils than it holds)
    if (nC == 1 && b1) MRPHY_K2C(1);
    else if (nC == 1) MRPHY_K2H(1, false);               // no b1 map: Bxy = rf, no complex product
    else if (nC <= 2 && b1) MRPHY_K2C(2);                // (round 3: 2 coils no longer pay for 8)
    else if (nC <= 4 && b1) MRPHY_K2C(4);
    else if (nC <= 8 && b1) MRPHY_K2C(8);
    else if (nC <= 16 && b1) MRPHY_K2C(16);
    else if (nC <= K2_MAXC && b1) MRPHY_K2C(32);
    else MRPHY_K2C(0);
#undef MRPHY_K2C
#undef MRPHY_K2H
#undef MRPHY_K2
    return launch_status();
}

inline int64_t k2b_waves(int64_t nM)
{
    const int64_t tiles = (nM + WAVE - 1) / WAVE;
    return tiles < K2B_MAX_WAVES ? tiles : K2B_MAX_WAVES;
}

template <typename T, typename CT>
int run_rfgr_bwd(const void* Mck, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, const void* gMo, void* gMi, void* grf, void* ggr, void* work,
                 int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    FusedBwdArgs<T> a;
    a.Mck = (const T*)Mck; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
    a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.gMo = (const T*)gMo; a.gMi = (T*)gMi;
    a.work = (T*)work; a.N = N; a.nM = nM; a.nT = nT; a.P = k2b_waves(nM);
    if (N * nM * nT == 0) return 0;
    if (N > 65535) return MRPHY_EINVAL;
    const dim3 grid((unsigned)a.P, (unsigned)N);
    if (b1) {
        if (E1.p) hipLaunchKernelGGL((k_bloch_rfgr_bwd<T, CT, true, true>), grid, dim3(WAVE), 0, st, a);
        else      hipLaunchKernelGGL((k_bloch_rfgr_bwd<T, CT, false, true>), grid, dim3(WAVE), 0, st, a);
    } else {                                             // no b1 map: Bxy = rf
        if (E1.p) hipLaunchKernelGGL((k_bloch_rfgr_bwd<T, CT, true, false>), grid, dim3(WAVE), 0, st, a);
        else      hipLaunchKernelGGL((k_bloch_rfgr_bwd<T, CT, false, false>), grid, dim3(WAVE), 0, st, a);
    }
    int e = launch_status();
    if (e) return e;
    if (grf || ggr) {
        hipLaunchKernelGGL((k_bloch_rfgr_bwd_p2<T>),
                           dim3((unsigned)((nT + P2_T - 1) / P2_T), 5, (unsigned)N),
                           dim3(P2_T * P2_G), 0, st, (const T*)work, (T*)grf, (T*)ggr, N, nT, a.P);
        e = launch_status();
    }
    return e;
}

inline int64_t k2b_mc_waves(int64_t nM)
{
    const int64_t tiles = (nM + WAVE - 1) / WAVE;
    return tiles < K2B_MC_MAX_WAVES ? tiles : K2B_MC_MAX_WAVES;
}

template <typename T, typename CT>
int run_rfgr_mc_bwd(const void* Mck, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                    const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                    const void* E1m1, const void* gMo, void* gMi, void* grf, void* ggr, void* work,
                    int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st)
{
    FusedBwdArgs<T> a;
    a.Mck = (const T*)Mck; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
    a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.gMo = (const T*)gMo; a.gMi = (T*)gMi;
    a.work = (T*)work; a.N = N; a.nM = nM; a.nT = nT; a.P = k2b_mc_waves(nM);
    if (N * nM * nT == 0) return 0;
    if (N > 65535) return MRPHY_EINVAL;
    const dim3 grid((unsigned)a.P, (unsigned)N);
    // the smallest coil capacity (2 / 4 / 8) that holds nC: the build's loops run over all of it, on zeros
#define MRPHY_K2BMC(MC_)                                                                                       \
    do {                                                                                                       \
        if (E1.p) hipLaunchKernelGGL((k_bloch_rfgr_bwd_mc<T, CT, true, MC_>), grid, dim3(WAVE), 0, st, a, (int)nC); \
        else      hipLaunchKernelGGL((k_bloch_rfgr_bwd_mc<T, CT, false, MC_>), grid, dim3(WAVE), 0, st, a, (int)nC); \
    } while (0)
    if (nC <= 2) MRPHY_K2BMC(2);
    else if (nC <= 4) MRPHY_K2BMC(4);
    else MRPHY_K2BMC(8);
#undef MRPHY_K2BMC
    int e = launch_status();
    if (e) return e;
    if (grf || ggr) {
        hipLaunchKernelGGL((k_bloch_rfgr_bwd_mc_p2<T>),
                           dim3((unsigned)((nT + P2_T - 1) / P2_T), (unsigned)(3 + 2 * nC), (unsigned)N),
                           dim3(P2_T * P2_G), 0, st, (const T*)work, (T*)grf, (T*)ggr, N, nT, a.P,
                           (int)nC);
        e = launch_status();
    }
    return e;
}

inline int check_common(int dtype, int64_t N, int64_t nM, int64_t nT)
{
    if (dtype != MRPHY_F32 && dtype != MRPHY_F64 && dtype != MRPHY_F32_C64 &&
        dtype != MRPHY_F32P && dtype != MRPHY_F32P_C64)
        return MRPHY_EINVAL;
    if (N < 0 || nM < 0 || nT < 0) return MRPHY_EINVAL;
    return 0;
}

#define MRPHY_DISPATCH(dtype, CALL)                          \
    switch (dtype) {                                         \
    case MRPHY_F32:     { using T = float;  using CT = float;  return CALL; } \
    case MRPHY_F64:     { using T = double; using CT = double; return CALL; } \
    case MRPHY_F32_C64: { using T = float;  using CT = double; return CALL; } \
    case MRPHY_F32P:     { using T = float;  using CT = prec_f32; return CALL; } \
    case MRPHY_F32P_C64: { using T = float;  using CT = prec_f64; return CALL; } \
    default: return MRPHY_EINVAL;                            \
    }

template <typename T, typename CT>
int run_beff2ab(const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* A, void* B, void* hist,
                int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    AbArgs<T> a;
    a.Beff = (const T*)Beff; a.A = (T*)A; a.B = (T*)B; a.hist = (T*)hist;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, sizeof(T));      // element alignment is enough (V16::utype)
    const dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
    if (hist) hipLaunchKernelGGL((k_beff2ab<T, CT, TC_FWD<T>, true>), grid, dim3(WAVE), 0, st, a);
    else      hipLaunchKernelGGL((k_beff2ab<T, CT, TC_FWD<T>, false>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

template <typename T, typename CT>
int run_beff2ab_bwd(const void* hist, const void* Beff, Bc g, Bc E1, Bc E2, const void* gA,
                    const void* gB, void* gBeff, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    AbBwdArgs<T> a;
    a.hist = (const T*)hist; a.Beff = (const T*)Beff; a.gA = (const T*)gA; a.gB = (const T*)gB;
    a.gBeff = (T*)gBeff; a.g = g; a.E1 = E1; a.E2 = E2;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, sizeof(T)) && aligned_to(gBeff, sizeof(T));
    const dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
    hipLaunchKernelGGL((k_beff2ab_bwd<T, CT, TC_BWD<T>>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}


}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int mrphy_abi_version(void) { return MRPHY_ABI_VERSION; }

#ifdef MRPHY_DEV_KNOBS
// dev build only: device buffer of 4 x uint64 per workgroup that the line kernels (K1, K1h, K3) fill
// with start / end / HW_ID / blockIdx; `cap` = workgroups it holds; null turns stamping off.
// dev build only: rotate which eighth of a buffer each XCD sweeps (experiment)
int mrphy_dev_set_xcd_shift(int shift)
{
    unsigned v = (unsigned)shift & 7u;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_shift), &v, sizeof(v));
}

int mrphy_dev_set_stamps(void* buf, int64_t cap)
{
    g_dev_stamps = (unsigned long long*)buf;
    g_dev_stamps_cap = buf ? cap : 0;
    return 0;
}
#endif

const char* mrphy_arch(void) { return "gfx950"; }

const char* mrphy_error_string(int code)
{
    switch (code) {
    case 0: return "success";
    case MRPHY_EINVAL: return "mrphy: invalid argument";
    case MRPHY_EALIGN: return "mrphy: pointer not aligned to its element size";
    case MRPHY_ENOSPC: return "mrphy: workspace too small";
    default: return hipGetErrorString((hipError_t)code);
    }
}

int mrphy_rfgr2beff(int dtype, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                    const void* loc, const void* df, int64_t df_sn, int64_t df_sm,
                    const void* gamma, int64_t gamma_sn, int64_t gamma_sm, const void* b1,
                    void* beff, int64_t N, int64_t nM, int64_t nT, int64_t nC, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (dtype != MRPHY_F32 && dtype != MRPHY_F64) return MRPHY_EINVAL;   // K0 has no constant type
    if (nC < 1 || (!b1 && nC != 1)) return MRPHY_EINVAL;
    if (N * nM * nT == 0) return 0;
    if (!rf || !gr || !loc || !beff || (df && !gamma)) return MRPHY_EINVAL;
    const size_t ts = tsize(dtype);
    if (!aligned_to(rf, ts) || !aligned_to(gr, ts) || !aligned_to(loc, ts) ||
        !aligned_to(beff, ts) || (b1 && !aligned_to(b1, ts)))
        return MRPHY_EALIGN;
    const Bc bdf = {df, df_sn, df_sm}, bgam = {gamma, gamma_sn, gamma_sm};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRPHY_F32)
        return run_rfgr2beff<float>(rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, beff, N, nM, nT, nC, st);
    return run_rfgr2beff<double>(rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, beff, N, nM, nT, nC, st);
}

size_t mrphy_rfgr2beff_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t nC)
{
    if (N <= 0 || nM <= 0 || nT <= 0 || nC < 1) return 0;
    // single coil: 3 sums x 3 nT per spin group (k_..._p1v).  2..32 coils: (3 + 2 nC) x nT per spin group
    // -- grad_gr's three rows and a re and an im row per coil -- followed by the packed coefficient rows of
    // the SGPR pass, (2 MC + 4) per spin for the padded coil count MC the launcher picks (the query cannot see
    // whether a b1 map will be passed; mrphy_rfgr2beff_bwd rejects nC >= 2 without one).  More coils: the
    // generic passes, (3 + 2 nC) x nT.  (The dev build also carries round 2's element-per-thread pass:
    // 2 MC sums x 3 nT per spin group.)
    const int cap = bwd_capacity(nC, true);
    if (cap) {
        size_t need = bwd_pack_offset(tsize(dtype), N, nM, nT, nC) +
                      (size_t)(N * nM * (2 * bwd_padded_coils(nC, true) + 4)) * tsize(dtype);
#ifdef MRPHY_DEV_KNOBS
        const size_t old_pass = (size_t)(bwd_spin_groups(nM) * N * (3 * 2 * cap) * nT) * tsize(dtype);
        if (old_pass > need) need = old_pass;
#endif
        return need;
    }
    const int64_t rows = (nC == 1) ? 9 : (3 + 2 * nC);
    return (size_t)(bwd_spin_groups(nM) * N * rows * nT) * tsize(dtype);
}

int mrphy_rfgr2beff_bwd(int dtype, const void* grad_beff, const void* loc, const void* b1,
                        void* grad_rf, void* grad_gr, void* work, size_t work_bytes, int64_t N,
                        int64_t nM, int64_t nT, int64_t nC, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || nC < 1) return MRPHY_EINVAL;
    // as mrphy_rfgr2beff: without a b1 map the field is Bxy = rf of ONE coil (the host sums a
    // multi-coil rf first), so there is no multi-coil gradient to form.  The workspace query
    // relies on this: nC >= 2 implies a map, i.e. the one-pass layout it sizes for.
    if (!b1 && nC != 1) return MRPHY_EINVAL;
    if (N * nT == 0) return 0;
    if (!grad_beff || !loc || !work) return MRPHY_EINVAL;
    if (work_bytes < mrphy_rfgr2beff_bwd_workspace(dtype, N, nM, nT, nC)) return MRPHY_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRPHY_F32)
        return run_rfgr2beff_bwd<float>(grad_beff, loc, b1, grad_rf, grad_gr, work, N, nM, nT, nC, st);
    return run_rfgr2beff_bwd<double>(grad_beff, loc, b1, grad_rf, grad_gr, work, N, nM, nT, nC, st);
}

size_t mrphy_blochsim_hist_bytes(int dtype, int64_t N, int64_t nM, int64_t nT)
{
    if (N <= 0 || nM <= 0 || nT <= 0) return 0;
    return (size_t)hist_elems(N, nM, nT) * tsize(dtype);
}

int mrphy_blochsim_fwd(int dtype, const void* Mi, const void* Beff, const void* g, int64_t g_sn,
                       int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                       int64_t E2_sn, int64_t E2_sm, const void* E1m1, void* Mo, void* Mpre,
                       int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (!Mi || !Mo || !g || (nT > 0 && !Beff)) return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr) || (E1 == nullptr) != (E1m1 == nullptr))
        return MRPHY_EINVAL;                              // both or neither (sims.py:68)
    const size_t ts = tsize(dtype), cs = csize(dtype);
    if (!aligned_to(Mi, ts) || !aligned_to(Mo, ts) || !aligned_to(Beff, ts) ||
        !aligned_to(g, cs) || (E1 && (!aligned_to(E1, cs) || !aligned_to(E2, cs))))
        return MRPHY_EALIGN;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_fwd<T, CT>(Mi, Beff, bg, b1, b2, E1m1, Mo, Mpre, N, nM, nT, st)));
}

int mrphy_blochsim_bwd(int dtype, const void* Mpre, const void* Beff, const void* g, int64_t g_sn,
                       int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                       int64_t E2_sn, int64_t E2_sm, const void* grad_Mo, void* grad_Mi,
                       void* grad_Beff, int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (!g || !grad_Mo || (nT > 0 && (!Beff || !Mpre))) return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr)) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_bwd<T, CT>(Mpre, Beff, bg, b1, b2, grad_Mo, grad_Mi, grad_Beff, nullptr,
                                          N, nM, nT, st)));
}

int mrphy_blochsim_bwd_consts(int dtype, const void* Mpre, const void* Beff, const void* g,
                              int64_t g_sn, int64_t g_sm, const void* E1, int64_t E1_sn,
                              int64_t E1_sm, const void* E2, int64_t E2_sn, int64_t E2_sm,
                              const void* grad_Mo, void* grad_Mi, void* grad_Beff, void* grad_consts,
                              int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (!g || !grad_Mo || !grad_consts || (nT > 0 && (!Beff || !Mpre))) return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr)) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_bwd<T, CT>(Mpre, Beff, bg, b1, b2, grad_Mo, grad_Mi, grad_Beff,
                                          grad_consts, N, nM, nT, st)));
}

int mrphy_blochsim_1step(int dtype, const void* M, const void* b, const void* g, int64_t g_sn,
                         int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm,
                         const void* E2, int64_t E2_sn, int64_t E2_sm, const void* E1m1,
                         void* Mout, int64_t N, int64_t nM, void* stream)
{
    // one step of the same integrator: Beff (N, nM, 1, 3) == b (N, nM, 3)
    return mrphy_blochsim_fwd(dtype, M, b, g, g_sn, g_sm, E1, E1_sn, E1_sm, E2, E2_sn, E2_sm,
                              E1m1, Mout, nullptr, N, nM, 1, stream);
}

int mrphy_blochsim_rfgr_fwd(int dtype, const void* Mi, const void* rf, int64_t rf_sn,
                            const void* gr, int64_t gr_sn, const void* loc, const void* df,
                            int64_t df_sn, int64_t df_sm, const void* gamma, int64_t gamma_sn,
                            int64_t gamma_sm, const void* b1, const void* g, int64_t g_sn,
                            int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm,
                            const void* E2, int64_t E2_sn, int64_t E2_sm, const void* E1m1,
                            void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                            int64_t nT, int64_t nC, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (nC < 1 || (!b1 && nC != 1) || (Mck && (ck_every < 8 || ck_every % 8 != 0)))
        return MRPHY_EINVAL;
    if (N * nM == 0) return 0;
    if (!Mi || !Mo || !loc || !g || (nT > 0 && (!rf || !gr)) || (df && !gamma))
        return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr) || (E1 == nullptr) != (E1m1 == nullptr))
        return MRPHY_EINVAL;
    const Bc bdf = {df, df_sn, df_sm}, bgam = {gamma, gamma_sn, gamma_sm};
    const Bc bg = {g, g_sn, g_sm}, be1 = {E1, E1_sn, E1_sm}, be2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_rfgr_fwd<T, CT>(Mi, rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, bg,
                                               be1, be2, E1m1, Mo, Mck, ck_every, N, nM, nT, nC,
                                               st)));
}

int64_t mrphy_blochsim_rfgr_ck_every(void) { return SEG; }

size_t mrphy_blochsim_rfgr_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT)
{
    if (N <= 0 || nM <= 0 || nT <= 0) return 0;
    return (size_t)(k2b_waves(nM) * N * 5 * nT) * tsize(dtype);
}

int mrphy_blochsim_rfgr_bwd(int dtype, const void* Mck, const void* rf, int64_t rf_sn,
                            const void* gr, int64_t gr_sn, const void* loc, const void* df,
                            int64_t df_sn, int64_t df_sm, const void* gamma, int64_t gamma_sn,
                            int64_t gamma_sm, const void* b1, const void* g, int64_t g_sn,
                            int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm,
                            const void* E2, int64_t E2_sn, int64_t E2_sm, const void* E1m1,
                            const void* grad_Mo, void* grad_Mi, void* grad_rf, void* grad_gr,
                            void* work, size_t work_bytes, int64_t N, int64_t nM, int64_t nT,
                            void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (nT % SEG != 0) return MRPHY_EINVAL;               // whole checkpoint segments only
    if (N * nM * nT == 0) return 0;
    if (!Mck || !rf || !gr || !loc || !g || !grad_Mo || !work || (df && !gamma)) return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr) || (E1 == nullptr) != (E1m1 == nullptr))
        return MRPHY_EINVAL;
    if (work_bytes < mrphy_blochsim_rfgr_bwd_workspace(dtype, N, nM, nT)) return MRPHY_ENOSPC;
    const Bc bdf = {df, df_sn, df_sm}, bgam = {gamma, gamma_sn, gamma_sm};
    const Bc bg = {g, g_sn, g_sm}, be1 = {E1, E1_sn, E1_sm}, be2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_rfgr_bwd<T, CT>(Mck, rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, bg, be1,
                                               be2, E1m1, grad_Mo, grad_Mi, grad_rf, grad_gr, work,
                                               N, nM, nT, st)));
}

int64_t mrphy_blochsim_rfgr_mc_max_coils(void) { return K2B_MAXC; }

size_t mrphy_blochsim_rfgr_mc_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t nC)
{
    if (N <= 0 || nM <= 0 || nT <= 0 || nC <= 0) return 0;
    return (size_t)(k2b_mc_waves(nM) * N * (3 + 2 * nC) * nT) * tsize(dtype);
}

int mrphy_blochsim_rfgr_mc_bwd(int dtype, const void* Mck, const void* rf, int64_t rf_sn,
                               const void* gr, int64_t gr_sn, const void* loc, const void* df,
                               int64_t df_sn, int64_t df_sm, const void* gamma, int64_t gamma_sn,
                               int64_t gamma_sm, const void* b1, const void* g, int64_t g_sn,
                               int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm,
                               const void* E2, int64_t E2_sn, int64_t E2_sm, const void* E1m1,
                               const void* grad_Mo, void* grad_Mi, void* grad_rf, void* grad_gr,
                               void* work, size_t work_bytes, int64_t N, int64_t nM, int64_t nT,
                               int64_t nC, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (nT % SEG != 0 || nC < 1 || nC > K2B_MAXC) return MRPHY_EINVAL;
    if (N * nM * nT == 0) return 0;
    if (!Mck || !rf || !gr || !loc || !b1 || !g || !grad_Mo || !work || (df && !gamma))
        return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr) || (E1 == nullptr) != (E1m1 == nullptr))
        return MRPHY_EINVAL;
    if (work_bytes < mrphy_blochsim_rfgr_mc_bwd_workspace(dtype, N, nM, nT, nC)) return MRPHY_ENOSPC;
    const Bc bdf = {df, df_sn, df_sm}, bgam = {gamma, gamma_sn, gamma_sm};
    const Bc bg = {g, g_sn, g_sm}, be1 = {E1, E1_sn, E1_sm}, be2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_rfgr_mc_bwd<T, CT>(Mck, rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, bg,
                                                  be1, be2, E1m1, grad_Mo, grad_Mi, grad_rf, grad_gr,
                                                  work, N, nM, nT, nC, st)));
}

static int freeprec_launch(int dtype, int dir, const void* Mi, const void* dur, int64_t dur_sn,
                           const void* T1, int64_t T1_sn, int64_t T1_sm, const void* T2,
                           int64_t T2_sn, int64_t T2_sm, const void* df, int64_t df_sn,
                           int64_t df_sm, void* Mo, int64_t N, int64_t nM, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || N < 0 || nM < 0) return MRPHY_EINVAL;
    if (N * nM == 0) return 0;
    if (!Mi || !Mo || !dur || ((T1 == nullptr) != (T2 == nullptr))) return MRPHY_EINVAL;
    FreePrecArgs a;
    a.Mi = Mi; a.Mo = Mo; a.dur = dur; a.dur_sn = dur_sn;
    a.T1 = Bc{T1, T1_sn, T1_sm}; a.T2 = Bc{T2, T2_sn, T2_sm}; a.df = Bc{df, df_sn, df_sm};
    a.rows = N * nM; a.nM = nM;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((a.rows + 255) / 256));
    if (dtype == MRPHY_F32) {
        if (dir > 0) hipLaunchKernelGGL((k_freeprec<float, 1>), grid, dim3(256), 0, st, a);
        else         hipLaunchKernelGGL((k_freeprec<float, -1>), grid, dim3(256), 0, st, a);
    } else {
        if (dir > 0) hipLaunchKernelGGL((k_freeprec<double, 1>), grid, dim3(256), 0, st, a);
        else         hipLaunchKernelGGL((k_freeprec<double, -1>), grid, dim3(256), 0, st, a);
    }
    return launch_status();
}

int mrphy_freeprec_fwd(int dtype, const void* Mi, const void* dur, int64_t dur_sn, const void* T1,
                       int64_t T1_sn, int64_t T1_sm, const void* T2, int64_t T2_sn, int64_t T2_sm,
                       const void* df, int64_t df_sn, int64_t df_sm, void* Mo, int64_t N,
                       int64_t nM, void* stream)
{
    return freeprec_launch(dtype, +1, Mi, dur, dur_sn, T1, T1_sn, T1_sm, T2, T2_sn, T2_sm, df, df_sn,
                           df_sm, Mo, N, nM, stream);
}

int mrphy_freeprec_bwd(int dtype, const void* grad_Mo, const void* dur, int64_t dur_sn,
                       const void* T1, int64_t T1_sn, int64_t T1_sm, const void* T2, int64_t T2_sn,
                       int64_t T2_sm, const void* df, int64_t df_sn, int64_t df_sm, void* grad_Mi,
                       int64_t N, int64_t nM, void* stream)
{
    return freeprec_launch(dtype, -1, grad_Mo, dur, dur_sn, T1, T1_sn, T1_sm, T2, T2_sn, T2_sm, df,
                           df_sn, df_sm, grad_Mi, N, nM, stream);
}

int mrphy_pulse_interp_linear(int dtype, int dir, const void* y, void* out, const void* lo,
                              const void* w, const void* dx, int64_t nch, int64_t nTo, int64_t nTn,
                              void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || nch < 0 || nTo < 0 || nTn < 0)
        return MRPHY_EINVAL;
    if (nch == 0 || (dir > 0 && nTn == 0) || (dir <= 0 && nTo == 0)) return 0;
    if (!y || !out || (nTn > 0 && (!lo || !w || !dx))) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dir > 0) {
        const dim3 grid((unsigned)((nTn + 255) / 256), (unsigned)nch);
        if (nch > 65535) return MRPHY_EINVAL;
        if (dtype == MRPHY_F32)
            hipLaunchKernelGGL((k_interp_lin_fwd<float>), grid, dim3(256), 0, st, (const float*)y,
                               (float*)out, (const int*)lo, (const double*)w, (const double*)dx, nch,
                               nTo, nTn);
        else
            hipLaunchKernelGGL((k_interp_lin_fwd<double>), grid, dim3(256), 0, st, (const double*)y,
                               (double*)out, (const int*)lo, (const double*)w, (const double*)dx, nch,
                               nTo, nTn);
    } else {
        const dim3 grid((unsigned)((nTo + 255) / 256), (unsigned)nch);
        if (nch > 65535) return MRPHY_EINVAL;
        if (dtype == MRPHY_F32)
            hipLaunchKernelGGL((k_interp_lin_bwd<float>), grid, dim3(256), 0, st, (const float*)y,
                               (float*)out, (const int*)lo, (const double*)w, (const double*)dx, nch,
                               nTo, nTn);
        else
            hipLaunchKernelGGL((k_interp_lin_bwd<double>), grid, dim3(256), 0, st, (const double*)y,
                               (double*)out, (const int*)lo, (const double*)w, (const double*)dx, nch,
                               nTo, nTn);
    }
    return launch_status();
}

int mrphy_pulse_interp_select(int dtype, int dir, const void* y, void* out, const void* sel,
                              int64_t nch, int64_t nTo, int64_t nTn, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || nch < 0 || nTo < 0 || nTn < 0)
        return MRPHY_EINVAL;
    if (nch == 0 || (dir > 0 && nTn == 0) || (dir <= 0 && nTo == 0)) return 0;
    if (nch > 65535) return MRPHY_EINVAL;
    if (!y || !out || (nTn > 0 && !sel)) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t npts = dir > 0 ? nTn : nTo;
    const dim3 grid((unsigned)((npts + 255) / 256), (unsigned)nch);
    if (dir > 0) {
        if (dtype == MRPHY_F32)
            hipLaunchKernelGGL((k_interp_sel_fwd<float>), grid, dim3(256), 0, st, (const float*)y,
                               (float*)out, (const int*)sel, nch, nTo, nTn);
        else
            hipLaunchKernelGGL((k_interp_sel_fwd<double>), grid, dim3(256), 0, st, (const double*)y,
                               (double*)out, (const int*)sel, nch, nTo, nTn);
    } else {
        if (dtype == MRPHY_F32)
            hipLaunchKernelGGL((k_interp_sel_bwd<float>), grid, dim3(256), 0, st, (const float*)y,
                               (float*)out, (const int*)sel, nch, nTo, nTn);
        else
            hipLaunchKernelGGL((k_interp_sel_bwd<double>), grid, dim3(256), 0, st, (const double*)y,
                               (double*)out, (const int*)sel, nch, nTo, nTn);
    }
    return launch_status();
}

int mrphy_beff2uphi(int dtype, const void* b, const void* g, int64_t g_sn, int64_t g_sm, void* U,
                    void* Phi, int64_t N, int64_t nM, void* stream)
{
    if (int e = check_common(dtype, N, nM, 0)) return e;
    if (dtype == MRPHY_F32P) dtype = MRPHY_F32;            // no time stepping here: same kernel
    if (dtype == MRPHY_F32P_C64) dtype = MRPHY_F32_C64;
    const int64_t rows = N * nM;
    if (rows == 0) return 0;
    if (!b || !g || !U || !Phi) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 255) / 256));
    switch (dtype) {
    case MRPHY_F32:
        hipLaunchKernelGGL((k_beff2uphi<float, float>), grid, dim3(256), 0, st, (const float*)b,
                           bg, (float*)U, (float*)Phi, rows, nM); break;
    case MRPHY_F64:
        hipLaunchKernelGGL((k_beff2uphi<double, double>), grid, dim3(256), 0, st,
                           (const double*)b, bg, (double*)U, (double*)Phi, rows, nM); break;
    default:
        hipLaunchKernelGGL((k_beff2uphi<float, double>), grid, dim3(256), 0, st, (const float*)b,
                           bg, (float*)U, (float*)Phi, rows, nM); break;
    }
    return launch_status();
}

int mrphy_uphirot(int dtype, const void* U, const void* Phi, const void* Vi, void* Vo,
                  int64_t rows, int64_t nV, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || rows < 0 || nV < 0) return MRPHY_EINVAL;
    if (rows * nV == 0) return 0;
    if (!U || !Phi || !Vi || !Vo || Vi == Vo) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows * nV + 255) / 256));
    if (dtype == MRPHY_F32)
        hipLaunchKernelGGL((k_uphirot<float>), grid, dim3(256), 0, st, (const float*)U,
                           (const float*)Phi, (const float*)Vi, (float*)Vo, rows, nV);
    else
        hipLaunchKernelGGL((k_uphirot<double>), grid, dim3(256), 0, st, (const double*)U,
                           (const double*)Phi, (const double*)Vi, (double*)Vo, rows, nV);
    return launch_status();
}

int mrphy_beff2uphi_bwd(int dtype, const void* b, const void* g, int64_t g_sn, int64_t g_sm,
                        const void* gU, const void* gPhi, void* gb, void* gg, int64_t N, int64_t nM,
                        void* stream)
{
    if (int e = check_common(dtype, N, nM, 0)) return e;
    if (dtype == MRPHY_F32P) dtype = MRPHY_F32;
    if (dtype == MRPHY_F32P_C64) dtype = MRPHY_F32_C64;
    const int64_t rows = N * nM;
    if (rows == 0 || (!gb && !gg)) return 0;
    if (!b || !g) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 255) / 256));
    switch (dtype) {
    case MRPHY_F32:
        hipLaunchKernelGGL((k_beff2uphi_bwd<float, float>), grid, dim3(256), 0, st, (const float*)b,
                           bg, (const float*)gU, (const float*)gPhi, (float*)gb, (float*)gg, rows, nM);
        break;
    case MRPHY_F64:
        hipLaunchKernelGGL((k_beff2uphi_bwd<double, double>), grid, dim3(256), 0, st,
                           (const double*)b, bg, (const double*)gU, (const double*)gPhi, (double*)gb,
                           (double*)gg, rows, nM);
        break;
    default:
        hipLaunchKernelGGL((k_beff2uphi_bwd<float, double>), grid, dim3(256), 0, st, (const float*)b,
                           bg, (const float*)gU, (const float*)gPhi, (float*)gb, (float*)gg, rows, nM);
        break;
    }
    return launch_status();
}

int mrphy_uphirot_bwd(int dtype, const void* U, const void* Phi, const void* Vi, const void* gVo,
                      void* gU, void* gPhi, void* gVi, int64_t rows, int64_t nV, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || rows < 0 || nV < 0) return MRPHY_EINVAL;
    if (rows * nV == 0 || (!gU && !gPhi && !gVi)) return 0;
    if (!U || !Phi || !Vi || !gVo || gVi == gVo) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 255) / 256));
    if (dtype == MRPHY_F32)
        hipLaunchKernelGGL((k_uphirot_bwd<float>), grid, dim3(256), 0, st, (const float*)U,
                           (const float*)Phi, (const float*)Vi, (const float*)gVo, (float*)gU,
                           (float*)gPhi, (float*)gVi, rows, nV);
    else
        hipLaunchKernelGGL((k_uphirot_bwd<double>), grid, dim3(256), 0, st, (const double*)U,
                           (const double*)Phi, (const double*)Vi, (const double*)gVo, (double*)gU,
                           (double*)gPhi, (double*)gVi, rows, nV);
    return launch_status();
}

int mrphy_mask_extract(int elem_bytes, const void* v, const int32_t* idx, void* out_, int64_t N,
                       int64_t nV, int64_t nM, int64_t K, void* stream)
{
    if ((elem_bytes != 4 && elem_bytes != 8) || N < 0 || nV < 0 || nM < 0 || K < 0 || nM > nV ||
        nV > INT32_MAX || N > 65535)
        return MRPHY_EINVAL;
    if (N * nM * K == 0) return 0;
    if (!v || !idx || !out_ || v == out_) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((nM * K + 255) / 256), (unsigned)N);
    if (elem_bytes == 4)
        hipLaunchKernelGGL((k_mask_extract<uint32_t>), grid, dim3(256), 0, st, (const uint32_t*)v,
                           idx, (uint32_t*)out_, nV, nM, K);
    else
        hipLaunchKernelGGL((k_mask_extract<uint64_t>), grid, dim3(256), 0, st, (const uint64_t*)v,
                           idx, (uint64_t*)out_, nV, nM, K);
    return launch_status();
}

int mrphy_mask_embed(int elem_bytes, const void* v_, const int32_t* inv, void* out, int64_t N,
                     int64_t nV, int64_t nM, int64_t K, int fill, uint64_t fillbits, void* stream)
{
    if ((elem_bytes != 4 && elem_bytes != 8) || N < 0 || nV < 0 || nM < 0 || K < 0 || nM > nV ||
        nV > INT32_MAX || N > 65535 || (fill != 0 && fill != 1))
        return MRPHY_EINVAL;
    if (N * nV * K == 0) return 0;
    if (!inv || !out || (nM > 0 && !v_) || v_ == out) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((nV * K + 255) / 256), (unsigned)N);
    if (elem_bytes == 4)
        hipLaunchKernelGGL((k_mask_embed<uint32_t>), grid, dim3(256), 0, st, (const uint32_t*)v_,
                           inv, (uint32_t*)out, nV, nM, K, fill, (uint32_t)fillbits);
    else
        hipLaunchKernelGGL((k_mask_embed<uint64_t>), grid, dim3(256), 0, st, (const uint64_t*)v_,
                           inv, (uint64_t*)out, nV, nM, K, fill, fillbits);
    return launch_status();
}

int mrphy_cube_loc(int dtype, const int32_t* idx, const void* fov, const void* ofst, void* loc_,
                   int64_t N, int64_t nM, int64_t nx, int64_t ny, int64_t nz, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || N < 0 || nM < 0 || nx < 1 || ny < 1 ||
        nz < 1 || nx * ny * nz > INT32_MAX || nM > nx * ny * nz || N > 65535)
        return MRPHY_EINVAL;
    if (N * nM == 0) return 0;
    if (!idx || !fov || !ofst || !loc_) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((nM + 255) / 256), (unsigned)N);
    if (dtype == MRPHY_F32)
        hipLaunchKernelGGL((k_cube_loc<float>), grid, dim3(256), 0, st, idx, (const float*)fov,
                           (const float*)ofst, (float*)loc_, nM, (int)nx, (int)ny, (int)nz);
    else
        hipLaunchKernelGGL((k_cube_loc<double>), grid, dim3(256), 0, st, idx, (const double*)fov,
                           (const double*)ofst, (double*)loc_, nM, (int)nx, (int)ny, (int)nz);
    return launch_status();
}

int mrphy_debug_xcc_map(int32_t* out, int64_t nblocks, void* stream)
{
    if (nblocks < 0 || nblocks > 0x7fffffff) return MRPHY_EINVAL;
    if (nblocks == 0) return 0;
    if (!out) return MRPHY_EINVAL;
    hipLaunchKernelGGL(k_xcc_map, dim3((unsigned)nblocks), dim3(64), 0, (hipStream_t)stream, out, nblocks);
    return launch_status();
}

int mrphy_beff2ab(int dtype, const void* Beff,
                  const void* g, int64_t g_sn, int64_t g_sm,
                  const void* E1, int64_t E1_sn, int64_t E1_sm,
                  const void* E2, int64_t E2_sn, int64_t E2_sm,
                  const void* E1m1, void* A, void* B,
                  int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (!A || !B || !g || !E1 || !E2 || !E1m1 || (nT > 0 && !Beff)) return MRPHY_EINVAL;
    const size_t ts = tsize(dtype), cs = csize(dtype);
    if (!aligned_to(A, ts) || !aligned_to(B, ts) || !aligned_to(Beff, ts) || !aligned_to(g, cs) ||
        !aligned_to(E1, cs) || !aligned_to(E2, cs) || !aligned_to(E1m1, cs))
        return MRPHY_EALIGN;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_beff2ab<T, CT>(Beff, bg, b1, b2, E1m1, A, B, nullptr, N, nM, nT, st)));
}

size_t mrphy_beff2ab_hist_bytes(int dtype, int64_t N, int64_t nM, int64_t nT)
{
    if (N <= 0 || nM <= 0 || nT <= 0) return 0;
    return (size_t)(((N * nM + WAVE - 1) / WAVE) * nT * AB_HIST_STEP) * tsize(dtype);
}

int mrphy_beff2ab_save(int dtype, const void* Beff,
                       const void* g, int64_t g_sn, int64_t g_sm,
                       const void* E1, int64_t E1_sn, int64_t E1_sm,
                       const void* E2, int64_t E2_sn, int64_t E2_sm,
                       const void* E1m1, void* A, void* B, void* hist,
                       int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (!A || !B || !g || !E1 || !E2 || !E1m1 || (nT > 0 && (!Beff || !hist))) return MRPHY_EINVAL;
    const size_t ts = tsize(dtype), cs = csize(dtype);
    if (!aligned_to(A, ts) || !aligned_to(B, ts) || !aligned_to(Beff, ts) || !aligned_to(hist, ts) ||
        !aligned_to(g, cs) || !aligned_to(E1, cs) || !aligned_to(E2, cs) || !aligned_to(E1m1, cs))
        return MRPHY_EALIGN;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_beff2ab<T, CT>(Beff, bg, b1, b2, E1m1, A, B, hist, N, nM, nT, st)));
}

int mrphy_beff2ab_bwd(int dtype, const void* hist, const void* Beff,
                      const void* g, int64_t g_sn, int64_t g_sm,
                      const void* E1, int64_t E1_sn, int64_t E1_sm,
                      const void* E2, int64_t E2_sn, int64_t E2_sm,
                      const void* grad_A, const void* grad_B, void* grad_Beff,
                      int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM * nT == 0) return 0;
    if (!hist || !Beff || !g || !E1 || !E2 || !grad_Beff) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_beff2ab_bwd<T, CT>(hist, Beff, bg, b1, b2, grad_A, grad_B, grad_Beff,
                                                  N, nM, nT, st)));
}

int mrphy_blochsim_ab(int dtype, const void* M, const void* A, const void* B, void* Mo,
                      int64_t rows, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || rows < 0) return MRPHY_EINVAL;
    if (rows == 0) return 0;
    if (!M || !A || !B || !Mo || M == Mo) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 255) / 256));
    if (dtype == MRPHY_F32)
        hipLaunchKernelGGL((k_ab_apply<float>), grid, dim3(256), 0, st, (const float*)M,
                           (const float*)A, (const float*)B, (float*)Mo, rows);
    else
        hipLaunchKernelGGL((k_ab_apply<double>), grid, dim3(256), 0, st, (const double*)M,
                           (const double*)A, (const double*)B, (double*)Mo, rows);
    return launch_status();
}

int mrphy_blochsim_ab_bwd(int dtype, const void* M, const void* A, const void* gMo, void* gM,
                          void* gA, int64_t rows, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || rows < 0) return MRPHY_EINVAL;
    if (rows == 0 || (!gM && !gA)) return 0;
    if (!gMo || (gM && !A) || (gA && !M)) return MRPHY_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 255) / 256));
    if (dtype == MRPHY_F32)
        hipLaunchKernelGGL((k_ab_apply_bwd<float>), grid, dim3(256), 0, st, (const float*)M,
                           (const float*)A, (const float*)gMo, (float*)gM, (float*)gA, rows);
    else
        hipLaunchKernelGGL((k_ab_apply_bwd<double>), grid, dim3(256), 0, st, (const double*)M,
                           (const double*)A, (const double*)gMo, (double*)gM, (double*)gA, rows);
    return launch_status();
}

}  // extern "C"
