// tu_blochsim_fwd.hip -- K1: launcher of mrphy_blochsim_fwd / _1step (line-granular and chunked forward kernels)
#include "host_common.hpp"

namespace {
#include "k_blochsim_fwd.hpp"
}  // namespace

namespace mrphy_i {

template <typename T, typename CT>
int run_fwd(const void* Mi, const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo,
            HistParts hist, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    FwdArgs<T> a;
    a.Mi = (const T*)Mi; a.Beff = (const T*)Beff; a.Mo = (T*)Mo; a.hist = hist;
    const bool Mpre = hist.n_parts > 0;            // the history-saving builds (K1h)
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    // vector path of the chunked kernel (16-B global accesses need element alignment only)
    a.vec_ok = aligned_to(Beff, sizeof(T));      // element alignment is enough (V16::utype)
    a.per_xcd = 0; a.xcd_rev = 0;
    if (a.rows == 0) return 0;
    dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
#ifdef MRPHY_DEV_KNOBS
    a.stamps = (int64_t)grid.x <= mrphy_i::g_dev_stamps_cap ? mrphy_i::g_dev_stamps : nullptr;
    a.prio_rot = prio_rot(); a.prio_shift = 0;
#endif
    if constexpr (sizeof(T) == 4) {
        const int v = fwd_variant();
        if (lines_shape_ok(Beff, nT) && v != 16 && v != 32) {
            // XCD-contiguous tile order pays where the kernel writes (history: 10.07 -> 8.75 ms at 128^3 x 1024).
            // For the read-only forward: plain order.  (Round 4, first half: behind a K0 that wrote Beff with nt stores,
            // K1 in plain order ran at 0.59 / 0.64 of HBM peak at 3.2 / 12.9 GB -- K0's dirty tail in the 256-MB
            // memory-side cache -- and the XCD-contiguous order recovered 0.73 / 0.74.  Second half: K0 now writes with
            // `sc1 nt` stores that leave nothing there, and behind THAT the plain order is the better one again: 0.78 /
            // 0.82 against 0.79 / 0.79; profiles/r04_k0_store_policy.json.  Dev knob MRPHY_K1_XCD keeps the other order.)
            const int k1x = k1_xcd(0);
            if (xcd_sweep() && (Mpre || k1x)) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; a.xcd_rev = !Mpre && k1x == 2; }
            // development knob MRPHY_FWD_VARIANT = OCC*100 + SPLIT*10 + NT selects a build.
            // measured on MI355X, 128^3 x 4096, no history (ms): 320 16.88 | 321 15.82 |
            // 330 17.14 | 331 15.72
#define MRPHY_LP(OCC_, SP_, NT_, SV_, PIN_)                                                      \
    do {                                                                                         \
        if (E1.p) hipLaunchKernelGGL((k_bloch_fwd_lines<CT, true, OCC_, SP_, NT_, SV_, PIN_>), grid, \
                                     dim3(WAVE), lds_pad(), st, a);                              \
        else      hipLaunchKernelGGL((k_bloch_fwd_lines<CT, false, OCC_, SP_, NT_, SV_, PIN_>), grid, \
                                     dim3(WAVE), lds_pad(), st, a);                              \
    } while (0)
#define MRPHY_L(OCC_, SP_, NT_, SV_) MRPHY_LP(OCC_, SP_, NT_, SV_, false)
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
                case 330: MRPHY_L(3, 3, false, false); return launch_status();
                case 321: MRPHY_L(3, 2, true, false); return launch_status();
                case 331: MRPHY_L(3, 3, true, false); return launch_status();
                case 341: MRPHY_L(3, 4, true, false); return launch_status();
                case 441: MRPHY_L(4, 4, true, false); return launch_status();
                case 1321: MRPHY_LP(3, 2, true, false, true); return launch_status();
                case 1331: MRPHY_LP(3, 3, true, false, true); return launch_status();
                case 1341: MRPHY_LP(3, 4, true, false, true); return launch_status();
                default: break;
                }
#endif
                // Round 4: pin_state (bloch_math.hpp) after every batch keeps the compiler from sinking the
                // rot_apply chains below the next batches' cold-path guards: 5-/6-step batches then need 102
                // VGPRs (3-/4-step: 90; unpinned: 146-164).  Same bits.  In plain tile order at 128^3 x 4096 the
                // pinned 5-/6-step build is the fastest precise build (15.83 vs 16.04 ms, 0.815 vs 0.804 of HBM
                // peak; isolated A/B 0.807-0.813 vs 0.800); the fast step and the XCD-contiguous order of
                // the smaller grids run best on the unpinned 3-/4-step schedule (profiles/r04_k1_pin_ab.json,
                // r04_k0k1_step_ab.json).
                if (CTr<CT>::precise && !a.per_xcd) MRPHY_LP(3, 2, true, false, true);
                else                                MRPHY_L(3, 3, true, false);
            }
#undef MRPHY_L
#undef MRPHY_LP
            return launch_status();
        }
    }
    if constexpr (sizeof(T) == 8) {
        // fp64 (the reference's own test precision): the line-granular kernel where the shape allows it
        // (round 4: the chunked kernel reads 1.22 x the algorithmic bytes)
        if (lines_shape_ok_f64(Beff, nT) && fwd_variant() != 16) {
            if (xcd_sweep() && (Mpre || k1_xcd(0))) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; }
#define MRPHY_L64(SV_)                                                                            \
    do {                                                                                          \
        if (E1.p) hipLaunchKernelGGL((k_bloch_fwd_lines_f64<CT, true, 3, true, SV_, true>), grid, \
                                     dim3(WAVE), lds_pad(), st, a);                               \
        else      hipLaunchKernelGGL((k_bloch_fwd_lines_f64<CT, false, 3, true, SV_, true>), grid, \
                                     dim3(WAVE), lds_pad(), st, a);                               \
    } while (0)
            if (Mpre) MRPHY_L64(true); else MRPHY_L64(false);
#undef MRPHY_L64
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

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_fwd<T_, CT_>(const void* Mi, const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo, HistParts hist, int64_t N, int64_t nM, int64_t nT, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
