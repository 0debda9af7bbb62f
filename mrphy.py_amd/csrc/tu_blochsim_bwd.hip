// tu_blochsim_bwd.hip -- K3: launcher of mrphy_blochsim_bwd / _bwd_consts (line-granular and chunked adjoint kernels)
#include "host_common.hpp"

namespace {
#include "k_blochsim_bwd.hpp"
}  // namespace

namespace mrphy_i {

template <typename T, typename CT>
int run_bwd(HistParts hist, const void* Beff, Bc g, Bc E1, Bc E2, const void* gMo, void* gMi,
            void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    BwdArgs<T> a;
    a.hist = hist; a.Beff = (const T*)Beff; a.gMo = (const T*)gMo;
    a.gMi = (T*)gMi; a.gBeff = (T*)gBeff; a.gC = (T*)gC;
    a.g = g; a.E1 = E1; a.E2 = E2;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, sizeof(T)) && (!gBeff || aligned_to(gBeff, sizeof(T)));
    a.per_xcd = 0;
    if (a.rows == 0) return 0;
    dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
#ifdef MRPHY_DEV_KNOBS
    a.stamps = (int64_t)grid.x <= mrphy_i::g_dev_stamps_cap ? mrphy_i::g_dev_stamps : nullptr;
    a.prio_rot = prio_rot(); a.prio_shift = 0;
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
#define MRPHY_LBP(OCC_, PIN_)                                                                    \
    do {                                                                                         \
        if (E1.p) hipLaunchKernelGGL((k_bloch_bwd_lines<CT, true, OCC_, true, PIN_>), grid,       \
                                     dim3(WAVE), lds_pad(), st, a);                              \
        else      hipLaunchKernelGGL((k_bloch_bwd_lines<CT, false, OCC_, true, PIN_>), grid,      \
                                     dim3(WAVE), lds_pad(), st, a);                              \
    } while (0)
#define MRPHY_LB(OCC_) MRPHY_LBP(OCC_, false)
            // same-box A/B at 128^3 x 1024 (ms), round 2: history fetched in-batch 13.28 | one batch
            // ahead: 2 waves/SIMD 12.83, 3 waves/SIMD 13.04 with 36 B/lane of spills; without forming
            // w, v in the adjoint step the 3-wave build has 134-136 VGPRs and no spills: 12.6-12.8, the
            // default (round 3, 64^3 x 2048: 2 | 3 | 4 waves 3.33 | 3.36 | 3.32 ms: no occupancy effect)
#ifdef MRPHY_DEV_KNOBS
            if (occ == 2) { MRPHY_LB(2); return launch_status(); }
            if (occ == 4) { MRPHY_LB(4); return launch_status(); }
            if (occ == 13) { MRPHY_LBP(3, true); return launch_status(); }
            if (occ == 14) { MRPHY_LBP(4, true); return launch_status(); }
#endif
            (void)occ;
            MRPHY_LB(3);
#undef MRPHY_LB
#undef MRPHY_LBP
            return launch_status();
        }
    }
    if constexpr (sizeof(T) == 8) {
        // fp64: the line-granular adjoint where the shape allows it (round 4; the chunked fp64 adjoint needs
        // 430-456 VGPRs = one wave per SIMD)
        if (lines_shape_ok_f64(Beff, nT) && (!gBeff || aligned_to(gBeff, 128)) && fwd_variant() != 16) {
            if (xcd_sweep()) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; }
            if (E1.p) hipLaunchKernelGGL((k_bloch_bwd_lines_f64<CT, true, 2, true, true>), grid, dim3(WAVE), lds_pad(), st, a);
            else      hipLaunchKernelGGL((k_bloch_bwd_lines_f64<CT, false, 2, true, true>), grid, dim3(WAVE), lds_pad(), st, a);
            return launch_status();
        }
    }
    hipLaunchKernelGGL((k_bloch_bwd<T, CT, TC_BWD<T>, false>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_bwd<T_, CT_>(HistParts hist, const void* Beff, Bc g, Bc E1, Bc E2, const void* gMo, void* gMi, void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
