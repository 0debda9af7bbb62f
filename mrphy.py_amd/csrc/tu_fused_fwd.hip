// tu_fused_fwd.hip -- K2: launcher of mrphy_blochsim_rfgr_fwd
#include "host_common.hpp"

namespace {
#include "k_fused_fwd.hpp"
}  // namespace

namespace mrphy_i {

template <typename T, typename CT>
int run_rfgr_fwd(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                 int64_t nT, int64_t nC, hipStream_t st)
{
    if constexpr (sizeof(T) == 4) {
        if (nC == 1)
            return run_rfgr_fwd1<T, CT>(Mi, rf, rf_sn, gr, gr_sn, loc, df, gam, b1, g, E1, E2, E1m1, Mo, Mck, ck_every,
                                        N, nM, nT, st);
    }
    FusedArgs<T> a;
    a.Mi = (const T*)Mi; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
    a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.Mo = (T*)Mo; a.Mck = (T*)Mck;
    a.ck_every = ck_every > 0 ? ck_every : 1;
    a.N = N; a.nM = nM; a.nT = nT; a.nC = nC;
    if (N * nM == 0) return 0;
    if (N > 65535) return MRPHY_EINVAL;
    const int64_t tiles = (nM + WAVE - 1) / WAVE;
    const dim3 grid((unsigned)tiles, (unsigned)N);
#ifdef MRPHY_DEV_KNOBS
    a.stamps = tiles * N <= mrphy_i::g_dev_stamps_cap ? mrphy_i::g_dev_stamps : nullptr;
    a.prio_rot = prio_rot(); a.prio_shift = env_int("MRPHY_PRIO_SHIFT", 3);
#endif
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
    // (one coil in float: tu_fused_fwd1.hip, above)
    if (nC == 1 && b1) { if constexpr (sizeof(T) == 8) MRPHY_K2C(1); }
    else if (nC == 1) { if constexpr (sizeof(T) == 8) MRPHY_K2H(1, false); }   // no b1 map: Bxy = rf, no complex product
    else if (nC <= 2 && b1) MRPHY_K2C(2);                // (round 3: 2 coils no longer pay for 8)
    else if (nC <= 4 && b1) MRPHY_K2C(4);
    else if (nC <= 8 && b1) MRPHY_K2C(8);
    else if (sizeof(T) == 4 && nC <= 16 && b1) { if constexpr (sizeof(T) == 4) MRPHY_K2C(16); }
    else if (sizeof(T) == 4 && nC <= 32 && b1) { if constexpr (sizeof(T) == 4) MRPHY_K2C(32); }
    else if (sizeof(T) == 4 && nC <= 40 && b1) { if constexpr (sizeof(T) == 4) MRPHY_K2C(40); }   // (round 4: no cliff at 33)
    else if (sizeof(T) == 4 && nC <= 48 && b1) { if constexpr (sizeof(T) == 4) MRPHY_K2C(48); }
    else if (sizeof(T) == 4 && nC <= K2_MAXC && b1) { if constexpr (sizeof(T) == 4) MRPHY_K2C(64); }
    // (fp64 with more than 8 coils: the 16- / 32-coil register builds would need 128-700 spilled VGPRs in
    // double precision; the host routes those to rfgr2beff + blochsim, and a direct caller gets the generic build)
    else MRPHY_K2C(0);
#undef MRPHY_K2C
#undef MRPHY_K2H
#undef MRPHY_K2
    return launch_status();
}

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_rfgr_fwd<T_, CT_>(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
