// tu_fused_fwd1.hip -- K2, one transmit coil, float: launcher called by run_rfgr_fwd (tu_fused_fwd.hip)
// A unit of its own because it is compiled with `-mllvm -amdgpu-sched-strategy=max-ilp` (_lib.py: UNIT_FLAGS):
// the ILP-first schedule of the step loop is 2-3 % faster for these builds (128^3 x 1024: 2.35 -> 2.28 ms, 128^3 x 4096:
// 9.5 -> 9.3 ms same box, profiles/r04_k2_ilp_ab.json) at 102-105 instead of 79-95 VGPRs, and costs the multi-coil
// builds (8 coils: 118 -> 199 VGPRs, +4 % time) and the fp64 ones what it gives these.
#include "host_common.hpp"

namespace {
#include "k_fused_fwd.hpp"
}  // namespace

namespace mrphy_i {

template <typename T, typename CT>
int run_rfgr_fwd1(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                  const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                  const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                  int64_t nT, hipStream_t st)
{
    if constexpr (sizeof(T) != 4) {
        return MRPHY_EINVAL;                             // fp64 stays in tu_fused_fwd.hip
    } else {
        FusedArgs<T> a;
        a.Mi = (const T*)Mi; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
        a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
        a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.Mo = (T*)Mo; a.Mck = (T*)Mck;
        a.ck_every = ck_every > 0 ? ck_every : 1;
        a.N = N; a.nM = nM; a.nT = nT; a.nC = 1;
        if (N * nM == 0) return 0;
        if (N > 65535) return MRPHY_EINVAL;
        const int64_t tiles = (nM + WAVE - 1) / WAVE;
        const dim3 grid((unsigned)tiles, (unsigned)N);
#ifdef MRPHY_DEV_KNOBS
        a.stamps = tiles * N <= mrphy_i::g_dev_stamps_cap ? mrphy_i::g_dev_stamps : nullptr;
        a.prio_rot = prio_rot(); a.prio_shift = env_int("MRPHY_PRIO_SHIFT", 3);
#endif
#define MRPHY_K2(CK_, RX_, HB_) \
    hipLaunchKernelGGL((k_bloch_rfgr_fwd<T, CT, 1, CK_, RX_, HB_>), grid, dim3(WAVE), 0, st, a)
#define MRPHY_K2H(HB_)                                                                  \
    do {                                                                                \
        if (ck) { if (rx) MRPHY_K2(true, true, HB_); else MRPHY_K2(true, false, HB_); }   \
        else    { if (rx) MRPHY_K2(false, true, HB_); else MRPHY_K2(false, false, HB_); } \
    } while (0)
        const bool ck = (Mck != nullptr), rx = (E1.p != nullptr);
        if (b1) MRPHY_K2H(true);
        else    MRPHY_K2H(false);                        // no b1 map: Bxy = rf, no complex product
#undef MRPHY_K2H
#undef MRPHY_K2
        return launch_status();
    }
}

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_rfgr_fwd1<T_, CT_>(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM, int64_t nT, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
