// tu_fused_mc_bwd.hip -- K2b, 2..8 transmit coils: launcher of mrphy_blochsim_rfgr_mc_bwd
#include "host_common.hpp"

namespace {
#include "k_fused_mc_bwd.hpp"
}  // namespace

namespace mrphy_i {

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

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_rfgr_mc_bwd<T_, CT_>(const void* Mck, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2, const void* E1m1, const void* gMo, void* gMi, void* grf, void* ggr, void* work, int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
