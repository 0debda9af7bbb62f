// tu_rfgr2beff_bwd.hip -- adjoint of K0: launcher of mrphy_rfgr2beff_bwd
#include "host_common.hpp"

namespace {
#include "k_rfgr2beff_bwd.hpp"
}  // namespace

namespace mrphy_i {

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
        // one block of coils c0 .. c0 + nCb - 1 (the whole pulse's if nC <= 32): pack, then the main pass
        auto launch_sgpr = [&](auto mc_tag, int64_t c0, int64_t nCb) -> int {
            constexpr int MC = decltype(mc_tag)::value;
            T* pk = reinterpret_cast<T*>(static_cast<char*>(work) + bwd_pack_offset(sizeof(T), N, nM, nT, nC));
            PackArgs<T> pa;
            pa.b1 = (const T*)b1; pa.loc = (const T*)loc; pa.pk = pk; pa.rows = N * nM; pa.nC = nCb; pa.MC = MC;
            pa.c0 = c0; pa.nCtot = nC;
            const int64_t words = N * nM * (2 * MC + 4);
            if ((words + 255) / 256 > 2147483647) return MRPHY_EINVAL;
            hipLaunchKernelGGL((k_pack_coefs<T>), dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, pa);
            int e = launch_status();
            if (e) return e;
            BeffBwdPkArgs<T> b;
            b.gB = a.gB; b.pk = pk; b.work = a.work; b.N = N; b.nM = nM; b.nT = nT; b.nC = nCb;
            b.spins_per_group = a.spins_per_group;
            b.K = (int)(3 + 2 * nC); b.rowR = (int)(3 + c0); b.rowI = (int)(3 + nC + c0);
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
            return launch_status();
        };
        auto launch_block = [&](int64_t c0, int64_t nCb) -> int {
            switch (bwd_padded_coils(nCb < 2 ? 2 : nCb, true)) {
            case 4:  return launch_sgpr(integral_constant<int, 4>{}, c0, nCb);
            case 8:  return launch_sgpr(integral_constant<int, 8>{}, c0, nCb);
            case 12: return launch_sgpr(integral_constant<int, 12>{}, c0, nCb);
            case 16: return launch_sgpr(integral_constant<int, 16>{}, c0, nCb);
            case 24: return launch_sgpr(integral_constant<int, 24>{}, c0, nCb);
            default: return launch_sgpr(integral_constant<int, 32>{}, c0, nCb);
            }
        };
        // coil counts above 32 (round 4): blocks of 32, each its own pass over grad_Beff -- the time grows by one
        // 32-coil pass per block, i.e. linearly, where the generic passes read grad_Beff nC + 1 times
        for (int64_t c0 = 0; c0 < nC; c0 += BWD_MAXC) {
            if (int e = launch_block(c0, (nC - c0 < BWD_MAXC) ? nC - c0 : (int64_t)BWD_MAXC)) return e;
        }
        hipLaunchKernelGGL((k_rfgr2beff_bwd_p2<T>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                           dim3(256), 0, st, a);
        return launch_status();
    }
    hipLaunchKernelGGL((k_rfgr2beff_bwd_p1<T>), dim3(tx, (unsigned)a.nSG, (unsigned)(N * (nC + 1))),
                       dim3(256), 0, st, a);
    int e = launch_status();
    if (e) return e;
    hipLaunchKernelGGL((k_rfgr2beff_bwd_p2<T>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                       dim3(256), 0, st, a);
    return launch_status();
}

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_rfgr2beff_bwd<T_>(const void* gB, const void* loc, const void* b1, void* grf, void* ggr, void* work, int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st);
MRPHY_FOR_DATA_TYPES(MRPHY_INST)
#undef MRPHY_INST
