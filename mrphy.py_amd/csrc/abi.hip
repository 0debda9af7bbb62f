// abi.hip -- the C ABI of libmrphy_hip.so (include/mrphy_hip.h): argument validation and dispatch on the
// dtype code to the launchers of the tu_*.hip units (internal.hpp).  No kernel is defined here.
#include "host_common.hpp"

#ifdef MRPHY_DEV_KNOBS
namespace mrphy_i {
unsigned long long* g_dev_stamps = nullptr;        // 4 x uint64 per workgroup, or null
int64_t g_dev_stamps_cap = 0;                      // workgroups the buffer holds
}
#endif

using namespace mrphy_i;

namespace {
#define MRPHY_DISPATCH(dtype, CALL)                          \
    switch (dtype) {                                         \
    case MRPHY_F32:     { using T = float;  using CT = float;  return CALL; } \
    case MRPHY_F64:     { using T = double; using CT = double; return CALL; } \
    case MRPHY_F32_C64: { using T = float;  using CT = double; return CALL; } \
    case MRPHY_F32P:     { using T = float;  using CT = prec_f32; return CALL; } \
    case MRPHY_F32P_C64: { using T = float;  using CT = prec_f64; return CALL; } \
    default: return MRPHY_EINVAL;                            \
    }

// the host's table of part pointers -> the by-value descriptor of the kernels (geom.hpp); MRPHY_E* or 0
int make_hist_parts(int dtype, void* const* parts, int64_t n_parts, int layout, int64_t N, int64_t nM, HistParts& h)
{
    h = HistParts{};
    if (n_parts < 0 || n_parts > HIST_MAX_PARTS) return MRPHY_EINVAL;
    if (layout != MRPHY_HIST_BLOCKED && layout != MRPHY_HIST_INTERLEAVED) return MRPHY_EINVAL;
    if (n_parts == 0 || !parts) return 0;
    for (int64_t i = 0; i < n_parts; ++i) {
        if (!parts[i]) return MRPHY_EINVAL;
        if (!aligned_to(parts[i], tsize(dtype))) return MRPHY_EALIGN;
        h.p[i] = parts[i];
    }
    h.n_parts = (int32_t)n_parts;
    h.interleaved = layout == MRPHY_HIST_INTERLEAVED;
    h.tiles_per_part = (uint32_t)hist_tiles_per_part(N, nM, n_parts);
    return 0;
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
    return mrphy_rfgr2beff_st(dtype, rf, rf_sn, gr, gr_sn, loc, df, df_sn, df_sm, gamma, gamma_sn, gamma_sm, b1, beff,
                              N, nM, nT, nC, MRPHY_STORE_AUTO, stream);
}

int mrphy_rfgr2beff_st(int dtype, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                       const void* loc, const void* df, int64_t df_sn, int64_t df_sm,
                       const void* gamma, int64_t gamma_sn, int64_t gamma_sm, const void* b1,
                       void* beff, int64_t N, int64_t nM, int64_t nT, int64_t nC, int store_policy, void* stream)
{
    if (store_policy < MRPHY_STORE_AUTO || store_policy > MRPHY_STORE_SC1NT) return MRPHY_EINVAL;
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
        return run_rfgr2beff<float>(rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, beff, N, nM, nT, nC, store_policy, st);
    return run_rfgr2beff<double>(rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, beff, N, nM, nT, nC, store_policy, st);
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

size_t mrphy_blochsim_hist_part_bytes(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t n_parts)
{
    if (N <= 0 || nM <= 0 || nT <= 0 || n_parts < 1 || n_parts > HIST_MAX_PARTS) return 0;
    return (size_t)(hist_tiles_per_part(N, nM, n_parts) * nT * HIST_STEP) * tsize(dtype);
}

int mrphy_blochsim_fwd_parts(int dtype, const void* Mi, const void* Beff, const void* g, int64_t g_sn,
                             int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                             int64_t E2_sn, int64_t E2_sm, const void* E1m1, void* Mo,
                             void* const* hist_parts, int64_t n_parts, int layout,
                             int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    HistParts hist;
    if (int e = make_hist_parts(dtype, hist_parts, n_parts, layout, N, nM, hist)) return e;
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
    MRPHY_DISPATCH(dtype, (run_fwd<T, CT>(Mi, Beff, bg, b1, b2, E1m1, Mo, hist, N, nM, nT, st)));
}
int mrphy_blochsim_fwd(int dtype, const void* Mi, const void* Beff, const void* g, int64_t g_sn,
                       int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                       int64_t E2_sn, int64_t E2_sm, const void* E1m1, void* Mo, void* Mpre,
                       int64_t N, int64_t nM, int64_t nT, void* stream)
{
    void* parts[1] = {Mpre};
    return mrphy_blochsim_fwd_parts(dtype, Mi, Beff, g, g_sn, g_sm, E1, E1_sn, E1_sm, E2, E2_sn, E2_sm, E1m1, Mo,
                                    Mpre ? parts : nullptr, Mpre ? 1 : 0, MRPHY_HIST_BLOCKED, N, nM, nT, stream);
}

int mrphy_blochsim_bwd_parts(int dtype, const void* const* hist_parts, int64_t n_parts, int layout,
                             const void* Beff, const void* g, int64_t g_sn, int64_t g_sm,
                             const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2, int64_t E2_sn,
                             int64_t E2_sm, const void* grad_Mo, void* grad_Mi, void* grad_Beff,
                             void* grad_consts, int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    HistParts hist;
    if (int e = make_hist_parts(dtype, const_cast<void* const*>(reinterpret_cast<const void* const*>(hist_parts)),
                                n_parts, layout, N, nM, hist)) return e;
    if (N * nM == 0) return 0;
    if (!g || !grad_Mo || (nT > 0 && (!Beff || hist.n_parts == 0))) return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr)) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_bwd<T, CT>(hist, Beff, bg, b1, b2, grad_Mo, grad_Mi, grad_Beff, grad_consts,
                                          N, nM, nT, st)));
}

int mrphy_blochsim_bwd(int dtype, const void* Mpre, const void* Beff, const void* g, int64_t g_sn,
                       int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                       int64_t E2_sn, int64_t E2_sm, const void* grad_Mo, void* grad_Mi,
                       void* grad_Beff, int64_t N, int64_t nM, int64_t nT, void* stream)
{
    const void* parts[1] = {Mpre};
    return mrphy_blochsim_bwd_parts(dtype, parts, Mpre ? 1 : 0, MRPHY_HIST_BLOCKED, Beff, g, g_sn, g_sm, E1, E1_sn,
                                    E1_sm, E2, E2_sn, E2_sm, grad_Mo, grad_Mi, grad_Beff, nullptr, N, nM, nT, stream);
}

int mrphy_blochsim_bwd_consts(int dtype, const void* Mpre, const void* Beff, const void* g,
                              int64_t g_sn, int64_t g_sm, const void* E1, int64_t E1_sn,
                              int64_t E1_sm, const void* E2, int64_t E2_sn, int64_t E2_sm,
                              const void* grad_Mo, void* grad_Mi, void* grad_Beff, void* grad_consts,
                              int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (!grad_consts) return MRPHY_EINVAL;
    const void* parts[1] = {Mpre};
    return mrphy_blochsim_bwd_parts(dtype, parts, Mpre ? 1 : 0, MRPHY_HIST_BLOCKED, Beff, g, g_sn, g_sm, E1, E1_sn,
                                    E1_sm, E2, E2_sn, E2_sm, grad_Mo, grad_Mi, grad_Beff, grad_consts, N, nM, nT,
                                    stream);
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
                                                  nullptr, N, nM, nT, st)));
}

int mrphy_beff2ab_bwd_consts(int dtype, const void* hist, const void* Beff,
                      const void* g, int64_t g_sn, int64_t g_sm,
                      const void* E1, int64_t E1_sn, int64_t E1_sm,
                      const void* E2, int64_t E2_sn, int64_t E2_sm,
                      const void* grad_A, const void* grad_B, void* grad_Beff, void* grad_consts,
                      int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (nT == 0) {
        // an empty pulse: A = I, B = 0 whatever the constants are, so their gradients are exact zeros (the reference's
        // autograd gives them, beffective.py:73-100) -- written here, not left to the caller's allocation
        if (!grad_consts) return MRPHY_EINVAL;
        return (int)hipMemsetAsync(grad_consts, 0, (size_t)(N * nM) * 4 * tsize(dtype), (hipStream_t)stream);
    }
    if (!hist || !Beff || !g || !E1 || !E2 || !grad_Beff || !grad_consts) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_beff2ab_bwd<T, CT>(hist, Beff, bg, b1, b2, grad_A, grad_B, grad_Beff,
                                                  grad_consts, N, nM, nT, st)));
}

}  // extern "C"
