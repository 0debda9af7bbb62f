// tu_aux.hip -- the elementwise entry points of the C ABI together with their kernels: freeprec, pulse
// interpolation, beff2uphi / uphirot and their adjoints, mask extract / embed, cube_loc, blochsim_ab.
// Two data types, no constant type: small enough to be one unit.
#include "host_common.hpp"

namespace {
#include "k_aux.hpp"
}  // namespace

extern "C" {

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

int mrphy_freeprec_bwd_consts(int dtype, const void* Mi, const void* grad_Mo, const void* dur, int64_t dur_sn,
                              const void* T1, int64_t T1_sn, int64_t T1_sm, const void* T2, int64_t T2_sn,
                              int64_t T2_sm, const void* df, int64_t df_sn, int64_t df_sm, void* grad_consts,
                              int64_t N, int64_t nM, void* stream)
{
    if ((dtype != MRPHY_F32 && dtype != MRPHY_F64) || N < 0 || nM < 0) return MRPHY_EINVAL;
    if (N * nM == 0) return 0;
    if (!Mi || !grad_Mo || !grad_consts || !dur || ((T1 == nullptr) != (T2 == nullptr))) return MRPHY_EINVAL;
    FreePrecArgs a;
    a.Mi = Mi; a.Mo = nullptr; a.dur = dur; a.dur_sn = dur_sn;
    a.T1 = Bc{T1, T1_sn, T1_sm}; a.T2 = Bc{T2, T2_sn, T2_sm}; a.df = Bc{df, df_sn, df_sm};
    a.rows = N * nM; a.nM = nM;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((a.rows + 255) / 256));
    if (dtype == MRPHY_F32) hipLaunchKernelGGL((k_freeprec_gc<float>), grid, dim3(256), 0, st, a, grad_Mo, grad_consts);
    else                    hipLaunchKernelGGL((k_freeprec_gc<double>), grid, dim3(256), 0, st, a, grad_Mo, grad_consts);
    return launch_status();
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
