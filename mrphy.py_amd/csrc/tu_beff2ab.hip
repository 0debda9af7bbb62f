// tu_beff2ab.hip -- launchers of mrphy_beff2ab / _save / _bwd
#include "host_common.hpp"

namespace {
#include "k_beff2ab.hpp"
}  // namespace

namespace mrphy_i {

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
                    const void* gB, void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    AbBwdArgs<T> a;
    a.hist = (const T*)hist; a.Beff = (const T*)Beff; a.gA = (const T*)gA; a.gB = (const T*)gB;
    a.gBeff = (T*)gBeff; a.gC = (T*)gC; a.g = g; a.E1 = E1; a.E2 = E2;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, sizeof(T)) && aligned_to(gBeff, sizeof(T));
    const dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
    // fp64: 8-step chunks (with 16 the build needs all 512 VGPRs and still spills; round 4)
    if (gC) hipLaunchKernelGGL((k_beff2ab_bwd<T, CT, (sizeof(T) == 8 ? 8 : TC_BWD<T>), true>), grid, dim3(WAVE), 0, st, a);
    else    hipLaunchKernelGGL((k_beff2ab_bwd<T, CT, (sizeof(T) == 8 ? 8 : TC_BWD<T>), false>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_beff2ab<T_, CT_>(const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* A, void* B, void* hist, int64_t N, int64_t nM, int64_t nT, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
#define MRPHY_INST(T_, CT_) template int mrphy_i::run_beff2ab_bwd<T_, CT_>(const void* hist, const void* Beff, Bc g, Bc E1, Bc E2, const void* gA, const void* gB, void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st);
MRPHY_FOR_DTYPES(MRPHY_INST)
#undef MRPHY_INST
