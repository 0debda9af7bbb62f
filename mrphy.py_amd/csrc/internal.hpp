// internal.hpp -- the launchers behind the C ABI.  abi.hip validates arguments and dispatches on the dtype
// code; the launchers (and the kernels they start) live in the tu_*.hip units, one explicit instantiation
// per dtype code, so that the library compiles as independent units in parallel (mrphy_amd._lib.build).
// Hidden visibility: none of this is part of the exported interface (include/mrphy_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "geom.hpp"

#pragma GCC visibility push(hidden)
namespace mrphy_i {
using mrphy::Bc;
using mrphy::HistParts;

template <typename T, typename CT>
int run_fwd(const void* Mi, const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo,
            HistParts hist, int64_t N, int64_t nM, int64_t nT, hipStream_t st);

template <typename T, typename CT>
int run_bwd(HistParts hist, const void* Beff, Bc g, Bc E1, Bc E2, const void* gMo, void* gMi,
            void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st);

template <typename T>
int run_rfgr2beff(const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc,
                  Bc df, Bc gam, const void* b1, void* beff, int64_t N, int64_t nM, int64_t nT,
                  int64_t nC, int store, hipStream_t st);

template <typename T>
int run_rfgr2beff_bwd(const void* gB, const void* loc, const void* b1, void* grf, void* ggr,
                      void* work, int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st);

template <typename T, typename CT>
int run_rfgr_fwd(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                 int64_t nT, int64_t nC, hipStream_t st);

// the one-coil float builds of K2 live in a unit of their own (tu_fused_fwd1.hip: compiled with the max-ILP
// scheduling strategy, which the multi-coil and fp64 builds pay for in registers)
template <typename T, typename CT>
int run_rfgr_fwd1(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                  const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                  const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                  int64_t nT, hipStream_t st);

template <typename T, typename CT>
int run_rfgr_bwd(const void* Mck, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, const void* gMo, void* gMi, void* grf, void* ggr, void* work,
                 int64_t N, int64_t nM, int64_t nT, hipStream_t st);

template <typename T, typename CT>
int run_rfgr_mc_bwd(const void* Mck, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                    const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                    const void* E1m1, const void* gMo, void* gMi, void* grf, void* ggr, void* work,
                    int64_t N, int64_t nM, int64_t nT, int64_t nC, hipStream_t st);

template <typename T, typename CT>
int run_beff2ab(const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* A, void* B, void* hist,
                int64_t N, int64_t nM, int64_t nT, hipStream_t st);

template <typename T, typename CT>
int run_beff2ab_bwd(const void* hist, const void* Beff, Bc g, Bc E1, Bc E2, const void* gA,
                    const void* gB, void* gBeff, void* gC, int64_t N, int64_t nM, int64_t nT, hipStream_t st);

#ifdef MRPHY_DEV_KNOBS
// dev build only (tools/build_dev.py): device buffer of 4 x uint64 per workgroup that the line kernels fill
extern unsigned long long* g_dev_stamps;
extern int64_t g_dev_stamps_cap;
#endif
}  // namespace mrphy_i
#pragma GCC visibility pop
