// host_common.hpp -- what every translation unit of libmrphy_hip.so starts with: the HIP runtime, the C ABI
// header, the shared geometry, the step math, the common device fragment (k_common.hpp, inside this
// unit's anonymous namespace: kernels have internal linkage, every unit carries its own code object)
// and the host-side helpers of the launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/mrphy_hip.h"
#include "geom.hpp"
#include "bloch_math.hpp"
#include "internal.hpp"

using namespace mrphy;

namespace {
#include "k_common.hpp"

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return (int)e;
}

inline size_t tsize(int dtype) { return dtype == MRPHY_F64 ? 8 : 4; }
inline size_t csize(int dtype) { return (dtype == MRPHY_F32 || dtype == MRPHY_F32P) ? 4 : 8; }

// steps per chunk of the chunked kernels.  fp32: 16 steps = 192 B per row.  fp64 forward: 8 steps =
// 192 B per row as well -- with 16 (384 B, 24 staging vectors = 96 VGPRs per lane) the fp64 forward
// builds need 354 VGPRs = ONE wave per SIMD; with 8, 244 = two (same-box A/B at 64^3 x 1024, round 3:
// K1 1.55 -> 1.42 ms, K1h 2.49 -> 2.40 ms).  The fp64 adjoint stays at 16: with 8 it got slower
// (3.92 -> 4.51 ms; 356 VGPRs either way is one wave per SIMD, and the smaller chunk doubles the
// barriers).  Putting the fp64 large-angle path (ocml sincos) behind a real call to shrink the
// kernels was tried too: the call's register convention spilled the HOT path (fused K2 0.78 -> 2.58 ms).
template <typename T> constexpr int TC_FWD = sizeof(T) == 8 ? 8 : 16;
template <typename T> constexpr int TC_BWD = 16;

// Development knobs exist only in the -DMRPHY_DEV_KNOBS build (tools/build_dev.py ->
// tools/libmrphy_hip_dev.so): environment variables that select alternative builds / block orders
// for A/B measurements (re-read at every launch, so one process can sweep them), and a
// per-workgroup time-stamp buffer.  The shipped library reads no
// environment variable and instantiates none of the alternatives.
#ifdef MRPHY_DEV_KNOBS
inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// MRPHY_K0_VARIANT = order*1000 + rows_per_block/8*10 + nt
inline int k0_variant() { return env_int("MRPHY_K0_VARIANT", 0); }
// MRPHY_BWD_VARIANT = waves per SIMD the K3 build is bounded for (2, 3, 4)
inline int bwd_variant() { return env_int("MRPHY_BWD_VARIANT", 0); }
// MRPHY_XCD_SWEEP=0 turns the XCD-contiguous tile order of the line kernels off
inline bool xcd_sweep() { return env_int("MRPHY_XCD_SWEEP", 1) != 0; }
// MRPHY_K1_XCD=0|1: XCD-contiguous tile order for the no-history K1 as well (each XCD reads the eighth of Beff
// that the same XCD slot of K0 wrote)
inline int k1_xcd(int dflt) { return env_int("MRPHY_K1_XCD", dflt); }      // 0 off, 1 forward, 2 reversed
// MRPHY_FWD_VARIANT = OCC*100 + SPLIT*10 + NT selects an alternative K1 build
inline int fwd_variant() { return env_int("MRPHY_FWD_VARIANT", 0); }
// MRPHY_K0_STEPS=0: multi-coil rfgr2beff on the element-per-thread builds instead of k_rfgr2beff_steps
inline bool k0_steps() { return env_int("MRPHY_K0_STEPS", 1) != 0; }
// MRPHY_K0_PK=0: exact coil counts on k_rfgr2beff_steps instead of the packed-scalar kernel k_rfgr2beff_pk
inline bool k0_pk() { return env_int("MRPHY_K0_PK", 1) != 0; }
// MRPHY_K0ADJ_TP: alternative multi-coil K0 adjoints (1|2|4: DPP pass, 0: element-per-thread pass, 12: SGPR pass, TP = 2)
inline int k0adj_tp() { return env_int("MRPHY_K0ADJ_TP", -1); }
// MRPHY_PRIO_ROT=N (re-read at every launch): rotate s_setprio with progress in the line kernels
inline int prio_rot() { return env_int("MRPHY_PRIO_ROT", 0); }
// MRPHY_LDS_PAD=bytes of dynamic LDS added to the line kernels' launches: caps the workgroups per CU
// (160 KB / (9216 + pad)) without touching the code -- occupancy experiments
inline unsigned lds_pad() { return (unsigned)env_int("MRPHY_LDS_PAD", 0); }
#else
constexpr int k0_variant() { return 0; }
constexpr int bwd_variant() { return 0; }
constexpr bool xcd_sweep() { return true; }
constexpr int fwd_variant() { return 0; }
constexpr int k1_xcd(int dflt) { return dflt; }
constexpr bool k0_steps() { return true; }
constexpr bool k0_pk() { return true; }
constexpr unsigned lds_pad() { return 0; }
#endif

inline int check_common(int dtype, int64_t N, int64_t nM, int64_t nT)
{
    if (dtype != MRPHY_F32 && dtype != MRPHY_F64 && dtype != MRPHY_F32_C64 &&
        dtype != MRPHY_F32P && dtype != MRPHY_F32P_C64)
        return MRPHY_EINVAL;
    if (N < 0 || nM < 0 || nT < 0) return MRPHY_EINVAL;
    return 0;
}

// rows on 128-B lines and whole 32-step periods: what the line-granular kernels need
inline bool lines_shape_ok(const void* Beff, int64_t nT)
{
    return aligned_to(Beff, 128) && nT > 0 && (nT % 32 == 0) && (768 * nT < (int64_t)4294967295);
}

// ... and for the fp64 line kernels: 16-step periods (a 128-B line = 16 doubles)
inline bool lines_shape_ok_f64(const void* Beff, int64_t nT)
{
    return aligned_to(Beff, 128) && nT > 0 && (nT % 16 == 0) && (1536 * nT < (int64_t)4294967295);
}

}  // namespace

// dtype code -> (T, CT); a unit instantiates its launchers for the codes in MRPHY_DT_MASK (bit = code)
#ifndef MRPHY_DT_MASK
#define MRPHY_DT_MASK 0x1f
#endif
#if MRPHY_DT_MASK & 1
#define MRPHY_IF_F32(X) X(float, float)
#else
#define MRPHY_IF_F32(X)
#endif
#if MRPHY_DT_MASK & 2
#define MRPHY_IF_F64(X) X(double, double)
#else
#define MRPHY_IF_F64(X)
#endif
#if MRPHY_DT_MASK & 4
#define MRPHY_IF_F32_C64(X) X(float, double)
#else
#define MRPHY_IF_F32_C64(X)
#endif
#if MRPHY_DT_MASK & 8
#define MRPHY_IF_F32P(X) X(float, prec_f32)
#else
#define MRPHY_IF_F32P(X)
#endif
#if MRPHY_DT_MASK & 16
#define MRPHY_IF_F32P_C64(X) X(float, prec_f64)
#else
#define MRPHY_IF_F32P_C64(X)
#endif
#define MRPHY_FOR_DTYPES(X) MRPHY_IF_F32(X) MRPHY_IF_F64(X) MRPHY_IF_F32_C64(X) MRPHY_IF_F32P(X) MRPHY_IF_F32P_C64(X)
// the units whose kernels have a data type only (K0 and its adjoint): bits 0 and 1
#define MRPHY_FOR_DATA_TYPES(X) MRPHY_IF_F32(X) MRPHY_IF_F64(X)
