// mrphy_hip.hip -- gfx950 kernels + C ABI (include/mrphy_hip.h) for the Bloch-simulation hot path.
//
// Execution model shared by the time-stepping kernels (K1 fwd, K3 bwd, K2 fused):
//   * one workgroup = ONE 64-lane wavefront = 64 consecutive spins (rows of the compact layout);
//     lane = spin, the magnetisation lives in 3 VGPRs for the whole pulse, time loop in-kernel;
//   * Beff is (rows, nT, 3): a spin's samples are contiguous in time, lanes are nT*12 B apart.
//     A chunk of TC steps x 64 spins is fetched with lanes running ALONG TIME (16 B per lane,
//     whole 128-B lines per row), parked in VGPRs while the previous chunk is being integrated
//     (register prefetch: the bytes in flight live in registers, not in LDS), then transposed
//     through a padded LDS tile so that each lane reads its own spin's samples with
//     conflict-free ds_read_b128;
//   * history / gradient tiles go back the same way (in place in the LDS tile, coalesced store).
// No MFMA anywhere: the only contraction (loc . gr) has K = 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "../../include/mrphy_hip.h"
#include "bloch_math.hpp"

using namespace mrphy;

namespace {

constexpr int WAVE = 64;

// broadcastable per-spin constant (see mrphy_hip.h)
struct Bc {
    const void* p;
    int64_t sn, sm;
};

template <typename CT>
__device__ __forceinline__ CT bc_load(const Bc& b, int64_t n, int64_t s)
{
    return reinterpret_cast<const CT*>(b.p)[n * b.sn + s * b.sm];
}

template <typename T, typename CT>
__device__ __forceinline__ SpinConst<T, CT> load_consts(const Bc& g, const Bc& E1, const Bc& E2,
                                                        const void* E1m1, int64_t n, int64_t s)
{
    SpinConst<T, CT> k;
    k.g = bc_load<CT>(g, n, s);
    k.relax = (E1.p != nullptr);
    if (k.relax) {
        k.e1 = bc_load<CT>(E1, n, s);
        k.e2 = bc_load<CT>(E2, n, s);
        Bc e = {E1m1, E1.sn, E1.sm};
        k.e1m1 = E1m1 ? bc_load<CT>(e, n, s) : CT(0);
    } else {
        k.e1 = k.e2 = CT(1);
        k.e1m1 = CT(0);
    }
    return k;
}

// ---------------------------------------------------------------------------------------------
// Chunk tile geometry: 64 rows x (3*TC) elements, LDS pitch padded by one 16-B slot so that
// "lane = row, same column" ds_read_b128 is conflict-free (slots per row is odd).
// ---------------------------------------------------------------------------------------------
template <typename T, int TC>
struct Tile {
    static constexpr int VE = V16<T>::N;            // elements per 16-B vector
    static constexpr int RL = 3 * TC;               // row length of a chunk, elements
    static constexpr int PITCH = RL + VE;           // padded LDS pitch, elements
    static constexpr int SPR = RL / VE;             // 16-B slots per row
    static constexpr int NL = SPR;                  // vector loads per lane per chunk
    static constexpr int ELEMS = WAVE * PITCH;
    static_assert(RL % VE == 0, "chunk row must be a whole number of 16-B slots");
    static_assert(SPR % 2 == 0, "pitch (SPR+1 slots) must be odd for conflict-free reads");
    using V = typename V16<T>::type;
};

// A chunk parked in registers (NL 16-B vectors per lane).  Passed and returned BY VALUE so that
// it is scalarised into VGPRs; through a pointer hipcc leaves it in scratch memory.
template <typename T, int TC>
struct Stage {
    typename Tile<T, TC>::V v[Tile<T, TC>::NL];
};

// global -> registers: lane `lane` fetches slots j = i*64 + lane of the 64 x SPR slot grid.
template <typename T, int TC>
__device__ __forceinline__ Stage<T, TC> chunk_fetch(const T* __restrict__ base, int64_t row0,
                                                    int64_t rows, int64_t rowlen, int64_t t0,
                                                    int lane)
{
    using TL = Tile<T, TC>;
    Stage<T, TC> st;
#pragma unroll
    for (int i = 0; i < TL::NL; ++i) {
        const int j = i * WAVE + lane;
        const int jr = j / TL::SPR, jc = j % TL::SPR;
        int64_t rr = row0 + jr;
        rr = rr < rows ? rr : rows - 1;
        const T* src = base + rr * rowlen + t0 * 3 + jc * TL::VE;
        st.v[i] = *reinterpret_cast<const typename TL::V*>(src);
    }
    return st;
}

template <typename T, int TC>
__device__ __forceinline__ void chunk_to_lds(T* tile, const Stage<T, TC> st, int lane)
{
    using TL = Tile<T, TC>;
#pragma unroll
    for (int i = 0; i < TL::NL; ++i) {
        const int j = i * WAVE + lane;
        const int jr = j / TL::SPR, jc = j % TL::SPR;
        *reinterpret_cast<typename TL::V*>(tile + jr * TL::PITCH + jc * TL::VE) = st.v[i];
    }
}

// LDS tile -> global, coalesced (the inverse mapping); rows beyond `rows` are skipped.
template <typename T, int TC>
__device__ __forceinline__ void chunk_store(const T* tile, T* __restrict__ base, int64_t row0,
                                            int64_t rows, int64_t rowlen, int64_t t0, int lane)
{
    using TL = Tile<T, TC>;
#pragma unroll
    for (int i = 0; i < TL::NL; ++i) {
        const int j = i * WAVE + lane;
        const int jr = j / TL::SPR, jc = j % TL::SPR;
        const int64_t rr = row0 + jr;
        const typename TL::V v =
            *reinterpret_cast<const typename TL::V*>(tile + jr * TL::PITCH + jc * TL::VE);
        if (rr < rows)
            *reinterpret_cast<typename TL::V*>(base + rr * rowlen + t0 * 3 + jc * TL::VE) = v;
    }
}

__device__ __forceinline__ void vec_unpack(const f32x4 v, float* o)
{
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void vec_unpack(const f64x2 v, double* o)
{
    o[0] = v.x; o[1] = v.y;
}
__device__ __forceinline__ f32x4 vec_pack(const float* o) { return f32x4{o[0], o[1], o[2], o[3]}; }
__device__ __forceinline__ f64x2 vec_pack(const double* o) { return f64x2{o[0], o[1]}; }

// ---------------------------------------------------------------------------------------------
// History of the forward sweep for the adjoint: the magnetisation BEFORE each step.  It is an
// internal buffer (never an API tensor), so it is laid out for the kernels, structure-of-arrays
// per 64-spin tile:  hist[tile][t][xyz][lane].  Every store / load is one fully coalesced 256-B
// wave access; no LDS transposition is needed on either side.
// ---------------------------------------------------------------------------------------------
constexpr int HIST_STEP = 3 * WAVE;               // elements per time step of one tile

template <typename T>
__device__ __forceinline__ void hist_store(T* hp, int64_t t, T mx, T my, T mz)
{
    T* q = hp + t * HIST_STEP;      // written once, read once by the adjoint: nt (plain: same time)
    __builtin_nontemporal_store(mx, q);
    __builtin_nontemporal_store(my, q + WAVE);
    __builtin_nontemporal_store(mz, q + 2 * WAVE);
}

template <typename T>
__device__ __forceinline__ void hist_load(const T* hp, int64_t t, T& mx, T& my, T& mz)
{
    const T* q = hp + t * HIST_STEP;
    mx = __builtin_nontemporal_load(q);
    my = __builtin_nontemporal_load(q + WAVE);
    mz = __builtin_nontemporal_load(q + 2 * WAVE);
}

// =============================================================================================
// K1: blochsim forward, materialised Beff.
// =============================================================================================
template <typename T>
struct FwdArgs {
    const T* Mi;
    const T* Beff;
    T* Mo;
    T* Mpre;
    Bc g, E1, E2;
    const void* E1m1;
    int64_t rows, nM, nT;
    int vec_ok;
    unsigned per_xcd;      // line kernels: > 0 -> block b works on spin tile (b % 8) * per_xcd + b / 8
};

// Blocks are dealt round-robin to the 8 XCDs; with this map each XCD walks its own contiguous
// eighth of the spin tiles (see run_rfgr2beff for what that is worth on the write side).
__device__ __forceinline__ int64_t xcd_tile(unsigned per_xcd)
{
    return per_xcd ? (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
}

template <typename T, typename CT, int TC, bool SAVE>
__global__ __launch_bounds__(WAVE) void k_bloch_fwd(FwdArgs<T> a)
{
    using TL = Tile<T, TC>;
    using V = typename TL::V;
    constexpr int VE = TL::VE;
    __shared__ __attribute__((aligned(16))) T tile[TL::ELEMS];

    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);

    T mx = a.Mi[rc * 3 + 0], my = a.Mi[rc * 3 + 1], mz = a.Mi[rc * 3 + 2];
    const int64_t rowlen = 3 * a.nT;
    int64_t t = 0;

    T* hp = SAVE ? a.Mpre + (int64_t)blockIdx.x * a.nT * HIST_STEP + lane : nullptr;
    if (a.vec_ok) {
        const int64_t nfull = a.nT / TC;
        Stage<T, TC> st;
        if (nfull > 0) st = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, 0, lane);
        T* myrow = tile + lane * TL::PITCH;
        for (int64_t c = 0; c < nfull; ++c) {
            __syncthreads();                         // tile free (previous chunk consumed)
            chunk_to_lds<T, TC>(tile, st, lane);
            __syncthreads();
            if (c + 1 < nfull)                       // next chunk flies while this one integrates
                st = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, (c + 1) * TC, lane);
#pragma unroll 1
            for (int tt = 0; tt < TC; tt += VE) {    // VE steps = 3 vectors = 48 B per lane
                T bb[3 * VE];
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3), bb);
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3 + VE), bb + VE);
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3 + 2 * VE), bb + 2 * VE);
                T gBx[VE], gBy[VE], gBz[VE];
#pragma unroll
                for (int q = 0; q < VE; ++q) { gBx[q] = bb[3 * q]; gBy[q] = bb[3 * q + 1]; gBz[q] = bb[3 * q + 2]; }
                Rot<T> rr[VE];
                rot_prepare<T, CT, VE>(k, gBx, gBy, gBz, rr);
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    if (SAVE) hist_store<T>(hp, c * TC + tt + q, mx, my, mz);
                    if (k.relax) rot_apply<true, T, CT>(k, rr[q], mx, my, mz);
                    else         rot_apply<false, T, CT>(k, rr[q], mx, my, mz);
                }
            }
        }
        t = nfull * TC;
    }
    // tail steps and the unaligned-shape path: each lane reads its own samples directly
    const T* bp = a.Beff + rc * rowlen;
    for (; t < a.nT; ++t) {
        if (SAVE) hist_store<T>(hp, t, mx, my, mz);
        bloch_step<T, CT>(k, bp[t * 3], bp[t * 3 + 1], bp[t * 3 + 2], mx, my, mz);
    }
    if (valid) { a.Mo[r * 3] = mx; a.Mo[r * 3 + 1] = my; a.Mo[r * 3 + 2] = mz; }
}


// =============================================================================================
// K1, line-granular variant (the headline path): float data, no history, rows 128-B aligned
// (Beff base % 128 == 0 and nT % 32 == 0).
//
// The chunked kernel above fetches 16 steps = 192 B per spin per chunk, i.e. one and a half
// cache lines: measured with FETCH_SIZE it reads 1.22x the algorithmic bytes, because the shared
// half line has usually left L2 when the next chunk asks for it.  Here the unit of transfer is
// ONE 128-B line per spin ("piece" = 32 floats = 10 2/3 steps):
//   * a piece of the 64-spin tile is 8 wave-loads; load i, lane l fetches 16 B of row 8i + l/8 at
//     byte 16*(l%8) of that row's line: every wave-load covers 8 rows x one WHOLE line;
//   * the next piece waits in 8 VGPR quads (32 VGPRs) while the current one is integrated;
//   * LDS tile 64 x (32+4) floats = 9 KB; lane = spin reads its row with conflict-free reads
//     (pitch 9 x 16 B, odd);
//   * a step needs 3 consecutive floats, so steps straddle piece boundaries; 3 pieces = 96 floats =
//     32 steps is the period: piece 0 holds steps 0-9 + 2 floats of step 10, piece 1 the rest of
//     step 10, steps 11-20 + 1 float of step 21, piece 2 the rest of step 21 and steps 22-31.  The
//     straddling floats travel in two carry registers.
// =============================================================================================
template <bool NT>
__device__ __forceinline__ f32x4 ldv(const f32x4* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// NA steps whose samples start at float `first` of this lane's LDS row, optionally preceded by a
// straddling step whose leading floats arrive in registers.
// SAVE: record the magnetisation before each step at hist[t], t = th, th+1, ...
template <bool RELAX, bool SAVE, typename CT, int NA>
__device__ __forceinline__ void lines_steps(const SpinConst<float, CT>& k, const float* q,
                                            float* hp, int64_t th, float& mx, float& my, float& mz)
{
    float Bx[NA], By[NA], Bz[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { Bx[j] = q[3 * j]; By[j] = q[3 * j + 1]; Bz[j] = q[3 * j + 2]; }
    Rot<float> r[NA];
    rot_prepare<float, CT, NA>(k, Bx, By, Bz, r);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        if (SAVE) hist_store<float>(hp, th + j, mx, my, mz);
        rot_apply<RELAX, float, CT>(k, r[j], mx, my, mz);
    }
}

// 1 straddling step (b0,b1,b2 given) + NA steps from q
template <bool RELAX, bool SAVE, typename CT, int NA>
__device__ __forceinline__ void lines_steps_carry(const SpinConst<float, CT>& k, float b0, float b1,
                                                  float b2, const float* q, float* hp, int64_t th,
                                                  float& mx, float& my, float& mz)
{
    float Bx[NA + 1], By[NA + 1], Bz[NA + 1];
    Bx[0] = b0; By[0] = b1; Bz[0] = b2;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        Bx[j + 1] = q[3 * j]; By[j + 1] = q[3 * j + 1]; Bz[j + 1] = q[3 * j + 2];
    }
    Rot<float> r[NA + 1];
    rot_prepare<float, CT, NA + 1>(k, Bx, By, Bz, r);
#pragma unroll
    for (int j = 0; j < NA + 1; ++j) {
        if (SAVE) hist_store<float>(hp, th + j, mx, my, mz);
        rot_apply<RELAX, float, CT>(k, r[j], mx, my, mz);
    }
}

// OCC: waves per SIMD the register allocation is bounded for.  SPLIT: sub-batches per piece
// (2: 5/6 steps prepared at once, 3: 3/4 steps -- fewer live registers).  NT: non-temporal loads.
// (Tried: three pieces in flight per wave instead of one -- 96 prefetch VGPRs, 2 waves/SIMD -- no
// gain at any grid size.)
template <typename CT, bool RELAX, int OCC, int SPLIT, bool NT, bool SAVE>
__global__ __launch_bounds__(WAVE, OCC) void k_bloch_fwd_lines(FwdArgs<float> a)
{
    using T = float;
    constexpr int PF = 32;                 // floats per piece = one 128-B line
    constexpr int PITCH = PF + 4;          // 9 slots of 16 B
    __shared__ __attribute__((aligned(16))) T tile[WAVE * PITCH];

    const int lane = threadIdx.x;
    const int64_t tile_id = xcd_tile(a.per_xcd);
    if (tile_id * WAVE >= a.rows) return;
    const int64_t row0 = tile_id * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
    T mx = a.Mi[rc * 3 + 0], my = a.Mi[rc * 3 + 1], mz = a.Mi[rc * 3 + 2];

    const int64_t rowlen = 3 * a.nT;                       // floats; multiple of 96
    const int64_t npieces = rowlen / PF;                   // multiple of 3
    const int frow = lane >> 3, fcol = (lane & 7) * 4;
    // wave-uniform base (SGPRs) + 32-bit per-lane offsets: loads use the saddr+voffset form and
    // need 8 VGPRs of addressing instead of 16 (host guarantees 64*rowlen < 2^31)
    const T* __restrict__ base = a.Beff + row0 * rowlen;
    const int64_t last = a.rows - 1 - row0;                // last valid row of this tile
    // byte offset of load i = min(off0 + i * ostride, olim): rows past the end of the last tile
    // re-read its last valid row (two VGPRs instead of eight precomputed offsets)
    const unsigned ostride = (unsigned)(8 * rowlen * sizeof(T));
    const unsigned off0 = (unsigned)(((frow < last ? frow : last) * rowlen + fcol) * sizeof(T));
    const unsigned olim = (unsigned)(((last < 63 ? last : 63) * rowlen + fcol) * sizeof(T));
// (o0 is laundered through an empty asm per piece, or the compiler hoists all eight offsets back
// into registers for the whole loop)
#define MRPHY_OFF(i) (min(o0 + (unsigned)(i) * ostride, olim))
    T* wr = tile + frow * PITCH + fcol;                    // + i*8*PITCH per load
    const T* my_ = tile + lane * PITCH;

    f32x4 st0[8];
#define MRPHY_FETCH(S, p)                                                                  \
    { unsigned o0 = off0; asm volatile("" : "+v"(o0));                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        S[i] = ldv<NT>(reinterpret_cast<const f32x4*>(                                      \
            reinterpret_cast<const char*>(base + (p) * PF) + MRPHY_OFF(i))); }
#define MRPHY_STAGE(S)                                                                     \
    __syncthreads();                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        *reinterpret_cast<f32x4*>(wr + i * 8 * PITCH) = S[i];                              \
    __syncthreads();

    T* hp = SAVE ? a.Mpre + tile_id * a.nT * HIST_STEP + lane : nullptr;
#define LS(NA_, Q_, TH_) lines_steps<RELAX, SAVE, CT, NA_>(k, my_ + (Q_), hp, t0 + (TH_), mx, my, mz)
#define LC(NA_, B0_, B1_, B2_, Q_, TH_) \
    lines_steps_carry<RELAX, SAVE, CT, NA_>(k, B0_, B1_, B2_, my_ + (Q_), hp, t0 + (TH_), mx, my, mz)
    if (npieces > 0) { MRPHY_FETCH(st0, 0) }
    T c0, c1;
    for (int64_t p = 0; p < npieces; p += 3) {
        const int64_t t0 = (p / 3) * 32;
        const bool more = p + 3 < npieces;
        // piece 0: steps 0..9 (floats 0..29), carry floats 30, 31
        MRPHY_STAGE(st0)
        MRPHY_FETCH(st0, p + 1)
        if (SPLIT == 2)      { LS(5, 0, 0); LS(5, 15, 5); }
        else if (SPLIT == 3) { LS(4, 0, 0); LS(3, 12, 4); LS(3, 21, 7); }
        else                 { LS(3, 0, 0); LS(3, 9, 3); LS(2, 18, 6); LS(2, 24, 8); }
        c0 = my_[30]; c1 = my_[31];
        // piece 1: step 10 = (c0, c1, f0); steps 11..20 from float 1; carry float 31
        MRPHY_STAGE(st0)
        MRPHY_FETCH(st0, p + 2)
        if (SPLIT == 2)      { LC(5, c0, c1, my_[0], 1, 10); LS(5, 16, 16); }
        else if (SPLIT == 3) { LC(3, c0, c1, my_[0], 1, 10); LS(4, 10, 14); LS(3, 22, 18); }
        else { LC(2, c0, c1, my_[0], 1, 10); LS(3, 7, 13); LS(3, 16, 16); LS(2, 25, 19); }
        c0 = my_[31];
        // piece 2: step 21 = (c0, f0, f1); steps 22..31 from float 2
        MRPHY_STAGE(st0)
        if (more) { MRPHY_FETCH(st0, p + 3) }
        if (SPLIT == 2)      { LC(5, c0, my_[0], my_[1], 2, 21); LS(5, 17, 27); }
        else if (SPLIT == 3) { LC(3, c0, my_[0], my_[1], 2, 21); LS(4, 11, 25); LS(3, 23, 29); }
        else { LC(2, c0, my_[0], my_[1], 2, 21); LS(3, 8, 24); LS(3, 17, 27); LS(2, 26, 30); }
    }
#undef LS
#undef LC
#undef MRPHY_FETCH
#undef MRPHY_STAGE
#undef MRPHY_OFF
    if (valid) { a.Mo[r * 3] = mx; a.Mo[r * 3 + 1] = my; a.Mo[r * 3 + 2] = mz; }
}

// =============================================================================================
// K3: blochsim backward.  Reads Beff and Mpre chunks (two tiles), sweeps time backwards,
// writes dL/dBeff in place into the Beff tile, stores it coalesced.
// =============================================================================================
template <typename T>
struct BwdArgs {
    const T* Mpre;
    const T* Beff;
    const T* gMo;
    T* gMi;
    T* gBeff;
    Bc g, E1, E2;
    int64_t rows, nM, nT;
    int vec_ok;
    unsigned per_xcd;
};

template <typename T, typename CT, int TC>
__global__ __launch_bounds__(WAVE) void k_bloch_bwd(BwdArgs<T> a)
{
    using TL = Tile<T, TC>;
    using V = typename TL::V;
    constexpr int VE = TL::VE;
    __shared__ __attribute__((aligned(16))) T tileB[TL::ELEMS];

    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, nullptr, n, s);

    T hx = a.gMo[rc * 3 + 0], hy = a.gMo[rc * 3 + 1], hz = a.gMo[rc * 3 + 2];
    const int64_t rowlen = 3 * a.nT;
    const int64_t nfull = a.vec_ok ? a.nT / TC : 0;
    const T* hp = a.Mpre + (int64_t)blockIdx.x * a.nT * HIST_STEP + lane;

    // tail first (we run time backwards)
    {
        const T* bp = a.Beff + rc * rowlen;
        T* gp = a.gBeff ? a.gBeff + rc * rowlen : nullptr;
        for (int64_t t = a.nT - 1; t >= nfull * TC; --t) {
            T gx, gy, gz, m0, m1, m2;
            hist_load<T>(hp, t, m0, m1, m2);
            bloch_step_adj<T, CT>(k, bp[t * 3], bp[t * 3 + 1], bp[t * 3 + 2], m0, m1, m2,
                                  hx, hy, hz, gx, gy, gz);
            if (gp && valid) { gp[t * 3] = gx; gp[t * 3 + 1] = gy; gp[t * 3 + 2] = gz; }
        }
    }
    if (nfull > 0) {
        Stage<T, TC> stB = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, (nfull - 1) * TC, lane);
        T* rowB = tileB + lane * TL::PITCH;
        for (int64_t c = nfull - 1; c >= 0; --c) {
            __syncthreads();
            chunk_to_lds<T, TC>(tileB, stB, lane);
            __syncthreads();
            if (c > 0) stB = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, (c - 1) * TC, lane);
#pragma unroll 1
            for (int tt = TC - VE; tt >= 0; tt -= VE) {
                T bb[3 * VE], gg[3 * VE];
                vec_unpack(*reinterpret_cast<const V*>(rowB + tt * 3), bb);
                vec_unpack(*reinterpret_cast<const V*>(rowB + tt * 3 + VE), bb + VE);
                vec_unpack(*reinterpret_cast<const V*>(rowB + tt * 3 + 2 * VE), bb + 2 * VE);
                T Bx[VE], By[VE], Bz[VE], M0[VE], M1[VE], M2[VE];
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    Bx[q] = bb[3 * q]; By[q] = bb[3 * q + 1]; Bz[q] = bb[3 * q + 2];
                    hist_load<T>(hp, c * TC + tt + q, M0[q], M1[q], M2[q]);
                }
                RotAdj<T> ra[VE];
                rot_prepare_adj<T, CT, VE>(k, Bx, By, Bz, ra);
#pragma unroll
                for (int q = VE - 1; q >= 0; --q) {
                    if (k.relax)
                        rot_apply_adj<true, T, CT>(k, ra[q], M0[q], M1[q], M2[q], hx, hy, hz,
                                                   gg[3 * q], gg[3 * q + 1], gg[3 * q + 2]);
                    else
                        rot_apply_adj<false, T, CT>(k, ra[q], M0[q], M1[q], M2[q], hx, hy, hz,
                                                    gg[3 * q], gg[3 * q + 1], gg[3 * q + 2]);
                }
                *reinterpret_cast<V*>(rowB + tt * 3) = vec_pack(gg);
                *reinterpret_cast<V*>(rowB + tt * 3 + VE) = vec_pack(gg + VE);
                *reinterpret_cast<V*>(rowB + tt * 3 + 2 * VE) = vec_pack(gg + 2 * VE);
            }
            if (a.gBeff) {
                __syncthreads();
                chunk_store<T, TC>(tileB, a.gBeff, row0, a.rows, rowlen, c * TC, lane);
            }
        }
    }
    if (valid && a.gMi) { a.gMi[r * 3] = hx; a.gMi[r * 3 + 1] = hy; a.gMi[r * 3 + 2] = hz; }
}

// =============================================================================================
// K3, line-granular variant: float, rows 128-B aligned (same conditions as k_bloch_fwd_lines).
// Beff arrives in 128-B pieces through the LDS tile exactly as in the forward kernel, the history
// comes straight from the SoA buffer, and dL/dBeff replaces Beff in the tile in place and leaves
// as whole lines.  Time runs backwards, so pieces are visited 2, 1, 0 within each 32-step period:
//   * a step is handled in the turn of the piece holding its LAST float; the leading floats of a
//     straddling step (1 or 2 of them, at the end of the previous piece) come from a tiny
//     per-lane "tail" load, issued a piece ahead (the line is fetched by the next piece anyway);
//   * the gradient components of those leading floats belong to the previous piece's tile, which
//     is staged next: they travel in two carry registers and are dropped into it then.
// =============================================================================================
// History of one batch of steps (at most HB_MAX), fetched ONE BATCH AHEAD of its use: a load issued
// at the start of the batch that consumes it has ~300 cycles to land, and -- vmcnt being in-order --
// waiting for it also waits for everything issued before it (the Beff prefetch of the turn, the
// grad_Beff stores of the previous piece).
constexpr int HB_MAX = 4;
struct HistBatch {
    float m0[HB_MAX], m1[HB_MAX], m2[HB_MAX];
};

template <int NA>
__device__ __forceinline__ void hist_fetch(const float* hp, int64_t th, HistBatch& h)
{
    static_assert(NA <= HB_MAX, "batch larger than HistBatch");
#pragma unroll
    for (int j = 0; j < NA; ++j) hist_load<float>(hp, th + j, h.m0[j], h.m1[j], h.m2[j]);
}

// NA steps (fields at q, history in h), time reversed; dL/dBeff replaces the fields in place.
template <bool RELAX, typename CT, int NA>
__device__ __forceinline__ void lines_adj(const SpinConst<float, CT>& k, float* q,
                                          const HistBatch& h, float& hx, float& hy, float& hz)
{
    float Bx[NA], By[NA], Bz[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { Bx[j] = q[3 * j]; By[j] = q[3 * j + 1]; Bz[j] = q[3 * j + 2]; }
    RotAdj<float> ra[NA];
    rot_prepare_adj<float, CT, NA>(k, Bx, By, Bz, ra);
#pragma unroll
    for (int j = NA - 1; j >= 0; --j) {
        float gx, gy, gz;
        rot_apply_adj<RELAX, float, CT>(k, ra[j], h.m0[j], h.m1[j], h.m2[j], hx, hy, hz, gx, gy, gz);
        q[3 * j] = gx; q[3 * j + 1] = gy; q[3 * j + 2] = gz;
    }
}

// NA steps from q plus, last in reversed time order, the straddling step whose field is
// (b0, b1, b2) and whose history is h[0]; its gradient is returned in (g0, g1, g2).
template <bool RELAX, typename CT, int NA>
__device__ __forceinline__ void lines_adj_carry(const SpinConst<float, CT>& k, float b0, float b1,
                                                float b2, float* q, const HistBatch& h,
                                                float& hx, float& hy, float& hz, float& g0,
                                                float& g1, float& g2)
{
    static_assert(NA + 1 <= HB_MAX, "batch larger than HistBatch");
    float Bx[NA + 1], By[NA + 1], Bz[NA + 1];
    Bx[0] = b0; By[0] = b1; Bz[0] = b2;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        Bx[j + 1] = q[3 * j]; By[j + 1] = q[3 * j + 1]; Bz[j + 1] = q[3 * j + 2];
    }
    RotAdj<float> ra[NA + 1];
    rot_prepare_adj<float, CT, NA + 1>(k, Bx, By, Bz, ra);
#pragma unroll
    for (int j = NA; j >= 1; --j) {
        float gx, gy, gz;
        rot_apply_adj<RELAX, float, CT>(k, ra[j], h.m0[j], h.m1[j], h.m2[j], hx, hy, hz, gx, gy, gz);
        q[3 * (j - 1)] = gx; q[3 * (j - 1) + 1] = gy; q[3 * (j - 1) + 2] = gz;
    }
    rot_apply_adj<RELAX, float, CT>(k, ra[0], h.m0[0], h.m1[0], h.m2[0], hx, hy, hz, g0, g1, g2);
}

template <typename CT, bool RELAX, int OCC, bool NT>
__global__ __launch_bounds__(WAVE, OCC) void k_bloch_bwd_lines(BwdArgs<float> a)
{
    using T = float;
    constexpr int PF = 32;
    constexpr int PITCH = PF + 4;
    __shared__ __attribute__((aligned(16))) T tile[WAVE * PITCH];

    const int lane = threadIdx.x;
    const int64_t tile_id = xcd_tile(a.per_xcd);
    if (tile_id * WAVE >= a.rows) return;
    const int64_t row0 = tile_id * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, nullptr, n, s);
    T hx = a.gMo[rc * 3 + 0], hy = a.gMo[rc * 3 + 1], hz = a.gMo[rc * 3 + 2];

    const int64_t rowlen = 3 * a.nT;
    const int64_t npieces = rowlen / PF;                   // multiple of 3
    const int frow = lane >> 3, fcol = (lane & 7) * 4;
    const T* __restrict__ base = a.Beff + row0 * rowlen;
    T* __restrict__ obase = a.gBeff ? a.gBeff + row0 * rowlen : nullptr;
    const int64_t last = a.rows - 1 - row0;
    unsigned off[8];
    bool rowok[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t rr = (i * 8 + frow) < last ? (i * 8 + frow) : last;
        off[i] = (unsigned)((rr * rowlen + fcol) * sizeof(T));
        rowok[i] = (i * 8 + frow) <= last;
    }
    T* wr = tile + frow * PITCH + fcol;
    T* my_ = tile + lane * PITCH;
    const T* hp = a.Mpre + tile_id * a.nT * HIST_STEP + lane;
    const T* rowp = a.Beff + rc * rowlen;                  // this lane's own row, for the tails

    f32x4 st[8];
#define MRPHY_FETCH(p)                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        st[i] = ldv<NT>(reinterpret_cast<const f32x4*>(                                     \
            reinterpret_cast<const char*>(base + (p) * PF) + off[i]));
#define MRPHY_STAGE()                                                                      \
    __syncthreads();                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        *reinterpret_cast<f32x4*>(wr + i * 8 * PITCH) = st[i];                             \
    __syncthreads();
#define MRPHY_STORE(p)                                                                     \
    if (obase) {                                                                           \
        __syncthreads();                                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                    \
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + i * 8 * PITCH);           \
            if (rowok[i])                                                                  \
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(                   \
                    reinterpret_cast<char*>(obase + (p) * PF) + off[i]));                  \
        }                                                                                  \
    }
    // Batches of a 32-step period in processing order (time reversed), steps [first, count]:
    //   piece p+2: [29,3] [25,4] [21,4: carry 21 + 22..24]    piece p+1: [18,3] [14,4] [10,4: carry]
    //   piece p  : [7,3] [4,3] [0,4]
    // H0/H1 alternate: each batch issues the history loads of the NEXT one before it computes.
    // First in a turn the order is: stage, next batch's history, next piece's Beff, compute.
#define LA(NA_, Q_, H_) lines_adj<RELAX, CT, NA_>(k, my_ + (Q_), H_, hx, hy, hz)
    HistBatch H0, H1;
    if (npieces > 0) {
        MRPHY_FETCH(npieces - 1)
        hist_fetch<3>(hp, (npieces / 3 - 1) * 32 + 29, H0);
    }
    for (int64_t p = npieces - 3; p >= 0; p -= 3) {
        const int64_t t0 = (p / 3) * 32;
        T g0, g1, g2;
        // ---- piece p+2: floats 64..95 of the period.  steps 31..22 (from float 2), then the
        //      straddling step 21 = (tail float 63 | floats 0, 1)
        const T tl63 = rowp[(p + 2) * PF - 1];
        MRPHY_STAGE()
        hist_fetch<4>(hp, t0 + 25, H1);
        MRPHY_FETCH(p + 1)
        LA(3, 23, H0);
        hist_fetch<4>(hp, t0 + 21, H0);
        LA(4, 11, H1);
        hist_fetch<3>(hp, t0 + 18, H1);
        lines_adj_carry<RELAX, CT, 3>(k, tl63, my_[0], my_[1], my_ + 2, H0, hx, hy, hz, g0, g1, g2);
        my_[0] = g1; my_[1] = g2;
        T cg31 = g0;                                       // -> float 31 of piece p+1
        MRPHY_STORE(p + 2)
        // ---- piece p+1: floats 32..63.  steps 20..11 (from float 1), straddling step 10 =
        //      (tail floats 30, 31 | float 0)
        const T tl30 = rowp[(p + 1) * PF - 2], tl31 = rowp[(p + 1) * PF - 1];
        MRPHY_STAGE()
        hist_fetch<4>(hp, t0 + 14, H0);
        MRPHY_FETCH(p)
        my_[31] = cg31;
        LA(3, 22, H1);
        hist_fetch<4>(hp, t0 + 10, H1);
        LA(4, 10, H0);
        hist_fetch<3>(hp, t0 + 7, H0);
        lines_adj_carry<RELAX, CT, 3>(k, tl30, tl31, my_[0], my_ + 1, H1, hx, hy, hz, g0, g1, g2);
        my_[0] = g2;
        MRPHY_STORE(p + 1)
        // ---- piece p: floats 0..31.  floats 30, 31 <- carried gradient of step 10; steps 9..0
        MRPHY_STAGE()
        hist_fetch<3>(hp, t0 + 4, H1);
        if (p > 0) { MRPHY_FETCH(p - 1) }
        my_[30] = g0; my_[31] = g1;
        LA(3, 21, H0);
        hist_fetch<4>(hp, t0 + 0, H0);
        LA(3, 12, H1);
        if (p > 0) hist_fetch<3>(hp, t0 - 32 + 29, H1);    // first batch of the next period
        LA(4, 0, H0);
        H0 = H1;
        MRPHY_STORE(p)
    }
#undef MRPHY_FETCH
#undef MRPHY_STAGE
#undef MRPHY_STORE
#undef LA
    if (valid && a.gMi) { a.gMi[r * 3] = hx; a.gMi[r * 3 + 1] = hy; a.gMi[r * 3 + 2] = hz; }
}

// =============================================================================================
// K0: rfgr2beff.  Pure HBM-write kernel: every thread owns VW consecutive elements of the
// (t, xyz) axis (one 16-B store), keeps their pulse samples in registers, and walks down ROWS
// spins; the per-spin operands (loc, b1, df/gamma) are wave-uniform loads.
// =============================================================================================
template <typename T>
struct BeffArgs {
    const T* rf;  int64_t rf_sn;     // (N|1, 2, nT, nC)
    const T* gr;  int64_t gr_sn;     // (N|1, 3, nT)
    const T* loc;                    // (N, nM, 3)
    Bc df, gam;                      // df.p may be null
    const T* b1;                     // (N, nM, 2, nC) or null
    T* beff;                         // (N, nM, nT, 3)
    int64_t nM, nT, nC;
    int rows_per_block;
    int nt;                          // non-temporal stores
    unsigned gy;                     // > 0: grid.x = spin tile * gy + time tile (time tile fastest)
    unsigned nblk, per_xcd;          // per_xcd > 0: block b works on tile (b % 8) * per_xcd + b / 8
};

constexpr int K0_THREADS = 256;
constexpr int K0_MAX_ROWS = 256;

// NC1 = true: single coil, pulse samples in registers.  NC1 = false: any nC, coil loop reads the
// rf samples from global memory (L1/L2 resident: 8*nC bytes per time point).
template <typename T, int VW, bool NC1>
__global__ __launch_bounds__(K0_THREADS) void k_rfgr2beff(BeffArgs<T> a)
{
    const int64_t L = 3 * a.nT;
    // grid: x = spin tile (can be large), y = tile of the (t, xyz) axis, z = batch entry
    unsigned tile = blockIdx.x;
    if (a.per_xcd) {
        tile = (blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3);
        if (tile >= a.nblk) return;
    }
    const unsigned by = a.gy ? tile % a.gy : blockIdx.y;
    const unsigned bx = a.gy ? tile / a.gy : tile;
    const int64_t e0 = ((int64_t)by * K0_THREADS + threadIdx.x) * VW;
    const int64_t n = blockIdx.z;
    const int64_t s0 = (int64_t)bx * a.rows_per_block;
    const int64_t s1 = (s0 + a.rows_per_block < a.nM) ? s0 + a.rows_per_block : a.nM;

    const T* rf = a.rf + n * a.rf_sn;
    const T* gr = a.gr + n * a.gr_sn;
    const int64_t nT = a.nT, nC = a.nC;

    // per-element pulse samples, fixed for the thread: element e = 3*t + c  (c: x, y, z)
    T rr[VW], ri[VW], px[VW], py[VW], pz[VW];
    int64_t tt[VW];
    int cc[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        const int64_t e = (e0 + j < L) ? e0 + j : L - 1;
        const int64_t t = e / 3;
        tt[j] = t; cc[j] = (int)(e - t * 3);
        px[j] = gr[t]; py[j] = gr[nT + t]; pz[j] = gr[2 * nT + t];
        rr[j] = NC1 ? rf[t] : T(0);
        ri[j] = NC1 ? rf[nT + t] : T(0);
    }

    // Per-spin operands of the block's rows go through LDS once: a global load inside the row loop
    // would need s_waitcnt vmcnt(0), which on gfx9-family parts also waits for the previous row's
    // store to be acknowledged (vmcnt counts stores, in order) -- one store round trip per row.
    __shared__ T sp[K0_MAX_ROWS][8];     // lx, ly, lz, df/gamma, b1r, b1i
    for (int64_t i = threadIdx.x; i < s1 - s0; i += K0_THREADS) {
        const int64_t s = s0 + i, row = n * a.nM + s;
        sp[i][0] = a.loc[row * 3]; sp[i][1] = a.loc[row * 3 + 1]; sp[i][2] = a.loc[row * 3 + 2];
        sp[i][3] = a.df.p ? bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s) : T(0);
        sp[i][4] = (NC1 && a.b1) ? a.b1[row * 2] : T(1);
        sp[i][5] = (NC1 && a.b1) ? a.b1[row * 2 + 1] : T(0);
    }
    __syncthreads();
    if (e0 >= L) return;

    for (int64_t s = s0; s < s1; ++s) {
        const int64_t row = n * a.nM + s;
        const T* q = sp[s - s0];
        const T lx = q[0], ly = q[1], lz = q[2], delta = q[3];
        T o[VW];
        if (NC1) {
            const T br = q[4], bi = q[5];
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                T Bx = T(0), By = T(0);
                field_xy_acc<T>(br, bi, rr[j], ri[j], Bx, By);
                const T Bz = field_z<T>(px[j], py[j], pz[j], lx, ly, lz, delta);
                o[j] = cc[j] == 0 ? Bx : (cc[j] == 1 ? By : Bz);
            }
        } else {
            const T* b1 = a.b1 + row * 2 * nC;    // [2][nC]
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                if (cc[j] == 2) {
                    o[j] = field_z<T>(px[j], py[j], pz[j], lx, ly, lz, delta);
                } else {
                    const T* qr = rf + tt[j] * nC;
                    const T* qi = rf + (nT + tt[j]) * nC;
                    T Bx = T(0), By = T(0);
                    for (int64_t c = 0; c < nC; ++c)
                        field_xy_acc<T>(b1[c], b1[nC + c], qr[c], qi[c], Bx, By);
                    o[j] = cc[j] == 0 ? Bx : By;
                }
            }
        }
        T* dst = a.beff + row * L + e0;
        if (VW == V16<T>::N) {
            if (a.nt) __builtin_nontemporal_store(vec_pack(o), reinterpret_cast<typename V16<T>::type*>(dst));
            else *reinterpret_cast<typename V16<T>::type*>(dst) = vec_pack(o);
        } else {
#pragma unroll
            for (int j = 0; j < VW; ++j)
                if (e0 + j < L) dst[j] = o[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Adjoint of K0 w.r.t. rf, gr: deterministic two-pass reduction over spins.
// Pass 1: block (time tile, spin group, batch*coil): thread = one time point, loops over the
// group's spins in order; partial sums -> work[(sg, n, 5, nC', nT)].  Pass 2: fixed-order sum.
// Rows of `work` per (sg, n): [gr_x, gr_y, gr_z, rf_r[c]..., rf_i[c]...].
// ---------------------------------------------------------------------------------------------
constexpr int BWD_GROUP = 256;        // spins per LDS sub-block of the K0-adjoint pass 1

template <typename T>
struct BeffBwdArgs {
    const T* gB;      // (N, nM, nT, 3)
    const T* loc;     // (N, nM, 3)
    const T* b1;      // (N, nM, 2, nC) or null
    T* work;          // (nSG, N, 3 + 2 nC, nT)
    T* grf;           // (N, 2, nT, nC) or null
    T* ggr;           // (N, 3, nT) or null
    int64_t N, nM, nT, nC, nSG, spins_per_group;
};

// Pass 1, single-coil fast path.  Thread = VW consecutive elements e = 3t + c of the (t, xyz) axis
// (one 16-B load per spin, fully coalesced), three running sums per element over the group's spins:
//   c = 0 or 1 (gBx / gBy):  (b1r*g, b1i*g, 0)          c = 2 (gBz):  (lx*g, ly*g, lz*g)
// written to work[(sg, n, k, e)], k = 0..2.  Pass 2 combines them per time point:
//   grad_gr[i][t] = A_i(t,2);  grad_rf_re[t] = A_0(t,0) + A_1(t,1);  grad_rf_im[t] = A_0(t,1) - A_1(t,0)
template <typename T, int VW>
__global__ __launch_bounds__(256) void k_rfgr2beff_bwd_p1v(BeffBwdArgs<T> a)
{
    const int64_t L = 3 * a.nT;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VW;
    const int64_t sg = blockIdx.y, n = blockIdx.z;
    const int64_t s0 = sg * a.spins_per_group;
    const int64_t s1 = (s0 + a.spins_per_group < a.nM) ? s0 + a.spins_per_group : a.nM;
    // the per-spin operands go through LDS, BWD_GROUP spins at a time, so that the row loop has
    // nothing but the gB stream in it and can keep U loads in flight per thread
    __shared__ T sp[BWD_GROUP][8];                     // lx, ly, lz, b1r, b1i
    const bool active = e0 < L;
    bool isz[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) isz[j] = ((e0 + j) % 3) == 2;
    T acc0[VW], acc1[VW], acc2[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) acc0[j] = acc1[j] = acc2[j] = T(0);
    constexpr int U = 8;
    auto accumulate = [&](const T* q, const T* g) {
        const T lx = q[0], ly = q[1], lz = q[2], br = q[3], bi = q[4];
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            acc0[j] += (isz[j] ? lx : br) * g[j];
            acc1[j] += (isz[j] ? ly : bi) * g[j];
            acc2[j] += (isz[j] ? lz : T(0)) * g[j];
        }
    };
    for (int64_t sb = s0; sb < s1; sb += BWD_GROUP) {
        const int64_t cnt = (s1 - sb < BWD_GROUP) ? s1 - sb : BWD_GROUP;
        __syncthreads();                               // previous sub-block consumed
        for (int64_t i = threadIdx.x; i < cnt; i += 256) {
            const int64_t row = n * a.nM + sb + i;
            sp[i][0] = a.loc[row * 3]; sp[i][1] = a.loc[row * 3 + 1]; sp[i][2] = a.loc[row * 3 + 2];
            sp[i][3] = a.b1 ? a.b1[row * 2] : T(1);
            sp[i][4] = a.b1 ? a.b1[row * 2 + 1] : T(0);
        }
        __syncthreads();
        if (!active) continue;
        const T* src0 = a.gB + (n * a.nM + sb) * L + e0;
        int64_t i = 0;
        if (VW == V16<T>::N) {
            for (; i + U <= cnt; i += U) {             // U rows' loads issued before the first use
                typename V16<T>::type v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = __builtin_nontemporal_load(
                        reinterpret_cast<const typename V16<T>::type*>(src0 + (i + u) * L));
#pragma unroll
                for (int u = 0; u < U; ++u) {          // same order as a plain loop: same sums
                    T g[VW];
                    vec_unpack(v[u], g);
                    accumulate(sp[i + u], g);
                }
            }
        }
        for (; i < cnt; ++i) {
            T g[VW];
            const T* src = src0 + i * L;
            if (VW == V16<T>::N) {
                vec_unpack(__builtin_nontemporal_load(
                               reinterpret_cast<const typename V16<T>::type*>(src)), g);
            } else {
#pragma unroll
                for (int j = 0; j < VW; ++j) g[j] = (e0 + j < L) ? src[j] : T(0);
            }
            accumulate(sp[i], g);
        }
    }
    if (!active) return;
    T* w = a.work + ((sg * a.N + n) * 3) * L;
#pragma unroll
    for (int j = 0; j < VW; ++j)
        if (e0 + j < L) { w[e0 + j] = acc0[j]; w[L + e0 + j] = acc1[j]; w[2 * L + e0 + j] = acc2[j]; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_rfgr2beff_bwd_p2v(BeffBwdArgs<T> a)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = blockIdx.z;
    if (t >= a.nT) return;
    const int64_t L = 3 * a.nT;
    T A[3][3];                                         // A[k][c]
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[kk][c] = T(0);
    for (int64_t sg = 0; sg < a.nSG; ++sg) {           // fixed order: deterministic
        const T* w = a.work + ((sg * a.N + n) * 3) * L + 3 * t;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int c = 0; c < 3; ++c) A[kk][c] += w[kk * L + c];
    }
    if (a.ggr) {
        a.ggr[(n * 3 + 0) * a.nT + t] = A[0][2];
        a.ggr[(n * 3 + 1) * a.nT + t] = A[1][2];
        a.ggr[(n * 3 + 2) * a.nT + t] = A[2][2];
    }
    if (a.grf) {                                        // nC == 1
        a.grf[(n * 2 + 0) * a.nT + t] = A[0][0] + A[1][1];
        a.grf[(n * 2 + 1) * a.nT + t] = A[0][1] - A[1][0];
    }
}

// Pass 1, any coil count (one block column per coil; strided scalar loads).
template <typename T>
__global__ __launch_bounds__(256) void k_rfgr2beff_bwd_p1(BeffBwdArgs<T> a)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t sg = blockIdx.y;
    const int64_t n = blockIdx.z / (a.nC + 1);
    const int64_t part = blockIdx.z % (a.nC + 1);     // 0: gradients, 1..nC: coil part-1
    if (t >= a.nT) return;
    const int64_t s0 = sg * a.spins_per_group;
    const int64_t s1 = (s0 + a.spins_per_group < a.nM) ? s0 + a.spins_per_group : a.nM;
    const int64_t K = 3 + 2 * a.nC;
    T* w = a.work + ((sg * a.N + n) * K) * a.nT;
    if (part == 0) {
        T ax = T(0), ay = T(0), az = T(0);
        for (int64_t s = s0; s < s1; ++s) {
            const int64_t row = n * a.nM + s;
            const T gz = a.gB[(row * a.nT + t) * 3 + 2];
            ax += a.loc[row * 3] * gz;
            ay += a.loc[row * 3 + 1] * gz;
            az += a.loc[row * 3 + 2] * gz;
        }
        w[0 * a.nT + t] = ax; w[1 * a.nT + t] = ay; w[2 * a.nT + t] = az;
    } else {
        const int64_t c = part - 1;
        T ar = T(0), ai = T(0);
        for (int64_t s = s0; s < s1; ++s) {
            const int64_t row = n * a.nM + s;
            const T gx = a.gB[(row * a.nT + t) * 3], gy = a.gB[(row * a.nT + t) * 3 + 1];
            T br = T(1), bi = T(0);
            if (a.b1) { br = a.b1[(row * 2) * a.nC + c]; bi = a.b1[(row * 2 + 1) * a.nC + c]; }
            ar += br * gx + bi * gy;
            ai += br * gy - bi * gx;
        }
        w[(3 + c) * a.nT + t] = ar;
        w[(3 + a.nC + c) * a.nT + t] = ai;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_rfgr2beff_bwd_p2(BeffBwdArgs<T> a)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t kk = blockIdx.y;       // row of the (3 + 2 nC) partial rows
    const int64_t n = blockIdx.z;
    if (t >= a.nT) return;
    const int64_t K = 3 + 2 * a.nC;
    T acc = T(0);
    for (int64_t sg = 0; sg < a.nSG; ++sg) acc += a.work[((sg * a.N + n) * K + kk) * a.nT + t];
    if (kk < 3) {
        if (a.ggr) a.ggr[(n * 3 + kk) * a.nT + t] = acc;
    } else if (a.grf) {
        const int64_t c = (kk - 3) % a.nC, ri = (kk - 3) / a.nC;
        a.grf[((n * 2 + ri) * a.nT + t) * a.nC + c] = acc;
    }
}


// =============================================================================================
// K2: fused rf,gr -> Mo.  No Beff in HBM: the pulse sample of step t is wave-uniform (one block
// = one wave = 64 spins of ONE batch entry, so rf/gr addresses are scalar loads) and the lane's
// own loc / df/gamma / b1 sit in registers.  The field is assembled exactly as K0 rounds it
// (B first, then g*B) so that K2 == K1(K0(.)) bit for bit.  VALU-bound, not HBM-bound.
// =============================================================================================
template <typename T>
struct FusedArgs {
    const T* Mi;
    const T* rf;  int64_t rf_sn;
    const T* gr;  int64_t gr_sn;
    const T* loc;
    Bc df, gam;
    const T* b1;
    Bc g, E1, E2;
    const void* E1m1;
    T* Mo;
    T* Mck;  int64_t ck_every;
    int64_t N, nM, nT, nC;
};

// CK: write checkpoints (every ck_every steps, a multiple of the 8-step chunk).  Kept out of the
// plain instantiation so that its step loop contains no store: the pulse loads are then provably
// unclobbered and become (batched) scalar loads.
template <typename T, typename CT, bool NC1, bool CK, bool RELAX>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_fwd(FusedArgs<T> a)
{
    constexpr int NS = 8;
    const int lane = threadIdx.x;
    const int64_t n = blockIdx.y;
    const int64_t s_ = (int64_t)blockIdx.x * WAVE + lane;
    const bool valid = s_ < a.nM;
    const int64_t s = valid ? s_ : a.nM - 1;
    const int64_t row = n * a.nM + s;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);

    T mx = a.Mi[row * 3], my = a.Mi[row * 3 + 1], mz = a.Mi[row * 3 + 2];
    const T lx = a.loc[row * 3], ly = a.loc[row * 3 + 1], lz = a.loc[row * 3 + 2];
    T delta = T(0);
    if (a.df.p) delta = bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s);
    T br = T(1), bi = T(0);
    if (NC1 && a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }

    const int64_t nT = a.nT, nC = a.nC;
    const T* __restrict__ rfr = a.rf + n * a.rf_sn;          // [nT][nC]
    const T* __restrict__ rfi = rfr + nT * nC;
    const T* __restrict__ gx = a.gr + n * a.gr_sn;
    const T* __restrict__ gy = gx + nT;
    const T* __restrict__ gz = gy + nT;
    const T* b1 = a.b1 ? a.b1 + row * 2 * nC : nullptr;
    const int64_t rows = a.N * a.nM;

    auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
        Bx = T(0); By = T(0);
        if (NC1) {
            field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
        } else {
            for (int64_t c = 0; c < nC; ++c)
                field_xy_acc<T>(b1[c], b1[nC + c], rfr[t * nC + c], rfi[t * nC + c], Bx, By);
        }
        Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
    };

    int64_t t0 = 0;
    for (; t0 + NS <= nT; t0 += NS) {
        if (CK && (t0 % a.ck_every) == 0 && valid) {
            T* c = a.Mck + ((t0 / a.ck_every) * rows + row) * 3;
            c[0] = mx; c[1] = my; c[2] = mz;
        }
        T Bx[NS], By[NS], Bz[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) field(t0 + j, Bx[j], By[j], Bz[j]);
        Rot<T> r[NS];
        rot_prepare<T, CT, NS>(k, Bx, By, Bz, r);
#pragma unroll
        for (int j = 0; j < NS; ++j) rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
    }
    for (; t0 < nT; ++t0) {                                   // nT % 8 tail
        if (CK && (t0 % a.ck_every) == 0 && valid) {
            T* c = a.Mck + ((t0 / a.ck_every) * rows + row) * 3;
            c[0] = mx; c[1] = my; c[2] = mz;
        }
        T Bx[1], By[1], Bz[1];
        field(t0, Bx[0], By[0], Bz[0]);
        Rot<T> r[1];
        rot_prepare<T, CT, 1>(k, Bx, By, Bz, r);
        rot_apply<RELAX, T, CT>(k, r[0], mx, my, mz);
    }
    if (valid) { a.Mo[row * 3] = mx; a.Mo[row * 3 + 1] = my; a.Mo[row * 3 + 2] = mz; }
}


// =============================================================================================
// K2b: adjoint of the fused kernel -- grad_Mo -> grad_Mi, grad_rf, grad_gr without Beff, history
// or grad_Beff in HBM (single-coil rf).  K2 leaves a checkpoint of M every SEG = 16 steps.  A wave
// walks the segments of its 64 spins backwards; per segment it
//   1. recomputes the 16 pre-step states from the checkpoint into registers (the very states the
//      forward pass went through, so no inversion error),
//   2. sweeps the adjoint over the 16 steps, re-assembling the field on the fly,
//   3. reduces the five per-step contributions
//        gr_x,y,z += loc_{x,y,z} * gBz     rf_re += b1r*gBx + b1i*gBy     rf_im += b1r*gBy - b1i*gBx
//      over its 64 spins with an LDS transpose-sum (80 rows x 64 lanes, slot-swizzled: conflict-free
//      ds_read_b128), and adds the 80 sums into ITS OWN row of the workspace.
// Waves are persistent (grid.x = min(tiles, 2048)) and take tiles w, w+P, ... in order, so every
// workspace row is accumulated in a fixed order; a second pass sums the rows in fixed order:
// deterministic, no float atomics.
// =============================================================================================
constexpr int SEG = 16;                      // steps per checkpoint segment
// Reduction tile: 80 rows x 64 lanes, NO padding (20480 B = exactly 1/8 of a CU's LDS, so 8 waves
// = 2 per SIMD are resident; with a padded pitch of 68 it was 21760 B -> 7 per CU, SIMD load
// 2:2:2:1).  Conflict-free row reads come from an XOR swizzle of the 16-B slot index instead:
// element (row, lane) lives in slot (lane/4) ^ (row & 15).
constexpr int RED_PITCH = WAVE;
constexpr int64_t K2B_MAX_WAVES = 256 * 8;   // resident waves: 8 per CU
__device__ __forceinline__ int red_idx(int row, int l)
{
    return row * RED_PITCH + ((((l >> 2) ^ (row & 15)) << 2) | (l & 3));
}

template <typename T>
struct FusedBwdArgs {
    const T* Mck;                    // (nT/SEG, N*nM, 3)
    const T* rf;  int64_t rf_sn;
    const T* gr;  int64_t gr_sn;
    const T* loc;
    Bc df, gam;
    const T* b1;                     // (N, nM, 2) or null
    Bc g, E1, E2;
    const void* E1m1;
    const T* gMo;
    T* gMi;                          // may be null
    T* work;                         // (P, N, 5, nT)
    int64_t N, nM, nT, P;
};

template <typename T, typename CT, bool RELAX>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_bwd(FusedBwdArgs<T> a)
{
    __shared__ __attribute__((aligned(16))) T red[5 * SEG * RED_PITCH];
    const int lane = threadIdx.x;
    const int64_t w = blockIdx.x, n = blockIdx.y;
    const int64_t nT = a.nT, rows = a.N * a.nM;
    const int64_t ntiles = (a.nM + WAVE - 1) / WAVE;
    const T* __restrict__ rfr = a.rf + n * a.rf_sn;
    const T* __restrict__ rfi = rfr + nT;
    const T* __restrict__ gx = a.gr + n * a.gr_sn;
    const T* __restrict__ gy = gx + nT;
    const T* __restrict__ gz = gy + nT;
    T* wsrow = a.work + ((w * a.N + n) * 5) * nT;
    bool first = true;

    for (int64_t tile = w; tile < ntiles; tile += a.P) {
        const int64_t s_ = tile * WAVE + lane;
        const bool valid = s_ < a.nM;
        const int64_t s = valid ? s_ : a.nM - 1;
        const int64_t row = n * a.nM + s;
        const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
        const T lx = a.loc[row * 3], ly = a.loc[row * 3 + 1], lz = a.loc[row * 3 + 2];
        T delta = T(0);
        if (a.df.p) delta = bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s);
        T br = T(1), bi = T(0);
        if (a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }
        const T vmask = valid ? T(1) : T(0);
        T hx = a.gMo[row * 3], hy = a.gMo[row * 3 + 1], hz = a.gMo[row * 3 + 2];

        auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
            Bx = T(0); By = T(0);
            field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
            Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
        };

        for (int64_t seg = nT / SEG - 1; seg >= 0; --seg) {
            const int64_t t0 = seg * SEG;
            const T* ck = a.Mck + (seg * rows + row) * 3;
            T mx = ck[0], my = ck[1], mz = ck[2];
            // 1. forward recompute, keeping the state before each step
            T M0[SEG], M1[SEG], M2[SEG];
#pragma unroll
            for (int sb = 0; sb < SEG / 4; ++sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                Rot<T> r[4];
                rot_prepare<T, CT, 4>(k, Bx, By, Bz, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    M0[sb * 4 + j] = mx; M1[sb * 4 + j] = my; M2[sb * 4 + j] = mz;
                    rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
                }
            }
            // 2. adjoint sweep, contributions to LDS
#pragma unroll
            for (int sb = SEG / 4 - 1; sb >= 0; --sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                RotAdj<T> ra[4];
                rot_prepare_adj<T, CT, 4>(k, Bx, By, Bz, ra);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int st = sb * 4 + j;
                    T g0, g1, g2;
                    rot_apply_adj<RELAX, T, CT>(k, ra[j], M0[st], M1[st], M2[st], hx, hy, hz,
                                                g0, g1, g2);
                    g0 *= vmask; g1 *= vmask; g2 *= vmask;
                    red[red_idx(0 * SEG + st, lane)] = lx * g2;
                    red[red_idx(1 * SEG + st, lane)] = ly * g2;
                    red[red_idx(2 * SEG + st, lane)] = lz * g2;
                    red[red_idx(3 * SEG + st, lane)] = br * g0 + bi * g1;
                    red[red_idx(4 * SEG + st, lane)] = br * g1 - bi * g0;
                }
            }
            __syncthreads();
            // 3. 80 row sums: lanes 0..63 take rows 0..63, lanes 0..15 rows 64..79
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int rrow = pass * WAVE + lane;
                if (rrow < 5 * SEG) {
                    T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);  // 4 chains for ILP; fixed order
#pragma unroll
                    for (int i = 0; i < WAVE; i += 4) {            // logical lanes i..i+3: one slot
                        const T* q = red + red_idx(rrow, i);
                        p0 += q[0]; p1 += q[1]; p2 += q[2]; p3 += q[3];
                    }
                    const T acc = (p0 + p1) + (p2 + p3);
                    T* dst = wsrow + (rrow / SEG) * nT + t0 + (rrow % SEG);
                    *dst = first ? acc : (*dst + acc);
                }
            }
            __syncthreads();
        }
        if (valid && a.gMi) { a.gMi[row * 3] = hx; a.gMi[row * 3 + 1] = hy; a.gMi[row * 3 + 2] = hz; }
        first = false;
    }
}

// Pass 2: sum the P workspace rows per (n, quantity, t) in a fixed order.  Block = 32 time points
// x 8 row groups (group g takes rows g, g+8, ...: 128-B coalesced reads per row), then the eight
// partial sums are combined through LDS in group order -- deterministic, and nT/32 * 5 blocks
// instead of nT/256 * 5 (40 blocks at nT = 2048 took 0.45 ms for 73 MB).
constexpr int P2_T = 32, P2_G = 8;
template <typename T>
__global__ __launch_bounds__(P2_T * P2_G) void k_bloch_rfgr_bwd_p2(const T* work, T* grf, T* ggr,
                                                                   int64_t N, int64_t nT, int64_t P)
{
    __shared__ T part[P2_G][P2_T];
    const int tl = threadIdx.x % P2_T, g = threadIdx.x / P2_T;
    const int64_t t = (int64_t)blockIdx.x * P2_T + tl;
    const int64_t q = blockIdx.y, n = blockIdx.z;
    T acc = T(0);
    if (t < nT)
        for (int64_t w = g; w < P; w += P2_G) acc += work[((w * N + n) * 5 + q) * nT + t];
    part[g][tl] = acc;
    __syncthreads();
    if (g != 0 || t >= nT) return;
    T sum = part[0][tl];
#pragma unroll
    for (int i = 1; i < P2_G; ++i) sum += part[i][tl];
    if (q < 3) { if (ggr) ggr[(n * 3 + q) * nT + t] = sum; }
    else if (grf) grf[(n * 2 + (q - 3)) * nT + t] = sum;
}


// =============================================================================================
// freeprec: free precession + relaxation for a duration `dur` -- mrphy.sims.FreePrec
// (reference sims.py:318-421; oracle form slowsims.py:134-174).  One thread per spin:
//   phi = -2 pi df dur (positive off-resonance dephases clockwise, sims.py:348-349)
//   Mxy <- R_z(phi) Mxy;   Mxy *= E2;   Mz <- Mz E1 - expm1(-dur/T1)        (sims.py:353-371)
// The adjoint (sims.py:400-419) is the transposed map applied to grad_Mo; it recomputes phi, E1, E2
// instead of saving five tensors.  DIR = +1 forward, -1 adjoint.
// =============================================================================================
__device__ __forceinline__ float  exp_(float a)   { return expf(a); }
__device__ __forceinline__ double exp_(double a)  { return exp(a); }
__device__ __forceinline__ float  expm1_(float a)  { return expm1f(a); }
__device__ __forceinline__ double expm1_(double a) { return expm1(a); }
__device__ __forceinline__ void sincos_full(float a, float* s, float* c)   { sincosf(a, s, c); }
__device__ __forceinline__ void sincos_full(double a, double* s, double* c) { sincos(a, s, c); }

struct FreePrecArgs {
    const void* Mi; void* Mo;
    const void* dur; int64_t dur_sn;      // (N|1,)
    Bc T1, T2, df;                        // T1.p == null: no relaxation; df.p == null: no precession
    int64_t rows, nM;
};

template <typename T, int DIR>
__global__ __launch_bounds__(256) void k_freeprec(FreePrecArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.rows) return;
    const int64_t n = r / a.nM, s = r % a.nM;
    const T* mi = reinterpret_cast<const T*>(a.Mi) + r * 3;
    T x = mi[0], y = mi[1], z = mi[2];
    const T dur = reinterpret_cast<const T*>(a.dur)[n * a.dur_sn];
    T cph = T(1), sph = T(0), e1 = T(1), e2 = T(1), e1m1 = T(0);
    if (a.df.p) {
        const T phi = T(-6.283185307179586476925) * bc_load<T>(a.df, n, s) * dur;
        sincos_full(phi, &sph, &cph);
    }
    if (a.T1.p) {
        const T a1 = -dur / bc_load<T>(a.T1, n, s), a2 = -dur / bc_load<T>(a.T2, n, s);
        e1 = exp_(a1); e1m1 = expm1_(a1); e2 = exp_(a2);
    }
    T ox, oy, oz;
    if (DIR > 0) {
        ox = (cph * x - sph * y) * e2;
        oy = (sph * x + cph * y) * e2;
        oz = z * e1 - e1m1;
    } else {
        const T gx = x * e2, gy = y * e2;
        ox = cph * gx + sph * gy;
        oy = cph * gy - sph * gx;
        oz = z * e1;
    }
    T* mo = reinterpret_cast<T*>(a.Mo) + r * 3;
    mo[0] = ox; mo[1] = oy; mo[2] = oz;
}


// =============================================================================================
// Pulse.interpT, linear (reference mobjs.py:177-220: numpy + scipy.interpolate.interp1d on the
// host).  The resampling grid depends only on (nT, dt_old, dt_new): the host supplies, per output
// sample j, lo[j] (index into the zero-prepended source, mobjs.py:204-207), w[j] = t_new - t_lo and
// dx[j] = t_hi - t_lo in fp64; the waveform itself never leaves the device.  Arithmetic as scipy's
// interp1d._call_linear: (y_hi - y_lo) in the data type, slope and product in fp64.
//   fwd: y_new[ch, j] = ((y_hi - y_lo)/dx[j]) * w[j] + y_lo
//   bwd: the transposed map in gather form (deterministic).
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_interp_lin_fwd(const T* y, T* out, const int* lo,
                                                        const double* w, const double* dx,
                                                        int64_t nch, int64_t nTo, int64_t nTn)
{
#pragma clang fp contract(off)                          // numpy rounds the product, then the sum
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ch = blockIdx.y;
    if (j >= nTn) return;
    const int l = lo[j];
    const T* row = y + ch * nTo;
    const T ylo = l == 0 ? T(0) : row[l - 1];          // sample 0 of the source is the prepended 0
    const T yhi = row[l];
    const T d = yhi - ylo;
    const double slope = double(d) / dx[j];
    const double prod = slope * w[j];
    out[ch * nTn + j] = T(prod + double(ylo));
}

// first j in [0, n) with lo[j] >= v  (lo is non-decreasing: a resampling grid)
__device__ __forceinline__ int64_t lower_bound_lo(const int* lo, int64_t n, int v)
{
    int64_t a = 0, b = n;
    while (a < b) {
        const int64_t m = (a + b) >> 1;
        if (lo[m] < v) a = m + 1; else b = m;
    }
    return a;
}

// Gather form of the transposed map: source sample i (row index i of y, i.e. l - 1 = i or l = i)
// receives  go[j] * a_j  from the outputs with lo[j] == i  and  go[j] * (1 - a_j)  from those with
// lo[j] == i + 1, a_j = w[j]/dx[j].  One thread per (channel, i), contributions added in j order
// with the rounding of a sequential scatter -- the same bits, nTo-fold parallel.
template <typename T>
__global__ __launch_bounds__(256) void k_interp_lin_bwd(const T* gout, T* gy, const int* lo,
                                                        const double* w, const double* dx,
                                                        int64_t nch, int64_t nTo, int64_t nTn)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ch = blockIdx.y;
    if (i >= nTo) return;
    const int64_t j0 = lower_bound_lo(lo, nTn, (int)i);
    const int64_t j1 = lower_bound_lo(lo, nTn, (int)i + 1);
    const int64_t j2 = lower_bound_lo(lo, nTn, (int)i + 2);
    const T* go = gout + ch * nTn;
    T acc = T(0);
    for (int64_t j = j0; j < j1; ++j) acc = T(double(acc) + double(go[j]) * (w[j] / dx[j]));
    for (int64_t j = j1; j < j2; ++j) acc = T(double(acc) + double(go[j]) * (1.0 - w[j] / dx[j]));
    gy[ch * nTo + i] = acc;
}

// =============================================================================================
// beff2uphi / uphirot: the two elementwise helpers of the reference's 1-step form.
// =============================================================================================
template <typename T, typename CT>
__global__ __launch_bounds__(256) void k_beff2uphi(const T* b, Bc g, T* U, T* Phi, int64_t rows,
                                                   int64_t nM)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const T x = b[r * 3], y = b[r * 3 + 1], z = b[r * 3 + 2];
    const T nrm = sqrt_(x * x + y * y + z * z);
    const T d = nrm > T(1e-12) ? nrm : T(1e-12);         // F.normalize eps (beffective.py:35)
    U[r * 3] = x / d; U[r * 3 + 1] = y / d; U[r * 3 + 2] = z / d;
    Phi[r] = T(-(CT(nrm) * bc_load<CT>(g, r / nM, r % nM)));
}

template <typename T>
__global__ __launch_bounds__(256) void k_uphirot(const T* U, const T* Phi, const T* Vi, T* Vo,
                                                 int64_t rows, int64_t nV)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * nV) return;
    const int64_t r = i / nV, v = i % nV;
    const T ux = U[r * 3], uy = U[r * 3 + 1], uz = U[r * 3 + 2];
    T sp, cp;
    sincos_(Phi[r], &sp, &cp);
    const T* vi = Vi + r * 3 * nV + v;
    const T x = vi[0], y = vi[nV], z = vi[2 * nV];
    const T ud = (T(1) - cp) * (ux * x + uy * y + uz * z);
    T* vo = Vo + r * 3 * nV + v;
    vo[0]      = cp * x + ud * ux + sp * (uy * z - uz * y);
    vo[nV]     = cp * y + ud * uy + sp * (uz * x - ux * z);
    vo[2 * nV] = cp * z + ud * uz + sp * (ux * y - uy * x);
}

// ---------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------
inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return (int)e;
}

inline size_t tsize(int dtype) { return dtype == MRPHY_F64 ? 8 : 4; }
inline size_t csize(int dtype) { return dtype == MRPHY_F32 ? 4 : 8; }

constexpr int TC_FWD = 16;
constexpr int TC_BWD = 16;

// =============================================================================================
// beff2ab (beffective.py:40-104): Hargreaves' A (3x3) and B (3) of a whole pulse per spin, i.e.
// the step map M -> relax(rotate(M)) applied to the four columns of [I | 0]; the -(E1-1) offset
// of the relaxation acts on the B column only.  Same streaming of Beff as the chunked K1, same
// rot_prepare / rot_apply, so column j of A equals blochsim(e_j) with a zero offset bit for bit
// and B equals blochsim(0).  ~4x the arithmetic of K1 per byte: VALU-bound.
// =============================================================================================
template <typename T>
struct AbArgs {
    const T* Beff;
    T* A;                  // (rows, 3, 3): A[r][i][j], i = xyz component, j = column
    T* B;                  // (rows, 3)
    Bc g, E1, E2;
    const void* E1m1;
    int64_t rows, nM, nT;
    int vec_ok;
};

template <typename T, typename CT, int TC>
__global__ __launch_bounds__(WAVE) void k_beff2ab(AbArgs<T> a)
{
    using TL = Tile<T, TC>;
    using V = typename TL::V;
    constexpr int VE = TL::VE;
    __shared__ __attribute__((aligned(16))) T tile[TL::ELEMS];

    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
    SpinConst<T, CT> kl = k;
    kl.e1m1 = CT(0);                                 // the A columns: linear part only

    T cx[4] = {T(1), T(0), T(0), T(0)}, cy[4] = {T(0), T(1), T(0), T(0)},
      cz[4] = {T(0), T(0), T(1), T(0)};
    const int64_t rowlen = 3 * a.nT;
    int64_t t = 0;
    if (a.vec_ok) {
        const int64_t nfull = a.nT / TC;
        Stage<T, TC> st;
        if (nfull > 0) st = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, 0, lane);
        T* myrow = tile + lane * TL::PITCH;
        for (int64_t c = 0; c < nfull; ++c) {
            __syncthreads();
            chunk_to_lds<T, TC>(tile, st, lane);
            __syncthreads();
            if (c + 1 < nfull)
                st = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, (c + 1) * TC, lane);
#pragma unroll 1
            for (int tt = 0; tt < TC; tt += VE) {
                T bb[3 * VE];
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3), bb);
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3 + VE), bb + VE);
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3 + 2 * VE), bb + 2 * VE);
                T gBx[VE], gBy[VE], gBz[VE];
#pragma unroll
                for (int q = 0; q < VE; ++q) { gBx[q] = bb[3 * q]; gBy[q] = bb[3 * q + 1]; gBz[q] = bb[3 * q + 2]; }
                Rot<T> rr[VE];
                rot_prepare<T, CT, VE>(k, gBx, gBy, gBz, rr);
#pragma unroll
                for (int q = 0; q < VE; ++q) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) rot_apply<true, T, CT>(kl, rr[q], cx[j], cy[j], cz[j]);
                    rot_apply<true, T, CT>(k, rr[q], cx[3], cy[3], cz[3]);
                }
            }
        }
        t = nfull * TC;
    }
    const T* bp = a.Beff + rc * rowlen;
    for (; t < a.nT; ++t) {
        const T bx_[1] = {bp[t * 3]}, by_[1] = {bp[t * 3 + 1]}, bz_[1] = {bp[t * 3 + 2]};
        Rot<T> r1[1];
        rot_prepare<T, CT, 1>(k, bx_, by_, bz_, r1);
#pragma unroll
        for (int j = 0; j < 3; ++j) rot_apply<true, T, CT>(kl, r1[0], cx[j], cy[j], cz[j]);
        rot_apply<true, T, CT>(k, r1[0], cx[3], cy[3], cz[3]);
    }
    if (valid) {
        T* A = a.A + r * 9;
#pragma unroll
        for (int j = 0; j < 3; ++j) { A[j] = cx[j]; A[3 + j] = cy[j]; A[6 + j] = cz[j]; }
        a.B[r * 3] = cx[3]; a.B[r * 3 + 1] = cy[3]; a.B[r * 3 + 2] = cz[3];
    }
}

// blochsim_ab (slowsims.py:117-131): Mo = A M + B per spin, and its adjoint
//   gM = A^T g,  gA[i][j] = g_i M_j,  (gB = g: the caller aliases it).
template <typename T>
__global__ __launch_bounds__(256) void k_ab_apply(const T* __restrict__ M, const T* __restrict__ A,
                                                  const T* __restrict__ B, T* __restrict__ Mo,
                                                  int64_t rows)
{
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const T mx = M[r * 3], my = M[r * 3 + 1], mz = M[r * 3 + 2];
    const T* a = A + r * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        Mo[r * 3 + i] = fma_(a[3 * i + 2], mz, fma_(a[3 * i + 1], my, a[3 * i] * mx)) + B[r * 3 + i];
}

template <typename T>
__global__ __launch_bounds__(256) void k_ab_apply_bwd(const T* __restrict__ M,
                                                      const T* __restrict__ A,
                                                      const T* __restrict__ g, T* __restrict__ gM,
                                                      T* __restrict__ gA, int64_t rows)
{
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const T g0 = g[r * 3], g1 = g[r * 3 + 1], g2 = g[r * 3 + 2];
    if (gM) {
        const T* a = A + r * 9;
#pragma unroll
        for (int j = 0; j < 3; ++j) gM[r * 3 + j] = fma_(a[6 + j], g2, fma_(a[3 + j], g1, a[j] * g0));
    }
    if (gA) {
        const T m[3] = {M[r * 3], M[r * 3 + 1], M[r * 3 + 2]};
        T* q = gA + r * 9;
#pragma unroll
        for (int j = 0; j < 3; ++j) { q[j] = g0 * m[j]; q[3 + j] = g1 * m[j]; q[6 + j] = g2 * m[j]; }
    }
}

// =============================================================================================
// Mask gather / scatter (mobjs.SpinArray.extract / embed, mobjs.py:512-553) through an index list
// built once per mask, and SpinCube._update_loc_ (mobjs.py:815-839).  Elements move as raw bits
// (E = 4- or 8-byte word), K = trailing elements per voxel.  grid.y = batch entry.
// =============================================================================================
template <typename E>
__global__ __launch_bounds__(256) void k_mask_extract(const E* __restrict__ v,
                                                      const int32_t* __restrict__ idx,
                                                      E* __restrict__ out, int64_t nV, int64_t nM,
                                                      int64_t K)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (e >= nM * K) return;
    const int64_t j = e / K, k = e - j * K;
    out[n * nM * K + e] = v[(n * nV + idx[j]) * K + k];
}

// fill: 0 = leave voxels outside the mask untouched, 1 = write `fillbits` there
template <typename E>
__global__ __launch_bounds__(256) void k_mask_embed(const E* __restrict__ v_,
                                                    const int32_t* __restrict__ inv,
                                                    E* __restrict__ out, int64_t nV, int64_t nM,
                                                    int64_t K, int fill, E fillbits)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (e >= nV * K) return;
    const int64_t p = e / K, k = e - p * K;
    const int32_t j = inv[p];
    if (j >= 0) out[n * nV * K + e] = v_[(n * nM + j) * K + k];
    else if (fill) out[n * nV * K + e] = fillbits;
}

// loc_[n, j, i] = fov[n, i] * ((c_i - dim_i / 2) / dim_i) + ofst[n, i],  c = unravel(idx[j]):
// the reference's arange/meshgrid/mask chain with the same three roundings (divide, multiply, add).
template <typename T>
__global__ __launch_bounds__(256) void k_cube_loc(const int32_t* __restrict__ idx,
                                                  const T* __restrict__ fov,
                                                  const T* __restrict__ ofst, T* __restrict__ loc_,
                                                  int64_t nM, int nx, int ny, int nz)
{
#pragma clang fp contract(off)
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (j >= nM) return;
    const int p = idx[j];
    const int iz = p % nz, iy = (p / nz) % ny, ix = p / (nz * ny);
    const T cx = (T(ix) - T(nx / 2)) / T(nx);
    const T cy = (T(iy) - T(ny / 2)) / T(ny);
    const T cz = (T(iz) - T(nz / 2)) / T(nz);
    T* q = loc_ + (n * nM + j) * 3;
    T px = fov[n * 3 + 0] * cx, py = fov[n * 3 + 1] * cy, pz = fov[n * 3 + 2] * cz;
    // -ffp-contract=fast lets the backend fuse this multiply with the add below whatever the
    // pragma says; the reference rounds twice (torch mul, then add).  Opaque pass-through:
    asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
    q[0] = px + ofst[n * 3 + 0];
    q[1] = py + ofst[n * 3 + 1];
    q[2] = pz + ofst[n * 3 + 2];
}

// development knob: MRPHY_K0_VARIANT = rows_per_block/8*10 + nt
inline int k0_variant()
{
    static const int v = [] { const char* e = getenv("MRPHY_K0_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}

inline int bwd_variant()
{
    static const int v = [] { const char* e = getenv("MRPHY_BWD_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}

// development knob: MRPHY_XCD_SWEEP=0 turns the XCD-contiguous tile order of the line kernels off
inline bool xcd_sweep()
{
    static const bool v = [] { const char* e = getenv("MRPHY_XCD_SWEEP"); return e ? atoi(e) != 0 : true; }();
    return v;
}

// development knob: MRPHY_FWD_VARIANT selects an alternative K1 build for A/B measurements
inline int fwd_variant()
{
    static const int v = [] { const char* e = getenv("MRPHY_FWD_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}

inline int64_t hist_elems(int64_t N, int64_t nM, int64_t nT)
{
    return ((N * nM + WAVE - 1) / WAVE) * nT * HIST_STEP;
}

// rows on 128-B lines and whole 32-step periods: what the line-granular kernels need
inline bool lines_shape_ok(const void* Beff, int64_t nT)
{
    return aligned_to(Beff, 128) && nT > 0 && (nT % 32 == 0) && (768 * nT < (int64_t)4294967295);
}

template <typename T, typename CT>
int run_fwd(const void* Mi, const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* Mo,
            void* Mpre, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    FwdArgs<T> a;
    a.Mi = (const T*)Mi; a.Beff = (const T*)Beff; a.Mo = (T*)Mo; a.Mpre = (T*)Mpre;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    // vector path of the chunked kernel: every row start and chunk start 16-B aligned
    a.vec_ok = aligned_to(Beff, 16) && ((3 * nT * sizeof(T)) % 16 == 0);
    a.per_xcd = 0;
    if (a.rows == 0) return 0;
    dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
    if constexpr (sizeof(T) == 4) {
        const int v = fwd_variant();
        if (lines_shape_ok(Beff, nT) && v != 16 && v != 32) {
            // XCD-contiguous tile order pays where the kernel writes (history: 10.07 -> 8.75 ms at
            // 128^3 x 1024); for the read-only forward it is neutral (15.70 vs 15.60 ms), left off.
            if (xcd_sweep() && Mpre) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; }
            // development knob MRPHY_FWD_VARIANT = OCC*100 + SPLIT*10 + NT selects a build.
            // measured on MI355X, 128^3 x 4096, no history (ms): 320 16.88 | 321 15.82 |
            // 330 17.14 | 331 15.72
#define MRPHY_L(OCC_, SP_, NT_, SV_)                                                             \
    do {                                                                                         \
        if (E1.p) hipLaunchKernelGGL((k_bloch_fwd_lines<CT, true, OCC_, SP_, NT_, SV_>), grid, \
                                     dim3(WAVE), 0, st, a);                                      \
        else      hipLaunchKernelGGL((k_bloch_fwd_lines<CT, false, OCC_, SP_, NT_, SV_>), grid, \
                                     dim3(WAVE), 0, st, a);                                      \
    } while (0)
            if (Mpre) {
                // with history: 3 waves/SIMD (the 4-wave build: 10.06 vs 8.72 ms at 128^3 x 1024)
                MRPHY_L(3, 3, true, true);
            } else {
                // 4 waves/SIMD (128 VGPRs; needs the 2-/3-step batches and the computed load
                // offsets to fit): a 64^3 grid -- or a 1/8 shard of 128^3 -- is 4096 tiles, exactly
                // the 4096 wave slots of the chip, instead of 1.33 rounds of 3072.
                // measured (ms): 64^3 x 4096: 331 2.48 | 441 2.29;  64^3 x 1024: 0.677 | 0.640;
                // 128^3 x 4096: 15.51 | 15.53 (431: 19.3 -- spills; 341: 15.86)
                switch (v) {
                case 320: MRPHY_L(3, 2, false, false); break;
                case 321: MRPHY_L(3, 2, true, false); break;
                case 330: MRPHY_L(3, 3, false, false); break;
                case 331: MRPHY_L(3, 3, true, false); break;
                case 341: MRPHY_L(3, 4, true, false); break;
                default:  MRPHY_L(4, 4, true, false); break;
                }
            }
#undef MRPHY_L
            return launch_status();
        }
    }
    if (Mpre)
        hipLaunchKernelGGL((k_bloch_fwd<T, CT, TC_FWD, true>), grid, dim3(WAVE), 0, st, a);
    else if (fwd_variant() == 32)
        hipLaunchKernelGGL((k_bloch_fwd<T, CT, 32, false>), grid, dim3(WAVE), 0, st, a);
    else
        hipLaunchKernelGGL((k_bloch_fwd<T, CT, TC_FWD, false>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

template <typename T, typename CT>
int run_bwd(const void* Mpre, const void* Beff, Bc g, Bc E1, Bc E2, const void* gMo, void* gMi,
            void* gBeff, int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    BwdArgs<T> a;
    a.Mpre = (const T*)Mpre; a.Beff = (const T*)Beff; a.gMo = (const T*)gMo;
    a.gMi = (T*)gMi; a.gBeff = (T*)gBeff;
    a.g = g; a.E1 = E1; a.E2 = E2;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, 16) && ((3 * nT * sizeof(T)) % 16 == 0) &&
               (!gBeff || aligned_to(gBeff, 16));
    a.per_xcd = 0;
    if (a.rows == 0) return 0;
    dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
    if constexpr (sizeof(T) == 4) {
        if (lines_shape_ok(Beff, nT) && (!gBeff || aligned_to(gBeff, 128)) &&
            fwd_variant() != 16) {
            if (xcd_sweep()) { a.per_xcd = (grid.x + 7) / 8; grid.x = a.per_xcd * 8; }
            const int occ = bwd_variant();
            // development knob MRPHY_BWD_VARIANT = waves per SIMD the build is bounded for (2, 3)
#define MRPHY_LB(OCC_)                                                                           \
    do {                                                                                         \
        if (E1.p) hipLaunchKernelGGL((k_bloch_bwd_lines<CT, true, OCC_, true>), grid,             \
                                     dim3(WAVE), 0, st, a);                                      \
        else      hipLaunchKernelGGL((k_bloch_bwd_lines<CT, false, OCC_, true>), grid,            \
                                     dim3(WAVE), 0, st, a);                                      \
    } while (0)
            // same-box A/B at 128^3 x 1024 (ms): history fetched in-batch 13.28 | one batch ahead:
            // 2 waves/SIMD 12.83, 3 waves/SIMD (36 B/lane of spills) 13.04
            if (occ == 3) MRPHY_LB(3); else MRPHY_LB(2);
#undef MRPHY_LB
            return launch_status();
        }
    }
    hipLaunchKernelGGL((k_bloch_bwd<T, CT, TC_BWD>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}

template <typename T>
int run_rfgr2beff(const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc,
                  Bc df, Bc gam, const void* b1, void* beff, int64_t N, int64_t nM, int64_t nT,
                  int64_t nC, hipStream_t st)
{
    BeffArgs<T> a;
    a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr; a.gr_sn = gr_sn;
    a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1; a.beff = (T*)beff;
    a.nM = nM; a.nT = nT; a.nC = nC;
    if (N * nM * nT == 0) return 0;
    // Block order matters more than anything else here.  Blocks are dealt round-robin to the 8 XCDs,
    // so block b works on tile (b % 8) * per_xcd + b / 8: every XCD (and its L2) sweeps its own
    // contiguous eighth of Beff, time tiles fastest, i.e. 8 linear write streams.
    // measured (128^3 x 4096, ms; v = order*1000 + rows/8*10 + nt):
    //   order 0 (spin tile fastest, grid y = time tile): 128 rows+nt 16.1-17.2 | 64 rows 16.6
    //   order 1 (time tile fastest, no XCD split):       128 rows+nt 19.3
    //   order 2 (XCD sweep): 8 rows+nt 16.4 | 16+nt 14.56 | 16 14.77 | 32+nt 14.90 | 48+nt 14.92
    //                        64 15.09 | 64+nt 17.45 | 128 15.20 | 128+nt 17.88
    a.rows_per_block = 16;
    a.nt = 1;
    int order = 2;
    if (k0_variant() > 0) {
        a.nt = (k0_variant() % 10) != 0; a.rows_per_block = (k0_variant() % 1000 / 10) * 8;
        order = k0_variant() / 1000;
    }
    if (a.rows_per_block < 8) a.rows_per_block = 64;
    if (a.rows_per_block > K0_MAX_ROWS) a.rows_per_block = K0_MAX_ROWS;
    constexpr int VWV = V16<T>::N;
    const int64_t L = 3 * nT;
    const bool vec = aligned_to(beff, 16) && ((L * sizeof(T)) % 16 == 0);
    const int vw = vec ? VWV : 1;
    const int64_t gy = (L + (int64_t)K0_THREADS * vw - 1) / ((int64_t)K0_THREADS * vw);
    if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
    const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
    a.gy = 0; a.nblk = 0; a.per_xcd = 0;
    if (order >= 1 && gx * gy < (int64_t(1) << 31) - 8) {
        a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
        grid = dim3(a.nblk, 1, (unsigned)N);
        if (order == 2) { a.per_xcd = (a.nblk + 7) / 8; grid.x = a.per_xcd * 8; }
    }
    const dim3 block(K0_THREADS);
    const bool nc1 = (nC == 1);
    if (vec) {
        if (nc1) hipLaunchKernelGGL((k_rfgr2beff<T, VWV, true>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_rfgr2beff<T, VWV, false>), grid, block, 0, st, a);
    } else {
        if (nc1) hipLaunchKernelGGL((k_rfgr2beff<T, 1, true>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_rfgr2beff<T, 1, false>), grid, block, 0, st, a);
    }
    return launch_status();
}

inline int64_t bwd_spin_groups(int64_t nM)
{
    int64_t g = (nM + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256) g = 256;
    return g;
}

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
        const bool vec = aligned_to(gB, 16) && ((L * sizeof(T)) % 16 == 0);
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
    hipLaunchKernelGGL((k_rfgr2beff_bwd_p1<T>), dim3(tx, (unsigned)a.nSG, (unsigned)(N * (nC + 1))),
                       dim3(256), 0, st, a);
    int e = launch_status();
    if (e) return e;
    hipLaunchKernelGGL((k_rfgr2beff_bwd_p2<T>), dim3(tx, (unsigned)(3 + 2 * nC), (unsigned)N),
                       dim3(256), 0, st, a);
    return launch_status();
}

template <typename T, typename CT>
int run_rfgr_fwd(const void* Mi, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, void* Mo, void* Mck, int64_t ck_every, int64_t N, int64_t nM,
                 int64_t nT, int64_t nC, hipStream_t st)
{
    FusedArgs<T> a;
    a.Mi = (const T*)Mi; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
    a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.Mo = (T*)Mo; a.Mck = (T*)Mck;
    a.ck_every = ck_every > 0 ? ck_every : 1;
    a.N = N; a.nM = nM; a.nT = nT; a.nC = nC;
    if (N * nM == 0) return 0;
    if (N > 65535) return MRPHY_EINVAL;
    const dim3 grid((unsigned)((nM + WAVE - 1) / WAVE), (unsigned)N);
#define MRPHY_K2(NC1_, CK_, RX_) \
    hipLaunchKernelGGL((k_bloch_rfgr_fwd<T, CT, NC1_, CK_, RX_>), grid, dim3(WAVE), 0, st, a)
    const bool nc1 = (nC == 1), ck = (Mck != nullptr), rx = (E1.p != nullptr);
    if (nc1) {
        if (ck) { if (rx) MRPHY_K2(true, true, true); else MRPHY_K2(true, true, false); }
        else    { if (rx) MRPHY_K2(true, false, true); else MRPHY_K2(true, false, false); }
    } else {
        if (ck) { if (rx) MRPHY_K2(false, true, true); else MRPHY_K2(false, true, false); }
        else    { if (rx) MRPHY_K2(false, false, true); else MRPHY_K2(false, false, false); }
    }
#undef MRPHY_K2
    return launch_status();
}

inline int64_t k2b_waves(int64_t nM)
{
    const int64_t tiles = (nM + WAVE - 1) / WAVE;
    return tiles < K2B_MAX_WAVES ? tiles : K2B_MAX_WAVES;
}

template <typename T, typename CT>
int run_rfgr_bwd(const void* Mck, const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn,
                 const void* loc, Bc df, Bc gam, const void* b1, Bc g, Bc E1, Bc E2,
                 const void* E1m1, const void* gMo, void* gMi, void* grf, void* ggr, void* work,
                 int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    FusedBwdArgs<T> a;
    a.Mck = (const T*)Mck; a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr;
    a.gr_sn = gr_sn; a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1; a.gMo = (const T*)gMo; a.gMi = (T*)gMi;
    a.work = (T*)work; a.N = N; a.nM = nM; a.nT = nT; a.P = k2b_waves(nM);
    if (N * nM * nT == 0) return 0;
    if (N > 65535) return MRPHY_EINVAL;
    const dim3 grid((unsigned)a.P, (unsigned)N);
    if (E1.p) hipLaunchKernelGGL((k_bloch_rfgr_bwd<T, CT, true>), grid, dim3(WAVE), 0, st, a);
    else      hipLaunchKernelGGL((k_bloch_rfgr_bwd<T, CT, false>), grid, dim3(WAVE), 0, st, a);
    int e = launch_status();
    if (e) return e;
    if (grf || ggr) {
        hipLaunchKernelGGL((k_bloch_rfgr_bwd_p2<T>),
                           dim3((unsigned)((nT + P2_T - 1) / P2_T), 5, (unsigned)N),
                           dim3(P2_T * P2_G), 0, st, (const T*)work, (T*)grf, (T*)ggr, N, nT, a.P);
        e = launch_status();
    }
    return e;
}

inline int check_common(int dtype, int64_t N, int64_t nM, int64_t nT)
{
    if (dtype != MRPHY_F32 && dtype != MRPHY_F64 && dtype != MRPHY_F32_C64) return MRPHY_EINVAL;
    if (N < 0 || nM < 0 || nT < 0) return MRPHY_EINVAL;
    return 0;
}

#define MRPHY_DISPATCH(dtype, CALL)                          \
    switch (dtype) {                                         \
    case MRPHY_F32:     { using T = float;  using CT = float;  return CALL; } \
    case MRPHY_F64:     { using T = double; using CT = double; return CALL; } \
    case MRPHY_F32_C64: { using T = float;  using CT = double; return CALL; } \
    default: return MRPHY_EINVAL;                            \
    }

template <typename T, typename CT>
int run_beff2ab(const void* Beff, Bc g, Bc E1, Bc E2, const void* E1m1, void* A, void* B,
                       int64_t N, int64_t nM, int64_t nT, hipStream_t st)
{
    AbArgs<T> a;
    a.Beff = (const T*)Beff; a.A = (T*)A; a.B = (T*)B;
    a.g = g; a.E1 = E1; a.E2 = E2; a.E1m1 = E1m1;
    a.rows = N * nM; a.nM = nM; a.nT = nT;
    a.vec_ok = aligned_to(Beff, 16) && ((3 * nT * sizeof(T)) % 16 == 0);
    const dim3 grid((unsigned)((a.rows + WAVE - 1) / WAVE));
    hipLaunchKernelGGL((k_beff2ab<T, CT, TC_FWD>), grid, dim3(WAVE), 0, st, a);
    return launch_status();
}


}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int mrphy_abi_version(void) { return MRPHY_ABI_VERSION; }

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
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (dtype == MRPHY_F32_C64) return MRPHY_EINVAL;      // K0 has no separate constant type
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
        return run_rfgr2beff<float>(rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, beff, N, nM, nT, nC, st);
    return run_rfgr2beff<double>(rf, rf_sn, gr, gr_sn, loc, bdf, bgam, b1, beff, N, nM, nT, nC, st);
}

size_t mrphy_rfgr2beff_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t nC)
{
    if (N <= 0 || nM <= 0 || nT <= 0 || nC < 1) return 0;
    const int64_t rows = (nC == 1) ? 9 : (3 + 2 * nC);   // single coil: 3 sums x 3nT elements
    return (size_t)(bwd_spin_groups(nM) * N * rows * nT) * tsize(dtype);
}

int mrphy_rfgr2beff_bwd(int dtype, const void* grad_beff, const void* loc, const void* b1,
                        void* grad_rf, void* grad_gr, void* work, size_t work_bytes, int64_t N,
                        int64_t nM, int64_t nT, int64_t nC, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (dtype == MRPHY_F32_C64 || nC < 1) return MRPHY_EINVAL;
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

int mrphy_blochsim_fwd(int dtype, const void* Mi, const void* Beff, const void* g, int64_t g_sn,
                       int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                       int64_t E2_sn, int64_t E2_sm, const void* E1m1, void* Mo, void* Mpre,
                       int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
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
    MRPHY_DISPATCH(dtype, (run_fwd<T, CT>(Mi, Beff, bg, b1, b2, E1m1, Mo, Mpre, N, nM, nT, st)));
}

int mrphy_blochsim_bwd(int dtype, const void* Mpre, const void* Beff, const void* g, int64_t g_sn,
                       int64_t g_sm, const void* E1, int64_t E1_sn, int64_t E1_sm, const void* E2,
                       int64_t E2_sn, int64_t E2_sm, const void* grad_Mo, void* grad_Mi,
                       void* grad_Beff, int64_t N, int64_t nM, int64_t nT, void* stream)
{
    if (int e = check_common(dtype, N, nM, nT)) return e;
    if (N * nM == 0) return 0;
    if (!g || !grad_Mo || (nT > 0 && (!Beff || !Mpre))) return MRPHY_EINVAL;
    if ((E1 == nullptr) != (E2 == nullptr)) return MRPHY_EINVAL;
    const Bc bg = {g, g_sn, g_sm}, b1 = {E1, E1_sn, E1_sm}, b2 = {E2, E2_sn, E2_sm};
    hipStream_t st = (hipStream_t)stream;
    MRPHY_DISPATCH(dtype, (run_bwd<T, CT>(Mpre, Beff, bg, b1, b2, grad_Mo, grad_Mi, grad_Beff,
                                          N, nM, nT, st)));
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

int mrphy_beff2uphi(int dtype, const void* b, const void* g, int64_t g_sn, int64_t g_sm, void* U,
                    void* Phi, int64_t N, int64_t nM, void* stream)
{
    if (int e = check_common(dtype, N, nM, 0)) return e;
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
    MRPHY_DISPATCH(dtype, (run_beff2ab<T, CT>(Beff, bg, b1, b2, E1m1, A, B, N, nM, nT, st)));
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
