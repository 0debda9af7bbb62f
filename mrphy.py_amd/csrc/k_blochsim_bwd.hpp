// k_blochsim_bwd.hpp -- K3 (blochsim adjoint): chunked and line-granular kernels
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

// =============================================================================================
// K3: blochsim backward.  Reads Beff and Mpre chunks (two tiles), sweeps time backwards,
// writes dL/dBeff in place into the Beff tile, stores it coalesced.
// =============================================================================================
template <typename T>
struct BwdArgs {
    HistParts hist;        // the history the forward wrote (same parts, same order)
    const T* Beff;
    const T* gMo;
    T* gMi;
    T* gBeff;
    Bc g, E1, E2;
    int64_t rows, nM, nT;
    int vec_ok;
    unsigned per_xcd;
    T* gC;                 // (rows, 4) [dL/dg, dL/dE1, dL/dE2, dL/dE1m1] per spin, or null (GC builds)
    MRPHY_STAMP_FIELD
};

// GC: also accumulate the gradients w.r.t. the per-spin constants (adj_const_accumulate) into a.gC.
template <typename T, typename CT, int TC, bool GC = false>
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
    adj_begin_rt<T, CT>(k, hx, hy, hz);
    const int64_t rowlen = 3 * a.nT;
    const int64_t nfull = a.vec_ok ? a.nT / TC : 0;
    const T* hp = hist_tile_base<T>(a.hist, (int64_t)blockIdx.x, a.nT) + lane;
    T acc[4] = {T(0), T(0), T(0), T(0)};

    // one adjoint step; GC builds also accumulate the constants' gradients
    auto step = [&](const RotAdj<T>& ra, T Bx, T By, T Bz, T m0, T m1, T m2, T& gx, T& gy, T& gz) {
        if constexpr (GC) {
            using R = typename CTr<CT>::reg;
            const T sx = hx, sy = hy, sz = hz;
            T dbx, dby, dbz;
            if (k.relax) rot_apply_adj_core<true, T, CT>(k, ra, m0, m1, m2, hx, hy, hz, dbx, dby, dbz);
            else         rot_apply_adj_core<false, T, CT>(k, ra, m0, m1, m2, hx, hy, hz, dbx, dby, dbz);
            gx = T(R(dbx) * k.g); gy = T(R(dby) * k.g); gz = T(R(dbz) * k.g);
            if (k.relax) adj_const_accumulate<true, T, CT>(ra, Bx, By, Bz, m0, m1, m2, sx, sy, sz, dbx, dby, dbz, acc);
            else         adj_const_accumulate<false, T, CT>(ra, Bx, By, Bz, m0, m1, m2, sx, sy, sz, dbx, dby, dbz, acc);
        } else {
            if (k.relax) rot_apply_adj<true, T, CT>(k, ra, m0, m1, m2, hx, hy, hz, gx, gy, gz);
            else         rot_apply_adj<false, T, CT>(k, ra, m0, m1, m2, hx, hy, hz, gx, gy, gz);
        }
    };

    // tail first (we run time backwards)
    {
        const T* bp = a.Beff + rc * rowlen;
        T* gp = a.gBeff ? a.gBeff + rc * rowlen : nullptr;
        for (int64_t t = a.nT - 1; t >= nfull * TC; --t) {
            T gx, gy, gz, m0, m1, m2;
            hist_load<T>(hp, t, m0, m1, m2);
            const T bx_[1] = {bp[t * 3]}, by_[1] = {bp[t * 3 + 1]}, bz_[1] = {bp[t * 3 + 2]};
            RotAdj<T> ra[1];
            rot_prepare_adj<T, CT, 1>(k, bx_, by_, bz_, ra);
            step(ra[0], bx_[0], by_[0], bz_[0], m0, m1, m2, gx, gy, gz);
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
                for (int q = VE - 1; q >= 0; --q)
                    step(ra[q], Bx[q], By[q], Bz[q], M0[q], M1[q], M2[q], gg[3 * q], gg[3 * q + 1],
                         gg[3 * q + 2]);
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
    adj_end_rt<T, CT>(k, hx, hy, hz);
    if (valid && a.gMi) { a.gMi[r * 3] = hx; a.gMi[r * 3 + 1] = hy; a.gMi[r * 3 + 2] = hz; }
    if constexpr (GC) {
        adj_const_finish<T, CT>(k, acc);
        if (valid && a.gC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a.gC[r * 4 + i] = acc[i];
        }
    }
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
template <typename T>
struct HistBatchT {
    T m0[HB_MAX], m1[HB_MAX], m2[HB_MAX];
};
using HistBatch = HistBatchT<float>;

template <int NA, typename T>
__device__ __forceinline__ void hist_fetch(const T* hp, int64_t th, HistBatchT<T>& h)
{
    static_assert(NA <= HB_MAX, "batch larger than HistBatch");
#pragma unroll
    for (int j = 0; j < NA; ++j) hist_load<T>(hp, th + j, h.m0[j], h.m1[j], h.m2[j]);
}

// NA steps (fields at q, history in h), time reversed; dL/dBeff replaces the fields in place.
template <bool RELAX, bool PIN, typename CT, int NA, typename T = float>
__device__ __forceinline__ void lines_adj(const SpinConst<T, CT>& k, T* q,
                                          const HistBatchT<T>& h, T& hx, T& hy, T& hz)
{
    T Bx[NA], By[NA], Bz[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { Bx[j] = q[3 * j]; By[j] = q[3 * j + 1]; Bz[j] = q[3 * j + 2]; }
    RotAdj<T> ra[NA];
    rot_prepare_adj<T, CT, NA>(k, Bx, By, Bz, ra);
#pragma unroll
    for (int j = NA - 1; j >= 0; --j) {
        T gx, gy, gz;
        rot_apply_adj<RELAX, T, CT>(k, ra[j], h.m0[j], h.m1[j], h.m2[j], hx, hy, hz, gx, gy, gz);
        q[3 * j] = gx; q[3 * j + 1] = gy; q[3 * j + 2] = gz;
    }
    if (PIN) pin_state(hx, hy, hz);
}

// NA steps from q plus, last in reversed time order, the straddling step whose field is
// (b0, b1, b2) and whose history is h[0]; its gradient is returned in (g0, g1, g2).
template <bool RELAX, bool PIN, typename CT, int NA, typename T = float>
__device__ __forceinline__ void lines_adj_carry(const SpinConst<T, CT>& k, T b0, T b1,
                                                T b2, T* q, const HistBatchT<T>& h,
                                                T& hx, T& hy, T& hz, T& g0,
                                                T& g1, T& g2)
{
    static_assert(NA + 1 <= HB_MAX, "batch larger than HistBatch");
    T Bx[NA + 1], By[NA + 1], Bz[NA + 1];
    Bx[0] = b0; By[0] = b1; Bz[0] = b2;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        Bx[j + 1] = q[3 * j]; By[j + 1] = q[3 * j + 1]; Bz[j + 1] = q[3 * j + 2];
    }
    RotAdj<T> ra[NA + 1];
    rot_prepare_adj<T, CT, NA + 1>(k, Bx, By, Bz, ra);
#pragma unroll
    for (int j = NA; j >= 1; --j) {
        T gx, gy, gz;
        rot_apply_adj<RELAX, T, CT>(k, ra[j], h.m0[j], h.m1[j], h.m2[j], hx, hy, hz, gx, gy, gz);
        q[3 * (j - 1)] = gx; q[3 * (j - 1) + 1] = gy; q[3 * (j - 1) + 2] = gz;
    }
    rot_apply_adj<RELAX, T, CT>(k, ra[0], h.m0[0], h.m1[0], h.m2[0], hx, hy, hz, g0, g1, g2);
    if (PIN) pin_state(hx, hy, hz);
}

template <typename CT, bool RELAX, int OCC, bool NT, bool PIN = false>
__global__ __launch_bounds__(WAVE, OCC) void k_bloch_bwd_lines(BwdArgs<float> a)
{
    using T = float;
    constexpr int PF = 32;
    constexpr int PITCH = PF + 4;
    __shared__ __attribute__((aligned(16))) T tile[WAVE * PITCH];

    const int lane = threadIdx.x;
    const int64_t tile_id = xcd_tile(a.per_xcd);
    if (tile_id * WAVE >= a.rows) return;
    MRPHY_STAMP_BEGIN()
    const int64_t row0 = tile_id * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, nullptr, n, s);
    T hx = a.gMo[rc * 3 + 0], hy = a.gMo[rc * 3 + 1], hz = a.gMo[rc * 3 + 2];
    adj_begin<RELAX, T, CT>(k, hx, hy, hz);

    const int64_t rowlen = 3 * a.nT;
    const int64_t npieces = rowlen / PF;                   // multiple of 3
    const int frow = lane >> 3, fcol = (lane & 7) * 4;
    const T* __restrict__ base = a.Beff + row0 * rowlen;
    T* __restrict__ obase = a.gBeff ? a.gBeff + row0 * rowlen : nullptr;
    const int64_t last = a.rows - 1 - row0;
    // byte offset of load / store i = min(off0 + i * ostride, olim), as in the forward kernel: two
    // VGPRs instead of eight precomputed offsets (o0 laundered per use, or LICM hoists all eight
    // back into registers); rows past the end of the last tile are clamped for the loads and
    // skipped for the stores
    const unsigned ostride = (unsigned)(8 * rowlen * sizeof(T));
    const unsigned off0 = (unsigned)(((frow < last ? frow : last) * rowlen + fcol) * sizeof(T));
    const unsigned olim = (unsigned)(((last < 63 ? last : 63) * rowlen + fcol) * sizeof(T));
    const int lastrow = (int)(last < 63 ? last : 63);
#define MRPHY_OFF(i) (min(o0 + (unsigned)(i) * ostride, olim))
    T* wr = tile + frow * PITCH + fcol;
    T* my_ = tile + lane * PITCH;
    const T* hp = hist_tile_base<T>(a.hist, tile_id, a.nT) + lane;
    const T* rowp = a.Beff + rc * rowlen;                  // this lane's own row, for the tails

    f32x4 st[8];
#define MRPHY_FETCH(p)                                                                     \
    { unsigned o0 = off0; asm volatile("" : "+v"(o0));                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        st[i] = ldv<NT>(reinterpret_cast<const f32x4*>(                                     \
            reinterpret_cast<const char*>(base + (p) * PF) + MRPHY_OFF(i))); }
#define MRPHY_STAGE()                                                                      \
    __syncthreads();                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        *reinterpret_cast<f32x4*>(wr + i * 8 * PITCH) = st[i];                             \
    __syncthreads();
#define MRPHY_STORE(p)                                                                     \
    if (obase) {                                                                           \
        __syncthreads();                                                                   \
        unsigned o0 = off0; asm volatile("" : "+v"(o0));                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                    \
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + i * 8 * PITCH);           \
            if (frow + 8 * i <= lastrow)                                                   \
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(                   \
                    reinterpret_cast<char*>(obase + (p) * PF) + MRPHY_OFF(i)));            \
        }                                                                                  \
    }
    // Batches of a 32-step period in processing order (time reversed), steps [first, count]:
    //   piece p+2: [29,3] [25,4] [21,4: carry 21 + 22..24]    piece p+1: [18,3] [14,4] [10,4: carry]
    //   piece p  : [7,3] [4,3] [0,4]
    // H0/H1 alternate: each batch issues the history loads of the NEXT one before it computes.
    // First in a turn the order is: stage, next batch's history, next piece's Beff, compute.
#define LA(NA_, Q_, H_) lines_adj<RELAX, PIN, CT, NA_>(k, my_ + (Q_), H_, hx, hy, hz)
    HistBatch H0, H1;
    if (npieces > 0) {
        MRPHY_FETCH(npieces - 1)
        hist_fetch<3, float>(hp, (npieces / 3 - 1) * 32 + 29, H0);
    }
    MRPHY_PRIO_INIT(a)
    for (int64_t p = npieces - 3; p >= 0; p -= 3) {
        const int64_t t0 = (p / 3) * 32;
        MRPHY_PRIO_TICK(a, p / 3)
        T g0, g1, g2;
        // ---- piece p+2: floats 64..95 of the period.  steps 31..22 (from float 2), then the
        //      straddling step 21 = (tail float 63 | floats 0, 1)
        const T tl63 = rowp[(p + 2) * PF - 1];
        MRPHY_STAGE()
        hist_fetch<4, float>(hp, t0 + 25, H1);
        MRPHY_FETCH(p + 1)
        LA(3, 23, H0);
        hist_fetch<4, float>(hp, t0 + 21, H0);
        LA(4, 11, H1);
        hist_fetch<3, float>(hp, t0 + 18, H1);
        lines_adj_carry<RELAX, PIN, CT, 3>(k, tl63, my_[0], my_[1], my_ + 2, H0, hx, hy, hz, g0, g1, g2);
        my_[0] = g1; my_[1] = g2;
        T cg31 = g0;                                       // -> float 31 of piece p+1
        MRPHY_STORE(p + 2)
        // ---- piece p+1: floats 32..63.  steps 20..11 (from float 1), straddling step 10 =
        //      (tail floats 30, 31 | float 0)
        const T tl30 = rowp[(p + 1) * PF - 2], tl31 = rowp[(p + 1) * PF - 1];
        MRPHY_STAGE()
        hist_fetch<4, float>(hp, t0 + 14, H0);
        MRPHY_FETCH(p)
        my_[31] = cg31;
        LA(3, 22, H1);
        hist_fetch<4, float>(hp, t0 + 10, H1);
        LA(4, 10, H0);
        hist_fetch<3, float>(hp, t0 + 7, H0);
        lines_adj_carry<RELAX, PIN, CT, 3>(k, tl30, tl31, my_[0], my_ + 1, H1, hx, hy, hz, g0, g1, g2);
        my_[0] = g2;
        MRPHY_STORE(p + 1)
        // ---- piece p: floats 0..31.  floats 30, 31 <- carried gradient of step 10; steps 9..0
        MRPHY_STAGE()
        hist_fetch<3, float>(hp, t0 + 4, H1);
        if (p > 0) { MRPHY_FETCH(p - 1) }
        my_[30] = g0; my_[31] = g1;
        LA(3, 21, H0);
        hist_fetch<4, float>(hp, t0 + 0, H0);
        LA(3, 12, H1);
        if (p > 0) hist_fetch<3, float>(hp, t0 - 32 + 29, H1);    // first batch of the next period
        LA(4, 0, H0);
        H0 = H1;
        MRPHY_STORE(p)
    }
#undef MRPHY_FETCH
#undef MRPHY_STAGE
#undef MRPHY_STORE
#undef MRPHY_OFF
#undef LA
    adj_end<RELAX, T, CT>(k, hx, hy, hz);
    if (valid && a.gMi) { a.gMi[r * 3] = hx; a.gMi[r * 3 + 1] = hy; a.gMi[r * 3 + 2] = hz; }
    MRPHY_STAMP_END(a, tile_id)
}


// =============================================================================================
// K3, line-granular, DOUBLE precision (round 4; see k_bloch_fwd_lines_f64 for the layout: a piece = 16
// doubles, period 3 pieces = 16 steps).  Time runs backwards, pieces 2, 1, 0 of a period:
//   piece p+2 (doubles 32-47 of the period): steps 15..11 (from double 1), then the straddling step
//            10 = (tail doubles 14, 15 of piece p+1 | double 0); its gradient components 0, 1 go to
//            doubles 14, 15 of piece p+1 (two carry registers), component 2 to double 0 here;
//   piece p+1: doubles 14, 15 <- carried; steps 9..6 (from double 2); straddling step 5 = (tail double
//            15 of piece p | doubles 0, 1): component 0 carried to double 15 of piece p;
//   piece p:   double 15 <- carried; steps 4..0.
// Batches of at most 2 steps (the fp64 RotAdj is 16 VGPRs per step); the history of each batch is fetched one
// batch ahead, as in the fp32 kernel: 244-248 VGPRs, two waves per SIMD, no scratch (the chunked fp64 adjoint:
// 430-456 VGPRs, one wave).
// =============================================================================================
template <typename CT, bool RELAX, int OCC, bool NT, bool PIN>
__global__ __launch_bounds__(WAVE, OCC) void k_bloch_bwd_lines_f64(BwdArgs<double> a)
{
    using T = double;
    constexpr int PF = 16;
    constexpr int PITCH = PF + 2;
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
    adj_begin<RELAX, T, CT>(k, hx, hy, hz);

    const int64_t rowlen = 3 * a.nT;
    const int64_t npieces = rowlen / PF;                   // multiple of 3
    const int frow = lane >> 3, fcol = (lane & 7) * 2;
    const T* __restrict__ base = a.Beff + row0 * rowlen;
    T* __restrict__ obase = a.gBeff ? a.gBeff + row0 * rowlen : nullptr;
    const int64_t last = a.rows - 1 - row0;
    const unsigned ostride = (unsigned)(8 * rowlen * sizeof(T));
    const unsigned off0 = (unsigned)(((frow < last ? frow : last) * rowlen + fcol) * sizeof(T));
    const unsigned olim = (unsigned)(((last < 63 ? last : 63) * rowlen + fcol) * sizeof(T));
    const int lastrow = (int)(last < 63 ? last : 63);
#define MRPHY_OFF(i) (min(o0 + (unsigned)(i) * ostride, olim))
    T* wr = tile + frow * PITCH + fcol;
    T* my_ = tile + lane * PITCH;
    const T* hp = hist_tile_base<T>(a.hist, tile_id, a.nT) + lane;
    const T* rowp = a.Beff + rc * rowlen;                  // this lane's own row, for the tails

    f64x2 st[8];
#define MRPHY_FETCH(p)                                                                     \
    { unsigned o0 = off0; asm volatile("" : "+v"(o0));                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        st[i] = ldv<NT>(reinterpret_cast<const f64x2*>(                                     \
            reinterpret_cast<const char*>(base + (p) * PF) + MRPHY_OFF(i))); }
#define MRPHY_STAGE()                                                                      \
    __syncthreads();                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        *reinterpret_cast<f64x2*>(wr + i * 8 * PITCH) = st[i];                             \
    __syncthreads();
#define MRPHY_STORE(p)                                                                     \
    if (obase) {                                                                           \
        __syncthreads();                                                                   \
        unsigned o0 = off0; asm volatile("" : "+v"(o0));                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                    \
            const f64x2 v = *reinterpret_cast<const f64x2*>(wr + i * 8 * PITCH);           \
            if (frow + 8 * i <= lastrow)                                                   \
                __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(                   \
                    reinterpret_cast<char*>(obase + (p) * PF) + MRPHY_OFF(i)));            \
        }                                                                                  \
    }
    // Batches of a 16-step period in processing order, steps [first, count]:
    //   piece p+2: [13,3] [10,3: carry 10 + 11, 12]   piece p+1: [8,2] [5,3: carry 5 + 6, 7]   piece p: [3,2] [0,3]
#define LA(NA_, Q_, H_) lines_adj<RELAX, PIN, CT, NA_, T>(k, my_ + (Q_), H_, hx, hy, hz)
    // Batches of at most two steps, processing order, steps [first, count]:
    //   piece p+2: [14,2] [12,2] [10,2: carry 10 + 11]   piece p+1: [8,2] [7,1] [5,2: carry 5 + 6]   piece p: [3,2] [1,2] [0,1]
    HistBatchT<T> H0, H1;
    if (npieces > 0) {
        MRPHY_FETCH(npieces - 1)
        hist_fetch<2, T>(hp, (npieces / 3 - 1) * 16 + 14, H0);
    }
    for (int64_t p = npieces - 3; p >= 0; p -= 3) {
        const int64_t t0 = (p / 3) * 16;
        T g0, g1, g2;
        // ---- piece p+2
        const T tl14 = rowp[(p + 2) * PF - 2], tl15 = rowp[(p + 2) * PF - 1];
        MRPHY_STAGE()
        hist_fetch<2, T>(hp, t0 + 12, H1);
        MRPHY_FETCH(p + 1)
        LA(2, 10, H0);                                     // steps 14, 15
        hist_fetch<2, T>(hp, t0 + 10, H0);
        LA(2, 4, H1);                                      // steps 12, 13
        hist_fetch<2, T>(hp, t0 + 8, H1);
        lines_adj_carry<RELAX, PIN, CT, 1, T>(k, tl14, tl15, my_[0], my_ + 1, H0, hx, hy, hz, g0, g1, g2);
        my_[0] = g2;
        const T cg14 = g0, cg15 = g1;
        MRPHY_STORE(p + 2)
        // ---- piece p+1
        const T tp15 = rowp[(p + 1) * PF - 1];
        MRPHY_STAGE()
        hist_fetch<1, T>(hp, t0 + 7, H0);
        MRPHY_FETCH(p)
        my_[14] = cg14; my_[15] = cg15;
        LA(2, 8, H1);                                      // steps 8, 9
        hist_fetch<2, T>(hp, t0 + 5, H1);
        LA(1, 5, H0);                                      // step 7
        hist_fetch<2, T>(hp, t0 + 3, H0);
        lines_adj_carry<RELAX, PIN, CT, 1, T>(k, tp15, my_[0], my_[1], my_ + 2, H1, hx, hy, hz, g0, g1, g2);
        my_[0] = g1; my_[1] = g2;
        const T cgp15 = g0;
        MRPHY_STORE(p + 1)
        // ---- piece p
        MRPHY_STAGE()
        hist_fetch<2, T>(hp, t0 + 1, H1);
        if (p > 0) { MRPHY_FETCH(p - 1) }
        my_[15] = cgp15;
        LA(2, 9, H0);                                      // steps 3, 4
        hist_fetch<1, T>(hp, t0 + 0, H0);
        LA(2, 3, H1);                                      // steps 1, 2
        if (p > 0) hist_fetch<2, T>(hp, t0 - 16 + 14, H1); // first batch of the next period
        LA(1, 0, H0);                                      // step 0
        H0 = H1;
        MRPHY_STORE(p)
    }
#undef MRPHY_FETCH
#undef MRPHY_STAGE
#undef MRPHY_STORE
#undef MRPHY_OFF
#undef LA
    adj_end<RELAX, T, CT>(k, hx, hy, hz);
    if (valid && a.gMi) { a.gMi[r * 3] = hx; a.gMi[r * 3 + 1] = hy; a.gMi[r * 3 + 2] = hz; }
}
