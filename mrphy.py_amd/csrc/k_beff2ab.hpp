// k_beff2ab.hpp -- beff2ab forward (optionally recording its history) and adjoint
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

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
    T* hist;               // SAVE: [tile][t][12][lane], the 3x4 state BEFORE step t (for the adjoint)
    Bc g, E1, E2;
    const void* E1m1;
    int64_t rows, nM, nT;
    int vec_ok;
};

// (AB_HIST_STEP = 12 * WAVE elements per time step of one tile: geom.hpp)

// state before step t of the four columns (x, y, z of column j at [3 j .. 3 j + 2]): twelve coalesced
// 256-B wave stores / loads, as the vector history of K1 / K3
template <typename T>
__device__ __forceinline__ void ab_hist_store(T* hp, int64_t t, const T (&cx)[4], const T (&cy)[4],
                                              const T (&cz)[4])
{
    T* q = hp + t * AB_HIST_STEP;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __builtin_nontemporal_store(cx[j], q + (3 * j) * WAVE);
        __builtin_nontemporal_store(cy[j], q + (3 * j + 1) * WAVE);
        __builtin_nontemporal_store(cz[j], q + (3 * j + 2) * WAVE);
    }
}

template <typename T, typename CT, int TC, bool SAVE>
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
    kl.e1m1 = typename CTr<CT>::reg(0);                                 // the A columns: linear part only

    T cx[4] = {T(1), T(0), T(0), T(0)}, cy[4] = {T(0), T(1), T(0), T(0)},
      cz[4] = {T(0), T(0), T(1), T(0)};
    const int64_t rowlen = 3 * a.nT;
    T* hp = SAVE ? a.hist + (int64_t)blockIdx.x * a.nT * AB_HIST_STEP + lane : nullptr;
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
                    if (SAVE) ab_hist_store<T>(hp, c * TC + tt + q, cx, cy, cz);
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
        if (SAVE) ab_hist_store<T>(hp, t, cx, cy, cz);
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

// ---------------------------------------------------------------------------------------------
// Adjoint of beff2ab: ONE backward sweep over Beff for all four columns (the reference gets it from
// autograd through its time loop, beffective.py:88-100).  The four columns share the step's
// rotation, so rot_prepare_adj runs once per step and rot_apply_adj four times; dL/dBeff is the
// sum of the four contributions.  The relaxation offset is an additive constant: it does not
// appear in the adjoint.  Same streaming as k_bloch_bwd: Beff chunks through the LDS tile,
// dL/dBeff written back in place and stored coalesced.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct AbBwdArgs {
    const T* hist;         // as written by k_beff2ab<SAVE>
    const T* Beff;
    const T* gA;           // (rows, 3, 3) or null (= 0)
    const T* gB;           // (rows, 3) or null (= 0)
    T* gBeff;              // (rows, nT, 3)
    Bc g, E1, E2;
    int64_t rows, nM, nT;
    int vec_ok;
    T* gC;                 // GC builds: (rows, 4) [dL/dg, dL/dE1, dL/dE2, dL/dE1m1] per spin
};

// GC (round 4): also the gradients w.r.t. the per-spin constants -- what autograd through the reference's time loop
// gives a caller who differentiates beff2ab w.r.t. E1, E2, gamma, dt (beffective.py:73-100): the four columns share
// each step's rotation, their contributions add; the relaxation offset acts on the B column only.
template <typename T, typename CT, int TC, bool GC = false>
__global__ __launch_bounds__(WAVE) void k_beff2ab_bwd(AbBwdArgs<T> a)
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

    T hx[4], hy[4], hz[4];                              // dL/d(state), column j
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        hx[j] = a.gA ? a.gA[rc * 9 + j] : T(0);
        hy[j] = a.gA ? a.gA[rc * 9 + 3 + j] : T(0);
        hz[j] = a.gA ? a.gA[rc * 9 + 6 + j] : T(0);
    }
    hx[3] = a.gB ? a.gB[rc * 3] : T(0);
    hy[3] = a.gB ? a.gB[rc * 3 + 1] : T(0);
    hz[3] = a.gB ? a.gB[rc * 3 + 2] : T(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) adj_begin<true, T, CT>(k, hx[j], hy[j], hz[j]);   // (no adj_end: h is not an output)

    const int64_t rowlen = 3 * a.nT;
    const int64_t nfull = a.vec_ok ? a.nT / TC : 0;
    const T* hp = a.hist + (int64_t)blockIdx.x * a.nT * AB_HIST_STEP + lane;

    T acc[4] = {T(0), T(0), T(0), T(0)};
    // one adjoint step for the four columns; returns dL/dB of this step
    auto step4 = [&](const RotAdj<T>& ra, T Bx, T By, T Bz, int64_t t, T& gx, T& gy, T& gz) {
        const T* q = hp + t * AB_HIST_STEP;
        gx = gy = gz = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T m0 = __builtin_nontemporal_load(q + (3 * j) * WAVE);
            const T m1 = __builtin_nontemporal_load(q + (3 * j + 1) * WAVE);
            const T m2 = __builtin_nontemporal_load(q + (3 * j + 2) * WAVE);
            T ax, ay, az;
            if constexpr (GC) {
                using R = typename CTr<CT>::reg;
                const T sx = hx[j], sy = hy[j], sz = hz[j];
                T dbx, dby, dbz;
                rot_apply_adj_core<true, T, CT>(k, ra, m0, m1, m2, hx[j], hy[j], hz[j], dbx, dby, dbz);
                ax = T(R(dbx) * k.g); ay = T(R(dby) * k.g); az = T(R(dbz) * k.g);
                adj_const_accumulate<true, T, CT>(ra, Bx, By, Bz, m0, m1, m2, sx, sy, sz, dbx, dby, dbz, acc, j == 3);
            } else {
                rot_apply_adj<true, T, CT>(k, ra, m0, m1, m2, hx[j], hy[j], hz[j], ax, ay, az);
            }
            gx += ax; gy += ay; gz += az;
        }
    };

    {   // tail first (time runs backwards)
        const T* bp = a.Beff + rc * rowlen;
        T* gp = a.gBeff + rc * rowlen;
        for (int64_t t = a.nT - 1; t >= nfull * TC; --t) {
            const T bx_[1] = {bp[t * 3]}, by_[1] = {bp[t * 3 + 1]}, bz_[1] = {bp[t * 3 + 2]};
            RotAdj<T> ra[1];
            rot_prepare_adj<T, CT, 1>(k, bx_, by_, bz_, ra);
            T gx, gy, gz;
            step4(ra[0], bx_[0], by_[0], bz_[0], t, gx, gy, gz);
            if (valid) { gp[t * 3] = gx; gp[t * 3 + 1] = gy; gp[t * 3 + 2] = gz; }
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
                T Bx[VE], By[VE], Bz[VE];
#pragma unroll
                for (int q = 0; q < VE; ++q) { Bx[q] = bb[3 * q]; By[q] = bb[3 * q + 1]; Bz[q] = bb[3 * q + 2]; }
                RotAdj<T> ra[VE];
                rot_prepare_adj<T, CT, VE>(k, Bx, By, Bz, ra);
#pragma unroll
                for (int q = VE - 1; q >= 0; --q)
                    step4(ra[q], Bx[q], By[q], Bz[q], c * TC + tt + q, gg[3 * q], gg[3 * q + 1], gg[3 * q + 2]);
                *reinterpret_cast<V*>(rowB + tt * 3) = vec_pack(gg);
                *reinterpret_cast<V*>(rowB + tt * 3 + VE) = vec_pack(gg + VE);
                *reinterpret_cast<V*>(rowB + tt * 3 + 2 * VE) = vec_pack(gg + 2 * VE);
            }
            __syncthreads();
            chunk_store<T, TC>(tileB, a.gBeff, row0, a.rows, rowlen, c * TC, lane);
        }
    }
    if constexpr (GC) {
        adj_const_finish<T, CT>(k, acc);
        if (valid && a.gC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a.gC[r * 4 + i] = acc[i];
        }
    }
}

