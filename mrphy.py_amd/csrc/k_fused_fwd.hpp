// k_fused_fwd.hpp -- K2 (fused rf,gr -> Mo)
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

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
    MRPHY_STAMP_FIELD
};

// CK: write checkpoints (every ck_every steps, a multiple of the 8-step chunk).  Kept out of the
// plain instantiation so that its step loop contains no store: the pulse loads are then provably
// unclobbered and become (batched) scalar loads.
// NCM: 1 = one coil (pulse samples are scalar loads); 2 / 4 / 8 / 16 / 32 = up to that many coils (the
// lane's b1 in 2 NCM registers, the chunk's rf samples staged in LDS and read as broadcasts; the
// coil sum is ONE ascending FMA chain whatever NCM is, so every capacity -- and K0 -- rounds alike);
// 0 = any number of coils (b1 and rf from memory inside the coil loop: slow, correctness path).
// Measured at 64^3 x 1024 before the 16 / 32 capacities existed: 8 coils 0.75 ms, 9 coils 5.85 ms,
// 16 coils 20.8 ms on the memory path (tools/ptx_timing.py).
constexpr int K2_MAXC = 64;                              // largest register/LDS coil capacity (48 / 64: float only)
// HB1 (one-coil builds): the coil has a b1 map.  Without one Bxy = rf (beffective.py:147-151): the
// build then skips the complex product -- 6 of the ~50 VALU instructions of a step; with b1 = (1, 0)
// the product returns rf bit for bit anyway, so results are unchanged.  A template parameter, not a
// run-time test: a wave-uniform branch in the field assembly broke the batching of the pulse's scalar
// loads (round 1: 6.6 -> 7.2 ms).
template <typename T, typename CT, int NCM, bool CK, bool RELAX, bool HB1 = true>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_fwd(FusedArgs<T> a)
{
    constexpr int NS = (sizeof(T) == 8 && NCM == 1) ? 4 : 8;   // fp64, one coil: 8 steps' pulse samples (80 SGPRs) spill to VGPR lanes
    constexpr bool NC1 = (NCM == 1);
    constexpr bool NCR = (NCM >= 2);                     // coils in registers / LDS
    constexpr int MC = NCR ? NCM : 1;                    // coil capacity of this instantiation
    static_assert(NCM == 0 || NCM == 1 || NCM == 2 || NCM == 4 || NCM == 8 || NCM == 16 || NCM == 32 || NCM == 40 ||
                  NCM == 48 || NCM == 64, "coil capacities: 2/4/8/16/32/40/48/64");
    static_assert(MC <= K2_MAXC, "capacity above K2_MAXC: the launcher would never select it");
    __shared__ __attribute__((aligned(16))) T srf[NCR ? 2 * NS * MC : 4];  // [re|im][j][c]
    const int lane = threadIdx.x;
    const int64_t n = blockIdx.y;
    MRPHY_STAMP_BEGIN()
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
    if (NC1 && HB1 && a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }

    const int64_t nT = a.nT, nC = a.nC;
    // The pulse is read-only for the whole launch and its addresses are wave-uniform: pointers into the
    // CONSTANT address space make the loads scalar (s_load, batched) whatever else the loop does.  With
    // plain global pointers the checkpoint-writing build could not prove that its stores leave the pulse
    // alone and fetched the samples with vector loads + v_readfirstlane (K2 with checkpoints: 0.82 ms
    // where the plain build's rate gives 0.60 at 64^3 x 2048).
    using CP = const T __attribute__((address_space(4)))*;
    CP rfr = (CP)(a.rf + n * a.rf_sn);                       // [nT][nC]
    CP rfi = rfr + nT * nC;
    CP gx = (CP)(a.gr + n * a.gr_sn);
    CP gy = gx + nT;
    CP gz = gy + nT;
    const T* b1 = a.b1 ? a.b1 + row * 2 * nC : nullptr;
    const int64_t rows = a.N * a.nM;
    T b1r[MC], b1i[MC];
    if (NCR) {
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            b1r[c] = (c < nC) ? b1[c] : T(0);
            b1i[c] = (c < nC) ? b1[nC + c] : T(0);
        }
    }
    // NCR: rf samples of steps [tb, tb + cnt) -> LDS as [step][MC], ZERO beyond nC: the coil loop
    // below then needs no `c < nC` test (b1r/b1i are zero there too; adding exact zeros changes
    // nothing), stays one basic block, and its broadcast reads are batched.  With the test it compiled
    // to a branch and an exposed LDS round trip per coil, as in K0 (8 coils: 0.81 ms at 64^3 x 1024).
    auto stage_rf = [&](int64_t tb, int cnt) {
        __syncthreads();
        for (int i = lane; i < cnt * MC; i += WAVE) {
            const int j = i / MC, c = i - j * MC;
            const bool on = c < (int)nC;
            srf[i] = on ? rfr[(tb + j) * nC + c] : T(0);
            srf[NS * MC + i] = on ? rfi[(tb + j) * nC + c] : T(0);
        }
        __syncthreads();
    };
    int64_t tstage = 0;                                       // first step held in srf

    auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
        Bx = T(0); By = T(0);
        if (NC1) {
            if (HB1) field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
            else     { Bx = rfr[t]; By = rfi[t]; }
        } else if (NCR) {
            const T* qr = srf + (t - tstage) * MC;
            const T* qi = qr + NS * MC;
#pragma unroll
            for (int c = 0; c < MC; ++c) field_xy_fma<T>(b1r[c], b1i[c], qr[c], qi[c], Bx, By);
        } else {
            for (int64_t c = 0; c < nC; ++c)
                field_xy_fma<T>(b1[c], b1[nC + c], rfr[t * nC + c], rfi[t * nC + c], Bx, By);
        }
        Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
    };

    // checkpoints: a running destination and the step of the next one -- `t0 % ck_every`, `t0 / ck_every` on
    // 64-bit run-time values were a software division on the scalar unit every 8 steps (round 3: +200 scalar
    // instructions per 16 steps in the ISA of the checkpoint build)
    // The checkpoint build: all of the prologue's vector loads are awaited HERE, before the loop.  Otherwise the
    // wait for them lands in the loop header (the join of the prologue and the back edge) as s_waitcnt
    // vmcnt(0), where it also waits, every 8 steps, for the checkpoint store of the iteration before.
    if (CK) __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
    int64_t ck_next = 0;
    T* ckp = CK ? a.Mck + row * 3 : nullptr;
    const int64_t ck_pitch = rows * 3;
    int64_t t0 = 0;
    MRPHY_PRIO_INIT(a)
    for (; t0 + NS <= nT; t0 += NS) {
        MRPHY_PRIO_TICK(a, 3u - ((unsigned)(t0 >> a.prio_shift) & 3u))
        if (NCR) { tstage = t0; stage_rf(t0, NS); }
        if (CK && t0 == ck_next) {
            if (valid) { ckp[0] = mx; ckp[1] = my; ckp[2] = mz; }
            ckp += ck_pitch; ck_next += a.ck_every;
        }
        T Bx[NS], By[NS], Bz[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) field(t0 + j, Bx[j], By[j], Bz[j]);
        Rot<T> r[NS];
        rot_prepare<T, CT, NS>(k, Bx, By, Bz, r);
#pragma unroll
        for (int j = 0; j < NS; ++j) rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
    }
    if (NCR && t0 < nT) { tstage = t0; stage_rf(t0, (int)(nT - t0)); }
    for (; t0 < nT; ++t0) {                                   // nT % 8 tail
        if (CK && t0 == ck_next) {
            if (valid) { ckp[0] = mx; ckp[1] = my; ckp[2] = mz; }
            ckp += ck_pitch; ck_next += a.ck_every;
        }
        T Bx[1], By[1], Bz[1];
        field(t0, Bx[0], By[0], Bz[0]);
        Rot<T> r[1];
        rot_prepare<T, CT, 1>(k, Bx, By, Bz, r);
        rot_apply<RELAX, T, CT>(k, r[0], mx, my, mz);
    }
    if (valid) { a.Mo[row * 3] = mx; a.Mo[row * 3 + 1] = my; a.Mo[row * 3 + 2] = mz; }
    MRPHY_STAMP_END(a, (int64_t)blockIdx.y * gridDim.x + blockIdx.x)
}

