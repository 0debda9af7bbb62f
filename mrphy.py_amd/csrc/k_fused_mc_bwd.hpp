// k_fused_mc_bwd.hpp -- K2b for parallel transmit (2..8 coils)
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once
#include "k_fused_bwd_common.hpp"

// =============================================================================================
// K2b for parallel transmit: nC <= K2B_MAXC coils, rf (N|1, 2, nT, nC), b1 (N, nM, 2, nC).
// Same sweep as the single-coil kernel.  The per-coil sums over the 64 spins of a tile
//     grad_rf_re[c][t] = sum_l b1r[c][l] gBx[t][l] + b1i[c][l] gBy[t][l]
//     grad_rf_im[c][t] = sum_l b1r[c][l] gBy[t][l] - b1i[c][l] gBx[t][l]
// are small dot products: the raw gBx, gBy rows of the segment (2 x 16 rows) sit in the reduction
// tile next to the three loc*gBz rows, the tile's b1 in a second LDS array (2 nC rows x 64), and
// lane (step, re|im, half of the spins) forms its dot product per coil in spin order; the two
// halves are added in fixed order.  Workspace rows per wave: [gr_x, gr_y, gr_z, (re, im) x nC].
// =============================================================================================
// (K2B_MAXC = 8; K2B_MC_MAX_WAVES = 256 * 8 -- 18 KB of LDS per wave -> 8 per CU = 2 per SIMD: geom.hpp)

// MC: coil capacity of the build (2 / 4 / 8, the smallest that holds nC).  b1 registers, the staged rf
// samples and the coefficient rows are ZERO beyond nC, so that neither the field's coil loop nor the
// reduction's has a `c < nC` test in it (round 3: with the test every coil was its own basic block -- a
// wave-uniform branch and an exposed LDS round trip per coil, twice per step in the field alone; K0 and
// K2 had been rid of that in round 2).  Adding exact zeros changes nothing (at most the sign of a zero
// sum): the recomputed states stay those of K2's forward.
template <typename T, typename CT, bool RELAX, int MC>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_bwd_mc(FusedBwdArgs<T> a, int nC)
{
    constexpr int K2B_NCF = 2 * MC + 3;             // coefficient rows: b1r[c], b1i[c], loc x y z
    // NOT generic in SEG: the workspace update and the dot products below take their step from the lane as
    // `lane >> 2` (64 lanes = 16 steps x (re|im) x (half of the spins)), while raw[], srf[] and the workspace
    // rows are sized from SEG.  With any other SEG lanes 32..63 would index steps past the segment (LDS reads
    // past raw[], global stores past t0 + SEG: the round-5 SEG = 8 dev build faulted exactly there).
    static_assert(SEG * 4 == WAVE && SEG == 16,
                  "k_bloch_rfgr_bwd_mc derives the step from the lane as lane >> 2: SEG must be 16 (geom.hpp)");
    // raw dL/dB rows of one segment: [gBx | gBy | gBz][step][lane], slot-swizzled like `red`
    __shared__ __attribute__((aligned(16))) T raw[3 * SEG * RED_PITCH];
    // the tile's coefficients [b1r c0..7 | b1i c0..7 | loc x y z][lane], zero for lanes past nM
    __shared__ __attribute__((aligned(16))) T cfs[K2B_NCF * WAVE];
    __shared__ __attribute__((aligned(16))) T srf[2 * SEG * MC];             // [re|im][step][c], zero beyond nC
    const int lane = threadIdx.x;
    const int64_t w = blockIdx.x, n = blockIdx.y;
    const int64_t nT = a.nT, rows = a.N * a.nM;
    const int64_t ntiles = (a.nM + WAVE - 1) / WAVE;
    const int nQ = 3 + 2 * nC;
    const T* __restrict__ rfr = a.rf + n * a.rf_sn;            // [nT][nC]
    const T* __restrict__ rfi = rfr + nT * nC;
    using CP = const T __attribute__((address_space(4)))*;     // wave-uniform gradient samples: scalar loads
    CP gx = (CP)(a.gr + n * a.gr_sn);
    CP gy = gx + nT;
    CP gz = gy + nT;
    T* wsrow = a.work + ((w * a.N + n) * nQ) * nT;
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
        T br[MC], bi[MC];
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            br[c] = (c < nC) ? a.b1[row * 2 * nC + c] : T(0);
            bi[c] = (c < nC) ? a.b1[row * 2 * nC + nC + c] : T(0);
        }
        const T vmask = valid ? T(1) : T(0);
        __syncthreads();                                   // previous tile's coefficients released
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            cfs[c * WAVE + lane] = br[c] * vmask;
            cfs[(MC + c) * WAVE + lane] = bi[c] * vmask;
        }
        cfs[(2 * MC + 0) * WAVE + lane] = lx * vmask;
        cfs[(2 * MC + 1) * WAVE + lane] = ly * vmask;
        cfs[(2 * MC + 2) * WAVE + lane] = lz * vmask;
        T hx = a.gMo[row * 3], hy = a.gMo[row * 3 + 1], hz = a.gMo[row * 3 + 2];
        adj_begin<RELAX, T, CT>(k, hx, hy, hz);

        int64_t tstage = 0;
        auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
            Bx = T(0); By = T(0);
            const T* qr = srf + (t - tstage) * MC;          // broadcast reads, batched: no test in the loop
            const T* qi = qr + SEG * MC;
#pragma unroll
            for (int c = 0; c < MC; ++c) field_xy_fma<T>(br[c], bi[c], qr[c], qi[c], Bx, By);
            Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
        };

        const int64_t nseg = nT / SEG;                      // checkpoint and workspace values are
        T cx = T(0), cy = T(0), cz = T(0);                  // fetched a segment ahead (see K2b)
        if (nseg > 0) {
            const T* ck = a.Mck + ((nseg - 1) * rows + row) * 3;
            cx = ck[0]; cy = ck[1]; cz = ck[2];
        }
        for (int64_t seg = nseg - 1; seg >= 0; --seg) {
            const int64_t t0 = seg * SEG;
            // the segment's rf samples (SEG * MC <= 128 floats per part) -> LDS; the barrier at the
            // end of the previous segment has released srf
            tstage = t0;
            for (int i = lane; i < SEG * MC; i += WAVE) {
                const int st_ = i / MC, c_ = i - st_ * MC;
                const bool on = c_ < nC;
                srf[i] = on ? rfr[(t0 + st_) * nC + c_] : T(0);
                srf[SEG * MC + i] = on ? rfi[(t0 + st_) * nC + c_] : T(0);
            }
            __syncthreads();
            T mx = cx, my = cy, mz = cz;
            if (seg > 0) {
                const T* ck = a.Mck + ((seg - 1) * rows + row) * 3;
                cx = ck[0]; cy = ck[1]; cz = ck[2];
            }
            // old workspace values of the rows this lane updates at the end of the segment
            const int st_w = lane >> 2, ri_w = (lane >> 1) & 1;
            const bool wr_w = (lane & 1) == 0;
            T* dst0 = wsrow + (3 + ri_w) * nT + t0 + st_w;        // + 2 c nT per coil
            T* dg0 = wsrow + ri_w * nT + t0 + st_w;               // grad_gr axis ri
            T* dg2 = wsrow + 2 * nT + t0 + st_w;                  // grad_gr axis z (ri == 0 lanes)
            T old[MC], oldg0 = T(0), oldg2 = T(0);
#pragma unroll
            for (int c = 0; c < MC; ++c)
                old[c] = (!first && wr_w && c < nC) ? dst0[2 * c * nT] : T(0);
            if (!first && wr_w) { oldg0 = *dg0; if (ri_w == 0) oldg2 = *dg2; }
            T M0[SEG], M1[SEG], M2[SEG], Sv[SEG], Cv[SEG];
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
                    Sv[sb * 4 + j] = r[j].S; Cv[sb * 4 + j] = r[j].C;     // reused by the sweep
                    rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
                }
            }
#pragma unroll
            for (int sb = SEG / 4 - 1; sb >= 0; --sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                RotAdj<T> ra[4];
                const T S4[4] = {Sv[sb * 4], Sv[sb * 4 + 1], Sv[sb * 4 + 2], Sv[sb * 4 + 3]};
                const T C4[4] = {Cv[sb * 4], Cv[sb * 4 + 1], Cv[sb * 4 + 2], Cv[sb * 4 + 3]};
                rot_prepare_adj_given<T, CT, 4>(k, Bx, By, Bz, S4, C4, ra);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int st = sb * 4 + j;
                    T g0, g1, g2;
                    rot_apply_adj<RELAX, T, CT>(k, ra[j], M0[st], M1[st], M2[st], hx, hy, hz,
                                                g0, g1, g2);
                    raw[red_idx(0 * SEG + st, lane)] = g0;   // lanes past nM: zero coefficients
                    raw[red_idx(1 * SEG + st, lane)] = g1;
                    raw[red_idx(2 * SEG + st, lane)] = g2;
                }
            }
            __syncthreads();
            // All sums over the tile's spins are dot products of a raw row with coefficient rows:
            //   lane = (step, kind, half of the spins), kind 0: re, 1: im  -> per coil c
            //     re: b1r[c].gBx + b1i[c].gBy        im: b1r[c].gBy - b1i[c].gBx
            //   and for grad_gr lane = (step, axis i < 3, -, half), kind 2:  loc_i . gBz
            // spins outer, accumulators inner; halves added in fixed order; one load round trip
            // for the workspace update.
            {
                const int st = lane >> 2, ri = (lane >> 1) & 1, half = lane & 1;
                T acc[MC], accg[2];                 // accg: this lane's 1-2 grad_gr axes
#pragma unroll
                for (int c = 0; c < MC; ++c) acc[c] = T(0);
                accg[0] = accg[1] = T(0);
                // grad_gr: (st, ri, half) lanes take axis ri (0: x, 1: y); axis z rides on ri == 0
                const T* l0 = cfs + (2 * MC + ri) * WAVE;
                const T* l2 = cfs + (2 * MC + 2) * WAVE;
#pragma unroll 2
                for (int i = half * 32; i < half * 32 + 32; i += 4) {
                    const T* qx = raw + red_idx(0 * SEG + st, i);
                    const T* qy = raw + red_idx(1 * SEG + st, i);
                    const T* qz = raw + red_idx(2 * SEG + st, i);
                    T pp[4], qq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        pp[u] = ri == 0 ? qx[u] : qy[u];
                        qq[u] = ri == 0 ? qy[u] : -qx[u];
                        accg[0] += l0[i + u] * qz[u];
                        accg[1] += l2[i + u] * qz[u];
                    }
#pragma unroll
                    for (int c = 0; c < MC; ++c) {
                        const T* b_r = cfs + c * WAVE + i;
                        const T* b_i = cfs + (MC + c) * WAVE + i;
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[c] += b_r[u] * pp[u] + b_i[u] * qq[u];
                    }
                }
                const bool wr = half == 0;
                // one wait for the old workspace values (requested at the start of the segment) instead of a
                // compiler-inserted s_waitcnt vmcnt(0) in front of EVERY store below -- which made each of the
                // 2 nC + 3 stores wait for the one before it: 19 store round trips per segment at 8 coils
                __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) (gfx9 encoding; expcnt / lgkmcnt untouched)
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    const T other = __shfl_xor(acc[c], 1);
                    const T sum = half == 0 ? acc[c] + other : other + acc[c];
                    if (wr && c < nC) dst0[2 * c * nT] = old[c] + sum;   // old = 0 on the first tile
                }
                {
                    const T o0 = __shfl_xor(accg[0], 1), o2 = __shfl_xor(accg[1], 1);
                    const T s0 = half == 0 ? accg[0] + o0 : o0 + accg[0];
                    const T s2 = half == 0 ? accg[1] + o2 : o2 + accg[1];
                    if (wr) {
                        *dg0 = oldg0 + s0;
                        if (ri == 0) *dg2 = oldg2 + s2;
                    }
                }
            }
            __syncthreads();
        }
        adj_end<RELAX, T, CT>(k, hx, hy, hz);
        if (valid && a.gMi) { a.gMi[row * 3] = hx; a.gMi[row * 3 + 1] = hy; a.gMi[row * 3 + 2] = hz; }
        first = false;
    }
}

// Pass 2 for nQ = 3 + 2 nC quantities; grad_rf is (N, 2, nT, nC).
template <typename T>
__global__ __launch_bounds__(P2_T * P2_G) void k_bloch_rfgr_bwd_mc_p2(const T* work, T* grf, T* ggr,
                                                                      int64_t N, int64_t nT,
                                                                      int64_t P, int nC)
{
    __shared__ T part[P2_G][P2_T];
    const int tl = threadIdx.x % P2_T, g = threadIdx.x / P2_T;
    const int64_t t = (int64_t)blockIdx.x * P2_T + tl;
    const int64_t q = blockIdx.y, n = blockIdx.z;
    const int nQ = 3 + 2 * nC;
    T acc = T(0);
    if (t < nT)
        for (int64_t w = g; w < P; w += P2_G) acc += work[((w * N + n) * nQ + q) * nT + t];
    part[g][tl] = acc;
    __syncthreads();
    if (g != 0 || t >= nT) return;
    T sum = part[0][tl];
#pragma unroll
    for (int i = 1; i < P2_G; ++i) sum += part[i][tl];
    if (q < 3) { if (ggr) ggr[(n * 3 + q) * nT + t] = sum; }
    else if (grf) {
        const int64_t c = (q - 3) / 2, ri = (q - 3) % 2;
        grf[((n * 2 + ri) * nT + t) * nC + c] = sum;
    }
}

