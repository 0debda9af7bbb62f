// k_fused_bwd.hpp -- K2b (adjoint of the fused kernel), one transmit coil
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once
#include "k_fused_bwd_common.hpp"

template <typename T, typename CT, bool RELAX, bool HB1 = true>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_bwd(FusedBwdArgs<T> a)
{
    __shared__ __attribute__((aligned(16))) T red[5 * SEG * RED_PITCH];
    const int lane = threadIdx.x;
    const int64_t w = blockIdx.x, n = blockIdx.y;
    const int64_t nT = a.nT, rows = a.N * a.nM;
    const int64_t ntiles = (a.nM + WAVE - 1) / WAVE;
    // read-only, wave-uniform pulse through the constant address space: scalar loads (see K2)
    using CP = const T __attribute__((address_space(4)))*;
    CP rfr = (CP)(a.rf + n * a.rf_sn);
    CP rfi = rfr + nT;
    CP gx = (CP)(a.gr + n * a.gr_sn);
    CP gy = gx + nT;
    CP gz = gy + nT;
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
        if (HB1 && a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }
        const T vmask = valid ? T(1) : T(0);
        // lanes past nM (they hold a copy of the last valid spin) start from a zero cotangent: the adjoint state and every
        // dL/dB they form stay exact zeros (all of it is linear in the state), so they add nothing to the row sums --
        // round 6: masked here, once per tile, instead of three multiplications per step
        T hx = a.gMo[row * 3] * vmask, hy = a.gMo[row * 3 + 1] * vmask, hz = a.gMo[row * 3 + 2] * vmask;
        adj_begin<RELAX, T, CT>(k, hx, hy, hz);

        auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
            Bx = T(0); By = T(0);
            if (HB1) field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
            else     { Bx = rfr[t]; By = rfi[t]; }               // no b1 map: Bxy = rf (as K2 / K0)
            Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
        };

        // Two global round trips per segment used to sit on the critical path: the checkpoint (used
        // at once by the recompute) and the read-modify-write of the workspace rows.  Both are now
        // issued a segment's worth of work ahead: the next checkpoint at the top of the current
        // segment, the old workspace values before the sweep that produces what is added to them.
        const int64_t nseg = nT / SEG;
        T cx = T(0), cy = T(0), cz = T(0);
        if (nseg > 0) {
            const T* ck = a.Mck + ((nseg - 1) * rows + row) * 3;
            cx = ck[0]; cy = ck[1]; cz = ck[2];
        }
        // rows 64..79 (the second pass of the row sums): four lanes per row, one of the four chains each
        const int r1 = WAVE + (lane >> 2);
        for (int64_t seg = nseg - 1; seg >= 0; --seg) {
            const int64_t t0 = seg * SEG;
            T mx = cx, my = cy, mz = cz;
            if (seg > 0) {
                const T* ck = a.Mck + ((seg - 1) * rows + row) * 3;
                cx = ck[0]; cy = ck[1]; cz = ck[2];
            }
            T* dst0 = wsrow + (lane / SEG) * nT + t0 + (lane % SEG);
            T* dst1 = wsrow + (r1 / SEG) * nT + t0 + (r1 % SEG);
            T old0 = T(0), old1 = T(0);
            if (!first) { old0 = *dst0; old1 = *dst1; }
            // 1. forward recompute, keeping the state before each step
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
                    const int st = sb * 4 + j;
                    M0[st] = mx; M1[st] = my; M2[st] = mz;
                    Sv[st] = r[j].S; Cv[st] = r[j].C;     // reused by the sweep
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
                const T S4[4] = {Sv[sb * 4], Sv[sb * 4 + 1], Sv[sb * 4 + 2], Sv[sb * 4 + 3]};
                const T C4[4] = {Cv[sb * 4], Cv[sb * 4 + 1], Cv[sb * 4 + 2], Cv[sb * 4 + 3]};
                rot_prepare_adj_given<T, CT, 4>(k, Bx, By, Bz, S4, C4, ra);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int st = sb * 4 + j;
                    T g0, g1, g2;
                    rot_apply_adj<RELAX, T, CT>(k, ra[j], M0[st], M1[st], M2[st], hx, hy, hz,
                                                g0, g1, g2);
#if defined(K2B_KNOCK) && (K2B_KNOCK & 1)   // knock-out experiment (wrong results; MRPHY_DEV_FLAGS=-DK2B_KNOCK=1|2|3): no LDS writes
                    { T q0 = lx * g2, q1 = ly * g2, q2 = lz * g2, q3 = HB1 ? br * g0 + bi * g1 : g0, q4 = HB1 ? br * g1 - bi * g0 : g1;
                      asm volatile("" :: "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4)); }
#else
                    red[red_idx(0 * SEG + st, lane)] = lx * g2;
                    red[red_idx(1 * SEG + st, lane)] = ly * g2;
                    red[red_idx(2 * SEG + st, lane)] = lz * g2;
                    red[red_idx(3 * SEG + st, lane)] = HB1 ? br * g0 + bi * g1 : g0;
                    red[red_idx(4 * SEG + st, lane)] = HB1 ? br * g1 - bi * g0 : g1;
#endif
                }
            }
#if defined(K2B_KNOCK) && (K2B_KNOCK & 2)   // knock-out experiment: no row sums, no workspace update
            if (seg > 0) continue;
#endif
            __syncthreads();
            // 3. 80 row sums: lanes 0..63 take rows 0..63, then lane (r, j) chain j of row 64 + r
            // (the old workspace values and the checkpoint were requested a segment ago: ONE explicit wait for
            // all vector loads here, or the compiler -- which loses count of them across the loops in between --
            // puts s_waitcnt vmcnt(0) in front of EACH store below, and every store then waits for the one before)
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0) (gfx9 encoding; expcnt / lgkmcnt untouched)
            {                                                      // pass 0: rows 0..63, one per lane
                T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);      // 4 chains for ILP; fixed order
#pragma unroll
                for (int i = 0; i < WAVE; i += 4) {                // logical lanes i..i+3: one slot
                    const T* q = red + red_idx(lane, i);
                    p0 += q[0]; p1 += q[1]; p2 += q[2]; p3 += q[3];
                }
                *dst0 = old0 + ((p0 + p1) + (p2 + p3));            // old = 0 on the wave's first tile
            }
            static_assert(5 * SEG - WAVE == WAVE / 4, "second pass: 16 rows x 4 chains = one wave");
            {   // pass 1: rows 64..79.  Round 5 ran the loop above once more with 16 of the 64 lanes active (64 adds and 16
                // 16-byte reads issued for a quarter of a wave); now lane (row, j) forms chain p_j of its row -- the same
                // sixteen additions in the same order -- and the four chains meet through DPP as (p0 + p1) + (p2 + p3):
                // the same bits (x + y == y + x), a quarter of the instructions
                const int j = lane & 3;
                T p = T(0);
#pragma unroll
                for (int i = 0; i < WAVE; i += 4) p += red[red_idx(r1, i) + j];
                p += __shfl_xor(p, 1);                             // lanes 0, 1: p0 + p1;  lanes 2, 3: p2 + p3
                p += __shfl_xor(p, 2);                             // (p0 + p1) + (p2 + p3)
                if (j == 0) *dst1 = old1 + p;
            }
            __syncthreads();
        }
        adj_end<RELAX, T, CT>(k, hx, hy, hz);
        if (valid && a.gMi) { a.gMi[row * 3] = hx; a.gMi[row * 3 + 1] = hy; a.gMi[row * 3 + 2] = hz; }
        first = false;
    }
}

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
