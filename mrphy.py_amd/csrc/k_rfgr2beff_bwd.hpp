// k_rfgr2beff_bwd.hpp -- adjoint of K0 (rfgr2beff) to rf, gr
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

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
    const bool fullv = e0 + VW <= L;                   // else: this thread straddles the row end
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
        if (VW == V16<T>::N && fullv) {
            for (; i + U <= cnt; i += U) {             // U rows' loads issued before the first use
                typename V16<T>::type v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = __builtin_nontemporal_load(
                        reinterpret_cast<const typename V16<T>::utype*>(src0 + (i + u) * L));
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
            if (VW == V16<T>::N && fullv) {
                vec_unpack(__builtin_nontemporal_load(
                               reinterpret_cast<const typename V16<T>::utype*>(src)), g);
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

// Pass 1 for 2..32 coils with a b1 map (coil capacity MC = 8 / 16 / 32): as p1v, but an element
// keeps KS = 2 MC running sums,
//   x or y element e:  S[c] = sum_s b1r[c][s] g[e],  S[MC + c] = sum_s b1i[c][s] g[e]
//   z element:         S[0..2] = sum_s loc[s] g[e]
// written to work[(sg, n, k, e)], k < 2 nC' (nC' = max(nC, 2): three loc sums need k = 0..2).
// Pass 2: grad_gr[i][t] = S_i(3t+2);  grad_rf_re[c][t] = S_c(3t) + S_{M+c}(3t+1);
//         grad_rf_im[c][t] = S_c(3t+1) - S_{M+c}(3t).   One pass over gB instead of nC + 1.
// (BWD_MAXC = 32, the largest coil capacity of the one-pass adjoint: geom.hpp)
#ifdef MRPHY_DEV_KNOBS   // the element-per-thread one-pass adjoint of round 2: A/B baseline of the dev build only
// Geometry of the one-pass multi-coil adjoint per coil capacity MC (8 / 16 / 32), read by pass 1,
// pass 2, the launcher and the workspace query alike: KS = 2 MC running sums per element (so the
// workspace holds KS rows of 3 nT per spin group), VW elements per thread chosen so that the
// accumulators stay at 64 registers, GROUP spins per LDS sub-block so that the coefficient rows
// (2 KS each) stay at 16 KB.  From 16 coils on the pass is VALU-bound (>= 32 FMAs per 4 B read).
template <typename T, int MC>
struct BwdGeom {
    static_assert(MC == 8 || MC == 16 || MC == 32, "coil capacities: 8/16/32");
    static constexpr int KS = 2 * MC;
    static constexpr int VWFULL = V16<T>::N;
    static constexpr int VW = MC == 8 ? VWFULL : (MC == 16 ? (VWFULL / 2 > 0 ? VWFULL / 2 : 1) : 1);
    static constexpr int GROUP = 1024 / MC;        // 128 / 64 / 32 spins: 2 KS GROUP = 4096 elements
    // blocks per CU the register allocation is bounded for: 4 (128 VGPRs) everywhere except fp64 at
    // capacity 32, whose 64 double accumulators alone are 128 VGPRs (it spilled 180 B/lane at 4)
    static constexpr int MINBLK = (sizeof(T) == 8 && MC == 32) ? 2 : 4;
};


template <typename T, int VW, int MC>
__global__ __launch_bounds__(256, (BwdGeom<T, MC>::MINBLK)) void k_rfgr2beff_bwd_p1mc(BeffBwdArgs<T> a)
{
    using G = BwdGeom<T, MC>;
    constexpr int KS = G::KS;
    constexpr int BWD_MC_GROUP = G::GROUP;
    static_assert(VW == 1 || VW == G::VW, "VW must come from BwdGeom");
    const int64_t L = 3 * a.nT;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VW;
    const int64_t sg = blockIdx.y, n = blockIdx.z;
    const int64_t s0 = sg * a.spins_per_group;
    const int64_t s1 = (s0 + a.spins_per_group < a.nM) ? s0 + a.spins_per_group : a.nM;
    const int nC = (int)a.nC;
    // one coefficient row per spin: [b1r c0..MC-1 | b1i c0..MC-1 | loc x y z 0 ...]; an x/y element
    // multiplies by the first half, a z element by the second -- no selects in the loop
    __shared__ __attribute__((aligned(16))) T sc[BWD_MC_GROUP][2 * KS];
    const bool active = e0 < L;
    const bool fullv = e0 + VW <= L;                   // else: this thread straddles the row end
    int half[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) half[j] = (((e0 + j) % 3) == 2) ? KS : 0;
    T acc[VW][KS];
#pragma unroll
    for (int j = 0; j < VW; ++j)
#pragma unroll
        for (int k = 0; k < KS; ++k) acc[j][k] = T(0);
    constexpr int U = 2;
    auto accumulate = [&](int i, const T* g) {
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const T* cf = sc[i] + half[j];
#pragma unroll
            for (int k = 0; k < KS; ++k) acc[j][k] += cf[k] * g[j];
        }
    };
    for (int64_t sb0 = s0; sb0 < s1; sb0 += BWD_MC_GROUP) {
        const int64_t cnt = (s1 - sb0 < BWD_MC_GROUP) ? s1 - sb0 : BWD_MC_GROUP;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < cnt * 2 * KS; i += 256) {
            const int64_t rr = i / (2 * KS), k = i - rr * 2 * KS;
            const int64_t row = n * a.nM + sb0 + rr;
            T v = T(0);
            if (k < MC)            { if (k < nC) v = a.b1[row * 2 * nC + k]; }
            else if (k < KS)       { if (k - MC < nC) v = a.b1[row * 2 * nC + nC + (k - MC)]; }
            else if (k < KS + 3)   v = a.loc[row * 3 + (k - KS)];
            sc[rr][k] = v;
        }
        __syncthreads();
        if (!active) continue;
        const T* src0 = a.gB + (n * a.nM + sb0) * L + e0;
        int64_t i = 0;
        if (VW == V16<T>::N && fullv) {
            for (; i + U <= cnt; i += U) {
                typename V16<T>::type v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = __builtin_nontemporal_load(
                        reinterpret_cast<const typename V16<T>::utype*>(src0 + (i + u) * L));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    T g[VW];
                    vec_unpack(v[u], g);
                    accumulate((int)(i + u), g);
                }
            }
        } else if (sizeof(T) == 4 && VW == 2 && fullv) {   // 16-coil capacity: 8-byte loads, U rows in flight
            for (; i + U <= cnt; i += U) {
                f32x2 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x2_u*>(src0 + (i + u) * L));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    T g[VW];
                    g[0] = T(v[u].x); g[VW - 1] = T(v[u].y);
                    accumulate((int)(i + u), g);
                }
            }
        }
        for (; i < cnt; ++i) {
            T g[VW];
            const T* src = src0 + i * L;
            if (VW == V16<T>::N && fullv) {
                vec_unpack(__builtin_nontemporal_load(
                               reinterpret_cast<const typename V16<T>::utype*>(src)), g);
            } else if (sizeof(T) == 4 && VW == 2 && fullv) {
                const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_u*>(src));
                g[0] = T(v.x); g[VW - 1] = T(v.y);
            } else {
#pragma unroll
                for (int j = 0; j < VW; ++j) g[j] = (e0 + j < L) ? src[j] : T(0);
            }
            accumulate((int)i, g);
        }
    }
    if (!active) return;
    T* w = a.work + ((sg * a.N + n) * KS) * L;
#pragma unroll
    for (int j = 0; j < VW; ++j)
        if (e0 + j < L) {
#pragma unroll
            for (int k = 0; k < KS; ++k) w[k * L + e0 + j] = acc[j][k];
        }
}

template <typename T, int MC>
__global__ __launch_bounds__(256) void k_rfgr2beff_bwd_p2mc(BeffBwdArgs<T> a)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t q = blockIdx.y;          // 0..2: grad_gr row; 3 + 2c + ri: grad_rf (c, re|im)
    const int64_t n = blockIdx.z;
    if (t >= a.nT) return;
    const int64_t L = 3 * a.nT;
    T acc = T(0);
    for (int64_t sg = 0; sg < a.nSG; ++sg) {           // fixed order: deterministic
        const T* w = a.work + ((sg * a.N + n) * (2 * MC)) * L + 3 * t;
        if (q < 3) {
            acc += w[q * L + 2];
        } else {
            const int64_t c = (q - 3) / 2, ri = (q - 3) % 2;
            const T* wr = w + c * L;
            const T* wi = w + (MC + c) * L;
            acc += ri == 0 ? (wr[0] + wi[1]) : (wr[1] - wi[0]);
        }
    }
    if (q < 3) { if (a.ggr) a.ggr[(n * 3 + q) * a.nT + t] = acc; }
    else if (a.grf) {
        const int64_t c = (q - 3) / 2, ri = (q - 3) % 2;
        a.grf[((n * 2 + ri) * a.nT + t) * a.nC + c] = acc;
    }
}

#endif  // MRPHY_DEV_KNOBS

// NE consecutive elements, element-aligned, non-temporal, in the widest pieces (16 B, 8 B, one element):
// a 3- or 6-element vector type would not do -- clang widens a 3-vector load to 4 elements, which at the
// last time point of the last row reads past the tensor
template <typename T, int NE>
struct ElemRun { T v[NE]; };
template <typename T, int NE>
__device__ __forceinline__ ElemRun<T, NE> load_run_nt(const T* p)
{
    ElemRun<T, NE> r;
    constexpr int VE = V16<T>::N;
    constexpr int NV = NE / VE;
#pragma unroll
    for (int v = 0; v < NV; ++v)
        vec_unpack(__builtin_nontemporal_load(reinterpret_cast<const typename V16<T>::utype*>(p + v * VE)), r.v + v * VE);
    constexpr int E1 = NV * VE;
    if constexpr (sizeof(T) == 4 && NE - E1 >= 2) {
        const f32x2 h = __builtin_nontemporal_load(reinterpret_cast<const f32x2_u*>(p + E1));
        r.v[E1] = T(h.x); r.v[E1 + 1] = T(h.y);
#pragma unroll
        for (int e = E1 + 2; e < NE; ++e) r.v[e] = __builtin_nontemporal_load(p + e);
    } else {
#pragma unroll
        for (int e = E1; e < NE; ++e) r.v[e] = __builtin_nontemporal_load(p + e);
    }
    return r;
}

#ifdef MRPHY_DEV_KNOBS   // the DPP version of the step-per-thread adjoint: A/B evidence of the dev build only
// =============================================================================================
// K0 adjoint for parallel transmit (2..32 coils with a b1 map), round 3: a thread owns whole TIME
// POINTS, and the row's b1 reaches the FMAs through DPP.
//
// p1mc above gives a thread VW elements of the (t, xyz) axis and 2 MC running sums per ELEMENT: a z
// element runs its 2 MC FMAs on zeros, gBx and gBy of one time point meet their b1 in different
// threads, and the sums per time point number 6 MC (so the thread tile shrinks to one element at 32
// coils: 64 LDS words per 4 bytes of gB).  Here a thread owns TP time points and keeps, per time
// point, exactly the sums the gradient has: 2 MC for grad_rf (re, im per coil) and 3 for grad_gr:
//     gRe[c] += b1r[c] gBx + b1i[c] gBy      gIm[c] += b1r[c] gBy - b1i[c] gBx      gG[i] += loc[i] gBz
// 4 MC + 3 FMAs per time point instead of 6 MC.  What would bound such a loop on this part is not the
// VALU but the LDS: a wave-uniform (broadcast) ds_read still delivers one word per two clocks per CU,
// 128 clocks for the 64 words of a 32-coil row against 262 VALU clocks per SIMD, four SIMDs per LDS.
// So the row's words are fetched ONCE per 16 (ds_read_b32, lane l takes word 16 v + l % 16: every
// 16-lane DPP row holds words 16 v .. 16 v + 15) and each FMA picks its word with the DPP modifier
// row_newbcast:k of v_fmac -- a broadcast inside the VALU operand path, no instruction of its own:
// 5 LDS reads per 32-coil row instead of 17 four-times-wider ones.  The compiler does not fold
// __builtin_amdgcn_update_dpp into the FMA (it emits v_mov_dpp + v_fmac: +50 % instructions), hence the
// asm blocks; each starts with s_nop 1 -- two wait states cover "VALU writes a VGPR, DPP reads it",
// should the register allocator ever put a copy in front (the hazard recogniser does not look into
// inline asm).  DPP reads lanes that EXEC disables as "no write": no thread leaves before the row loop
// ends.  Output: work[(sg, n, 3 + 2 nC, nT)] -- the layout of the generic pass 2, which sums the spin
// groups in fixed order: deterministic, no atomics.
// =============================================================================================
#define MRPHY_DPP_TAIL " row_mask:0xf bank_mask:0xf\n\t"
// one coil k of a group of 8 (named asm operands r0-r7 = gRe, i0-i7 = gIm, vr / vi = the vectors holding
// b1r / b1i of the group, gx / gy = the thread's gBx / gBy), L = the coil's lane in its 16-lane DPP row
#define MRPHY_DPP_RE1(OP, k, L) OP " %[r" #k "], %[vr], %[gx] row_newbcast:" #L MRPHY_DPP_TAIL
#define MRPHY_DPP_RE2(OP, k, L) OP " %[r" #k "], %[vi], %[gy] row_newbcast:" #L MRPHY_DPP_TAIL
#define MRPHY_DPP_IM1(OP, k, L) OP " %[i" #k "], %[vr], %[gy] row_newbcast:" #L MRPHY_DPP_TAIL
#define MRPHY_DPP_IM2(OP, k, L) OP " %[i" #k "], -%[vi], %[gx] row_newbcast:" #L MRPHY_DPP_TAIL
#define MRPHY_DPP_X(M, OP, L0, L1, L2, L3, L4, L5, L6, L7)                                           \
    M(OP, 0, L0) M(OP, 1, L1) M(OP, 2, L2) M(OP, 3, L3) M(OP, 4, L4) M(OP, 5, L5) M(OP, 6, L6) M(OP, 7, L7)
#define MRPHY_DPP_X_(M, OP, ...) MRPHY_DPP_X(M, OP, __VA_ARGS__)
#define MRPHY_DPP_LO 0, 1, 2, 3, 4, 5, 6, 7
#define MRPHY_DPP_HI 8, 9, 10, 11, 12, 13, 14, 15
// the four passes in turn, so that the two FMAs into one accumulator are 16 instructions apart
#define MRPHY_DPP_COILS8(OP, LRS, LIS)                                                               \
    "s_nop 1\n\t" MRPHY_DPP_X_(MRPHY_DPP_RE1, OP, LRS) MRPHY_DPP_X_(MRPHY_DPP_IM1, OP, LRS)           \
    MRPHY_DPP_X_(MRPHY_DPP_RE2, OP, LIS) MRPHY_DPP_X_(MRPHY_DPP_IM2, OP, LIS)
#define MRPHY_DPP_OPERANDS(aR, aI, vr_, vi_, gx_, gy_)                                               \
    : [r0] "+v"(aR[0]), [r1] "+v"(aR[1]), [r2] "+v"(aR[2]), [r3] "+v"(aR[3]), [r4] "+v"(aR[4]),      \
      [r5] "+v"(aR[5]), [r6] "+v"(aR[6]), [r7] "+v"(aR[7]), [i0] "+v"(aI[0]), [i1] "+v"(aI[1]),      \
      [i2] "+v"(aI[2]), [i3] "+v"(aI[3]), [i4] "+v"(aI[4]), [i5] "+v"(aI[5]), [i6] "+v"(aI[6]),      \
      [i7] "+v"(aI[7])                                                                               \
    : [vr] "v"(vr_), [vi] "v"(vi_), [gx] "v"(gx_), [gy] "v"(gy_)

// 8 coils' worth of the rf sums.  LR / LI: 0 or 8 = the group's first lane in the b1r / b1i vector.
template <int LR, int LI>
__device__ __forceinline__ void dpp_coils8(float* aR, float* aI, float vr, float vi, float gx, float gy)
{
    static_assert((LR == 0 || LR == 8) && (LI == 0 || LI == 8), "lane group");
    if constexpr (LR == 0 && LI == 0)
        asm(MRPHY_DPP_COILS8("v_fmac_f32_dpp", MRPHY_DPP_LO, MRPHY_DPP_LO) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
    else if constexpr (LR == 8 && LI == 8)
        asm(MRPHY_DPP_COILS8("v_fmac_f32_dpp", MRPHY_DPP_HI, MRPHY_DPP_HI) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
    else if constexpr (LR == 0 && LI == 8)
        asm(MRPHY_DPP_COILS8("v_fmac_f32_dpp", MRPHY_DPP_LO, MRPHY_DPP_HI) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
    else
        asm(MRPHY_DPP_COILS8("v_fmac_f32_dpp", MRPHY_DPP_HI, MRPHY_DPP_LO) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
}
template <int LR, int LI>
__device__ __forceinline__ void dpp_coils8(double* aR, double* aI, double vr, double vi, double gx, double gy)
{
    static_assert((LR == 0 || LR == 8) && (LI == 0 || LI == 8), "lane group");
    if constexpr (LR == 0 && LI == 0)
        asm(MRPHY_DPP_COILS8("v_fmac_f64_dpp", MRPHY_DPP_LO, MRPHY_DPP_LO) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
    else if constexpr (LR == 8 && LI == 8)
        asm(MRPHY_DPP_COILS8("v_fmac_f64_dpp", MRPHY_DPP_HI, MRPHY_DPP_HI) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
    else if constexpr (LR == 0 && LI == 8)
        asm(MRPHY_DPP_COILS8("v_fmac_f64_dpp", MRPHY_DPP_LO, MRPHY_DPP_HI) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
    else
        asm(MRPHY_DPP_COILS8("v_fmac_f64_dpp", MRPHY_DPP_HI, MRPHY_DPP_LO) MRPHY_DPP_OPERANDS(aR, aI, vr, vi, gx, gy));
}
// grad_gr: gG[i] += loc[i] gBz, loc in lanes 0..2 of its vector
__device__ __forceinline__ void dpp_loc3(float* aG, float vl, float gz)
{
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %3, %4 row_newbcast:0" MRPHY_DPP_TAIL
        "v_fmac_f32_dpp %1, %3, %4 row_newbcast:1" MRPHY_DPP_TAIL
        "v_fmac_f32_dpp %2, %3, %4 row_newbcast:2" MRPHY_DPP_TAIL
        : "+v"(aG[0]), "+v"(aG[1]), "+v"(aG[2]) : "v"(vl), "v"(gz));
}
__device__ __forceinline__ void dpp_loc3(double* aG, double vl, double gz)
{
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %3, %4 row_newbcast:0" MRPHY_DPP_TAIL
        "v_fmac_f64_dpp %1, %3, %4 row_newbcast:1" MRPHY_DPP_TAIL
        "v_fmac_f64_dpp %2, %3, %4 row_newbcast:2" MRPHY_DPP_TAIL
        : "+v"(aG[0]), "+v"(aG[1]), "+v"(aG[2]) : "v"(vl), "v"(gz));
}

// Geometry, read by the kernel and its launcher.  TP time points per thread (template parameter of the
// kernel: the launcher picks it per capacity); a staged row = [b1r 0..MC-1 | b1i 0..MC-1 | loc x y z, 0 x 13],
// zero beyond nC, i.e. NV = MC / 8 + 1 DPP vectors of 16 words.
template <typename T, int MC>
struct BwdStepGeom {
    static_assert(MC == 8 || MC == 16 || MC == 32, "coil capacities: 8/16/32");
    static constexpr int NVB = MC / 8;               // vectors holding b1
    static constexpr int PW = 2 * MC + 16;           // words per staged row
    static constexpr int GROUP = 64;                 // rows per LDS stage: 64 PW words = 20 KB fp32 at 32 coils
    static constexpr int U = 4;                      // rows whose gB loads are in flight per thread
};

template <typename T, int MC, int TP>
__global__ __launch_bounds__(256, 2) void k_rfgr2beff_bwd_steps(BeffBwdArgs<T> a)
{
    using G = BwdStepGeom<T, MC>;
    constexpr int NE = 3 * TP;                         // contiguous elements of a row the thread reads
    using gvec = ElemRun<T, NE>;
    const int64_t L = 3 * a.nT, nT = a.nT;             // the launcher guarantees nT >= TP
    const int64_t t0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * TP;
    const int64_t sg = blockIdx.y, n = blockIdx.z;
    const int64_t s0 = sg * a.spins_per_group;
    const int64_t s1 = (s0 + a.spins_per_group < a.nM) ? s0 + a.spins_per_group : a.nM;
    const int nC = (int)a.nC;
    __shared__ __attribute__((aligned(16))) T sc[G::GROUP][G::PW];
    // every thread stays in the loop (DPP wants all lanes enabled): one at or beyond the row end re-reads
    // the row's last TP time points (slot q of the thread = time point tr + q) and stores only its own
    const int64_t tr = (t0 + TP <= nT) ? t0 : nT - TP;
    const int l16 = threadIdx.x & 15;

    T aR[TP][MC], aI[TP][MC], aG[TP][3];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
#pragma unroll
        for (int c = 0; c < MC; ++c) aR[j][c] = aI[j][c] = T(0);
        aG[j][0] = aG[j][1] = aG[j][2] = T(0);
    }
    for (int64_t sb0 = s0; sb0 < s1; sb0 += G::GROUP) {
        const int cnt = (int)((s1 - sb0 < G::GROUP) ? s1 - sb0 : G::GROUP);
        // stage the rows' coefficient words; rows beyond cnt are ZERO, so that the row loop below can run
        // whole groups of U rows (a row past the end re-reads the last row's gB against zeros)
        const T* b1s = a.b1 + (n * a.nM + sb0) * 2 * nC;
        const T* locs = a.loc + (n * a.nM + sb0) * 3;
        __syncthreads();
        for (int i = threadIdx.x; i < G::GROUP * G::PW; i += 256) {
            const int r = i / G::PW, k = i - r * G::PW;
            const int part = k >= MC, c = k - part * MC;             // meaningful for k < 2 MC
            const bool isb = (k < 2 * MC) && (c < nC) && (r < cnt);
            const bool isl = (k >= 2 * MC) && (k < 2 * MC + 3) && (r < cnt);
            T v = T(0);
            if (isb) v = b1s[r * 2 * nC + part * nC + c];
            if (isl) v = locs[r * 3 + (k - 2 * MC)];
            sc[r][k] = v;
        }
        __syncthreads();
        const T* src0 = a.gB + (n * a.nM + sb0) * L + 3 * tr;
        for (int i = 0; i < cnt; i += G::U) {
            gvec g[G::U];
#pragma unroll
            for (int u = 0; u < G::U; ++u) {
                const int ir = (i + u < cnt) ? i + u : cnt - 1;
                g[u] = load_run_nt<T, NE>(src0 + (int64_t)ir * L);
            }
#pragma unroll
            for (int u = 0; u < G::U; ++u) {
                T vb[G::NVB + 1];
#pragma unroll
                for (int v = 0; v <= G::NVB; ++v) vb[v] = sc[i + u][16 * v + l16];
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const T gx = g[u].v[3 * j], gy = g[u].v[3 * j + 1], gz = g[u].v[3 * j + 2];
                    // coil 8 cg + k: b1r is word 8 cg + k of the row, b1i word MC + 8 cg + k
#pragma unroll
                    for (int cg = 0; cg < MC / 8; ++cg) {
                        if constexpr (MC == 8)
                            dpp_coils8<0, 8>(&aR[j][0], &aI[j][0], vb[0], vb[0], gx, gy);
                        else if (cg % 2 == 0)
                            dpp_coils8<0, 0>(&aR[j][8 * cg], &aI[j][8 * cg], vb[cg / 2], vb[MC / 16 + cg / 2], gx, gy);
                        else
                            dpp_coils8<8, 8>(&aR[j][8 * cg], &aI[j][8 * cg], vb[cg / 2], vb[MC / 16 + cg / 2], gx, gy);
                    }
                    dpp_loc3(&aG[j][0], vb[G::NVB], gz);
                }
            }
        }
    }
    const int64_t K = 3 + 2 * (int64_t)nC;
    T* w = a.work + ((sg * a.N + n) * K) * nT;
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        const int64_t t = tr + q;
        if (t < t0) continue;                           // a tail thread's re-read time points: not its own
        w[0 * nT + t] = aG[q][0]; w[1 * nT + t] = aG[q][1]; w[2 * nT + t] = aG[q][2];
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < nC) { w[(3 + c) * nT + t] = aR[q][c]; w[(3 + nC + c) * nT + t] = aI[q][c]; }
    }
}

#endif  // MRPHY_DEV_KNOBS

// =============================================================================================
// The same blocking with the row's coefficients in SGPRs (round 3, second version; the default).
// The DPP build above removed the LDS wall, but measured (round 3: profiles/r03_valu_operand_rates.txt) a v_fmac with a DPP
// source issues at HALF the rate of a plain one on this part (2.0 vs 1.0 ns per wave-instruction and
// SIMD; v_mov_dpp or v_readlane in front of plain FMAs cost 14-19 cycles each) -- and a v_fmac whose
// source is an SGPR runs at the full rate.  A row's b1 and loc are wave-uniform, so they belong in
// SGPRs: a small pre-pass (k_pack_coefs) writes them once, zero-padded, to pk[row][2 MC + 4] =
// [b1r 0..MC-1 | b1i 0..MC-1 | loc x y z, 0] in the workspace; the main pass reads a row with scalar
// loads (constant address space: s_load_dwordx16, batched, all in bounds thanks to the padding) and
// every FMA takes its coefficient straight from an SGPR.  No LDS, no barriers, no inline asm.  The
// capacities are fine-grained (4, 8, 12, 16, 24, 32: no LDS or register tile depends on them here), so
// a coil count pays for at most a third more coils than it has.  (Skipping the coil groups beyond nC
// with wave-uniform branches inside ONE 32-coil build was tried first: the compiler sinks the scalar
// loads into the branches, three exposed scalar-load round trips per row.)
// =============================================================================================
template <typename T>
struct PackArgs {
    const T* b1; const T* loc; T* pk;
    int64_t rows, nC; int MC;
    int64_t c0, nCtot;      // this block of coils: c0 .. c0 + nC - 1 of nCtot (round 4: coil counts above 32 in blocks)
};
template <typename T>
__global__ __launch_bounds__(256) void k_pack_coefs(PackArgs<T> a)
{
    const int PW = 2 * a.MC + 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.rows * PW) return;
    const int64_t r = i / PW;
    const int k = (int)(i - r * PW);
    const int part = k >= a.MC, c = k - part * a.MC;
    T v = T(0);
    if (k < 2 * a.MC) { if (c < a.nC) v = a.b1[r * 2 * a.nCtot + part * a.nCtot + a.c0 + c]; }
    else if (k < 2 * a.MC + 3) v = a.loc[r * 3 + (k - 2 * a.MC)];
    a.pk[i] = v;
}

template <typename T>
struct BeffBwdPkArgs {
    const T* gB;      // (N, nM, nT, 3)
    const T* pk;      // (N nM, 2 MC + 4): packed coefficient rows
    T* work;          // (nSG, N, 3 + 2 nCtot, nT)
    int64_t N, nM, nT, nC, spins_per_group;
    int K, rowR, rowI;    // workspace rows per (spin group, n): K = 3 + 2 nCtot; this block's coil c -> rows rowR + c
                          // (= 3 + c0 + c) and rowI + c (= 3 + nCtot + c0 + c); rowR == 3 also writes grad_gr's rows
};

#ifndef K0ADJ_U
#define K0ADJ_U 4
#endif
template <typename T, int MC, int TP>
__global__ __launch_bounds__(256, 2) void k_rfgr2beff_bwd_sgpr(BeffBwdPkArgs<T> a)
{
    // U rows' gB loads are issued together, ahead of the arithmetic on them.  (Requesting the NEXT group
    // before computing this one -- a register double buffer -- was slower at every coil count: the
    // compiler splits and scatters the loads through the group, 0.61 -> 1.04 ms at 2 coils; U = 8: no gain.)
    constexpr int NE = 3 * TP, PW = 2 * MC + 4, U = K0ADJ_U, H = MC / 2;
    static_assert(MC % 2 == 0, "coil pairs");
    using gvec = ElemRun<T, NE>;
    using CP = const T __attribute__((address_space(4)))*;
    typedef T V2 __attribute__((ext_vector_type(2)));   // a coil PAIR: v_pk_fma_f32 for float
    const int64_t L = 3 * a.nT, nT = a.nT;             // the launcher guarantees nT >= TP
    const int64_t t0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * TP;
    const int64_t sg = blockIdx.y, n = blockIdx.z;
    const int64_t s0 = sg * a.spins_per_group;
    const int64_t s1 = (s0 + a.spins_per_group < a.nM) ? s0 + a.spins_per_group : a.nM;
    const int nC = (int)a.nC;
    // a thread at or beyond the row end re-reads the row's last TP time points (slot q = time point
    // tr + q) and stores only its own
    const int64_t tr = (t0 + TP <= nT) ? t0 : nT - TP;
    V2 aR[TP][H], aI[TP][H];
    T aG[TP][3];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
#pragma unroll
        for (int k = 0; k < H; ++k) aR[j][k] = aI[j][k] = V2{T(0), T(0)};
        aG[j][0] = aG[j][1] = aG[j][2] = T(0);
    }
    const int64_t cnt = s1 - s0;
    const T* src0 = a.gB + (n * a.nM + s0) * L + 3 * tr;
    CP pk0 = (CP)(a.pk + (n * a.nM + s0) * PW);
    for (int64_t i = 0; i < cnt; i += U) {
        gvec g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t ir = (i + u < cnt) ? i + u : cnt - 1;
            g[u] = load_run_nt<T, NE>(src0 + ir * L);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // no branch in here (it would sink the U loads above into the rows that use them): a row past
            // the group's end re-reads the last row, against zeroed gB
            const bool on = i + u < cnt;
            CP q = pk0 + (on ? i + u : cnt - 1) * PW;
            T cf[PW];
#pragma unroll
            for (int k = 0; k < PW; ++k) cf[k] = q[k];  // scalar loads, all issued before the first FMA
#pragma unroll
            for (int e = 0; e < NE; ++e) g[u].v[e] = on ? g[u].v[e] : T(0);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const T gx = g[u].v[3 * j], gy = g[u].v[3 * j + 1], gz = g[u].v[3 * j + 2];
                const V2 gx2 = {gx, gx}, gy2 = {gy, gy}, ngx2 = {-gx, -gx};
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    const V2 br = {cf[2 * k], cf[2 * k + 1]}, bi = {cf[MC + 2 * k], cf[MC + 2 * k + 1]};
                    aR[j][k] = __builtin_elementwise_fma(bi, gy2, __builtin_elementwise_fma(br, gx2, aR[j][k]));
                    aI[j][k] = __builtin_elementwise_fma(bi, ngx2, __builtin_elementwise_fma(br, gy2, aI[j][k]));
                }
                aG[j][0] = fma_(cf[2 * MC], gz, aG[j][0]);
                aG[j][1] = fma_(cf[2 * MC + 1], gz, aG[j][1]);
                aG[j][2] = fma_(cf[2 * MC + 2], gz, aG[j][2]);
            }
        }
    }
    T* w = a.work + ((sg * a.N + n) * a.K) * nT;
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        const int64_t t = tr + q;
        if (t < t0) continue;                           // a tail thread's re-read time points: not its own
        if (a.rowR == 3) { w[0 * nT + t] = aG[q][0]; w[1 * nT + t] = aG[q][1]; w[2 * nT + t] = aG[q][2]; }
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < nC) {
                w[(a.rowR + c) * nT + t] = (c & 1) ? aR[q][c / 2].y : aR[q][c / 2].x;
                w[(a.rowI + c) * nT + t] = (c & 1) ? aI[q][c / 2].y : aI[q][c / 2].x;
            }
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

