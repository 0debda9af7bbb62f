// k_rfgr2beff.hpp -- K0 (rfgr2beff) and its adjoint to rf, gr
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

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
    // coil blocking (k_rfgr2beff_steps only): this launch adds coils [c0, c0 + nC) of nCt; acc: the chain of Bx, By
    // continues from the values in beff (the launches before this one) instead of starting at zero
    int64_t c0, nCt;
    int acc;
    int rows_per_block;
    int nt;                          // store policy (k_common.hpp: store_pol): 0 plain, 1 nt, 2 sc1 nt
    unsigned gy;                     // > 0: grid.x = spin tile * gy + time tile (time tile fastest)
    unsigned nblk, per_xcd;          // per_xcd > 0: block b works on tile (b % 8) * per_xcd + b / 8
};

constexpr int K0_THREADS = 256;
constexpr int K0_MAX_ROWS = 256;

// Geometry of the multi-coil builds, read by BOTH the kernel and its launcher (run_rfgr2beff) so
// that they cannot disagree: coil capacity MC (= NCM for NCM >= 8), elements per thread VW -- the
// thread keeps the rf samples of its VW time points for all MC coils in 2 VW MC registers, so VW
// shrinks as MC grows (64 registers throughout) -- and the most rows of b1 a block stages in LDS.
// From 16 coils on the kernel is VALU-bound (>= 64 FMAs per 12 B written): the narrower stores
// that come with a smaller VW are not what limits it.
constexpr int K0_MC_ROWS = 64;                       // rows per block, multi-coil builds (LDS: 2 MC floats each)
template <typename T, int NCM>
struct K0Geom {
    static constexpr int MC = NCM >= 8 ? NCM : 1;
    static constexpr int VWFULL = V16<T>::N;         // 16-byte stores: 4 floats / 2 doubles
    static constexpr int VW = NCM <= 8 ? VWFULL : (NCM == 16 ? (VWFULL / 2 > 0 ? VWFULL / 2 : 1) : 1);
    static constexpr int ROWS = NCM >= 8 ? K0_MC_ROWS : K0_MAX_ROWS;
    static_assert(NCM == 0 || NCM == 1 || NCM == 8 || NCM == 16 || NCM == 32, "coil capacities: 8/16/32");
};

// NC1 = true: single coil, pulse samples in registers.  NC1 = false: any nC, coil loop reads the
// rf samples from global memory (L1/L2 resident: 8*nC bytes per time point).
// NCM: 1 = one coil; 8 / 16 / 32 = up to that many coils (the thread's rf samples of all coils in
// registers, the rows' b1 in LDS; one ascending FMA chain over the coils whatever the capacity, as
// in K2 / K2b); 0 = any coil count (rf and b1 re-read from memory per element and row: slow --
// 64^3 x 1024: 8 coils 2.1 ms, 9 coils 15.5 ms, 16 coils 43 ms before the 16 / 32 capacities).
// VW is K0Geom<T, NCM>::VW on the vector path (1 on the unaligned one); the launcher passes it.
constexpr int K0_MAXC = 32;                          // largest register/LDS coil capacity
template <typename T, int VW, int NCM>
__global__ __launch_bounds__(K0_THREADS) void k_rfgr2beff(BeffArgs<T> a)
{
    constexpr bool NC1 = (NCM == 1);
    constexpr bool NCR = (NCM >= 8);                 // coils in registers / LDS
    constexpr int MC = K0Geom<T, NCM>::MC;
    static_assert(VW == 1 || VW == K0Geom<T, NCM>::VW, "VW must come from K0Geom");
    const int64_t L = 3 * a.nT;
    // grid: x = spin tile (can be large), y = tile of the (t, xyz) axis, z = batch entry
    unsigned tile = blockIdx.x;
    if (a.per_xcd) {
        tile = MRPHY_XCD_SLOT(blockIdx.x) * a.per_xcd + (blockIdx.x >> 3);
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
    // Coil operands of the thread's elements, selected ONCE per thread so that the row loop is
    // branch-free and forms only the component an element needs -- 2 FMAs per coil instead of 4:
    //   x element:  Bx = sum_c b1r rfr + b1i (-rfi)      (cu, cv) = ( rfr, -rfi)
    //   y element:  By = sum_c b1r rfi + b1i   rfr       (cu, cv) = ( rfi,  rfr)
    //   z element:  (0, 0): the coil sum is discarded
    // acc = fma(b1r, cu, fma(b1i, cv, acc)) in ascending c is, product for product and rounding for
    // rounding, the Bx resp. By chain of field_xy_fma (b1i (-rfi) and (-b1i) rfi are the same real
    // number): bit-identical to K2 / K2b.  Coils c >= nC carry zeros on both sides (adding an exact
    // zero changes nothing but, at most, the sign of a zero sum), so the loop needs no `c < nC` test.
    // (Round 2: the guarded loop compiled to a branch and an exposed LDS round trip PER COIL --
    // ds_read2, s_waitcnt lgkmcnt(0), 4 FMAs, s_cbranch -- and the x/y-vs-z test to divergent code
    // that formed Bx AND By and kept one: 8 coils took 2.1 ms at 64^3 x 1024, 4x the write time.)
    T cu[NCR ? VW : 1][MC], cv[NCR ? VW : 1][MC];
    if (NCR) {
#pragma unroll
        for (int j = 0; j < VW; ++j)
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                const T rr_ = (c < nC) ? rf[tt[j] * nC + c] : T(0);
                const T ri_ = (c < nC) ? rf[(nT + tt[j]) * nC + c] : T(0);
                cu[j][c] = cc[j] == 0 ? rr_ : (cc[j] == 1 ? ri_ : T(0));
                cv[j][c] = cc[j] == 0 ? -ri_ : (cc[j] == 1 ? rr_ : T(0));
            }
    }
    // rows' b1: [re c0..MC-1 | im c0..MC-1], ZERO beyond nC (the row loop reads all MC).  The launcher
    // keeps rows_per_block <= K0Geom::ROWS (= the first dimension here) and nC <= MC for this build.
    __shared__ __attribute__((aligned(16))) T sb1[NCR ? K0Geom<T, NCM>::ROWS : 1][2 * MC];
    if (NCR) {
        for (int64_t i = threadIdx.x; i < (s1 - s0) * 2 * MC; i += K0_THREADS) {
            const int64_t rr_ = i / (2 * MC), k_ = i - rr_ * 2 * MC;
            const int64_t part = k_ / MC, c = k_ - part * MC;
            sb1[rr_][k_] = (c < nC) ? a.b1[(n * a.nM + s0 + rr_) * 2 * nC + part * nC + c] : T(0);
        }
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
        } else if (NCR) {
            const T* b = sb1[s - s0];                    // wave-uniform: broadcast reads, batched
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                T acc = T(0);
#pragma unroll
                for (int c = 0; c < MC; ++c)
                    acc = fma_(b[c], cu[j][c], fma_(b[MC + c], cv[j][c], acc));
                const T Bz = field_z<T>(px[j], py[j], pz[j], lx, ly, lz, delta);
                o[j] = cc[j] == 2 ? Bz : acc;
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
                        field_xy_fma<T>(b1[c], b1[nC + c], qr[c], qi[c], Bx, By);
                    o[j] = cc[j] == 0 ? Bx : By;
                }
            }
        }
        T* dst = a.beff + row * L + e0;
        if (VW == V16<T>::N && e0 + VW <= L) {        // a thread straddling the row end: per element
            store_pol(reinterpret_cast<typename V16<T>::utype*>(dst), (typename V16<T>::utype)vec_pack(o), a.nt);
        } else if (sizeof(T) == 4 && VW == 2 && e0 + VW <= L) {           // 16-coil build: 8-byte stores
            const f32x2 v = {float(o[0]), float(o[VW - 1])};
            store_pol(reinterpret_cast<f32x2_u*>(dst), (f32x2_u)v, a.nt);
        } else {
#pragma unroll
            for (int j = 0; j < VW; ++j)
                if (e0 + j < L) dst[j] = o[j];
        }
    }
}

// =============================================================================================
// K0 for parallel transmit (2..32 coils with a b1 map), round 3: a thread owns whole TIME POINTS.
//
// The element-per-thread builds above give a thread VW consecutive elements of the (t, xyz) axis:
// with coils that wastes a third of the arithmetic (a z element runs the coil loop on zeros), forms
// Bx and By of one time point in different threads -- each fetching the row's b1 from LDS for itself --
// and at 32 coils serves ONE element per 64 LDS words (7.8 ms at 64^3 x 1024, slower than 16 coils
// per coil).  Here a thread owns TP consecutive time points, i.e. 3 TP consecutive elements: the rf
// samples of its time points for all MC coils sit in 2 MC TP = 64 registers (TP = 4 / 2 / 1 at capacity
// 8 / 16 / 32), a row's b1 (2 MC LDS words, broadcast reads) serves Bx AND By of TP time points, and
// Bz needs no coil loop at all: 4 MC FMAs per time point instead of 6 MC, a third of the LDS reads per
// element.  Bx, By are field_xy_fma's chains (ascending coils), so K0 stays bit-identical to K2 / K2b.
// A thread's 12 TP bytes are contiguous (TP = 4: three 16-B stores; 2: 16 + 8; 1: one 12-B store).
// =============================================================================================
template <typename T, int MC>
struct K0StepGeom {
    static_assert(MC == 8 || MC == 16 || MC == 32 || MC == 40 || MC == 48 || MC == 64, "coil capacities");
    // time points per thread: 64 rf registers up to 32 coils; beyond (round 4: 40 / 48 / 64 coils, float only) one
    // time point and 2 MC rf registers -- 33 coils used to fall off a cliff onto the generic kernel (129 ms at
    // 64^3 x 1024 against 1.24 ms for 32)
    static constexpr int TP = MC <= 32 ? 32 / MC : 1;
    static constexpr int VW = 3 * TP;                // elements per thread
    // rows per block: the thread's 64 rf registers are filled with strided (uncoalesced) loads once per
    // block, so a block must walk far more rows than the 16 of the single-coil kernel to amortise them
    // (dev knob MRPHY_K0_VARIANT picks fewer for A/B); LDS: ROWS (2 MC + 4) words <= 34 KB
    static constexpr int ROWS = MC <= 32 ? 128 : 64;
};

template <typename T, int MC>
__global__ __launch_bounds__(K0_THREADS) void k_rfgr2beff_steps(BeffArgs<T> a)
{
    using G = K0StepGeom<T, MC>;
    constexpr int TP = G::TP;
    const int64_t L = 3 * a.nT;
    unsigned tile = blockIdx.x;
    if (a.per_xcd) {
        tile = MRPHY_XCD_SLOT(blockIdx.x) * a.per_xcd + (blockIdx.x >> 3);
        if (tile >= a.nblk) return;
    }
    const unsigned by = a.gy ? tile % a.gy : blockIdx.y;
    const unsigned bx = a.gy ? tile / a.gy : tile;
    const int64_t t0 = ((int64_t)by * K0_THREADS + threadIdx.x) * TP;     // first time point of the thread
    const int64_t n = blockIdx.z;
    const int64_t s0 = (int64_t)bx * a.rows_per_block;
    const int64_t s1 = (s0 + a.rows_per_block < a.nM) ? s0 + a.rows_per_block : a.nM;
    const T* rf = a.rf + n * a.rf_sn;
    const T* gr = a.gr + n * a.gr_sn;
    const int64_t nT = a.nT, nC = a.nC, nCt = a.nCt, c0 = a.c0;

    // the thread's pulse samples: TP time points x MC coils (zero beyond nC), gradient samples
    T rr[TP][MC], ri[TP][MC], px[TP], py[TP], pz[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int64_t t = (t0 + j < nT) ? t0 + j : nT - 1;
        px[j] = gr[t]; py[j] = gr[nT + t]; pz[j] = gr[2 * nT + t];
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            rr[j][c] = (c < nC) ? rf[t * nCt + c0 + c] : T(0);
            ri[j][c] = (c < nC) ? rf[(nT + t) * nCt + c0 + c] : T(0);
        }
    }
    // rows' b1: [re c0..MC-1 | im c0..MC-1], ZERO beyond nC; per-spin loc and df/gamma
    __shared__ __attribute__((aligned(16))) T sb1[G::ROWS][2 * MC];
    __shared__ __attribute__((aligned(16))) T sp[G::ROWS][4];
    for (int64_t i = threadIdx.x; i < (s1 - s0) * 2 * MC; i += K0_THREADS) {
        const int64_t r_ = i / (2 * MC), k_ = i - r_ * 2 * MC;
        const int64_t part = k_ / MC, c = k_ - part * MC;
        sb1[r_][k_] = (c < nC) ? a.b1[(n * a.nM + s0 + r_) * 2 * nCt + part * nCt + c0 + c] : T(0);
    }
    for (int64_t i = threadIdx.x; i < s1 - s0; i += K0_THREADS) {
        const int64_t s = s0 + i, row = n * a.nM + s;
        sp[i][0] = a.loc[row * 3]; sp[i][1] = a.loc[row * 3 + 1]; sp[i][2] = a.loc[row * 3 + 2];
        sp[i][3] = a.df.p ? bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s) : T(0);
    }
    __syncthreads();
    if (t0 >= nT) return;
    const bool full = t0 + TP <= nT;                   // else: this thread straddles the row end

    for (int64_t s = s0; s < s1; ++s) {
        const T* q = sp[s - s0];
        const T lx = q[0], ly = q[1], lz = q[2], delta = q[3];
        const T* b = sb1[s - s0];                      // wave-uniform: broadcast reads, batched
        T o[3 * TP];
        T* dst = a.beff + (n * a.nM + s) * L + 3 * t0;
        if (a.acc) {                                   // a later coil block: the chains go on from the stored Bx, By
            if (full) {
                constexpr int VE = V16<T>::N, NV = (3 * TP) / VE;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const typename V16<T>::utype w = *reinterpret_cast<const typename V16<T>::utype*>(dst + v * VE);
#pragma unroll
                    for (int e = 0; e < VE; ++e) o[v * VE + e] = w[e];
                }
#pragma unroll
                for (int e = NV * VE; e < 3 * TP; ++e) o[e] = dst[e];
            } else {
#pragma unroll
                for (int e = 0; e < 3 * TP; ++e) o[e] = (3 * t0 + e < L) ? dst[e] : T(0);
            }
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            T Bx = T(0), By = T(0);
            if (a.acc) { Bx = o[3 * j]; By = o[3 * j + 1]; }
#pragma unroll
            for (int c = 0; c < MC; ++c) field_xy_fma<T>(b[c], b[MC + c], rr[j][c], ri[j][c], Bx, By);
            o[3 * j] = Bx; o[3 * j + 1] = By;
            o[3 * j + 2] = field_z<T>(px[j], py[j], pz[j], lx, ly, lz, delta);
        }
        if (full) {
            constexpr int VE = V16<T>::N;              // elements per 16-B vector: 4 floats / 2 doubles
            constexpr int NV = (3 * TP) / VE;          // whole vectors; the remainder goes element-wise
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                store_pol(reinterpret_cast<typename V16<T>::utype*>(dst + v * VE), (typename V16<T>::utype)vec_pack(o + v * VE), a.nt);
            }
#pragma unroll
            for (int e = NV * VE; e < 3 * TP; ++e) {
                store_pol(dst + e, o[e], a.nt);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 3 * TP; ++e)
                if (3 * t0 + e < L) dst[e] = o[e];
        }
    }
}

// =============================================================================================
// K0 for parallel transmit, coil counts 4 / 8 / 12 / 16 (exactly): the rows' b1 as SCALAR operands
// of packed FMAs over PAIRS OF TIME POINTS (round 3; see the operand-path table in docs/LABNOTES.md).
//
// k_rfgr2beff_steps reads a row's b1 from LDS with wave-uniform (broadcast) reads: one word per two clocks
// per CU, 128 clocks for a 32-coil row against 256 VALU clocks per SIMD and four SIMDs per LDS -- LDS-bound
// from 16 coils on.  A row's b1 is wave-uniform, so it belongs in SGPRs, and the one FMA form that takes
// a different scalar operand every instruction at FULL rate is v_pk_fma_f32 with an SGPR source.  A
// thread owns two consecutive time points; its rf samples sit in register PAIRS {rf[t0][c], rf[t0+1][c]};
// per coil   {Bx(t0), Bx(t0+1)} = fma({b1r, b1r}, {rr0, rr1}, fma({-b1i, -b1i}, {ri0, ri1}, {Bx, Bx}))
// -- per time point exactly field_xy_fma's chain in ascending coil order, so the output is bit-identical
// to every other multi-coil build (K0, K2, K2b; asserted).  b1 rows are read straight from the caller's
// tensor, 2 nC words each: in bounds without padding because the coil count is exact (a template
// parameter); other coil counts take k_rfgr2beff_steps.  No LDS for b1; loc and df/gamma stay in LDS.
// Shipped up to 16 coils: at 24 / 32 the rows' scalar loads (256 B per wave and row, 3.3-3.6 B/ns per CU when
// waves walk their own rows) cost more than the LDS broadcasts they replace (1.30 / 1.60 vs 1.16 / 1.32 ms).
// =============================================================================================
template <typename T, int NC>
__global__ __launch_bounds__(K0_THREADS, 2) void k_rfgr2beff_pk(BeffArgs<T> a)
{
    static_assert(NC % 4 == 0 && NC >= 4 && NC <= 32, "exact coil counts: multiples of 4 up to 32");
    constexpr int ROWS = 128;
    typedef T V2 __attribute__((ext_vector_type(2)));
    using CP = const T __attribute__((address_space(4)))*;
    const int64_t L = 3 * a.nT;
    unsigned tile = blockIdx.x;
    if (a.per_xcd) {
        tile = MRPHY_XCD_SLOT(blockIdx.x) * a.per_xcd + (blockIdx.x >> 3);
        if (tile >= a.nblk) return;
    }
    const unsigned by = a.gy ? tile % a.gy : blockIdx.y;
    const unsigned bx = a.gy ? tile / a.gy : tile;
    const int64_t t0 = ((int64_t)by * K0_THREADS + threadIdx.x) * 2;      // first of the thread's two time points
    const int64_t n = blockIdx.z;
    const int64_t s0 = (int64_t)bx * a.rows_per_block;
    const int64_t s1 = (s0 + a.rows_per_block < a.nM) ? s0 + a.rows_per_block : a.nM;
    const T* rf = a.rf + n * a.rf_sn;
    const T* gr = a.gr + n * a.gr_sn;
    const int64_t nT = a.nT;
    const int64_t ta = (t0 < nT) ? t0 : nT - 1, tb = (t0 + 1 < nT) ? t0 + 1 : nT - 1;
    V2 rr[NC], ri[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        rr[c] = V2{rf[ta * NC + c], rf[tb * NC + c]};
        ri[c] = V2{rf[(nT + ta) * NC + c], rf[(nT + tb) * NC + c]};
    }
    const T pxa = gr[ta], pya = gr[nT + ta], pza = gr[2 * nT + ta];
    const T pxb = gr[tb], pyb = gr[nT + tb], pzb = gr[2 * nT + tb];
    __shared__ __attribute__((aligned(16))) T sp[ROWS][4];
    for (int64_t i = threadIdx.x; i < s1 - s0; i += K0_THREADS) {
        const int64_t s = s0 + i, row = n * a.nM + s;
        sp[i][0] = a.loc[row * 3]; sp[i][1] = a.loc[row * 3 + 1]; sp[i][2] = a.loc[row * 3 + 2];
        sp[i][3] = a.df.p ? bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s) : T(0);
    }
    __syncthreads();
    if (t0 >= nT) return;
    const bool full = t0 + 2 <= nT;
    CP b1 = (CP)(a.b1 + (n * a.nM + s0) * 2 * NC);
    for (int64_t s = s0; s < s1; ++s) {
        const T* q = sp[s - s0];
        const T lx = q[0], ly = q[1], lz = q[2], delta = q[3];
        CP b = b1 + (s - s0) * 2 * NC;                    // wave-uniform: scalar loads
        V2 Bx = {T(0), T(0)}, By = {T(0), T(0)};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const T br = b[c], bi = b[NC + c];
            const V2 br2 = {br, br}, bi2 = {bi, bi}, nbi2 = {-bi, -bi};
            Bx = __builtin_elementwise_fma(br2, rr[c], __builtin_elementwise_fma(nbi2, ri[c], Bx));
            By = __builtin_elementwise_fma(br2, ri[c], __builtin_elementwise_fma(bi2, rr[c], By));
        }
        T o[6];
        o[0] = Bx.x; o[1] = By.x; o[2] = field_z<T>(pxa, pya, pza, lx, ly, lz, delta);
        o[3] = Bx.y; o[4] = By.y; o[5] = field_z<T>(pxb, pyb, pzb, lx, ly, lz, delta);
        T* dst = a.beff + (n * a.nM + s) * L + 3 * t0;
        if (full) {
            if constexpr (sizeof(T) == 4) {               // 24 B at an 8-byte boundary: three 8-byte stores
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const f32x2 w = {float(o[2 * v]), float(o[2 * v + 1])};
                    store_pol(reinterpret_cast<f32x2_u*>(dst + 2 * v), (f32x2_u)w, a.nt);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    store_pol(reinterpret_cast<typename V16<T>::utype*>(dst + 2 * v), (typename V16<T>::utype)vec_pack(o + 2 * v), a.nt);
                }
            }
        } else {
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];  // the row's last time point (nT odd)
        }
    }
}
