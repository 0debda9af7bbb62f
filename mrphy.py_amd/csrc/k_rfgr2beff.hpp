// k_rfgr2beff.hpp -- K0 (rfgr2beff) and its adjoint to rf, gr
// Fragment of the single translation unit mrphy_hip.hip: included there INSIDE its anonymous
// namespace, after <hip/hip_runtime.h>, include/mrphy_hip.h and bloch_math.hpp.  Not a standalone
// header.

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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 f32x2_u __attribute__((aligned(4)));

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
            if (a.nt) __builtin_nontemporal_store(vec_pack(o), reinterpret_cast<typename V16<T>::utype*>(dst));
            else *reinterpret_cast<typename V16<T>::utype*>(dst) = vec_pack(o);
        } else if (sizeof(T) == 4 && VW == 2 && e0 + VW <= L) {           // 16-coil build: 8-byte stores
            const f32x2 v = {float(o[0]), float(o[VW - 1])};
            if (a.nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x2_u*>(dst));
            else *reinterpret_cast<f32x2_u*>(dst) = v;
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
    static_assert(MC == 8 || MC == 16 || MC == 32, "coil capacities: 8/16/32");
    static constexpr int TP = 32 / MC;               // time points per thread: 64 rf registers
    static constexpr int VW = 3 * TP;                // elements per thread
    // rows per block: the thread's 64 rf registers are filled with strided (uncoalesced) loads once per
    // block, so a block must walk far more rows than the 16 of the single-coil kernel to amortise them
    // (dev knob MRPHY_K0_VARIANT picks fewer for A/B); LDS: ROWS (2 MC + 4) words <= 34 KB
    static constexpr int ROWS = 128;
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
    const int64_t nT = a.nT, nC = a.nC;

    // the thread's pulse samples: TP time points x MC coils (zero beyond nC), gradient samples
    T rr[TP][MC], ri[TP][MC], px[TP], py[TP], pz[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int64_t t = (t0 + j < nT) ? t0 + j : nT - 1;
        px[j] = gr[t]; py[j] = gr[nT + t]; pz[j] = gr[2 * nT + t];
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            rr[j][c] = (c < nC) ? rf[t * nC + c] : T(0);
            ri[j][c] = (c < nC) ? rf[(nT + t) * nC + c] : T(0);
        }
    }
    // rows' b1: [re c0..MC-1 | im c0..MC-1], ZERO beyond nC; per-spin loc and df/gamma
    __shared__ __attribute__((aligned(16))) T sb1[G::ROWS][2 * MC];
    __shared__ __attribute__((aligned(16))) T sp[G::ROWS][4];
    for (int64_t i = threadIdx.x; i < (s1 - s0) * 2 * MC; i += K0_THREADS) {
        const int64_t r_ = i / (2 * MC), k_ = i - r_ * 2 * MC;
        const int64_t part = k_ / MC, c = k_ - part * MC;
        sb1[r_][k_] = (c < nC) ? a.b1[(n * a.nM + s0 + r_) * 2 * nC + part * nC + c] : T(0);
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
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            T Bx = T(0), By = T(0);
#pragma unroll
            for (int c = 0; c < MC; ++c) field_xy_fma<T>(b[c], b[MC + c], rr[j][c], ri[j][c], Bx, By);
            o[3 * j] = Bx; o[3 * j + 1] = By;
            o[3 * j + 2] = field_z<T>(px[j], py[j], pz[j], lx, ly, lz, delta);
        }
        T* dst = a.beff + (n * a.nM + s) * L + 3 * t0;
        if (full) {
            constexpr int VE = V16<T>::N;              // elements per 16-B vector: 4 floats / 2 doubles
            constexpr int NV = (3 * TP) / VE;          // whole vectors; the remainder goes element-wise
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (a.nt) __builtin_nontemporal_store(vec_pack(o + v * VE), reinterpret_cast<typename V16<T>::utype*>(dst + v * VE));
                else *reinterpret_cast<typename V16<T>::utype*>(dst + v * VE) = vec_pack(o + v * VE);
            }
#pragma unroll
            for (int e = NV * VE; e < 3 * TP; ++e) {
                if (a.nt) __builtin_nontemporal_store(o[e], dst + e);
                else dst[e] = o[e];
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
// of packed FMAs over PAIRS OF TIME POINTS (round 3; see the operand-path table in DESIGN.md).
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
                    if (a.nt) __builtin_nontemporal_store(w, reinterpret_cast<f32x2_u*>(dst + 2 * v));
                    else *reinterpret_cast<f32x2_u*>(dst + 2 * v) = w;
                }
            } else {
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    if (a.nt) __builtin_nontemporal_store(vec_pack(o + 2 * v), reinterpret_cast<typename V16<T>::utype*>(dst + 2 * v));
                    else *reinterpret_cast<typename V16<T>::utype*>(dst + 2 * v) = vec_pack(o + 2 * v);
                }
            }
        } else {
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];  // the row's last time point (nT odd)
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
constexpr int BWD_MAXC = 32;           // largest coil capacity of the one-pass adjoint
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
// The DPP build above removed the LDS wall, but measured (tools/dbg/dpp_rate.hip) a v_fmac with a DPP
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
    if (k < 2 * a.MC) { if (c < a.nC) v = a.b1[r * 2 * a.nC + part * a.nC + c]; }
    else if (k < 2 * a.MC + 3) v = a.loc[r * 3 + (k - 2 * a.MC)];
    a.pk[i] = v;
}

template <typename T>
struct BeffBwdPkArgs {
    const T* gB;      // (N, nM, nT, 3)
    const T* pk;      // (N nM, 2 MC + 4): packed coefficient rows
    T* work;          // (nSG, N, 3 + 2 nC, nT)
    int64_t N, nM, nT, nC, spins_per_group;
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
    const int64_t K = 3 + 2 * (int64_t)nC;
    T* w = a.work + ((sg * a.N + n) * K) * nT;
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        const int64_t t = tr + q;
        if (t < t0) continue;                           // a tail thread's re-read time points: not its own
        w[0 * nT + t] = aG[q][0]; w[1 * nT + t] = aG[q][1]; w[2 * nT + t] = aG[q][2];
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < nC) {
                w[(3 + c) * nT + t] = (c & 1) ? aR[q][c / 2].y : aR[q][c / 2].x;
                w[(3 + nC + c) * nT + t] = (c & 1) ? aI[q][c / 2].y : aI[q][c / 2].x;
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
