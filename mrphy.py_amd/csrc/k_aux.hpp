// k_aux.hpp -- freeprec, interpT, beff2uphi/uphirot, beff2ab/blochsim_ab, mask gather/scatter, cube_loc
// Fragment of the single translation unit mrphy_hip.hip: included there INSIDE its anonymous
// namespace, after <hip/hip_runtime.h>, include/mrphy_hip.h and bloch_math.hpp.  Not a standalone
// header.

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

// ---------------------------------------------------------------------------------------------
// Pulse.interpT with the one-tap kinds of scipy's interp1d ('nearest', 'nearest-up', 'previous',
// 'next', 'zero'; reference mobjs.py:201,214-215 passes `kind` through): every output sample IS one
// sample of the zero-prepended source, chosen by the grid alone.  The host supplies sel[j] in
// [0, nTo] -- 0 = the prepended zero sample, k >= 1 = y[k - 1] -- taken from scipy itself for that
// grid (mrphy_amd/interp.py), non-decreasing in j; no arithmetic, so the result is bit-identical to
// the reference's.  The host validates the range of sel before launching: every read is in bounds.
//   fwd: out[ch, j] = sel[j] ? y[ch, sel[j] - 1] : 0
//   bwd: gy[ch, i]  = sum over { j : sel[j] == i + 1 } of gout[ch, j], added in j order
//        (two binary searches in the monotone sel: gather form, deterministic)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_interp_sel_fwd(const T* y, T* out, const int* sel,
                                                        int64_t nch, int64_t nTo, int64_t nTn)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ch = blockIdx.y;
    if (j >= nTn) return;
    const int k = sel[j];
    out[ch * nTn + j] = k == 0 ? T(0) : y[ch * nTo + (k - 1)];
}

template <typename T>
__global__ __launch_bounds__(256) void k_interp_sel_bwd(const T* gout, T* gy, const int* sel,
                                                        int64_t nch, int64_t nTo, int64_t nTn)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ch = blockIdx.y;
    if (i >= nTo) return;
    const int64_t j0 = lower_bound_lo(sel, nTn, (int)i + 1);
    const int64_t j1 = lower_bound_lo(sel, nTn, (int)i + 2);
    const T* go = gout + ch * nTn;
    double acc = 0.0;
    for (int64_t j = j0; j < j1; ++j) acc += double(go[j]);
    gy[ch * nTo + i] = T(acc);
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

// Adjoints of the two helpers (the reference gets them from autograd over plain torch ops,
// beffective.py:35-36, utils.py:351-357).
//   beff2uphi: n = |b|, d = max(n, eps), U = b/d, Phi = -n g
//     n > eps : gb = gU/n - b (b.gU)/n^3 - gPhi g b/n        (F.normalize + norm backward)
//     n <= eps: gb = gU/eps - gPhi g b/n   (clamp_min cuts the U branch from the norm; the Phi
//                                            branch stays; at n = 0 torch's norm backward gives 0)
//     gg (per spin, optional) = -gPhi n
template <typename T, typename CT>
__global__ __launch_bounds__(256) void k_beff2uphi_bwd(const T* b, Bc g, const T* gU, const T* gPhi,
                                                       T* gb, T* gg, int64_t rows, int64_t nM)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const T x = b[r * 3], y = b[r * 3 + 1], z = b[r * 3 + 2];
    const T ux = gU ? gU[r * 3] : T(0), uy = gU ? gU[r * 3 + 1] : T(0), uz = gU ? gU[r * 3 + 2] : T(0);
    const T gp = gPhi ? gPhi[r] : T(0);
    const T nrm = sqrt_(x * x + y * y + z * z);
    const CT gam = bc_load<CT>(g, r / nM, r % nM);
    T ox, oy, oz;
    if (nrm > T(1e-12)) {
        const T rn = T(1) / nrm;
        const T bu = (x * ux + y * uy + z * uz) * rn * rn;          // (b.gU)/n^2
        const T kp = T(CT(gp) * gam);
        ox = (ux - x * bu - kp * x) * rn;
        oy = (uy - y * bu - kp * y) * rn;
        oz = (uz - z * bu - kp * z) * rn;
    } else {
        ox = ux * T(1e12); oy = uy * T(1e12); oz = uz * T(1e12);
        if (nrm > T(0)) {                   // 0 < n <= eps: the clamp cuts the U branch only; torch's
            const T kp = T(CT(gp) * gam) / nrm;          // norm backward still passes -gPhi g b/n
            ox -= kp * x; oy -= kp * y; oz -= kp * z;
        }
    }
    if (gb) { gb[r * 3] = ox; gb[r * 3 + 1] = oy; gb[r * 3 + 2] = oz; }
    if (gg) gg[r] = -(gp * nrm);
}

//   uphirot: Vo = c V + (1-c)(U.V) U + s UxV, G = dL/dVo; per row, summed over the nV vectors:
//     gV   = c G + (1-c)(U.G) U - s UxG
//     gPhi = sum_v G.(-s V + s (U.V) U + c UxV)
//     gU   = sum_v (1-c)[(U.V) G + (U.G) V] + s VxG          (U treated as free, as autograd does)
template <typename T>
__global__ __launch_bounds__(256) void k_uphirot_bwd(const T* U, const T* Phi, const T* Vi,
                                                     const T* G, T* gU, T* gPhi, T* gVi,
                                                     int64_t rows, int64_t nV)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const T ux = U[r * 3], uy = U[r * 3 + 1], uz = U[r * 3 + 2];
    T sp, cp;
    sincos_(Phi[r], &sp, &cp);
    const T omc = T(1) - cp;
    T ax = T(0), ay = T(0), az = T(0), ap = T(0);
    for (int64_t v = 0; v < nV; ++v) {
        const T* vi = Vi + r * 3 * nV + v;
        const T* gi = G + r * 3 * nV + v;
        const T x = vi[0], y = vi[nV], z = vi[2 * nV];
        const T gx = gi[0], gy = gi[nV], gz = gi[2 * nV];
        const T uv = ux * x + uy * y + uz * z, ug = ux * gx + uy * gy + uz * gz;
        const T cx = uy * z - uz * y, cy = uz * x - ux * z, cz = ux * y - uy * x;      // U x V
        if (gVi) {
            T* o = gVi + r * 3 * nV + v;
            o[0]      = cp * gx + omc * ug * ux - sp * (uy * gz - uz * gy);
            o[nV]     = cp * gy + omc * ug * uy - sp * (uz * gx - ux * gz);
            o[2 * nV] = cp * gz + omc * ug * uz - sp * (ux * gy - uy * gx);
        }
        ap += gx * (sp * (uv * ux - x) + cp * cx) + gy * (sp * (uv * uy - y) + cp * cy)
            + gz * (sp * (uv * uz - z) + cp * cz);
        ax += omc * (uv * gx + ug * x) + sp * (y * gz - z * gy);
        ay += omc * (uv * gy + ug * y) + sp * (z * gx - x * gz);
        az += omc * (uv * gz + ug * z) + sp * (x * gy - y * gx);
    }
    if (gU) { gU[r * 3] = ax; gU[r * 3 + 1] = ay; gU[r * 3 + 2] = az; }
    if (gPhi) gPhi[r] = ap;
}

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

constexpr int AB_HIST_STEP = 12 * WAVE;            // elements per time step of one tile

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
};

template <typename T, typename CT, int TC>
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

    // one adjoint step for the four columns; returns dL/dB of this step
    auto step4 = [&](const RotAdj<T>& ra, int64_t t, T& gx, T& gy, T& gz) {
        const T* q = hp + t * AB_HIST_STEP;
        gx = gy = gz = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T m0 = __builtin_nontemporal_load(q + (3 * j) * WAVE);
            const T m1 = __builtin_nontemporal_load(q + (3 * j + 1) * WAVE);
            const T m2 = __builtin_nontemporal_load(q + (3 * j + 2) * WAVE);
            T ax, ay, az;
            rot_apply_adj<true, T, CT>(k, ra, m0, m1, m2, hx[j], hy[j], hz[j], ax, ay, az);
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
            step4(ra[0], t, gx, gy, gz);
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
                    step4(ra[q], c * TC + tt + q, gg[3 * q], gg[3 * q + 1], gg[3 * q + 2]);
                *reinterpret_cast<V*>(rowB + tt * 3) = vec_pack(gg);
                *reinterpret_cast<V*>(rowB + tt * 3 + VE) = vec_pack(gg + VE);
                *reinterpret_cast<V*>(rowB + tt * 3 + 2 * VE) = vec_pack(gg + 2 * VE);
            }
            __syncthreads();
            chunk_store<T, TC>(tileB, a.gBeff, row0, a.rows, rowlen, c * TC, lane);
        }
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

// =============================================================================================
// Diagnostic: which XCD (HW_REG_XCC_ID, bits 3:0) each workgroup of a 1-D grid runs on.  The write
// kernels assume only that blocks b and b + 8 share an XCD (round-robin dealing); this shows what a
// process actually gets.
// =============================================================================================
__global__ __launch_bounds__(64) void k_xcc_map(int32_t* out, int64_t nblocks)
{
    if (threadIdx.x == 0 && (int64_t)blockIdx.x < nblocks)
        out[blockIdx.x] = (int32_t)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);   // XCC_ID
}
