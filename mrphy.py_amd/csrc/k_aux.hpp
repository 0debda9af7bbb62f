// k_aux.hpp -- freeprec, interpT, beff2uphi/uphirot, beff2ab/blochsim_ab, mask gather/scatter, cube_loc
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

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


// Gradients of freeprec w.r.t. dur, T1, T2, df per spin (round 4): what the reference's autograd returns through
// the plain torch ops of slowsims.freeprec (slowsims.py:151-174).  With phi = -2 pi df dur, a_i = -dur / T_i,
// E_i = exp(a_i), (x, y, z) the input and g the cotangent of the output:
//     dL/dphi = E2 [ g_y (c x - s y) - g_x (s x + c y) ]       dL/dE2 = g_x (c x - s y) + g_y (s x + c y)
//     dL/dE1  = g_z (z - 1)                                    (Mz' = z E1 + 1 - E1)
//     dL/ddf  = -2 pi dur dL/dphi        dL/dT_i = dL/dE_i E_i dur / T_i^2
//     dL/ddur = -2 pi df dL/dphi - dL/dE1 E1 / T1 - dL/dE2 E2 / T2
template <typename T>
__global__ __launch_bounds__(256) void k_freeprec_gc(FreePrecArgs a, const void* gMo_, void* gC_)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.rows) return;
    const int64_t n = r / a.nM, s = r % a.nM;
    const T* mi = reinterpret_cast<const T*>(a.Mi) + r * 3;
    const T* go = reinterpret_cast<const T*>(gMo_) + r * 3;
    const T x = mi[0], y = mi[1], z = mi[2], gx = go[0], gy = go[1], gz = go[2];
    const T dur = reinterpret_cast<const T*>(a.dur)[n * a.dur_sn];
    const T twopi = T(6.283185307179586476925);
    T cph = T(1), sph = T(0), e1 = T(1), e2 = T(1), df = T(0), t1 = T(1), t2 = T(1);
    if (a.df.p) {
        df = bc_load<T>(a.df, n, s);
        sincos_full(-twopi * df * dur, &sph, &cph);
    }
    if (a.T1.p) {
        t1 = bc_load<T>(a.T1, n, s); t2 = bc_load<T>(a.T2, n, s);
        e1 = exp_(-dur / t1); e2 = exp_(-dur / t2);
    }
    const T rx = cph * x - sph * y, ry = sph * x + cph * y;       // rotated, before relaxation
    const T dphi = a.df.p ? e2 * (gy * rx - gx * ry) : T(0);
    const T dE2 = a.T1.p ? gx * rx + gy * ry : T(0);
    const T dE1 = a.T1.p ? gz * (z - T(1)) : T(0);
    const T da1 = dE1 * e1, da2 = dE2 * e2;                       // dL/da_i
    T* o = reinterpret_cast<T*>(gC_) + r * 4;
    o[0] = -twopi * df * dphi - da1 / t1 - da2 / t2;
    o[1] = da1 * dur / (t1 * t1);
    o[2] = da2 * dur / (t2 * t2);
    o[3] = -twopi * dur * dphi;
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

