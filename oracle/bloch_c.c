/* bloch_c.c -- plain-C, double-precision restatement of the hot path.  TEST INFRASTRUCTURE, NOT
 * PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 *
 * Written from the reference's formulas (tianrluo/MRphy.py v0.2.0), in the reference's own form
 * -- axis/angle Rodrigues rotation with sin and cos -- so that it is independent of both torch
 * (oracle/bloch_oracle.py restates the same path with ATen ops) and of the HIP kernels (which
 * use the un-normalised S(x), C(x) form):
 *
 *   field assembly     beffective.py:137-165   Bz = loc . gr + df/gamma,  Bxy = sum_c b1_c * rf_c
 *   one step           sims.py:100-124 / slowsims.py:42-51
 *       phi = -|B| g (g = gamma 2 pi dt), u = B/max(|B|, 1e-12),
 *       M <- cos(phi) M + (1 - cos(phi)) (u.M) u + sin(phi) u x M            (utils.py:333-359)
 *       M <- (E2 Mx, E2 My, E1 Mz - (E1 - 1))
 *
 * Per-spin constants g, E1, E2, E1m1 are passed in (rows of N*nM doubles), as the kernels take
 * them.  OpenMP over spins.  Build: make -C oracle  ->  oracle/libbloch_c.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline void step(double m[3], double Bx, double By, double Bz, double g, int relax,
                        double e1, double e2, double e1m1)
{
    const double nrm = sqrt(Bx * Bx + By * By + Bz * Bz);
    const double d = nrm > 1e-12 ? nrm : 1e-12;
    const double ux = Bx / d, uy = By / d, uz = Bz / d;
    const double phi = -nrm * g;
    const double c = cos(phi), s = sin(phi);
    const double um = ux * m[0] + uy * m[1] + uz * m[2];
    const double cx = uy * m[2] - uz * m[1], cy = uz * m[0] - ux * m[2], cz = ux * m[1] - uy * m[0];
    double x = c * m[0] + (1 - c) * um * ux + s * cx;
    double y = c * m[1] + (1 - c) * um * uy + s * cy;
    double z = c * m[2] + (1 - c) * um * uz + s * cz;
    if (relax) { x *= e2; y *= e2; z = z * e1 - e1m1; }
    m[0] = x; m[1] = y; m[2] = z;
}

/* rf (N|1, 2, nT, nC), gr (N|1, 3, nT): batch strides rf_sn, gr_sn (0 = broadcast);
 * loc (N, nM, 3); dfg (N*nM) = df/gamma per spin or NULL; b1 (N, nM, 2, nC) or NULL (=> nC == 1,
 * Bxy = rf).  beff (N, nM, nT, 3). */
void oracle_rfgr2beff_f64(const double* rf, int64_t rf_sn, const double* gr, int64_t gr_sn,
                          const double* loc, const double* dfg, const double* b1, double* beff,
                          int64_t N, int64_t nM, int64_t nT, int64_t nC)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N * nM; ++r) {
        const int64_t n = r / nM;
        const double* rfr = rf + n * rf_sn;
        const double* rfi = rfr + nT * nC;
        const double* g = gr + n * gr_sn;
        const double lx = loc[r * 3], ly = loc[r * 3 + 1], lz = loc[r * 3 + 2];
        const double dz = dfg ? dfg[r] : 0.0;
        for (int64_t t = 0; t < nT; ++t) {
            double Bx = 0, By = 0;
            for (int64_t c = 0; c < nC; ++c) {
                const double br = b1 ? b1[(r * 2) * nC + c] : 1.0;
                const double bi = b1 ? b1[(r * 2 + 1) * nC + c] : 0.0;
                Bx += br * rfr[t * nC + c] - bi * rfi[t * nC + c];
                By += br * rfi[t * nC + c] + bi * rfr[t * nC + c];
            }
            double* o = beff + (r * nT + t) * 3;
            o[0] = Bx; o[1] = By;
            o[2] = lx * g[t] + ly * g[nT + t] + lz * g[2 * nT + t] + dz;
        }
    }
}

/* Mi, Mo (rows, 3); beff (rows, nT, 3); g, E1, E2, E1m1 (rows) -- E1 == NULL: no relaxation. */
void oracle_blochsim_f64(const double* Mi, const double* beff, const double* g, const double* E1,
                         const double* E2, const double* E1m1, double* Mo, int64_t rows, int64_t nT)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        double m[3] = {Mi[r * 3], Mi[r * 3 + 1], Mi[r * 3 + 2]};
        const double* b = beff + r * nT * 3;
        for (int64_t t = 0; t < nT; ++t)
            step(m, b[t * 3], b[t * 3 + 1], b[t * 3 + 2], g[r], E1 != NULL, E1 ? E1[r] : 1.0,
                 E2 ? E2[r] : 1.0, E1m1 ? E1m1[r] : 0.0);
        Mo[r * 3] = m[0]; Mo[r * 3 + 1] = m[1]; Mo[r * 3 + 2] = m[2];
    }
}

/* The two together without the (rows, nT, 3) tensor: whole configs fit in memory on the host. */
void oracle_blochsim_rfgr_f64(const double* Mi, const double* rf, int64_t rf_sn, const double* gr,
                              int64_t gr_sn, const double* loc, const double* dfg, const double* b1,
                              const double* g, const double* E1, const double* E2,
                              const double* E1m1, double* Mo, int64_t N, int64_t nM, int64_t nT,
                              int64_t nC)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N * nM; ++r) {
        const int64_t n = r / nM;
        const double* rfr = rf + n * rf_sn;
        const double* rfi = rfr + nT * nC;
        const double* gg = gr + n * gr_sn;
        const double lx = loc[r * 3], ly = loc[r * 3 + 1], lz = loc[r * 3 + 2];
        const double dz = dfg ? dfg[r] : 0.0;
        double m[3] = {Mi[r * 3], Mi[r * 3 + 1], Mi[r * 3 + 2]};
        for (int64_t t = 0; t < nT; ++t) {
            double Bx = 0, By = 0;
            for (int64_t c = 0; c < nC; ++c) {
                const double br = b1 ? b1[(r * 2) * nC + c] : 1.0;
                const double bi = b1 ? b1[(r * 2 + 1) * nC + c] : 0.0;
                Bx += br * rfr[t * nC + c] - bi * rfi[t * nC + c];
                By += br * rfi[t * nC + c] + bi * rfr[t * nC + c];
            }
            const double Bz = lx * gg[t] + ly * gg[nT + t] + lz * gg[2 * nT + t] + dz;
            step(m, Bx, By, Bz, g[r], E1 != NULL, E1 ? E1[r] : 1.0, E2 ? E2[r] : 1.0,
                 E1m1 ? E1m1[r] : 0.0);
        }
        Mo[r * 3] = m[0]; Mo[r * 3 + 1] = m[1]; Mo[r * 3 + 2] = m[2];
    }
}

/* The field of one step in SINGLE precision, the way the reference forms its fp32 Beff tensor
 * (beffective.py:137-167: loc @ gr accumulated with fused multiply-adds as the BLAS kernels do -- K = 3 --,
 * + df/gamma; Bxy = rf, or the complex b1 product for one coil, or FMA chains for several).  ONE function for the
 * forward, the gradient and the exported field (oracle_field_f32), and that export is pinned bit for bit to the
 * reference's own Beff rows (tests/golden/big_beff_rows_f32.npz, round 4): the "same fp32 field" yardstick of the
 * all-spins tests is then the reference's field, not merely the kernels'. */
static inline void field_f32_at(const double* rfr, const double* rfi, const double* gg, const double* b1,
                                int64_t r, int64_t t, int64_t nT, int64_t nC, float lx, float ly, float lz,
                                float dz, float* Bx_, float* By_, float* Bz_)
{
    float Bx = 0, By = 0;
    if (!b1) { Bx = (float)rfr[t * nC]; By = (float)rfi[t * nC]; }
    else if (nC == 1) {
        const float br = (float)b1[r * 2], bi = (float)b1[r * 2 + 1];
        const float rr = (float)rfr[t], ri = (float)rfi[t];
        Bx = 0.0f + fmaf(br, rr, -(bi * ri));
        By = 0.0f + fmaf(br, ri, bi * rr);
    } else {
        for (int64_t c = 0; c < nC; ++c) {
            const float br = (float)b1[(r * 2) * nC + c], bi = (float)b1[(r * 2 + 1) * nC + c];
            const float rr = (float)rfr[t * nC + c], ri = (float)rfi[t * nC + c];
            Bx = fmaf(br, rr, fmaf(-bi, ri, Bx));
            By = fmaf(br, ri, fmaf(bi, rr, By));
        }
    }
    const float gx = (float)gg[t], gy = (float)gg[nT + t], gz = (float)gg[2 * nT + t];
    *Bx_ = Bx; *By_ = By;
    *Bz_ = fmaf(gz, lz, fmaf(gy, ly, gx * lx)) + dz;
}

/* the single-precision field itself, (N, nM, nT, 3) floats */
void oracle_field_f32(const double* rf, int64_t rf_sn, const double* gr, int64_t gr_sn, const double* loc,
                      const double* dfg, const double* b1, float* beff, int64_t N, int64_t nM, int64_t nT,
                      int64_t nC)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N * nM; ++r) {
        const int64_t n = r / nM;
        const double* rfr = rf + n * rf_sn;
        const double* rfi = rfr + nT * nC;
        const double* gg = gr + n * gr_sn;
        const float lx = (float)loc[r * 3], ly = (float)loc[r * 3 + 1], lz = (float)loc[r * 3 + 2];
        const float dz = dfg ? (float)dfg[r] : 0.0f;
        for (int64_t t = 0; t < nT; ++t)
            field_f32_at(rfr, rfi, gg, b1, r, t, nT, nC, lx, ly, lz, dz, beff + (r * nT + t) * 3,
                         beff + (r * nT + t) * 3 + 1, beff + (r * nT + t) * 3 + 2);
    }
}

/* As oracle_blochsim_rfgr_f64, but the field of every step is first formed in SINGLE precision the
 * way the reference forms its fp32 Beff tensor (beffective.py:137-167: loc @ gr accumulated with
 * fused multiply-adds as the BLAS kernels do -- K = 3 --, + df/gamma; Bxy = rf, or the complex b1
 * product for one coil, or FMA chains for several), then integrated in double.  All inputs must hold
 * float values; dfg = (float)df / (float)gamma, formed by the caller in float.  This is "exact
 * arithmetic on the same fp32 field": what an fp32 run of blochsim can be asked to reproduce. */
void oracle_blochsim_rfgr_f32field(const double* Mi, const double* rf, int64_t rf_sn, const double* gr,
                                   int64_t gr_sn, const double* loc, const double* dfg,
                                   const double* b1, const double* g, const double* E1,
                                   const double* E2, const double* E1m1, double* Mo, int64_t N,
                                   int64_t nM, int64_t nT, int64_t nC)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N * nM; ++r) {
        const int64_t n = r / nM;
        const double* rfr = rf + n * rf_sn;
        const double* rfi = rfr + nT * nC;
        const double* gg = gr + n * gr_sn;
        const float lx = (float)loc[r * 3], ly = (float)loc[r * 3 + 1], lz = (float)loc[r * 3 + 2];
        const float dz = dfg ? (float)dfg[r] : 0.0f;
        double m[3] = {Mi[r * 3], Mi[r * 3 + 1], Mi[r * 3 + 2]};
        for (int64_t t = 0; t < nT; ++t) {
            float Bx, By, Bz;
            field_f32_at(rfr, rfi, gg, b1, r, t, nT, nC, lx, ly, lz, dz, &Bx, &By, &Bz);
            step(m, (double)Bx, (double)By, (double)Bz, g[r], E1 != NULL, E1 ? E1[r] : 1.0,
                 E2 ? E2[r] : 1.0, E1m1 ? E1m1[r] : 0.0);
        }
        Mo[r * 3] = m[0]; Mo[r * 3 + 1] = m[1]; Mo[r * 3 + 2] = m[2];
    }
}

/* ---------------------------------------------------------------------------------------------
 * Adjoint (round 3): the explicit Jacobian of sims.py:135-269, restated in the reference's own
 * axis/angle form.  With b = g B, phi = max(|b|, 1e-12), u = b / phi, (s, c1) = (sin phi,
 * cos phi - 1), ht = E h (adjoint relaxation, sims.py:172), m0 the magnetisation BEFORE the step,
 * mt = R m0 its rotation (sims.py:109-121; the reference recovers it as E^-1 m1, sims.py:174-177):
 *     h0      = ht + c1 (ht - (u.ht) u) + s (u x ht)                             sims.py:218-227
 *     dL/dB_t = -g { s/phi (m0 x ht) + c1/phi ((u.m0) ht + (u.ht) m0)
 *                    - [ (mt - s/phi m0).(u x ht) + 2 c1/phi (u.ht)(u.m0) ] u }  sims.py:229-259
 * (the reference pre-multiplies h by -g, sims.py:194, and divides it out of grad_Mi, sims.py:267:
 * the sweep is linear in h, so scaling the field gradient instead is the same thing -- and is
 * right for per-spin g, where sims.py:267 is not).  Everything in double.
 * ------------------------------------------------------------------------------------------- */
#include <stdlib.h>

static inline void rot_fwd(const double m[3], const double u[3], double s, double c1, double o[3])
{
    const double um = u[0] * m[0] + u[1] * m[1] + u[2] * m[2];
    const double cx = u[1] * m[2] - u[2] * m[1], cy = u[2] * m[0] - u[0] * m[2],
                 cz = u[0] * m[1] - u[1] * m[0];
    o[0] = m[0] - s * cx + c1 * (m[0] - um * u[0]);
    o[1] = m[1] - s * cy + c1 * (m[1] - um * u[1]);
    o[2] = m[2] - s * cz + c1 * (m[2] - um * u[2]);
}

/* one adjoint step: h <- dL/dm0, gB <- dL/dB_t */
static inline void step_adj(const double m0[3], double Bx, double By, double Bz, double g, int relax,
                            double e1, double e2, double h[3], double gB[3])
{
    const double b[3] = {g * Bx, g * By, g * Bz};
    const double nrm = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    const double phi = nrm > 1e-12 ? nrm : 1e-12;
    const double u[3] = {b[0] / phi, b[1] / phi, b[2] / phi};
    const double s = sin(phi), c1 = cos(phi) - 1.0;
    double ht[3] = {h[0], h[1], h[2]};
    if (relax) { ht[0] *= e2; ht[1] *= e2; ht[2] *= e1; }
    double mt[3];
    rot_fwd(m0, u, s, c1, mt);
    const double um = u[0] * m0[0] + u[1] * m0[1] + u[2] * m0[2];
    const double uh = u[0] * ht[0] + u[1] * ht[1] + u[2] * ht[2];
    const double x[3] = {u[1] * ht[2] - u[2] * ht[1], u[2] * ht[0] - u[0] * ht[2],
                         u[0] * ht[1] - u[1] * ht[0]};                       /* u x ht */
    const double mh[3] = {m0[1] * ht[2] - m0[2] * ht[1], m0[2] * ht[0] - m0[0] * ht[2],
                          m0[0] * ht[1] - m0[1] * ht[0]};                    /* m0 x ht */
    const double sp = s / phi, cp = c1 / phi;
    const double k = (mt[0] - sp * m0[0]) * x[0] + (mt[1] - sp * m0[1]) * x[1] +
                     (mt[2] - sp * m0[2]) * x[2] + 2.0 * cp * uh * um;
    for (int i = 0; i < 3; ++i) {
        gB[i] = -g * (sp * mh[i] + cp * (um * ht[i] + uh * m0[i]) - k * u[i]);
        h[i] = ht[i] + c1 * (ht[i] - uh * u[i]) + s * x[i];
    }
}

/* blochsim forward + adjoint over a materialised field: gMi (rows, 3), gBeff (rows, nT, 3). */
void oracle_blochsim_bwd_f64(const double* Mi, const double* beff, const double* g, const double* E1,
                             const double* E2, const double* E1m1, const double* gMo, double* gMi,
                             double* gBeff, int64_t rows, int64_t nT)
{
#pragma omp parallel
    {
        double* hist = (double*)malloc(sizeof(double) * 3 * (size_t)(nT > 0 ? nT : 1));
#pragma omp for schedule(static)
        for (int64_t r = 0; r < rows; ++r) {
            double m[3] = {Mi[r * 3], Mi[r * 3 + 1], Mi[r * 3 + 2]};
            const double* b = beff + r * nT * 3;
            const int relax = E1 != NULL;
            const double e1 = E1 ? E1[r] : 1.0, e2 = E2 ? E2[r] : 1.0, o = E1m1 ? E1m1[r] : 0.0;
            for (int64_t t = 0; t < nT; ++t) {
                hist[3 * t] = m[0]; hist[3 * t + 1] = m[1]; hist[3 * t + 2] = m[2];
                step(m, b[t * 3], b[t * 3 + 1], b[t * 3 + 2], g[r], relax, e1, e2, o);
            }
            double h[3] = {gMo[r * 3], gMo[r * 3 + 1], gMo[r * 3 + 2]};
            for (int64_t t = nT - 1; t >= 0; --t)
                step_adj(hist + 3 * t, b[t * 3], b[t * 3 + 1], b[t * 3 + 2], g[r], relax, e1, e2, h,
                         gBeff + (r * nT + t) * 3);
            gMi[r * 3] = h[0]; gMi[r * 3 + 1] = h[1]; gMi[r * 3 + 2] = h[2];
        }
        free(hist);
    }
}

/* Forward + adjoint of blochsim(Mi, rfgr2beff(rf, gr, ...)) without the (rows, nT, 3) tensors:
 *   Mo (rows, 3), gMi (rows, 3), grf (N|1 -> always N entries, 2, nT, nC), ggr (N, 3, nT),
 * the pulse gradients summed over the spins of each batch entry (chain rule through
 * beffective.py:137-165: gr_k[t] += loc_k gBz, rf_c[t] += conj(b1_c) (gBx + i gBy)).
 * field_f32 != 0: the field of every step is the single-precision one of
 * oracle_blochsim_rfgr_f32field (what an fp32 Beff tensor holds), integrated and differentiated in
 * double -- "exact arithmetic on the same fp32 field".  Whole configs run in seconds. */
void oracle_blochsim_rfgr_grad(const double* Mi, const double* rf, int64_t rf_sn, const double* gr,
                               int64_t gr_sn, const double* loc, const double* dfg, const double* b1,
                               const double* g, const double* E1, const double* E2,
                               const double* E1m1, const double* gMo, double* Mo, double* gMi,
                               double* grf, double* ggr, int64_t N, int64_t nM, int64_t nT,
                               int64_t nC, int field_f32)
{
    const int64_t nrf = 2 * nT * nC, ngr = 3 * nT;
    for (int64_t i = 0; i < N * nrf; ++i) grf[i] = 0.0;
    for (int64_t i = 0; i < N * ngr; ++i) ggr[i] = 0.0;
#pragma omp parallel
    {
        double* hist = (double*)malloc(sizeof(double) * 6 * (size_t)(nT > 0 ? nT : 1));
        double* B = hist + 3 * (nT > 0 ? nT : 1);
        double* arf = (double*)calloc((size_t)(N * nrf > 0 ? N * nrf : 1), sizeof(double));
        double* agr = (double*)calloc((size_t)(N * ngr > 0 ? N * ngr : 1), sizeof(double));
#pragma omp for schedule(static)
        for (int64_t r = 0; r < N * nM; ++r) {
            const int64_t n = r / nM;
            const double* rfr = rf + n * rf_sn;
            const double* rfi = rfr + nT * nC;
            const double* gg = gr + n * gr_sn;
            const double lx = loc[r * 3], ly = loc[r * 3 + 1], lz = loc[r * 3 + 2];
            const double dz = dfg ? dfg[r] : 0.0;
            const int relax = E1 != NULL;
            const double e1 = E1 ? E1[r] : 1.0, e2 = E2 ? E2[r] : 1.0, o = E1m1 ? E1m1[r] : 0.0;
            double m[3] = {Mi[r * 3], Mi[r * 3 + 1], Mi[r * 3 + 2]};
            for (int64_t t = 0; t < nT; ++t) {
                double Bx = 0, By = 0, Bz;
                if (field_f32) {
                    float fx, fy, fz;
                    field_f32_at(rfr, rfi, gg, b1, r, t, nT, nC, (float)lx, (float)ly, (float)lz, (float)dz, &fx, &fy, &fz);
                    Bx = fx; By = fy; Bz = fz;
                } else {
                    for (int64_t c = 0; c < nC; ++c) {
                        const double br = b1 ? b1[(r * 2) * nC + c] : 1.0;
                        const double bi = b1 ? b1[(r * 2 + 1) * nC + c] : 0.0;
                        Bx += br * rfr[t * nC + c] - bi * rfi[t * nC + c];
                        By += br * rfi[t * nC + c] + bi * rfr[t * nC + c];
                    }
                    Bz = lx * gg[t] + ly * gg[nT + t] + lz * gg[2 * nT + t] + dz;
                }
                B[3 * t] = Bx; B[3 * t + 1] = By; B[3 * t + 2] = Bz;
                hist[3 * t] = m[0]; hist[3 * t + 1] = m[1]; hist[3 * t + 2] = m[2];
                step(m, Bx, By, Bz, g[r], relax, e1, e2, o);
            }
            if (Mo) { Mo[r * 3] = m[0]; Mo[r * 3 + 1] = m[1]; Mo[r * 3 + 2] = m[2]; }
            double h[3] = {gMo[r * 3], gMo[r * 3 + 1], gMo[r * 3 + 2]};
            double* ar = arf + n * nrf;
            double* ai = ar + nT * nC;
            double* ag = agr + n * ngr;
            for (int64_t t = nT - 1; t >= 0; --t) {
                double gB[3];
                step_adj(hist + 3 * t, B[3 * t], B[3 * t + 1], B[3 * t + 2], g[r], relax, e1, e2, h, gB);
                ag[t] += lx * gB[2]; ag[nT + t] += ly * gB[2]; ag[2 * nT + t] += lz * gB[2];
                for (int64_t c = 0; c < nC; ++c) {
                    const double br = b1 ? b1[(r * 2) * nC + c] : 1.0;
                    const double bi = b1 ? b1[(r * 2 + 1) * nC + c] : 0.0;
                    ar[t * nC + c] += br * gB[0] + bi * gB[1];
                    ai[t * nC + c] += br * gB[1] - bi * gB[0];
                }
            }
            if (gMi) { gMi[r * 3] = h[0]; gMi[r * 3 + 1] = h[1]; gMi[r * 3 + 2] = h[2]; }
        }
#pragma omp critical
        {
            for (int64_t i = 0; i < N * nrf; ++i) grf[i] += arf[i];
            for (int64_t i = 0; i < N * ngr; ++i) ggr[i] += agr[i];
        }
        free(hist); free(arf); free(agr);
    }
}
