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
            const float Bz = fmaf(gz, lz, fmaf(gy, ly, gx * lx)) + dz;
            step(m, (double)Bx, (double)By, (double)Bz, g[r], E1 != NULL, E1 ? E1[r] : 1.0,
                 E2 ? E2[r] : 1.0, E1m1 ? E1m1[r] : 0.0);
        }
        Mo[r * 3] = m[0]; Mo[r * 3 + 1] = m[1]; Mo[r * 3 + 2] = m[2];
    }
}
