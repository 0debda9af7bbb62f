// Device-math check of the fp32 step, fast vs precise, against fp64 on the device:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize tools/dbg/step_precision.hip -o tools/dbg/step_precision
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../mrphy.py_amd/csrc/bloch_math.hpp"
using namespace mrphy;

template <typename CT>
__global__ void k_run(const float* B, int nT, float g, float e1, float e2, float e1m1, float* out, int relax)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    SpinConst<float, CT> k;
    k.g = g; k.e1 = e1; k.e2 = e2; k.e1m1 = e1m1; k.d1 = e1 - 1.f; k.d2 = e2 - 1.f; k.relax = relax != 0;
    float mx = 0.f, my = 0.f, mz = 1.f;
    const float sc = 1.f + 0.001f * s;                     // each thread: slightly different field
    for (int t = 0; t < nT; ++t)
        bloch_step<float, CT>(k, B[3 * t] * sc, B[3 * t + 1] * sc, B[3 * t + 2] * sc, mx, my, mz);
    out[3 * s] = mx; out[3 * s + 1] = my; out[3 * s + 2] = mz;
}

__global__ void k_ref(const float* B, int nT, float g, float e1, float e2, float e1m1, double* out, int relax)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    double mx = 0, my = 0, mz = 1;
    const float sc = 1.f + 0.001f * s;
    for (int t = 0; t < nT; ++t) {
        const double bx = (double)(B[3 * t] * sc) * g, by = (double)(B[3 * t + 1] * sc) * g, bz = (double)(B[3 * t + 2] * sc) * g;
        const double x = bx * bx + by * by + bz * bz, ph = sqrt(x);
        const double S = ph > 1e-8 ? sin(ph) / ph : 1 - x / 6, C = ph > 1e-4 ? (1 - cos(ph)) / x : 0.5 - x / 24;
        const double wx = by * mz - bz * my, wy = bz * mx - bx * mz, wz = bx * my - by * mx;
        const double vx = by * wz - bz * wy, vy = bz * wx - bx * wz, vz = bx * wy - by * wx;
        mx += -S * wx + C * vx; my += -S * wy + C * vy; mz += -S * wz + C * vz;
        if (relax) { mx *= e2; my *= e2; mz = mz * e1 - e1m1; }
    }
    out[3 * s] = mx; out[3 * s + 1] = my; out[3 * s + 2] = mz;
}

int main()
{
    const int nT = 4096, nS = 256;
    std::vector<float> B(3 * nT);
    for (int t = 0; t < nT; ++t) {
        B[3 * t] = 0.2f * cosf(6.2831853f * t / nT); B[3 * t + 1] = 0.2f * sinf(6.2831853f * t / nT);
        B[3 * t + 2] = 14.f * atanf(t - nT / 2.f) * 0.6366f + 3.f;
    }
    float *dB, *o; double* od;
    hipMalloc(&dB, B.size() * 4); hipMalloc(&o, nS * 12); hipMalloc(&od, nS * 24);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const float g = 6.2831853f * 4257.6f * 4e-6f, e1 = expf(-4e-6f / 1.0f), e2 = expf(-4e-6f / 0.06f), e1m1 = e1 - 1.f;
    std::vector<float> h(nS * 3); std::vector<double> r(nS * 3);
    for (int relax = 0; relax < 2; ++relax) {
        hipLaunchKernelGGL(k_ref, dim3(nS / 64), dim3(64), 0, 0, dB, nT, g, e1, e2, e1m1, od, relax);
        hipMemcpy(r.data(), od, nS * 24, hipMemcpyDeviceToHost);
        for (int mode = 0; mode < 2; ++mode) {
            if (mode == 0) hipLaunchKernelGGL((k_run<float>), dim3(nS / 64), dim3(64), 0, 0, dB, nT, g, e1, e2, e1m1, o, relax);
            else           hipLaunchKernelGGL((k_run<prec_f32>), dim3(nS / 64), dim3(64), 0, 0, dB, nT, g, e1, e2, e1m1, o, relax);
            hipMemcpy(h.data(), o, nS * 12, hipMemcpyDeviceToHost);
            double num = 0, den = 0, mc[3] = {0, 0, 0};
            for (int i = 0; i < nS * 3; ++i) {
                num += (h[i] - r[i]) * (h[i] - r[i]); den += r[i] * r[i];
                if (fabs(h[i] - r[i]) > mc[i % 3]) mc[i % 3] = fabs(h[i] - r[i]);
            }
            printf("relax %d  %-8s rel-L2 vs fp64 %.3e  max abs err x %.2e y %.2e z %.2e   (spin 0: %.9g %.9g %.9g | ref %.12g %.12g %.12g)\n",
                   relax, mode ? "precise" : "fast", sqrt(num / den), mc[0], mc[1], mc[2], h[0], h[1], h[2], r[0], r[1], r[2]);
        }
    }
    return 0;
}
