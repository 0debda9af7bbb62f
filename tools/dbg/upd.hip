// op-by-op comparison of the precise update between the device and the host (IEEE fmaf)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
struct Tr { float a, s, es, p, ep, lo, q, eq, r; };
__host__ __device__ inline Tr upd(float m, float w, float v, float S, float C, float E, float off)
{
#pragma clang fp contract(off)
    Tr t;
    t.a = fmaf(-S, w, m);
    t.s = fmaf(C, v, t.a);
    t.es = fmaf(C, v, t.a - t.s);
    t.p = t.s * E;
    t.ep = fmaf(t.s, E, -t.p);
    t.lo = fmaf(t.es, E, t.ep);
    t.q = t.p - off;
    t.eq = (t.p - t.q) - off;
    t.r = t.q + (t.lo + t.eq);
    return t;
}
__global__ void k(const float* in, Tr* o, int n, float E, float off)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = upd(in[5 * i], in[5 * i + 1], in[5 * i + 2], in[5 * i + 3], in[5 * i + 4], E, off);
}
int main()
{
    const int n = 1 << 16;
    std::vector<float> in(5 * n); unsigned st = 12345;
    auto U = [&]() { st = st * 1664525u + 1013904223u; return 2.f * (st >> 8) / 16777216.f - 1.f; };
    for (int i = 0; i < n; ++i) {
        in[5 * i] = 0.9989f + 1e-4f * U(); in[5 * i + 1] = 0.02f * U(); in[5 * i + 2] = 0.05f * U();
        in[5 * i + 3] = 0.74f + 0.01f * U(); in[5 * i + 4] = 0.43f + 0.01f * U();
    }
    float* di; Tr* d; (void)hipMalloc(&di, in.size() * 4); (void)hipMalloc(&d, n * sizeof(Tr));
    (void)hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    const float E = expf(-4e-6f), off = E - 1.f;
    k<<<n / 256, 256>>>(di, d, n, E, off);
    std::vector<Tr> h(n); (void)hipMemcpy(h.data(), d, n * sizeof(Tr), hipMemcpyDeviceToHost);
    const char* names[9] = {"a", "s", "es", "p", "ep", "lo", "q", "eq", "r"};
    int diff[9] = {0}; int shown = 0;
    for (int i = 0; i < n; ++i) {
        Tr c = upd(in[5 * i], in[5 * i + 1], in[5 * i + 2], in[5 * i + 3], in[5 * i + 4], E, off);
        const float* x = (const float*)&c; const float* y = (const float*)&h[i];
        for (int j = 0; j < 9; ++j) if (memcmp(&x[j], &y[j], 4)) {
            ++diff[j];
            if (shown < 6) { printf("i %d %s: host %.9g device %.9g\n", i, names[j], x[j], y[j]); ++shown; }
        }
    }
    for (int j = 0; j < 9; ++j) printf("%s differs in %d of %d\n", names[j], diff[j], n);
    return 0;
}
