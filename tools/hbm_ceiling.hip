// Streaming read / write ceilings of the device with minimal kernels (no arithmetic to speak of):
//   read : every lane sums 16-B non-temporal loads, 8 in flight, blocks walk contiguous 32-KB spans
//   write: every lane stores 16-B vectors
// with plain block order and with the XCD-contiguous order used by the product kernels.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o tools/hbm_ceiling && tools/hbm_ceiling [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TPB = 256, UNROLL = 8;   // one block: 256 lanes x 8 x 16 B = 32 KB

__device__ inline size_t span_of(unsigned per_xcd)
{
    return per_xcd ? (size_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : (size_t)blockIdx.x;
}

template <bool NT>
__global__ __launch_bounds__(TPB) void k_read(const f32x4* __restrict__ p, float* out, size_t nspan, unsigned per_xcd)
{
    const size_t sp = span_of(per_xcd);
    if (sp >= nspan) return;
    const f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    f32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = NT ? __builtin_nontemporal_load(q + i * TPB) : q[i * TPB];
    f32x4 s = v[0];
#pragma unroll
    for (int i = 1; i < UNROLL; ++i) s += v[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;   // never true for zero-filled input
}

template <bool NT>
__global__ __launch_bounds__(TPB) void k_write(f32x4* __restrict__ p, size_t nspan, unsigned per_xcd)
{
    const size_t sp = span_of(per_xcd);
    if (sp >= nspan) return;
    f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if (NT) __builtin_nontemporal_store(v, q + i * TPB); else q[i * TPB] = v;
    }
}

// NR read streams + 1 write stream of the same size (NR = 1: copy, NR = 2: add)
template <int NR>
__global__ __launch_bounds__(TPB) void k_mix(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                             f32x4* __restrict__ o, size_t nspan, unsigned per_xcd)
{
    const size_t sp = span_of(per_xcd);
    if (sp >= nspan) return;
    const size_t off = sp * (TPB * UNROLL) + threadIdx.x;
    f32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = __builtin_nontemporal_load(a + off + i * TPB);
    if (NR == 2) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) v[i] += __builtin_nontemporal_load(b + off + i * TPB);
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) __builtin_nontemporal_store(v[i], o + off + i * TPB);
}

template <typename F>
static double time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 32;
    const size_t bytes = gib << 30, nspan = bytes / (TPB * UNROLL * 16);
    f32x4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 0, bytes));
    for (int xcd = 0; xcd < 2; ++xcd) {
        const unsigned per = xcd ? (unsigned)((nspan + 7) / 8) : 0u;
        const unsigned grid = xcd ? per * 8 : (unsigned)nspan;
        double t;
        t = time_ms([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(TPB), 0, 0, buf, out, nspan, per); });
        printf("%zu GiB %-14s read  nt    %7.3f ms  %6.3f TB/s\n", gib, xcd ? "xcd-contiguous" : "plain order", t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(TPB), 0, 0, buf, out, nspan, per); });
        printf("%zu GiB %-14s read  plain %7.3f ms  %6.3f TB/s\n", gib, xcd ? "xcd-contiguous" : "plain order", t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(TPB), 0, 0, buf, nspan, per); });
        printf("%zu GiB %-14s write nt    %7.3f ms  %6.3f TB/s\n", gib, xcd ? "xcd-contiguous" : "plain order", t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(TPB), 0, 0, buf, nspan, per); });
        printf("%zu GiB %-14s write plain %7.3f ms  %6.3f TB/s\n", gib, xcd ? "xcd-contiguous" : "plain order", t, bytes / t / 1e9);
    }
    // mixed traffic: three buffers of gib/3 each
    const size_t b3 = (bytes / 3) & ~(size_t)((TPB * UNROLL * 16) - 1), ns3 = b3 / (TPB * UNROLL * 16);
    f32x4 *pa = buf, *pb = buf + b3 / 16, *po = buf + 2 * (b3 / 16);
    for (int xcd = 0; xcd < 2; ++xcd) {
        const unsigned per = xcd ? (unsigned)((ns3 + 7) / 8) : 0u;
        const unsigned grid = xcd ? per * 8 : (unsigned)ns3;
        double t;
        t = time_ms([&] { hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(TPB), 0, 0, pa, pb, po, ns3, per); });
        printf("%zu GiB %-14s 1R:1W copy  %7.3f ms  %6.3f TB/s\n", gib, xcd ? "xcd-contiguous" : "plain order", t, 2.0 * b3 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_mix<2>, dim3(grid), dim3(TPB), 0, 0, pa, pb, po, ns3, per); });
        printf("%zu GiB %-14s 2R:1W add   %7.3f ms  %6.3f TB/s\n", gib, xcd ? "xcd-contiguous" : "plain order", t, 3.0 * b3 / t / 1e9);
    }
    return 0;
}
