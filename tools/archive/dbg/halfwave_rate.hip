// Does a wave64 VALU instruction with the upper 32 lanes of EXEC off cost less issue time?  And how does the issue
// rate of DEPENDENT fmac chains depend on the waves per SIMD?  (K2 at one-round grids: 4 waves per SIMD.)
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/halfwave_rate.hip -o tools/dbg/halfwave_rate && tools/dbg/halfwave_rate
#include <hip/hip_runtime.h>
#include <cstdio>
// CH independent accumulator chains, 64 fmacs per loop trip in total
template <int CH, bool HALF>
__global__ __launch_bounds__(64) void k(float* out, const float* in, int iters)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x] + i;
    const float g = in[threadIdx.x + 64];
    if (!HALF || threadIdx.x < 32) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 64; ++j)
                asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a[j % CH]) : "v"(g));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH, bool HALF>
static void run(float* out, float* in, int wps)
{
    const int iters = 2048, blocks = 256 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH, HALF>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("chains %d  %s  waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD, %.2f ns per instruction of one wave\n",
           CH, HALF ? "lanes 0-31 only" : "all 64 lanes  ", wps, ms, ms * 1e6 / (64.0 * iters * wps), ms * 1e6 / (64.0 * iters));
}
int main()
{
    float *out, *in;
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    for (int wps : {1, 2, 4, 6, 8}) {
        run<1, false>(out, in, wps); run<2, false>(out, in, wps); run<4, false>(out, in, wps); run<8, false>(out, in, wps);
        run<1, true>(out, in, wps); run<8, true>(out, in, wps);
    }
    return 0;
}
