// Issue rate of v_fmac_f32 with a DPP row_newbcast source against plain VGPR / SGPR sources (MI355X):
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/dpp_rate.hip -o tools/dbg/dpp_rate && tools/dbg/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
#define R64(X) R8(R8(X))
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* cyc)
{
    float a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = in[threadIdx.x + 256], g = in[threadIdx.x + 512];
    float sb = in[blockIdx.x & 1];   // wave-uniform
    sb = __builtin_amdgcn_readfirstlane(sb);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(R8(
                "v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(g));
        } else if (MODE == 1) {
            asm volatile(R8(
                "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(g));
        } else if (MODE == 2) {
            asm volatile(R8(
                "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb), "v"(g));
        } else if (MODE == 3) {   // v_mov_dpp + 2 fmacs (what the compiler makes of update_dpp)
            float t;
            asm volatile(R8(
                "v_mov_b32_dpp %10, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32 %0, %10, %9\n v_fmac_f32 %1, %10, %9\n v_fmac_f32 %2, %10, %9\n v_fmac_f32 %3, %10, %9\n"
                "v_fmac_f32 %4, %10, %9\n v_fmac_f32 %5, %10, %9\n v_fmac_f32 %6, %10, %9\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(g), "v"(t));
        } else {   // v_readlane to SGPR, then fmac with the SGPR
            asm volatile(R8(
                "v_readlane_b32 s40, %8, 3\n"
                "v_fmac_f32 %0, s40, %9\n v_fmac_f32 %1, s40, %9\n v_fmac_f32 %2, s40, %9\n v_fmac_f32 %3, s40, %9\n"
                "v_fmac_f32 %4, s40, %9\n v_fmac_f32 %5, s40, %9\n v_fmac_f32 %6, s40, %9\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(g) : "s40");
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float *out, *in; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 4096);
    const int iters = 4096;
    for (int waves = 1; waves <= 2; ++waves)
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256 * 2 * waves;                        // 2 x waves blocks of 4 waves per CU
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc); break;
            default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double insts = (double)iters * 64;                    // VALU instructions per wave
        const char* nm[] = {"fmac_dpp row_newbcast", "fmac vgpr", "fmac sgpr", "mov_dpp + 7 fmac", "readlane + 7 fmac"};
        // waves per SIMD = blocks * 4 / (256 CUs * 4 SIMDs)
        const double wps = blocks * 4.0 / 1024.0;
        printf("%-24s waves/SIMD %.0f: %.3f ms  -> %.2f ns per wave-instruction per SIMD (s_memtime delta %lld)\n", nm[mode], wps, ms,
               ms * 1e6 / (insts * wps), c);
    }
    return 0;
}
