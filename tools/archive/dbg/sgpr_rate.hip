// v_fmac_f32 with SGPR sources: distinct SGPRs per instruction, and the scalar-load rate that feeds them.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/sgpr_rate.hip -o tools/dbg/sgpr_rate && tools/dbg/sgpr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
// 64 fmacs on 16 accumulators from 64 distinct SGPRs s[36:99]
#define F(a, s) "v_fmac_f32 %" #a ", s" #s ", %16\n"
#define F16(b) F(0, b##0) F(1, b##1) F(2, b##2) F(3, b##3) F(4, b##4) F(5, b##5) F(6, b##6) F(7, b##7) F(8, b##8) F(9, b##9)
#define FMAS \
    F(0,36) F(1,37) F(2,38) F(3,39) F(4,40) F(5,41) F(6,42) F(7,43) F(8,44) F(9,45) F(10,46) F(11,47) F(12,48) F(13,49) F(14,50) F(15,51) \
    F(0,52) F(1,53) F(2,54) F(3,55) F(4,56) F(5,57) F(6,58) F(7,59) F(8,60) F(9,61) F(10,62) F(11,63) F(12,64) F(13,65) F(14,66) F(15,67) \
    F(0,68) F(1,69) F(2,70) F(3,71) F(4,72) F(5,73) F(6,74) F(7,75) F(8,76) F(9,77) F(10,78) F(11,79) F(12,80) F(13,81) F(14,82) F(15,83) \
    F(0,84) F(1,85) F(2,86) F(3,87) F(4,88) F(5,89) F(6,90) F(7,91) F(8,92) F(9,93) F(10,94) F(11,95) F(12,96) F(13,97) F(14,98) F(15,99)
#define CLOB "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99"
#define ACCS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
// MODE 0: FMAs only (SGPRs loaded once).  1: + 4 x s_load_dwordx16 per 64 FMAs, one fixed 256-B row
// (scalar cache hits).  2: rows walking through a big buffer (stride 272 B, each wave its own region).
// 3: as 2 but 128 FMAs per 256 B loaded (the 32-coil adjoint's ratio)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, const float* big, int iters, long long stride_rows)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x] + i;
    float g = in[threadIdx.x + 512];
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) / 64);
    const float* p = big + (long long)wave * stride_rows * 68;
    asm volatile("s_load_dwordx16 s[36:51], %0, 0x0\n s_load_dwordx16 s[52:67], %0, 0x40\n"
                 "s_load_dwordx16 s[68:83], %0, 0x80\n s_load_dwordx16 s[84:99], %0, 0xc0\n s_waitcnt lgkmcnt(0)\n"
                 :: "s"(p) : CLOB, "memory");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(FMAS : ACCS : "v"(g) : CLOB);
        } else {
            const float* q = (MODE == 1) ? p : p + (long long)(i % (int)stride_rows) * 68;
            asm volatile("s_load_dwordx16 s[36:51], %17, 0x0\n s_load_dwordx16 s[52:67], %17, 0x40\n"
                         "s_load_dwordx16 s[68:83], %17, 0x80\n s_load_dwordx16 s[84:99], %17, 0xc0\n s_waitcnt lgkmcnt(0)\n"
                         FMAS : ACCS : "v"(g), "s"(q) : CLOB, "memory");
            if (MODE == 3) asm volatile(FMAS : ACCS : "v"(g) : CLOB);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out, *in, *big;
    const long long stride_rows = 1024;                      // rows per wave region
    const int blocks = 256 * 4;                              // 4 waves per SIMD
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096);
    const size_t bigbytes = (size_t)blocks * 4 * stride_rows * 68 * 4 + 4096;
    hipMalloc(&big, bigbytes); hipMemset(in, 0, 4096); hipMemset(big, 0, bigbytes);
    const int iters = 4096;
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, in, big, iters, stride_rows); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, in, big, iters, stride_rows); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, in, big, iters, stride_rows); break;
            default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, in, big, iters, stride_rows); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fm = (mode == 3 ? 128.0 : 64.0) * iters;          // FMAs per wave
        const char* nm[] = {"64 fmac, 64 distinct SGPRs, no loads", "4 x s_load_dwordx16 (same row) + 64 fmac",
                            "4 x s_load_dwordx16 (walking rows) + 64 fmac", "4 x s_load_dwordx16 (walking rows) + 128 fmac"};
        printf("%-48s %.3f ms -> %.2f ns per fmac per SIMD (4 waves/SIMD); scalar bytes per CU: %.2f B/ns\n", nm[mode], ms,
               ms * 1e6 / (fm * 4), mode ? 16.0 * iters * 256.0 / (ms * 1e6) : 0.0);
    }
    return 0;
}
