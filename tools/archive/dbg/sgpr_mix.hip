// Does an SGPR-source VALU instruction cost more than a VGPR-source one inside a stream of VGPR-source
// instructions?  Groups of 8 fmacs of which K take (distinct) SGPR sources, K = 0, 1, 2, 4, 8.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/sgpr_mix.hip -o tools/dbg/sgpr_mix && tools/dbg/sgpr_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define V(a) "v_fmac_f32 %" #a ", %8, %9\n"
#define S(a, s) "v_fmac_f32 %" #a ", s" #s ", %9\n"
#define G0 V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7)
#define G1(b) S(0, b) V(1) V(2) V(3) V(4) V(5) V(6) V(7)
#define G2(b, c) S(0, b) V(1) V(2) V(3) S(4, c) V(5) V(6) V(7)
#define G4(b, c, d, e) S(0, b) V(1) S(2, c) V(3) S(4, d) V(5) S(6, e) V(7)
#define G8(b0,b1,b2,b3,b4,b5,b6,b7) S(0,b0) S(1,b1) S(2,b2) S(3,b3) S(4,b4) S(5,b5) S(6,b6) S(7,b7)
#define CLOB "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51"
#define OPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(g) : CLOB
template <int K>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x] + i;
    float g = in[threadIdx.x + 512], b = in[threadIdx.x + 256];
    asm volatile("s_load_dwordx16 s[36:51], %0, 0x0\n s_waitcnt lgkmcnt(0)\n" :: "s"(in) : CLOB, "memory");
    for (int i = 0; i < iters; ++i) {
        if (K == 0) asm volatile(G0 G0 G0 G0 G0 G0 G0 G0 : OPS);
        if (K == 1) asm volatile(G1(36) G1(37) G1(38) G1(39) G1(40) G1(41) G1(42) G1(43) : OPS);
        if (K == 2) asm volatile(G2(36, 37) G2(38, 39) G2(40, 41) G2(42, 43) G2(44, 45) G2(46, 47) G2(48, 49) G2(50, 51) : OPS);
        if (K == 4) asm volatile(G4(36, 37, 38, 39) G4(40, 41, 42, 43) G4(44, 45, 46, 47) G4(48, 49, 50, 51)
                                 G4(36, 37, 38, 39) G4(40, 41, 42, 43) G4(44, 45, 46, 47) G4(48, 49, 50, 51) : OPS);
        if (K == 8) asm volatile(G8(36,37,38,39,40,41,42,43) G8(44,45,46,47,48,49,50,51) G8(36,37,38,39,40,41,42,43) G8(44,45,46,47,48,49,50,51)
                                 G8(36,37,38,39,40,41,42,43) G8(44,45,46,47,48,49,50,51) G8(36,37,38,39,40,41,42,43) G8(44,45,46,47,48,49,50,51) : OPS);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&in, 8192); (void)hipMemset(in, 0, 8192);
    const int iters = 4096;
    for (int waves = 1; waves <= 4; waves *= 2)
    for (int K : {0, 1, 2, 4, 8}) {
        const int blocks = 256 * waves;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            switch (K) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, in, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, in, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, in, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, in, iters); break;
            default: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, in, iters); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d, %d of 8 fmacs with an SGPR source: %.3f ms -> %.2f ns per fmac per SIMD\n", waves, K, ms,
               ms * 1e6 / (64.0 * iters * waves));
    }
    return 0;
}
