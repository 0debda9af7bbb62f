// v_pk_fma_f32 issue rate with an SGPR-pair source (distinct pair per instruction) against all-VGPR sources.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/pk_rate.hip -o tools/dbg/pk_rate && tools/dbg/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define P(a, s) "v_pk_fma_f32 %" #a ", s[" #s "], %8, %" #a " op_sel_hi:[1,0,1]\n"
#define V(a) "v_pk_fma_f32 %" #a ", %9, %8, %" #a " op_sel_hi:[1,0,1]\n"
#define PKS P(0,36:37) P(1,38:39) P(2,40:41) P(3,42:43) P(4,44:45) P(5,46:47) P(6,48:49) P(7,50:51) \
            P(0,52:53) P(1,54:55) P(2,56:57) P(3,58:59) P(4,60:61) P(5,62:63) P(6,64:65) P(7,66:67)
#define PKV V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7)
#define CLOB "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters)
{
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{in[threadIdx.x] + i, in[threadIdx.x] - i};
    f2 g = {in[threadIdx.x + 512], in[threadIdx.x + 513]};
    f2 b = {in[threadIdx.x + 256], in[threadIdx.x + 257]};
    asm volatile("s_load_dwordx16 s[36:51], %0, 0x0\n s_load_dwordx16 s[52:67], %0, 0x40\n s_waitcnt lgkmcnt(0)\n" :: "s"(in) : CLOB, "memory");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile(PKS PKS PKS PKS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(g), "v"(b) : CLOB);
        else
            asm volatile(PKV PKV PKV PKV : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(g), "v"(b) : CLOB);
    }
    f2 s = {0, 0};
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
int main()
{
    float *out, *in;
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 8192); hipMemset(in, 0, 8192);
    const int iters = 4096, blocks = 256 * 4;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
            else           hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = 64.0 * iters;
        printf("%-40s %.3f ms -> %.2f ns per v_pk_fma_f32 per SIMD (= 2 FMAs), 4 waves/SIMD\n",
               mode == 0 ? "pk_fma, SGPR-pair source (distinct)" : "pk_fma, VGPR sources", ms, ms * 1e6 / (insts * 4));
    }
    return 0;
}
