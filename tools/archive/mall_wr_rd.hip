// What a streaming WRITE leaves in the 256-MB memory-side cache (MALL / Infinity Cache) and what the READ that
// follows pays for it.  One buffer; a writer (16-B stores, XCD-contiguous block order as K0 writes Beff) and
// then a reader (16-B loads, 8 in flight) with each combination of the cache-policy bits of the gfx950
// global_load / global_store encodings (sc0, sc1, nt), in plain and XCD-contiguous block order; the first read
// after the write against the second.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_wr_rd.hip -o tools/mall_wr_rd && tools/mall_wr_rd [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TPB = 256, UNROLL = 8;

__device__ inline size_t span_of(unsigned per_xcd)
{
    return per_xcd ? (size_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : (size_t)blockIdx.x;
}

#define LOAD_ASM(BITS) asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v[i]) : "v"(q + i * TPB) : "memory")
#define STORE_ASM(BITS) asm volatile("global_store_dwordx4 %0, %1, off " BITS : : "v"(q + i * TPB), "v"(val) : "memory")

template <int POL>
__global__ __launch_bounds__(TPB) void k_read(const f32x4* __restrict__ p, float* out, size_t nspan, unsigned per_xcd)
{
    const size_t sp = span_of(per_xcd);
    if (sp >= nspan) return;
    const f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    f32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if (POL == 0) LOAD_ASM("");
        if (POL == 1) LOAD_ASM("nt");
        if (POL == 2) LOAD_ASM("sc0");
        if (POL == 3) LOAD_ASM("sc1");
        if (POL == 4) LOAD_ASM("sc0 sc1");
        if (POL == 5) LOAD_ASM("sc0 sc1 nt");
        if (POL == 6) LOAD_ASM("sc1 nt");
        if (POL == 7) LOAD_ASM("sc0 nt");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = v[0];
#pragma unroll
    for (int i = 1; i < UNROLL; ++i) s += v[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}

template <int POL>
__global__ __launch_bounds__(TPB) void k_write(f32x4* __restrict__ p, size_t nspan, unsigned per_xcd)
{
    const size_t sp = span_of(per_xcd);
    if (sp >= nspan) return;
    f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    const f32x4 val = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if (POL == 0) STORE_ASM("");
        if (POL == 1) STORE_ASM("nt");
        if (POL == 2) STORE_ASM("sc0");
        if (POL == 3) STORE_ASM("sc1");
        if (POL == 4) STORE_ASM("sc0 sc1");
        if (POL == 5) STORE_ASM("sc0 sc1 nt");
        if (POL == 6) STORE_ASM("sc1 nt");
        if (POL == 7) STORE_ASM("sc0 nt");
    }
}

static const char* POLN[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0 nt"};

static float ms_of(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

template <int WP, int RP>
static void combo(f32x4* buf, float* out, size_t bytes, size_t nspan, bool rd_xcd, bool wr_xcd = true)
{
    const unsigned wper = wr_xcd ? (unsigned)((nspan + 7) / 8) : 0u, wgrid = wr_xcd ? wper * 8 : (unsigned)nspan;
    const unsigned rper = rd_xcd ? (unsigned)((nspan + 7) / 8) : 0u, rgrid = rd_xcd ? rper * 8 : (unsigned)nspan;
    hipEvent_t e[4]; for (auto& x : e) CK(hipEventCreate(&x));
    double w = 1e30, r1 = 1e30, r2 = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(k_write<WP>, dim3(wgrid), dim3(TPB), 0, 0, buf, nspan, wper);
        CK(hipEventRecord(e[1]));
        hipLaunchKernelGGL(k_read<RP>, dim3(rgrid), dim3(TPB), 0, 0, buf, out, nspan, rper);
        CK(hipEventRecord(e[2]));
        hipLaunchKernelGGL(k_read<RP>, dim3(rgrid), dim3(TPB), 0, 0, buf, out, nspan, rper);
        CK(hipEventRecord(e[3]));
        CK(hipDeviceSynchronize());
        if (rep) {
            w = fmin(w, ms_of(e[0], e[1])); r1 = fmin(r1, ms_of(e[1], e[2])); r2 = fmin(r2, ms_of(e[2], e[3]));
        }
    }
    printf("write %-10s (%s) %6.3f ms %5.2f TB/s | read %-10s (%s) first %6.3f ms %5.2f TB/s  second %6.3f ms %5.2f TB/s  penalty %+6.3f ms\n",
           POLN[WP], wr_xcd ? "xcd" : "lin", w, bytes / w / 1e9, POLN[RP], rd_xcd ? "xcd" : "lin", r1, bytes / r1 / 1e9, r2,
           bytes / r2 / 1e9, r1 - r2);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 3.0;
    const size_t bytes = ((size_t)(gib * (1 << 30))) & ~(size_t)(TPB * UNROLL * 16 - 1), nspan = bytes / (TPB * UNROLL * 16);
    f32x4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 0, bytes));
    printf("buffer %.2f GiB\n", bytes / double(1 << 30));
    // reader policies after the product's writer (nt, XCD-contiguous)
    combo<1, 0>(buf, out, bytes, nspan, false); combo<1, 1>(buf, out, bytes, nspan, false);
    combo<1, 2>(buf, out, bytes, nspan, false); combo<1, 3>(buf, out, bytes, nspan, false);
    combo<1, 4>(buf, out, bytes, nspan, false); combo<1, 5>(buf, out, bytes, nspan, false);
    combo<1, 6>(buf, out, bytes, nspan, false); combo<1, 7>(buf, out, bytes, nspan, false);
    combo<1, 1>(buf, out, bytes, nspan, true);  combo<1, 5>(buf, out, bytes, nspan, true);
    // writer policies before the product's reader (nt, plain order)
    combo<0, 1>(buf, out, bytes, nspan, false); combo<2, 1>(buf, out, bytes, nspan, false);
    combo<3, 1>(buf, out, bytes, nspan, false); combo<4, 1>(buf, out, bytes, nspan, false);
    combo<5, 1>(buf, out, bytes, nspan, false); combo<6, 1>(buf, out, bytes, nspan, false);
    combo<7, 1>(buf, out, bytes, nspan, false);
    combo<1, 1>(buf, out, bytes, nspan, false, false); combo<5, 1>(buf, out, bytes, nspan, false, false);
    return 0;
}
