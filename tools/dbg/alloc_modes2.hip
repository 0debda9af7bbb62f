// Slow-to-write physical regions: does the rate depend on HOW the 8 XCDs' streams are laid over the
// window?  One 240-GiB arena; for windows 0..5 (24 GiB each) write with
//   chunk G: XCD k writes chunks k, k+8, k+16, ... of G bytes (G = window/8: the product kernels'
//            "XCD-contiguous" order), each chunk linearly;
//   skew   : G = window/8, XCD k starts k * skew bytes into its eighth (wraps around).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TPB = 256, UNROLL = 8;            // span = 32 KiB
__global__ __launch_bounds__(TPB) void k_write(f32x4* __restrict__ p, size_t nspan, size_t spc, size_t skew_spans)
{
    // block b: xcd = b % 8, j = b / 8 -> j-th span of that XCD
    const size_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const size_t per = nspan / 8;
    if (j >= per) return;
    size_t sp;
    if (spc >= per) {                         // one chunk per XCD (+ optional skew)
        sp = xcd * per + (j + xcd * skew_spans) % per;
    } else {
        const size_t c = j / spc, r = j % spc;
        sp = (c * 8 + xcd) * spc + r;
    }
    f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) __builtin_nontemporal_store(v, q + i * TPB);
}
static float timeit(void* p, size_t bytes, size_t chunk, size_t skew)
{
    const size_t span = TPB * UNROLL * 16, nspan = bytes / span;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_write, dim3((unsigned)nspan), dim3(TPB), 0, 0, (f32x4*)p, nspan, chunk / span, skew / span);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    return best;
}
int main()
{
    const size_t W = (size_t)24 << 30;
    char* arena; CK(hipMalloc((void**)&arena, (size_t)240 << 30));
    const size_t chunks[] = {W / 8, (size_t)1 << 30, (size_t)256 << 20, (size_t)64 << 20, (size_t)16 << 20, (size_t)4 << 20, (size_t)1 << 20, (size_t)256 << 10, (size_t)32 << 10};
    printf("TB/s per window 0..5          ");
    for (int w = 0; w < 6; ++w) printf("   w%d  ", w);
    printf("\n");
    for (size_t g : chunks) {
        printf("chunk %10zu KiB         ", g >> 10);
        for (int w = 0; w < 6; ++w) printf(" %6.2f", W / timeit(arena + w * W, W, g, 0) / 1e9);
        printf("\n");
    }
    const size_t skews[] = {(size_t)32 << 10, (size_t)1 << 20, (size_t)33 << 20, (size_t)128 << 20, (size_t)384 << 20};
    for (size_t s : skews) {
        printf("eighths, skew %8zu KiB   ", s >> 10);
        for (int w = 0; w < 6; ++w) printf(" %6.2f", W / timeit(arena + w * W, W, W / 8, s) / 1e9);
        printf("\n");
    }
    return 0;
}
