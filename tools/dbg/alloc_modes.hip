// Is the ~15 % slower WRITE rate a property of an allocation (physical placement)?
//  A: 12 x { hipMalloc(24 GiB), write it 3x (XCD-contiguous streaming stores), hipFree }
//  B: one 240-GiB arena, the same write into 24-GiB windows at 10 offsets (twice)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TPB = 256, UNROLL = 8;
__global__ __launch_bounds__(TPB) void k_write(f32x4* __restrict__ p, size_t nspan, unsigned per_xcd)
{
    const size_t sp = (size_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (sp >= nspan) return;
    f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) __builtin_nontemporal_store(v, q + i * TPB);
}
__global__ __launch_bounds__(TPB) void k_read(const f32x4* __restrict__ p, float* out, size_t nspan, unsigned per_xcd)
{
    const size_t sp = (size_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (sp >= nspan) return;
    const f32x4* q = p + sp * (TPB * UNROLL) + threadIdx.x;
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += __builtin_nontemporal_load(q + i * TPB);
    if (s.x == 12345.678f) out[0] = 1.f;
}
static float timeit(bool wr, void* p, float* out, size_t bytes)
{
    const size_t nspan = bytes / (TPB * UNROLL * 16);
    const unsigned per_xcd = (unsigned)((nspan + 7) / 8);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a));
        if (wr) hipLaunchKernelGGL(k_write, dim3(per_xcd * 8), dim3(TPB), 0, 0, (f32x4*)p, nspan, per_xcd);
        else    hipLaunchKernelGGL(k_read, dim3(per_xcd * 8), dim3(TPB), 0, 0, (const f32x4*)p, out, nspan, per_xcd);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    return best;
}
int main()
{
    const size_t W = (size_t)24 << 30;
    float* out; CK(hipMalloc(&out, 4));
    printf("A: fresh allocations of 24 GiB\n");
    void* keep[4] = {0, 0, 0, 0};
    for (int i = 0; i < 12; ++i) {
        void* p; CK(hipMalloc(&p, W));
        const float w = timeit(true, p, out, W), r = timeit(false, p, out, W);
        printf("  alloc %2d at %p: write %6.3f ms = %5.2f TB/s   read %6.3f ms = %5.2f TB/s\n", i, p, w, W / w / 1e9, r, W / r / 1e9);
        if (i % 3 == 0 && i / 3 < 4) keep[i / 3] = p;      // keep some alive so the next ones land elsewhere
        else CK(hipFree(p));
    }
    for (int i = 0; i < 4; ++i) if (keep[i]) CK(hipFree(keep[i]));
    printf("B: windows of one 240-GiB arena\n");
    char* arena; CK(hipMalloc((void**)&arena, (size_t)240 << 30));
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < 10; ++i) {
            const float w = timeit(true, arena + i * W, out, W);
            printf("  rep %d window %2d (+%3d GiB): write %6.3f ms = %5.2f TB/s\n", rep, i, i * 24, w, W / w / 1e9);
        }
    return 0;
}
