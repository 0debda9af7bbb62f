// HBM rate of the adjoint kernel's (K3) access pattern without any arithmetic, as a function of the
// width of the contiguous run each row is WRITTEN with:
//   per wave = one tile of 64 rows x ROWLEN bytes; time runs backwards; per turn the wave
//     reads  RPIECE = 128 B of every row of B       (8 wave-loads of 8 rows x one line, as K3)
//     reads  64 x 128 B of the tile's history H     (contiguous, SoA: 8 wave-loads of 1 KB)
//     and every WPIECE/128 turns writes WPIECE bytes of every row of G (wave-stores of
//     64*16/WPIECE rows x WPIECE bytes), non-temporal;
//   tiles in XCD-contiguous order; WAVES = waves per SIMD the kernel is bounded for (LDS-limited).
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_k3pattern.hip -o tools/hbm_k3pattern && tools/hbm_k3pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t ROWLEN = 12288;          // bytes per row: nT = 1024

template <int WPIECE, int WAVES, bool WRITE, bool HIST>
__global__ __launch_bounds__(64, WAVES) void k_pat(const char* __restrict__ B, const char* __restrict__ H,
                                                   char* __restrict__ G, float* out, unsigned per_xcd,
                                                   size_t ntiles)
{
    constexpr int K = WPIECE / 128;                 // turns per store
    constexpr int LPR = WPIECE / 16, RPS = 64 / LPR; // lanes per row / rows per wave-store
    extern __shared__ char pad[];                   // occupancy knob
    const size_t tile = (size_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const char* pb = B + tile * 64 * ROWLEN + (size_t)(lane >> 3) * ROWLEN + (lane & 7) * 16;
    const char* ph = H + tile * 64 * ROWLEN + lane * 16;
    char* pg = G + tile * 64 * ROWLEN + (size_t)(lane / LPR) * ROWLEN + (lane % LPR) * 16;
    const int nturn = ROWLEN / 128;
    f32x4 cur[8], nxt[8], hc[8];
    f32x4 acc[K][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = __builtin_nontemporal_load((const f32x4*)(pb + (size_t)(nturn - 1) * 128 + (size_t)i * 8 * ROWLEN));
    for (int t = nturn - 1; t >= 0; t -= K) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int tt = t - k;
            if (HIST) {
#pragma unroll
                for (int i = 0; i < 8; ++i) hc[i] = __builtin_nontemporal_load((const f32x4*)(ph + (size_t)tt * 64 * 128 + i * 1024));
            }
            if (tt > 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) nxt[i] = __builtin_nontemporal_load((const f32x4*)(pb + (size_t)(tt - 1) * 128 + (size_t)i * 8 * ROWLEN));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[k][i] = HIST ? cur[i] + hc[i] : cur[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
        }
        if (WRITE) {
            // K pieces x 8 quads per lane = K*8 wave-stores of RPS rows x WPIECE bytes
            const size_t off = (size_t)(t - (K - 1)) * 128;
#pragma unroll
            for (int j = 0; j < K * 8; ++j)
                __builtin_nontemporal_store(acc[j % K][j / K], (f32x4*)(pg + off + (size_t)j * RPS * ROWLEN));
        } else {
            f32x4 s = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) s += acc[k][i];
            if (s.x == 12345.678f) out[0] = 1.f;
        }
    }
}

template <int WPIECE, int WAVES, bool WRITE, bool HIST>
static void run(const char* B, const char* H, char* G, float* out, size_t ntiles, const char* label)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned per_xcd = (unsigned)((ntiles + 7) / 8);
    const size_t lds = WAVES >= 4 ? 9216 : (WAVES == 3 ? 12288 : 18432);
    auto launch = [&]() {
        hipLaunchKernelGGL((k_pat<WPIECE, WAVES, WRITE, HIST>), dim3(per_xcd * 8), dim3(64), lds, 0, B, H, G, out, per_xcd, ntiles);
    };
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double bytes = (double)ntiles * 64 * ROWLEN * (1 + (HIST ? 1 : 0) + (WRITE ? 1 : 0));
    printf("%-28s write run %4d B, %d waves/SIMD: %7.3f ms  %6.3f TB/s\n", label, WRITE ? WPIECE : 0, WAVES, best, bytes / best / 1e9);
}

int main()
{
    const size_t ntiles = 32768, bytes = ntiles * 64 * ROWLEN;      // 25.8 GB each, as 128^3 x 1024
    char *B, *H, *G; float* out;
    CK(hipMalloc(&B, bytes)); CK(hipMalloc(&H, bytes)); CK(hipMalloc(&G, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(B, 0, bytes)); CK(hipMemset(H, 0, bytes));
    run<128, 2, false, true>(B, H, G, out, ntiles, "read B + H (K3 w/o output)");
    run<128, 2, true, true>(B, H, G, out, ntiles, "K3 pattern");
    run<256, 2, true, true>(B, H, G, out, ntiles, "K3 pattern");
    run<512, 2, true, true>(B, H, G, out, ntiles, "K3 pattern");
    run<128, 3, true, true>(B, H, G, out, ntiles, "K3 pattern");
    run<256, 3, true, true>(B, H, G, out, ntiles, "K3 pattern");
    run<128, 4, true, true>(B, H, G, out, ntiles, "K3 pattern");
    run<128, 2, true, false>(B, H, G, out, ntiles, "read B, write G (1R:1W)");
    run<512, 2, true, false>(B, H, G, out, ntiles, "read B, write G (1R:1W)");
    return 0;
}
