// vmm_alloc.cpp -- experiment (round 6): a device allocator that builds every large allocation from CHUNKS separately
// created physical allocations (hipMemCreate), mapped next to each other in one reserved virtual range -- one contiguous
// tensor for the caller, several allocations for the driver.  Question: does a block built this way behave, for the
// kernels that stream writes into it (K0, K1h, K3), like a history dealt to separately allocated parts (profiles/
// r06_hist_parts_*.json: the fast placement mode in 12 of 16 fresh processes against 4 of 22 for one hipMalloc)?
// Signatures are those of torch.cuda.memory.CUDAPluggableAllocator.
//
// What round 5's attempt got wrong is not known (gpurun_out/r05c/vmm.log: a GPU memory fault at the first mapping that
// had a 16-GiB spacer handle between the halves; the tool was never committed).  This one: the granularity the driver
// RECOMMENDS, every sub-range mapped before hipMemSetAccess covers the whole reservation, no unmapped holes inside a
// reservation, no spacer handles, every call checked (a failure returns nullptr, never a half-built range).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Block {
    size_t total = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
    std::vector<size_t> slots;            // virtual slot of the i-th created chunk
    bool plain = false;
};
std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;
int g_chunks = 4;                      // > 0: this many chunks per allocation;  0: chunks of g_chunk_bytes
size_t g_chunk_bytes = 0;
int g_permute = 0;                     // 1: chunk j (in order of creation) is mapped at slot (j * stride) mod n, stride ~ n / 1.618;  2: a pseudo-random shuffle
size_t g_min_bytes = (size_t)256 << 20;

#define VMM_TRY(x)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "vmm_alloc: %s -> %s\n", #x, hipGetErrorString(e_));                \
            ok = false;                                                                         \
        }                                                                                       \
    } while (0)
}  // namespace

extern "C" {

void mrphy_vmm_config(int chunks, size_t min_bytes)
{
    std::lock_guard<std::mutex> l(g_mu);
    g_chunks = chunks < 1 ? 1 : chunks;
    g_chunk_bytes = 0;
    g_permute = 0;
    g_min_bytes = min_bytes;
}

// chunks of `chunk_bytes` each (as many as the allocation needs), mapped in creation order or -- permute = 1 -- scattered
// over the virtual range by a stride permutation: consecutive virtual chunks are then far apart in the order the
// driver created their physical memory in
void mrphy_vmm_config_chunk_bytes(size_t chunk_bytes, int permute, size_t min_bytes)
{
    std::lock_guard<std::mutex> l(g_mu);
    g_chunks = 0;
    g_chunk_bytes = chunk_bytes;
    g_permute = permute;
    g_min_bytes = min_bytes;
}

void* mrphy_vmm_alloc(size_t size, int device, hipStream_t)
{
    if (size == 0) return nullptr;
    int chunks, permute;
    size_t min_bytes, chunk_bytes;
    { std::lock_guard<std::mutex> l(g_mu); chunks = g_chunks; min_bytes = g_min_bytes; chunk_bytes = g_chunk_bytes; permute = g_permute; }
    bool ok = true;
    int prev = 0;
    VMM_TRY(hipGetDevice(&prev));
    VMM_TRY(hipSetDevice(device));
    void* out = nullptr;
    Block b;
    if (size < min_bytes || chunks == 1) {
        VMM_TRY(hipMalloc(&out, size));
        if (!ok) out = nullptr;
        b.plain = true;
        b.total = size;
    } else {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        VMM_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        if (gran == 0) gran = (size_t)2 << 20;
        if (chunks == 0) {
            const size_t cb = (chunk_bytes + gran - 1) / gran * gran;
            chunks = (int)((size + cb - 1) / cb);
        }
        const size_t per = ((size + chunks - 1) / chunks + gran - 1) / gran * gran;
        const size_t total = per * chunks;
        void* va = nullptr;
        if (ok) VMM_TRY(hipMemAddressReserve(&va, total, gran, nullptr, 0));
        // slot of the j-th created chunk: identity, or a stride permutation (stride coprime with the chunk count)
        long stride = 1;
        if (permute && chunks > 2) {
            stride = (long)(chunks / 1.6180339887);
            auto gcd = [](long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; };
            while (stride > 1 && gcd(stride, chunks) != 1) --stride;
        }
        b.slots.resize(chunks);
        std::vector<size_t> perm(chunks);
        for (int i = 0; i < chunks; ++i) perm[i] = (size_t)(((long)i * stride) % chunks);
        if (permute == 2) {                         // a pseudo-random shuffle (xorshift64, fixed seed: reproducible)
            for (int i = 0; i < chunks; ++i) perm[i] = (size_t)i;
            uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)chunks;
            for (int i = chunks - 1; i > 0; --i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const int j = (int)(x % (uint64_t)(i + 1));
                const size_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
            }
        }
        for (int i = 0; ok && i < chunks; ++i) {
            const size_t slot = perm[i];
            hipMemGenericAllocationHandle_t h;
            VMM_TRY(hipMemCreate(&h, per, &prop, 0));
            if (!ok) break;
            VMM_TRY(hipMemMap((char*)va + slot * per, per, 0, h, 0));
            if (!ok) { (void)hipMemRelease(h); break; }
            b.handles.push_back(h);
            b.sizes.push_back(per);
            b.slots[i] = slot;
        }
        if (ok) {
            hipMemAccessDesc acc = {};
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = device;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            VMM_TRY(hipMemSetAccess(va, total, &acc, 1));
        }
        if (!ok) {                                  // undo whatever was built
            for (size_t i = 0; i < b.handles.size(); ++i) {
                (void)hipMemUnmap((char*)va + b.slots[i] * b.sizes[i], b.sizes[i]);
                (void)hipMemRelease(b.handles[i]);
            }
            if (va) (void)hipMemAddressFree(va, total);
            (void)hipGetLastError();
            out = nullptr;
        } else {
            out = va;
            b.total = total;
        }
    }
    (void)hipSetDevice(prev);
    if (out) { std::lock_guard<std::mutex> l(g_mu); g_blocks[out] = b; }
    return out;
}

void mrphy_vmm_free(void* p, size_t, int device, hipStream_t)
{
    if (!p) return;
    Block b;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_blocks.find(p);
        if (it == g_blocks.end()) { fprintf(stderr, "vmm_alloc: free of an unknown pointer %p\n", p); return; }
        b = it->second;
        g_blocks.erase(it);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();                  // nothing may still be using the range
    if (b.plain) {
        (void)hipFree(p);
    } else {
        for (size_t i = 0; i < b.handles.size(); ++i) {
            (void)hipMemUnmap((char*)p + b.slots[i] * b.sizes[i], b.sizes[i]);
            (void)hipMemRelease(b.handles[i]);
        }
        (void)hipMemAddressFree(p, b.total);
    }
    (void)hipSetDevice(prev);
}

int mrphy_vmm_live_blocks(void)
{
    std::lock_guard<std::mutex> l(g_mu);
    return (int)g_blocks.size();
}

}  // extern "C"
