// k_common.hpp -- broadcast descriptors, chunk tiles, history buffer layout, XCD tile order
#pragma once
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, this file first).  Not a standalone header.


// -DMRPHY_DEV_KNOBS build only (tools/): per-workgroup time stamps for the wave timeline of the
// line kernels.  stamps[4 id + {0, 1, 2, 3}] = start, end (s_memrealtime: 100 MHz), HW_ID | XCC_ID << 32,
// blockIdx.x.  The shipped library has neither the argument field nor the code.
#ifdef MRPHY_DEV_KNOBS
#define MRPHY_STAMP_FIELD unsigned long long* stamps; int prio_rot; int prio_shift;
#define MRPHY_STAMP_BEGIN() const unsigned long long stamp_t0_ = __builtin_amdgcn_s_memrealtime();
#define MRPHY_STAMP_END(a_, id_)                                                                  \
    if ((a_).stamps && (threadIdx.x & 63) == 0) {                                                 \
        unsigned long long* q_ = (a_).stamps + 4 * (id_);                                          \
        q_[0] = stamp_t0_;                                                                         \
        q_[1] = __builtin_amdgcn_s_memrealtime();                                                  \
        q_[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                    \
                ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf) << 32);     \
        q_[3] = blockIdx.x;                                                                        \
    }
// experiment: rotate the wave's issue priority with its progress, (period + wave slot on the SIMD) & 3,
// so that the waves sharing a SIMD advance at the same average rate (the arbiter prefers the oldest)
#define MRPHY_PRIO_INIT(a_) const unsigned prio_slot_ = (a_).prio_rot == 1 ? (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xfu) : 0u;
#define MRPHY_PRIO_TICK(a_, period_)                                                             \
    if ((a_).prio_rot) {                                                                          \
        switch ((prio_slot_ + (unsigned)(period_)) & 3u) {                                        \
        case 0: __builtin_amdgcn_s_setprio(0); break;                                             \
        case 1: __builtin_amdgcn_s_setprio(1); break;                                             \
        case 2: __builtin_amdgcn_s_setprio(2); break;                                             \
        default: __builtin_amdgcn_s_setprio(3); break;                                            \
        }                                                                                         \
    }
#else
#define MRPHY_STAMP_FIELD
#define MRPHY_STAMP_BEGIN()
#define MRPHY_STAMP_END(a_, id_)
#define MRPHY_PRIO_INIT(a_)
#define MRPHY_PRIO_TICK(a_, period_)
#endif

// (Bc, the broadcastable per-spin constant of mrphy_hip.h, is in geom.hpp)
#define MRPHY_XCD_SLOT(b_) ((b_) & 7u)

template <typename CT>
__device__ __forceinline__ CT bc_load(const Bc& b, int64_t n, int64_t s)
{
    return reinterpret_cast<const CT*>(b.p)[n * b.sn + s * b.sm];
}

template <typename T, typename CT>
__device__ __forceinline__ SpinConst<T, CT> load_consts(const Bc& g, const Bc& E1, const Bc& E2,
                                                        const void* E1m1, int64_t n, int64_t s)
{
    using M = typename CTr<CT>::mem;              // type of the constants in memory
    using R = typename CTr<CT>::reg;
    SpinConst<T, CT> k;
    k.g = bc_load<M>(g, n, s);
    k.relax = (E1.p != nullptr);
    if (k.relax) {
        k.e1 = bc_load<M>(E1, n, s);
        k.e2 = bc_load<M>(E2, n, s);
        Bc e = {E1m1, E1.sn, E1.sm};
        k.e1m1 = E1m1 ? bc_load<M>(e, n, s) : R(0);
    } else {
        k.e1 = k.e2 = R(1);
        k.e1m1 = R(0);
    }
    k.d1 = k.e1 - R(1);
    k.d2 = k.e2 - R(1);
    return k;
}

// ---------------------------------------------------------------------------------------------
// Chunk tile geometry: 64 rows x (3*TC) elements, LDS pitch padded by one 16-B slot so that
// "lane = row, same column" ds_read_b128 is conflict-free (slots per row is odd).
// ---------------------------------------------------------------------------------------------
template <typename T, int TC>
struct Tile {
    static constexpr int VE = V16<T>::N;            // elements per 16-B vector
    static constexpr int RL = 3 * TC;               // row length of a chunk, elements
    static constexpr int PITCH = RL + VE;           // padded LDS pitch, elements
    static constexpr int SPR = RL / VE;             // 16-B slots per row
    static constexpr int NL = SPR;                  // vector loads per lane per chunk
    static constexpr int ELEMS = WAVE * PITCH;
    static_assert(RL % VE == 0, "chunk row must be a whole number of 16-B slots");
    static_assert(SPR % 2 == 0, "pitch (SPR+1 slots) must be odd for conflict-free reads");
    using V = typename V16<T>::type;
};

// A chunk parked in registers (NL 16-B vectors per lane).  Passed and returned BY VALUE so that
// it is scalarised into VGPRs; through a pointer hipcc leaves it in scratch memory.
template <typename T, int TC>
struct Stage {
    typename Tile<T, TC>::V v[Tile<T, TC>::NL];
};

// global -> registers: lane `lane` fetches slots j = i*64 + lane of the 64 x SPR slot grid.
template <typename T, int TC>
__device__ __forceinline__ Stage<T, TC> chunk_fetch(const T* __restrict__ base, int64_t row0,
                                                    int64_t rows, int64_t rowlen, int64_t t0,
                                                    int lane)
{
    using TL = Tile<T, TC>;
    Stage<T, TC> st;
#pragma unroll
    for (int i = 0; i < TL::NL; ++i) {
        const int j = i * WAVE + lane;
        const int jr = j / TL::SPR, jc = j % TL::SPR;
        int64_t rr = row0 + jr;
        rr = rr < rows ? rr : rows - 1;
        const T* src = base + rr * rowlen + t0 * 3 + jc * TL::VE;
        st.v[i] = *reinterpret_cast<const typename V16<T>::utype*>(src);   // element-aligned
    }
    return st;
}

template <typename T, int TC>
__device__ __forceinline__ void chunk_to_lds(T* tile, const Stage<T, TC> st, int lane)
{
    using TL = Tile<T, TC>;
#pragma unroll
    for (int i = 0; i < TL::NL; ++i) {
        const int j = i * WAVE + lane;
        const int jr = j / TL::SPR, jc = j % TL::SPR;
        *reinterpret_cast<typename TL::V*>(tile + jr * TL::PITCH + jc * TL::VE) = st.v[i];
    }
}

// LDS tile -> global, coalesced (the inverse mapping); rows beyond `rows` are skipped.
template <typename T, int TC>
__device__ __forceinline__ void chunk_store(const T* tile, T* __restrict__ base, int64_t row0,
                                            int64_t rows, int64_t rowlen, int64_t t0, int lane)
{
    using TL = Tile<T, TC>;
#pragma unroll
    for (int i = 0; i < TL::NL; ++i) {
        const int j = i * WAVE + lane;
        const int jr = j / TL::SPR, jc = j % TL::SPR;
        const int64_t rr = row0 + jr;
        const typename TL::V v =
            *reinterpret_cast<const typename TL::V*>(tile + jr * TL::PITCH + jc * TL::VE);
        if (rr < rows)
            *reinterpret_cast<typename V16<T>::utype*>(base + rr * rowlen + t0 * 3 + jc * TL::VE) = v;
    }
}

__device__ __forceinline__ void vec_unpack(const f32x4 v, float* o)
{
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void vec_unpack(const f64x2 v, double* o)
{
    o[0] = v.x; o[1] = v.y;
}
__device__ __forceinline__ f32x4 vec_pack(const float* o) { return f32x4{o[0], o[1], o[2], o[3]}; }
__device__ __forceinline__ f64x2 vec_pack(const double* o) { return f64x2{o[0], o[1]}; }

// ---------------------------------------------------------------------------------------------
// History of the forward sweep for the adjoint: the magnetisation BEFORE each step.  It is an
// internal buffer (never an API tensor), so it is laid out for the kernels, structure-of-arrays
// per 64-spin tile:  hist[tile][t][xyz][lane].  Every store / load is one fully coalesced 256-B
// wave access; no LDS transposition is needed on either side.
// ---------------------------------------------------------------------------------------------
// (HIST_STEP = 3 * WAVE elements per time step of one tile: geom.hpp)
// Since ABI 5 the tiles may live in several separately allocated parts (HistParts, geom.hpp): where tile `tile` starts.
// Wave-uniform arithmetic, once per wave; the select chain keeps the pointer table in SGPRs (a dynamically indexed
// kernarg array is apt to be copied to scratch, and no kernel of this library has a private segment).
template <typename T>
__device__ __forceinline__ T* hist_tile_base(const HistParts& h, int64_t tile, int64_t nT)
{
    unsigned part = 0, local = (unsigned)tile;
    if (h.n_parts > 1) {
        const unsigned t = (unsigned)tile;
        if (h.interleaved) { part = t % (unsigned)h.n_parts; local = t / (unsigned)h.n_parts; }
        else               { part = t / h.tiles_per_part;    local = t - part * h.tiles_per_part; }
    }
    void* b = h.p[0];
#pragma unroll
    for (int i = 1; i < HIST_MAX_PARTS; ++i) b = part == (unsigned)i ? h.p[i] : b;
    return reinterpret_cast<T*>(b) + (int64_t)local * nT * HIST_STEP;
}

template <typename T>
__device__ __forceinline__ void hist_store(T* hp, int64_t t, T mx, T my, T mz)
{
    T* q = hp + t * HIST_STEP;      // written once, read once by the adjoint: nt (plain: same time)
    __builtin_nontemporal_store(mx, q);
    __builtin_nontemporal_store(my, q + WAVE);
    __builtin_nontemporal_store(mz, q + 2 * WAVE);
}

template <typename T>
__device__ __forceinline__ void hist_load(const T* hp, int64_t t, T& mx, T& my, T& mz)
{
    const T* q = hp + t * HIST_STEP;
    mx = __builtin_nontemporal_load(q);
    my = __builtin_nontemporal_load(q + WAVE);
    mz = __builtin_nontemporal_load(q + 2 * WAVE);
}

// Blocks are dealt round-robin to the 8 XCDs; with this map each XCD walks its own contiguous
// eighth of the spin tiles (see run_rfgr2beff for what that is worth on the write side).
__device__ __forceinline__ int64_t xcd_tile(unsigned per_xcd, bool reversed = false)
{
    if (!per_xcd) return (int64_t)blockIdx.x;
    const unsigned k = blockIdx.x >> 3;
    return (int64_t)MRPHY_XCD_SLOT(blockIdx.x) * per_xcd + (reversed ? per_xcd - 1 - k : k);
}

template <bool NT>
__device__ __forceinline__ f32x4 ldv(const f32x4* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

template <bool NT>
__device__ __forceinline__ f64x2 ldv(const f64x2* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 f32x2_u __attribute__((aligned(4)));


// ---------------------------------------------------------------------------------------------
// Store with a cache policy (round 4).  pol 0: plain; 1: non-temporal (`nt`); 2: `sc1 nt` -- an agent-scope (write-through)
// streaming store, which does NOT leave the line in the 256-MB memory-side cache.  K0 writes Beff with it (below
// 8 GB): with plain or nt stores its eight XCD streams end with their last 32 MB each resident and dirty there, and
// the K1 that follows pays for their eviction -- 0.59 instead of 0.78 of HBM peak at 64^3 x 1024, 0.68 instead of
// 0.82 on a 1/8 shard (tools/k0var_step_ab.py, profiles/r04_k0_store_policy.json).  The encoding has no builtin
// (__builtin_nontemporal_store gives `nt` alone; scoped atomics are 4 bytes wide): one inline instruction per width.
// ---------------------------------------------------------------------------------------------
// Only 16-B stores take the write-through encodings: a narrower `sc1` store is one fabric write each (the guide: dwordx2 2.7 x,
// dword ~6 x the dwordx4 time per byte; measured here: the 8-coil K0 with 8-B `sc1 nt` stores 0.74 -> 4.7 ms at 64^3 x 1024),
// so 4- and 8-B stores fall back to the plain `nt` hint whatever the policy says.
// The instruction is invisible to LLVM's hazard recogniser: gfx940+ wants two wait states between a VMEM store of more than 64 bits
// and a VALU write of its data VGPRs.  The `s_nop 1` that follows it inside the same asm statement supplies them wherever the
// statement is inlined (round 4 was safe only because a branch happened to follow every such store: ADVICE r4).
#define MRPHY_STORE_ASM(BITS)                                                                                         \
    do {                                                                                                          \
        if constexpr (sizeof(V) == 16)      asm volatile("global_store_dwordx4 %0, %1, off " BITS "\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory"); \
        else __builtin_nontemporal_store(v, dst);                                                                 \
    } while (0)
template <typename V>
__device__ __forceinline__ void store_pol(V* dst, const V v, int pol)
{
    if (pol == 2) {
        MRPHY_STORE_ASM("sc1 nt");
#ifdef MRPHY_DEV_KNOBS
    } else if (pol == 3) { MRPHY_STORE_ASM("sc1");
    } else if (pol == 4) { MRPHY_STORE_ASM("sc0 sc1");
    } else if (pol == 5) { MRPHY_STORE_ASM("sc0 sc1 nt");
    } else if (pol == 6) { MRPHY_STORE_ASM("sc0 nt");
    } else if (pol == 7) { MRPHY_STORE_ASM("sc0");
#endif
    } else if (pol) {
        __builtin_nontemporal_store(v, dst);
    } else {
        *dst = v;
    }
}
