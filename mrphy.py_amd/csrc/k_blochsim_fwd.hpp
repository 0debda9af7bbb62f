// k_blochsim_fwd.hpp -- K1 (blochsim forward): chunked and line-granular kernels
// Fragment: included INSIDE a translation unit's anonymous namespace, after host_common.hpp (HIP runtime,
// include/mrphy_hip.h, geom.hpp, bloch_math.hpp, k_common.hpp).  Not a standalone header.
#pragma once

// =============================================================================================
// K1: blochsim forward, materialised Beff.
// =============================================================================================
template <typename T>
struct FwdArgs {
    const T* Mi;
    const T* Beff;
    T* Mo;
    HistParts hist;        // SAVE builds: where the magnetisation before each step goes (n_parts = 0: not wanted)
    Bc g, E1, E2;
    const void* E1m1;
    int64_t rows, nM, nT;
    int vec_ok;
    unsigned per_xcd;      // line kernels: > 0 -> block b works on spin tile (b % 8) * per_xcd + b / 8
    int xcd_rev;           // ... or on (b % 8) * per_xcd + per_xcd - 1 - b / 8: each XCD walks its eighth from the end
    MRPHY_STAMP_FIELD
};

template <typename T, typename CT, int TC, bool SAVE>
__global__ __launch_bounds__(WAVE) void k_bloch_fwd(FwdArgs<T> a)
{
    using TL = Tile<T, TC>;
    using V = typename TL::V;
    constexpr int VE = TL::VE;
    __shared__ __attribute__((aligned(16))) T tile[TL::ELEMS];

    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);

    T mx = a.Mi[rc * 3 + 0], my = a.Mi[rc * 3 + 1], mz = a.Mi[rc * 3 + 2];
    const int64_t rowlen = 3 * a.nT;
    int64_t t = 0;

    T* hp = SAVE ? hist_tile_base<T>(a.hist, (int64_t)blockIdx.x, a.nT) + lane : nullptr;
    if (a.vec_ok) {
        const int64_t nfull = a.nT / TC;
        Stage<T, TC> st;
        if (nfull > 0) st = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, 0, lane);
        T* myrow = tile + lane * TL::PITCH;
        for (int64_t c = 0; c < nfull; ++c) {
            __syncthreads();                         // tile free (previous chunk consumed)
            chunk_to_lds<T, TC>(tile, st, lane);
            __syncthreads();
            if (c + 1 < nfull)                       // next chunk flies while this one integrates
                st = chunk_fetch<T, TC>(a.Beff, row0, a.rows, rowlen, (c + 1) * TC, lane);
#pragma unroll 1
            for (int tt = 0; tt < TC; tt += VE) {    // VE steps = 3 vectors = 48 B per lane
                T bb[3 * VE];
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3), bb);
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3 + VE), bb + VE);
                vec_unpack(*reinterpret_cast<const V*>(myrow + tt * 3 + 2 * VE), bb + 2 * VE);
                T gBx[VE], gBy[VE], gBz[VE];
#pragma unroll
                for (int q = 0; q < VE; ++q) { gBx[q] = bb[3 * q]; gBy[q] = bb[3 * q + 1]; gBz[q] = bb[3 * q + 2]; }
                Rot<T> rr[VE];
                rot_prepare<T, CT, VE>(k, gBx, gBy, gBz, rr);
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    if (SAVE) hist_store<T>(hp, c * TC + tt + q, mx, my, mz);
                    if (k.relax) rot_apply<true, T, CT>(k, rr[q], mx, my, mz);
                    else         rot_apply<false, T, CT>(k, rr[q], mx, my, mz);
                }
            }
        }
        t = nfull * TC;
    }
    // tail steps and the unaligned-shape path: each lane reads its own samples directly
    const T* bp = a.Beff + rc * rowlen;
    for (; t < a.nT; ++t) {
        if (SAVE) hist_store<T>(hp, t, mx, my, mz);
        bloch_step<T, CT>(k, bp[t * 3], bp[t * 3 + 1], bp[t * 3 + 2], mx, my, mz);
    }
    if (valid) { a.Mo[r * 3] = mx; a.Mo[r * 3 + 1] = my; a.Mo[r * 3 + 2] = mz; }
}


// =============================================================================================
// K1, line-granular variant (the headline path): float data, no history, rows 128-B aligned
// (Beff base % 128 == 0 and nT % 32 == 0).
//
// The chunked kernel above fetches 16 steps = 192 B per spin per chunk, i.e. one and a half
// cache lines: measured with FETCH_SIZE it reads 1.22x the algorithmic bytes, because the shared
// half line has usually left L2 when the next chunk asks for it.  Here the unit of transfer is
// ONE 128-B line per spin ("piece" = 32 floats = 10 2/3 steps):
//   * a piece of the 64-spin tile is 8 wave-loads; load i, lane l fetches 16 B of row 8i + l/8 at
//     byte 16*(l%8) of that row's line: every wave-load covers 8 rows x one WHOLE line;
//   * the next piece waits in 8 VGPR quads (32 VGPRs) while the current one is integrated;
//   * LDS tile 64 x (32+4) floats = 9 KB; lane = spin reads its row with conflict-free reads
//     (pitch 9 x 16 B, odd);
//   * a step needs 3 consecutive floats, so steps straddle piece boundaries; 3 pieces = 96 floats =
//     32 steps is the period: piece 0 holds steps 0-9 + 2 floats of step 10, piece 1 the rest of
//     step 10, steps 11-20 + 1 float of step 21, piece 2 the rest of step 21 and steps 22-31.  The
//     straddling floats travel in two carry registers.
// =============================================================================================
// NA steps whose samples start at float `first` of this lane's LDS row, optionally preceded by a
// straddling step whose leading floats arrive in registers.
// SAVE: record the magnetisation before each step at hist[t], t = th, th+1, ...
template <bool RELAX, bool SAVE, bool PIN, typename CT, int NA, typename T = float>
__device__ __forceinline__ void lines_steps(const SpinConst<T, CT>& k, const T* q,
                                            T* hp, int64_t th, T& mx, T& my, T& mz)
{
    T Bx[NA], By[NA], Bz[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { Bx[j] = q[3 * j]; By[j] = q[3 * j + 1]; Bz[j] = q[3 * j + 2]; }
    Rot<T> r[NA];
    rot_prepare<T, CT, NA>(k, Bx, By, Bz, r);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        if (SAVE) hist_store<T>(hp, th + j, mx, my, mz);
        rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
    }
    if (PIN) pin_state(mx, my, mz);
}

// 1 straddling step (b0,b1,b2 given) + NA steps from q
template <bool RELAX, bool SAVE, bool PIN, typename CT, int NA, typename T = float>
__device__ __forceinline__ void lines_steps_carry(const SpinConst<T, CT>& k, T b0, T b1,
                                                  T b2, const T* q, T* hp, int64_t th,
                                                  T& mx, T& my, T& mz)
{
    T Bx[NA + 1], By[NA + 1], Bz[NA + 1];
    Bx[0] = b0; By[0] = b1; Bz[0] = b2;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        Bx[j + 1] = q[3 * j]; By[j + 1] = q[3 * j + 1]; Bz[j + 1] = q[3 * j + 2];
    }
    Rot<T> r[NA + 1];
    rot_prepare<T, CT, NA + 1>(k, Bx, By, Bz, r);
#pragma unroll
    for (int j = 0; j < NA + 1; ++j) {
        if (SAVE) hist_store<T>(hp, th + j, mx, my, mz);
        rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
    }
    if (PIN) pin_state(mx, my, mz);
}

// OCC: waves per SIMD the register allocation is bounded for.  SPLIT: sub-batches per piece
// (2: 5/6 steps prepared at once, 3: 3/4 steps -- fewer live registers).  NT: non-temporal loads.
// (Tried: three pieces in flight per wave instead of one -- 96 prefetch VGPRs, 2 waves/SIMD -- no
// gain at any grid size.)
template <typename CT, bool RELAX, int OCC, int SPLIT, bool NT, bool SAVE, bool PIN = false>
__global__ __launch_bounds__(WAVE, OCC) void k_bloch_fwd_lines(FwdArgs<float> a)
{
    using T = float;
    constexpr int PF = 32;                 // floats per piece = one 128-B line
    constexpr int PITCH = PF + 4;          // 9 slots of 16 B
    __shared__ __attribute__((aligned(16))) T tile[WAVE * PITCH];

    const int lane = threadIdx.x;
    const int64_t tile_id = xcd_tile(a.per_xcd, a.xcd_rev != 0);
    if (tile_id * WAVE >= a.rows) return;
    MRPHY_STAMP_BEGIN()
    const int64_t row0 = tile_id * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
    T mx = a.Mi[rc * 3 + 0], my = a.Mi[rc * 3 + 1], mz = a.Mi[rc * 3 + 2];

    const int64_t rowlen = 3 * a.nT;                       // floats; multiple of 96
    const int64_t npieces = rowlen / PF;                   // multiple of 3
    const int frow = lane >> 3, fcol = (lane & 7) * 4;
    // wave-uniform base (SGPRs) + 32-bit per-lane offsets: loads use the saddr+voffset form and
    // need 8 VGPRs of addressing instead of 16 (host guarantees 64*rowlen < 2^31)
    const T* __restrict__ base = a.Beff + row0 * rowlen;
    const int64_t last = a.rows - 1 - row0;                // last valid row of this tile
    // byte offset of load i = min(off0 + i * ostride, olim): rows past the end of the last tile
    // re-read its last valid row (two VGPRs instead of eight precomputed offsets)
    const unsigned ostride = (unsigned)(8 * rowlen * sizeof(T));
    const unsigned off0 = (unsigned)(((frow < last ? frow : last) * rowlen + fcol) * sizeof(T));
    const unsigned olim = (unsigned)(((last < 63 ? last : 63) * rowlen + fcol) * sizeof(T));
// (o0 is laundered through an empty asm per piece, or the compiler hoists all eight offsets back
// into registers for the whole loop)
#define MRPHY_OFF(i) (min(o0 + (unsigned)(i) * ostride, olim))
    T* wr = tile + frow * PITCH + fcol;                    // + i*8*PITCH per load
    const T* my_ = tile + lane * PITCH;

    f32x4 st0[8];
#define MRPHY_FETCH(S, p)                                                                  \
    { unsigned o0 = off0; asm volatile("" : "+v"(o0));                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        S[i] = ldv<NT>(reinterpret_cast<const f32x4*>(                                      \
            reinterpret_cast<const char*>(base + (p) * PF) + MRPHY_OFF(i))); }
#define MRPHY_STAGE(S)                                                                     \
    __syncthreads();                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        *reinterpret_cast<f32x4*>(wr + i * 8 * PITCH) = S[i];                              \
    __syncthreads();

    T* hp = SAVE ? hist_tile_base<T>(a.hist, tile_id, a.nT) + lane : nullptr;
#define LS(NA_, Q_, TH_) lines_steps<RELAX, SAVE, PIN, CT, NA_>(k, my_ + (Q_), hp, t0 + (TH_), mx, my, mz)
#define LC(NA_, B0_, B1_, B2_, Q_, TH_) \
    lines_steps_carry<RELAX, SAVE, PIN, CT, NA_>(k, B0_, B1_, B2_, my_ + (Q_), hp, t0 + (TH_), mx, my, mz)
    if (npieces > 0) { MRPHY_FETCH(st0, 0) }
    T c0, c1;
    MRPHY_PRIO_INIT(a)
    for (int64_t p = 0; p < npieces; p += 3) {
        const int64_t t0 = (p / 3) * 32;
        MRPHY_PRIO_TICK(a, p / 3)
        const bool more = p + 3 < npieces;
        // piece 0: steps 0..9 (floats 0..29), carry floats 30, 31
        MRPHY_STAGE(st0)
        MRPHY_FETCH(st0, p + 1)
        if (SPLIT == 2)      { LS(5, 0, 0); LS(5, 15, 5); }
        else if (SPLIT == 3) { LS(4, 0, 0); LS(3, 12, 4); LS(3, 21, 7); }
        else                 { LS(3, 0, 0); LS(3, 9, 3); LS(2, 18, 6); LS(2, 24, 8); }
        c0 = my_[30]; c1 = my_[31];
        // piece 1: step 10 = (c0, c1, f0); steps 11..20 from float 1; carry float 31
        MRPHY_STAGE(st0)
        MRPHY_FETCH(st0, p + 2)
        if (SPLIT == 2)      { LC(5, c0, c1, my_[0], 1, 10); LS(5, 16, 16); }
        else if (SPLIT == 3) { LC(3, c0, c1, my_[0], 1, 10); LS(4, 10, 14); LS(3, 22, 18); }
        else { LC(2, c0, c1, my_[0], 1, 10); LS(3, 7, 13); LS(3, 16, 16); LS(2, 25, 19); }
        c0 = my_[31];
        // piece 2: step 21 = (c0, f0, f1); steps 22..31 from float 2
        MRPHY_STAGE(st0)
        if (more) { MRPHY_FETCH(st0, p + 3) }
        if (SPLIT == 2)      { LC(5, c0, my_[0], my_[1], 2, 21); LS(5, 17, 27); }
        else if (SPLIT == 3) { LC(3, c0, my_[0], my_[1], 2, 21); LS(4, 11, 25); LS(3, 23, 29); }
        else { LC(2, c0, my_[0], my_[1], 2, 21); LS(3, 8, 24); LS(3, 17, 27); LS(2, 26, 30); }
    }
#undef LS
#undef LC
#undef MRPHY_FETCH
#undef MRPHY_STAGE
#undef MRPHY_OFF
    if (valid) { a.Mo[r * 3] = mx; a.Mo[r * 3 + 1] = my; a.Mo[r * 3 + 2] = mz; }
    MRPHY_STAMP_END(a, tile_id)
}


// =============================================================================================
// K1, line-granular, DOUBLE precision (round 4): the reference's own tests run in fp64
// (tests/test_sims.py:16), and the chunked kernel above reads fp64 rows 192 B at a time -- one and a
// half lines, 1.22 x the algorithmic bytes, 4.1-4.5 TB/s.  Same scheme as k_bloch_fwd_lines with the
// numbers of an 8-byte element: a piece = one 128-B line per spin = 16 doubles = 5 1/3 steps, period
// 3 pieces = 48 doubles = 16 steps (rows 128-B aligned, nT % 16 == 0):
//     piece 0: steps 0-4 (doubles 0-14), double 15 carried;
//     piece 1: step 5 = (carry, d0, d1), steps 6-9 (doubles 2-13), doubles 14, 15 carried;
//     piece 2: step 10 = (carry, carry, d0), steps 11-15 (doubles 1-15).
// Batches of at most two steps (a Rot<double> is 10 VGPRs): 150-160 VGPRs, three waves per SIMD, no scratch.
// A wave-load is still 8 rows x one whole line (lane l: row 8i + l/8, 16 B at byte 16 (l % 8)); LDS
// tile 64 x (16 + 2) doubles = 9 KB, pitch 9 x 16 B (odd: conflict-free row reads).
// =============================================================================================
template <typename CT, bool RELAX, int OCC, bool NT, bool SAVE, bool PIN>
__global__ __launch_bounds__(WAVE, OCC) void k_bloch_fwd_lines_f64(FwdArgs<double> a)
{
    using T = double;
    constexpr int PF = 16;                 // doubles per piece = one 128-B line
    constexpr int PITCH = PF + 2;          // 9 slots of 16 B
    __shared__ __attribute__((aligned(16))) T tile[WAVE * PITCH];

    const int lane = threadIdx.x;
    const int64_t tile_id = xcd_tile(a.per_xcd, a.xcd_rev != 0);
    if (tile_id * WAVE >= a.rows) return;
    const int64_t row0 = tile_id * WAVE;
    const int64_t r = row0 + lane;
    const bool valid = r < a.rows;
    const int64_t rc = valid ? r : a.rows - 1;
    const int64_t n = rc / a.nM, s = rc % a.nM;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
    T mx = a.Mi[rc * 3 + 0], my = a.Mi[rc * 3 + 1], mz = a.Mi[rc * 3 + 2];

    const int64_t rowlen = 3 * a.nT;                       // doubles; multiple of 48
    const int64_t npieces = rowlen / PF;                   // multiple of 3
    const int frow = lane >> 3, fcol = (lane & 7) * 2;
    const T* __restrict__ base = a.Beff + row0 * rowlen;
    const int64_t last = a.rows - 1 - row0;
    const unsigned ostride = (unsigned)(8 * rowlen * sizeof(T));
    const unsigned off0 = (unsigned)(((frow < last ? frow : last) * rowlen + fcol) * sizeof(T));
    const unsigned olim = (unsigned)(((last < 63 ? last : 63) * rowlen + fcol) * sizeof(T));
#define MRPHY_OFF(i) (min(o0 + (unsigned)(i) * ostride, olim))
    T* wr = tile + frow * PITCH + fcol;
    const T* my_ = tile + lane * PITCH;

    f64x2 st0[8];
#define MRPHY_FETCH(S, p)                                                                  \
    { unsigned o0 = off0; asm volatile("" : "+v"(o0));                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        S[i] = ldv<NT>(reinterpret_cast<const f64x2*>(                                      \
            reinterpret_cast<const char*>(base + (p) * PF) + MRPHY_OFF(i))); }
#define MRPHY_STAGE(S)                                                                     \
    __syncthreads();                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
        *reinterpret_cast<f64x2*>(wr + i * 8 * PITCH) = S[i];                              \
    __syncthreads();

    T* hp = SAVE ? hist_tile_base<T>(a.hist, tile_id, a.nT) + lane : nullptr;
#define LS(NA_, Q_, TH_) lines_steps<RELAX, SAVE, PIN, CT, NA_, T>(k, my_ + (Q_), hp, t0 + (TH_), mx, my, mz)
#define LC(NA_, B0_, B1_, B2_, Q_, TH_) \
    lines_steps_carry<RELAX, SAVE, PIN, CT, NA_, T>(k, B0_, B1_, B2_, my_ + (Q_), hp, t0 + (TH_), mx, my, mz)
    if (npieces > 0) { MRPHY_FETCH(st0, 0) }
    T c0, c1;
    for (int64_t p = 0; p < npieces; p += 3) {
        const int64_t t0 = (p / 3) * 16;
        const bool more = p + 3 < npieces;
        // piece 0: steps 0..4 (doubles 0..14), carry double 15
        MRPHY_STAGE(st0)
        MRPHY_FETCH(st0, p + 1)
        LS(2, 0, 0); LS(2, 6, 2); LS(1, 12, 4);
        c0 = my_[15];
        // piece 1: step 5 = (c0, d0, d1); steps 6..9 from double 2; carry doubles 14, 15
        MRPHY_STAGE(st0)
        MRPHY_FETCH(st0, p + 2)
        LC(1, c0, my_[0], my_[1], 2, 5); LS(2, 5, 7); LS(1, 11, 9);
        c0 = my_[14]; c1 = my_[15];
        // piece 2: step 10 = (c0, c1, d0); steps 11..15 from double 1
        MRPHY_STAGE(st0)
        if (more) { MRPHY_FETCH(st0, p + 3) }
        LC(1, c0, c1, my_[0], 1, 10); LS(2, 4, 12); LS(2, 10, 14);
    }
#undef LS
#undef LC
#undef MRPHY_FETCH
#undef MRPHY_STAGE
#undef MRPHY_OFF
    if (valid) { a.Mo[r * 3] = mx; a.Mo[r * 3 + 1] = my; a.Mo[r * 3 + 2] = mz; }
}
