// k_fused.hpp -- K2 (fused rf,gr -> Mo) and K2b (its adjoint)
// Fragment of the single translation unit mrphy_hip.hip: included there INSIDE its anonymous
// namespace, after <hip/hip_runtime.h>, include/mrphy_hip.h and bloch_math.hpp.  Not a standalone
// header.

// =============================================================================================
// K2: fused rf,gr -> Mo.  No Beff in HBM: the pulse sample of step t is wave-uniform (one block
// = one wave = 64 spins of ONE batch entry, so rf/gr addresses are scalar loads) and the lane's
// own loc / df/gamma / b1 sit in registers.  The field is assembled exactly as K0 rounds it
// (B first, then g*B) so that K2 == K1(K0(.)) bit for bit.  VALU-bound, not HBM-bound.
// =============================================================================================
template <typename T>
struct FusedArgs {
    const T* Mi;
    const T* rf;  int64_t rf_sn;
    const T* gr;  int64_t gr_sn;
    const T* loc;
    Bc df, gam;
    const T* b1;
    Bc g, E1, E2;
    const void* E1m1;
    T* Mo;
    T* Mck;  int64_t ck_every;
    int64_t N, nM, nT, nC;
};

// CK: write checkpoints (every ck_every steps, a multiple of the 8-step chunk).  Kept out of the
// plain instantiation so that its step loop contains no store: the pulse loads are then provably
// unclobbered and become (batched) scalar loads.
// NCM: 1 = one coil (pulse samples are scalar loads); 2 / 4 / 8 / 16 / 32 = up to that many coils (the
// lane's b1 in 2 NCM registers, the chunk's rf samples staged in LDS and read as broadcasts; the
// coil sum is ONE ascending FMA chain whatever NCM is, so every capacity -- and K0 -- rounds alike);
// 0 = any number of coils (b1 and rf from memory inside the coil loop: slow, correctness path).
// Measured at 64^3 x 1024 before the 16 / 32 capacities existed: 8 coils 0.75 ms, 9 coils 5.85 ms,
// 16 coils 20.8 ms on the memory path (tools/ptx_timing.py).
constexpr int K2_MAXC = 32;                              // largest register/LDS coil capacity
// HB1 (one-coil builds): the coil has a b1 map.  Without one Bxy = rf (beffective.py:147-151): the
// build then skips the complex product -- 6 of the ~50 VALU instructions of a step; with b1 = (1, 0)
// the product returns rf bit for bit anyway, so results are unchanged.  A template parameter, not a
// run-time test: a wave-uniform branch in the field assembly broke the batching of the pulse's scalar
// loads (round 1: 6.6 -> 7.2 ms).
template <typename T, typename CT, int NCM, bool CK, bool RELAX, bool HB1 = true>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_fwd(FusedArgs<T> a)
{
    constexpr int NS = 8;
    constexpr bool NC1 = (NCM == 1);
    constexpr bool NCR = (NCM >= 2);                     // coils in registers / LDS
    constexpr int MC = NCR ? NCM : 1;                    // coil capacity of this instantiation
    static_assert(NCM == 0 || NCM == 1 || NCM == 2 || NCM == 4 || NCM == 8 || NCM == 16 || NCM == 32,
                  "coil capacities: 2/4/8/16/32");
    static_assert(MC <= K2_MAXC, "capacity above K2_MAXC: the launcher would never select it");
    __shared__ __attribute__((aligned(16))) T srf[NCR ? 2 * NS * MC : 4];  // [re|im][j][c]
    const int lane = threadIdx.x;
    const int64_t n = blockIdx.y;
    const int64_t s_ = (int64_t)blockIdx.x * WAVE + lane;
    const bool valid = s_ < a.nM;
    const int64_t s = valid ? s_ : a.nM - 1;
    const int64_t row = n * a.nM + s;
    const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);

    T mx = a.Mi[row * 3], my = a.Mi[row * 3 + 1], mz = a.Mi[row * 3 + 2];
    const T lx = a.loc[row * 3], ly = a.loc[row * 3 + 1], lz = a.loc[row * 3 + 2];
    T delta = T(0);
    if (a.df.p) delta = bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s);
    T br = T(1), bi = T(0);
    if (NC1 && HB1 && a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }

    const int64_t nT = a.nT, nC = a.nC;
    // The pulse is read-only for the whole launch and its addresses are wave-uniform: pointers into the
    // CONSTANT address space make the loads scalar (s_load, batched) whatever else the loop does.  With
    // plain global pointers the checkpoint-writing build could not prove that its stores leave the pulse
    // alone and fetched the samples with vector loads + v_readfirstlane (K2 with checkpoints: 0.82 ms
    // where the plain build's rate gives 0.60 at 64^3 x 2048).
    using CP = const T __attribute__((address_space(4)))*;
    CP rfr = (CP)(a.rf + n * a.rf_sn);                       // [nT][nC]
    CP rfi = rfr + nT * nC;
    CP gx = (CP)(a.gr + n * a.gr_sn);
    CP gy = gx + nT;
    CP gz = gy + nT;
    const T* b1 = a.b1 ? a.b1 + row * 2 * nC : nullptr;
    const int64_t rows = a.N * a.nM;
    T b1r[MC], b1i[MC];
    if (NCR) {
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            b1r[c] = (c < nC) ? b1[c] : T(0);
            b1i[c] = (c < nC) ? b1[nC + c] : T(0);
        }
    }
    // NCR: rf samples of steps [tb, tb + cnt) -> LDS as [step][MC], ZERO beyond nC: the coil loop
    // below then needs no `c < nC` test (b1r/b1i are zero there too; adding exact zeros changes
    // nothing), stays one basic block, and its broadcast reads are batched.  With the test it compiled
    // to a branch and an exposed LDS round trip per coil, as in K0 (8 coils: 0.81 ms at 64^3 x 1024).
    auto stage_rf = [&](int64_t tb, int cnt) {
        __syncthreads();
        for (int i = lane; i < cnt * MC; i += WAVE) {
            const int j = i / MC, c = i - j * MC;
            const bool on = c < (int)nC;
            srf[i] = on ? rfr[(tb + j) * nC + c] : T(0);
            srf[NS * MC + i] = on ? rfi[(tb + j) * nC + c] : T(0);
        }
        __syncthreads();
    };
    int64_t tstage = 0;                                       // first step held in srf

    auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
        Bx = T(0); By = T(0);
        if (NC1) {
            if (HB1) field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
            else     { Bx = rfr[t]; By = rfi[t]; }
        } else if (NCR) {
            const T* qr = srf + (t - tstage) * MC;
            const T* qi = qr + NS * MC;
#pragma unroll
            for (int c = 0; c < MC; ++c) field_xy_fma<T>(b1r[c], b1i[c], qr[c], qi[c], Bx, By);
        } else {
            for (int64_t c = 0; c < nC; ++c)
                field_xy_fma<T>(b1[c], b1[nC + c], rfr[t * nC + c], rfi[t * nC + c], Bx, By);
        }
        Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
    };

    // checkpoints: a running destination and the step of the next one -- `t0 % ck_every`, `t0 / ck_every` on
    // 64-bit run-time values were a software division on the scalar unit every 8 steps (round 3: +200 scalar
    // instructions per 16 steps in the ISA of the checkpoint build)
    // The checkpoint build: all of the prologue's vector loads are awaited HERE, before the loop.  Otherwise the
    // wait for them lands in the loop header (the join of the prologue and the back edge) as s_waitcnt
    // vmcnt(0), where it also waits, every 8 steps, for the checkpoint store of the iteration before.
    if (CK) __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
    int64_t ck_next = 0;
    T* ckp = CK ? a.Mck + row * 3 : nullptr;
    const int64_t ck_pitch = rows * 3;
    int64_t t0 = 0;
    for (; t0 + NS <= nT; t0 += NS) {
        if (NCR) { tstage = t0; stage_rf(t0, NS); }
        if (CK && t0 == ck_next) {
            if (valid) { ckp[0] = mx; ckp[1] = my; ckp[2] = mz; }
            ckp += ck_pitch; ck_next += a.ck_every;
        }
        T Bx[NS], By[NS], Bz[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) field(t0 + j, Bx[j], By[j], Bz[j]);
        Rot<T> r[NS];
        rot_prepare<T, CT, NS>(k, Bx, By, Bz, r);
#pragma unroll
        for (int j = 0; j < NS; ++j) rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
    }
    if (NCR && t0 < nT) { tstage = t0; stage_rf(t0, (int)(nT - t0)); }
    for (; t0 < nT; ++t0) {                                   // nT % 8 tail
        if (CK && t0 == ck_next) {
            if (valid) { ckp[0] = mx; ckp[1] = my; ckp[2] = mz; }
            ckp += ck_pitch; ck_next += a.ck_every;
        }
        T Bx[1], By[1], Bz[1];
        field(t0, Bx[0], By[0], Bz[0]);
        Rot<T> r[1];
        rot_prepare<T, CT, 1>(k, Bx, By, Bz, r);
        rot_apply<RELAX, T, CT>(k, r[0], mx, my, mz);
    }
    if (valid) { a.Mo[row * 3] = mx; a.Mo[row * 3 + 1] = my; a.Mo[row * 3 + 2] = mz; }
}


// =============================================================================================
// K2b: adjoint of the fused kernel -- grad_Mo -> grad_Mi, grad_rf, grad_gr without Beff, history
// or grad_Beff in HBM (single-coil rf).  K2 leaves a checkpoint of M every SEG = 16 steps.  A wave
// walks the segments of its 64 spins backwards; per segment it
//   1. recomputes the 16 pre-step states from the checkpoint into registers (the very states the
//      forward pass went through, so no inversion error),
//   2. sweeps the adjoint over the 16 steps, re-assembling the field on the fly,
//   3. reduces the five per-step contributions
//        gr_x,y,z += loc_{x,y,z} * gBz     rf_re += b1r*gBx + b1i*gBy     rf_im += b1r*gBy - b1i*gBx
//      over its 64 spins with an LDS transpose-sum (80 rows x 64 lanes, slot-swizzled: conflict-free
//      ds_read_b128), and adds the 80 sums into ITS OWN row of the workspace.
// Waves are persistent (grid.x = min(tiles, 2048)) and take tiles w, w+P, ... in order, so every
// workspace row is accumulated in a fixed order; a second pass sums the rows in fixed order:
// deterministic, no float atomics.
// =============================================================================================
constexpr int SEG = 16;                      // steps per checkpoint segment
// Reduction tile: 80 rows x 64 lanes, NO padding (20480 B = exactly 1/8 of a CU's LDS, so 8 waves
// = 2 per SIMD are resident; with a padded pitch of 68 it was 21760 B -> 7 per CU, SIMD load
// 2:2:2:1).  Conflict-free row reads come from an XOR swizzle of the 16-B slot index instead:
// element (row, lane) lives in slot (lane/4) ^ (row & 15).
constexpr int RED_PITCH = WAVE;
constexpr int64_t K2B_MAX_WAVES = 256 * 8;   // resident waves: 8 per CU
__device__ __forceinline__ int red_idx(int row, int l)
{
    return row * RED_PITCH + ((((l >> 2) ^ (row & 15)) << 2) | (l & 3));
}

template <typename T>
struct FusedBwdArgs {
    const T* Mck;                    // (nT/SEG, N*nM, 3)
    const T* rf;  int64_t rf_sn;
    const T* gr;  int64_t gr_sn;
    const T* loc;
    Bc df, gam;
    const T* b1;                     // (N, nM, 2) or null
    Bc g, E1, E2;
    const void* E1m1;
    const T* gMo;
    T* gMi;                          // may be null
    T* work;                         // (P, N, 5, nT)
    int64_t N, nM, nT, P;
};

template <typename T, typename CT, bool RELAX, bool HB1 = true>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_bwd(FusedBwdArgs<T> a)
{
    __shared__ __attribute__((aligned(16))) T red[5 * SEG * RED_PITCH];
    const int lane = threadIdx.x;
    const int64_t w = blockIdx.x, n = blockIdx.y;
    const int64_t nT = a.nT, rows = a.N * a.nM;
    const int64_t ntiles = (a.nM + WAVE - 1) / WAVE;
    // read-only, wave-uniform pulse through the constant address space: scalar loads (see K2)
    using CP = const T __attribute__((address_space(4)))*;
    CP rfr = (CP)(a.rf + n * a.rf_sn);
    CP rfi = rfr + nT;
    CP gx = (CP)(a.gr + n * a.gr_sn);
    CP gy = gx + nT;
    CP gz = gy + nT;
    T* wsrow = a.work + ((w * a.N + n) * 5) * nT;
    bool first = true;

    for (int64_t tile = w; tile < ntiles; tile += a.P) {
        const int64_t s_ = tile * WAVE + lane;
        const bool valid = s_ < a.nM;
        const int64_t s = valid ? s_ : a.nM - 1;
        const int64_t row = n * a.nM + s;
        const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
        const T lx = a.loc[row * 3], ly = a.loc[row * 3 + 1], lz = a.loc[row * 3 + 2];
        T delta = T(0);
        if (a.df.p) delta = bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s);
        T br = T(1), bi = T(0);
        if (HB1 && a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }
        const T vmask = valid ? T(1) : T(0);
        T hx = a.gMo[row * 3], hy = a.gMo[row * 3 + 1], hz = a.gMo[row * 3 + 2];
        adj_begin<RELAX, T, CT>(k, hx, hy, hz);

        auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
            Bx = T(0); By = T(0);
            if (HB1) field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
            else     { Bx = rfr[t]; By = rfi[t]; }               // no b1 map: Bxy = rf (as K2 / K0)
            Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
        };

        // Two global round trips per segment used to sit on the critical path: the checkpoint (used
        // at once by the recompute) and the read-modify-write of the workspace rows.  Both are now
        // issued a segment's worth of work ahead: the next checkpoint at the top of the current
        // segment, the old workspace values before the sweep that produces what is added to them.
        const int64_t nseg = nT / SEG;
        T cx = T(0), cy = T(0), cz = T(0);
        if (nseg > 0) {
            const T* ck = a.Mck + ((nseg - 1) * rows + row) * 3;
            cx = ck[0]; cy = ck[1]; cz = ck[2];
        }
        const int r1 = WAVE + lane;                        // this lane's second row, if < 5 * SEG
        for (int64_t seg = nseg - 1; seg >= 0; --seg) {
            const int64_t t0 = seg * SEG;
            T mx = cx, my = cy, mz = cz;
            if (seg > 0) {
                const T* ck = a.Mck + ((seg - 1) * rows + row) * 3;
                cx = ck[0]; cy = ck[1]; cz = ck[2];
            }
            T* dst0 = wsrow + (lane / SEG) * nT + t0 + (lane % SEG);
            T* dst1 = wsrow + (r1 / SEG) * nT + t0 + (r1 % SEG);
            T old0 = T(0), old1 = T(0);
            if (!first) { old0 = *dst0; if (r1 < 5 * SEG) old1 = *dst1; }
            // 1. forward recompute, keeping the state before each step
            T M0[SEG], M1[SEG], M2[SEG], Sv[SEG], Cv[SEG];
#pragma unroll
            for (int sb = 0; sb < SEG / 4; ++sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                Rot<T> r[4];
                rot_prepare<T, CT, 4>(k, Bx, By, Bz, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    M0[sb * 4 + j] = mx; M1[sb * 4 + j] = my; M2[sb * 4 + j] = mz;
                    Sv[sb * 4 + j] = r[j].S; Cv[sb * 4 + j] = r[j].C;     // reused by the sweep
                    rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
                }
            }
            // 2. adjoint sweep, contributions to LDS
#pragma unroll
            for (int sb = SEG / 4 - 1; sb >= 0; --sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                RotAdj<T> ra[4];
                const T S4[4] = {Sv[sb * 4], Sv[sb * 4 + 1], Sv[sb * 4 + 2], Sv[sb * 4 + 3]};
                const T C4[4] = {Cv[sb * 4], Cv[sb * 4 + 1], Cv[sb * 4 + 2], Cv[sb * 4 + 3]};
                rot_prepare_adj_given<T, CT, 4>(k, Bx, By, Bz, S4, C4, ra);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int st = sb * 4 + j;
                    T g0, g1, g2;
                    rot_apply_adj<RELAX, T, CT>(k, ra[j], M0[st], M1[st], M2[st], hx, hy, hz,
                                                g0, g1, g2);
                    g0 *= vmask; g1 *= vmask; g2 *= vmask;
                    red[red_idx(0 * SEG + st, lane)] = lx * g2;
                    red[red_idx(1 * SEG + st, lane)] = ly * g2;
                    red[red_idx(2 * SEG + st, lane)] = lz * g2;
                    red[red_idx(3 * SEG + st, lane)] = HB1 ? br * g0 + bi * g1 : g0;
                    red[red_idx(4 * SEG + st, lane)] = HB1 ? br * g1 - bi * g0 : g1;
                }
            }
            __syncthreads();
            // 3. 80 row sums: lanes 0..63 take rows 0..63, lanes 0..15 rows 64..79
            // (the old workspace values and the checkpoint were requested a segment ago: ONE explicit wait for
            // all vector loads here, or the compiler -- which loses count of them across the loops in between --
            // puts s_waitcnt vmcnt(0) in front of EACH store below, and every store then waits for the one before)
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0) (gfx9 encoding; expcnt / lgkmcnt untouched)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int rrow = pass * WAVE + lane;
                if (rrow < 5 * SEG) {
                    T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);  // 4 chains for ILP; fixed order
#pragma unroll
                    for (int i = 0; i < WAVE; i += 4) {            // logical lanes i..i+3: one slot
                        const T* q = red + red_idx(rrow, i);
                        p0 += q[0]; p1 += q[1]; p2 += q[2]; p3 += q[3];
                    }
                    const T acc = (p0 + p1) + (p2 + p3);
                    if (pass == 0) *dst0 = old0 + acc;             // old = 0 on the wave's first tile
                    else           *dst1 = old1 + acc;
                }
            }
            __syncthreads();
        }
        adj_end<RELAX, T, CT>(k, hx, hy, hz);
        if (valid && a.gMi) { a.gMi[row * 3] = hx; a.gMi[row * 3 + 1] = hy; a.gMi[row * 3 + 2] = hz; }
        first = false;
    }
}

// Pass 2: sum the P workspace rows per (n, quantity, t) in a fixed order.  Block = 32 time points
// x 8 row groups (group g takes rows g, g+8, ...: 128-B coalesced reads per row), then the eight
// partial sums are combined through LDS in group order -- deterministic, and nT/32 * 5 blocks
// instead of nT/256 * 5 (40 blocks at nT = 2048 took 0.45 ms for 73 MB).
constexpr int P2_T = 32, P2_G = 8;
template <typename T>
__global__ __launch_bounds__(P2_T * P2_G) void k_bloch_rfgr_bwd_p2(const T* work, T* grf, T* ggr,
                                                                   int64_t N, int64_t nT, int64_t P)
{
    __shared__ T part[P2_G][P2_T];
    const int tl = threadIdx.x % P2_T, g = threadIdx.x / P2_T;
    const int64_t t = (int64_t)blockIdx.x * P2_T + tl;
    const int64_t q = blockIdx.y, n = blockIdx.z;
    T acc = T(0);
    if (t < nT)
        for (int64_t w = g; w < P; w += P2_G) acc += work[((w * N + n) * 5 + q) * nT + t];
    part[g][tl] = acc;
    __syncthreads();
    if (g != 0 || t >= nT) return;
    T sum = part[0][tl];
#pragma unroll
    for (int i = 1; i < P2_G; ++i) sum += part[i][tl];
    if (q < 3) { if (ggr) ggr[(n * 3 + q) * nT + t] = sum; }
    else if (grf) grf[(n * 2 + (q - 3)) * nT + t] = sum;
}

// =============================================================================================
// K2b for parallel transmit: nC <= K2B_MAXC coils, rf (N|1, 2, nT, nC), b1 (N, nM, 2, nC).
// Same sweep as the single-coil kernel.  The per-coil sums over the 64 spins of a tile
//     grad_rf_re[c][t] = sum_l b1r[c][l] gBx[t][l] + b1i[c][l] gBy[t][l]
//     grad_rf_im[c][t] = sum_l b1r[c][l] gBy[t][l] - b1i[c][l] gBx[t][l]
// are small dot products: the raw gBx, gBy rows of the segment (2 x 16 rows) sit in the reduction
// tile next to the three loc*gBz rows, the tile's b1 in a second LDS array (2 nC rows x 64), and
// lane (step, re|im, half of the spins) forms its dot product per coil in spin order; the two
// halves are added in fixed order.  Workspace rows per wave: [gr_x, gr_y, gr_z, (re, im) x nC].
// =============================================================================================
constexpr int K2B_MAXC = 8;
constexpr int64_t K2B_MC_MAX_WAVES = 256 * 8;    // 18 KB of LDS per wave -> 8 per CU = 2 per SIMD

// MC: coil capacity of the build (2 / 4 / 8, the smallest that holds nC).  b1 registers, the staged rf
// samples and the coefficient rows are ZERO beyond nC, so that neither the field's coil loop nor the
// reduction's has a `c < nC` test in it (round 3: with the test every coil was its own basic block -- a
// wave-uniform branch and an exposed LDS round trip per coil, twice per step in the field alone; K0 and
// K2 had been rid of that in round 2).  Adding exact zeros changes nothing (at most the sign of a zero
// sum): the recomputed states stay those of K2's forward.
template <typename T, typename CT, bool RELAX, int MC>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_bwd_mc(FusedBwdArgs<T> a, int nC)
{
    constexpr int K2B_NCF = 2 * MC + 3;             // coefficient rows: b1r[c], b1i[c], loc x y z
    // raw dL/dB rows of one segment: [gBx | gBy | gBz][step][lane], slot-swizzled like `red`
    __shared__ __attribute__((aligned(16))) T raw[3 * SEG * RED_PITCH];
    // the tile's coefficients [b1r c0..7 | b1i c0..7 | loc x y z][lane], zero for lanes past nM
    __shared__ __attribute__((aligned(16))) T cfs[K2B_NCF * WAVE];
    __shared__ __attribute__((aligned(16))) T srf[2 * SEG * MC];             // [re|im][step][c], zero beyond nC
    const int lane = threadIdx.x;
    const int64_t w = blockIdx.x, n = blockIdx.y;
    const int64_t nT = a.nT, rows = a.N * a.nM;
    const int64_t ntiles = (a.nM + WAVE - 1) / WAVE;
    const int nQ = 3 + 2 * nC;
    const T* __restrict__ rfr = a.rf + n * a.rf_sn;            // [nT][nC]
    const T* __restrict__ rfi = rfr + nT * nC;
    using CP = const T __attribute__((address_space(4)))*;     // wave-uniform gradient samples: scalar loads
    CP gx = (CP)(a.gr + n * a.gr_sn);
    CP gy = gx + nT;
    CP gz = gy + nT;
    T* wsrow = a.work + ((w * a.N + n) * nQ) * nT;
    bool first = true;

    for (int64_t tile = w; tile < ntiles; tile += a.P) {
        const int64_t s_ = tile * WAVE + lane;
        const bool valid = s_ < a.nM;
        const int64_t s = valid ? s_ : a.nM - 1;
        const int64_t row = n * a.nM + s;
        const SpinConst<T, CT> k = load_consts<T, CT>(a.g, a.E1, a.E2, a.E1m1, n, s);
        const T lx = a.loc[row * 3], ly = a.loc[row * 3 + 1], lz = a.loc[row * 3 + 2];
        T delta = T(0);
        if (a.df.p) delta = bc_load<T>(a.df, n, s) / bc_load<T>(a.gam, n, s);
        T br[MC], bi[MC];
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            br[c] = (c < nC) ? a.b1[row * 2 * nC + c] : T(0);
            bi[c] = (c < nC) ? a.b1[row * 2 * nC + nC + c] : T(0);
        }
        const T vmask = valid ? T(1) : T(0);
        __syncthreads();                                   // previous tile's coefficients released
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            cfs[c * WAVE + lane] = br[c] * vmask;
            cfs[(MC + c) * WAVE + lane] = bi[c] * vmask;
        }
        cfs[(2 * MC + 0) * WAVE + lane] = lx * vmask;
        cfs[(2 * MC + 1) * WAVE + lane] = ly * vmask;
        cfs[(2 * MC + 2) * WAVE + lane] = lz * vmask;
        T hx = a.gMo[row * 3], hy = a.gMo[row * 3 + 1], hz = a.gMo[row * 3 + 2];
        adj_begin<RELAX, T, CT>(k, hx, hy, hz);

        int64_t tstage = 0;
        auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
            Bx = T(0); By = T(0);
            const T* qr = srf + (t - tstage) * MC;          // broadcast reads, batched: no test in the loop
            const T* qi = qr + SEG * MC;
#pragma unroll
            for (int c = 0; c < MC; ++c) field_xy_fma<T>(br[c], bi[c], qr[c], qi[c], Bx, By);
            Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
        };

        const int64_t nseg = nT / SEG;                      // checkpoint and workspace values are
        T cx = T(0), cy = T(0), cz = T(0);                  // fetched a segment ahead (see K2b)
        if (nseg > 0) {
            const T* ck = a.Mck + ((nseg - 1) * rows + row) * 3;
            cx = ck[0]; cy = ck[1]; cz = ck[2];
        }
        for (int64_t seg = nseg - 1; seg >= 0; --seg) {
            const int64_t t0 = seg * SEG;
            // the segment's rf samples (SEG * MC <= 128 floats per part) -> LDS; the barrier at the
            // end of the previous segment has released srf
            tstage = t0;
            for (int i = lane; i < SEG * MC; i += WAVE) {
                const int st_ = i / MC, c_ = i - st_ * MC;
                const bool on = c_ < nC;
                srf[i] = on ? rfr[(t0 + st_) * nC + c_] : T(0);
                srf[SEG * MC + i] = on ? rfi[(t0 + st_) * nC + c_] : T(0);
            }
            __syncthreads();
            T mx = cx, my = cy, mz = cz;
            if (seg > 0) {
                const T* ck = a.Mck + ((seg - 1) * rows + row) * 3;
                cx = ck[0]; cy = ck[1]; cz = ck[2];
            }
            // old workspace values of the rows this lane updates at the end of the segment
            const int st_w = lane >> 2, ri_w = (lane >> 1) & 1;
            const bool wr_w = (lane & 1) == 0;
            T* dst0 = wsrow + (3 + ri_w) * nT + t0 + st_w;        // + 2 c nT per coil
            T* dg0 = wsrow + ri_w * nT + t0 + st_w;               // grad_gr axis ri
            T* dg2 = wsrow + 2 * nT + t0 + st_w;                  // grad_gr axis z (ri == 0 lanes)
            T old[MC], oldg0 = T(0), oldg2 = T(0);
#pragma unroll
            for (int c = 0; c < MC; ++c)
                old[c] = (!first && wr_w && c < nC) ? dst0[2 * c * nT] : T(0);
            if (!first && wr_w) { oldg0 = *dg0; if (ri_w == 0) oldg2 = *dg2; }
            T M0[SEG], M1[SEG], M2[SEG], Sv[SEG], Cv[SEG];
#pragma unroll
            for (int sb = 0; sb < SEG / 4; ++sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                Rot<T> r[4];
                rot_prepare<T, CT, 4>(k, Bx, By, Bz, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    M0[sb * 4 + j] = mx; M1[sb * 4 + j] = my; M2[sb * 4 + j] = mz;
                    Sv[sb * 4 + j] = r[j].S; Cv[sb * 4 + j] = r[j].C;     // reused by the sweep
                    rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
                }
            }
#pragma unroll
            for (int sb = SEG / 4 - 1; sb >= 0; --sb) {
                T Bx[4], By[4], Bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) field(t0 + sb * 4 + j, Bx[j], By[j], Bz[j]);
                RotAdj<T> ra[4];
                const T S4[4] = {Sv[sb * 4], Sv[sb * 4 + 1], Sv[sb * 4 + 2], Sv[sb * 4 + 3]};
                const T C4[4] = {Cv[sb * 4], Cv[sb * 4 + 1], Cv[sb * 4 + 2], Cv[sb * 4 + 3]};
                rot_prepare_adj_given<T, CT, 4>(k, Bx, By, Bz, S4, C4, ra);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int st = sb * 4 + j;
                    T g0, g1, g2;
                    rot_apply_adj<RELAX, T, CT>(k, ra[j], M0[st], M1[st], M2[st], hx, hy, hz,
                                                g0, g1, g2);
                    raw[red_idx(0 * SEG + st, lane)] = g0;   // lanes past nM: zero coefficients
                    raw[red_idx(1 * SEG + st, lane)] = g1;
                    raw[red_idx(2 * SEG + st, lane)] = g2;
                }
            }
            __syncthreads();
            // All sums over the tile's spins are dot products of a raw row with coefficient rows:
            //   lane = (step, kind, half of the spins), kind 0: re, 1: im  -> per coil c
            //     re: b1r[c].gBx + b1i[c].gBy        im: b1r[c].gBy - b1i[c].gBx
            //   and for grad_gr lane = (step, axis i < 3, -, half), kind 2:  loc_i . gBz
            // spins outer, accumulators inner; halves added in fixed order; one load round trip
            // for the workspace update.
            {
                const int st = lane >> 2, ri = (lane >> 1) & 1, half = lane & 1;
                T acc[MC], accg[2];                 // accg: this lane's 1-2 grad_gr axes
#pragma unroll
                for (int c = 0; c < MC; ++c) acc[c] = T(0);
                accg[0] = accg[1] = T(0);
                // grad_gr: (st, ri, half) lanes take axis ri (0: x, 1: y); axis z rides on ri == 0
                const T* l0 = cfs + (2 * MC + ri) * WAVE;
                const T* l2 = cfs + (2 * MC + 2) * WAVE;
#pragma unroll 2
                for (int i = half * 32; i < half * 32 + 32; i += 4) {
                    const T* qx = raw + red_idx(0 * SEG + st, i);
                    const T* qy = raw + red_idx(1 * SEG + st, i);
                    const T* qz = raw + red_idx(2 * SEG + st, i);
                    T pp[4], qq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        pp[u] = ri == 0 ? qx[u] : qy[u];
                        qq[u] = ri == 0 ? qy[u] : -qx[u];
                        accg[0] += l0[i + u] * qz[u];
                        accg[1] += l2[i + u] * qz[u];
                    }
#pragma unroll
                    for (int c = 0; c < MC; ++c) {
                        const T* b_r = cfs + c * WAVE + i;
                        const T* b_i = cfs + (MC + c) * WAVE + i;
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[c] += b_r[u] * pp[u] + b_i[u] * qq[u];
                    }
                }
                const bool wr = half == 0;
                // one wait for the old workspace values (requested at the start of the segment) instead of a
                // compiler-inserted s_waitcnt vmcnt(0) in front of EVERY store below -- which made each of the
                // 2 nC + 3 stores wait for the one before it: 19 store round trips per segment at 8 coils
                __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) (gfx9 encoding; expcnt / lgkmcnt untouched)
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    const T other = __shfl_xor(acc[c], 1);
                    const T sum = half == 0 ? acc[c] + other : other + acc[c];
                    if (wr && c < nC) dst0[2 * c * nT] = old[c] + sum;   // old = 0 on the first tile
                }
                {
                    const T o0 = __shfl_xor(accg[0], 1), o2 = __shfl_xor(accg[1], 1);
                    const T s0 = half == 0 ? accg[0] + o0 : o0 + accg[0];
                    const T s2 = half == 0 ? accg[1] + o2 : o2 + accg[1];
                    if (wr) {
                        *dg0 = oldg0 + s0;
                        if (ri == 0) *dg2 = oldg2 + s2;
                    }
                }
            }
            __syncthreads();
        }
        adj_end<RELAX, T, CT>(k, hx, hy, hz);
        if (valid && a.gMi) { a.gMi[row * 3] = hx; a.gMi[row * 3 + 1] = hy; a.gMi[row * 3 + 2] = hz; }
        first = false;
    }
}

// Pass 2 for nQ = 3 + 2 nC quantities; grad_rf is (N, 2, nT, nC).
template <typename T>
__global__ __launch_bounds__(P2_T * P2_G) void k_bloch_rfgr_bwd_mc_p2(const T* work, T* grf, T* ggr,
                                                                      int64_t N, int64_t nT,
                                                                      int64_t P, int nC)
{
    __shared__ T part[P2_G][P2_T];
    const int tl = threadIdx.x % P2_T, g = threadIdx.x / P2_T;
    const int64_t t = (int64_t)blockIdx.x * P2_T + tl;
    const int64_t q = blockIdx.y, n = blockIdx.z;
    const int nQ = 3 + 2 * nC;
    T acc = T(0);
    if (t < nT)
        for (int64_t w = g; w < P; w += P2_G) acc += work[((w * N + n) * nQ + q) * nT + t];
    part[g][tl] = acc;
    __syncthreads();
    if (g != 0 || t >= nT) return;
    T sum = part[0][tl];
#pragma unroll
    for (int i = 1; i < P2_G; ++i) sum += part[i][tl];
    if (q < 3) { if (ggr) ggr[(n * 3 + q) * nT + t] = sum; }
    else if (grf) {
        const int64_t c = (q - 3) / 2, ri = (q - 3) % 2;
        grf[((n * 2 + ri) * nT + t) * nC + c] = sum;
    }
}
