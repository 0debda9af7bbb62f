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
template <typename T, typename CT, bool NC1, bool CK, bool RELAX>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_fwd(FusedArgs<T> a)
{
    constexpr int NS = 8;
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
    if (NC1 && a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }

    const int64_t nT = a.nT, nC = a.nC;
    const T* __restrict__ rfr = a.rf + n * a.rf_sn;          // [nT][nC]
    const T* __restrict__ rfi = rfr + nT * nC;
    const T* __restrict__ gx = a.gr + n * a.gr_sn;
    const T* __restrict__ gy = gx + nT;
    const T* __restrict__ gz = gy + nT;
    const T* b1 = a.b1 ? a.b1 + row * 2 * nC : nullptr;
    const int64_t rows = a.N * a.nM;

    auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
        Bx = T(0); By = T(0);
        if (NC1) {
            field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
        } else {
            for (int64_t c = 0; c < nC; ++c)
                field_xy_acc<T>(b1[c], b1[nC + c], rfr[t * nC + c], rfi[t * nC + c], Bx, By);
        }
        Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
    };

    int64_t t0 = 0;
    for (; t0 + NS <= nT; t0 += NS) {
        if (CK && (t0 % a.ck_every) == 0 && valid) {
            T* c = a.Mck + ((t0 / a.ck_every) * rows + row) * 3;
            c[0] = mx; c[1] = my; c[2] = mz;
        }
        T Bx[NS], By[NS], Bz[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) field(t0 + j, Bx[j], By[j], Bz[j]);
        Rot<T> r[NS];
        rot_prepare<T, CT, NS>(k, Bx, By, Bz, r);
#pragma unroll
        for (int j = 0; j < NS; ++j) rot_apply<RELAX, T, CT>(k, r[j], mx, my, mz);
    }
    for (; t0 < nT; ++t0) {                                   // nT % 8 tail
        if (CK && (t0 % a.ck_every) == 0 && valid) {
            T* c = a.Mck + ((t0 / a.ck_every) * rows + row) * 3;
            c[0] = mx; c[1] = my; c[2] = mz;
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

template <typename T, typename CT, bool RELAX>
__global__ __launch_bounds__(WAVE) void k_bloch_rfgr_bwd(FusedBwdArgs<T> a)
{
    __shared__ __attribute__((aligned(16))) T red[5 * SEG * RED_PITCH];
    const int lane = threadIdx.x;
    const int64_t w = blockIdx.x, n = blockIdx.y;
    const int64_t nT = a.nT, rows = a.N * a.nM;
    const int64_t ntiles = (a.nM + WAVE - 1) / WAVE;
    const T* __restrict__ rfr = a.rf + n * a.rf_sn;
    const T* __restrict__ rfi = rfr + nT;
    const T* __restrict__ gx = a.gr + n * a.gr_sn;
    const T* __restrict__ gy = gx + nT;
    const T* __restrict__ gz = gy + nT;
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
        if (a.b1) { br = a.b1[row * 2]; bi = a.b1[row * 2 + 1]; }
        const T vmask = valid ? T(1) : T(0);
        T hx = a.gMo[row * 3], hy = a.gMo[row * 3 + 1], hz = a.gMo[row * 3 + 2];

        auto field = [&](int64_t t, T& Bx, T& By, T& Bz) {
            Bx = T(0); By = T(0);
            field_xy_acc<T>(br, bi, rfr[t], rfi[t], Bx, By);
            Bz = field_z<T>(gx[t], gy[t], gz[t], lx, ly, lz, delta);
        };

        for (int64_t seg = nT / SEG - 1; seg >= 0; --seg) {
            const int64_t t0 = seg * SEG;
            const T* ck = a.Mck + (seg * rows + row) * 3;
            T mx = ck[0], my = ck[1], mz = ck[2];
            // 1. forward recompute, keeping the state before each step
            T M0[SEG], M1[SEG], M2[SEG];
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
                rot_prepare_adj<T, CT, 4>(k, Bx, By, Bz, ra);
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
                    red[red_idx(3 * SEG + st, lane)] = br * g0 + bi * g1;
                    red[red_idx(4 * SEG + st, lane)] = br * g1 - bi * g0;
                }
            }
            __syncthreads();
            // 3. 80 row sums: lanes 0..63 take rows 0..63, lanes 0..15 rows 64..79
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
                    T* dst = wsrow + (rrow / SEG) * nT + t0 + (rrow % SEG);
                    *dst = first ? acc : (*dst + acc);
                }
            }
            __syncthreads();
        }
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
