// bloch_math.hpp -- per-spin Bloch step (rotation + relaxation) and its adjoint, device side.
//
// Physics restated from the reference (tianrluo/MRphy.py v0.2.0):
//   forward step   mrphy/sims.py:100-124   (rotation by -phi about u, then E-relaxation)
//   adjoint step   mrphy/sims.py:204-259
//   axis/angle     mrphy/beffective.py:35-36, Rodrigues mrphy/utils.py:351-357
//
// Formulation used here (algebraically identical, no axis normalisation, no 0/0 at phi = 0):
// with b = gamma*2*pi*dt * B  (rad), x = b.b = phi^2,
//     S(x) = sin(phi)/phi,   C(x) = (1 - cos(phi))/phi^2      (both entire functions of x)
//     m1   = m - S (b x m) + C (b x (b x m))
// which equals the reference's  m - sin(phi) (u x m) + (cos(phi)-1) (m - (u.m) u)  with
// u = b/phi, because b x (b x m) = b (b.m) - phi^2 m.  At phi = 0 the reference clamps phi to
// 1e-12 and gets u = 0, m1 = m (sims.py:101-102); here S = 1, C = 1/2, b = 0 give the same.
#pragma once
#include <hip/hip_runtime.h>

namespace mrphy {

// 16-byte vectors.  Native clang vectors, NOT HIP's float4/double2: those are struct/union
// wrappers that keep register arrays of them from being scalarised (they land in scratch).
typedef float  f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
// ...and the same vectors with element alignment only, for GLOBAL accesses: global_load/store
// dwordx4 need no more than dword alignment on this ISA, so rows that start at any element (pulse
// lengths with nT % 4 != 0) still move as 16-B vectors.  (LDS accesses keep the aligned types.)
typedef f32x4 f32x4_u __attribute__((aligned(4)));
typedef f64x2 f64x2_u __attribute__((aligned(8)));
template <typename T> struct V16;
template <> struct V16<float>  { using type = f32x4; using utype = f32x4_u; static constexpr int N = 4; };
template <> struct V16<double> { using type = f64x2; using utype = f64x2_u; static constexpr int N = 2; };

__device__ __forceinline__ float  fma_(float a, float b, float c)    { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }
// Compact float sincos for the cold large-angle path (a >= 0): quadrant reduction in fp64
// (k = rint(a*2/pi), r = a - k*pi/2 with a two-term pi/2: error < 1e-16*k, fine up to a ~ 1e9 rad),
// then the classic single-precision kernels on [-pi/4, pi/4] (Cephes sinf/cosf coefficients,
// < 1 ulp there).  Inline and ~30 instructions: ocml's sincosf would bring its Payne-Hanek branch
// and ~60 VGPRs into every unrolled step.
__device__ __forceinline__ void sincos_(float a, float* s, float* c)
{
#pragma clang fp contract(off)
    const double ad = (double)a;
    const double kd = rint(ad * 0.63661977236758134308);             // 2/pi
    double rd = fma(kd, -1.57079632679489655800, ad);                // pi/2 hi
    rd = fma(kd, -6.12323399573676603587e-17, rd);                   // pi/2 lo
    const float r = (float)rd, z = r * r;
    const int q = (int)((long long)kd & 3);
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sr = fmaf(ps * z, r, r);                             // sin r
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cr = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));          // cos r
    const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
}
// Compact double sincos for the same cold path (a >= 0, up to ~1e6 rad): quadrant reduction with a three-part
// pi/2 and FMAs (error ~ k 2^-110), then the classic double-precision kernels on [-pi/4, pi/4] (fdlibm's
// __kernel_sin / __kernel_cos coefficients, < 1 ulp there).  Round 4: ocml's sincos(double) brought its
// Payne-Hanek branch -- a private-memory table walk: 132 B of scratch per lane in every fp64 kernel that inlines
// the step, and some 60 VGPRs live across the cold branch -- into kernels whose arguments never exceed a few
// tens of radians per step.
__device__ __forceinline__ void sincos_(double a, double* s, double* c)
{
#pragma clang fp contract(off)
    const double kd = rint(a * 0.63661977236758134308);              // 2/pi
    double r = fma(kd, -1.57079632679489655800e+00, a);              // pi/2, first 53 bits
    r = fma(kd, -6.12323399573676603587e-17, r);                     // next 53
    r = fma(kd, 1.49738490485916983294e-33, r);                      // and the rest (pi/2 = hi + mid - this)
    const int q = (int)((long long)kd & 3);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sr = fma(ps * z, r, r);                             // sin r
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cr = fma(pc * z, z, fma(z, -0.5, 1.0));             // cos r
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ float  sqrt_(float a)  { return sqrtf(a); }
__device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
__device__ __forceinline__ float  tiny_(float)  { return 1e-30f; }
__device__ __forceinline__ double tiny_(double) { return 1e-300; }

// ---------------------------------------------------------------------------------------------
// Rotation coefficients.  General path: one sincos at the HALF angle h = phi/2,
//     sin(phi) = 2 sh ch,  1 - cos(phi) = 2 sh^2   =>   S = (sh/h) ch,  C = (sh/h)^2 / 2
// which has no cancellation for small phi (the reference's cos(phi) - 1 does).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void rot_coeffs_general(T x, T& S, T& C, T& cosphi)
{
#pragma clang fp contract(off)
    T h = T(0.5) * sqrt_(x);
    h = h > tiny_(T(0)) ? h : tiny_(T(0));
    T sh, ch;
    sincos_(h, &sh, &ch);
    const T q = sh / h;
    S = q * ch;
    C = (T(0.5) * q) * q;
    cosphi = fma_(T(-2) * sh, sh, T(1));
}

// Polynomial fast path (float only): S and C as polynomials in x = phi^2 on [0, X_POLY].
// Near-minimax fits (Remez exchange in fp64, tools/fit_poly.py): degree 6 / 5 with approximation
// error 7.4e-10 / 4.5e-9, below the rounding of the fp32 Horner sum itself (1.4e-7 / 3.9e-8
// absolute -- the same as the degree-9 Taylor polynomials they replace, which cost 18 FMAs).
// No transcendental, no sqrt, no division: 11 FMAs.  The kernels are VALU-issue bound
// (SQ_ACTIVE_INST_VALU ~ 99 % of the cycles at 46 instructions per spin-step), so every FMA
// removed here is time.
constexpr float X_POLY = 9.8696044f;   // pi^2: phi <= pi per step

__device__ __forceinline__ void rot_coeffs_poly(float x, float& S, float& C)
{
    float s = 1.361460111e-10f;
    s = fmaf(s, x, -2.472925686e-08f);
    s = fmaf(s, x,  2.753590024e-06f);
    s = fmaf(s, x, -1.984053670e-04f);
    s = fmaf(s, x,  8.333321661e-03f);
    s = fmaf(s, x, -1.666666567e-01f);
    S = fmaf(s, x, 1.0f);
    float c = -1.773506675e-09f;
    c = fmaf(c, x,  2.721793635e-07f);
    c = fmaf(c, x, -2.478447095e-05f);
    c = fmaf(c, x,  1.388849691e-03f);
    c = fmaf(c, x, -4.166663438e-02f);
    C = fmaf(c, x, 0.5f);
}

// Precise fp32 path: near-minimax degree-7 / degree-6 fits on [0, pi^2] (tools/fit_poly.py --fp64:
// approximation error 6.8e-12 / 4.7e-11) evaluated in fp64 from the fp32 argument and rounded
// once.  13 fp64 FMAs: about 2.5x the issue slots of the 11 fp32 FMAs of rot_coeffs_poly(float).
__device__ __forceinline__ void rot_coeffs_poly_precise(float xf, float& S, float& C)
{
    const double x = (double)xf;
    double s = -6.61101325761093948e-13;
    s = fma(s, x,  1.58967818563764548e-10);
    s = fma(s, x, -2.50387235830435598e-08);
    s = fma(s, x,  2.75567047781280402e-06);
    s = fma(s, x, -1.98412544903332901e-04);
    s = fma(s, x,  8.33333314478239967e-03);
    s = fma(s, x, -1.66666666578391104e-01);
    s = fma(s, x,  9.99999999993218314e-01);
    double c = 9.92981381380750462e-12;
    c = fma(c, x, -2.06726098894046996e-09);
    c = fma(c, x,  2.75437529915325051e-07);
    c = fma(c, x, -2.48011224326891956e-05);
    c = fma(c, x,  1.38888812875101854e-03);
    c = fma(c, x, -4.16666662001028004e-02);
    c = fma(c, x,  4.99999999953231744e-01);
    S = (float)s;
    C = (float)c;
}

// fp64 polynomial path: degree-12 interpolants at Chebyshev nodes on [0, pi^2], fitted in 50-digit
// arithmetic (tools/fit_poly64.py): approximation error < 1e-22, fp64 Horner error <= 2e-16.  The
// half-angle sincos form costs several hundred instructions per step in double and made the fp64
// kernels VALU-bound; it remains the path beyond pi^2.
// d = a * b + c as the THREE-address VOP3 instruction.  Left to itself the compiler selects the two-address
// v_fmac_f64 for these Horner chains and then, wherever its two-address pass does not convert them back, copies
// each coefficient (loop-invariant, in VGPRs: VOP3 has no 64-bit literal) into the accumulator's register first:
// 22 v_mov_b64 per 2-step batch in half the batches of the fp64 line kernels, 17 % of their VALU instructions
// (round 4, found in the ISA of k_bloch_fwd_lines_f64).
__device__ __forceinline__ double fma3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ void rot_coeffs_poly(double x, double& S, double& C)
{
    double s = 5.8830316106046321201e-26, c = 2.2776106052303377242e-27;
    s = fma3(s, x, -3.852402864994572414e-23);
    c = fma3(c, x, -1.6060843946810907021e-24);
    s = fma3(s, x, 1.9570355135893241057e-20);
    c = fma3(c, x, 8.8958637814668820037e-22);
    s = fma3(s, x, -8.2206078130557639829e-18);
    c = fma3(c, x, -4.1103077796065230472e-19);
    s = fma3(s, x, 2.8114570576322788928e-15);
    c = fma3(c, x, 1.5619206262606141008e-16);
    s = fma3(s, x, -7.6471637221312281391e-13);
    c = fma3(c, x, -4.7794773289098115937e-14);
    s = fma3(s, x, 1.6059043836494938496e-10);
    c = fma3(c, x, 1.147074559761245822e-11);
    s = fma3(s, x, -2.5052108385434337514e-8);
    c = fma3(c, x, -2.0876756987865449157e-9);
    s = fma3(s, x, 2.7557319223985783598e-6);
    c = fma3(c, x, 2.7557319223985852219e-7);
    s = fma3(s, x, -0.00019841269841269840346);
    c = fma3(c, x, -0.000024801587301587301256);
    s = fma3(s, x, 0.0083333333333333333292);
    c = fma3(c, x, 0.0013888888888888888887);
    s = fma3(s, x, -0.16666666666666666667);
    c = fma3(c, x, -0.041666666666666666667);
    s = fma(s, x, 1.0);
    c = fma(c, x, 0.5);
    S = s;
    C = c;
}

__device__ __forceinline__ void rot_dcoeffs_poly(double x, double& dS, double& dC)
{
    double s = -1.0963785217554045421e-27, c = -3.9380224385056999573e-29;
    s = fma3(s, x, 7.7090525825738789645e-25);
    c = fma3(c, x, 2.9663884481186015663e-26);
    s = fma3(s, x, -4.2545394390502402012e-22);
    c = fma3(c, x, -1.7727615028214082275e-23);
    s = fma3(s, x, 1.9572893549246436302e-19);
    c = fma3(c, x, 8.8967754892338863547e-21);
    s = fma3(s, x, -7.3985713811976039322e-17);
    c = fma3(c, x, -3.6992857469065027348e-18);
    s = fma3(s, x, 2.2491658017978594663e-14);
    c = fma3(c, x, 1.249536556924918865e-15);
    s = fma3(s, x, -5.3530146122172694503e-12);
    c = fma3(c, x, -3.3456341326522178321e-13);
    s = fma3(s, x, 9.6354263020916897487e-10);
    c = fma3(c, x, 6.882447358637406556e-11);
    s = fma3(s, x, -1.2526054192720840836e-7);
    c = fma3(c, x, -1.0438378493934043278e-8);
    s = fma3(s, x, 0.000011022927689594356101);
    c = fma3(c, x, 1.1022927689594356207e-6);
    s = fma3(s, x, -0.00059523809523809523802);
    c = fma3(c, x, -0.000074404761904761904759);
    s = fma3(s, x, 0.016666666666666666667);
    c = fma3(c, x, 0.0027777777777777777778);
    s = fma3(s, x, -0.16666666666666666667);
    c = fma3(c, x, -0.041666666666666666667);
    dS = s;
    dC = c;
}

// Which path a LANE takes depends only on its own x (x <= X_POLY: polynomial, else the half-angle
// sincos form), never on its wave neighbours, so results are independent of tiling and batching.
// The sincos form is evaluated only if some lane of the wave needs it (wave-uniform guard): it is
// cold for |B| <= pi/(gamma 2pi dt) ~ 29 G per step at dt = 4 us.
__device__ __forceinline__ void rot_coeffs_fixup(float x, float& S, float& C)
{
    float s, c, cp;
    rot_coeffs_general<float>(x, s, c, cp);
    if (x > X_POLY) { S = s; C = c; }
}

template <typename T>
__device__ __forceinline__ void rot_coeffs(T x, T& S, T& C);

template <>
__device__ __forceinline__ void rot_coeffs<float>(float x, float& S, float& C)
{
    rot_coeffs_poly(x, S, C);
    if (__builtin_amdgcn_ballot_w64(x > X_POLY) != 0ull) rot_coeffs_fixup(x, S, C);
}

template <>
__device__ __forceinline__ void rot_coeffs<double>(double x, double& S, double& C)
{
    rot_coeffs_poly(x, S, C);
    if (__builtin_amdgcn_ballot_w64(x > double(X_POLY)) != 0ull) {
        double s, c, cp;
        rot_coeffs_general<double>(x, s, c, cp);
        if (x > double(X_POLY)) { S = s; C = c; }
    }
}

// dS/dx and dC/dx for the adjoint.  Closed forms (cos(phi) - S)/(2x) and (S/2 - C)/x cancel for
// small x, so x < 1 uses the Taylor series  sum_{k>=1} (-1)^k k x^(k-1) / (2k+1)!  (resp. (2k+2)!).
template <typename T>
__device__ __forceinline__ void rot_coeffs_grad(T x, T& S, T& C, T& dS, T& dC)
{
#pragma clang fp contract(off)
    T cp;
    rot_coeffs_general<T>(x, S, C, cp);
    if (x < T(1)) {
        // k = 10 ... 1; k*x^(k-1)/(2k+1)! < 1e-18 at k = 10, x = 1 (enough for double).
        T ds = T( 10.0 / 51090942171709440000.0);                  //  10/21!
        ds = fma_(ds, x, T(-9.0  / 121645100408832000.0));             //  -9/19!
        ds = fma_(ds, x, T( 8.0  / 355687428096000.0));                //   8/17!
        ds = fma_(ds, x, T(-7.0  / 1307674368000.0));                  //  -7/15!
        ds = fma_(ds, x, T( 6.0  / 6227020800.0));                     //   6/13!
        ds = fma_(ds, x, T(-5.0  / 39916800.0));                       //  -5/11!
        ds = fma_(ds, x, T( 4.0  / 362880.0));                         //   4/9!
        ds = fma_(ds, x, T(-3.0  / 5040.0));                           //  -3/7!
        ds = fma_(ds, x, T( 2.0  / 120.0));                            //   2/5!
        ds = fma_(ds, x, T(-1.0  / 6.0));                              //  -1/3!
        T dc = T( 10.0 / 1124000727777607680000.0);                //  10/22!
        dc = fma_(dc, x, T(-9.0  / 2432902008176640000.0));            //  -9/20!
        dc = fma_(dc, x, T( 8.0  / 6402373705728000.0));               //   8/18!
        dc = fma_(dc, x, T(-7.0  / 20922789888000.0));                 //  -7/16!
        dc = fma_(dc, x, T( 6.0  / 87178291200.0));                    //   6/14!
        dc = fma_(dc, x, T(-5.0  / 479001600.0));                      //  -5/12!
        dc = fma_(dc, x, T( 4.0  / 3628800.0));                        //   4/10!
        dc = fma_(dc, x, T(-3.0  / 40320.0));                          //  -3/8!
        dc = fma_(dc, x, T( 2.0  / 720.0));                            //   2/6!
        dc = fma_(dc, x, T(-1.0  / 24.0));                             //  -1/4!
        dS = ds;
        dC = dc;
    } else {
        const T rx = T(1) / x;
        dS = (cp - S) * (T(0.5) * rx);
        dC = (T(0.5) * S - C) * rx;
    }
}

// ---------------------------------------------------------------------------------------------
// Field assembly (beffective.py:137-165), shared by K0 (which stores it) and K2 (which consumes
// it in registers) so that both round identically: explicit FMAs, no further contraction.
//   Bz = loc . gr + df/gamma        as  fma(gz,lz, fma(gy,ly, gx*lx)) + delta
//   Bx += b1r*rfr - b1i*rfi,  By += b1r*rfi + b1i*rfr      (one coil)
// ---------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ T field_z(T gx, T gy, T gz, T lx, T ly, T lz, T delta)
{
#pragma clang fp contract(off)
    const T dot = fma_(gz, lz, fma_(gy, ly, gx * lx));
    return dot + delta;
}

template <typename T>
__device__ __forceinline__ void field_xy_acc(T br, T bi, T rr, T ri, T& Bx, T& By)
{
#pragma clang fp contract(off)
    const T px = bi * ri, py = bi * rr;
    const T tx = fma_(br, rr, -px), ty = fma_(br, ri, py);
    Bx = Bx + tx;
    By = By + ty;
}

// Coil sum for parallel transmit (2 or more coils): the same products accumulated with FMA chains,
// 4 instructions per coil instead of 6.  Every multi-coil kernel (K0, K2, K2b) uses THIS form, so
// they agree bit for bit with each other; the one-coil form above is the one that is bit-identical
// to the reference.
template <typename T>
__device__ __forceinline__ void field_xy_fma(T br, T bi, T rr, T ri, T& Bx, T& By)
{
#pragma clang fp contract(off)
    Bx = fma_(br, rr, fma_(-bi, ri, Bx));
    By = fma_(br, ri, fma_(bi, rr, By));
}

// ---------------------------------------------------------------------------------------------
// Precision of the fp32 step, carried in the constant-type parameter CT of every kernel:
//   float / double      "fast": the fp32 step as in round 1 (degree-6/5 fp32 polynomials for S, C;
//                       m <- E (m - S w + C v) with three fp32 roundings per component)
//   prec_f32 / prec_f64 "precise" (constants in memory as float / double): S, C from degree-7/6
//                       polynomials evaluated in fp64 and rounded once (<= 0.5 ulp instead of the
//                       1.4e-7 / 3.9e-8 Horner error, which enters the state multiplied by phi and
//                       phi^2), and the rounding errors of the last accumulation and of the
//                       relaxation product carried into the result (error-free transformations with
//                       FMAs).  Measured on the headline workload (128^3 x 4096, phi up to 2.6 rad per
//                       step), relative L2 error against exact arithmetic on the same fp32 field and
//                       constants: fast 2.0e-5 (the reference's own fp32 runs: 2.6e-5 / 2.9e-5),
//                       precise 4.8e-6 (tools/precision_emul.py reproduces both on the CPU).
// ---------------------------------------------------------------------------------------------
struct prec_f32 {};
struct prec_f64 {};
template <typename CT> struct CTr { using mem = CT; using reg = CT; static constexpr bool precise = false; };
template <> struct CTr<prec_f32> { using mem = float;  using reg = float;  static constexpr bool precise = true; };
template <> struct CTr<prec_f64> { using mem = double; using reg = double; static constexpr bool precise = true; };


// ---------------------------------------------------------------------------------------------
// Per-spin constants, as the host computed them in the reference's dtype (sims.py:62,74-76).
// CT may be wider than T (fp32 data with the reference's fp64 default gamma/dt): products with
// constants are then formed in CT and rounded to T, which is what ATen's type promotion does
// for `torch.mul(g, Beff, out=fp32)` and `m1.mul_(E)` (sims.py:64,77).
// ---------------------------------------------------------------------------------------------
template <typename T, typename CT>
struct SpinConst {
    using R = typename CTr<CT>::reg;
    R g;            // gamma * 2 pi * dt  [rad/Gauss]
    R e1, e2, e1m1; // exp(-dt/T1), exp(-dt/T2), E1 - 1 (as the host formed it)
    R d1, d2;       // e1 - 1, e2 - 1 formed here (exact): the precise step's relaxation increments
    bool relax;
};

template <typename T, typename CT>
__device__ __forceinline__ void scale_b(const SpinConst<T, CT>& k, T Bx, T By, T Bz,
                                        T& bx, T& by, T& bz)
{
    using R = typename CTr<CT>::reg;
    bx = T(R(Bx) * k.g);
    by = T(R(By) * k.g);
    bz = T(R(Bz) * k.g);
}

// The step and its adjoint are written with explicit FMAs and contraction OFF, so that every
// kernel that inlines them (K1, K2, 1step; K3) performs bit-identical arithmetic: the fused kernel
// equals rfgr2beff + blochsim bit for bit, and results do not depend on how hipcc happens to
// contract a*b+c in one instantiation or another.
template <typename T>
__device__ __forceinline__ void cross_(T ax, T ay, T az, T bx, T by, T bz, T& cx, T& cy, T& cz)
{
#pragma clang fp contract(off)
    cx = fma_(ay, bz, -(az * by));
    cy = fma_(az, bx, -(ax * bz));
    cz = fma_(ax, by, -(ay * bx));
}

template <typename T>
__device__ __forceinline__ T dot_(T ax, T ay, T az, T bx, T by, T bz)
{
#pragma clang fp contract(off)
    return fma_(az, bz, fma_(ay, by, ax * bx));
}

// ---------------------------------------------------------------------------------------------
// Forward step in two halves.
//   rot_prepare: everything that does not depend on M -- b = g*B, x = b.b, S(x), C(x).  A kernel
//                calls it for a whole batch of steps at once: the polynomial chains of different
//                steps are independent, which is where the instruction-level parallelism is.
//   rot_apply:   the short M-dependent chain -- w = b x m, v = b x w, m += -S w + C v, relax.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Rot {
    T bx, by, bz, S, C;
};

template <typename T, typename CT, int NS>
__device__ __forceinline__ void rot_prepare(const SpinConst<T, CT>& k, const T (&Bx)[NS],
                                            const T (&By)[NS], const T (&Bz)[NS], Rot<T> (&r)[NS])
{
#pragma clang fp contract(off)
    if constexpr (sizeof(T) == 4) {
        T x[NS];
        bool big = false;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            scale_b<T, CT>(k, Bx[j], By[j], Bz[j], r[j].bx, r[j].by, r[j].bz);
            x[j] = dot_(r[j].bx, r[j].by, r[j].bz, r[j].bx, r[j].by, r[j].bz);
            if constexpr (CTr<CT>::precise) rot_coeffs_poly_precise(x[j], r[j].S, r[j].C);
            else                            rot_coeffs_poly(x[j], r[j].S, r[j].C);
            big = big || (x[j] > X_POLY);
        }
        if (__builtin_amdgcn_ballot_w64(big) != 0ull) {            // cold
#pragma unroll
            for (int j = 0; j < NS; ++j)
                if (__builtin_amdgcn_ballot_w64(x[j] > X_POLY) != 0ull)
                    rot_coeffs_fixup(x[j], r[j].S, r[j].C);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            scale_b<T, CT>(k, Bx[j], By[j], Bz[j], r[j].bx, r[j].by, r[j].bz);
            const T x = dot_(r[j].bx, r[j].by, r[j].bz, r[j].bx, r[j].by, r[j].bz);
            rot_coeffs<T>(x, r[j].S, r[j].C);
        }
    }
}

// One component of the precise update  m <- E (m - S w + C v) [- off]:
//   a = m - S w;  s = a + C v;   es = (a - s) + C v      rounding error of s (a - s is exact
//                                                         unless s cancels; |C v| reaches 2|m|)
//   r = s (E - 1) [- off]                                 the relaxation INCREMENT: |r| <= 1e-4 |s|,
//                                                         so its own rounding is ~1e-12
//   m = s + (es + r)                                      one rounding of the whole expression
//                                                         (es + r formed by one fma: fma(s, E-1, es [- off]))
// E - 1 is exact in the constants' type (E in [1/2, 1]).  Against the plain form -- three roundings
// (a, s, s E), a fourth on z (- off), where the sum p - off of a product of O(1) and an offset of
// O(1e-6) also drifts systematically: T1 recovery stalls below half an ulp per step -- this is one
// rounding of a and one of the result.  tools/precision_emul.py: 128^3 x 4096 on the headline
// pulse, relative L2 from exact arithmetic 2.0e-5 -> 4.8e-6 (seeded M0), 2.5e-5 -> 1.7e-6
// (M0 = z), together with the fp64-evaluated S, C.
template <bool RELAX, bool OFFSET, typename R>
__device__ __forceinline__ float update_precise(float m, float w, float v, float S, float C, R Em1, R off)
{
#pragma clang fp contract(off)
    const float a = fmaf(-S, w, m);
    const float s = fmaf(C, v, a);
    const float es = fmaf(C, v, a - s);
    if (!RELAX) return s + es;
    if constexpr (sizeof(R) == 4) {
        // es + r in ONE fma (round 3): r = s (E - 1) [- off] is never needed rounded on its own --
        // one instruction and one rounding fewer per component than  r = s*Em1;  es + r
        return s + fmaf(s, Em1, OFFSET ? es - off : es);
    } else {
        const float r = float(OFFSET ? fma(double(s), Em1, -off) : double(s) * Em1);
        return s + (es + r);
    }
}

template <bool RELAX, typename T, typename CT>
__device__ __forceinline__ void rot_apply(const SpinConst<T, CT>& k, const Rot<T>& r,
                                          T& mx, T& my, T& mz)
{
#pragma clang fp contract(off)
    using R = typename CTr<CT>::reg;
    T wx, wy, wz, vx, vy, vz;
    cross_(r.bx, r.by, r.bz, mx, my, mz, wx, wy, wz);    // w = b x m
    cross_(r.bx, r.by, r.bz, wx, wy, wz, vx, vy, vz);    // v = b x w
    if constexpr (CTr<CT>::precise && sizeof(T) == 4) {
        mx = update_precise<RELAX, false, R>(mx, wx, vx, r.S, r.C, k.d2, R(0));
        my = update_precise<RELAX, false, R>(my, wy, vy, r.S, r.C, k.d2, R(0));
        mz = update_precise<RELAX, true, R>(mz, wz, vz, r.S, r.C, k.d1, k.e1m1);
        return;
    }
    mx = fma_(r.C, vx, fma_(-r.S, wx, mx));
    my = fma_(r.C, vy, fma_(-r.S, wy, my));
    mz = fma_(r.C, vz, fma_(-r.S, wz, mz));
    if (RELAX) {                                         // two roundings on z, as sims.py:77
        mx = T(R(mx) * k.e2);
        my = T(R(my) * k.e2);
        mz = T(R(mz) * k.e1);
        mz = T(R(mz) - k.e1m1);
    }
}

// Pin the state to a program point: an empty volatile asm that "rewrites" the three registers.  The chain
// that produces them must be complete before it, and -- having side effects as far as the compiler knows --
// it is neither sunk past the branch that follows (the wave-uniform guard of the NEXT batch's cold
// large-angle path) nor merged with its neighbours.  Without it the compiler sinks the rot_apply chains of a
// batch below the next batches' guards, keeping their Rot registers (5 per step) alive across them: the
// no-history K1 then needs 146 VGPRs where its history-saving twin -- whose stores pin the order the same
// way -- needs 98 (round 4; 88 with the pin, no scratch).  No instruction is emitted.
__device__ __forceinline__ void pin_state(float& a, float& b, float& c)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void pin_state(double& a, double& b, double& c)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
}

// One forward step: M <- relax(rotate(M, B))  (= rot_prepare<1> + rot_apply, same arithmetic).
template <typename T, typename CT>
__device__ __forceinline__ void bloch_step(const SpinConst<T, CT>& k, T Bx, T By, T Bz,
                                           T& mx, T& my, T& mz)
{
    const T bx_[1] = {Bx}, by_[1] = {By}, bz_[1] = {Bz};
    Rot<T> r[1];
    rot_prepare<T, CT, 1>(k, bx_, by_, bz_, r);
    if (k.relax) rot_apply<true, T, CT>(k, r[0], mx, my, mz);
    else         rot_apply<false, T, CT>(k, r[0], mx, my, mz);
}

// ---------------------------------------------------------------------------------------------
// Adjoint step, in the same two halves as the forward one.
//   L = ht.m1,  ht = E*h,  m1 = m - S w + C v,  w = b x m,  v = b x w
//   dL/db = -S (m x ht) + C[(b.m) ht + (b.ht) m - 2 (ht.m) b] + 2 b [ -S' (ht.w) + C' (ht.v) ]
//   dL/dm = ht + S (b x ht) + C (b x (b x ht))          (rotation by +phi)
// This is sims.py:204-259 with the -gamma*2*pi*dt pre-scaling of h (sims.py:194) folded out.
// S' = dS/dx, C' = dC/dx: degree-5 near-minimax polynomials on [0, X_POLY] (tools/fit_poly.py;
// abs error 2.8e-8 / 3.8e-9 in float), closed forms (cos(phi) - S)/(2x) and (S/2 - C)/x beyond,
// where they no longer cancel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot_dcoeffs_poly(float x, float& dS, float& dC)
{
    float s = 8.170263910e-10f;
    s = fmaf(s, x, -1.236781628e-07f);
    s = fmaf(s, x,  1.101494763e-05f);
    s = fmaf(s, x, -5.952198408e-04f);
    s = fmaf(s, x,  1.666665077e-02f);
    dS = fmaf(s, x, -1.666666716e-01f);
    float c = 5.958734201e-11f;
    c = fmaf(c, x, -1.033830355e-08f);
    c = fmaf(c, x,  1.101787234e-06f);
    c = fmaf(c, x, -7.440360059e-05f);
    c = fmaf(c, x,  2.777776914e-03f);
    dC = fmaf(c, x, -4.166666791e-02f);
}

template <typename T>
struct RotAdj {
    T bx, by, bz, x, S, C, dS, dC;     // x = b.b
};

template <typename T, typename CT, int NS>
__device__ __forceinline__ void rot_prepare_adj(const SpinConst<T, CT>& k, const T (&Bx)[NS],
                                                const T (&By)[NS], const T (&Bz)[NS],
                                                RotAdj<T> (&r)[NS])
{
#pragma clang fp contract(off)
    if constexpr (sizeof(T) == 4) {
        T x[NS];
        bool big = false;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            scale_b<T, CT>(k, Bx[j], By[j], Bz[j], r[j].bx, r[j].by, r[j].bz);
            x[j] = dot_(r[j].bx, r[j].by, r[j].bz, r[j].bx, r[j].by, r[j].bz);
            r[j].x = x[j];
            // S, C drive the propagation of h (errors accumulate over the sweep): precise mode takes
            // the fp64-evaluated, once-rounded values of the forward step.  S', C' only enter dL/dB
            // of their own step (nothing accumulates): fp32 polynomials in both modes.
            if constexpr (CTr<CT>::precise) rot_coeffs_poly_precise(x[j], r[j].S, r[j].C);
            else                            rot_coeffs_poly(x[j], r[j].S, r[j].C);
            rot_dcoeffs_poly(x[j], r[j].dS, r[j].dC);
            big = big || (x[j] > X_POLY);
        }
        if (__builtin_amdgcn_ballot_w64(big) != 0ull) {            // cold
#pragma unroll
            for (int j = 0; j < NS; ++j)
                if (__builtin_amdgcn_ballot_w64(x[j] > X_POLY) != 0ull) {
                    T S, C, cp;
                    rot_coeffs_general<T>(x[j], S, C, cp);
                    const T rx = T(1) / x[j];
                    if (x[j] > X_POLY) {
                        r[j].S = S; r[j].C = C;
                        r[j].dS = (cp - S) * (T(0.5) * rx);
                        r[j].dC = (T(0.5) * S - C) * rx;
                    }
                }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            scale_b<T, CT>(k, Bx[j], By[j], Bz[j], r[j].bx, r[j].by, r[j].bz);
            const T x = dot_(r[j].bx, r[j].by, r[j].bz, r[j].bx, r[j].by, r[j].bz);
            r[j].x = x;
            rot_coeffs_poly(x, r[j].S, r[j].C);
            rot_dcoeffs_poly(x, r[j].dS, r[j].dC);
            if (__builtin_amdgcn_ballot_w64(x > T(X_POLY)) != 0ull) {      // cold: the closed forms (no cancellation
                T S, C, cp;                                                // beyond pi^2; rot_coeffs_grad's x >= 1 branch)
                rot_coeffs_general<T>(x, S, C, cp);
                const T rx = T(1) / x;
                if (x > T(X_POLY)) {
                    r[j].S = S; r[j].C = C;
                    r[j].dS = (cp - S) * (T(0.5) * rx);
                    r[j].dC = (T(0.5) * S - C) * rx;
                }
            }
        }
    }
}

// rot_prepare_adj with S, C GIVEN: a kernel that has just run rot_prepare on the same fields (the
// fused adjoint recomputes the forward states of a segment first) already holds the step's S, C --
// the same values rot_prepare_adj would form (same x, same polynomial, same cold path) -- so the
// sweep only needs b, x and the derivatives.  In precise mode that spares the 13 fp64 FMAs + 3
// conversions per step a second time.  Bit-identical to rot_prepare_adj.
template <typename T, typename CT, int NS>
__device__ __forceinline__ void rot_prepare_adj_given(const SpinConst<T, CT>& k, const T (&Bx)[NS],
                                                      const T (&By)[NS], const T (&Bz)[NS],
                                                      const T (&S)[NS], const T (&C)[NS],
                                                      RotAdj<T> (&r)[NS])
{
#pragma clang fp contract(off)
    bool big = false;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        scale_b<T, CT>(k, Bx[j], By[j], Bz[j], r[j].bx, r[j].by, r[j].bz);
        r[j].x = dot_(r[j].bx, r[j].by, r[j].bz, r[j].bx, r[j].by, r[j].bz);
        r[j].S = S[j];
        r[j].C = C[j];
        rot_dcoeffs_poly(r[j].x, r[j].dS, r[j].dC);
        big = big || (r[j].x > T(X_POLY));
    }
    if (__builtin_amdgcn_ballot_w64(big) != 0ull) {                // cold: as rot_prepare_adj
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (__builtin_amdgcn_ballot_w64(r[j].x > T(X_POLY)) != 0ull) {
                if constexpr (sizeof(T) == 4) {
                    T Sg, Cg, cp;
                    rot_coeffs_general<T>(r[j].x, Sg, Cg, cp);
                    const T rx = T(1) / r[j].x;
                    if (r[j].x > X_POLY) {
                        r[j].dS = (cp - Sg) * (T(0.5) * rx);
                        r[j].dC = (T(0.5) * Sg - Cg) * rx;
                    }
                } else {
                    T Sg, Cg, cp;
                    rot_coeffs_general<T>(r[j].x, Sg, Cg, cp);
                    const T rx = T(1) / r[j].x;
                    if (r[j].x > T(X_POLY)) {
                        r[j].dS = (cp - Sg) * (T(0.5) * rx);
                        r[j].dC = (T(0.5) * Sg - Cg) * rx;
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// What the adjoint sweep carries from step to step.
//   plain modes (fast fp32, fp64): h = dL/dM_after; every step forms ht = E h, then h <- R^T ht.
//   precise fp32 mode (round 3):   the carried state IS ht ("t-state").  Its recursion
//       t <- E (R^T t)
//   has the forward step's "rotate, then relax" shape (rotation by +phi, no offset), so it takes
//   the forward's compensated update (update_precise): one rounding of `a` and one of the result
//   per component and step, instead of the five of  E h, cos(phi) t, + S p, + C (b.t) b.  The
//   incoming cotangent is scaled by E once (adj_begin) and the outgoing one divided by E once
//   (adj_end); both are exact to half an ulp and nothing accumulates.  (E must not be zero, i.e.
//   T2 > dt / 100 in fp32 -- as in the reference, whose adjoint divides by E too, sims.py:174-177.)
// Measured on MI355X (tools/grad_parity.py, config 5 = 64^3 x 2048, all spins, against fp64
// differentiation of the same function on the same fp32 field and constants): with the fp32
// S, C and the plain update grad_M0 was 1.24e-5 from exact (2.2e-5 at nT = 4096) -- the fast
// forward step's error level; see docs/LABNOTES.md ("The precise adjoint") for the figures of this form.
// ---------------------------------------------------------------------------------------------
template <typename T, typename CT>
struct AdjMode { static constexpr bool tstate = CTr<CT>::precise && sizeof(T) == 4; };

template <bool RELAX, typename T, typename CT>
__device__ __forceinline__ void adj_begin(const SpinConst<T, CT>& k, T& hx, T& hy, T& hz)
{
#pragma clang fp contract(off)
    using R = typename CTr<CT>::reg;
    if constexpr (AdjMode<T, CT>::tstate && RELAX) {
        hx = T(R(hx) * k.e2);
        hy = T(R(hy) * k.e2);
        hz = T(R(hz) * k.e1);
    }
}

template <bool RELAX, typename T, typename CT>
__device__ __forceinline__ void adj_end(const SpinConst<T, CT>& k, T& hx, T& hy, T& hz)
{
#pragma clang fp contract(off)
    using R = typename CTr<CT>::reg;
    if constexpr (AdjMode<T, CT>::tstate && RELAX) {
        hx = T(R(hx) / k.e2);
        hy = T(R(hy) / k.e2);
        hz = T(R(hz) / k.e1);
    }
}

// run-time form of the pair for kernels that branch on k.relax per step
template <typename T, typename CT>
__device__ __forceinline__ void adj_begin_rt(const SpinConst<T, CT>& k, T& hx, T& hy, T& hz)
{
    if (k.relax) adj_begin<true, T, CT>(k, hx, hy, hz);
}
template <typename T, typename CT>
__device__ __forceinline__ void adj_end_rt(const SpinConst<T, CT>& k, T& hx, T& hy, T& hz)
{
    if (k.relax) adj_end<true, T, CT>(k, hx, hy, hz);
}

// In: m = magnetisation BEFORE the step, h = the carried adjoint state after the step (above).
// Out: h <- the carried state before the step, g = dL/dB.
// (core: dL/db of the SCALED field b = g B comes back raw in (dbx, dby, dbz); rot_apply_adj below scales it to
// dL/dB, the GC builds also feed it to adj_const_accumulate)
template <bool RELAX, typename T, typename CT>
__device__ __forceinline__ void rot_apply_adj_core(const SpinConst<T, CT>& k, const RotAdj<T>& r,
                                                   T mx, T my, T mz, T& hx, T& hy, T& hz,
                                                   T& dbx, T& dby, T& dbz)
{
#pragma clang fp contract(off)
    using R = typename CTr<CT>::reg;
    const T bx = r.bx, by = r.by, bz = r.bz, S = r.S, C = r.C;
    T tx = hx, ty = hy, tz = hz;
    if constexpr (!AdjMode<T, CT>::tstate) {
        if (RELAX) {
            tx = T(R(hx) * k.e2);
            ty = T(R(hy) * k.e2);
            tz = T(R(hz) * k.e1);
        }
    }
    // w = b x m and v = b x w never need forming:  ht.w = b.(m x ht),  ht.v = (b.ht)(b.m) - x (ht.m),
    // b x (b x ht) = (b.ht) b - x ht  (14 of 77 instructions less than the literal form)
    T cx, cy, cz;
    cross_(mx, my, mz, tx, ty, tz, cx, cy, cz);          // c = m x ht
    const T bm = dot_(bx, by, bz, mx, my, mz);
    const T bt = dot_(bx, by, bz, tx, ty, tz);
    const T tm = dot_(tx, ty, tz, mx, my, mz);
    const T tw = dot_(bx, by, bz, cx, cy, cz);           // ht.w
    const T tv = fma_(bt, bm, -(r.x * tm));              // ht.v
    const T kb = T(2) * (fma_(r.dC, tv, -(r.dS * tw)) - C * tm);           // coefficient of b
    dbx = fma_(kb, bx, fma_(C, fma_(bm, tx, bt * mx), -(S * cx)));
    dby = fma_(kb, by, fma_(C, fma_(bm, ty, bt * my), -(S * cy)));
    dbz = fma_(kb, bz, fma_(C, fma_(bm, tz, bt * mz), -(S * cz)));
    if constexpr (AdjMode<T, CT>::tstate) {
        // t <- E (t + S (b x t) + C (b x (b x t))), compensated as the forward update:
        // w' = t x b = -(b x t), v = w' x b = b x (b x t);  a = t - S w', s = a + C v, ...
        T wx, wy, wz, vx, vy, vz;
        cross_(tx, ty, tz, bx, by, bz, wx, wy, wz);
        cross_(wx, wy, wz, bx, by, bz, vx, vy, vz);
        hx = update_precise<RELAX, false, R>(tx, wx, vx, S, C, k.d2, R(0));
        hy = update_precise<RELAX, false, R>(ty, wy, vy, S, C, k.d2, R(0));
        hz = update_precise<RELAX, false, R>(tz, wz, vz, S, C, k.d1, R(0));
        return;
    }
    T px, py, pz;                                        // h0 = cos(phi) ht + S (b x ht) + C (b.ht) b
    cross_(bx, by, bz, tx, ty, tz, px, py, pz);
    const T cph = fma_(-C, r.x, T(1));                   // cos(phi) = 1 - C x
    const T cbt = C * bt;
    hx = fma_(cbt, bx, fma_(S, px, cph * tx));
    hy = fma_(cbt, by, fma_(S, py, cph * ty));
    hz = fma_(cbt, bz, fma_(S, pz, cph * tz));
}

template <bool RELAX, typename T, typename CT>
__device__ __forceinline__ void rot_apply_adj(const SpinConst<T, CT>& k, const RotAdj<T>& r,
                                              T mx, T my, T mz, T& hx, T& hy, T& hz,
                                              T& gx, T& gy, T& gz)
{
#pragma clang fp contract(off)
    using R = typename CTr<CT>::reg;
    T dbx, dby, dbz;
    rot_apply_adj_core<RELAX, T, CT>(k, r, mx, my, mz, hx, hy, hz, dbx, dby, dbz);
    gx = T(R(dbx) * k.g);
    gy = T(R(dby) * k.g);
    gz = T(R(dbz) * k.g);
}

// ---------------------------------------------------------------------------------------------
// Gradients w.r.t. the per-spin constants (round 3) -- what autograd through the reference's plain
// torch ops hands a caller of slowsims.blochsim / blochsim_1step who differentiates w.r.t. T1, T2,
// gamma, dt (slowsims.py:86-112, 42-51).  With u = R m (rotated, before relaxation) and
// m' = (E2 ux, E2 uy, E1 uz - E1m1), b = g B, summed over the steps:
//     dL/dE2 = hx ux + hy uy,   dL/dE1 = hz uz,   dL/dE1m1 = -hz,   dL/dg = dL/db . B
// (dL/db raw, from rot_apply_adj_core: round 4 -- round 3 recovered it as (dL/dB . B) / g, which is 0/0 for a spin
// with g = 0, e.g. a gamma = 0 padding spin, and poisoned the summed gradient)
// where h = dL/dm'.  `s` is the state the sweep carries INTO this step's adjoint: h itself in the
// plain modes, t = E h in the precise fp32 mode -- adj_const_finish divides that E out once.
// acc = [dL/dg, dL/dE1, dL/dE2, dL/dE1m1].
// ---------------------------------------------------------------------------------------------
template <bool RELAX, typename T, typename CT>
__device__ __forceinline__ void adj_const_accumulate(const RotAdj<T>& r, T Bx, T By, T Bz,
                                                     T mx, T my, T mz, T sx, T sy, T sz,
                                                     T dbx, T dby, T dbz, T (&acc)[4], bool offset = true)
{
#pragma clang fp contract(off)
    acc[0] = fma_(dbz, Bz, fma_(dby, By, fma_(dbx, Bx, acc[0])));
    if (RELAX) {
        T wx, wy, wz, vx, vy, vz;
        cross_(r.bx, r.by, r.bz, mx, my, mz, wx, wy, wz);
        cross_(r.bx, r.by, r.bz, wx, wy, wz, vx, vy, vz);
        const T ux = fma_(r.C, vx, fma_(-r.S, wx, mx));
        const T uy = fma_(r.C, vy, fma_(-r.S, wy, my));
        const T uz = fma_(r.C, vz, fma_(-r.S, wz, mz));
        acc[1] = fma_(sz, uz, acc[1]);
        acc[2] = fma_(sy, uy, fma_(sx, ux, acc[2]));
        if (offset) acc[3] = acc[3] - sz;          // (beff2ab: the offset acts on the B column only)
    }
}

template <typename T, typename CT>
__device__ __forceinline__ void adj_const_finish(const SpinConst<T, CT>& k, T (&acc)[4])
{
    using R = typename CTr<CT>::reg;
    if (k.relax && AdjMode<T, CT>::tstate) {
        acc[1] = T(R(acc[1]) / k.e1);
        acc[2] = T(R(acc[2]) / k.e2);
        acc[3] = T(R(acc[3]) / k.e1);
    }
}

}  // namespace mrphy
