/* mrphy_hip.h -- C ABI of libmrphy_hip.so, the MI355X (gfx950) Bloch-simulation hot path.
 *
 * This is the drop-in boundary for ONE path of tianrluo/MRphy.py (reference v0.2.0):
 *
 *     mrphy.beffective.rfgr2beff   (reference mrphy/beffective.py:107-168)
 *     mrphy.sims.blochsim          (reference mrphy/sims.py:272-315; BlochSim.forward :32-132,
 *                                   BlochSim.backward :135-269)
 *     mrphy.slowsims.blochsim_1step(reference mrphy/slowsims.py:15-54)
 *
 * plus, in the order of SURVEY.md section 8(f), the callers and helpers either side of it:
 * sims.freeprec (sims.py:318-458), Pulse.interpT linear (mobjs.py:177-220), SpinArray.extract /
 * embed and SpinCube._update_loc_ (mobjs.py:512-553, 815-839), beffective.beff2ab and
 * slowsims.blochsim_ab (beffective.py:40-104, slowsims.py:117-131); and the fused rf,gr -> Mo
 * kernel with its adjoint (one transmit coil, or 2..8 with a b1 map), which is what
 * SpinArray.applypulse composes (mobjs.py:435-446).
 *
 * The reference has no FFI of its own (it is pure Python over ATen); the entry points below are
 * what a ctypes binding for those functions binds to.  INTEGRATION.md shows the reference-side
 * stub.  Conventions:
 *
 *   - plain pointers + sizes only; every pointer is a DEVICE address (hipMalloc'ed / a torch
 *     tensor's data_ptr()), never dereferenced on the host;
 *   - the library allocates nothing, keeps no state and is re-entrant (autograd calls the
 *     backward entry points from another thread); scratch space is passed in by the caller;
 *   - `stream` is a hipStream_t (0 = the null stream); launches are asynchronous, no host sync;
 *   - return value: 0 on success, otherwise a hipError_t value (mrphy_error_string() renders
 *     it) or one of the negative MRPHY_E* codes for argument errors caught on the host;
 *   - "spins" are rows r = n*nM + s of the compact layout (N batches x nM spins);
 *   - a *broadcastable per-spin constant* is passed as (ptr, stride_n, stride_m) in ELEMENTS:
 *     element (n, s) lives at ptr[n*stride_n + s*stride_m]; stride 0 broadcasts.  This covers
 *     the reference's "() | (N|1, nM|1)" shapes and torch's stride-0 expanded views
 *     (mobjs.SpinArray keeps T1_/T2_/gamma_ that way) without materialising them.
 *
 * dtype codes: data type T of M / Beff / rf / gr / loc and constant type CT of the
 * host-computed constants (gamma*2*pi*dt, E1, E2, E1-1):
 */
#ifndef MRPHY_HIP_H
#define MRPHY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* History of the contract.  A caller should check mrphy_abi_version() == MRPHY_ABI_VERSION.
 *   1  round 1: dtype codes 0..2.
 *   2  round 2-3: dtype codes 3 and 4 (precise fp32 step) accepted by every integrating entry
 *      point; mrphy_rfgr2beff_bwd_workspace sizes the one-pass layout for 2..32 coils (round 3:
 *      partial sums + packed per-spin coefficient rows; callers that ask the query see no change);
 *      mrphy_rfgr2beff_bwd returns MRPHY_EINVAL for nC >= 2 without a b1 map; under codes 3 / 4 the
 *      adjoint entry points (blochsim_bwd, blochsim_rfgr_*bwd, beff2ab_bwd) carry the adjoint state
 *      with the compensated update as well (round 3) -- same arguments, different (better) bits.
 *      No environment variable changes what the shipped library runs (the development knobs of
 *      rounds 1-2 exist only in the -DMRPHY_DEV_KNOBS build of tools/).
 *   3  round 4: mrphy_freeprec_bwd_consts and mrphy_beff2ab_bwd_consts (gradients w.r.t. the constants that the
 *      reference's autograd supplies through slowsims.freeprec and beffective.beff2ab); the gamma*2*pi*dt column of
 *      mrphy_blochsim_bwd_consts is finite for spins with gamma*2*pi*dt == 0 (it was 0/0); nothing else changed.
 *      Domain of the adjoint entry points under codes 3 / 4 (since ABI 2): E1 and E2 must be non-zero -- the sweep
 *      carries E h and divides by E once at the end, as the reference's adjoint divides at every step
 *      (sims.py:174-177); an E that underflowed to 0 (T < dt/100 in fp32) yields NaN gradients.  The Python layer
 *      raises for such constants; a direct caller uses codes 0 / 2 for them.
 *   4  round 4: mrphy_rfgr2beff_st -- mrphy_rfgr2beff with the cache policy of its stores chosen by the caller
 *      (MRPHY_STORE_*); mrphy_rfgr2beff is that call with MRPHY_STORE_AUTO.  Same bits under every policy.
 *   5  round 6: the history of the blochsim forward / adjoint may live in 1..8 separately allocated parts
 *      (mrphy_blochsim_hist_part_bytes, mrphy_blochsim_fwd_parts, mrphy_blochsim_bwd_parts); the single-pointer entry
 *      points are the one-part case, unchanged.  RCCL helpers for a ctypes-only multi-GPU consumer
 *      live in a library of their own beside this one (include/mrphy_comm.h, libmrphy_comm.so).  Same bits. */
#define MRPHY_ABI_VERSION 5

#define MRPHY_F32      0  /* T = float,  CT = float                                          */
#define MRPHY_F64      1  /* T = double, CT = double                                         */
#define MRPHY_F32_C64  2  /* T = float,  CT = double: the reference's behaviour when fp32 data
                             meets its fp64 default constants (sims.py:62-64,74-77 promote)  */
#define MRPHY_F32P     3  /* as MRPHY_F32 with the PRECISE fp32 step: S = sin(phi)/phi and
                             C = (1-cos(phi))/phi^2 evaluated in fp64 and rounded once, rounding
                             errors of the update and of the relaxation product carried (FMA
                             error-free transformations).  128^3 x 4096, phi <= 2.6 rad per step:
                             4.8e-6 relative L2 from exact arithmetic (MRPHY_F32: 2.0e-5, the
                             reference's own fp32 runs 2.6e-5); ~1.8x the arithmetic per step.
                             Accepted wherever MRPHY_F32 is by the entry points that integrate
                             (blochsim_fwd/_bwd/_1step, blochsim_rfgr_*, beff2ab).  The adjoint
                             sweeps take the same treatment (round 3): they carry t = E h, whose
                             recursion t <- E R^T t has the forward step's shape, through the same
                             compensated update with the same once-rounded S, C.  64^3 x 2048
                             (BASELINE configs[4]), all spins, against fp64 differentiation of
                             the same function: grad_M0 4.0e-6, grad_rf 2.9e-7, grad_gr 1.8e-6
                             (MRPHY_F32: 1.24e-5, 4.0e-6, 1.26e-5)                            */
#define MRPHY_F32P_C64 4  /* as MRPHY_F32_C64 with the precise step                          */

#define MRPHY_EINVAL  (-1)  /* bad argument (null pointer, negative size, unknown dtype)     */
#define MRPHY_EALIGN  (-2)  /* a pointer is not aligned to its element size                  */
#define MRPHY_ENOSPC  (-3)  /* workspace too small                                           */

/* Library identity / diagnostics. */
int         mrphy_abi_version(void);
const char* mrphy_error_string(int code);
/* Compiled offload architecture, e.g. "gfx950". */
const char* mrphy_arch(void);

/* Diagnostic (no reference counterpart): out[b] = id (0..7) of the XCD that workgroup b of a 1-D
 * grid of `nblocks` one-wave workgroups ran on (HW_REG_XCC_ID).  The kernels that write large
 * tensors give every XCD a contiguous share of the output by assuming that blocks b and b + 8
 * share an XCD; bench.py records whether the process really got that dealing. */
int mrphy_debug_xcc_map(int32_t* out, int64_t nblocks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K0  rfgr2beff -- replaces mrphy.beffective.rfgr2beff (beffective.py:107-168).
 *
 *   beff[n,s,t,0] = sum_c b1[n,s,0,c]*rf[n,0,t,c] - b1[n,s,1,c]*rf[n,1,t,c]
 *   beff[n,s,t,1] = sum_c b1[n,s,0,c]*rf[n,1,t,c] + b1[n,s,1,c]*rf[n,0,t,c]
 *   beff[n,s,t,2] = loc[n,s,:] . gr[n,:,t] + df[n,s]/gamma[n,s]
 *
 *   rf   (N|1, 2, nT, nC) contiguous, batch stride rf_sn elements (0 broadcasts the pulse)
 *   gr   (N|1, 3, nT)     contiguous, batch stride gr_sn
 *   loc  (N, nM, 3)       contiguous
 *   df, gamma             broadcastable per-spin constants; df may be NULL (no off-resonance)
 *   b1   (N, nM, 2, nC)   contiguous, or NULL: then nC must be 1 and Bx,By = rf (the host sums
 *                          a multi-coil rf over coils first, beffective.py:148-149)
 *   beff (N, nM, nT, 3)   contiguous output
 * Any nC: up to 64 coils (32 in fp64) in one launch; beyond, in blocks of that many coils whose launches
 * continue the ascending FMA chains of Bx, By from the values stored by the launch before (one extra read +
 * write pass over beff per block; the same bits as a single chain).
 * ------------------------------------------------------------------------------------------- */
int mrphy_rfgr2beff(int dtype,
                    const void* rf, int64_t rf_sn,
                    const void* gr, int64_t gr_sn,
                    const void* loc,
                    const void* df, int64_t df_sn, int64_t df_sm,
                    const void* gamma, int64_t gamma_sn, int64_t gamma_sm,
                    const void* b1,
                    void* beff,
                    int64_t N, int64_t nM, int64_t nT, int64_t nC,
                    void* stream);

/* The same with the cache policy of the Beff stores chosen by the caller (ABI 4).  What the stores leave in the
 * 256-MB memory-side cache decides how fast the kernel that reads Beff next starts (DESIGN.md "K1 right behind
 * K0"), what the writer pays for each encoding depends on the box and the block, and nothing a process can read
 * tells which: a caller that owns the block can time both (mrphy_amd.workspace.BeffArena does).
 *   MRPHY_STORE_AUTO   what mrphy_rfgr2beff does: SC1NT below 8 GB of Beff, NT from there up (round 5)
 *   MRPHY_STORE_PLAIN  cached stores
 *   MRPHY_STORE_NT     the non-temporal hint
 *   MRPHY_STORE_SC1NT  agent-scope write-through + non-temporal (16-byte stores; narrower ones take NT)
 * Any other value: MRPHY_EINVAL.  The results do not depend on the policy. */
#define MRPHY_STORE_AUTO  (-1)
#define MRPHY_STORE_PLAIN 0
#define MRPHY_STORE_NT    1
#define MRPHY_STORE_SC1NT 2
int mrphy_rfgr2beff_st(int dtype,
                       const void* rf, int64_t rf_sn,
                       const void* gr, int64_t gr_sn,
                       const void* loc,
                       const void* df, int64_t df_sn, int64_t df_sm,
                       const void* gamma, int64_t gamma_sn, int64_t gamma_sm,
                       const void* b1,
                       void* beff,
                       int64_t N, int64_t nM, int64_t nT, int64_t nC,
                       int store_policy,
                       void* stream);

/* Adjoint of K0 w.r.t. rf and gr (what autograd derives from beffective.py:137,160-165):
 *
 *   grad_gr[n,i,t]   = sum_s loc[n,s,i] * gB[n,s,t,2]
 *   grad_rf[n,0,t,c] = sum_s b1[n,s,0,c]*gB[n,s,t,0] + b1[n,s,1,c]*gB[n,s,t,1]
 *   grad_rf[n,1,t,c] = sum_s b1[n,s,0,c]*gB[n,s,t,1] - b1[n,s,1,c]*gB[n,s,t,0]
 *
 * Outputs are per-batch (N, 2, nT, nC) / (N, 3, nT); a broadcast pulse is reduced over n by the
 * caller.  Deterministic two-pass reduction over spins (fixed order, no float atomics).
 * `work` must hold mrphy_rfgr2beff_bwd_workspace(...) bytes.  b1 == NULL requires nC == 1, as in
 * mrphy_rfgr2beff (MRPHY_EINVAL otherwise).  With a map, 2..32 coils take ONE pass over grad_beff
 * (a thread owns a time point and its 2 nC + 3 running sums; the spins' b1 and loc, packed and
 * zero-padded into the tail of `work` by a pre-pass, reach the FMAs as scalar operands); more coils
 * take nC + 1 passes.  The workspace size depends on the path: always ask the query.
 */
size_t mrphy_rfgr2beff_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t nC);
int mrphy_rfgr2beff_bwd(int dtype,
                        const void* grad_beff,
                        const void* loc,
                        const void* b1,
                        void* grad_rf, void* grad_gr,
                        void* work, size_t work_bytes,
                        int64_t N, int64_t nM, int64_t nT, int64_t nC,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1  blochsim forward -- replaces mrphy.sims.BlochSim.forward (sims.py:32-132) with the
 * wrapper logic of sims.blochsim (sims.py:305-313) done by the caller.
 *
 * Per step t, with bt = g*Beff[n,s,t,:]  (g = gamma*2*pi*dt, computed by the host in the
 * reference's dtype, sims.py:62):
 *     rotate M about bt by -|bt| (Rodrigues, sims.py:100-121), then, if E1 != NULL,
 *     M <- (E2*Mx, E2*My, E1*Mz - E1m1)                           (sims.py:74-77,124)
 *
 *   Mi    (N, nM, 3)       contiguous, not modified
 *   Beff  (N, nM, nT, 3)   contiguous, not modified
 *   g, E1, E2, E1m1        broadcastable per-spin constants of type CT; E1==E2==E1m1==NULL
 *                          disables relaxation (reference T1=T2=None); E1m1 shares E1's strides
 *   Mo    (N, nM, 3)       contiguous output, final magnetisation
 *   Mpre                   optional (NULL = not wanted) history buffer of
 *                          mrphy_blochsim_hist_bytes() bytes: the magnetisation BEFORE each step,
 *                          the only history mrphy_blochsim_bwd needs (12 B per spin-step; the
 *                          reference keeps 40, sims.py:84-88).  Its layout is internal to the
 *                          library (per 64-spin tile, structure of arrays): pass it back to
 *                          mrphy_blochsim_bwd unchanged.
 * ------------------------------------------------------------------------------------------- */
size_t mrphy_blochsim_hist_bytes(int dtype, int64_t N, int64_t nM, int64_t nT);
int mrphy_blochsim_fwd(int dtype,
                       const void* Mi, const void* Beff,
                       const void* g,  int64_t g_sn,  int64_t g_sm,
                       const void* E1, int64_t E1_sn, int64_t E1_sm,
                       const void* E2, int64_t E2_sn, int64_t E2_sm,
                       const void* E1m1,
                       void* Mo, void* Mpre,
                       int64_t N, int64_t nM, int64_t nT,
                       void* stream);

/* K1h with the history in parts (ABI 5) -- the same forward; the history (sims.py:84-88) goes to `n_parts`
 * (1..MRPHY_HIST_MAX_PARTS) SEPARATELY ALLOCATED buffers of mrphy_blochsim_hist_part_bytes() bytes each instead of
 * one buffer.  Why: the rate at which a kernel can stream writes into a block depends on where the driver put the
 * block's physical pages (0.62 or 0.75 of HBM peak for K1h); a write stream spread over two
 * blocks from different allocations runs in the fast mode where one block of the same total size usually does not.
 * The history is internal to the library, so nothing forces it to be one allocation.
 *
 *   hist_parts   HOST array of n_parts device pointers (read during the call only); NULL or n_parts = 0: no history
 *   layout       how the 64-spin tiles are dealt to the parts:
 *                  MRPHY_HIST_BLOCKED      tile t -> part t / ceil(tiles / n_parts): with the XCD-contiguous block
 *                                          order of K1h / K3 every part is written by its own group of XCDs
 *                  MRPHY_HIST_INTERLEAVED  tile t -> part t % n_parts
 *                pass the same parts, in the same order, with the same layout to mrphy_blochsim_bwd_parts.
 * Results are bit-identical to mrphy_blochsim_fwd / _bwd (the kernels are the same; only a tile's base address differs).
 */
#define MRPHY_HIST_MAX_PARTS 8
#define MRPHY_HIST_BLOCKED      0
#define MRPHY_HIST_INTERLEAVED  1
size_t mrphy_blochsim_hist_part_bytes(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t n_parts);
int mrphy_blochsim_fwd_parts(int dtype,
                             const void* Mi, const void* Beff,
                             const void* g,  int64_t g_sn,  int64_t g_sm,
                             const void* E1, int64_t E1_sn, int64_t E1_sm,
                             const void* E2, int64_t E2_sn, int64_t E2_sm,
                             const void* E1m1,
                             void* Mo,
                             void* const* hist_parts, int64_t n_parts, int layout,
                             int64_t N, int64_t nM, int64_t nT,
                             void* stream);

/* K3 reading a history in parts: mrphy_blochsim_bwd (grad_consts = NULL) or mrphy_blochsim_bwd_consts
 * (grad_consts (N, nM, 4)) -- see below -- over the parts mrphy_blochsim_fwd_parts filled. */
int mrphy_blochsim_bwd_parts(int dtype,
                             const void* const* hist_parts, int64_t n_parts, int layout,
                             const void* Beff,
                             const void* g,  int64_t g_sn,  int64_t g_sm,
                             const void* E1, int64_t E1_sn, int64_t E1_sm,
                             const void* E2, int64_t E2_sn, int64_t E2_sm,
                             const void* grad_Mo,
                             void* grad_Mi, void* grad_Beff, void* grad_consts,
                             int64_t N, int64_t nM, int64_t nT,
                             void* stream);

/* K3  blochsim backward -- replaces mrphy.sims.BlochSim.backward (sims.py:135-269).
 *
 *   grad_Mo   (N, nM, 3)      contiguous
 *   grad_Mi   (N, nM, 3)      output, may be NULL
 *   grad_Beff (N, nM, nT, 3)  output, may be NULL; never aliases a saved tensor (the reference
 *                             overwrites its saved gamma*Beff, sims.py:239-264 -- not replicated)
 * Mpre is the history buffer mrphy_blochsim_fwd filled for the same (dtype, N, nM, nT).
 * E1m1 is not needed: the rotated, pre-relaxation magnetisation is recomputed from Mpre.
 */
int mrphy_blochsim_bwd(int dtype,
                       const void* Mpre, const void* Beff,
                       const void* g,  int64_t g_sn,  int64_t g_sm,
                       const void* E1, int64_t E1_sn, int64_t E1_sm,
                       const void* E2, int64_t E2_sn, int64_t E2_sm,
                       const void* grad_Mo,
                       void* grad_Mi, void* grad_Beff,
                       int64_t N, int64_t nM, int64_t nT,
                       void* stream);

/* As mrphy_blochsim_bwd, and in the same sweep the gradients w.r.t. the per-spin constants (round 3):
 *   grad_consts (N, nM, 4)  output: [dL/d(gamma*2*pi*dt), dL/dE1, dL/dE2, dL/d(E1-1)] per spin,
 *                           E1 and E1-1 treated as the two separate inputs they are.
 * The reference's slowsims.blochsim / blochsim_1step are plain differentiable torch ops
 * (slowsims.py:86-112, 42-51), so its callers get gradients w.r.t. T1, T2, gamma, dt; the host layer
 * chains these four to them (mrphy_amd.slowsims).  Without relaxation (E1 = E2 = NULL) only the first
 * entry is meaningful (the others are 0).  gamma*2*pi*dt must be non-zero.  Any shape / alignment
 * (the chunked adjoint kernel); about 25 more VALU operations per step than mrphy_blochsim_bwd.
 */
int mrphy_blochsim_bwd_consts(int dtype,
                              const void* Mpre, const void* Beff,
                              const void* g,  int64_t g_sn,  int64_t g_sm,
                              const void* E1, int64_t E1_sn, int64_t E1_sm,
                              const void* E2, int64_t E2_sn, int64_t E2_sm,
                              const void* grad_Mo,
                              void* grad_Mi, void* grad_Beff, void* grad_consts,
                              int64_t N, int64_t nM, int64_t nT,
                              void* stream);

/* blochsim_1step -- replaces mrphy.slowsims.blochsim_1step (slowsims.py:15-54): one step with
 * caller-supplied E1, E1-1, E2, gamma*2*pi*dt.  M (N,nM,3), b (N,nM,3) -> Mout (N,nM,3).
 * Mout may alias M (the reference mutates M in place on its all-zero-field branch).
 */
int mrphy_blochsim_1step(int dtype,
                         const void* M, const void* b,
                         const void* g,  int64_t g_sn,  int64_t g_sm,
                         const void* E1, int64_t E1_sn, int64_t E1_sm,
                         const void* E2, int64_t E2_sn, int64_t E2_sm,
                         const void* E1m1,
                         void* Mout,
                         int64_t N, int64_t nM,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2  fused rf,gr -> Mo: mrphy.mobjs.SpinArray.applypulse's back-to-back
 * rfgr2beff + blochsim (mobjs.py:435-446) without materialising Beff (N,nM,nT,3) in HBM.
 * Same operands as K0 and K1 together (`gamma` is the rfgr2beff one that divides df, `g` the
 * blochsim one).  Mck (nCk, N, nM, 3), nCk = ceil(nT/ck_every), receives the magnetisation
 * before steps 0, ck_every, 2*ck_every, ... when not NULL (checkpoints for an adjoint sweep;
 * ck_every must be a positive multiple of 8, the kernel's step batch).
 * Coils: register / LDS builds for 1, 2, 4, 8 (both precisions) and 16 ... 64 coils (float); more coils
 * than that run a generic build that re-reads rf and b1 from memory inside the coil loop (correct, two
 * orders of magnitude slower): call mrphy_rfgr2beff + mrphy_blochsim_fwd there, as the Python layer does.
 * ------------------------------------------------------------------------------------------- */
int mrphy_blochsim_rfgr_fwd(int dtype,
                            const void* Mi,
                            const void* rf, int64_t rf_sn,
                            const void* gr, int64_t gr_sn,
                            const void* loc,
                            const void* df, int64_t df_sn, int64_t df_sm,
                            const void* gamma, int64_t gamma_sn, int64_t gamma_sm,
                            const void* b1,
                            const void* g,  int64_t g_sn,  int64_t g_sm,
                            const void* E1, int64_t E1_sn, int64_t E1_sm,
                            const void* E2, int64_t E2_sn, int64_t E2_sm,
                            const void* E1m1,
                            void* Mo, void* Mck, int64_t ck_every,
                            int64_t N, int64_t nM, int64_t nT, int64_t nC,
                            void* stream);

/* K2b  adjoint of K2 (single-coil rf: rf (N|1, 2, nT, 1), b1 (N, nM, 2, 1) or NULL):
 *   grad_Mo (N,nM,3) -> grad_Mi (N,nM,3; may be NULL), grad_rf (N,2,nT), grad_gr (N,3,nT) (either may
 *   be NULL), per batch entry; a broadcast pulse is reduced over n by the caller.
 * Mck must be the checkpoints K2 wrote with ck_every = mrphy_blochsim_rfgr_ck_every(); nT must be a
 * multiple of it.  Each checkpoint segment is recomputed forward in registers, then swept
 * backwards; the reduction over spins is deterministic (per-wave partial rows in `work`, summed in
 * fixed order; no float atomics).  `work` must hold mrphy_blochsim_rfgr_bwd_workspace() bytes.
 */
int64_t mrphy_blochsim_rfgr_ck_every(void);
size_t mrphy_blochsim_rfgr_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT);
int mrphy_blochsim_rfgr_bwd(int dtype,
                            const void* Mck,
                            const void* rf, int64_t rf_sn,
                            const void* gr, int64_t gr_sn,
                            const void* loc,
                            const void* df, int64_t df_sn, int64_t df_sm,
                            const void* gamma, int64_t gamma_sn, int64_t gamma_sm,
                            const void* b1,
                            const void* g,  int64_t g_sn,  int64_t g_sm,
                            const void* E1, int64_t E1_sn, int64_t E1_sm,
                            const void* E2, int64_t E2_sn, int64_t E2_sm,
                            const void* E1m1,
                            const void* grad_Mo,
                            void* grad_Mi, void* grad_rf, void* grad_gr,
                            void* work, size_t work_bytes,
                            int64_t N, int64_t nM, int64_t nT,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * freeprec -- mrphy.sims.FreePrec forward / backward (sims.py:318-421; wrapper freeprec :424-458;
 * oracle form slowsims.py:134-174): precession by -2 pi df dur about z, then relaxation
 *     Mxy *= exp(-dur/T2),  Mz <- Mz exp(-dur/T1) - expm1(-dur/T1).
 *   Mi / Mo (N, nM, 3); dur (N|1,) with element stride dur_sn (0 broadcasts); T1, T2, df
 *   broadcastable per-spin constants of the DATA type; T1 == T2 == NULL: no relaxation;
 *   df == NULL: no precession.  The backward maps grad_Mo -> grad_Mi (the only input the
 *   reference differentiates, sims.py:321) and recomputes phi, E1, E2 instead of saving them.
 * ------------------------------------------------------------------------------------------- */
int mrphy_freeprec_fwd(int dtype, const void* Mi,
                       const void* dur, int64_t dur_sn,
                       const void* T1, int64_t T1_sn, int64_t T1_sm,
                       const void* T2, int64_t T2_sn, int64_t T2_sm,
                       const void* df, int64_t df_sn, int64_t df_sm,
                       void* Mo, int64_t N, int64_t nM, void* stream);
int mrphy_freeprec_bwd(int dtype, const void* grad_Mo,
                       const void* dur, int64_t dur_sn,
                       const void* T1, int64_t T1_sn, int64_t T1_sm,
                       const void* T2, int64_t T2_sn, int64_t T2_sm,
                       const void* df, int64_t df_sn, int64_t df_sm,
                       void* grad_Mi, int64_t N, int64_t nM, void* stream);

/* Gradients of freeprec w.r.t. its constants, per spin -- what autograd through the reference's plain torch ops
 * gives a caller of mrphy.slowsims.freeprec who differentiates w.r.t. dur, T1, T2, df (slowsims.py:151-174):
 *     grad_consts (N, nM, 4) = [dL/d dur, dL/d T1, dL/d T2, dL/d df]   (0 where the operand is NULL);
 * the caller sums each column over the axes its operand broadcasts along.  Mi is the INPUT magnetisation of the
 * forward call, grad_Mo the cotangent of its output. */
int mrphy_freeprec_bwd_consts(int dtype, const void* Mi, const void* grad_Mo,
                              const void* dur, int64_t dur_sn,
                              const void* T1, int64_t T1_sn, int64_t T1_sm,
                              const void* T2, int64_t T2_sn, int64_t T2_sm,
                              const void* df, int64_t df_sn, int64_t df_sm,
                              void* grad_consts, int64_t N, int64_t nM, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Linear time resampling of a pulse -- mrphy.mobjs.Pulse.interpT(kind='linear') (mobjs.py:177-220)
 * without the device -> host -> scipy -> device round trip.  `y` holds nch channels of nTo samples
 * (rf and gr rows), `out` nch x nTn.  The host supplies the grid (it depends on nTo, dt_old, dt_new
 * only): lo[j] = index of the left neighbour in the ZERO-PREPENDED source (mobjs.py:204-207; 0 means
 * the prepended sample), w[j] = t_new[j] - t[lo[j]], dx[j] = t[lo[j]+1] - t[lo[j]], fp64.
 * dir > 0: out = interp(y).  dir <= 0: the adjoint, `y` = grad wrt the resampled pulse (nch x nTn),
 * `out` = grad wrt the source (nch x nTo).
 * ------------------------------------------------------------------------------------------- */
int mrphy_pulse_interp_linear(int dtype, int dir, const void* y, void* out,
                              const void* lo, const void* w, const void* dx,
                              int64_t nch, int64_t nTo, int64_t nTn, void* stream);

/* The one-tap kinds of the same resampling -- mobjs.Pulse.interpT(kind=...) passes `kind` to
 * scipy.interpolate.interp1d (mobjs.py:201,214-215): 'nearest', 'nearest-up', 'previous', 'next',
 * 'zero'.  Each output sample is one sample of the zero-prepended source: sel[j] in [0, nTo]
 * (int32; 0 = the prepended zero, k >= 1 = y[., k - 1]), NON-DECREASING in j, computed by the host
 * for the grid (it depends on nTo, dt_old, dt_new and the kind only).  The caller guarantees the
 * range of sel; no arithmetic is done, so results equal the reference's bit for bit.
 * dir > 0: out (nch x nTn) = select(y (nch x nTo)).  dir <= 0: the adjoint, `y` = grad w.r.t. the
 * resampled pulse (nch x nTn), `out` = grad w.r.t. the source (nch x nTo), summed in j order. */
int mrphy_pulse_interp_select(int dtype, int dir, const void* y, void* out, const void* sel,
                              int64_t nch, int64_t nTo, int64_t nTn, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The two helpers mrphy.slowsims.blochsim_1step is written with in the reference.
 *
 * beff2uphi -- mrphy.beffective.beff2u\u03d5 (beffective.py:18-37):
 *     U = b / max(|b|, 1e-12),  Phi = -|b| * g          b (N,nM,3) -> U (N,nM,3), Phi (N,nM)
 * uphirot   -- mrphy.utils.u\u03d5rot (utils.py:333-359), Rodrigues rotation of nV vectors per row:
 *     Vo = cos(Phi) Vi + (1 - cos(Phi)) (U.Vi) U + sin(Phi) U x Vi
 *     U (rows,3), Phi (rows), Vi/Vo (rows, 3, nV) contiguous; Vo must not alias Vi.
 * ------------------------------------------------------------------------------------------- */
int mrphy_beff2uphi(int dtype, const void* b,
                    const void* g, int64_t g_sn, int64_t g_sm,
                    void* U, void* Phi, int64_t N, int64_t nM, void* stream);
int mrphy_uphirot(int dtype, const void* U, const void* Phi, const void* Vi, void* Vo,
                  int64_t rows, int64_t nV, void* stream);

/* Their adjoints.  The reference differentiates both through autograd over plain torch ops
 * (beffective.py:35-36: F.normalize + torch.norm; utils.py:351-357), which slowsims.blochsim_1step
 * -- the reference's implicit-Jacobian path -- relies on.
 *
 * beff2uphi_bwd: grad_U (N,nM,3) and grad_Phi (N,nM) (either may be NULL = zero) ->
 *     grad_b (N,nM,3) (NULL: skipped) and, if grad_g != NULL, the PER-SPIN gradient w.r.t. g,
 *     (N,nM), which the caller sums down to g's broadcast shape.
 * uphirot_bwd: grad_Vo (rows,3,nV) -> grad_U (rows,3), grad_Phi (rows) (summed over the nV
 *     vectors; U is treated as a free vector, as autograd does), grad_Vi (rows,3,nV); any output
 *     may be NULL. */
int mrphy_beff2uphi_bwd(int dtype, const void* b,
                        const void* g, int64_t g_sn, int64_t g_sm,
                        const void* grad_U, const void* grad_Phi, void* grad_b, void* grad_g,
                        int64_t N, int64_t nM, void* stream);
int mrphy_uphirot_bwd(int dtype, const void* U, const void* Phi, const void* Vi,
                      const void* grad_Vo, void* grad_U, void* grad_Phi, void* grad_Vi,
                      int64_t rows, int64_t nV, void* stream);

/* K2b for parallel transmit (nC = 1 .. mrphy_blochsim_rfgr_mc_max_coils() coils): as
 * mrphy_blochsim_rfgr_bwd, with rf (N|1, 2, nT, nC), b1 (N, nM, 2, nC) (required) and
 * grad_rf (N, 2, nT, nC).  The reference reaches these gradients through autograd over
 * rfgr2beff's complex coil sum (beffective.py:153-165) and BlochSim.backward (sims.py:135-269).
 * `work` must hold mrphy_blochsim_rfgr_mc_bwd_workspace() bytes.
 */
int64_t mrphy_blochsim_rfgr_mc_max_coils(void);
size_t mrphy_blochsim_rfgr_mc_bwd_workspace(int dtype, int64_t N, int64_t nM, int64_t nT, int64_t nC);
int mrphy_blochsim_rfgr_mc_bwd(int dtype,
                               const void* Mck,
                               const void* rf, int64_t rf_sn,
                               const void* gr, int64_t gr_sn,
                               const void* loc,
                               const void* df, int64_t df_sn, int64_t df_sm,
                               const void* gamma, int64_t gamma_sn, int64_t gamma_sm,
                               const void* b1,
                               const void* g,  int64_t g_sn,  int64_t g_sm,
                               const void* E1, int64_t E1_sn, int64_t E1_sm,
                               const void* E2, int64_t E2_sn, int64_t E2_sm,
                               const void* E1m1,
                               const void* grad_Mo,
                               void* grad_Mi, void* grad_rf, void* grad_gr,
                               void* work, size_t work_bytes,
                               int64_t N, int64_t nM, int64_t nT, int64_t nC,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * the steps either side of the path in SpinArray.applypulse (mobjs.py:427-433,449).
 *
 * The reference gathers/scatters with a boolean mask (`v[mask]`, `out[mask] = v_`), which has to
 * count the mask on the host at every call.  Here the mask is turned ONCE into two int32 lists
 *     idx (nM)  -- voxel number (row-major over *Nd) of compact spin j
 *     inv (nV)  -- compact spin number of voxel p, or -1 outside the mask
 * (mrphy_amd.masks.MaskIndex builds them with torch) and the kernels move raw 4- or 8-byte
 * elements; K = number of trailing elements per voxel (3 for M/loc, 1 for maps, 2*nC for b1Map).
 *
 * mask_extract -- SpinArray.extract (mobjs.py:532-553):  out_[n, j, :] = v[n, idx[j], :]
 *     v (N, nV, K) -> out_ (N, nM, K)
 * mask_embed   -- SpinArray.embed (mobjs.py:512-530):    out[n, p, :] = v_[n, inv[p], :] inside
 *     the mask; outside it `fillbits` when fill = 1 (a fresh output: the reference fills NaN),
 *     untouched when fill = 0 (the reference's `out=` form).        v_ (N, nM, K) -> out (N, nV, K)
 * cube_loc     -- SpinCube._update_loc_ (mobjs.py:815-839), 3-D grids:
 *     loc_[n, j, i] = fov[n, i] * ((c_i - dim_i / 2) / dim_i) + ofst[n, i],  c = unravel(idx[j])
 *     fov, ofst (N, 3) -> loc_ (N, nM, 3); same three roundings as the reference's expression.
 * Limits: nV < 2^31, N <= 65535.
 * ------------------------------------------------------------------------------------------- */
int mrphy_mask_extract(int elem_bytes, const void* v, const int32_t* idx, void* out_, int64_t N,
                       int64_t nV, int64_t nM, int64_t K, void* stream);
int mrphy_mask_embed(int elem_bytes, const void* v_, const int32_t* inv, void* out, int64_t N,
                     int64_t nV, int64_t nM, int64_t K, int fill, uint64_t fillbits, void* stream);
int mrphy_cube_loc(int dtype, const int32_t* idx, const void* fov, const void* ofst, void* loc_,
                   int64_t N, int64_t nM, int64_t nx, int64_t ny, int64_t nz, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hargreaves' A/B propagation (doi:10.1002/mrm.1170).
 *
 * beff2ab -- mrphy.beffective.beff2ab (beffective.py:40-104): the step map
 *     M -> relax(rotate(M, Beff[t]))  is affine; its composition over the pulse is  M -> A M + B.
 *     The kernel carries the four columns of [I | 0] through the nT steps with the arithmetic of
 *     mrphy_blochsim_fwd (the -(E1-1) offset acts on the B column only), so column j of A is
 *     bit-identical to blochsim(e_j) with a zero offset and B to blochsim(0).
 *     Beff (N,nM,nT,3); constants as for mrphy_blochsim_fwd but E1, E2, E1m1 are mandatory (the
 *     reference takes E1, E2 -- not T1, T2 -- and always relaxes; E1 = E2 = 1, E1m1 = 0 is "no
 *     relaxation");  A (N,nM,3,3) row-major A[i][j] (i = xyz of the result), B (N,nM,3).
 * blochsim_ab -- mrphy.slowsims.blochsim_ab (slowsims.py:117-131):  Mo = A M + B  per spin;
 *     rows = N*nM, M/Mo/B (rows,3), A (rows,3,3) contiguous.
 * blochsim_ab_bwd -- its adjoint: gM = A^T gMo (needs A), gA[i][j] = gMo_i M_j (needs M); either
 *     output may be NULL; the gradient w.r.t. B is gMo itself.
 * ------------------------------------------------------------------------------------------- */
int mrphy_beff2ab(int dtype, const void* Beff,
                  const void* g, int64_t g_sn, int64_t g_sm,
                  const void* E1, int64_t E1_sn, int64_t E1_sm,
                  const void* E2, int64_t E2_sn, int64_t E2_sm,
                  const void* E1m1, void* A, void* B,
                  int64_t N, int64_t nM, int64_t nT, void* stream);

/* beff2ab with its adjoint in ONE backward sweep over Beff (the reference differentiates its time
 * loop with autograd, beffective.py:88-100).  mrphy_beff2ab_save = mrphy_beff2ab that also records
 * the 3x4 state before every step in `hist` (opaque layout, mrphy_beff2ab_hist_bytes bytes: 48 B per
 * spin-step in fp32); mrphy_beff2ab_bwd turns grad_A (N,nM,3,3) / grad_B (N,nM,3) (either may be
 * NULL = zero) into grad_Beff (N,nM,nT,3): the four columns share each step's rotation, so Beff is
 * read once and grad_Beff written once, where four blochsim adjoints would read it four times,
 * write four gradients and add them up. */
size_t mrphy_beff2ab_hist_bytes(int dtype, int64_t N, int64_t nM, int64_t nT);
int mrphy_beff2ab_save(int dtype, const void* Beff,
                       const void* g, int64_t g_sn, int64_t g_sm,
                       const void* E1, int64_t E1_sn, int64_t E1_sm,
                       const void* E2, int64_t E2_sn, int64_t E2_sm,
                       const void* E1m1, void* A, void* B, void* hist,
                       int64_t N, int64_t nM, int64_t nT, void* stream);
int mrphy_beff2ab_bwd(int dtype, const void* hist, const void* Beff,
                      const void* g, int64_t g_sn, int64_t g_sm,
                      const void* E1, int64_t E1_sn, int64_t E1_sm,
                      const void* E2, int64_t E2_sn, int64_t E2_sm,
                      const void* grad_A, const void* grad_B, void* grad_Beff,
                      int64_t N, int64_t nM, int64_t nT, void* stream);

/* mrphy_beff2ab_bwd that also returns the gradients w.r.t. the per-spin constants, as the reference's autograd
 * through its time loop does for a caller who differentiates beff2ab w.r.t. E1, E2, gamma, dt (beffective.py:73-100):
 *     grad_consts (N, nM, 4) = [dL/d(gamma 2 pi dt), dL/dE1, dL/dE2, dL/d(E1-1)]
 * (the host chains them through its own expressions for those four, as for mrphy_blochsim_bwd_consts). */
int mrphy_beff2ab_bwd_consts(int dtype, const void* hist, const void* Beff,
                             const void* g, int64_t g_sn, int64_t g_sm,
                             const void* E1, int64_t E1_sn, int64_t E1_sm,
                             const void* E2, int64_t E2_sn, int64_t E2_sm,
                             const void* grad_A, const void* grad_B, void* grad_Beff,
                             void* grad_consts, int64_t N, int64_t nM, int64_t nT, void* stream);
int mrphy_blochsim_ab(int dtype, const void* M, const void* A, const void* B, void* Mo,
                      int64_t rows, void* stream);
int mrphy_blochsim_ab_bwd(int dtype, const void* M, const void* A, const void* gMo, void* gM,
                          void* gA, int64_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRPHY_HIP_H */
