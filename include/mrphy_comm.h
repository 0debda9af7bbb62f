/* mrphy_comm.h -- C ABI of libmrphy_comm.so: the two collectives of the spin-sharded Bloch simulation over RCCL.
 *
 * SURVEY.md section 8(b) lists RCCL helpers among what the drop-in library exports, section 8(e) what they carry: spins
 * are independent, so the compact spin axis is cut into `nranks` contiguous blocks (one process per GPU) and NOTHING is
 * exchanged during the time loop;
 *
 *     forward            ONE all-gather of the final magnetisation  Mo (N, nM/nranks, 3)   (3.1 MB per rank at 128^3)
 *     pulse-design step  ONE all-reduce(sum) of grad_rf (N, 2, nT[, nC]) and grad_gr (N, 3, nT), flattened (20 nT bytes)
 *
 * The reference (tianrluo/MRphy.py) has no distributed code at all; these entry points are the MI355X scale-out of its
 * spin axis for a consumer that binds the C ABI directly (ctypes, cgo, JNI ...) and has no torch.distributed.  The
 * Python layer's default route is torch.distributed (backend "nccl" = RCCL; mrphy_amd/dist.py), which can also be told
 * to go through these (dist.use_c_abi).  A separate library beside libmrphy_hip.so so that the kernels' library does not
 * depend on RCCL; it links librccl.so.1 -- in a process that has loaded PyTorch-ROCm that is PyTorch's own copy.
 *
 * Conventions as in mrphy_hip.h: plain pointers and sizes, device pointers from the caller, `stream` a hipStream_t, calls
 * asynchronous on that stream (no host sync), return value 0 or a negative MRPHY_COMM_E* / 1000 + ncclResult_t.
 * Rendezvous: rank 0 calls mrphy_comm_unique_id and hands the 128 bytes to the other ranks by any channel it has
 * (a file, a socket, MPI, a torch.distributed broadcast); every rank then calls mrphy_comm_init on ITS device
 * (hipSetDevice first), which blocks until all ranks have arrived.
 */
#ifndef MRPHY_COMM_H
#define MRPHY_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRPHY_COMM_ABI_VERSION 1
#define MRPHY_COMM_UNIQUE_ID_BYTES 128      /* NCCL_UNIQUE_ID_BYTES */
#define MRPHY_COMM_EINVAL (-1)              /* bad argument */
#define MRPHY_COMM_NCCL_BASE 1000           /* 1000 + ncclResult_t for an RCCL failure */

/* data type codes: the T of mrphy_hip.h's dtype codes */
#define MRPHY_COMM_F32 0
#define MRPHY_COMM_F64 1

int         mrphy_comm_abi_version(void);
const char* mrphy_comm_error_string(int code);

/* rank 0: fill `id` (MRPHY_COMM_UNIQUE_ID_BYTES bytes, host memory) for one communicator */
int mrphy_comm_unique_id(void* id);

/* every rank: join the communicator of `id` as `rank` of `nranks` on the current device; *comm receives the handle */
int mrphy_comm_init(const void* id, int nranks, int rank, void** comm);
int mrphy_comm_destroy(void* comm);

/* Forward: all-gather of the final magnetisation.  send: this rank's block, `count` ELEMENTS (= N * spins_per_rank * 3);
 * recv: nranks * count elements, rank r's block at recv + r * count (the caller pads ragged blocks to the largest:
 * blocks differ by at most one spin).  send may lie inside recv at its own slot (in-place). */
int mrphy_comm_allgather_spins(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream);

/* Pulse-design backward: in-place all-reduce(sum) of `count` elements -- grad_rf and grad_gr of the replicated pulse,
 * flattened into one buffer by the caller (one collective, latency-bound). */
int mrphy_comm_allreduce_pulse_grads(void* comm, void* buf, int64_t count, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRPHY_COMM_H */
