// comm.cpp -- libmrphy_comm.so: the C ABI of include/mrphy_comm.h over RCCL (host code only: RCCL launches its own kernels).
// One all-gather (final magnetisation) and one all-reduce (pulse gradients) per step is all the sharded simulation
// exchanges; both go straight to ncclAllGather / ncclAllReduce on the caller's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/mrphy_comm.h"

static_assert(MRPHY_COMM_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {
inline int rc(ncclResult_t r) { return r == ncclSuccess ? 0 : MRPHY_COMM_NCCL_BASE + (int)r; }
inline bool dtype_of(int code, ncclDataType_t& t)
{
    if (code == MRPHY_COMM_F32) { t = ncclFloat32; return true; }
    if (code == MRPHY_COMM_F64) { t = ncclFloat64; return true; }
    return false;
}
}  // namespace

extern "C" {

int mrphy_comm_abi_version(void) { return MRPHY_COMM_ABI_VERSION; }

const char* mrphy_comm_error_string(int code)
{
    if (code == 0) return "success";
    if (code == MRPHY_COMM_EINVAL) return "mrphy_comm: invalid argument";
    if (code >= MRPHY_COMM_NCCL_BASE) return ncclGetErrorString((ncclResult_t)(code - MRPHY_COMM_NCCL_BASE));
    return "mrphy_comm: unknown error";
}

int mrphy_comm_unique_id(void* id)
{
    if (!id) return MRPHY_COMM_EINVAL;
    ncclUniqueId u;
    const ncclResult_t r = ncclGetUniqueId(&u);
    if (r == ncclSuccess) memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return rc(r);
}

int mrphy_comm_init(const void* id, int nranks, int rank, void** comm)
{
    if (!id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return MRPHY_COMM_EINVAL;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, nranks, u, rank);
    *comm = r == ncclSuccess ? (void*)c : nullptr;
    return rc(r);
}

int mrphy_comm_destroy(void* comm)
{
    return comm ? rc(ncclCommDestroy((ncclComm_t)comm)) : 0;
}

int mrphy_comm_allgather_spins(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream)
{
    ncclDataType_t t;
    if (!comm || count < 0 || !dtype_of(dtype, t)) return MRPHY_COMM_EINVAL;
    if (count == 0) return 0;
    if (!send || !recv) return MRPHY_COMM_EINVAL;
    return rc(ncclAllGather(send, recv, (size_t)count, t, (ncclComm_t)comm, (hipStream_t)stream));
}

int mrphy_comm_allreduce_pulse_grads(void* comm, void* buf, int64_t count, int dtype, void* stream)
{
    ncclDataType_t t;
    if (!comm || count < 0 || !dtype_of(dtype, t)) return MRPHY_COMM_EINVAL;
    if (count == 0) return 0;
    if (!buf) return MRPHY_COMM_EINVAL;
    return rc(ncclAllReduce(buf, buf, (size_t)count, t, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

}  // extern "C"
