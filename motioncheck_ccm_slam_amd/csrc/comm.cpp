// comm.cpp -- RCCL communicator attached to a context (multi-GPU global BA).
#include "ccm_internal.h"
#include <rccl/rccl.h>

// `ext`: a caller-supplied transport (ccm_comm_attach) used instead of RCCL
struct CommState { ncclComm_t comm = nullptr; int n_ranks = 1, rank = 0; bool has_ext = false; ccm_comm_transport ext{}; };

void comm_state_free(ccm_ctx* c)
{
    if (!c || !c->comm) return;
    if (c->comm->comm) (void)ncclCommDestroy(c->comm->comm);
    if (c->comm->has_ext && c->comm->ext.destroy) c->comm->ext.destroy(c->comm->ext.user);
    delete c->comm;
    c->comm = nullptr;
}

extern "C" {

int ccm_comm_unique_id(uint8_t id[CCM_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= CCM_COMM_ID_BYTES, "id buffer too small");
    if (!id) return CCM_E_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return CCM_E_COMM;
    std::memset(id, 0, CCM_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return CCM_OK;
}

int ccm_comm_init(ccm_ctx* c, const uint8_t id[CCM_COMM_ID_BYTES], int n_ranks, int rank)
{
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return c ? ccm_fail(c, CCM_E_ARG, "bad communicator arguments") : CCM_E_ARG;
    CCM_HIP(c, hipSetDevice(c->device));
    comm_state_free(c);
    c->comm = new CommState();
    c->comm->n_ranks = n_ranks; c->comm->rank = rank;
    // A one-rank job needs no communicator.  CCM_COMM_RCCL_SINGLE=1 creates one all the same, so that the RCCL
    // calls of the sharded path (ncclCommInitRank, in-place ncclAllReduce on the context's stream) can be run on a
    // machine with one GPU (tests/test_ba_gpu.py).
    const char* single = getenv("CCM_COMM_RCCL_SINGLE");
    if (n_ranks == 1 && !(single && single[0] == '1')) return CCM_OK;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&c->comm->comm, n_ranks, u, rank);
    if (r != ncclSuccess) { c->comm->comm = nullptr; return ccm_fail(c, CCM_E_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r)); }
    return CCM_OK;
}

int ccm_comm_attach(ccm_ctx* c, const ccm_comm_transport* t, int n_ranks, int rank)
{
    if (!c || !t || !t->allreduce_f64 || !t->allreduce_u8_max || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return c ? ccm_fail(c, CCM_E_ARG, "bad transport arguments") : CCM_E_ARG;
    comm_state_free(c);
    c->comm = new CommState();
    c->comm->n_ranks = n_ranks; c->comm->rank = rank;
    c->comm->has_ext = true; c->comm->ext = *t;
    return CCM_OK;
}

int ccm_comm_destroy(ccm_ctx* c)
{
    if (!c) return CCM_E_ARG;
    comm_state_free(c);
    return CCM_OK;
}

}  // extern "C"

// used by ba_host.cpp
int comm_ranks(const ccm_ctx* c) { return c->comm ? c->comm->n_ranks : 1; }
int comm_rank(const ccm_ctx* c) { return c->comm ? c->comm->rank : 0; }
int comm_allreduce_f64(ccm_ctx* c, double* dev, size_t n, bool max_op)
{
    if (!c->comm || (c->comm->n_ranks == 1 && !c->comm->comm)) return CCM_OK;
    if (c->comm->has_ext) {
        if (c->comm->ext.allreduce_f64(c->comm->ext.user, dev, n, max_op ? 1 : 0, (void*)c->stream)) return ccm_fail(c, CCM_E_COMM, "the attached transport's all-reduce (f64, %zu) failed", n);
        return CCM_OK;
    }
    ncclResult_t r = ncclAllReduce(dev, dev, n, ncclDouble, max_op ? ncclMax : ncclSum, c->comm->comm, c->stream);
    if (r != ncclSuccess) return ccm_fail(c, CCM_E_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
    return CCM_OK;
}
int comm_allreduce_u8_max(ccm_ctx* c, uint8_t* dev, size_t n)
{
    if (!c->comm || (c->comm->n_ranks == 1 && !c->comm->comm)) return CCM_OK;
    if (c->comm->has_ext) {
        if (c->comm->ext.allreduce_u8_max(c->comm->ext.user, dev, n, (void*)c->stream)) return ccm_fail(c, CCM_E_COMM, "the attached transport's all-reduce (u8 max, %zu) failed", n);
        return CCM_OK;
    }
    ncclResult_t r = ncclAllReduce(dev, dev, n, ncclUint8, ncclMax, c->comm->comm, c->stream);
    if (r != ncclSuccess) return ccm_fail(c, CCM_E_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
    return CCM_OK;
}
