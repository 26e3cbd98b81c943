// comm.cpp -- RCCL communicator attached to a context (multi-GPU global BA).
#include "ccm_internal.h"
#include <rccl/rccl.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>

// Rehearsal transport: ranks = processes of this host exchanging through one POSIX shared-memory segment.  It exists
// so that the sharded global BA (partition, block-pattern union, partial sums) can be run end to end on a machine
// with ONE GPU; the production transport is RCCL (ccm_comm_init).
struct ShmHeader { std::atomic<unsigned> magic; pthread_barrier_t bar; };
struct ShmComm { ShmHeader* hdr = nullptr; char* slots = nullptr; size_t cap = 0, map_bytes = 0; std::string name; bool owner = false; };

struct CommState { ncclComm_t comm = nullptr; int n_ranks = 1, rank = 0; ShmComm* shm = nullptr; };

void comm_state_free(ccm_ctx* c)
{
    if (!c || !c->comm) return;
    if (c->comm->comm) (void)ncclCommDestroy(c->comm->comm);
    if (c->comm->shm) {
        ShmComm* s = c->comm->shm;
        if (s->hdr) munmap(s->hdr, s->map_bytes);
        if (s->owner) shm_unlink(s->name.c_str());
        delete s;
    }
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

int ccm_comm_init_shm(ccm_ctx* c, const char* name, int n_ranks, int rank, size_t capacity_bytes)
{
    if (!c || !name || n_ranks < 1 || rank < 0 || rank >= n_ranks || capacity_bytes == 0) return c ? ccm_fail(c, CCM_E_ARG, "bad communicator arguments") : CCM_E_ARG;
    comm_state_free(c);
    c->comm = new CommState();
    c->comm->n_ranks = n_ranks; c->comm->rank = rank;
    if (n_ranks == 1) return CCM_OK;
    ShmComm* s = new ShmComm();
    c->comm->shm = s;
    s->name = name; s->cap = (capacity_bytes + 63) & ~(size_t)63; s->owner = rank == 0;
    const size_t head = (sizeof(ShmHeader) + 4095) & ~(size_t)4095;
    s->map_bytes = head + s->cap * n_ranks;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)s->map_bytes) != 0) { if (fd >= 0) close(fd); return ccm_fail(c, CCM_E_COMM, "shm_open/ftruncate(%s) failed", name); }
    } else {
        for (int tries = 0; tries < 3000 && fd < 0; tries++) {           // wait (up to 30 s) for rank 0
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (fd < 0) return ccm_fail(c, CCM_E_COMM, "shm_open(%s): rank 0 never created the segment", name);
        for (int tries = 0; tries < 3000; tries++) {                     // until rank 0 has sized it
            off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)s->map_bytes) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    void* m = mmap(nullptr, s->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ccm_fail(c, CCM_E_COMM, "mmap of the communicator segment failed");
    s->hdr = static_cast<ShmHeader*>(m);
    s->slots = static_cast<char*>(m) + head;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&s->hdr->bar, &a, (unsigned)n_ranks);
        pthread_barrierattr_destroy(&a);
        s->hdr->magic.store(0xCC3A11u, std::memory_order_release);
    } else {
        int tries = 0;
        while (s->hdr->magic.load(std::memory_order_acquire) != 0xCC3A11u) {
            if (++tries > 3000) return ccm_fail(c, CCM_E_COMM, "rank 0 never initialised the communicator segment");
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    pthread_barrier_wait(&s->hdr->bar);
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
// host-staged all-reduce of the rehearsal transport: every rank adds the ranks' slots in rank order (same bits everywhere)
template <class T, class Op>
static int shm_allreduce(ccm_ctx* c, T* dev, size_t n, Op op)
{
    ShmComm* s = c->comm->shm;
    const size_t bytes = n * sizeof(T);
    if (bytes > s->cap) return ccm_fail(c, CCM_E_COMM, "all-reduce of %zu bytes exceeds the segment's %zu per rank", bytes, s->cap);
    T* mine = reinterpret_cast<T*>(s->slots + s->cap * c->comm->rank);
    CCM_HIP(c, hipMemcpyAsync(mine, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    pthread_barrier_wait(&s->hdr->bar);
    std::vector<T> tot(n);
    const T* r0 = reinterpret_cast<const T*>(s->slots);
    for (size_t i = 0; i < n; i++) tot[i] = r0[i];
    for (int r = 1; r < c->comm->n_ranks; r++) {
        const T* rr = reinterpret_cast<const T*>(s->slots + s->cap * r);
        for (size_t i = 0; i < n; i++) tot[i] = op(tot[i], rr[i]);
    }
    pthread_barrier_wait(&s->hdr->bar);                                  // everybody has read every slot
    CCM_HIP(c, hipMemcpyAsync(dev, tot.data(), bytes, hipMemcpyHostToDevice, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    return CCM_OK;
}

int comm_allreduce_f64(ccm_ctx* c, double* dev, size_t n, bool max_op)
{
    if (!c->comm || (c->comm->n_ranks == 1 && !c->comm->comm)) return CCM_OK;
    if (c->comm->shm)
        return max_op ? shm_allreduce(c, dev, n, [](double a, double b) { return a > b ? a : b; })
                      : shm_allreduce(c, dev, n, [](double a, double b) { return a + b; });
    ncclResult_t r = ncclAllReduce(dev, dev, n, ncclDouble, max_op ? ncclMax : ncclSum, c->comm->comm, c->stream);
    if (r != ncclSuccess) return ccm_fail(c, CCM_E_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
    return CCM_OK;
}
int comm_allreduce_u8_max(ccm_ctx* c, uint8_t* dev, size_t n)
{
    if (!c->comm || (c->comm->n_ranks == 1 && !c->comm->comm)) return CCM_OK;
    if (c->comm->shm) return shm_allreduce(c, dev, n, [](uint8_t a, uint8_t b) { return a > b ? a : b; });
    ncclResult_t r = ncclAllReduce(dev, dev, n, ncclUint8, ncclMax, c->comm->comm, c->stream);
    if (r != ncclSuccess) return ccm_fail(c, CCM_E_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
    return CCM_OK;
}
