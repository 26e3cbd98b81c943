// shm_transport.cpp -- TEST SCAFFOLDING, not part of libccm_hot.so.
//
// A ccm_comm_transport (include/ccm_hot.h) whose ranks are processes of one host exchanging through a POSIX shared-memory
// segment, attached with ccm_comm_attach.  It exists so that the sharded global BA (landmark partition, block-pattern union,
// partial reduced systems, collective stop flag) can be run end to end on a machine with ONE GPU, where two RCCL ranks
// cannot share the device; the production transport is RCCL (ccm_comm_init).  All waits are bounded: a peer that died or
// never arrived makes the collective return an error after TIMEOUT_S instead of hanging the test.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/ccm_hot.h"

namespace {
constexpr double TIMEOUT_S = 120.0;
struct Header { std::atomic<unsigned> magic, count, gen; };
struct Shm {
    Header* hdr = nullptr; char* slots = nullptr; size_t cap = 0, map_bytes = 0; std::string name; bool owner = false;
    int n_ranks = 1, rank = 0;
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// sense-reversing barrier on two atomics in the segment, with a deadline
int barrier(Shm* s)
{
    const unsigned g = s->hdr->gen.load(std::memory_order_acquire);
    if (s->hdr->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)s->n_ranks) {
        s->hdr->count.store(0, std::memory_order_relaxed);
        s->hdr->gen.fetch_add(1, std::memory_order_release);
        return 0;
    }
    const double t0 = now();
    int spins = 0;
    while (s->hdr->gen.load(std::memory_order_acquire) == g) {
        if (++spins > 2000) { std::this_thread::sleep_for(std::chrono::microseconds(50)); if (now() - t0 > TIMEOUT_S) return -1; }
    }
    return 0;
}

template <class T, class Op>
int allreduce(Shm* s, T* dev, size_t n, hipStream_t st, Op op)
{
    const size_t bytes = n * sizeof(T);
    if (bytes > s->cap) { fprintf(stderr, "[shm transport] all-reduce of %zu bytes exceeds the segment's %zu per rank\n", bytes, s->cap); return -1; }
    T* mine = reinterpret_cast<T*>(s->slots + s->cap * s->rank);
    if (hipMemcpyAsync(mine, dev, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
    if (barrier(s)) return -3;
    std::vector<T> tot(n);                                   // every rank adds the slots in rank order: same bits everywhere
    const T* r0 = reinterpret_cast<const T*>(s->slots);
    for (size_t i = 0; i < n; i++) tot[i] = r0[i];
    for (int r = 1; r < s->n_ranks; r++) {
        const T* rr = reinterpret_cast<const T*>(s->slots + s->cap * r);
        for (size_t i = 0; i < n; i++) tot[i] = op(tot[i], rr[i]);
    }
    if (barrier(s)) return -3;                               // everybody has read every slot
    if (hipMemcpyAsync(dev, tot.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
    return 0;
}

int ar_f64(void* u, double* dev, size_t n, int max_op, void* st)
{
    Shm* s = static_cast<Shm*>(u);
    return max_op ? allreduce(s, dev, n, (hipStream_t)st, [](double a, double b) { return a > b ? a : b; })
                  : allreduce(s, dev, n, (hipStream_t)st, [](double a, double b) { return a + b; });
}
int ar_u8(void* u, uint8_t* dev, size_t n, void* st)
{
    return allreduce(static_cast<Shm*>(u), dev, n, (hipStream_t)st, [](uint8_t a, uint8_t b) { return a > b ? a : b; });
}
void destroy(void* u)
{
    Shm* s = static_cast<Shm*>(u);
    if (s->hdr) munmap(s->hdr, s->map_bytes);
    if (s->owner) shm_unlink(s->name.c_str());
    delete s;
}
}  // namespace

// Fills *out; returns 0, or a negative code when the segment cannot be set up within the deadline.
extern "C" int shm_transport_create(const char* name, int n_ranks, int rank, size_t capacity_bytes, ccm_comm_transport* out)
{
    if (!name || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks || capacity_bytes == 0) return -1;
    Shm* s = new Shm();
    s->name = name; s->cap = (capacity_bytes + 63) & ~(size_t)63; s->owner = rank == 0; s->n_ranks = n_ranks; s->rank = rank;
    const size_t head = 4096;
    s->map_bytes = head + s->cap * n_ranks;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)s->map_bytes) != 0) { if (fd >= 0) close(fd); delete s; return -2; }
    } else {
        const double t0 = now();
        while (fd < 0) {                                                   // wait for rank 0
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) { if (now() - t0 > TIMEOUT_S) { delete s; return -3; } std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
        }
        for (;;) {                                                         // until rank 0 has sized it
            if (lseek(fd, 0, SEEK_END) >= (off_t)s->map_bytes) break;
            if (now() - t0 > TIMEOUT_S) { close(fd); delete s; return -4; }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    void* m = mmap(nullptr, s->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete s; return -5; }
    s->hdr = static_cast<Header*>(m);
    s->slots = static_cast<char*>(m) + head;
    if (rank == 0) {
        s->hdr->count.store(0); s->hdr->gen.store(0);
        s->hdr->magic.store(0xCC3A11u, std::memory_order_release);
    } else {
        const double t0 = now();
        while (s->hdr->magic.load(std::memory_order_acquire) != 0xCC3A11u) {
            if (now() - t0 > TIMEOUT_S) { destroy(s); return -6; }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    if (barrier(s)) { destroy(s); return -7; }
    out->user = s; out->allreduce_f64 = ar_f64; out->allreduce_u8_max = ar_u8; out->destroy = destroy;
    return 0;
}
