// ccm_internal.h -- shared between the host-side translation units of libccm_hot.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ccm_hot.h"

struct OrbState;
struct MatchState;
struct BaState;
struct CommState;

struct ccm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    OrbState* orb = nullptr;
    MatchState* match = nullptr;
    BaState* ba = nullptr;
    CommState* comm = nullptr;
};

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

int ccm_fail(ccm_ctx* c, int code, const char* fmt, ...);

#define CCM_HIP(c, expr)                                                                   \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ccm_fail((c), CCM_E_DEVICE, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                            hipGetErrorString(e__));                                       \
    } while (0)

#define CCM_RESERVE(c, buf, bytes)                                                        \
    do {                                                                                  \
        if ((buf).reserve(bytes))                                                         \
            return ccm_fail((c), CCM_E_NOMEM, "%s:%d device alloc of %zu bytes failed",   \
                            __FILE__, __LINE__, (size_t)(bytes));                         \
    } while (0)

void orb_state_free(OrbState*);
void match_state_free(MatchState*);
void ba_state_free(BaState*);
void comm_state_free(ccm_ctx*);
