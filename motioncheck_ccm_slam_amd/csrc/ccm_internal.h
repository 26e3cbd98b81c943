// ccm_internal.h -- shared between the host-side translation units of libccm_hot.so.
#pragma once
#include <hip/hip_runtime.h>
#include <roctracer/roctx.h>
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
struct PoseState;
struct Sim3State;
struct EssState;

struct ProfLabel { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; };

struct ccm_ctx {
    bool prof_on = false;
    ProfLabel prof[CCM_PROF_COUNT];
    int device = 0;
    hipStream_t stream = nullptr;
    // Two auxiliary streams per context, created on first use and SHARED by its subsystems (ccm_aux_stream below): the runtime maps
    // streams onto a handful of hardware queues (4 by default), and streams that share a queue run one after the other -- when the
    // extractor's copy streams and the bundle adjustment's side stream were separate objects, the fifth stream of the process put the
    // coarse inversion into the PCG's queue and the solve took 30 instead of 18 ms (round 3).
    //   aux[0]  lowest priority: the extractor's uploads, the bundle adjustment's coarse inversion
    //   aux[1]  default priority: the extractor's downloads
    hipStream_t aux[2] = { nullptr, nullptr };
    std::string err;
    OrbState* orb = nullptr;
    MatchState* match = nullptr;
    BaState* ba = nullptr;
    CommState* comm = nullptr;
    PoseState* pose = nullptr;
    Sim3State* sim3 = nullptr;
    EssState* ess = nullptr;
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
hipStream_t ccm_aux_stream(ccm_ctx* c, int which);      // nullptr if it cannot be created

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

// Brackets the launches issued while it is alive with two events when profiling is on.
struct ProfScope {
    ccm_ctx* c; int label; hipEvent_t stop = nullptr;
    ProfScope(ccm_ctx* ctx, int lab) : c(ctx), label(lab)
    {
        if (!c->prof_on) return;
        ProfLabel& L = c->prof[label];
        if (L.used == L.ev.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            L.ev.emplace_back(a, b);
        }
        (void)hipEventRecord(L.ev[L.used].first, c->stream);
        stop = L.ev[L.used].second;
        L.used++;
    }
    ~ProfScope() { if (stop) (void)hipEventRecord(stop, c->stream); }
};

// Named range in rocprofv3 --marker-trace / roctracer timelines (the reference's only instrumentation is
// g2o's G2OBatchStatistics, block_solver.hpp:441-453, mirrored by the timers of ccm_ba_result; these ranges show where a
// call's host time goes next to the kernels).  A push/pop pair costs ~50 ns when no tool is attached.
struct RoctxRange {
    bool open = true;
    explicit RoctxRange(const char* name) { roctxRangePushA(name); }
    void end() { if (open) { roctxRangePop(); open = false; } }       // close before the scope ends (early returns still balance)
    ~RoctxRange() { end(); }
    RoctxRange(const RoctxRange&) = delete; RoctxRange& operator=(const RoctxRange&) = delete;
};

void orb_state_free(OrbState*);
void match_state_free(MatchState*);
void ba_state_free(BaState*);
void comm_state_free(ccm_ctx*);
void pose_state_free(PoseState*);
void sim3_state_free(Sim3State*);
void ess_state_free(EssState*);
