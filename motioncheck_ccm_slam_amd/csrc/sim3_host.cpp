// sim3_host.cpp -- C ABI of the batched Optimizer::OptimizeSim3 (cslam/src/Optimizer.cpp:867-1062); the whole
// schedule of every problem runs in one kernel launch (sim3_kernels.hip).
#include "ccm_internal.h"
#include <algorithm>

struct Sim3Dev {
    int n_problems; double* sim3; const int* fix_scale; const double* K1; const double* K2; const int* first;
    const double* P1; const double* P2; const double* obs1; const double* obs2; const double* info1; const double* info2;
    const float* th2; double* err; uint8_t* inlier; int* n_in;
};
void sim3_launch(hipStream_t, const Sim3Dev&);

struct Sim3State { DevBuf sim3, fix, K1, K2, first, P1, P2, o1, o2, i1, i2, th2, err, inl, nin; };
void sim3_state_free(Sim3State* s)
{
    if (!s) return;
    DevBuf* all[] = { &s->sim3, &s->fix, &s->K1, &s->K2, &s->first, &s->P1, &s->P2, &s->o1, &s->o2, &s->i1, &s->i2, &s->th2, &s->err, &s->inl, &s->nin };
    for (DevBuf* b : all) b->release();
    delete s;
}

extern "C" int ccm_optimize_sim3(ccm_ctx* c, ccm_sim3_problem* pb)
{
    RoctxRange roctx_("ccm_optimize_sim3");
    if (!c || !pb) return CCM_E_ARG;
    if (pb->n_problems == 0) return CCM_OK;
    if (pb->n_problems < 0 || !pb->sim3 || !pb->fix_scale || !pb->K1 || !pb->K2 || !pb->first || !pb->th2 || !pb->n_inliers)
        return ccm_fail(c, CCM_E_ARG, "bad Sim3 problem");
    const int F = pb->n_problems;
    if (pb->first[0] != 0) return ccm_fail(c, CCM_E_ARG, "first[0] must be 0");
    for (int f = 0; f < F; f++) if (pb->first[f + 1] < pb->first[f]) return ccm_fail(c, CCM_E_ARG, "first[] must be non-decreasing");
    const size_t T = (size_t)pb->first[F];
    if (T > 0 && (!pb->P1 || !pb->P2 || !pb->obs1 || !pb->obs2 || !pb->info1 || !pb->info2 || !pb->inlier))
        return ccm_fail(c, CCM_E_ARG, "bad Sim3 problem");
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->sim3) c->sim3 = new Sim3State();
    Sim3State& S = *c->sim3;
    hipStream_t st = c->stream;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(S.sim3, pb->sim3, (size_t)F * 64))) return rc;
    if ((rc = up(S.fix, pb->fix_scale, (size_t)F * 4))) return rc;
    if ((rc = up(S.K1, pb->K1, (size_t)F * 32))) return rc;
    if ((rc = up(S.K2, pb->K2, (size_t)F * 32))) return rc;
    if ((rc = up(S.first, pb->first, ((size_t)F + 1) * 4))) return rc;
    if ((rc = up(S.th2, pb->th2, (size_t)F * 4))) return rc;
    if ((rc = up(S.P1, pb->P1, T * 24))) return rc;
    if ((rc = up(S.P2, pb->P2, T * 24))) return rc;
    if ((rc = up(S.o1, pb->obs1, T * 16))) return rc;
    if ((rc = up(S.o2, pb->obs2, T * 16))) return rc;
    if ((rc = up(S.i1, pb->info1, T * 8))) return rc;
    if ((rc = up(S.i2, pb->info2, T * 8))) return rc;
    CCM_RESERVE(c, S.err, std::max<size_t>(T * 32, 16)); CCM_RESERVE(c, S.inl, std::max<size_t>(T, 16));
    CCM_RESERVE(c, S.nin, (size_t)F * 4);
    Sim3Dev D{ F, S.sim3.as<double>(), S.fix.as<int>(), S.K1.as<double>(), S.K2.as<double>(), S.first.as<int>(), S.P1.as<double>(),
               S.P2.as<double>(), S.o1.as<double>(), S.o2.as<double>(), S.i1.as<double>(), S.i2.as<double>(), S.th2.as<float>(),
               S.err.as<double>(), S.inl.as<uint8_t>(), S.nin.as<int>() };
    sim3_launch(st, D);
    CCM_HIP(c, hipGetLastError());
    CCM_HIP(c, hipMemcpyAsync(pb->sim3, S.sim3.p, (size_t)F * 64, hipMemcpyDeviceToHost, st));
    if (T) CCM_HIP(c, hipMemcpyAsync(pb->inlier, S.inl.p, T, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipMemcpyAsync(pb->n_inliers, S.nin.p, (size_t)F * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));
    return CCM_OK;
}
