// pose_host.cpp -- C ABI of the batched pose-only optimisation (Optimizer::PoseOptimizationClient,
// src/Optimizer.cpp:215-347); the whole schedule runs in one kernel launch (pose_kernels.hip).
#include "ccm_internal.h"
#include <algorithm>

struct PoseDev {
    int n_frames; double* poses; const double* intr; const int* first; const double* pts; const double* obs;
    const double* info; double* err; uint8_t* outlier; int* n_inliers;
};
void pose_launch(hipStream_t, const PoseDev&);

struct PoseState { DevBuf poses, intr, first, pts, obs, info, err, outlier, ninl; };
static PoseState* pose_state(ccm_ctx* c)
{
    // lives with the BA state's lifetime rules: allocated on first use, freed with the context
    static_assert(sizeof(void*) == 8, "64-bit only");
    if (!c->pose) c->pose = new PoseState();
    return c->pose;
}
void pose_state_free(PoseState* s)
{
    if (!s) return;
    DevBuf* all[] = { &s->poses, &s->intr, &s->first, &s->pts, &s->obs, &s->info, &s->err, &s->outlier, &s->ninl };
    for (DevBuf* b : all) b->release();
    delete s;
}

extern "C" int ccm_pose_optimize(ccm_ctx* c, ccm_pose_problem* pb)
{
    RoctxRange roctx_("ccm_pose_optimize");
    if (!c || !pb) return CCM_E_ARG;
    if (pb->n_frames == 0) return CCM_OK;
    if (pb->n_frames < 0 || !pb->poses || !pb->intr || !pb->first || !pb->n_inliers) return ccm_fail(c, CCM_E_ARG, "bad pose problem");
    const int F = pb->n_frames;
    if (pb->first[0] != 0) return ccm_fail(c, CCM_E_ARG, "first[0] must be 0");
    for (int f = 0; f < F; f++) if (pb->first[f + 1] < pb->first[f]) return ccm_fail(c, CCM_E_ARG, "first[] must be non-decreasing");
    const size_t T = (size_t)pb->first[F];
    if (T > 0 && (!pb->points || !pb->obs || !pb->info || !pb->outlier)) return ccm_fail(c, CCM_E_ARG, "bad pose problem");
    CCM_HIP(c, hipSetDevice(c->device));
    PoseState& S = *pose_state(c);
    hipStream_t st = c->stream;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(S.poses, pb->poses, (size_t)F * 56))) return rc;
    if ((rc = up(S.intr, pb->intr, (size_t)F * 32))) return rc;
    if ((rc = up(S.first, pb->first, ((size_t)F + 1) * 4))) return rc;
    if ((rc = up(S.pts, pb->points, T * 24))) return rc;
    if ((rc = up(S.obs, pb->obs, T * 16))) return rc;
    if ((rc = up(S.info, pb->info, T * 8))) return rc;
    CCM_RESERVE(c, S.err, std::max<size_t>(T * 16, 16)); CCM_RESERVE(c, S.outlier, std::max<size_t>(T, 16));
    CCM_RESERVE(c, S.ninl, (size_t)F * 4);
    PoseDev D{ F, S.poses.as<double>(), S.intr.as<double>(), S.first.as<int>(), S.pts.as<double>(), S.obs.as<double>(),
               S.info.as<double>(), S.err.as<double>(), S.outlier.as<uint8_t>(), S.ninl.as<int>() };
    pose_launch(st, D);
    CCM_HIP(c, hipGetLastError());
    CCM_HIP(c, hipMemcpyAsync(pb->poses, S.poses.p, (size_t)F * 56, hipMemcpyDeviceToHost, st));
    if (T) CCM_HIP(c, hipMemcpyAsync(pb->outlier, S.outlier.p, T, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipMemcpyAsync(pb->n_inliers, S.ninl.p, (size_t)F * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));
    return CCM_OK;
}
