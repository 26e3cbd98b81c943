// ess_host.cpp -- C ABI of the 7-DoF pose-graph optimisation inside Optimizer::OptimizeEssentialGraphLoopClosure /
// OptimizeEssentialGraphMapFusion (cslam/src/Optimizer.cpp:1064-1331, :1333-1574) and of the map point correction
// that follows it (:1300-1330).  Levenberg-Marquardt as g2o runs it (userLambdaInit 1e-16, :1073); the normal
// equations are assembled dense on the device and factorised by rocSOLVER (the reference's sparse Cholesky solves
// the same system; at a few thousand keyframes the dense factorisation is tens of milliseconds).
#include "ccm_internal.h"
#include <rocsolver/rocsolver.h>
#include <algorithm>
#include <cfloat>
#include <cmath>

void ess_launch_errors(hipStream_t, int ne, const int* ei, const int* ej, const double* meas, const double* sim3, const uint8_t* fixed, int fix_scale,
                       int variants, double* err);
void ess_launch_system(hipStream_t, int ne, int nv, const int* ei, const int* ej, const int* fidx, const int* inc_ptr, const int* inc_list,
                       const double* err, double* blocks, double* grad, int N, double* H, double* b);
void ess_launch_update(hipStream_t, int nv, const int* fidx, const double* x, int fix_scale, double* sim3);
void ess_launch_chi2(hipStream_t, int ne, const double* err, int stride, double* part, double* out);
void ess_launch_add_lambda(hipStream_t, int N, double lambda, double* H);
void ess_launch_correct(hipStream_t, int np, const int* ref, const double* s_old, const double* s_new, double* pts);

struct EssState {
    rocblas_handle blas = nullptr;
    DevBuf sim3, save, fixed, ei, ej, meas, fidx, incp, incl, err, blocks, grad, H, Hs, b, x, part, scal, info;
    DevBuf pts, ref, sold, snew;
};
void ess_state_free(EssState* s)
{
    if (!s) return;
    if (s->blas) (void)rocblas_destroy_handle(s->blas);
    DevBuf* all[] = { &s->sim3, &s->save, &s->fixed, &s->ei, &s->ej, &s->meas, &s->fidx, &s->incp, &s->incl, &s->err, &s->blocks, &s->grad, &s->H, &s->Hs,
                      &s->b, &s->x, &s->part, &s->scal, &s->info, &s->pts, &s->ref, &s->sold, &s->snew };
    for (DevBuf* b : all) b->release();
    delete s;
}

extern "C" int ccm_optimize_essential_graph(ccm_ctx* c, ccm_essential_graph* g)
{
    RoctxRange roctx_("ccm_optimize_essential_graph");
    if (!c || !g) return CCM_E_ARG;
    g->iterations_done = 0; g->chi2_initial = 0; g->chi2_final = 0;
    const int nv = g->n_vertices, ne = g->n_edges;
    if (nv < 0 || ne < 0 || (nv > 0 && (!g->sim3 || !g->fixed)) || (ne > 0 && (!g->edge_i || !g->edge_j || !g->measurement)))
        return ccm_fail(c, CCM_E_ARG, "bad essential graph");
    for (int k = 0; k < ne; k++)
        if (g->edge_i[k] < 0 || g->edge_i[k] >= nv || g->edge_j[k] < 0 || g->edge_j[k] >= nv || g->edge_i[k] == g->edge_j[k])
            return ccm_fail(c, CCM_E_ARG, "essential graph edge %d: bad vertex ids", k);
    std::vector<int> fidx(std::max(nv, 1), -1), incp(nv + 1, 0), incl(std::max(2 * ne, 1));
    int nf = 0;
    for (int v = 0; v < nv; v++) if (!g->fixed[v]) fidx[v] = nf++;
    const int N = 7 * nf;
    if (N == 0 || ne == 0 || g->iterations <= 0) return CCM_OK;
    if ((size_t)N * N * 8 > ((size_t)48 << 30)) return ccm_fail(c, CCM_E_ARG, "essential graph too large for the dense solve (%d free keyframes)", nf);
    for (int k = 0; k < ne; k++) { incp[g->edge_i[k] + 1]++; incp[g->edge_j[k] + 1]++; }
    for (int v = 0; v < nv; v++) incp[v + 1] += incp[v];
    {
        std::vector<int> fill(incp.begin(), incp.end() - 1);
        for (int k = 0; k < ne; k++) { incl[fill[g->edge_i[k]]++] = (k << 1); incl[fill[g->edge_j[k]]++] = (k << 1) | 1; }   // edge order per vertex
    }
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->ess) c->ess = new EssState();
    EssState& S = *c->ess;
    hipStream_t st = c->stream;
    if (!S.blas) {
        if (rocblas_create_handle(&S.blas) != rocblas_status_success) { S.blas = nullptr; return ccm_fail(c, CCM_E_DEVICE, "rocblas_create_handle failed"); }
        rocblas_set_stream(S.blas, st);
    }
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(S.sim3, g->sim3, (size_t)nv * 64))) return rc;
    if ((rc = up(S.fixed, g->fixed, (size_t)nv))) return rc;
    if ((rc = up(S.ei, g->edge_i, (size_t)ne * 4))) return rc;
    if ((rc = up(S.ej, g->edge_j, (size_t)ne * 4))) return rc;
    if ((rc = up(S.meas, g->measurement, (size_t)ne * 64))) return rc;
    if ((rc = up(S.fidx, fidx.data(), (size_t)nv * 4))) return rc;
    if ((rc = up(S.incp, incp.data(), ((size_t)nv + 1) * 4))) return rc;
    if ((rc = up(S.incl, incl.data(), (size_t)2 * ne * 4))) return rc;
    const int nbp = (ne + 255) / 256;
    CCM_RESERVE(c, S.save, (size_t)nv * 64); CCM_RESERVE(c, S.err, (size_t)ne * 29 * 56); CCM_RESERVE(c, S.blocks, (size_t)ne * 147 * 8);
    CCM_RESERVE(c, S.grad, (size_t)ne * 14 * 8); CCM_RESERVE(c, S.H, (size_t)N * N * 8); CCM_RESERVE(c, S.Hs, (size_t)N * N * 8);
    CCM_RESERVE(c, S.b, (size_t)N * 8); CCM_RESERVE(c, S.x, (size_t)N * 8); CCM_RESERVE(c, S.part, (size_t)nbp * 8); CCM_RESERVE(c, S.scal, 64);
    CCM_RESERVE(c, S.info, 16);
    double* H = S.H.as<double>(); double* Hs = S.Hs.as<double>();
    const int fs = g->fix_scale ? 1 : 0;
    std::vector<double> bh(N), xh(N);
    double lambda = 0, ni = 2;
    int nBad = 0;
    double cur = 0;
    for (int it = 0; it < g->iterations; it++) {
        // computeActiveErrors + buildSystem
        ess_launch_errors(st, ne, S.ei.as<int>(), S.ej.as<int>(), S.meas.as<double>(), S.sim3.as<double>(), S.fixed.as<uint8_t>(), fs, 29, S.err.as<double>());
        ess_launch_chi2(st, ne, S.err.as<double>(), 29 * 7, S.part.as<double>(), S.scal.as<double>());
        CCM_HIP(c, hipMemsetAsync(H, 0, (size_t)N * N * 8, st));
        ess_launch_system(st, ne, nv, S.ei.as<int>(), S.ej.as<int>(), S.fidx.as<int>(), S.incp.as<int>(), S.incl.as<int>(), S.err.as<double>(),
                          S.blocks.as<double>(), S.grad.as<double>(), N, H, S.b.as<double>());
        CCM_HIP(c, hipGetLastError());
        CCM_HIP(c, hipMemcpyAsync(&cur, S.scal.p, 8, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipMemcpyAsync(bh.data(), S.b.p, (size_t)N * 8, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        if (it == 0) { g->chi2_initial = cur; lambda = 1e-16; ni = 2; nBad = 0; }          // setUserLambdaInit(1e-16)
        const double ini = cur;
        double rho = 0;
        int qmax = 0;
        do {
            CCM_HIP(c, hipMemcpyAsync(S.save.p, S.sim3.p, (size_t)nv * 64, hipMemcpyDeviceToDevice, st));   // push()
            CCM_HIP(c, hipMemcpyAsync(Hs, H, (size_t)N * N * 8, hipMemcpyDeviceToDevice, st));
            ess_launch_add_lambda(st, N, lambda, Hs);
            CCM_HIP(c, hipMemcpyAsync(S.x.p, S.b.p, (size_t)N * 8, hipMemcpyDeviceToDevice, st));
            int* info_dev = S.info.as<int>();
            if (rocsolver_dpotrf(S.blas, rocblas_fill_lower, (rocblas_int)N, Hs, (rocblas_int)N, info_dev) != rocblas_status_success)
                return ccm_fail(c, CCM_E_DEVICE, "rocsolver_dpotrf failed");
            int info = 0;
            CCM_HIP(c, hipMemcpyAsync(&info, info_dev, 4, hipMemcpyDeviceToHost, st));
            CCM_HIP(c, hipStreamSynchronize(st));
            const bool ok2 = info == 0;
            double temp = DBL_MAX;
            std::fill(xh.begin(), xh.end(), 0.0);
            if (ok2) {
                if (rocsolver_dpotrs(S.blas, rocblas_fill_lower, (rocblas_int)N, 1, Hs, (rocblas_int)N, S.x.as<double>(), (rocblas_int)N) != rocblas_status_success)
                    return ccm_fail(c, CCM_E_DEVICE, "rocsolver_dpotrs failed");
                ess_launch_update(st, nv, S.fidx.as<int>(), S.x.as<double>(), fs, S.sim3.as<double>());
                ess_launch_errors(st, ne, S.ei.as<int>(), S.ej.as<int>(), S.meas.as<double>(), S.sim3.as<double>(), S.fixed.as<uint8_t>(), fs, 1, S.err.as<double>());
                ess_launch_chi2(st, ne, S.err.as<double>(), 7, S.part.as<double>(), S.scal.as<double>());
                CCM_HIP(c, hipGetLastError());
                CCM_HIP(c, hipMemcpyAsync(&temp, S.scal.p, 8, hipMemcpyDeviceToHost, st));
                CCM_HIP(c, hipMemcpyAsync(xh.data(), S.x.p, (size_t)N * 8, hipMemcpyDeviceToHost, st));
                CCM_HIP(c, hipStreamSynchronize(st));
            }
            double scale = 1e-3;
            for (int j = 0; j < N; j++) scale += xh[j] * (lambda * xh[j] + bh[j]);
            rho = ok2 ? (cur - temp) / scale : -1.0;
            if (rho > 0 && std::isfinite(temp)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha); ni = 2; cur = temp;
            } else {
                lambda *= ni; ni *= 2;
                CCM_HIP(c, hipMemcpyAsync(S.sim3.p, S.save.p, (size_t)nv * 64, hipMemcpyDeviceToDevice, st));   // pop()
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        g->iterations_done = it + 1;
        if (qmax == 10 || rho == 0) break;
        if ((ini - cur) * 1e3 < ini) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    g->chi2_final = cur;
    CCM_HIP(c, hipMemcpyAsync(g->sim3, S.sim3.p, (size_t)nv * 64, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));
    return CCM_OK;
}

extern "C" int ccm_correct_map_points(ccm_ctx* c, int n_points, double* points, const int32_t* ref_vertex, int n_vertices, const double* sim3_before,
                                      const double* sim3_after)
{
    if (!c) return CCM_E_ARG;
    if (n_points == 0) return CCM_OK;
    if (n_points < 0 || n_vertices <= 0 || !points || !ref_vertex || !sim3_before || !sim3_after) return ccm_fail(c, CCM_E_ARG, "bad map point correction");
    for (int i = 0; i < n_points; i++) if (ref_vertex[i] >= n_vertices) return ccm_fail(c, CCM_E_ARG, "map point %d: reference keyframe out of range", i);
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->ess) c->ess = new EssState();
    EssState& S = *c->ess;
    hipStream_t st = c->stream;
    CCM_RESERVE(c, S.pts, (size_t)n_points * 24); CCM_RESERVE(c, S.ref, (size_t)n_points * 4);
    CCM_RESERVE(c, S.sold, (size_t)n_vertices * 64); CCM_RESERVE(c, S.snew, (size_t)n_vertices * 64);
    CCM_HIP(c, hipMemcpyAsync(S.pts.p, points, (size_t)n_points * 24, hipMemcpyHostToDevice, st));
    CCM_HIP(c, hipMemcpyAsync(S.ref.p, ref_vertex, (size_t)n_points * 4, hipMemcpyHostToDevice, st));
    CCM_HIP(c, hipMemcpyAsync(S.sold.p, sim3_before, (size_t)n_vertices * 64, hipMemcpyHostToDevice, st));
    CCM_HIP(c, hipMemcpyAsync(S.snew.p, sim3_after, (size_t)n_vertices * 64, hipMemcpyHostToDevice, st));
    ess_launch_correct(st, n_points, S.ref.as<int>(), S.sold.as<double>(), S.snew.as<double>(), S.pts.as<double>());
    CCM_HIP(c, hipGetLastError());
    CCM_HIP(c, hipMemcpyAsync(points, S.pts.p, (size_t)n_points * 24, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));
    return CCM_OK;
}
