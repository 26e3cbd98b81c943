// ess_host.cpp -- C ABI of the 7-DoF pose-graph optimisation inside Optimizer::OptimizeEssentialGraphLoopClosure /
// OptimizeEssentialGraphMapFusion (cslam/src/Optimizer.cpp:1064-1331, :1333-1574) and of the map point correction
// that follows it (:1300-1330).  Levenberg-Marquardt as g2o runs it (userLambdaInit 1e-16, :1073).  The normal equations
// live block-sparse (7x7 blocks) on the pattern of the graph and are solved by a block-sparse Cholesky factorisation on the
// device, like the reference's BlockSolver_7_3 + sparse LinearSolverEigen (:1072-1074): a fill-reducing ordering computed
// once per call on the host, then per LM trial a numeric factorisation and two triangular solves (ess_kernels.hip, k_essp_*).
// No dense matrix, no library solver: 2000 keyframes / 8000 edges take ~25 MB instead of the 1.57 GB dense system.
#include "ccm_internal.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

void ess_launch_errors(hipStream_t, int ne, const int* ei, const int* ej, const double* meas, const double* sim3, const uint8_t* fixed, int fix_scale,
                       int variants, double* err);
void ess_launch_blocks(hipStream_t, int ne, int nv, const int* fidx, const int* inc_ptr, const int* inc_list, const double* err, double* blocks,
                       double* grad, double* b);
void essp_launch_assemble(hipStream_t, int ntargets, int ncol, const int* aptr, const int* alist, const double* blocks, double* D, double* Lb);
void essp_launch_factor_round(hipStream_t, const int* cols, int n, const int* colptr, int ncol, const int* tptr, const int* tpa, const int* tpb,
                              double lambda, double* D, double* Lb, int* bad);
void essp_launch_forward_round(hipStream_t, const int* cols, int n, const int* perm, const int* rptr, const int* rslot, const int* rcol,
                               const double* D, const double* Lb, const double* b, double* y);
void essp_launch_backward_round(hipStream_t, const int* cols, int n, const int* perm, const int* colptr, const int* rowidx, const double* D,
                                const double* Lb, const double* y, double* xp, double* x);
void ess_launch_update(hipStream_t, int nv, const int* fidx, const double* x, int fix_scale, double* sim3);
void ess_launch_chi2(hipStream_t, int ne, const double* err, int stride, double* part, double* out);
void ess_launch_correct(hipStream_t, int np, const int* ref, const double* s_old, const double* s_new, double* pts);

struct EssState {
    DevBuf sim3, save, fixed, ei, ej, meas, fidx, incp, incl, err, blocks, grad, b, x, part, scal, info;
    DevBuf Da, La, D, Lb, y, xp;                                   // assembled system, its factor (in place), solve vectors
    DevBuf perm, colptr, rowidx, cols, aptr, alist, tptr, tpa, tpb, rptr, rslot, rcol;
    DevBuf pts, ref, sold, snew;
};
void ess_state_free(EssState* s)
{
    if (!s) return;
    DevBuf* all[] = { &s->sim3, &s->save, &s->fixed, &s->ei, &s->ej, &s->meas, &s->fidx, &s->incp, &s->incl, &s->err, &s->blocks, &s->grad,
                      &s->b, &s->x, &s->part, &s->scal, &s->info, &s->Da, &s->La, &s->D, &s->Lb, &s->y, &s->xp, &s->perm, &s->colptr, &s->rowidx,
                      &s->cols, &s->aptr, &s->alist, &s->tptr, &s->tpa, &s->tpb, &s->rptr, &s->rslot, &s->rcol, &s->pts, &s->ref, &s->sold, &s->snew };
    for (DevBuf* b : all) b->release();
    delete s;
}

#include "ess_symbolic.h"

extern "C" int ccm_optimize_essential_graph(ccm_ctx* c, ccm_essential_graph* g)
{
    RoctxRange roctx_("ccm_optimize_essential_graph");
    if (!c || !g) return CCM_E_ARG;
    g->iterations_done = 0; g->chi2_initial = 0; g->chi2_final = 0; g->factor_blocks = 0; g->factor_rounds = 0; g->solver_bytes = 0;
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
    EssSymbolic Y;
    ess_symbolic(nf, ne, g->edge_i, g->edge_j, fidx, Y);
    const int ntargets = nf + Y.nnz, n_rounds = (int)Y.round_ptr.size() - 1;
    g->factor_blocks = ntargets; g->factor_rounds = n_rounds;
    g->solver_bytes = 2LL * ntargets * 49 * 8 + 4LL * (Y.tpa.size() + Y.tpb.size() + Y.alist.size() + Y.rslot.size() + Y.rcol.size() + Y.rowidx.size() + 4 * (size_t)ntargets);
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
    auto upi = [&](DevBuf& b, const std::vector<int>& v) -> int { return up(b, v.data(), v.size() * 4); };
    if ((rc = upi(S.perm, Y.perm)) || (rc = upi(S.colptr, Y.colptr)) || (rc = upi(S.rowidx, Y.rowidx)) || (rc = upi(S.cols, Y.cols)) ||
        (rc = upi(S.aptr, Y.aptr)) || (rc = upi(S.alist, Y.alist)) || (rc = upi(S.tptr, Y.tptr)) || (rc = upi(S.tpa, Y.tpa)) || (rc = upi(S.tpb, Y.tpb)) ||
        (rc = upi(S.rptr, Y.rptr)) || (rc = upi(S.rslot, Y.rslot)) || (rc = upi(S.rcol, Y.rcol))) return rc;
    const int nbp = (ne + 255) / 256;
    const size_t dbytes = (size_t)nf * 49 * 8, lbytes = std::max<size_t>((size_t)Y.nnz * 49 * 8, 16);
    CCM_RESERVE(c, S.save, (size_t)nv * 64); CCM_RESERVE(c, S.err, (size_t)ne * 29 * 56); CCM_RESERVE(c, S.blocks, (size_t)ne * 147 * 8);
    CCM_RESERVE(c, S.grad, (size_t)ne * 14 * 8); CCM_RESERVE(c, S.Da, dbytes); CCM_RESERVE(c, S.La, lbytes); CCM_RESERVE(c, S.D, dbytes); CCM_RESERVE(c, S.Lb, lbytes);
    CCM_RESERVE(c, S.b, (size_t)N * 8); CCM_RESERVE(c, S.x, (size_t)N * 8); CCM_RESERVE(c, S.y, (size_t)N * 8); CCM_RESERVE(c, S.xp, (size_t)N * 8);
    CCM_RESERVE(c, S.part, (size_t)nbp * 8); CCM_RESERVE(c, S.scal, 64); CCM_RESERVE(c, S.info, 16);
    const int fs = g->fix_scale ? 1 : 0;
    std::vector<double> bh(N), xh(N);
    double lambda = 0, ni = 2;
    int nBad = 0;
    double cur = 0;
    for (int it = 0; it < g->iterations; it++) {
        // computeActiveErrors + buildSystem
        ess_launch_errors(st, ne, S.ei.as<int>(), S.ej.as<int>(), S.meas.as<double>(), S.sim3.as<double>(), S.fixed.as<uint8_t>(), fs, 29, S.err.as<double>());
        ess_launch_chi2(st, ne, S.err.as<double>(), 29 * 7, S.part.as<double>(), S.scal.as<double>());
        ess_launch_blocks(st, ne, nv, S.fidx.as<int>(), S.incp.as<int>(), S.incl.as<int>(), S.err.as<double>(), S.blocks.as<double>(),
                          S.grad.as<double>(), S.b.as<double>());
        essp_launch_assemble(st, ntargets, nf, S.aptr.as<int>(), S.alist.as<int>(), S.blocks.as<double>(), S.Da.as<double>(), S.La.as<double>());
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
            // numeric factorisation of (H + lambda I), one launch per round of independent columns
            CCM_HIP(c, hipMemcpyAsync(S.D.p, S.Da.p, dbytes, hipMemcpyDeviceToDevice, st));
            if (Y.nnz) CCM_HIP(c, hipMemcpyAsync(S.Lb.p, S.La.p, (size_t)Y.nnz * 49 * 8, hipMemcpyDeviceToDevice, st));
            int* info_dev = S.info.as<int>();
            CCM_HIP(c, hipMemsetAsync(info_dev, 0, 4, st));
            for (int r = 0; r < n_rounds; r++)
                essp_launch_factor_round(st, S.cols.as<int>() + Y.round_ptr[r], Y.round_ptr[r + 1] - Y.round_ptr[r], S.colptr.as<int>(), nf, S.tptr.as<int>(),
                                         S.tpa.as<int>(), S.tpb.as<int>(), lambda, S.D.as<double>(), S.Lb.as<double>(), info_dev);
            int info = 0;
            CCM_HIP(c, hipMemcpyAsync(&info, info_dev, 4, hipMemcpyDeviceToHost, st));
            CCM_HIP(c, hipStreamSynchronize(st));
            const bool ok2 = info == 0;
            double temp = DBL_MAX;
            std::fill(xh.begin(), xh.end(), 0.0);
            if (ok2) {
                for (int r = 0; r < n_rounds; r++)
                    essp_launch_forward_round(st, S.cols.as<int>() + Y.round_ptr[r], Y.round_ptr[r + 1] - Y.round_ptr[r], S.perm.as<int>(), S.rptr.as<int>(),
                                              S.rslot.as<int>(), S.rcol.as<int>(), S.D.as<double>(), S.Lb.as<double>(), S.b.as<double>(), S.y.as<double>());
                for (int r = n_rounds - 1; r >= 0; r--)
                    essp_launch_backward_round(st, S.cols.as<int>() + Y.round_ptr[r], Y.round_ptr[r + 1] - Y.round_ptr[r], S.perm.as<int>(), S.colptr.as<int>(),
                                               S.rowidx.as<int>(), S.D.as<double>(), S.Lb.as<double>(), S.y.as<double>(), S.xp.as<double>(), S.x.as<double>());
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
