// ba_host.cpp -- Levenberg-Marquardt driver of the reprojection BA and its C ABI (include/ccm_hot.h).
//
// Mirrors, step for step, what the reference runs through g2o for Optimizer::BundleAdjustmentClient,
// LocalBundleAdjustmentClient and MapFusionGBA (src/Optimizer.cpp:32-212, 349-644, 646-865):
//   SparseOptimizer::optimize            cslam/thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-419
//   OptimizationAlgorithmLevenberg::solve  .../core/optimization_algorithm_levenberg.cpp:61-189
//   BlockSolver<6,3>::buildSystem/solve  .../core/block_solver.hpp:354-486, 502-604
// The reduced camera system is solved by dense Cholesky (rocSOLVER potrf/potrs) in place of
// LinearSolverEigen's SimplicialLDLT (solvers/linear_solver_eigen.h:106-136): any exact SPD solve is
// equivalent up to rounding.  With an RCCL communicator attached (ccm_comm_init) the landmarks are
// sharded over the ranks and the reduced system is summed with one all-reduce per LM trial.
#include "ccm_internal.h"
#include "ba_types.h"
#include "ba_math.h"
#include <rocsolver/rocsolver.h>
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <numeric>

int comm_ranks(const ccm_ctx* c);
int comm_rank(const ccm_ctx* c);
int comm_allreduce_f64(ccm_ctx* c, double* dev, size_t n, bool max_op);

void ba_launch_pose_rt(hipStream_t, const BaDev&);
int ba_errors_blocks(const BaDev&);
void ba_launch_errors(hipStream_t, const BaDev&, double hd, double* partial, double* out);
void ba_launch_linearize(hipStream_t, const BaDev&, double hd);
void ba_launch_init_reduced(hipStream_t, const BaDev&, double lambda_diag);
void ba_launch_add_diag(hipStream_t, const BaDev&, double v);
void ba_launch_schur(hipStream_t, const BaDev&, double lambda);
void ba_launch_backsub(hipStream_t, const BaDev&);
void ba_launch_update(hipStream_t, const BaDev&);
int ba_scale_blocks(const BaDev&);
void ba_launch_scale(hipStream_t, const BaDev&, double lambda, int add_pose_lambda, double* partial, double* out);
void ba_launch_diag(hipStream_t, const BaDev&, double* tmp_ll, double* pp_diag, double* out_ll_max);
void ba_launch_outliers(hipStream_t, const BaDev&, double th, uint8_t* flag);
void ba_launch_deactivate(hipStream_t, const BaDev&, const uint8_t* flag);

struct BaState {
    rocblas_handle blas = nullptr;
    DevBuf poses, Rt, intr, free_of, pose_of_free, points, edge_pose, edge_point, obs, info, active, err,
           pt_first, pose_first, pose_edges, Hpp, bp, Hll, bl, Hpl, Dinv, Hs, bs, x, save_poses, save_points,
           partial, scal, flags, info_dev, tmp_ll, pp_diag, gather;
};
void ba_state_free(BaState* s)
{
    if (!s) return;
    if (s->blas) (void)rocblas_destroy_handle(s->blas);
    DevBuf* all[] = { &s->poses, &s->Rt, &s->intr, &s->free_of, &s->pose_of_free, &s->points, &s->edge_pose, &s->edge_point,
                      &s->obs, &s->info, &s->active, &s->err, &s->pt_first, &s->pose_first, &s->pose_edges, &s->Hpp, &s->bp,
                      &s->Hll, &s->bl, &s->Hpl, &s->Dinv, &s->Hs, &s->bs, &s->x, &s->save_poses, &s->save_points,
                      &s->partial, &s->scal, &s->flags, &s->info_dev, &s->tmp_ll, &s->pp_diag, &s->gather };
    for (DevBuf* b : all) b->release();
    delete s;
}

namespace {
using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

template <class T> int upload(ccm_ctx* c, DevBuf& b, const T* src, size_t n)
{
    CCM_RESERVE(c, b, std::max<size_t>(n * sizeof(T), 16));
    if (n) CCM_HIP(c, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return CCM_OK;
}
}  // namespace

extern "C" {

int ccm_pose_from_mat4f(const float* T, double* pose)
{
    if (!T || !pose) return CCM_E_ARG;
    double R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = (double)T[i * 4 + j];   // Converter.cc:40-56
    ba_R_to_quat(R, pose);
    ba_quat_normalize(pose);
    for (int i = 0; i < 3; i++) pose[4 + i] = (double)T[i * 4 + 3];
    return CCM_OK;
}

int ccm_pose_to_mat4f(const double* pose, float* T)
{
    if (!T || !pose) return CCM_E_ARG;
    double R[9];
    ba_quat_to_R(pose, R);                                                                     // Converter.cc:86-93
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)pose[4 + i];
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    return CCM_OK;
}

int ccm_ba_solve(ccm_ctx* c, ccm_ba_problem* pb, const ccm_ba_options* opt, ccm_ba_result* res)
{
    if (!c || !pb || !opt) return CCM_E_ARG;
    if (pb->n_poses <= 0 || pb->n_points < 0 || pb->n_edges < 0 || !pb->poses || !pb->intr ||
        (pb->n_points > 0 && !pb->points) || (pb->n_edges > 0 && (!pb->edge_pose || !pb->edge_point || !pb->obs || !pb->info)))
        return ccm_fail(c, CCM_E_ARG, "bad BA problem");
    const int P = pb->n_poses, Lall = pb->n_points, Eall = pb->n_edges;
    for (int e = 0; e < Eall; e++)
        if (pb->edge_pose[e] < 0 || pb->edge_pose[e] >= P || pb->edge_point[e] < 0 || pb->edge_point[e] >= Lall)
            return ccm_fail(c, CCM_E_ARG, "edge %d references a vertex out of range", e);
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->ba) c->ba = new BaState();
    BaState& S = *c->ba;
    if (!S.blas) {
        if (rocblas_create_handle(&S.blas) != rocblas_status_success) { S.blas = nullptr; return ccm_fail(c, CCM_E_DEVICE, "rocblas_create_handle failed"); }
        rocblas_set_stream(S.blas, c->stream);
    }
    hipStream_t st = c->stream;
    const int ranks = comm_ranks(c), rank = comm_rank(c);
    ccm_ba_result local{};
    if (!res) res = &local;
    uint8_t* outlier_out = res->edge_outlier;
    *res = ccm_ba_result{};
    res->edge_outlier = outlier_out;

    // ---- vertices
    std::vector<int> free_of(P), pose_of_free;
    for (int p = 0; p < P; p++) {
        const bool fx = pb->fixed && pb->fixed[p];
        free_of[p] = fx ? -1 : (int)pose_of_free.size();
        if (!fx) pose_of_free.push_back(p);
    }
    const int nfree = (int)pose_of_free.size();
    const long long n = 6LL * nfree;

    // ---- landmark shard of this rank: contiguous range balanced by Schur cost k(k+1)/2 + k
    std::vector<int> deg(Lall, 0);
    for (int e = 0; e < Eall; e++) deg[pb->edge_point[e]]++;
    int l0 = 0, l1 = Lall;
    if (ranks > 1) {
        double total = 0;
        for (int l = 0; l < Lall; l++) total += 0.5 * deg[l] * (deg[l] + 1) + deg[l] + 1;
        double acc = 0; int r = 0; l0 = 0; l1 = Lall;
        std::vector<int> cut(ranks + 1, Lall); cut[0] = 0;
        for (int l = 0; l < Lall; l++) {
            acc += 0.5 * deg[l] * (deg[l] + 1) + deg[l] + 1;
            while (r + 1 < ranks && acc >= total * (r + 1) / ranks) { cut[++r] = l + 1; }
        }
        l0 = cut[rank]; l1 = cut[rank + 1];
    }
    const int L = l1 - l0;
    // local edges sorted by (landmark, pose); perm[k] = original edge id
    std::vector<int> perm;
    perm.reserve(Eall / ranks + 16);
    for (int e = 0; e < Eall; e++) if (pb->edge_point[e] >= l0 && pb->edge_point[e] < l1) perm.push_back(e);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
        if (pb->edge_point[a] != pb->edge_point[b]) return pb->edge_point[a] < pb->edge_point[b];
        return pb->edge_pose[a] < pb->edge_pose[b];
    });
    const int E = (int)perm.size();
    std::vector<int> e_pose(E), e_pt(E), pt_first(L + 1, 0);
    std::vector<double> e_obs(2 * (size_t)E), e_info(E);
    for (int k = 0; k < E; k++) {
        const int e = perm[k];
        e_pose[k] = pb->edge_pose[e]; e_pt[k] = pb->edge_point[e] - l0;
        e_obs[2 * k] = pb->obs[2 * e]; e_obs[2 * k + 1] = pb->obs[2 * e + 1]; e_info[k] = pb->info[e];
        pt_first[e_pt[k] + 1]++;
    }
    for (int l = 0; l < L; l++) pt_first[l + 1] += pt_first[l];
    std::vector<int> pose_first(nfree + 1, 0), pose_edges;
    for (int k = 0; k < E; k++) if (free_of[e_pose[k]] >= 0) pose_first[free_of[e_pose[k]] + 1]++;
    for (int f = 0; f < nfree; f++) pose_first[f + 1] += pose_first[f];
    pose_edges.resize(pose_first[nfree]);
    {
        std::vector<int> fill(pose_first.begin(), pose_first.end() - 1);
        for (int k = 0; k < E; k++) { const int f = free_of[e_pose[k]]; if (f >= 0) pose_edges[fill[f]++] = k; }
    }

    // ---- device buffers
    int rc;
    if ((rc = upload(c, S.poses, pb->poses, 7 * (size_t)P))) return rc;
    if ((rc = upload(c, S.intr, pb->intr, 4 * (size_t)P))) return rc;
    if ((rc = upload(c, S.free_of, free_of.data(), P))) return rc;
    if ((rc = upload(c, S.pose_of_free, pose_of_free.data(), nfree))) return rc;
    if ((rc = upload(c, S.points, pb->points + 3 * (size_t)l0, 3 * (size_t)L))) return rc;
    if ((rc = upload(c, S.edge_pose, e_pose.data(), E))) return rc;
    if ((rc = upload(c, S.edge_point, e_pt.data(), E))) return rc;
    if ((rc = upload(c, S.obs, e_obs.data(), 2 * (size_t)E))) return rc;
    if ((rc = upload(c, S.info, e_info.data(), E))) return rc;
    if ((rc = upload(c, S.pt_first, pt_first.data(), L + 1))) return rc;
    if ((rc = upload(c, S.pose_first, pose_first.data(), nfree + 1))) return rc;
    if ((rc = upload(c, S.pose_edges, pose_edges.data(), pose_edges.size()))) return rc;
    const size_t nxl = (size_t)n + 3 * (size_t)L;
    CCM_RESERVE(c, S.Rt, 12 * (size_t)P * 8);
    CCM_RESERVE(c, S.active, std::max<size_t>(E, 16)); CCM_RESERVE(c, S.flags, std::max<size_t>(E, 16));
    CCM_RESERVE(c, S.err, std::max<size_t>(2 * (size_t)E * 8, 16));
    CCM_RESERVE(c, S.Hpp, std::max<size_t>(36 * (size_t)nfree * 8, 16)); CCM_RESERVE(c, S.bp, std::max<size_t>((size_t)n * 8, 16));
    CCM_RESERVE(c, S.Hll, std::max<size_t>(9 * (size_t)L * 8, 16)); CCM_RESERVE(c, S.bl, std::max<size_t>(3 * (size_t)L * 8, 16));
    CCM_RESERVE(c, S.Hpl, std::max<size_t>(18 * (size_t)E * 8, 16)); CCM_RESERVE(c, S.Dinv, std::max<size_t>(9 * (size_t)L * 8, 16));
    // Hs and bs are contiguous so that one all-reduce covers both
    CCM_RESERVE(c, S.Hs, ((size_t)n * n + (size_t)n + 8) * 8);
    CCM_RESERVE(c, S.x, std::max<size_t>(nxl * 8, 16));
    CCM_RESERVE(c, S.save_poses, 7 * (size_t)P * 8); CCM_RESERVE(c, S.save_points, std::max<size_t>(3 * (size_t)L * 8, 16));
    const size_t nb_max = (size_t)std::max((E + 255) / 256, (int)((nxl + 255) / 256)) + 8;
    CCM_RESERVE(c, S.partial, nb_max * 8); CCM_RESERVE(c, S.scal, 64 * 8); CCM_RESERVE(c, S.info_dev, 64);
    CCM_RESERVE(c, S.tmp_ll, std::max<size_t>((size_t)L * 8, 16)); CCM_RESERVE(c, S.pp_diag, std::max<size_t>((size_t)n * 8, 16));
    CCM_HIP(c, hipMemsetAsync(S.active.p, 1, std::max(E, 1), st));
    CCM_HIP(c, hipMemsetAsync(S.err.p, 0, std::max<size_t>(2 * (size_t)E * 8, 16), st));
    CCM_HIP(c, hipMemsetAsync(S.x.p, 0, std::max<size_t>(nxl * 8, 16), st));
    CCM_HIP(c, hipStreamSynchronize(st));    // host staging vectors stay alive until here

    BaDev D{};
    D.P = P; D.L = L; D.E = E; D.nfree = nfree;
    D.poses = S.poses.as<double>(); D.Rt = S.Rt.as<double>(); D.intr = S.intr.as<double>();
    D.free_of = S.free_of.as<int>(); D.pose_of_free = S.pose_of_free.as<int>(); D.points = S.points.as<double>();
    D.edge_pose = S.edge_pose.as<int>(); D.edge_point = S.edge_point.as<int>(); D.obs = S.obs.as<double>(); D.info = S.info.as<double>();
    D.active = S.active.as<uint8_t>(); D.err = S.err.as<double>(); D.pt_first = S.pt_first.as<int>();
    D.pose_first = S.pose_first.as<int>(); D.pose_edges = S.pose_edges.as<int>();
    D.Hpp = S.Hpp.as<double>(); D.bp = S.bp.as<double>(); D.Hll = S.Hll.as<double>(); D.bl = S.bl.as<double>();
    D.Hpl = S.Hpl.as<double>(); D.Dinv = S.Dinv.as<double>();
    D.Hs = S.Hs.as<double>(); D.bs = D.Hs + (size_t)n * n; D.x = S.x.as<double>();
    double* scal = S.scal.as<double>();      // [0] chi2, [1] scale, [2] Hll max, ...
    double* partial = S.partial.as<double>();
    int* info_dev = S.info_dev.as<int>();

    auto stop_requested = [&]() { return opt->stop_flag && *opt->stop_flag; };
    // chi2 (+ optionally scale) of the current state, summed over ranks
    auto eval_chi2 = [&](double hd, bool with_scale, double lambda, double* chi, double* scale) -> int {
        ba_launch_pose_rt(st, D);
        if (E > 0) ba_launch_errors(st, D, hd, partial, scal);
        else CCM_HIP(c, hipMemsetAsync(scal, 0, 8, st));
        if (with_scale) ba_launch_scale(st, D, lambda, rank == 0 ? 1 : 0, partial, scal + 1);
        int r = comm_allreduce_f64(c, scal, with_scale ? 2 : 1, false);
        if (r) return r;
        double h[2] = { 0, 0 };
        CCM_HIP(c, hipMemcpyAsync(h, scal, with_scale ? 16 : 8, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        *chi = h[0]; if (scale) *scale = h[1];
        return CCM_OK;
    };

    double huber = opt->huber_delta > 0 ? opt->huber_delta : 0.0;
    bool first_eval = true;
    for (int stage = 0; stage < 2 && !res->stopped; stage++) {
        const int iterations = stage == 0 ? opt->iterations : opt->iterations2;
        if (iterations <= 0) { if (stage == 0) continue; else break; }
        if (stage == 1) {
            // src/Optimizer.cpp:546-563: chi2 > th or non-positive depth -> level 1; every kernel dropped
            ba_launch_pose_rt(st, D);
            if (E > 0) { ba_launch_outliers(st, D, opt->outlier_chi2, S.flags.as<uint8_t>()); ba_launch_deactivate(st, D, S.flags.as<uint8_t>()); }
            huber = 0.0;
        }
        double lambda = 0, ni = 2;
        int nBad = 0;
        for (int it = 0; it < iterations; it++) {
            if (stop_requested()) { res->stopped = 1; break; }                    // !terminate(), sparse_optimizer.cpp:376
            auto t0 = clk::now();
            double currentChi = 0;
            if ((rc = eval_chi2(huber, false, 0, &currentChi, nullptr))) return rc;
            const double iniChi = currentChi;
            if (first_eval) { res->chi2_initial = currentChi; first_eval = false; }
            ba_launch_linearize(st, D, huber);                                      // buildSystem
            if (it == 0) {                                                          // computeLambdaInit
                ba_launch_diag(st, D, S.tmp_ll.as<double>(), S.pp_diag.as<double>(), scal + 2);
                if (L == 0) CCM_HIP(c, hipMemsetAsync(scal + 2, 0, 8, st));
                if ((rc = comm_allreduce_f64(c, scal + 2, 1, true))) return rc;
                if ((rc = comm_allreduce_f64(c, S.pp_diag.as<double>(), (size_t)n, false))) return rc;
                std::vector<double> dg((size_t)n + 1);
                CCM_HIP(c, hipMemcpyAsync(dg.data(), S.pp_diag.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
                CCM_HIP(c, hipMemcpyAsync(&dg[n], scal + 2, 8, hipMemcpyDeviceToHost, st));
                CCM_HIP(c, hipStreamSynchronize(st));
                double md = 0;
                for (double v : dg) md = std::max(md, std::fabs(v));
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            CCM_HIP(c, hipStreamSynchronize(st));
            res->t_linearize += secs(t0, clk::now());
            double rho = 0;
            int qmax = 0;
            do {
                auto t1 = clk::now();
                CCM_HIP(c, hipMemcpyAsync(S.save_poses.p, D.poses, 7 * (size_t)P * 8, hipMemcpyDeviceToDevice, st));   // push
                if (L) CCM_HIP(c, hipMemcpyAsync(S.save_points.p, D.points, 3 * (size_t)L * 8, hipMemcpyDeviceToDevice, st));
                CCM_HIP(c, hipMemsetAsync(D.Hs, 0, ((size_t)n * n + (size_t)n) * 8 + 8, st));
                ba_launch_init_reduced(st, D, 0.0);
                if (L > 0) ba_launch_schur(st, D, lambda);
                if ((rc = comm_allreduce_f64(c, D.Hs, (size_t)n * n + (size_t)n, false))) return rc;
                ba_launch_add_diag(st, D, lambda);
                CCM_HIP(c, hipStreamSynchronize(st));
                auto t2 = clk::now();
                res->t_schur += secs(t1, t2);
                int ok2 = 1;
                if (n > 0) {
                    CCM_HIP(c, hipMemcpyAsync(D.x, D.bs, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
                    // row-major upper triangle == column-major lower triangle
                    if (rocsolver_dpotrf(S.blas, rocblas_fill_lower, (rocblas_int)n, D.Hs, (rocblas_int)n, info_dev) != rocblas_status_success)
                        return ccm_fail(c, CCM_E_DEVICE, "rocsolver_dpotrf failed");
                    int info = 0;
                    CCM_HIP(c, hipMemcpyAsync(&info, info_dev, 4, hipMemcpyDeviceToHost, st));
                    CCM_HIP(c, hipStreamSynchronize(st));
                    ok2 = info == 0;
                    if (ok2 && rocsolver_dpotrs(S.blas, rocblas_fill_lower, (rocblas_int)n, 1, D.Hs, (rocblas_int)n, D.x, (rocblas_int)n) != rocblas_status_success)
                        return ccm_fail(c, CCM_E_DEVICE, "rocsolver_dpotrs failed");
                    CCM_HIP(c, hipStreamSynchronize(st));
                }
                auto t3 = clk::now();
                res->t_solve += secs(t2, t3);
                res->trials++;
                double tempChi = DBL_MAX, scale = 0;
                if (ok2) {
                    if (L > 0) ba_launch_backsub(st, D);
                    ba_launch_update(st, D);
                    if ((rc = eval_chi2(huber, true, lambda, &tempChi, &scale))) return rc;
                }
                scale += 1e-3;
                rho = ok2 ? (currentChi - tempChi) / scale : -1.0;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha);
                    ni = 2; currentChi = tempChi;                                   // discardTop
                } else {
                    lambda *= ni; ni *= 2;
                    CCM_HIP(c, hipMemcpyAsync(D.poses, S.save_poses.p, 7 * (size_t)P * 8, hipMemcpyDeviceToDevice, st));   // pop
                    if (L) CCM_HIP(c, hipMemcpyAsync(D.points, S.save_points.p, 3 * (size_t)L * 8, hipMemcpyDeviceToDevice, st));
                }
                qmax++;
                CCM_HIP(c, hipStreamSynchronize(st));
                res->t_update += secs(t3, clk::now());
            } while (rho < 0 && qmax < 10 && !stop_requested());
            res->iterations_done++;
            res->chi2_final = currentChi; res->lambda_final = lambda;
            if (qmax == 10 || rho == 0) break;                                       // Terminate
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;         // stop criterion :154-161
            if (nBad >= 3) break;
        }
    }

    // ---- results
    CCM_HIP(c, hipMemcpyAsync(pb->poses, D.poses, 7 * (size_t)P * 8, hipMemcpyDeviceToHost, st));
    if (ranks == 1) {
        if (L) CCM_HIP(c, hipMemcpyAsync(pb->points, D.points, 3 * (size_t)L * 8, hipMemcpyDeviceToHost, st));
    } else {
        // every rank fills its landmark range of a zeroed full-size buffer; a sum all-reduce is the all-gather
        CCM_RESERVE(c, S.gather, std::max<size_t>(3 * (size_t)Lall * 8, 16));
        CCM_HIP(c, hipMemsetAsync(S.gather.p, 0, 3 * (size_t)Lall * 8, st));
        if (L) CCM_HIP(c, hipMemcpyAsync(S.gather.as<double>() + 3 * (size_t)l0, D.points, 3 * (size_t)L * 8, hipMemcpyDeviceToDevice, st));
        if ((rc = comm_allreduce_f64(c, S.gather.as<double>(), 3 * (size_t)Lall, false))) return rc;
        CCM_HIP(c, hipMemcpyAsync(pb->points, S.gather.p, 3 * (size_t)Lall * 8, hipMemcpyDeviceToHost, st));
    }
    if (outlier_out) {
        ba_launch_pose_rt(st, D);
        std::vector<uint8_t> fl(std::max(E, 1));
        if (E > 0) {
            ba_launch_outliers(st, D, opt->outlier_chi2, S.flags.as<uint8_t>());
            CCM_HIP(c, hipMemcpyAsync(fl.data(), S.flags.p, E, hipMemcpyDeviceToHost, st));
        }
        CCM_HIP(c, hipStreamSynchronize(st));
        if (ranks == 1) {
            for (int k = 0; k < E; k++) outlier_out[perm[k]] = fl[k];
        } else {
            // flags of the other ranks' edges: exchange as doubles through the same collective
            std::vector<double> full(Eall, 0.0);
            for (int k = 0; k < E; k++) full[perm[k]] = fl[k];
            CCM_RESERVE(c, S.gather, std::max<size_t>((size_t)Eall * 8, 16));
            CCM_HIP(c, hipMemcpyAsync(S.gather.p, full.data(), (size_t)Eall * 8, hipMemcpyHostToDevice, st));
            if ((rc = comm_allreduce_f64(c, S.gather.as<double>(), Eall, false))) return rc;
            CCM_HIP(c, hipMemcpyAsync(full.data(), S.gather.p, (size_t)Eall * 8, hipMemcpyDeviceToHost, st));
            CCM_HIP(c, hipStreamSynchronize(st));
            for (int e = 0; e < Eall; e++) outlier_out[e] = full[e] != 0.0;
        }
    }
    CCM_HIP(c, hipStreamSynchronize(st));
    CCM_HIP(c, hipGetLastError());
    return CCM_OK;
}

}  // extern "C"
