// ba_host.cpp -- Levenberg-Marquardt driver of the reprojection BA and its C ABI (include/ccm_hot.h).
//
// Mirrors, step for step, what the reference runs through g2o for Optimizer::BundleAdjustmentClient,
// LocalBundleAdjustmentClient and MapFusionGBA (src/Optimizer.cpp:32-212, 349-644, 646-865):
//   SparseOptimizer::optimize            cslam/thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-419
//   OptimizationAlgorithmLevenberg::solve  .../core/optimization_algorithm_levenberg.cpp:61-189
//   BlockSolver<6,3>::buildSystem/solve  .../core/block_solver.hpp:354-486, 502-604
// The reduced camera system is solved by a dense inverse (in-house block Gauss-Jordan, ba_sparse.hip) for small maps and
// by a two-level preconditioned CG on the packed blocks for large ones, in place of LinearSolverEigen's SimplicialLDLT
// (solvers/linear_solver_eigen.h:106-136): any SPD solve of sufficient accuracy is equivalent up to rounding.  With an RCCL communicator attached (ccm_comm_init) the landmarks are
// sharded over the ranks and the reduced system is summed with one all-reduce per LM trial.
#include "ccm_internal.h"
#include "ba_types.h"
#include "ba_math.h"
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <memory>
#include <numeric>
#include <new>
#include <system_error>
#include <thread>

int comm_ranks(const ccm_ctx* c);
int comm_rank(const ccm_ctx* c);
int comm_allreduce_f64(ccm_ctx* c, double* dev, size_t n, bool max_op);

void ba_launch_pose_rt(hipStream_t, const BaDev&);
int ba_errors_blocks(const BaDev&);
void ba_launch_errors(hipStream_t, const BaDev&, double hd, double* partial, double* out);
void ba_launch_linearize(hipStream_t, const BaDev&, double hd, double lambda, bool keep_hpl, bool landmarks_only);
void ba_launch_lin_pose(hipStream_t, const BaDev&, double hd);
int comm_allreduce_u8_max(ccm_ctx* c, uint8_t* dev, size_t n);
// ba_sparse.hip
size_t sp_scan_temp_bytes(size_t n);
hipError_t sp_scan_int(hipStream_t, void* tmp, size_t tmp_bytes, const int* in, int* out, size_t n);
hipError_t sp_scan_flags(hipStream_t, void* tmp, size_t tmp_bytes, const uint8_t* in, int* out, size_t n);
size_t sp_sort_temp_bytes(size_t n);
hipError_t sp_sort_u64(hipStream_t, void* tmp, size_t tmp_bytes, const unsigned* kin, unsigned* kout, const unsigned long long* vin,
                       unsigned long long* vout, size_t n, int bits);
hipError_t sp_sort_u32(hipStream_t, void* tmp, size_t tmp_bytes, const unsigned* kin, unsigned* kout, const unsigned* vin, unsigned* vout, size_t n);
void sp_launch_pair_count(hipStream_t, const BaDev&, int* cnt);
void sp_launch_pair_fill(hipStream_t, const BaDev&, const int* off, unsigned* key, unsigned long long* val);
void sp_launch_mark(hipStream_t, const unsigned* key, long long np, int nfree, uint8_t* map);
void sp_launch_block_coords(hipStream_t, const uint8_t* map, const int* id, long long n2, int nfree, int* br, int* bc, int* diag);
void sp_launch_pair_block(hipStream_t, const unsigned* key, const int* id, long long np, unsigned* out);
void sp_launch_seg_bounds(hipStream_t, const unsigned* sk, long long np, int* st, int* en);
void sp_launch_row_entries(hipStream_t, const int* br, const int* bc, int nb, int nfree, unsigned* key, unsigned* val);
void sp_launch_row_ptr(hipStream_t, const unsigned* skey, int n_ent, int nfree, int* row_ptr);
void sp_launch_dinv(hipStream_t, const BaDev&, double lambda);
void sp_launch_schur_blocks(hipStream_t, const BaDev&, const double* Y, const unsigned long long* pairs, const int* st, const int* en,
                            const int* br, const int* bc, int nb, double* Hb);
void sp_launch_bschur(hipStream_t, const BaDev&, double* bs);
void sp_launch_add_lambda(hipStream_t, const int* diag, int nfree, double lambda, double* Hb);
void sp_launch_to_dense(hipStream_t, const double* Hb, const int* br, const int* bc, int nb, long long n, double* Hs);
int dense_small_max();
int dense_launch_small_solve(hipStream_t, const double* Hb, const int* blk_row, const int* blk_col, int nb, int n, const double* b, double* x, int* bad, double lambda);
size_t pcg_minv_bytes(int nfree);
hipError_t pcg_launch_minv(hipStream_t, const double* Hb, const int* blk_row, const int* blk_col, int nb, int nfree, double* Minv, int* bad);
void pcg_launch_init(hipStream_t, const double* b, const double* Minv, int nfree, double* w, double* part, double* sc, const PcgCoarse& C);
void pcg_launch_iter(hipStream_t, const double* Hb, const int* row_ptr, const unsigned* ekey, const unsigned* eval, const double* Minv,
                     int nfree, double* w, double* pap_part, double* part, double* sc, int parity, const PcgCoarse& C);
void pcg_launch_publish(hipStream_t, int nfree, double* part, double* sc, const PcgCoarse& C);
size_t ppcg_state_doubles(int nfree);
size_t ppcg_ca_doubles(int nfree);
bool ppcg_supported(int nfree);
void ppcg_launch_expand(hipStream_t, const double* Hb, const unsigned* ekey, const unsigned* eval, int n_ent, int nfree, double* Hf, int* ecol);
hipError_t ppcg_launch_init(hipStream_t, const double* b, const double* Minv, const int* row_ptr, int nfree, double* wb, double* part, double* sc,
                            const PcgCoarse& C, const PpcgBufs& B);
void ppcg_launch_iter(hipStream_t, const double* Minv, const int* row_ptr, int nfree, double* wb, double* part, double* sc, const PcgCoarse& C, const PpcgBufs& B);
void ppcg_launch_publish(hipStream_t, const double* part, int nfree, double* sc);
int pcg_coarse_dim(int nfree);
int pcg_coarse_parts(int nfree);
void pcg_launch_coarse_mark(hipStream_t, const int* blk_row, const int* blk_col, int nb, int nfree, uint8_t* aggmap);
hipError_t pcg_launch_coarse_build(hipStream_t, const double* Hb, const uint8_t* map, const int* id, int nfree, const double* svec, const double* cen,
                                   const int* pairs, int npairs, double* Ac);
int pcg_coarse_aggregates(int nfree);
int pcg_coarse_agg_keyframes(int nfree);
void pcg_launch_coarse_mirror(hipStream_t, double* A, int ncp);
void pcg_launch_coarse_invert(hipStream_t, double* A, int ncp, double* D, int* bad);
int pcg_coarse_pitch(int nfree);
int dense_pitch(long long n);
void dense_launch_solve(hipStream_t, double* A, int n, int lda, const double* b, double* x, int* bad);
void ba_launch_backsub(hipStream_t, const BaDev&, double lambda);
void ba_launch_index_check(hipStream_t, const int* edge_pose, const int* edge_point, int E, int P, int L, int* flags, int* pt_first);
void ba_launch_index_pose_keys(hipStream_t, const int* edge_pose, const int* free_of, int E, int P, int nfree, unsigned* key, unsigned* val);
void ba_launch_index_pose_first(hipStream_t, const unsigned* skey, int E, int nfree, int* pose_first);
void ba_launch_update(hipStream_t, const BaDev&, double* save_poses, double* save_points);
int ba_scale_blocks(const BaDev&);
void ba_launch_scale(hipStream_t, const BaDev&, double lambda, int add_pose_lambda, double* partial, double* out);
void ba_launch_errors_scale(hipStream_t, const BaDev&, double hd, double lambda, int add_pose_lambda, double* partial, double* out);
void ba_launch_diag(hipStream_t, const BaDev&, double* tmp_ll, double* pp_diag, double* out_ll_max);
void ba_launch_outliers(hipStream_t, const BaDev&, double th, uint8_t* flag);
void ba_launch_deactivate(hipStream_t, const BaDev&, const uint8_t* flag);

struct BaState {
    hipStream_t side = nullptr;            // coarse-level inversion, concurrent with the PCG of the current trial
    hipEvent_t ev_hb = nullptr;            // the side stream has finished reading the reduced system
    hipEvent_t ev_inv = nullptr;           // the side stream has finished an inversion
    hipEvent_t ev_up = nullptr;            // the side stream has finished uploading observations, information values and points
    hipEvent_t ev_copy = nullptr;          // the main stream has copied the last inverse out of the side stream's work matrix
    hipEvent_t ev_chi = nullptr;           // a trial's chi2 has reached the host (the stream goes on with the next iteration's pose-side linearisation)
    std::vector<hipEvent_t> clock_ev;      // phase timers of large problems (events instead of stream synchronisations)
    double* pinned = nullptr;          // 16 doubles of page-locked host memory for small device->host reads
    DevBuf poses, Rt, intr, free_of, pose_of_free, points, edge_pose, edge_point, obs, info, active, err,
           pt_first, pose_first, pose_edges, Hpp, bp, Hpp2, bp2, Hll, bl, Hpl, Dinv, Hs, bs, x, save_poses, save_points,
           partial, scal, flags, info_dev, tmp_ll, pp_diag, gather,
           sp_cnt, sp_off, sp_key, sp_val, sp_key2, sp_val2, sp_map, sp_id, sp_tmp, blk_row, blk_col, diag_id, seg_start, seg_end,
           ent_key, ent_val, ent_key2, ent_val2, row_ptr, Hb, Y, db, Minv, pcg_w, pcg_pap, pcg_part, pcg_sc, pcg_aci, pcg_coarse, pcg_acw, pcg_svec, pcg_hf, pcg_ecol, pcg_ca, pcg_aggmap, pcg_pairs, ce;
};
void ba_state_free(BaState* s)
{
    if (!s) return;
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->ev_hb) (void)hipEventDestroy(s->ev_hb);
    if (s->ev_inv) (void)hipEventDestroy(s->ev_inv);
    if (s->ev_up) (void)hipEventDestroy(s->ev_up);
    if (s->ev_copy) (void)hipEventDestroy(s->ev_copy);
    if (s->ev_chi) (void)hipEventDestroy(s->ev_chi);
    for (hipEvent_t e : s->clock_ev) (void)hipEventDestroy(e);
    if (s->pinned) (void)hipHostFree(s->pinned);
    DevBuf* all[] = { &s->poses, &s->Rt, &s->intr, &s->free_of, &s->pose_of_free, &s->points, &s->edge_pose, &s->edge_point,
                      &s->obs, &s->info, &s->active, &s->err, &s->pt_first, &s->pose_first, &s->pose_edges, &s->Hpp, &s->bp, &s->Hpp2, &s->bp2,
                      &s->Hll, &s->bl, &s->Hpl, &s->Dinv, &s->Hs, &s->bs, &s->x, &s->save_poses, &s->save_points,
                      &s->partial, &s->scal, &s->flags, &s->info_dev, &s->tmp_ll, &s->pp_diag, &s->gather,
                      &s->sp_cnt, &s->sp_off, &s->sp_key, &s->sp_val, &s->sp_key2, &s->sp_val2, &s->sp_map, &s->sp_id, &s->sp_tmp, &s->blk_row,
                      &s->blk_col, &s->diag_id, &s->seg_start, &s->seg_end, &s->ent_key, &s->ent_val, &s->ent_key2, &s->ent_val2, &s->row_ptr,
                      &s->Hb, &s->Y, &s->db, &s->Minv, &s->pcg_w, &s->pcg_pap, &s->pcg_part, &s->pcg_sc, &s->pcg_aci, &s->pcg_coarse, &s->pcg_acw, &s->pcg_svec, &s->pcg_hf, &s->pcg_ecol, &s->pcg_ca, &s->pcg_aggmap, &s->pcg_pairs, &s->ce };
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

// Landmark ranges of the ranks of a sharded global BA: contiguous, balanced by the Schur cost
// k(k+1)/2 + k + 1 of a k-observation landmark.  cuts[r] .. cuts[r+1]-1 belong to rank r.  Host only.
int ccm_ba_landmark_cuts(const int32_t* edge_point, int n_edges, int n_points, int n_ranks, int32_t* cuts)
{
    if (!cuts || n_ranks < 1 || n_points < 0 || n_edges < 0 || (n_edges > 0 && !edge_point)) return CCM_E_ARG;
    std::vector<int> deg(n_points, 0);
    for (int e = 0; e < n_edges; e++) {
        if (edge_point[e] < 0 || edge_point[e] >= n_points) return CCM_E_ARG;
        deg[edge_point[e]]++;
    }
    double total = 0;
    for (int l = 0; l < n_points; l++) total += 0.5 * deg[l] * (deg[l] + 1) + deg[l] + 1;
    for (int r = 0; r <= n_ranks; r++) cuts[r] = n_points;
    cuts[0] = 0;
    double acc = 0;
    int r = 0;
    for (int l = 0; l < n_points; l++) {
        acc += 0.5 * deg[l] * (deg[l] + 1) + deg[l] + 1;
        while (r + 1 < n_ranks && acc >= total * (r + 1) / n_ranks) cuts[++r] = l + 1;
    }
    return CCM_OK;
}

static int ba_solve_impl(ccm_ctx* c, ccm_ba_problem* pb, const ccm_ba_options* opt, ccm_ba_result* res);
// No C++ exception may cross the C ABI (std::terminate would take the host process -- a SLAM server -- down): allocation failures of
// the host-side index vectors and anything else thrown below come back as status codes.
int ccm_ba_solve(ccm_ctx* c, ccm_ba_problem* pb, const ccm_ba_options* opt, ccm_ba_result* res)
{
    try { return ba_solve_impl(c, pb, opt, res); }
    catch (const std::bad_alloc&) { return c ? ccm_fail(c, CCM_E_NOMEM, "ccm_ba_solve: host allocation failed") : CCM_E_NOMEM; }
    catch (const std::exception& e) { return c ? ccm_fail(c, CCM_E_DEVICE, "ccm_ba_solve: %s", e.what()) : CCM_E_DEVICE; }
    catch (...) { return c ? ccm_fail(c, CCM_E_DEVICE, "ccm_ba_solve: unknown exception") : CCM_E_DEVICE; }
}

static int ba_solve_impl(ccm_ctx* c, ccm_ba_problem* pb, const ccm_ba_options* opt, ccm_ba_result* res)
{
    RoctxRange roctx_("ccm_ba_solve");
    if (!c || !pb || !opt) return CCM_E_ARG;
    if (pb->n_poses <= 0 || pb->n_points < 0 || pb->n_edges < 0 || !pb->poses || !pb->intr ||
        (pb->n_points > 0 && !pb->points) || (pb->n_edges > 0 && (!pb->edge_pose || !pb->edge_point || !pb->obs || !pb->info)))
        return ccm_fail(c, CCM_E_ARG, "bad BA problem");
    const int P = pb->n_poses, Lall = pb->n_points, Eall = pb->n_edges;
    // (every edge's vertex indices are range-checked by the first pass over the edge list below, before anything indexes with them)
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->ba) c->ba = new BaState();
    BaState& S = *c->ba;
    if (!S.pinned && hipHostMalloc((void**)&S.pinned, 16 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        S.pinned = nullptr;
        return ccm_fail(c, CCM_E_NOMEM, "hipHostMalloc failed");
    }
    hipStream_t st = c->stream;
    if (S.side) CCM_HIP(c, hipStreamSynchronize(S.side));      // an earlier call that ended on an error may have left work there
    const int ranks = comm_ranks(c), rank = comm_rank(c);
    ccm_ba_result local{};
    if (!res) res = &local;
    uint8_t* outlier_out = res->edge_outlier;
    *res = ccm_ba_result{};
    res->edge_outlier = outlier_out;

    const bool dbg_t = getenv("CCM_DEBUG") != nullptr;
    auto t_setup0 = clk::now();
    auto lap = [&](const char* what) {
        if (!dbg_t) return;
        (void)hipStreamSynchronize(st);
        auto t = clk::now();
        fprintf(stderr, "[ccm] setup %-28s %.3f ms\n", what, std::chrono::duration<double>(t - t_setup0).count() * 1e3);
        t_setup0 = t;
    };
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
    int l0 = 0, l1 = Lall;
    if (ranks > 1) {
        std::vector<int32_t> cut(ranks + 1);
        if (ccm_ba_landmark_cuts(pb->edge_point, Eall, Lall, ranks, cut.data())) return ccm_fail(c, CCM_E_ARG, "an edge references a landmark out of range");
        l0 = cut[rank]; l1 = cut[rank + 1];
    }
    const int L = l1 - l0;
    // local edges sorted by (landmark, pose); perm[k] = original edge id
    // (a comparison sort of the 1.8 M edges of config 5 took 20 ms, a fifth of the whole call: a graph extracted landmark by landmark
    // arrives sorted already, which one pass detects; otherwise a stable counting sort by landmark and an insertion sort of each
    // landmark's handful of observations by keyframe give the same order in linear time)
    // The passes over the edge list below were 4 ms of host time at config 5 (1.8 M edges): large graphs deal them to a few threads
    // (contiguous slices; every result is the same as the serial loop's).
    const int NT = Eall >= 400000 ? (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    auto pfor = [&](auto&& fn) {
        if (NT == 1) { fn(0); return; }
        std::vector<std::thread> th;
        int started = 1;                                   // slices 1 .. started-1 have a thread
        try { for (int t = 1; t < NT; t++) { th.emplace_back([&fn, t]() { fn(t); }); started = t + 1; } }
        catch (const std::system_error&) {}                // thread or process limit: the remaining slices run here, same results
        fn(0);
        for (int t = started; t < NT; t++) fn(t);
        for (auto& x : th) x.join();
    };
    auto slice = [&](long long total, int t) { return std::pair<long long, long long>(total * t / NT, total * (t + 1) / NT); };
    // Large unsharded maps: the raw edge list goes up as it is, and the device checks it (vertex ranges, order) and makes the index
    // structures (landmark -> edges, free keyframe -> edges by a stable radix sort).  A list that turns out unsorted takes the host path.
    int rc;
    bool dev_indexed = false;
    static const bool host_index_only = getenv("CCM_BA_HOST_INDEX") && atoi(getenv("CCM_BA_HOST_INDEX")) != 0;
    if (ranks == 1 && Eall >= 400000 && nfree > 0 && Lall > 0 && !host_index_only) {
        if ((rc = upload(c, S.free_of, free_of.data(), P))) return rc;
        if ((rc = upload(c, S.edge_pose, pb->edge_pose, Eall))) return rc;
        if ((rc = upload(c, S.edge_point, pb->edge_point, Eall))) return rc;
        CCM_RESERVE(c, S.pt_first, ((size_t)Lall + 2) * 4); CCM_RESERVE(c, S.pose_first, ((size_t)nfree + 2) * 4);
        CCM_RESERVE(c, S.pose_edges, (size_t)Eall * 4 + 16); CCM_RESERVE(c, S.info_dev, 64);
        CCM_RESERVE(c, S.sp_key, (size_t)Eall * 4 + 16); CCM_RESERVE(c, S.sp_key2, (size_t)Eall * 4 + 16); CCM_RESERVE(c, S.sp_off, (size_t)Eall * 4 + 16);
        const size_t ix_tmp = sp_sort_temp_bytes((size_t)Eall);
        CCM_RESERVE(c, S.sp_tmp, ix_tmp + 256);
        int* flags = S.info_dev.as<int>() + 8;
        const int init_flags[2] = { 0, 0x7FFFFFFF };
        CCM_HIP(c, hipMemcpyAsync(flags, init_flags, 8, hipMemcpyHostToDevice, st));
        ba_launch_index_check(st, S.edge_pose.as<int>(), S.edge_point.as<int>(), Eall, P, Lall, flags, S.pt_first.as<int>());
        ba_launch_index_pose_keys(st, S.edge_pose.as<int>(), S.free_of.as<int>(), Eall, P, nfree, S.sp_key.as<unsigned>(), S.sp_off.as<unsigned>());
        CCM_HIP(c, sp_sort_u32(st, S.sp_tmp.p, ix_tmp, S.sp_key.as<unsigned>(), S.sp_key2.as<unsigned>(), S.sp_off.as<unsigned>(),
                               S.pose_edges.as<unsigned>(), (size_t)Eall));
        ba_launch_index_pose_first(st, S.sp_key2.as<unsigned>(), Eall, nfree, S.pose_first.as<int>());
        int got[2] = { 0, 0 };
        CCM_HIP(c, hipMemcpyAsync(got, flags, 8, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        if (got[1] != 0x7FFFFFFF) return ccm_fail(c, CCM_E_ARG, "edge %d references a vertex out of range", got[1]);
        dev_indexed = got[0] == 0;
    }
    std::vector<int> perm;
    bool sorted = true;
    int n_local = dev_indexed ? Eall : 0;
    if (!dev_indexed) {
        std::vector<int> cnt(NT, 0), bad(NT, 0), first_l(NT, -1), first_p(NT, -1), last_l(NT, -1), last_p(NT, -1);
        std::vector<long long> out_of_range(NT, -1);
        pfor([&](int t) {
            const auto r = slice(Eall, t);
            int prev_l = -1, prev_p = -1, c = 0, b = 0;
            for (long long e = r.first; e < r.second; e++) {
                const int l = pb->edge_point[e];
                const int p = pb->edge_pose[e];
                if (p < 0 || p >= P || l < 0 || l >= Lall) { if (out_of_range[t] < 0) out_of_range[t] = e; continue; }
                if (l < l0 || l >= l1) continue;
                if (c == 0) { first_l[t] = l; first_p[t] = p; }
                else if (l < prev_l || (l == prev_l && p < prev_p)) b = 1;
                prev_l = l; prev_p = p; c++;
            }
            cnt[t] = c; bad[t] = b; last_l[t] = prev_l; last_p[t] = prev_p;
        });
        for (int t = 0; t < NT; t++)
            if (out_of_range[t] >= 0) return ccm_fail(c, CCM_E_ARG, "edge %lld references a vertex out of range", out_of_range[t]);
        int pl = -1, pp = -1;
        for (int t = 0; t < NT; t++) {
            if (bad[t]) sorted = false;
            if (cnt[t]) {
                if (first_l[t] < pl || (first_l[t] == pl && first_p[t] < pp)) sorted = false;      // across the slice boundary
                pl = last_l[t]; pp = last_p[t];
            }
            n_local += cnt[t];
        }
    }
    // a sorted, unsharded edge list is used where it lies (no index vector, no staging copies: 6 ms at config 5)
    const bool direct = sorted && n_local == Eall && l0 == 0;
    if (!direct) {
        perm.reserve(n_local);
        for (int e = 0; e < Eall; e++) { const int l = pb->edge_point[e]; if (l >= l0 && l < l1) perm.push_back(e); }
    }
    if (!sorted) {
        std::vector<int> first(L + 2, 0), out(perm.size());
        for (int e : perm) first[pb->edge_point[e] - l0 + 1]++;
        for (int l = 0; l < L; l++) first[l + 1] += first[l];
        {
            std::vector<int> fill(first.begin(), first.end() - 1);
            for (int e : perm) out[fill[pb->edge_point[e] - l0]++] = e;          // stable: ties keep the input order
        }
        for (int l = 0; l < L; l++)
            for (int a = first[l] + 1; a < first[l + 1]; a++) {                 // stable insertion sort by keyframe
                const int v = out[a], pv = pb->edge_pose[v];
                int b = a - 1;
                while (b >= first[l] && pb->edge_pose[out[b]] > pv) { out[b + 1] = out[b]; b--; }
                out[b + 1] = v;
            }
        perm.swap(out);
    }
    const int E = n_local;
    // Per-phase timers need a stream synchronisation at every phase boundary.  On a large graph that costs nothing next to the phases; on a
    // local BA (tens of keyframes, launch-bound) the four extra round trips per LM trial were a quarter of the call, so small problems
    // skip them: their timers still add up to the wall time, but a phase's GPU time is booked where the next necessary sync happens.
    static const bool force_timers = getenv("CCM_BA_TIMERS") && atoi(getenv("CCM_BA_TIMERS")) != 0;
    const bool fine_timers = force_timers || E >= 200000;
    std::vector<int> e_pose_v, e_pt_v, pt_first(L + 1, 0);
    std::vector<double> e_obs_v, e_info_v;
    if (!direct) {
        e_pose_v.resize(E); e_pt_v.resize(E); e_obs_v.resize(2 * (size_t)E); e_info_v.resize(E);
        for (int k = 0; k < E; k++) {
            const int e = perm[k];
            e_pose_v[k] = pb->edge_pose[e]; e_pt_v[k] = pb->edge_point[e] - l0;
            e_obs_v[2 * k] = pb->obs[2 * e]; e_obs_v[2 * k + 1] = pb->obs[2 * e + 1]; e_info_v[k] = pb->info[e];
        }
    }
    const int32_t* e_pose = direct ? pb->edge_pose : e_pose_v.data();
    const int32_t* e_pt = direct ? pb->edge_point : e_pt_v.data();
    const double* e_obs = direct ? pb->obs : e_obs_v.data();
    const double* e_info = direct ? pb->info : e_info_v.data();
    // pt_first: the edges are sorted by landmark, so a landmark's first edge is where the landmark index changes;
    // pose_edges: stable counting sort of the edges by free keyframe, slice by slice
    std::vector<int> pose_first(nfree + 1, 0);
    std::unique_ptr<int[]> pose_edges;                        // every slot is written below: no zero-fill of 7 MB
    size_t n_pose_edges = 0;
    if (!dev_indexed) {
        std::vector<std::vector<int>> hist(NT, std::vector<int>(nfree + 1, 0));
        pfor([&](int t) {
            const auto r = slice(E, t);
            std::vector<int>& h = hist[t];
            for (long long k = r.first; k < r.second; k++) {
                const int l = e_pt[k];
                const int lp = k > 0 ? e_pt[k - 1] : -1;
                for (int q = lp + 1; q <= l; q++) pt_first[q] = (int)k;           // landmarks without edges in between start here too
                const int f = free_of[e_pose[k]];
                if (f >= 0) h[f]++;
            }
        });
        const int last = E > 0 ? e_pt[E - 1] : -1;
        for (int q = last + 1; q <= L; q++) pt_first[q] = E;
        // slice t's first slot for keyframe f = all earlier keyframes + f's edges in earlier slices
        int acc = 0;
        for (int f = 0; f < nfree; f++) {
            pose_first[f] = acc;
            for (int t = 0; t < NT; t++) { const int c = hist[t][f]; hist[t][f] = acc; acc += c; }
        }
        pose_first[nfree] = acc;
        n_pose_edges = (size_t)acc; pose_edges.reset(new int[std::max<size_t>(n_pose_edges, 1)]);
        pfor([&](int t) {
            const auto r = slice(E, t);
            std::vector<int>& fill = hist[t];
            for (long long k = r.first; k < r.second; k++) { const int f = free_of[e_pose[k]]; if (f >= 0) pose_edges[fill[f]++] = (int)k; }
        });
    }

    lap("host: sort + index edges");
    // ---- device buffers
    if ((rc = upload(c, S.poses, pb->poses, 7 * (size_t)P))) return rc;
    if ((rc = upload(c, S.intr, pb->intr, 4 * (size_t)P))) return rc;
    if ((rc = upload(c, S.pose_of_free, pose_of_free.data(), nfree))) return rc;
    // Large maps taken where they lie: observations, information values and points (48 of the 65 MB at config 5) are not needed
    // before the first linearisation, so they go up on the context's first auxiliary stream while this stream builds the block
    // structure of the reduced system from the index arrays (pair enumeration, radix sorts: 0.9 ms at config 5).
    static const bool split_off = getenv("CCM_BA_SPLIT_UPLOAD") && atoi(getenv("CCM_BA_SPLIT_UPLOAD")) == 0;
    const bool split_up = direct && ranks == 1 && Eall >= 400000 && nfree > 0 && !split_off;
    if (split_up) {
        hipStream_t up_st = ccm_aux_stream(c, 0);               // (the stream the coarse inversion uses later in the call: S.side)
        if (!up_st) return ccm_fail(c, CCM_E_DEVICE, "hipStreamCreate failed");
        if (!S.ev_up) CCM_HIP(c, hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
        CCM_RESERVE(c, S.points, std::max<size_t>(3 * (size_t)L * 8, 16)); CCM_RESERVE(c, S.obs, std::max<size_t>(2 * (size_t)E * 8, 16));
        CCM_RESERVE(c, S.info, std::max<size_t>((size_t)E * 8, 16));
        if (L) CCM_HIP(c, hipMemcpyAsync(S.points.p, pb->points + 3 * (size_t)l0, 3 * (size_t)L * 8, hipMemcpyHostToDevice, up_st));
        if (E) CCM_HIP(c, hipMemcpyAsync(S.obs.p, e_obs, 2 * (size_t)E * 8, hipMemcpyHostToDevice, up_st));
        if (E) CCM_HIP(c, hipMemcpyAsync(S.info.p, e_info, (size_t)E * 8, hipMemcpyHostToDevice, up_st));
        CCM_HIP(c, hipEventRecord(S.ev_up, up_st));
    } else {
        if ((rc = upload(c, S.points, pb->points + 3 * (size_t)l0, 3 * (size_t)L))) return rc;
        if ((rc = upload(c, S.obs, e_obs, 2 * (size_t)E))) return rc;
        if ((rc = upload(c, S.info, e_info, E))) return rc;
    }
    // whatever way this function is left, the copies out of the caller's arrays have finished by then
    struct UploadDone { hipEvent_t e; ~UploadDone() { if (e) (void)hipEventSynchronize(e); } } upload_done{ split_up ? S.ev_up : nullptr };
    if (!dev_indexed) {                                        // (the device path has these already)
        if ((rc = upload(c, S.free_of, free_of.data(), P))) return rc;
        if ((rc = upload(c, S.edge_pose, e_pose, E))) return rc;
        if ((rc = upload(c, S.edge_point, e_pt, E))) return rc;
        if ((rc = upload(c, S.pt_first, pt_first.data(), L + 1))) return rc;
        if ((rc = upload(c, S.pose_first, pose_first.data(), nfree + 1))) return rc;
        if ((rc = upload(c, S.pose_edges, pose_edges.get(), n_pose_edges))) return rc;
    }
    const size_t nxl = (size_t)n + 3 * (size_t)L;
    CCM_RESERVE(c, S.Rt, 12 * (size_t)P * 8);
    CCM_RESERVE(c, S.active, std::max<size_t>(E, 16)); CCM_RESERVE(c, S.flags, std::max<size_t>(E, 16));
    CCM_RESERVE(c, S.err, std::max<size_t>(2 * (size_t)E * 8, 16));
    CCM_RESERVE(c, S.Hpp, std::max<size_t>(36 * (size_t)nfree * 8, 16)); CCM_RESERVE(c, S.bp, std::max<size_t>((size_t)n * 8, 16));
    CCM_RESERVE(c, S.Hpp2, std::max<size_t>(36 * (size_t)nfree * 8, 16)); CCM_RESERVE(c, S.bp2, std::max<size_t>((size_t)n * 8, 16));   // the LM loop's look-ahead (below)
    CCM_RESERVE(c, S.Hll, std::max<size_t>(9 * (size_t)L * 8, 16)); CCM_RESERVE(c, S.bl, std::max<size_t>(3 * (size_t)L * 8, 16));
    CCM_RESERVE(c, S.Hpl, std::max<size_t>(18 * (size_t)E * 8, 16)); CCM_RESERVE(c, S.Dinv, std::max<size_t>(9 * (size_t)L * 8, 16));
    CCM_RESERVE(c, S.x, std::max<size_t>(nxl * 8, 16));
    CCM_RESERVE(c, S.save_poses, 7 * (size_t)P * 8); CCM_RESERVE(c, S.save_points, std::max<size_t>(3 * (size_t)L * 8, 16));
    const size_t nb_max = (size_t)(E + 255) / 256 + (nxl + 255) / 256 + 8;             // (ba_launch_errors_scale keeps both kernels' partial sums)
    CCM_RESERVE(c, S.partial, nb_max * 8); CCM_RESERVE(c, S.scal, 64 * 8); CCM_RESERVE(c, S.info_dev, 64);
    CCM_RESERVE(c, S.tmp_ll, std::max<size_t>((size_t)L * 8, 16)); CCM_RESERVE(c, S.pp_diag, std::max<size_t>((size_t)n * 8, 16));
    CCM_HIP(c, hipMemsetAsync(S.active.p, 1, std::max(E, 1), st));
    CCM_HIP(c, hipMemsetAsync(S.err.p, 0, std::max<size_t>(2 * (size_t)E * 8, 16), st));
    CCM_HIP(c, hipMemsetAsync(S.x.p, 0, std::max<size_t>(nxl * 8, 16), st));
    CCM_HIP(c, hipStreamSynchronize(st));    // host staging vectors stay alive until here
    lap("upload + allocate");

    BaDev D{};
    D.P = P; D.L = L; D.E = E; D.nfree = nfree;
    D.poses = S.poses.as<double>(); D.Rt = S.Rt.as<double>(); D.intr = S.intr.as<double>();
    D.free_of = S.free_of.as<int>(); D.pose_of_free = S.pose_of_free.as<int>(); D.points = S.points.as<double>();
    D.edge_pose = S.edge_pose.as<int>(); D.edge_point = S.edge_point.as<int>(); D.obs = S.obs.as<double>(); D.info = S.info.as<double>();
    D.active = S.active.as<uint8_t>(); D.err = S.err.as<double>(); D.pt_first = S.pt_first.as<int>();
    D.pose_first = S.pose_first.as<int>(); D.pose_edges = S.pose_edges.as<int>();
    D.Hpp = S.Hpp.as<double>(); D.bp = S.bp.as<double>(); D.Hll = S.Hll.as<double>(); D.bl = S.bl.as<double>();
    D.Hpl = S.Hpl.as<double>(); D.Dinv = S.Dinv.as<double>();
    D.Hs = nullptr; D.bs = nullptr; D.x = S.x.as<double>();
    double* scal = S.scal.as<double>();      // [0] chi2, [1] scale, [2] stop flag (collective), [3] rank 0's verdict, [5] Hll max
    double* partial = S.partial.as<double>();
    int* info_dev = S.info_dev.as<int>();

    // ---- block-sparse structure of the reduced camera system (once per call; see ba_sparse.hip)
    if ((long long)nfree * nfree >= (1LL << 32)) return ccm_fail(c, CCM_E_ARG, "too many free keyframes (%d) for 32-bit block keys", nfree);
    const long long n2 = (long long)nfree * nfree;
    long long NP = 0;
    int nb = 0;
    if (nfree > 0) {
        const size_t scan_tmp = sp_scan_temp_bytes((size_t)std::max<long long>(n2, L + 1));
        CCM_RESERVE(c, S.sp_cnt, ((size_t)L + 2) * 4); CCM_RESERVE(c, S.sp_off, ((size_t)L + 2) * 4);
        CCM_RESERVE(c, S.sp_tmp, scan_tmp + 256);
        CCM_HIP(c, hipMemsetAsync(S.sp_cnt.p, 0, ((size_t)L + 2) * 4, st));
        if (L > 0) sp_launch_pair_count(st, D, S.sp_cnt.as<int>());
        CCM_HIP(c, sp_scan_int(st, S.sp_tmp.p, scan_tmp, S.sp_cnt.as<int>(), S.sp_off.as<int>(), (size_t)L + 1));
        int np_i = 0;
        CCM_HIP(c, hipMemcpyAsync(&np_i, S.sp_off.as<int>() + L, 4, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        NP = np_i;
        CCM_RESERVE(c, S.sp_key, std::max<size_t>((size_t)NP * 4, 16)); CCM_RESERVE(c, S.sp_val, std::max<size_t>((size_t)NP * 8, 16));
        CCM_RESERVE(c, S.sp_key2, std::max<size_t>((size_t)NP * 4, 16)); CCM_RESERVE(c, S.sp_val2, std::max<size_t>((size_t)NP * 8, 16));
        if (NP > 0) sp_launch_pair_fill(st, D, S.sp_off.as<int>(), S.sp_key.as<unsigned>(), S.sp_val.as<unsigned long long>());
        CCM_RESERVE(c, S.sp_map, (size_t)n2 + 16); CCM_RESERVE(c, S.sp_id, ((size_t)n2 + 2) * 4);
        CCM_HIP(c, hipMemsetAsync(S.sp_map.p, 0, (size_t)n2 + 16, st));
        sp_launch_mark(st, S.sp_key.as<unsigned>(), NP, nfree, S.sp_map.as<uint8_t>());
        if ((rc = comm_allreduce_u8_max(c, S.sp_map.as<uint8_t>(), (size_t)n2))) return rc;      // union pattern over ranks
        CCM_HIP(c, sp_scan_flags(st, S.sp_tmp.p, scan_tmp, S.sp_map.as<uint8_t>(), S.sp_id.as<int>(), (size_t)n2 + 1));
        CCM_HIP(c, hipMemcpyAsync(&nb, S.sp_id.as<int>() + n2, 4, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        CCM_RESERVE(c, S.blk_row, (size_t)nb * 4); CCM_RESERVE(c, S.blk_col, (size_t)nb * 4); CCM_RESERVE(c, S.diag_id, (size_t)nfree * 4);
        {
            // the scale columns of the prolongation: keyframe translations as they are now, and each aggregate's mean
            const int A = pcg_coarse_agg_keyframes(nfree), nagg = pcg_coarse_aggregates(nfree);
            std::vector<double> sv(3 * (size_t)nfree + 3 * (size_t)nagg, 0.0);
            for (int f = 0; f < nfree; f++) for (int q = 0; q < 3; q++) sv[3 * (size_t)f + q] = pb->poses[7 * (size_t)pose_of_free[f] + 4 + q];
            for (int I = 0; I < nagg; I++) {
                const int f0 = I * A, f1 = std::min(nfree, f0 + A);
                for (int q = 0; q < 3; q++) {
                    double m = 0;
                    for (int f = f0; f < f1; f++) m += sv[3 * (size_t)f + q];
                    sv[3 * (size_t)nfree + 3 * I + q] = m / std::max(1, f1 - f0);
                }
            }
            CCM_RESERVE(c, S.pcg_svec, sv.size() * 8 + 64);
            CCM_HIP(c, hipMemcpyAsync(S.pcg_svec.p, sv.data(), sv.size() * 8, hipMemcpyHostToDevice, st));
            CCM_HIP(c, hipStreamSynchronize(st));                                                         // sv is a local (the stream is idle here: nb has just been read)
        }
        sp_launch_block_coords(st, S.sp_map.as<uint8_t>(), S.sp_id.as<int>(), n2, nfree, S.blk_row.as<int>(), S.blk_col.as<int>(), S.diag_id.as<int>());
        // this rank's pairs sorted by target block (stable: fixed summation order)
        const size_t sort_tmp = sp_sort_temp_bytes((size_t)std::max<long long>(NP, 2LL * nb));
        CCM_RESERVE(c, S.sp_tmp, std::max(sort_tmp, scan_tmp) + 256);
        sp_launch_pair_block(st, S.sp_key.as<unsigned>(), S.sp_id.as<int>(), NP, S.sp_key2.as<unsigned>());
        int bits = 1; while ((1LL << bits) < nb + 1) bits++;
        if (NP > 0) CCM_HIP(c, sp_sort_u64(st, S.sp_tmp.p, sort_tmp, S.sp_key2.as<unsigned>(), S.sp_key.as<unsigned>(),
                                           S.sp_val.as<unsigned long long>(), S.sp_val2.as<unsigned long long>(), (size_t)NP, bits));
        CCM_RESERVE(c, S.seg_start, (size_t)nb * 4); CCM_RESERVE(c, S.seg_end, (size_t)nb * 4);
        CCM_HIP(c, hipMemsetAsync(S.seg_start.p, 0, (size_t)nb * 4, st)); CCM_HIP(c, hipMemsetAsync(S.seg_end.p, 0, (size_t)nb * 4, st));
        sp_launch_seg_bounds(st, S.sp_key.as<unsigned>(), NP, S.seg_start.as<int>(), S.seg_end.as<int>());
        // symmetric row lists for the mat-vec
        CCM_RESERVE(c, S.ent_key, (size_t)nb * 8); CCM_RESERVE(c, S.ent_val, (size_t)nb * 8);
        CCM_RESERVE(c, S.ent_key2, (size_t)nb * 8); CCM_RESERVE(c, S.ent_val2, (size_t)nb * 8);
        sp_launch_row_entries(st, S.blk_row.as<int>(), S.blk_col.as<int>(), nb, nfree, S.ent_key.as<unsigned>(), S.ent_val.as<unsigned>());
        CCM_HIP(c, sp_sort_u32(st, S.sp_tmp.p, sort_tmp, S.ent_key.as<unsigned>(), S.ent_key2.as<unsigned>(), S.ent_val.as<unsigned>(),
                               S.ent_val2.as<unsigned>(), (size_t)2 * nb));
        CCM_RESERVE(c, S.row_ptr, (size_t)nfree * 8);
        CCM_HIP(c, hipMemsetAsync(S.row_ptr.p, 0, (size_t)nfree * 8, st));
        sp_launch_row_ptr(st, S.ent_key2.as<unsigned>(), 2 * nb, nfree, S.row_ptr.as<int>());
        CCM_RESERVE(c, S.Hb, (36 * (size_t)nb + (size_t)n + 8) * 8);          // blocks, then bschur: one all-reduce covers both
        CCM_RESERVE(c, S.Minv, pcg_minv_bytes(nfree)); CCM_RESERVE(c, S.pcg_w, std::max(6 * (size_t)n, ppcg_state_doubles(nfree)) * 8 + 64);
        CCM_RESERVE(c, S.pcg_pap, (size_t)nfree * 8 + 64); CCM_RESERVE(c, S.pcg_part, ((size_t)nfree + (size_t)n / 192 + 4) * 3 * 8 + 64);
        CCM_RESERVE(c, S.pcg_sc, 64 * 8);
        {
            const size_t nc = (size_t)pcg_coarse_dim(nfree), ncp = (size_t)pcg_coarse_pitch(nfree);
            CCM_RESERVE(c, S.pcg_aci, ncp * ncp * 8 + 64); CCM_RESERVE(c, S.pcg_acw, (ncp * ncp + 48 * 48) * 8 + 64);   // + one block of scratch
            const size_t nrp = ((size_t)n / 192 + 2) * 4 * 7;                                                     // block partials of P^T r (ba_sparse.hip: PCG_UPD_TPB, PCG_RSLOTS, PCG_CDOF)
            CCM_RESERVE(c, S.pcg_coarse, (nrp + nc + (size_t)pcg_coarse_parts(nfree) + 64) * 8);                // P^T r, yc, cpart
        }
        CCM_HIP(c, hipGetLastError());
    }
    CCM_RESERVE(c, S.Y, std::max<size_t>(18 * (size_t)E * 8, 16)); CCM_RESERVE(c, S.db, std::max<size_t>(3 * (size_t)L * 8, 16));
    CCM_RESERVE(c, S.ce, std::max<size_t>(6 * (size_t)E * 8, 16));
    D.Z = S.Y.as<double>(); D.db = S.db.as<double>(); D.ce = S.ce.as<double>();
    double* Hb = S.Hb.as<double>();
    D.bs = nfree > 0 ? Hb + 36 * (size_t)nb : nullptr;
    res->schur_blocks = nb; res->schur_pairs = NP;
    // dense solve for small systems (exact, and cheaper than PCG start-up), PCG on the packed blocks otherwise
    static const int dense_max = getenv("CCM_BA_DENSE_MAX") ? atoi(getenv("CCM_BA_DENSE_MAX")) : 1536;
    const bool use_pcg = n > dense_max;
    // second preconditioner level (ba_sparse.hip): on for systems with at least 64 coarse unknowns
    static const bool want_coarse = !(getenv("CCM_PCG_COARSE") && atoi(getenv("CCM_PCG_COARSE")) == 0);
    PcgCoarse PC{};
    std::vector<int> coarse_pairs;                         // (I, J), I <= J
    const int nc = nfree > 0 ? pcg_coarse_dim(nfree) : 0, ncp = nfree > 0 ? pcg_coarse_pitch(nfree) : 0;
    if (use_pcg && want_coarse && nc >= 64 && nc <= 2304) {      // beyond: the cubic inversion would outlast an LM trial (more than 24 576 free keyframes)
        if (!S.side) {
            // the context's low-priority auxiliary stream: the inversion has a whole LM trial to finish, the PCG kernels it shares the GPU
            // with are the critical path
            if (!(S.side = ccm_aux_stream(c, 0))) return ccm_fail(c, CCM_E_DEVICE, "hipStreamCreate failed");
            CCM_HIP(c, hipEventCreateWithFlags(&S.ev_hb, hipEventDisableTiming));
            CCM_HIP(c, hipEventCreateWithFlags(&S.ev_inv, hipEventDisableTiming));
            CCM_HIP(c, hipEventCreateWithFlags(&S.ev_copy, hipEventDisableTiming));
        }
        PC.Aci = S.pcg_aci.as<double>();
        PC.rc = S.pcg_coarse.as<double>();
        PC.yc = PC.rc + ((size_t)n / 192 + 2) * 4 * 7;
        PC.cpart = PC.yc + nc;
        PC.svec = S.pcg_svec.as<double>();
        PC.cen = PC.svec + 3 * (size_t)nfree;
    }
    // Which iteration: the pipelined one (two kernels per iteration, ba_sparse.hip) for the tolerances a BA asks for; its recurrences
    // stall near a relative residual of 1e-9, so a caller that wants more than 1e-7 gets the classic four-kernel iteration.
    static const double env_tol = getenv("CCM_PCG_TOL") ? atof(getenv("CCM_PCG_TOL")) : 0.0;
    const double pcg_tol = env_tol > 0 ? env_tol : (opt->pcg_tol > 0 ? opt->pcg_tol : 1e-6);   // relative residual (default: see ccm_hot.h)
    static const int env_pipe = getenv("CCM_PCG_PIPELINED") ? atoi(getenv("CCM_PCG_PIPELINED")) : -1;            // test switch: 0 / 1 force
    const bool pipelined = use_pcg && nfree > 0 && ppcg_supported(nfree) && (env_pipe >= 0 ? env_pipe != 0 : pcg_tol >= 1e-7);
    PpcgBufs PB{};
    if (pipelined) {
        CCM_RESERVE(c, S.pcg_hf, 36 * 2 * (size_t)nb * 8 + 64); CCM_RESERVE(c, S.pcg_ecol, 2 * (size_t)nb * 4 + 64);
        CCM_RESERVE(c, S.pcg_ca, ppcg_ca_doubles(nfree) * 8 + 64);
        PB.Hf = S.pcg_hf.as<double>(); PB.ecol = S.pcg_ecol.as<int>(); PB.CA = S.pcg_ca.as<double>();
    }
    res->pcg_pipelined = pipelined ? 1 : 0;
    // The PCG inner loop is three or four small dependent kernels per iteration and is launch-bound when issued one by
    // one: capture a chunk of iterations (+ the scalar publication) into a HIP graph and replay it.
    // Chunk lengths (even: the r.z slot parity is the same at the start of every chunk): graphs of 8 and of 2 iterations; a host
    // round trip launches as many of them as the contraction observed so far says are still needed (rounded up to 2), then one
    // and looks again (k_pcg_direction publishes the scalars after every iteration) -- with the hat-function coarse level a trial takes 20-100 iterations of 45 us, so
    // iterations past convergence cost more than round trips.
    const int pcg_len[2] = {8, 2};
    // Graphs: [level][length]; level 0 = the cluster level alone (first trial of a call: no coarse inverse exists yet), 1 = both levels.
    PcgCoarse PC0{};
    hipGraph_t pcg_graph[4] = {nullptr, nullptr, nullptr, nullptr}; hipGraphExec_t pcg_exec[4] = {nullptr, nullptr, nullptr, nullptr};
    struct GraphGuard { hipGraph_t* g; hipGraphExec_t* e; ~GraphGuard() { for (int i = 0; i < 4; i++) { if (e[i]) (void)hipGraphExecDestroy(e[i]); if (g[i]) (void)hipGraphDestroy(g[i]); } } } graph_guard{pcg_graph, pcg_exec};
    lap("block structure (pairs, sort)");
    // (Captured on the context's second auxiliary stream, which is idle, while the main stream is still sorting the pair lists: the 0.3 ms
    //  of host time the capture takes used to be idle time of the GPU.)
    hipStream_t cap_st = (use_pcg && nfree > 0) ? ccm_aux_stream(c, 1) : nullptr;
    if (cap_st) {
        for (int gi = 0; gi < (PC.Aci ? 4 : 2); gi++) {
            const PcgCoarse& pc = (gi >> 1) ? PC : PC0;
            if (hipStreamBeginCapture(cap_st, hipStreamCaptureModeRelaxed) == hipSuccess) {
                for (int k = 0; k < pcg_len[gi & 1]; k++) {
                    if (pipelined) ppcg_launch_iter(cap_st, S.Minv.as<double>(), S.row_ptr.as<int>(), nfree, S.pcg_w.as<double>(), S.pcg_part.as<double>(), S.pcg_sc.as<double>(), pc, PB);
                    else pcg_launch_iter(cap_st, Hb, S.row_ptr.as<int>(), S.ent_key2.as<unsigned>(), S.ent_val2.as<unsigned>(), S.Minv.as<double>(),
                                         nfree, S.pcg_w.as<double>(), S.pcg_pap.as<double>(), S.pcg_part.as<double>(), S.pcg_sc.as<double>(), k & 1, pc);
                }
                hipError_t e1 = hipStreamEndCapture(cap_st, &pcg_graph[gi]);
                hipError_t e2 = e1 == hipSuccess ? hipGraphInstantiate(&pcg_exec[gi], pcg_graph[gi], nullptr, nullptr, 0) : e1;
                if (e2 != hipSuccess) {
                    pcg_exec[gi] = nullptr;                     // fall back to plain launches
                    if (getenv("CCM_DEBUG")) fprintf(stderr, "[ccm] PCG graph capture failed: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
                    (void)hipGetLastError();
                }
            } else { if (getenv("CCM_DEBUG")) fprintf(stderr, "[ccm] hipStreamBeginCapture failed\n"); (void)hipGetLastError(); }
            if (getenv("CCM_DEBUG")) fprintf(stderr, "[ccm] PCG graph %d %s\n", gi, pcg_exec[gi] ? "ready" : "not used");
        }
    }
    // The coarse inversion is 75 small launches (0.2 ms of host time per LM trial, during which the host does not answer the PCG's round
    // trips): captured once, replayed with one call.
    hipGraph_t inv_graph = nullptr; hipGraphExec_t inv_exec = nullptr;
    struct InvGraphGuard { hipGraph_t& g; hipGraphExec_t& e; ~InvGraphGuard() { if (e) (void)hipGraphExecDestroy(e); if (g) (void)hipGraphDestroy(g); } } inv_guard{inv_graph, inv_exec};
    if (PC.Aci && use_pcg) {
        double* Aw = S.pcg_acw.as<double>();
        if (hipStreamBeginCapture(S.side, hipStreamCaptureModeRelaxed) == hipSuccess) {
            pcg_launch_coarse_invert(S.side, Aw, ncp, Aw + (size_t)ncp * ncp, info_dev + 4);
            pcg_launch_coarse_mirror(S.side, Aw, ncp);
            hipError_t e1 = hipStreamEndCapture(S.side, &inv_graph);
            hipError_t e2 = e1 == hipSuccess ? hipGraphInstantiate(&inv_exec, inv_graph, nullptr, nullptr, 0) : e1;
            if (e2 != hipSuccess) { inv_exec = nullptr; (void)hipGetLastError(); }
        } else (void)hipGetLastError();
    }
    if (PC.Aci) {
        // the aggregate pairs that hold a block (the grid of the coarse matrix's assembly): fixed for the call
        const int nagg = pcg_coarse_aggregates(nfree);
        CCM_RESERVE(c, S.pcg_aggmap, (size_t)nagg * nagg + 16);
        CCM_HIP(c, hipMemsetAsync(S.pcg_aggmap.p, 0, (size_t)nagg * nagg, st));
        pcg_launch_coarse_mark(st, S.blk_row.as<int>(), S.blk_col.as<int>(), nb, nfree, S.pcg_aggmap.as<uint8_t>());
        std::vector<uint8_t> am((size_t)nagg * nagg);
        CCM_HIP(c, hipMemcpyAsync(am.data(), S.pcg_aggmap.p, am.size(), hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        for (int I = 0; I < nagg; I++)
            for (int J = I; J < nagg; J++) if (am[(size_t)I * nagg + J]) { coarse_pairs.push_back(I); coarse_pairs.push_back(J); }
        CCM_RESERVE(c, S.pcg_pairs, coarse_pairs.size() * 4 + 16);
        CCM_HIP(c, hipMemcpyAsync(S.pcg_pairs.p, coarse_pairs.data(), coarse_pairs.size() * 4, hipMemcpyHostToDevice, st));
        CCM_HIP(c, hipStreamSynchronize(st));
    }
    lap("PCG graph capture");
    if (split_up) CCM_HIP(c, hipStreamWaitEvent(st, S.ev_up, 0));          // observations, information values, points have arrived
    bool hb_in_use = false;                                // the side stream is still reading this trial's reduced system
    bool coarse_ready = false, coarse_pending = false;     // an inverse is in Aci / an inversion is running on the side stream
    bool copy_recorded = false;                            // ev_copy has been recorded in this call

    // *pbStopFlag (sparse_optimizer.cpp:376, optimization_algorithm_levenberg.cpp:149).  With several ranks the decision must be
    // the same everywhere or a rank would leave the loop while the others wait in the next all-reduce: every rank's sample of
    // its own flag rides on the chi2 all-reduce (a sum: non-zero = some rank saw it), and the loop tests that collective value.
    bool stop_collective = false;
    auto stop_requested = [&]() { return ranks > 1 ? stop_collective : (opt->stop_flag && *opt->stop_flag); };
    auto sync_stop = [&]() -> int {                            // a dedicated exchange where no chi2 evaluation precedes the test
        if (ranks <= 1) return CCM_OK;
        S.pinned[13] = (opt->stop_flag && *opt->stop_flag) ? 1.0 : 0.0;
        CCM_HIP(c, hipMemcpyAsync(scal + 2, S.pinned + 13, 8, hipMemcpyHostToDevice, st));
        int r = comm_allreduce_f64(c, scal + 2, 1, false);
        if (r) return r;
        CCM_HIP(c, hipMemcpyAsync(S.pinned + 10, scal + 2, 8, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        stop_collective = S.pinned[10] > 0.0;
        return CCM_OK;
    };
    // chi2 (+ optionally scale) of the current state, summed over ranks
    // Look-ahead of the LM loop (one rank): a trial is nearly always accepted, and the keyframes' side of the next linearisation (Hpp, bp)
    // depends on nothing but the state the trial has just produced -- so it is enqueued right behind the trial's chi2, into a second pair
    // of buffers, and runs while the host waits for the chi2 (an event, not the stream), decides and launches the landmark side.  An accepted
    // trial's iteration swaps the buffers in and launches the landmarks alone; a rejected one never looks at them.  Same kernel on the same
    // state: the values are those the iteration would have computed itself.
    double* Hpp_alt = S.Hpp2.as<double>();
    double* bp_alt = S.bp2.as<double>();
    static const bool no_ahead = getenv("CCM_BA_NO_LOOKAHEAD") && atoi(getenv("CCM_BA_NO_LOOKAHEAD")) != 0;                  // test / A-B switch
    if (!S.ev_chi) CCM_HIP(c, hipEventCreateWithFlags(&S.ev_chi, hipEventDisableTiming));
    auto eval_chi2 = [&](double hd, bool with_scale, double lambda, double* chi, double* scale, bool rt_current = false, bool look_ahead = false) -> int {
        if (!rt_current) ba_launch_pose_rt(st, D);                             // (k_ba_update leaves the matrices of the poses it moved)
        if (with_scale) ba_launch_errors_scale(st, D, hd, lambda, rank == 0 ? 1 : 0, partial, scal);
        else if (E > 0) ba_launch_errors(st, D, hd, partial, scal);
        else CCM_HIP(c, hipMemsetAsync(scal, 0, 8, st));
        if (ranks > 1) {
            if (!with_scale) CCM_HIP(c, hipMemsetAsync(scal + 1, 0, 8, st));
            S.pinned[13] = (opt->stop_flag && *opt->stop_flag) ? 1.0 : 0.0;
            CCM_HIP(c, hipMemcpyAsync(scal + 2, S.pinned + 13, 8, hipMemcpyHostToDevice, st));
            int r = comm_allreduce_f64(c, scal, 3, false);
            if (r) return r;
            CCM_HIP(c, hipMemcpyAsync(S.pinned + 8, scal, 24, hipMemcpyDeviceToHost, st));
            CCM_HIP(c, hipStreamSynchronize(st));
            stop_collective = S.pinned[10] > 0.0;
        } else {
            CCM_HIP(c, hipMemcpyAsync(S.pinned + 8, scal, with_scale ? 16 : 8, hipMemcpyDeviceToHost, st));
            if (look_ahead) {
                CCM_HIP(c, hipEventRecord(S.ev_chi, st));
                BaDev D2 = D;
                D2.Hpp = Hpp_alt; D2.bp = bp_alt;
                { ProfScope ps(c, CCM_PROF_BA_LINEARIZE); ba_launch_lin_pose(st, D2, hd); }
                CCM_HIP(c, hipEventSynchronize(S.ev_chi));             // (polling the event or the page-locked slots instead measured the same)
            } else CCM_HIP(c, hipStreamSynchronize(st));
        }
        *chi = S.pinned[8]; if (scale) *scale = S.pinned[9];
        return CCM_OK;
    };

    // Phase timers (t_linearize / t_schur / t_solve / t_update).  Small problems: host clock, booked where the next necessary
    // synchronisation falls.  Large ones (fine_timers): an event on the stream at every phase boundary, read once at the end of the
    // call -- round 2 synchronised the stream there instead, four idle gaps of 20-30 us per LM trial once a trial took 3 ms.
    std::vector<int> clock_phase;                          // phase the interval ENDING at event i belongs to (-1: none)
    auto tick = [&](int phase) {
        if (!fine_timers) return;
        const size_t i = clock_phase.size();
        if (i == S.clock_ev.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return; S.clock_ev.push_back(e); }
        if (hipEventRecord(S.clock_ev[i], st) == hipSuccess) clock_phase.push_back(phase);
    };
    double huber = opt->huber_delta > 0 ? opt->huber_delta : 0.0;
    bool first_eval = true;
    CCM_HIP(c, hipMemcpyAsync(S.save_poses.p, D.poses, 7 * (size_t)P * 8, hipMemcpyDeviceToDevice, st));     // the fixed keyframes' entries of the saved state
    tick(-1);
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
        if ((rc = sync_stop())) return rc;
        // computeActiveErrors + activeRobustChi2 at the top of an iteration (optimization_algorithm_levenberg.cpp:75-80) evaluate the state the
        // last ACCEPTED trial left, which that trial has just evaluated -- same kernels, same state, same sums: the value, the errors
        // and the pose matrices are carried over instead of computed again (one host round trip and three launches per iteration; a
        // local BA is bound by exactly those).  After a rejected last trial (state restored) and with several ranks (the stop flag
        // rides on this evaluation's all-reduce) the evaluation runs as before.
        bool chi_carried = false;
        double carried_chi = 0;
        bool ahead_ready = false;                            // Hpp_alt / bp_alt hold the pose side of the current state
        for (int it = 0; it < iterations; it++) {
            if (stop_requested()) { res->stopped = 1; break; }                    // !terminate(), sparse_optimizer.cpp:376
            auto t0 = clk::now();
            RoctxRange lin_("ba:linearize");
            double currentChi = 0;
            if (chi_carried && ranks == 1) currentChi = carried_chi;
            else if ((rc = eval_chi2(huber, false, 0, &currentChi, nullptr))) return rc;
            chi_carried = false;
            const double iniChi = currentChi;
            if (first_eval) { res->chi2_initial = currentChi; first_eval = false; }
            // buildSystem.  From the second iteration on lambda is known here, and the landmarks' share of the first trial's Schur step
            // (Dinv, db, Z, ce) is computed by the same kernel
            const bool fused_schur = it > 0 && nfree > 0 && lambda > 0;
            // (and without Hpl, which only a repeated trial reads: see k_ba_lin_landmark MODE 2)
            static const bool keep_hpl = getenv("CCM_BA_KEEP_HPL") && atoi(getenv("CCM_BA_KEEP_HPL")) != 0;          // test / A-B switch
            const bool use_ahead = ahead_ready;
            ahead_ready = false;
            if (use_ahead) { std::swap(D.Hpp, Hpp_alt); std::swap(D.bp, bp_alt); }
            { ProfScope ps(c, CCM_PROF_BA_LINEARIZE); ba_launch_linearize(st, D, huber, fused_schur ? lambda : 0.0, keep_hpl, use_ahead); }
            bool landmark_share_ready = fused_schur;
            bool hpl_valid = !fused_schur || keep_hpl;
            if (it == 0) {                                                          // computeLambdaInit
                ba_launch_diag(st, D, S.tmp_ll.as<double>(), S.pp_diag.as<double>(), scal + 5);
                if (L == 0) CCM_HIP(c, hipMemsetAsync(scal + 5, 0, 8, st));
                if ((rc = comm_allreduce_f64(c, scal + 5, 1, true))) return rc;
                if ((rc = comm_allreduce_f64(c, S.pp_diag.as<double>(), (size_t)n, false))) return rc;
                std::vector<double> dg((size_t)n + 1);
                CCM_HIP(c, hipMemcpyAsync(dg.data(), S.pp_diag.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
                CCM_HIP(c, hipMemcpyAsync(&dg[n], scal + 5, 8, hipMemcpyDeviceToHost, st));
                CCM_HIP(c, hipStreamSynchronize(st));
                double md = 0;
                for (double v : dg) md = std::max(md, std::fabs(v));
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            tick(0);
            lin_.end();
            if (!fine_timers) res->t_linearize += secs(t0, clk::now());
            double rho = 0;
            int qmax = 0;
            do {
                auto t1 = clk::now();
                RoctxRange trial_("ba:trial (schur + solve + update)");
                // (push: k_ba_update saves the state it is about to change; the fixed keyframes' poses were copied once, above)
                int ok2 = 1;
                bool updated = false, looked_ahead = false;
                // The dense solve's verdict ("not positive definite") of a small, single-rank problem is read together with the
                // trial's chi2 instead of in a round trip of its own (a local BA is launch- and round-trip-bound: three host syncs
                // per trial were a fifth of the call): the trial's update is applied as if the solve had succeeded and, if it had
                // not, discarded exactly like a rejected step (the saved poses and points come back).
                volatile int* dense_info = reinterpret_cast<volatile int*>(S.pinned + 14);   // pinned: a copy to pageable memory would synchronise
                bool dense_info_pending = false;
                auto t2 = t1;
                if (nfree > 0) {
                    if (!landmark_share_ready) {
                        ProfScope ps(c, CCM_PROF_BA_DINV_Y);
                        if (!hpl_valid) {
                            // a rejected trial of an iteration linearised without Hpl: the state is the linearisation point again (pop),
                            // the pose matrices are the rejected trial's
                            ba_launch_pose_rt(st, D);
                            ba_launch_linearize(st, D, huber, 0.0, true, true);
                            hpl_valid = true;
                        }
                        sp_launch_dinv(st, D, lambda);
                    }
                    landmark_share_ready = false;                          // a repeated trial has another lambda
                    { ProfScope ps(c, CCM_PROF_BA_SCHUR_BLOCKS);
                      sp_launch_schur_blocks(st, D, S.Y.as<double>(), S.sp_val2.as<unsigned long long>(), S.seg_start.as<int>(), S.seg_end.as<int>(),
                                             S.blk_row.as<int>(), S.blk_col.as<int>(), nb, Hb); }
                    { ProfScope ps(c, CCM_PROF_BA_BSCHUR); sp_launch_bschur(st, D, D.bs); }
                    if ((rc = comm_allreduce_f64(c, Hb, 36 * (size_t)nb + (size_t)n, false))) return rc;
                    // (a system small enough for k_dense_small_solve gets its damping there)
                    static const bool no_small = getenv("CCM_BA_NO_SMALL_SOLVE") && atoi(getenv("CCM_BA_NO_SMALL_SOLVE")) != 0;   // test switch
                    const bool small_solve = !use_pcg && !no_small && n <= dense_small_max();
                    if (!small_solve) sp_launch_add_lambda(st, S.diag_id.as<int>(), nfree, lambda, Hb);
                    tick(1);
                    t2 = clk::now();
                    if (!fine_timers) res->t_schur += secs(t1, t2);
                    bool solved = false;
                    if (use_pcg) {
                        int* bad = info_dev + 1;
                        CCM_HIP(c, hipMemsetAsync(bad, 0, 4, st));
                        CCM_HIP(c, pcg_launch_minv(st, Hb, S.blk_row.as<int>(), S.blk_col.as<int>(), nb, nfree, S.Minv.as<double>(), bad));
                        // Coarse level, pipelined: the inverse this trial uses was computed from the PREVIOUS trial's system on
                        // the side stream while that trial's PCG ran (a preconditioner may be stale: lambda differs by the LM
                        // factor, H by one relinearisation); this trial's system starts the next inversion.  Which inverse a
                        // trial uses depends only on the trial number, never on timing, so all ranks do the same.
                        volatile int* cinfo = reinterpret_cast<volatile int*>(S.pinned + 15);
                        bool coarse_unverified = false;
                        if (PC.Aci && coarse_pending) {
                            // No host round trip: the main stream waits for the inversion (normally over for milliseconds), copies the
                            // inverse, and the verdict of the inversion is read together with the first scalars of this trial's PCG.
                            CCM_HIP(c, hipStreamWaitEvent(st, S.ev_inv, 0));
                            CCM_HIP(c, hipMemcpyAsync(PC.Aci, S.pcg_acw.p, (size_t)ncp * ncp * 8, hipMemcpyDeviceToDevice, st));
                            CCM_HIP(c, hipMemcpyAsync(S.pinned + 15, info_dev + 4, 4, hipMemcpyDeviceToHost, st));
                            CCM_HIP(c, hipEventRecord(S.ev_copy, st));                   // the next inversion overwrites the work matrix: it waits for this copy
                            coarse_pending = false; coarse_ready = true; coarse_unverified = true; copy_recorded = true;
                        }
                        const bool more_trials_planned = it + 1 < iterations || (stage == 0 && opt->iterations2 > 0);
                        // The next inversion (assembly of Ac from this trial's system, block Gauss-Jordan: 68 small launches, 0.2 ms
                        // of host time) goes to the side stream right after this trial's first chunk of PCG iterations has been
                        // launched, so the host enqueues it while the GPU is already iterating.
                        bool side_todo = PC.Aci && more_trials_planned;
                        double inv_host_ms = 0;
                        auto start_inversion = [&]() -> int {
                            auto ti0 = clk::now();
                            double* Aw = S.pcg_acw.as<double>();
                            if (copy_recorded) CCM_HIP(c, hipStreamWaitEvent(S.side, S.ev_copy, 0));
                            CCM_HIP(c, hipMemsetAsync(info_dev + 4, 0, 4, S.side));
                            CCM_HIP(c, pcg_launch_coarse_build(S.side, Hb, S.sp_map.as<uint8_t>(), S.sp_id.as<int>(), nfree, PC.svec, PC.cen, S.pcg_pairs.as<int>(),
                                                               (int)(coarse_pairs.size() / 2), Aw));
                            CCM_HIP(c, hipEventRecord(S.ev_hb, S.side));                 // awaited before the next trial overwrites Hb
                            hb_in_use = true;
                            if (inv_exec) CCM_HIP(c, hipGraphLaunch(inv_exec, S.side));
                            else {
                                pcg_launch_coarse_invert(S.side, Aw, ncp, Aw + (size_t)ncp * ncp, info_dev + 4);
                                pcg_launch_coarse_mirror(S.side, Aw, ncp);
                            }
                            CCM_HIP(c, hipEventRecord(S.ev_inv, S.side));
                            CCM_HIP(c, hipGetLastError());
                            coarse_pending = true; side_todo = false;
                            inv_host_ms = secs(ti0, clk::now()) * 1e3;
                            return CCM_OK;
                        };
                        const int max_it = 40 * 8 + (int)std::min<long long>(n, 4000);
                        const double tol2 = pcg_tol * pcg_tol;
                        volatile double* sc = S.pinned;
                        int itc = 0;
                        volatile int& badh = *reinterpret_cast<volatile int*>(S.pinned + 11);   // page-locked: a copy to the stack would stall the host until it is done
                        badh = 0;
                        CCM_HIP(c, hipMemcpyAsync(S.pinned + 11, bad, 4, hipMemcpyDeviceToHost, st));
                        if (pipelined) ppcg_launch_expand(st, Hb, S.ent_key2.as<unsigned>(), S.ent_val2.as<unsigned>(), 2 * nb, nfree, PB.Hf, PB.ecol);
                        // one run of the iteration: 0 = converged, 1 = not positive definite (or, pipelined, the recurrences broke down), 2 = no convergence
                        // (3 = the coarse inverse this trial was to use came from a system that was not positive definite: run again without it)
                        auto pcg_run = [&](bool pip) -> int {
                            const PcgCoarse& pcu = coarse_ready ? PC : PC0;
                            const int glv = coarse_ready ? 2 : 0;
                            if (pip) CCM_HIP(c, ppcg_launch_init(st, D.bs, S.Minv.as<double>(), S.row_ptr.as<int>(), nfree, S.pcg_w.as<double>(), S.pcg_part.as<double>(),
                                                                 S.pcg_sc.as<double>(), pcu, PB));
                            else pcg_launch_init(st, D.bs, S.Minv.as<double>(), nfree, S.pcg_w.as<double>(), S.pcg_part.as<double>(), S.pcg_sc.as<double>(), pcu);
                            int last_len = 0;
                            double rr_prev = 0;
                            for (;;) {
                                CCM_HIP(c, hipMemcpyAsync(S.pinned, S.pcg_sc.p, 5 * sizeof(double), hipMemcpyDeviceToHost, st));
                                CCM_HIP(c, hipStreamSynchronize(st));
                                if (coarse_unverified) { coarse_unverified = false; if (*cinfo != 0) return 3; }
                                if (badh || !(sc[3] > 0.0) || !std::isfinite(sc[2])) return 1;
                                if (sc[2] <= tol2 * sc[1]) return 0;
                                if (itc >= max_it) return 2;
                                // iterations still needed if |r|^2 keeps contracting as over the last round trip; without an estimate
                                // (first round trip, stagnation): 16, or 32 while there is no coarse level
                                const double rr = sc[2], target = tol2 * sc[1];
                                int need = coarse_ready ? 16 : 32;
                                if (last_len > 0 && rr_prev > 0 && rr < rr_prev) {
                                    const double est = std::log(rr / target) / (std::log(rr_prev / rr) / last_len);
                                    need = (int)std::min(16.0, std::max(2.0, std::ceil(est)));       // (CG speeds up as it goes: a longer forecast overshoots)
                                }
                                const int n8 = need / pcg_len[0], n2 = (need - n8 * pcg_len[0] + pcg_len[1] - 1) / pcg_len[1];
                                rr_prev = rr; last_len = n8 * pcg_len[0] + n2 * pcg_len[1];
                                for (int which = 0; which < 2; which++) {
                                    hipGraphExec_t gexec = pip == pipelined ? pcg_exec[glv + which] : nullptr;     // the graphs hold the iteration chosen for this call
                                    for (int rpt = 0; rpt < (which ? n2 : n8); rpt++) {
                                        if (gexec) CCM_HIP(c, hipGraphLaunch(gexec, st));
                                        else
                                            for (int k = 0; k < pcg_len[which]; k++) {
                                                if (pip) ppcg_launch_iter(st, S.Minv.as<double>(), S.row_ptr.as<int>(), nfree, S.pcg_w.as<double>(), S.pcg_part.as<double>(), S.pcg_sc.as<double>(), pcu, PB);
                                                else pcg_launch_iter(st, Hb, S.row_ptr.as<int>(), S.ent_key2.as<unsigned>(), S.ent_val2.as<unsigned>(), S.Minv.as<double>(),
                                                                     nfree, S.pcg_w.as<double>(), S.pcg_pap.as<double>(), S.pcg_part.as<double>(), S.pcg_sc.as<double>(), k & 1, pcu);
                                            }
                                    }
                                }
                                // |r|^2 of the iterate the chunk ended on (the classic kernels publish it with every direction, one update late:
                                // k_pcg_scalars brings sc[2], sc[3] up to date as well)
                                if (pip) ppcg_launch_publish(st, S.pcg_part.as<double>(), nfree, S.pcg_sc.as<double>());
                                else pcg_launch_publish(st, nfree, S.pcg_part.as<double>(), S.pcg_sc.as<double>(), pcu);
                                itc += last_len;
                                if (side_todo && (rc = start_inversion())) return -rc;
                            }
                        };
                        int status = pcg_run(pipelined);
                        if (status == 3) { coarse_ready = false; status = pcg_run(pipelined); }
                        if (status < 0) return -status;
                        if (status == 1 && pipelined && !badh) {                 // a breakdown of the recurrences is not a verdict on the matrix: ask the classic iteration
                            res->pcg_fallbacks++;
                            status = pcg_run(false);
                            if (status < 0) return -status;
                        }
                        if (status == 1) { ok2 = 0; solved = true; }             // not positive definite
                        else if (status == 0) solved = true;
                        if (side_todo && (rc = start_inversion())) return rc;
                        res->pcg_iterations += itc;
                        if (getenv("CCM_DEBUG")) fprintf(stderr, "[ccm] PCG trial: %d iterations, %.3f ms (host time of the side-stream enqueue %.3f), rel.res %.2e, |b| %.4e, lambda %.3e\n", itc, secs(t2, clk::now()) * 1e3, inv_host_ms, std::sqrt(sc[2] / std::max((double)sc[1], 1e-300)), std::sqrt((double)sc[1]), lambda);
                        if (solved && ok2) CCM_HIP(c, hipMemcpyAsync(D.x, S.pcg_w.p, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
                        if (!solved) res->pcg_fallbacks++;
                    }
                    if (!solved) {
                        // dense solve by the in-house block Gauss-Jordan (see dense_launch_solve for why not rocSOLVER)
                        if (small_solve) {
                            // a local BA's system: factored and solved by one workgroup in LDS, one launch (see k_dense_small_solve)
                            if (dense_launch_small_solve(st, Hb, S.blk_row.as<int>(), S.blk_col.as<int>(), nb, (int)n, D.bs, D.x, info_dev, lambda))
                                return ccm_fail(c, CCM_E_DEVICE, "k_dense_small_solve: LDS request refused");
                        } else {
                        CCM_HIP(c, hipMemsetAsync(info_dev, 0, 4, st));
                        const size_t npd = (size_t)dense_pitch(n);
                        CCM_RESERVE(c, S.Hs, (npd * npd + 48 * 48 + 8) * 8);
                        double* Hs = S.Hs.as<double>();
                        CCM_HIP(c, hipMemsetAsync(Hs, 0, npd * npd * 8, st));
                        sp_launch_to_dense(st, Hb, S.blk_row.as<int>(), S.blk_col.as<int>(), nb, (long long)npd, Hs);   // row-major upper block triangle, pitch npd
                        dense_launch_solve(st, Hs, (int)n, (int)npd, D.bs, D.x, info_dev);
                        }
                        CCM_HIP(c, hipMemcpyAsync(S.pinned + 14, info_dev, 4, hipMemcpyDeviceToHost, st));
                        if (ranks == 1 && !fine_timers) dense_info_pending = true;
                        else { CCM_HIP(c, hipStreamSynchronize(st)); ok2 = *dense_info == 0; }
                    }
                } else {
                    // no free keyframe: only the landmark inverse is needed for the back-substitution
                    sp_launch_dinv(st, D, lambda);
                    CCM_HIP(c, hipStreamSynchronize(st));
                    tick(1);
                    t2 = clk::now();
                    if (!fine_timers) res->t_schur += secs(t1, t2);
                }
                if (hb_in_use) { CCM_HIP(c, hipEventSynchronize(S.ev_hb)); hb_in_use = false; }     // long over: the assembly is the side stream's first 0.2 ms
                if (ranks > 1 && nfree > 0) {
                    // Every rank has solved the same reduced system.  Rank 0's increment -- and its verdict on positive
                    // definiteness -- is the one all ranks apply, so their poses stay bit-identical whatever a rank's solver
                    // did: the others contribute zeros to a sum all-reduce (x + 0 is exact).
                    double* flag = scal + 3;
                    S.pinned[12] = (rank == 0 && ok2) ? 1.0 : 0.0;
                    CCM_HIP(c, hipMemcpyAsync(flag, S.pinned + 12, 8, hipMemcpyHostToDevice, st));
                    if (rank != 0 || !ok2) CCM_HIP(c, hipMemsetAsync(D.x, 0, (size_t)n * 8, st));
                    if ((rc = comm_allreduce_f64(c, D.x, (size_t)n, false))) return rc;
                    if ((rc = comm_allreduce_f64(c, flag, 1, false))) return rc;
                    CCM_HIP(c, hipMemcpyAsync(S.pinned + 12, flag, 8, hipMemcpyDeviceToHost, st));
                    CCM_HIP(c, hipStreamSynchronize(st));
                    ok2 = S.pinned[12] != 0.0;
                }
                tick(2);
                auto t3 = clk::now();
                if (!fine_timers) res->t_solve += secs(t2, t3);
                res->trials++;
                double tempChi = DBL_MAX, scale = 0;
                if (ok2) {
                    if (L > 0) { ProfScope ps(c, CCM_PROF_BA_BACKSUB); ba_launch_backsub(st, D, lambda); }
                    ba_launch_update(st, D, S.save_poses.as<double>(), S.save_points.as<double>());
                    updated = true;
                    looked_ahead = ranks == 1 && nfree > 0 && !no_ahead && it + 1 < iterations;
                    if ((rc = eval_chi2(huber, true, lambda, &tempChi, &scale, true, looked_ahead))) return rc;     // synchronises the stream (or, looking ahead, waits for the chi2 alone)
                    if (dense_info_pending && *dense_info != 0) { ok2 = 0; tempChi = DBL_MAX; scale = 0; }
                }
                scale += 1e-3;
                rho = ok2 ? (currentChi - tempChi) / scale : -1.0;
                // test switch: the first trial of iteration N is treated as rejected (the repeated-trial path -- pop, another lambda,
                // Hpl rebuilt -- on graphs whose trials are all accepted)
                static const int reject_at = getenv("CCM_BA_TEST_REJECT_AT") ? atoi(getenv("CCM_BA_TEST_REJECT_AT")) : -1;
                if (it == reject_at && qmax == 0) rho = -1.0;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha);
                    ni = 2; currentChi = tempChi;                                   // discardTop
                    ahead_ready = looked_ahead;
                } else {
                    lambda *= ni; ni *= 2;
                    if (updated) {                                                  // pop (a trial whose solve failed has not moved anything)
                        CCM_HIP(c, hipMemcpyAsync(D.poses, S.save_poses.p, 7 * (size_t)P * 8, hipMemcpyDeviceToDevice, st));
                        if (L) CCM_HIP(c, hipMemcpyAsync(D.points, S.save_points.p, 3 * (size_t)L * 8, hipMemcpyDeviceToDevice, st));
                    }
                }
                qmax++;
                tick(3);
                if (!fine_timers) res->t_update += secs(t3, clk::now());
            } while (rho < 0 && qmax < 10 && !stop_requested());
            if (rho > 0 && std::isfinite(currentChi)) { chi_carried = true; carried_chi = currentChi; }      // the last trial was accepted: currentChi is its chi2
            res->iterations_done++;
            res->chi2_final = currentChi; res->lambda_final = lambda;
            if (qmax == 10 || rho == 0) break;                                       // Terminate
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;         // stop criterion :154-161
            if (nBad >= 3) break;
        }
    }

    if (S.side) CCM_HIP(c, hipStreamSynchronize(S.side));
    if (fine_timers && clock_phase.size() > 1) {
        CCM_HIP(c, hipStreamSynchronize(st));
        double* tp[4] = { &res->t_linearize, &res->t_schur, &res->t_solve, &res->t_update };
        for (size_t i = 1; i < clock_phase.size(); i++) {
            float ms = 0.f;
            if (clock_phase[i] >= 0 && hipEventElapsedTime(&ms, S.clock_ev[i - 1], S.clock_ev[i]) == hipSuccess) *tp[clock_phase[i]] += 1e-3 * (double)ms;
        }
    }
    lap("LM loop");
    // ---- results
    CCM_HIP(c, hipMemcpyAsync(pb->poses, D.poses, 7 * (size_t)P * 8, hipMemcpyDeviceToHost, st));
    if (ranks == 1) {
        // (a page-locked landing area + a threaded copy for pageable destinations was measured: 1.19 against 0.75 ms for config 5's 4.8 MB --
        //  the first touch of a freshly allocated destination costs the same either way, and the threads cost their creation)
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
        const bool straight = ranks == 1 && direct;               // the caller's array is in the device's edge order
        std::vector<uint8_t> fl(straight ? 1 : std::max(E, 1));
        if (E > 0) {
            ba_launch_outliers(st, D, opt->outlier_chi2, S.flags.as<uint8_t>());
            CCM_HIP(c, hipMemcpyAsync(straight ? outlier_out : fl.data(), S.flags.p, E, hipMemcpyDeviceToHost, st));
        }
        CCM_HIP(c, hipStreamSynchronize(st));
        if (straight) {
            // (downloaded in place)
        } else if (ranks == 1) {
            for (int k = 0; k < E; k++) outlier_out[perm[k]] = fl[k];
        } else {
            // flags of the other ranks' edges: exchange as doubles through the same collective
            std::vector<double> full(Eall, 0.0);
            for (int k = 0; k < E; k++) full[direct ? k : perm[k]] = fl[k];
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
    lap("download results");
    return CCM_OK;
}

}  // extern "C"
