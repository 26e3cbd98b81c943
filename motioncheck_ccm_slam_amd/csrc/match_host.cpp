// match_host.cpp -- C ABI of the matcher (include/ccm_hot.h): brute-force Hamming search,
// ORBmatcher::SearchByBoW (cslam/src/ORBmatcher.cpp:178-306, 565-698; on the device, k_bow_greedy) and the windowed matchers.
#include "ccm_internal.h"
#include <algorithm>
#include <climits>
#include <cmath>
#include <numeric>

void match_launch_bf(hipStream_t, const uint8_t* q, long long q_pair_bytes, const uint8_t* t, long long t_pair_bytes,
                     int nq, int nt, int n_pairs, const int* nq_n, const int* nt_n, int n_split, int variant,
                     unsigned* part_best, int* part_second, int* bi, int* bd, int* sd);
void match_launch_ranges(hipStream_t, const uint8_t* d1, const uint8_t* d2, const int* order2, const int* start,
                         const int* len, const long long* off, int n1, unsigned short* dist);
struct BowDev {                                  // must match match_kernels.hip
    int n_groups, n1;
    const int* ga; const int* gae; const int* gb; const int* gbe;
    const int* ord1; const int* ord2;
    const uint8_t* d1; const uint8_t* d2;
    const uint8_t* valid1; const uint8_t* valid2;
    const float* angle1; const float* angle2;
    uint8_t* taken; int* match12; int* bin_of; int* hist;
    float nnratio; int th, strict_th, check_ori;
};
void match_launch_bow(hipStream_t, const BowDev&);

struct WindowBufs;
void match_window_free(WindowBufs*);
struct MatchState {
    WindowBufs* win = nullptr;                  // windowed matchers
    DevBuf q, t, nqn, ntn, bi, bd, sd;          // brute force staging
    DevBuf part_best, part_second;              // per-split partial results
    DevBuf d1, d2, order2, start, len, off, dist; // BoW staging
    DevBuf order1, grp[4], bv1, bv2, ba1, ba2, taken, m12, binof, hist;   // device-side SearchByBoW
};
void match_state_free(MatchState* s)
{
    if (!s) return;
    DevBuf* all[] = { &s->q, &s->t, &s->nqn, &s->ntn, &s->bi, &s->bd, &s->sd, &s->d1, &s->d2, &s->order2, &s->start, &s->len, &s->off, &s->dist, &s->part_best, &s->part_second,
                      &s->order1, &s->grp[0], &s->grp[1], &s->grp[2], &s->grp[3], &s->bv1, &s->bv2, &s->ba1, &s->ba2, &s->taken, &s->m12, &s->binof, &s->hist };
    for (DevBuf* b : all) b->release();
    match_window_free(s->win);
    delete s;
}

extern "C" {

// ORBmatcher::DescriptorDistance, ORBmatcher.cpp:1653-1669
int ccm_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4);
        std::memcpy(&y, b + 4 * i, 4);
        d += __builtin_popcount(x ^ y);
    }
    return d;
}

// ORBmatcher.cpp:247-249 (Frame overload) / :641-643 (KF-KF overload)
int ccm_ratio_test(int best_dist, int second_dist, float nnratio, int th, int strict)
{
    const bool pass = strict ? (best_dist < th) : (best_dist <= th);
    return pass && (static_cast<float>(best_dist) < nnratio * static_cast<float>(second_dist)) ? 1 : 0;
}

int ccm_hamming_match_dev(ccm_ctx* c, const uint8_t* q_dev, int nq, size_t q_pair_stride, const uint8_t* t_dev, int nt,
                          size_t t_pair_stride, int n_pairs, const int32_t* nq_n_dev, const int32_t* nt_n_dev,
                          int32_t* best_idx_dev, int32_t* best_dist_dev, int32_t* second_dist_dev)
{
    RoctxRange roctx_("ccm_hamming_match_dev");
    if (!c) return CCM_E_ARG;
    if (n_pairs == 0 || nq == 0) return CCM_OK;
    if (n_pairs < 0 || nq < 0 || nt < 0 || nt > 65535 || !q_dev || (!t_dev && nt > 0) || !best_idx_dev || !best_dist_dev || !second_dist_dev)
        return ccm_fail(c, CCM_E_ARG, "bad matcher arguments (nt must be <= 65535)");
    if (((uintptr_t)q_dev | (uintptr_t)t_dev) & 15) return ccm_fail(c, CCM_E_ARG, "descriptor arrays must be 16-byte aligned");
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->match) c->match = new MatchState();
    MatchState& M = *c->match;
    // Few pairs cannot fill 256 CUs with one workgroup each: split the train rows of a pair over several
    // workgroups (exact merge afterwards) until there are about four workgroups per CU.
    static const int env_split = getenv("CCM_BF_SPLIT") ? atoi(getenv("CCM_BF_SPLIT")) : 0;
    // default: the matrix-core kernel (variant 3) while its per-train table fits LDS (nt <= 2048), else the VALU kernel
    static const int variant = getenv("CCM_BF_VARIANT") ? atoi(getenv("CCM_BF_VARIANT")) : 3;
    int n_split = env_split > 0 ? env_split : 1;
    if (env_split <= 0) while (n_split < 8 && (long long)n_pairs * n_split < 1024 && nt / (n_split * 2) >= 128) n_split *= 2;
    if (n_split > 1) {
        CCM_RESERVE(c, M.part_best, (size_t)n_pairs * n_split * nq * 4);
        CCM_RESERVE(c, M.part_second, (size_t)n_pairs * n_split * nq * 4);
    }
    ProfScope ps(c, CCM_PROF_HAMMING_BF);
    match_launch_bf(c->stream, q_dev, (long long)q_pair_stride * 32, t_dev, (long long)t_pair_stride * 32, nq, nt, n_pairs,
                    nq_n_dev, nt_n_dev, n_split, variant, M.part_best.as<unsigned>(), M.part_second.as<int>(),
                    best_idx_dev, best_dist_dev, second_dist_dev);
    CCM_HIP(c, hipGetLastError());
    return CCM_OK;
}

int ccm_hamming_match(ccm_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int n_pairs, const int32_t* nq_n,
                      const int32_t* nt_n, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    RoctxRange roctx_("ccm_hamming_match");
    if (!c) return CCM_E_ARG;
    if (n_pairs == 0 || nq == 0) return CCM_OK;
    if (n_pairs < 0 || nq < 0 || nt < 0 || !q || (!t && nt > 0) || !best_idx || !best_dist || !second_dist)
        return ccm_fail(c, CCM_E_ARG, "bad matcher arguments");
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->match) c->match = new MatchState();
    MatchState& M = *c->match;
    const size_t qb = (size_t)n_pairs * nq * 32, tb = (size_t)n_pairs * nt * 32, ob = (size_t)n_pairs * nq * 4;
    CCM_RESERVE(c, M.q, qb); CCM_RESERVE(c, M.t, std::max<size_t>(tb, 32));
    CCM_RESERVE(c, M.bi, ob); CCM_RESERVE(c, M.bd, ob); CCM_RESERVE(c, M.sd, ob);
    CCM_RESERVE(c, M.nqn, (size_t)n_pairs * 4); CCM_RESERVE(c, M.ntn, (size_t)n_pairs * 4);
    CCM_HIP(c, hipMemcpyAsync(M.q.p, q, qb, hipMemcpyHostToDevice, c->stream));
    if (tb) CCM_HIP(c, hipMemcpyAsync(M.t.p, t, tb, hipMemcpyHostToDevice, c->stream));
    if (nq_n) CCM_HIP(c, hipMemcpyAsync(M.nqn.p, nq_n, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    if (nt_n) CCM_HIP(c, hipMemcpyAsync(M.ntn.p, nt_n, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    int rc = ccm_hamming_match_dev(c, M.q.as<uint8_t>(), nq, nq, M.t.as<uint8_t>(), nt, nt, n_pairs,
                                   nq_n ? M.nqn.as<int32_t>() : nullptr, nt_n ? M.ntn.as<int32_t>() : nullptr,
                                   M.bi.as<int32_t>(), M.bd.as<int32_t>(), M.sd.as<int32_t>());
    if (rc) return rc;
    CCM_HIP(c, hipMemcpyAsync(best_idx, M.bi.p, ob, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipMemcpyAsync(best_dist, M.bd.p, ob, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipMemcpyAsync(second_dist, M.sd.p, ob, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    return CCM_OK;
}

// ORBmatcher::ComputeThreeMaxima, ORBmatcher.cpp:1607-1648
static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// Side-2 candidates of every side-1 feature = the features of the same vocabulary node (DBoW2::FeatureVector walk of
// SearchByBoW / SearchForTriangulation), with their Hamming distances from k_hamming_ranges.
struct BowRanges {
    std::vector<int> ord1, ord2, start, len;
    std::vector<long long> off;
    std::vector<unsigned short> dist;
};
static int bow_ranges(ccm_ctx* c, const uint8_t* desc1, const int32_t* node1, const uint8_t* valid1, int n1,
                      const uint8_t* desc2, const int32_t* node2, int n2, BowRanges& R)
{
    if (!c->match) c->match = new MatchState();
    MatchState& M = *c->match;
    std::vector<int>&ord1 = R.ord1, &ord2 = R.ord2, &start = R.start, &len = R.len;
    std::vector<long long>& off = R.off;
    std::vector<unsigned short>& dist = R.dist;
    // FeatureVector order: node ascending, feature index ascending inside a node (DBoW2 fills it so)
    ord1.assign(n1, 0); ord2.assign(n2, 0);
    std::iota(ord1.begin(), ord1.end(), 0); std::iota(ord2.begin(), ord2.end(), 0);
    auto by_node = [](const int32_t* node) { return [node](int a, int b) { return node[a] != node[b] ? node[a] < node[b] : a < b; }; };
    std::stable_sort(ord1.begin(), ord1.end(), by_node(node1));
    std::stable_sort(ord2.begin(), ord2.end(), by_node(node2));
    // per side-1 feature: the slice of ord2 holding its node (features without a node have id < 0)
    start.assign(n1, 0); len.assign(n1, 0); off.assign(n1 + 1, 0);
    {
        size_t b = 0;
        for (size_t a = 0; a < ord1.size();) {
            const int nd = node1[ord1[a]];
            size_t ae = a; while (ae < ord1.size() && node1[ord1[ae]] == nd) ae++;
            while (b < ord2.size() && node2[ord2[b]] < nd) b++;
            size_t be = b; while (be < ord2.size() && node2[ord2[be]] == nd) be++;
            if (nd >= 0) for (size_t i = a; i < ae; i++) if (valid1[ord1[i]]) { start[ord1[i]] = (int)b; len[ord1[i]] = (int)(be - b); }
            a = ae; b = be;
        }
    }
    for (int i = 0; i < n1; i++) off[i + 1] = off[i] + len[i];
    const long long total = off[n1];
    dist.assign((size_t)std::max<long long>(total, 1), 0);
    if (total > 0) {
        CCM_RESERVE(c, M.d1, (size_t)n1 * 32); CCM_RESERVE(c, M.d2, (size_t)n2 * 32);
        CCM_RESERVE(c, M.order2, (size_t)n2 * 4); CCM_RESERVE(c, M.start, (size_t)n1 * 4); CCM_RESERVE(c, M.len, (size_t)n1 * 4);
        CCM_RESERVE(c, M.off, (size_t)(n1 + 1) * 8); CCM_RESERVE(c, M.dist, (size_t)total * 2);
        CCM_HIP(c, hipMemcpyAsync(M.d1.p, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice, c->stream));
        CCM_HIP(c, hipMemcpyAsync(M.d2.p, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, c->stream));
        CCM_HIP(c, hipMemcpyAsync(M.order2.p, ord2.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, c->stream));
        CCM_HIP(c, hipMemcpyAsync(M.start.p, start.data(), (size_t)n1 * 4, hipMemcpyHostToDevice, c->stream));
        CCM_HIP(c, hipMemcpyAsync(M.len.p, len.data(), (size_t)n1 * 4, hipMemcpyHostToDevice, c->stream));
        CCM_HIP(c, hipMemcpyAsync(M.off.p, off.data(), (size_t)(n1 + 1) * 8, hipMemcpyHostToDevice, c->stream));
        match_launch_ranges(c->stream, M.d1.as<uint8_t>(), M.d2.as<uint8_t>(), M.order2.as<int>(), M.start.as<int>(),
                            M.len.as<int>(), M.off.as<long long>(), n1, M.dist.as<unsigned short>());
        CCM_HIP(c, hipGetLastError());
        CCM_HIP(c, hipMemcpyAsync(dist.data(), M.dist.p, (size_t)total * 2, hipMemcpyDeviceToHost, c->stream));
        CCM_HIP(c, hipStreamSynchronize(c->stream));
    }
    return CCM_OK;
}

int ccm_match_bow(ccm_ctx* c, const ccm_bow_options* o, const uint8_t* desc1, const int32_t* node1, const uint8_t* valid1,
                  const float* angle1, int n1, const uint8_t* desc2, const int32_t* node2, const uint8_t* valid2,
                  const float* angle2, int n2, int32_t* match12)
{
    RoctxRange roctx_("ccm_match_bow");
    if (!c || !o) return CCM_E_ARG;
    if (n1 < 0 || n2 < 0 || (n1 > 0 && (!desc1 || !node1 || !valid1 || !match12)) || (n2 > 0 && (!desc2 || !node2)) ||
        (o->check_ori && n1 > 0 && n2 > 0 && (!angle1 || !angle2)))
        return ccm_fail(c, CCM_E_ARG, "bad SearchByBoW arguments");
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->match) c->match = new MatchState();
    MatchState& M = *c->match;
    // FeatureVector order: node ascending, feature index ascending inside a node (DBoW2 fills it so); the merge walk of
    // :201-298 pairs the runs of equal node ids -- one group per common node
    std::vector<int> ord1(n1), ord2(n2);
    std::iota(ord1.begin(), ord1.end(), 0); std::iota(ord2.begin(), ord2.end(), 0);
    auto by_node = [](const int32_t* node) { return [node](int a, int b) { return node[a] != node[b] ? node[a] < node[b] : a < b; }; };
    std::stable_sort(ord1.begin(), ord1.end(), by_node(node1));
    std::stable_sort(ord2.begin(), ord2.end(), by_node(node2));
    std::vector<int> grp[4];
    {
        size_t pa = 0, pb = 0;
        while (pa < ord1.size() && pb < ord2.size()) {
            const int na = node1[ord1[pa]], nb2 = node2[ord2[pb]];
            if (na < 0) { pa++; continue; }                              // features without a node are in no FeatureVector entry
            if (nb2 < 0) { pb++; continue; }
            if (na < nb2) { while (pa < ord1.size() && node1[ord1[pa]] == na) pa++; continue; }
            if (nb2 < na) { while (pb < ord2.size() && node2[ord2[pb]] == nb2) pb++; continue; }
            size_t ae = pa, be = pb;
            while (ae < ord1.size() && node1[ord1[ae]] == na) ae++;
            while (be < ord2.size() && node2[ord2[be]] == na) be++;
            grp[0].push_back((int)pa); grp[1].push_back((int)ae); grp[2].push_back((int)pb); grp[3].push_back((int)be);
            pa = ae; pb = be;
        }
    }
    const int ng = (int)grp[0].size();
    hipStream_t st = c->stream;
    auto up = [&](DevBuf& bf, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, bf, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(bf.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(M.d1, desc1, (size_t)n1 * 32)) || (rc = up(M.d2, desc2, (size_t)n2 * 32))) return rc;
    if ((rc = up(M.order1, ord1.data(), (size_t)n1 * 4)) || (rc = up(M.order2, ord2.data(), (size_t)n2 * 4))) return rc;
    for (int k = 0; k < 4; k++) if ((rc = up(M.grp[k], grp[k].data(), (size_t)ng * 4))) return rc;
    if ((rc = up(M.bv1, valid1, (size_t)n1))) return rc;
    if (valid2 && (rc = up(M.bv2, valid2, (size_t)n2))) return rc;
    if (o->check_ori && ((rc = up(M.ba1, angle1, (size_t)n1 * 4)) || (rc = up(M.ba2, angle2, (size_t)n2 * 4)))) return rc;
    CCM_RESERVE(c, M.taken, (size_t)n2); CCM_RESERVE(c, M.m12, (size_t)n1 * 4); CCM_RESERVE(c, M.binof, (size_t)n1 * 4); CCM_RESERVE(c, M.hist, 32 * 4);
    CCM_HIP(c, hipMemsetAsync(M.taken.p, 0, (size_t)n2, st));
    CCM_HIP(c, hipMemsetAsync(M.m12.p, 0xFF, (size_t)n1 * 4, st));
    CCM_HIP(c, hipMemsetAsync(M.hist.p, 0, 32 * 4, st));
    BowDev B{ ng, n1, M.grp[0].as<int>(), M.grp[1].as<int>(), M.grp[2].as<int>(), M.grp[3].as<int>(), M.order1.as<int>(), M.order2.as<int>(),
              M.d1.as<uint8_t>(), M.d2.as<uint8_t>(), M.bv1.as<uint8_t>(), valid2 ? M.bv2.as<uint8_t>() : nullptr,
              M.ba1.as<float>(), M.ba2.as<float>(), M.taken.as<uint8_t>(), M.m12.as<int>(), M.binof.as<int>(), M.hist.as<int>(),
              o->nnratio, o->th, o->strict_th, o->check_ori };
    match_launch_bow(st, B);
    CCM_HIP(c, hipGetLastError());
    int nmatches = 0;
    CCM_HIP(c, hipMemcpyAsync(match12, M.m12.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipMemcpyAsync(&nmatches, M.hist.as<int>() + 30, 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));
    return nmatches;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// F1: windowed matching.  The GPU enumerates, for every query, the features inside its
// search window with their Hamming distances (k_window_candidates).  The acceptance runs on the device too for the matchers
// a server batches over many map points: Fuse / SearchBySim3 (k_window_select: no coupling between queries), and
// SearchByProjection(Frame, map points) / SearchByProjection(KF, Scw) / SearchByProjection(Frame, Frame | KeyFrame)
// (k_window_greedy: the reference's order-dependent occupancy bookkeeping resolved by claim rounds, bit-identical to the
// sequential loop; the frame matcher's rotation histogram and its three maxima in the same kernel).  Only the
// initialisation matcher (called once per map) keeps its acceptance loop on the host.
struct WinGrid {
    int n, cols, rows; float min_x, min_y, inv_w, inv_h;
    const float* kx; const float* ky; const int* oct; const uint8_t* desc; const int* cell_first; const int* cell_items;
};
void match_launch_window(hipStream_t, const WinGrid&, int nq, const float* qx, const float* qy, const float* qr, const int* minl,
                         const int* maxl, const uint8_t* qdesc, int cap, int* ci, int* cd, int* cn);

void match_launch_window_select_batch(hipStream_t, const WinGrid* grids, const int* q_kf, int nq, const float* qx, const float* qy, const float* qr,
                                      const int* minl, const int* maxl, const uint8_t* qdesc, const float* inv_sigma2, int accept_th, int* best_idx, int* best_dist);
void match_launch_window_select(hipStream_t, const WinGrid&, int nq, const float* qx, const float* qy, const float* qr, const int* minl,
                                const int* maxl, const uint8_t* qdesc, const float* inv_sigma2, int accept_th, int* best_idx, int* best_dist);
struct GreedyArgs {
    int nq, n, cap; const int* ci; const int* cd; const int* cn; const uint8_t* active; const int* qlevel; const int* oct; const uint8_t* qflag;
    uint8_t* flag; float nnratio; int* out; int* status;
    int orb_dist, check_ori; const float* q_angle; const float* f_angle; int* ev;
    const struct GreedyKf* kfs;
};
struct GreedyKf { int q0, nq, f0, n; };
int match_launch_window_greedy_batch(hipStream_t, const GreedyArgs&, int n_kf, int max_n);
void match_launch_window_batch(hipStream_t, const WinGrid* grids, const int* q_kf, int nq, const float* qx, const float* qy, const float* qr, const int* minl,
                               const int* maxl, const uint8_t* qdesc, int cap, int* ci, int* cd, int* cn);
size_t match_window_greedy_lds(int n, int nq);
int match_launch_window_greedy(hipStream_t, int mode, const GreedyArgs&);

struct WindowBufs { DevBuf kx, ky, oct, desc, cfirst, citems, qx, qy, qr, minl, maxl, qdesc, ci, cd, cn, sel_i, sel_d, is2, act, qlvl, qflag, flag, out, status, qang, fang, ev,
                           grids, qkf, gkf; };
// LDS the single-workgroup acceptance kernel may ask for (claim + flag per feature, one byte per query); larger problems take the
// host loops below
static const size_t kGreedyLdsMax = 150 * 1024;

// mode 0: candidate lists to the host (ci / cd / cn); 1: lists stay in HBM for k_window_greedy; 2: no lists, k_window_select
// leaves one (index, distance) per query in W.sel_i / W.sel_d
static int window_run(ccm_ctx* c, const ccm_frame_grid* f, int nq, const float* qx, const float* qy, const float* qr,
                      const int32_t* minl, const int32_t* maxl, const uint8_t* qdesc, int cap, int mode,
                      std::vector<int32_t>& ci, std::vector<int32_t>& cd, std::vector<int32_t>& cn,
                      const float* inv_sigma2 = nullptr, int n_levels = 0, int accept_th = 0)
{
    if (!c->match) c->match = new MatchState();
    if (!c->match->win) c->match->win = new WindowBufs();
    WindowBufs& W = *c->match->win;
    const int n = f->n, cells = f->grid_cols * f->grid_rows;
    // Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cpp:103-118, 255-266): mGrid[x][y], features in index order
    std::vector<int> cell(n), first(cells + 1, 0), items(std::max(n, 1));
    for (int i = 0; i < n; i++) {
        const int px = (int)std::round((f->kp_x[i] - f->min_x) * f->inv_w), py = (int)std::round((f->kp_y[i] - f->min_y) * f->inv_h);
        cell[i] = (px < 0 || px >= f->grid_cols || py < 0 || py >= f->grid_rows) ? -1 : px * f->grid_rows + py;
        if (cell[i] >= 0) first[cell[i] + 1]++;
    }
    for (int k = 0; k < cells; k++) first[k + 1] += first[k];
    { std::vector<int> fill(first.begin(), first.end() - 1); for (int i = 0; i < n; i++) if (cell[i] >= 0) items[fill[cell[i]]++] = i; }
    hipStream_t st = c->stream;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(W.kx, f->kp_x, (size_t)n * 4)) || (rc = up(W.ky, f->kp_y, (size_t)n * 4)) || (rc = up(W.oct, f->kp_octave, (size_t)n * 4)) ||
        (rc = up(W.desc, f->desc, (size_t)n * 32)) || (rc = up(W.cfirst, first.data(), first.size() * 4)) ||
        (rc = up(W.citems, items.data(), (size_t)n * 4)) || (rc = up(W.qx, qx, (size_t)nq * 4)) || (rc = up(W.qy, qy, (size_t)nq * 4)) ||
        (rc = up(W.qr, qr, (size_t)nq * 4)) || (rc = up(W.minl, minl, (size_t)nq * 4)) || (rc = up(W.maxl, maxl, (size_t)nq * 4)) ||
        (rc = up(W.qdesc, qdesc, (size_t)nq * 32)))
        return rc;
    WinGrid G{ n, f->grid_cols, f->grid_rows, f->min_x, f->min_y, f->inv_w, f->inv_h, W.kx.as<float>(), W.ky.as<float>(), W.oct.as<int>(),
               W.desc.as<uint8_t>(), W.cfirst.as<int>(), W.citems.as<int>() };
    if (mode == 2) {
        CCM_RESERVE(c, W.sel_i, (size_t)nq * 4); CCM_RESERVE(c, W.sel_d, (size_t)nq * 4);
        if (inv_sigma2 && (rc = up(W.is2, inv_sigma2, (size_t)n_levels * 4))) return rc;
        match_launch_window_select(st, G, nq, W.qx.as<float>(), W.qy.as<float>(), W.qr.as<float>(), W.minl.as<int>(), W.maxl.as<int>(),
                                   W.qdesc.as<uint8_t>(), inv_sigma2 ? W.is2.as<float>() : nullptr, accept_th, W.sel_i.as<int>(), W.sel_d.as<int>());
        CCM_HIP(c, hipGetLastError());
        return CCM_OK;
    }
    CCM_RESERVE(c, W.ci, (size_t)nq * cap * 4); CCM_RESERVE(c, W.cd, (size_t)nq * cap * 4); CCM_RESERVE(c, W.cn, (size_t)nq * 4);
    match_launch_window(st, G, nq, W.qx.as<float>(), W.qy.as<float>(), W.qr.as<float>(), W.minl.as<int>(), W.maxl.as<int>(),
                        W.qdesc.as<uint8_t>(), cap, W.ci.as<int>(), W.cd.as<int>(), W.cn.as<int>());
    CCM_HIP(c, hipGetLastError());
    if (mode == 1) return CCM_OK;
    ci.resize((size_t)nq * cap); cd.resize((size_t)nq * cap); cn.resize(nq);
    CCM_HIP(c, hipMemcpyAsync(ci.data(), W.ci.p, ci.size() * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipMemcpyAsync(cd.data(), W.cd.p, cd.size() * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipMemcpyAsync(cn.data(), W.cn.p, cn.size() * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));
    return CCM_OK;
}

static int window_candidates(ccm_ctx* c, const ccm_frame_grid* f, int nq, const float* qx, const float* qy, const float* qr,
                             const int32_t* minl, const int32_t* maxl, const uint8_t* qdesc, int cap,
                             std::vector<int32_t>& ci, std::vector<int32_t>& cd, std::vector<int32_t>& cn)
{
    return window_run(c, f, nq, qx, qy, qr, minl, maxl, qdesc, cap, 0, ci, cd, cn);
}

// The order-dependent acceptance on the device (k_window_greedy): candidate lists stay in HBM, in: per-query active / level / flag and
// the per-feature flags; out: `out` (n_out ints, pre-set to -1), the updated flags, the number of matches.  Returns the match count,
// or < 0 on error.  Lists longer than `cap` make the kernel report the needed length and the call repeats once.
static int window_greedy(ccm_ctx* c, const ccm_frame_grid* f, int nq, const float* qx, const float* qy, const float* qr,
                         const int32_t* minl, const int32_t* maxl, const uint8_t* qdesc, int mode, const uint8_t* active, const int32_t* qlevel,
                         const uint8_t* qflag, uint8_t* flag, float nnratio, int32_t* out, int n_out,
                         int orb_dist = 0, int check_ori = 0, const float* q_angle = nullptr, const float* f_angle = nullptr)
{
    std::vector<int32_t> d0, d1, d2;
    int cap = 64;
    for (int attempt = 0; attempt < 3; attempt++) {
        int rc = window_run(c, f, nq, qx, qy, qr, minl, maxl, qdesc, cap, 1, d0, d1, d2);
        if (rc) return rc;
        WindowBufs& W = *c->match->win;
        hipStream_t st = c->stream;
        auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
            CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
            if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
            return CCM_OK;
        };
        if ((rc = up(W.act, active, (size_t)nq)) || (rc = up(W.qflag, qflag, (size_t)nq)) || (rc = up(W.flag, flag, (size_t)f->n))) return rc;
        if (qlevel && (rc = up(W.qlvl, qlevel, (size_t)nq * 4))) return rc;
        if (mode == 2) {
            CCM_RESERVE(c, W.ev, std::max<size_t>((size_t)nq * 4, 16));
            if (check_ori && ((rc = up(W.qang, q_angle, (size_t)nq * 4)) || (rc = up(W.fang, f_angle, (size_t)f->n * 4)))) return rc;
        }
        CCM_RESERVE(c, W.out, std::max<size_t>((size_t)n_out * 4, 16)); CCM_RESERVE(c, W.status, 16);
        CCM_HIP(c, hipMemsetAsync(W.out.p, 0xFF, (size_t)n_out * 4, st));
        GreedyArgs A{ nq, f->n, cap, W.ci.as<int>(), W.cd.as<int>(), W.cn.as<int>(), W.act.as<uint8_t>(), qlevel ? W.qlvl.as<int>() : nullptr,
                      W.oct.as<int>(), W.qflag.as<uint8_t>(), W.flag.as<uint8_t>(), nnratio, W.out.as<int>(), W.status.as<int>(),
                      orb_dist, check_ori, W.qang.as<float>(), W.fang.as<float>(), W.ev.as<int>(), nullptr };
        if (match_launch_window_greedy(st, mode, A)) return ccm_fail(c, CCM_E_DEVICE, "k_window_greedy: LDS request refused");
        CCM_HIP(c, hipGetLastError());
        int status[3] = { 0, 0, 0 };
        CCM_HIP(c, hipMemcpyAsync(status, W.status.p, 12, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        if (status[0] < 0) { cap = status[1]; continue; }                        // rare: a denser window than expected
        CCM_HIP(c, hipMemcpyAsync(out, W.out.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipMemcpyAsync(flag, W.flag.p, (size_t)f->n, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        return status[0];
    }
    return ccm_fail(c, CCM_E_CAPACITY, "window candidate lists keep overflowing");
}

void match_window_free(WindowBufs* w)
{
    if (!w) return;
    DevBuf* all[] = { &w->kx, &w->ky, &w->oct, &w->desc, &w->cfirst, &w->citems, &w->qx, &w->qy, &w->qr, &w->minl, &w->maxl, &w->qdesc, &w->ci, &w->cd, &w->cn,
                      &w->sel_i, &w->sel_d, &w->is2, &w->act, &w->qlvl, &w->qflag, &w->flag, &w->out, &w->status, &w->qang, &w->fang, &w->ev, &w->grids, &w->qkf, &w->gkf };
    for (DevBuf* b : all) b->release();
    delete w;
}

extern "C" {

int ccm_window_candidates(ccm_ctx* c, const ccm_frame_grid* f, int nq, const float* qx, const float* qy, const float* qr,
                          const int32_t* min_level, const int32_t* max_level, const uint8_t* qdesc, int cap,
                          int32_t* cand_idx, int32_t* cand_dist, int32_t* cand_n)
{
    if (!c || !f) return CCM_E_ARG;
    if (nq == 0) return CCM_OK;
    if (nq < 0 || cap < 1 || f->n < 0 || f->grid_cols < 1 || f->grid_rows < 1 || !qx || !qy || !qr || !min_level || !max_level || !qdesc ||
        !cand_idx || !cand_dist || !cand_n || (f->n > 0 && (!f->kp_x || !f->kp_y || !f->kp_octave || !f->desc)))
        return ccm_fail(c, CCM_E_ARG, "bad window-search arguments");
    CCM_HIP(c, hipSetDevice(c->device));
    std::vector<int32_t> ci, cd, cn;
    int rc = window_candidates(c, f, nq, qx, qy, qr, min_level, max_level, qdesc, cap, ci, cd, cn);
    if (rc) return rc;
    std::memcpy(cand_idx, ci.data(), ci.size() * 4); std::memcpy(cand_dist, cd.data(), cd.size() * 4); std::memcpy(cand_n, cn.data(), cn.size() * 4);
    for (int q = 0; q < nq; q++) if (cn[q] > cap) return ccm_fail(c, CCM_E_CAPACITY, "query %d has %d candidates, cap %d", q, cn[q], cap);
    return CCM_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<mpptr>&, th), ORBmatcher.cpp:71-148
int ccm_search_by_projection(ccm_ctx* c, const ccm_frame_grid* f, const float* scale_factors, int n_mp, const uint8_t* in_view,
                             const int32_t* level, const float* view_cos, const float* proj_x, const float* proj_y,
                             const uint8_t* mp_desc, const uint8_t* mp_has_obs, uint8_t* occupied, float th, float nnratio,
                             int32_t* match)
{
    if (!c || !f) return CCM_E_ARG;
    if (f->n < 0 || n_mp < 0 || (f->n > 0 && (!match || !occupied)) ||
        (n_mp > 0 && (!scale_factors || !in_view || !level || !view_cos || !proj_x || !proj_y || !mp_desc || !mp_has_obs)))
        return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection arguments");
    for (int i = 0; i < f->n; i++) match[i] = -1;
    if (n_mp == 0 || f->n == 0) return 0;
    CCM_HIP(c, hipSetDevice(c->device));
    const bool bFactor = th != 1.0;
    std::vector<float> qr(n_mp); std::vector<int32_t> minl(n_mp), maxl(n_mp);
    for (int m = 0; m < n_mp; m++) {
        if (!in_view[m]) { qr[m] = -1.f; minl[m] = 0; maxl[m] = 0; continue; }
        float r = view_cos[m] > 0.998 ? 2.5f : 4.0f;                          // RadiusByViewingCos :150-156
        if (bFactor) r *= th;
        qr[m] = r * scale_factors[level[m]];
        minl[m] = level[m] - 1; maxl[m] = level[m];
    }
    static const bool host_accept = getenv("CCM_WINDOW_HOST_ACCEPT") && atoi(getenv("CCM_WINDOW_HOST_ACCEPT")) != 0;   // test switch
    if (!host_accept && match_window_greedy_lds(f->n, n_mp) <= kGreedyLdsMax)
        return window_greedy(c, f, n_mp, proj_x, proj_y, qr.data(), minl.data(), maxl.data(), mp_desc, 0, in_view, nullptr, mp_has_obs, occupied,
                             nnratio, match, f->n);
    int cap = 64;
    std::vector<int32_t> ci, cd, cn;
    for (;;) {
        int rc = window_candidates(c, f, n_mp, proj_x, proj_y, qr.data(), minl.data(), maxl.data(), mp_desc, cap, ci, cd, cn);
        if (rc) return rc;
        int mx = 0;
        for (int v : cn) mx = std::max(mx, v);
        if (mx <= cap) break;
        cap = mx;                                                              // rare: a denser window than expected
    }
    int nmatches = 0;
    for (int m = 0; m < n_mp; m++) {
        if (!in_view[m] || cn[m] == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < cn[m]; k++) {
            const int idx = ci[(size_t)m * cap + k];
            if (occupied[idx]) continue;                                       // mvpMapPoints[idx] with Observations() > 0
            const int dist = cd[(size_t)m * cap + k];
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = f->kp_octave[idx]; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = f->kp_octave[idx]; bestDist2 = dist; }
        }
        if (bestDist <= 100) {                                                 // TH_HIGH
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match[bestIdx] = m;
            occupied[bestIdx] = mp_has_obs[m];
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th), ORBmatcher.cpp:1350-1476
int ccm_search_by_projection_frame(ccm_ctx* c, const ccm_frame_grid* f, const float* cur_angle, const float* scale_factors, int n_last,
                                   const uint8_t* valid, const float* u, const float* v, const int32_t* last_octave, const float* last_angle,
                                   const uint8_t* mp_desc, const uint8_t* mp_has_obs, uint8_t* occupied, float th, int check_ori,
                                   int orb_dist, int32_t* match)
{
    if (!c || !f) return CCM_E_ARG;
    if (f->n < 0 || n_last < 0 || (f->n > 0 && (!match || !occupied || (check_ori && !cur_angle))) ||
        (n_last > 0 && (!scale_factors || !valid || !u || !v || !last_octave || !mp_desc || !mp_has_obs || (check_ori && !last_angle))))
        return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection(frame, frame) arguments");
    for (int i = 0; i < f->n; i++) match[i] = -1;
    if (n_last == 0 || f->n == 0) return 0;
    CCM_HIP(c, hipSetDevice(c->device));
    std::vector<float> qr(n_last); std::vector<int32_t> minl(n_last), maxl(n_last);
    for (int i = 0; i < n_last; i++) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = 0; continue; }
        qr[i] = th * scale_factors[last_octave[i]];                           // :1401
        minl[i] = last_octave[i] - 1; maxl[i] = last_octave[i] + 1;           // :1405
    }
    static const bool host_accept = getenv("CCM_WINDOW_HOST_ACCEPT") && atoi(getenv("CCM_WINDOW_HOST_ACCEPT")) != 0;   // test switch
    if (!host_accept && match_window_greedy_lds(f->n, n_last) <= kGreedyLdsMax)
        return window_greedy(c, f, n_last, u, v, qr.data(), minl.data(), maxl.data(), mp_desc, 2, valid, nullptr, mp_has_obs, occupied, 0.f, match, f->n,
                             orb_dist, check_ori, last_angle, cur_angle);
    int cap = 64;
    std::vector<int32_t> ci, cd, cn;
    for (;;) {
        int rc = window_candidates(c, f, n_last, u, v, qr.data(), minl.data(), maxl.data(), mp_desc, cap, ci, cd, cn);
        if (rc) return rc;
        int mx = 0;
        for (int k : cn) mx = std::max(mx, k);
        if (mx <= cap) break;
        cap = mx;
    }
    const int HISTO = 30;
    std::vector<int> rot[HISTO];
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    for (int i = 0; i < n_last; i++) {
        if (!valid[i] || cn[i] == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < cn[i]; k++) {
            const int i2 = ci[(size_t)i * cap + k];
            if (occupied[i2]) continue;
            const int dist = cd[(size_t)i * cap + k];
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {                                            // TH_HIGH (:1432) / ORBdist (:1556)
            match[bestIdx2] = i;
            occupied[bestIdx2] = mp_has_obs[i];
            nmatches++;
            if (check_ori) {
                float r = last_angle[i] - cur_angle[bestIdx2];
                if (r < 0.0) r += 360.0f;
                int bin = (int)std::round(r * factor);
                if (bin == HISTO) bin = 0;
                rot[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(rot, HISTO, i1, i2, i3);
        for (int b = 0; b < HISTO; b++) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int idx : rot[b]) { match[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchForInitialization, ORBmatcher.cpp:448-563
int ccm_search_for_initialization(ccm_ctx* c, int n1, const int32_t* oct1, const uint8_t* desc1, const float* angle1,
                                  const ccm_frame_grid* f2, const float* angle2, float* prev_matched_xy, int window, float nnratio,
                                  int check_ori, int32_t* matches12)
{
    if (!c || !f2) return CCM_E_ARG;
    if (n1 < 0 || f2->n < 0 || (n1 > 0 && (!oct1 || !desc1 || !prev_matched_xy || !matches12 || (check_ori && !angle1))) ||
        (check_ori && f2->n > 0 && !angle2))
        return ccm_fail(c, CCM_E_ARG, "bad SearchForInitialization arguments");
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || f2->n == 0) return 0;
    CCM_HIP(c, hipSetDevice(c->device));
    std::vector<float> qx(n1), qy(n1), qr(n1); std::vector<int32_t> minl(n1), maxl(n1);
    for (int i = 0; i < n1; i++) {
        qx[i] = prev_matched_xy[2 * i]; qy[i] = prev_matched_xy[2 * i + 1];
        qr[i] = oct1[i] > 0 ? -1.f : (float)window;                             // :464-466 only level-0 features
        minl[i] = oct1[i]; maxl[i] = oct1[i];
    }
    int cap = 128;
    std::vector<int32_t> ci, cd, cn;
    for (;;) {
        int rc = window_candidates(c, f2, n1, qx.data(), qy.data(), qr.data(), minl.data(), maxl.data(), desc1, cap, ci, cd, cn);
        if (rc) return rc;
        int mx = 0;
        for (int k : cn) mx = std::max(mx, k);
        if (mx <= cap) break;
        cap = mx;
    }
    const int HISTO = 30;
    std::vector<int> rot[HISTO];
    const float factor = 1.0f / HISTO;
    std::vector<int> matched_dist(f2->n, INT32_MAX), m21(f2->n, -1);
    int nmatches = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        if (oct1[i1] > 0 || cn[i1] == 0) continue;
        int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
        for (int k = 0; k < cn[i1]; k++) {
            const int i2 = ci[(size_t)i1 * cap + k], dist = cd[(size_t)i1 * cap + k];
            if (matched_dist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= 50) {                                                   // TH_LOW
            if (bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { matches12[m21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2; m21[bestIdx2] = i1; matched_dist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    float r = angle1[i1] - angle2[bestIdx2];
                    if (r < 0.0) r += 360.0f;
                    int bin = (int)std::round(r * factor);
                    if (bin == HISTO) bin = 0;
                    rot[bin].push_back(i1);
                }
            }
        }
    }
    if (check_ori) {
        int a1, a2, a3;
        three_maxima(rot, HISTO, a1, a2, a3);
        for (int b = 0; b < HISTO; b++) {
            if (b == a1 || b == a2 || b == a3) continue;
            for (int idx1 : rot[b]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) { prev_matched_xy[2 * i1] = f2->kp_x[matches12[i1]]; prev_matched_xy[2 * i1 + 1] = f2->kp_y[matches12[i1]]; }
    return nmatches;
}

// Selection loop of ORBmatcher::Fuse, both overloads (ORBmatcher.cpp:914-955 and :1072-1100)
int ccm_fuse_select(ccm_ctx* c, const ccm_frame_grid* kf, const float* scale_factors, const float* inv_level_sigma2, int n_mp,
                    const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc, float th,
                    int chi2_check, int accept_th, int32_t* best_idx, int32_t* best_dist)
{
    if (!c || !kf) return CCM_E_ARG;
    if (n_mp < 0 || kf->n < 0 || (n_mp > 0 && (!valid || !u || !v || !level || !mp_desc || !best_idx || !best_dist || !scale_factors)) ||
        (chi2_check && !inv_level_sigma2))
        return ccm_fail(c, CCM_E_ARG, "bad Fuse arguments");
    for (int m = 0; m < n_mp; m++) { best_idx[m] = -1; best_dist[m] = 256; }
    if (n_mp == 0 || kf->n == 0) return CCM_OK;
    CCM_HIP(c, hipSetDevice(c->device));
    std::vector<float> qr(n_mp); std::vector<int32_t> lo(n_mp), hi(n_mp);
    int n_levels = 1;
    for (int m = 0; m < n_mp; m++) {
        qr[m] = valid[m] ? th * scale_factors[level[m]] : -1.f;                                  // :909 / :1068
        lo[m] = level[m] - 1; hi[m] = level[m];                                                  // :925-926 kpLevel in [level - 1, level]
    }
    for (int i = 0; i < kf->n; i++) n_levels = std::max(n_levels, kf->kp_octave[i] + 1);
    // the whole selection runs on the device (k_window_select): one (index, distance) per map point comes back, no candidate list
    std::vector<int32_t> d0, d1, d2;
    int rc = window_run(c, kf, n_mp, u, v, qr.data(), lo.data(), hi.data(), mp_desc, 0, 2, d0, d1, d2, chi2_check ? inv_level_sigma2 : nullptr,
                        n_levels, accept_th);
    if (rc) return rc;
    WindowBufs& W = *c->match->win;
    CCM_HIP(c, hipMemcpyAsync(best_idx, W.sel_i.p, (size_t)n_mp * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipMemcpyAsync(best_dist, W.sel_d.p, (size_t)n_mp * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    for (int m = 0; m < n_mp; m++) if (!valid[m]) { best_idx[m] = -1; best_dist[m] = 256; }
    return CCM_OK;
}

// The selection of ccm_fuse_select for n_kf keyframes in one launch: what n_kf sequential calls return (the selection reads the map
// points' projections and descriptors and the keyframe's features only -- what an earlier keyframe's Replace / AddObservation
// changes is which points the CALLER still applies, src/ORBmatcher.cpp:884-886, :958-990).  Map points projected into keyframe k are
// rows mp_first[k] .. mp_first[k + 1] - 1 of valid / u / v / level / mp_desc and of the outputs.
int ccm_fuse_select_batch(ccm_ctx* c, int n_kf, const ccm_frame_grid* kfs, const float* scale_factors, const float* inv_level_sigma2,
                          const int32_t* mp_first, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                          const uint8_t* mp_desc, float th, int chi2_check, int accept_th, int32_t* best_idx, int32_t* best_dist)
{
    if (!c) return CCM_E_ARG;
    if (n_kf < 0 || (n_kf > 0 && (!kfs || !mp_first))) return ccm_fail(c, CCM_E_ARG, "bad Fuse batch arguments");
    if (n_kf == 0) return CCM_OK;
    const int n_mp = mp_first[n_kf];
    if (mp_first[0] != 0 || n_mp < 0 || (n_mp > 0 && (!valid || !u || !v || !level || !mp_desc || !best_idx || !best_dist || !scale_factors)) ||
        (chi2_check && !inv_level_sigma2))
        return ccm_fail(c, CCM_E_ARG, "bad Fuse batch arguments");
    for (int k = 0; k < n_kf; k++) if (mp_first[k + 1] < mp_first[k] || kfs[k].n < 0) return ccm_fail(c, CCM_E_ARG, "bad Fuse batch arguments");
    for (int m = 0; m < n_mp; m++) { best_idx[m] = -1; best_dist[m] = 256; }
    if (n_mp == 0) return CCM_OK;
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->match) c->match = new MatchState();
    if (!c->match->win) c->match->win = new WindowBufs();
    WindowBufs& W = *c->match->win;
    // per keyframe: Frame::AssignFeaturesToGrid (as window_run), everything concatenated; a keyframe's items index its own features
    std::vector<int> feat_first(n_kf + 1, 0), cell_first_off(n_kf + 1, 0);
    for (int k = 0; k < n_kf; k++) { feat_first[k + 1] = feat_first[k] + kfs[k].n; cell_first_off[k + 1] = cell_first_off[k] + kfs[k].grid_cols * kfs[k].grid_rows + 1; }
    const int NF = feat_first[n_kf];
    std::vector<float> kx(std::max(NF, 1)), ky(std::max(NF, 1)), qr(n_mp);
    std::vector<int32_t> oct(std::max(NF, 1)), items(std::max(NF, 1)), cfirst(cell_first_off[n_kf]), lo(n_mp), hi(n_mp), qkf(n_mp);
    std::vector<uint8_t> fdesc((size_t)std::max(NF, 1) * 32);
    int n_levels = 1;
    for (int k = 0; k < n_kf; k++) {
        const ccm_frame_grid& f = kfs[k];
        const int n = f.n, cells = f.grid_cols * f.grid_rows, f0 = feat_first[k];
        int* first = cfirst.data() + cell_first_off[k];
        std::vector<int> cell(n);
        for (int q = 0; q <= cells; q++) first[q] = 0;
        for (int i = 0; i < n; i++) {
            const int px = (int)std::round((f.kp_x[i] - f.min_x) * f.inv_w), py = (int)std::round((f.kp_y[i] - f.min_y) * f.inv_h);
            cell[i] = (px < 0 || px >= f.grid_cols || py < 0 || py >= f.grid_rows) ? -1 : px * f.grid_rows + py;
            if (cell[i] >= 0) first[cell[i] + 1]++;
            kx[f0 + i] = f.kp_x[i]; ky[f0 + i] = f.kp_y[i]; oct[f0 + i] = f.kp_octave[i];
            n_levels = std::max(n_levels, f.kp_octave[i] + 1);
        }
        if (n) memcpy(fdesc.data() + (size_t)f0 * 32, f.desc, (size_t)n * 32);
        for (int q = 0; q < cells; q++) first[q + 1] += first[q];
        { std::vector<int> fill(first, first + cells); for (int i = 0; i < n; i++) if (cell[i] >= 0) items[f0 + fill[cell[i]]++] = i; }
        for (int m = mp_first[k]; m < mp_first[k + 1]; m++) {
            qkf[m] = k;
            qr[m] = (valid[m] && n > 0) ? th * scale_factors[level[m]] : -1.f;                    // :909 / :1068
            lo[m] = level[m] - 1; hi[m] = level[m];                                               // :925-926
        }
    }
    hipStream_t st = c->stream;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(W.kx, kx.data(), (size_t)NF * 4)) || (rc = up(W.ky, ky.data(), (size_t)NF * 4)) || (rc = up(W.oct, oct.data(), (size_t)NF * 4)) ||
        (rc = up(W.desc, fdesc.data(), (size_t)NF * 32)) || (rc = up(W.cfirst, cfirst.data(), cfirst.size() * 4)) || (rc = up(W.citems, items.data(), (size_t)NF * 4)) ||
        (rc = up(W.qx, u, (size_t)n_mp * 4)) || (rc = up(W.qy, v, (size_t)n_mp * 4)) || (rc = up(W.qr, qr.data(), (size_t)n_mp * 4)) ||
        (rc = up(W.minl, lo.data(), (size_t)n_mp * 4)) || (rc = up(W.maxl, hi.data(), (size_t)n_mp * 4)) || (rc = up(W.qdesc, mp_desc, (size_t)n_mp * 32)) ||
        (rc = up(W.qkf, qkf.data(), (size_t)n_mp * 4)))
        return rc;
    if (chi2_check && (rc = up(W.is2, inv_level_sigma2, (size_t)n_levels * 4))) return rc;
    std::vector<WinGrid> grids(n_kf);
    for (int k = 0; k < n_kf; k++) {
        const ccm_frame_grid& f = kfs[k];
        grids[k] = WinGrid{ f.n, f.grid_cols, f.grid_rows, f.min_x, f.min_y, f.inv_w, f.inv_h, W.kx.as<float>() + feat_first[k], W.ky.as<float>() + feat_first[k],
                            W.oct.as<int>() + feat_first[k], W.desc.as<uint8_t>() + (size_t)feat_first[k] * 32, W.cfirst.as<int>() + cell_first_off[k],
                            W.citems.as<int>() + feat_first[k] };
    }
    if ((rc = up(W.grids, grids.data(), grids.size() * sizeof(WinGrid)))) return rc;
    CCM_RESERVE(c, W.sel_i, (size_t)n_mp * 4); CCM_RESERVE(c, W.sel_d, (size_t)n_mp * 4);
    match_launch_window_select_batch(st, W.grids.as<WinGrid>(), W.qkf.as<int>(), n_mp, W.qx.as<float>(), W.qy.as<float>(), W.qr.as<float>(), W.minl.as<int>(),
                                     W.maxl.as<int>(), W.qdesc.as<uint8_t>(), chi2_check ? W.is2.as<float>() : nullptr, accept_th, W.sel_i.as<int>(), W.sel_d.as<int>());
    CCM_HIP(c, hipGetLastError());
    CCM_HIP(c, hipMemcpyAsync(best_idx, W.sel_i.p, (size_t)n_mp * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipMemcpyAsync(best_dist, W.sel_d.p, (size_t)n_mp * 4, hipMemcpyDeviceToHost, st));
    CCM_HIP(c, hipStreamSynchronize(st));                       // (the staging vectors above stay alive until here)
    for (int m = 0; m < n_mp; m++) if (!valid[m] || kfs[qkf[m]].n == 0) { best_idx[m] = -1; best_dist[m] = 256; }
    return CCM_OK;
}

// ORBmatcher::SearchBySim3, ORBmatcher.cpp:1124-1348: two selection passes (<= TH_HIGH) and the agreement check
int ccm_search_by_sim3(ccm_ctx* c, const ccm_frame_grid* kf1, const float* scale_factors1, const ccm_frame_grid* kf2, const float* scale_factors2,
                       const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* mp_desc1,
                       const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* mp_desc2,
                       float th, int32_t* match12)
{
    if (!c || !kf1 || !kf2) return CCM_E_ARG;
    if (kf1->n < 0 || kf2->n < 0 || (kf1->n > 0 && !match12)) return ccm_fail(c, CCM_E_ARG, "bad SearchBySim3 arguments");
    const int n1 = kf1->n, n2 = kf2->n;
    std::vector<int32_t> m1(std::max(n1, 1)), d1(std::max(n1, 1)), m2(std::max(n2, 1)), d2(std::max(n2, 1));
    // map points of KF1 (one per feature of KF1) are searched in KF2, and vice versa
    int rc = ccm_fuse_select(c, kf2, scale_factors2, nullptr, n1, valid1, u1, v1, level1, mp_desc1, th, 0, 100, m1.data(), d1.data());
    if (rc) return rc;
    rc = ccm_fuse_select(c, kf1, scale_factors1, nullptr, n2, valid2, u2, v2, level2, mp_desc2, th, 0, 100, m2.data(), d2.data());
    if (rc) return rc;
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }     // :1330-1345
    }
    return nFound;
}

// ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th), ORBmatcher.cpp:308-446
int ccm_search_by_projection_sim3(ccm_ctx* c, const ccm_frame_grid* kf, const float* scale_factors, int n_mp, const uint8_t* valid,
                                  const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc, const uint8_t* observed,
                                  uint8_t* matched, float th, int32_t* best_idx)
{
    if (!c || !kf) return CCM_E_ARG;
    if (n_mp < 0 || kf->n < 0 || (n_mp > 0 && (!valid || !u || !v || !level || !mp_desc || !observed || !best_idx || !scale_factors)) ||
        (kf->n > 0 && !matched))
        return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection(kf, Scw) arguments");
    for (int m = 0; m < n_mp; m++) best_idx[m] = -1;
    if (n_mp == 0 || kf->n == 0) return 0;
    CCM_HIP(c, hipSetDevice(c->device));
    std::vector<float> qr(n_mp); std::vector<int32_t> none(n_mp, -1);
    for (int m = 0; m < n_mp; m++) qr[m] = valid[m] ? th * scale_factors[level[m]] : -1.f;       // :380
    static const bool host_accept = getenv("CCM_WINDOW_HOST_ACCEPT") && atoi(getenv("CCM_WINDOW_HOST_ACCEPT")) != 0;   // test switch
    if (!host_accept && match_window_greedy_lds(kf->n, n_mp) <= kGreedyLdsMax)
        return window_greedy(c, kf, n_mp, u, v, qr.data(), none.data(), none.data(), mp_desc, 1, valid, level, observed, matched, 0.f, best_idx, n_mp);
    int cap = 64;
    std::vector<int32_t> ci, cd, cn;
    for (;;) {
        int rc = window_candidates(c, kf, n_mp, u, v, qr.data(), none.data(), none.data(), mp_desc, cap, ci, cd, cn);
        if (rc) return rc;
        int mx = 0;
        for (int k : cn) mx = std::max(mx, k);
        if (mx <= cap) break;
        cap = mx;
    }
    int nmatches = 0;
    for (int m = 0; m < n_mp; m++) {                       // sequential: vpMatched grows while the points are visited
        if (!valid[m]) continue;
        const int lvl = level[m];
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < cn[m]; k++) {
            const int idx = ci[(size_t)m * cap + k];
            if (matched[idx]) continue;                                                          // :394
            const int kpLevel = kf->kp_octave[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            const int dist = cd[(size_t)m * cap + k];
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) {                                                                    // TH_LOW
            best_idx[m] = bestIdx;
            if (!observed[m]) { matched[bestIdx] = 1; nmatches++; }                              // :436-440
        }
    }
    return nmatches;
}

// ccm_search_by_projection_sim3 for n_kf keyframes in ONE launch of each kernel (the loop closer matches the loop points into every
// keyframe connected to the current one, src/LoopFinder.cpp / MapMatcher.cpp: one SearchByProjection(pKF, Scw, ...) per keyframe).
// Keyframes are independent problems -- each has its own vpMatched -- so workgroup k of k_window_greedy takes keyframe k.
int ccm_search_by_projection_sim3_batch(ccm_ctx* c, int n_kf, const ccm_frame_grid* kfs, const float* scale_factors, const int32_t* mp_first,
                                        const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc,
                                        const uint8_t* observed, uint8_t* matched, float th, int32_t* best_idx, int32_t* n_matches)
{
    if (!c) return CCM_E_ARG;
    if (n_kf < 0 || (n_kf > 0 && (!kfs || !mp_first || !n_matches))) return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection(kf, Scw) batch arguments");
    if (n_kf == 0) return 0;
    const int n_mp = mp_first[n_kf];
    if (mp_first[0] != 0 || n_mp < 0 || (n_mp > 0 && (!valid || !u || !v || !level || !mp_desc || !observed || !best_idx || !scale_factors)))
        return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection(kf, Scw) batch arguments");
    std::vector<int> feat_first(n_kf + 1, 0), cell_first_off(n_kf + 1, 0);
    int max_n = 0;
    for (int k = 0; k < n_kf; k++) {
        if (mp_first[k + 1] < mp_first[k] || kfs[k].n < 0 || kfs[k].grid_cols < 1 || kfs[k].grid_rows < 1)
            return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection(kf, Scw) batch arguments");
        feat_first[k + 1] = feat_first[k] + kfs[k].n; cell_first_off[k + 1] = cell_first_off[k] + kfs[k].grid_cols * kfs[k].grid_rows + 1;
        max_n = std::max(max_n, kfs[k].n);
        n_matches[k] = 0;
    }
    const int NF = feat_first[n_kf];
    if (NF > 0 && !matched) return ccm_fail(c, CCM_E_ARG, "bad SearchByProjection(kf, Scw) batch arguments");
    for (int m = 0; m < n_mp; m++) best_idx[m] = -1;
    if (n_mp == 0 || NF == 0) return 0;
    static const bool host_accept = getenv("CCM_WINDOW_HOST_ACCEPT") && atoi(getenv("CCM_WINDOW_HOST_ACCEPT")) != 0;   // test switch
    if (host_accept || match_window_greedy_lds(max_n, 0) > kGreedyLdsMax) {          // keyframe by keyframe through the single entry point
        int total = 0;
        for (int k = 0; k < n_kf; k++) {
            const int q0 = mp_first[k], nq = mp_first[k + 1] - q0;
            const int r = ccm_search_by_projection_sim3(c, &kfs[k], scale_factors, nq, valid + q0, u + q0, v + q0, level + q0, mp_desc + (size_t)q0 * 32,
                                                        observed + q0, matched + feat_first[k], th, best_idx + q0);
            if (r < 0) return r;
            n_matches[k] = r; total += r;
        }
        return total;
    }
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->match) c->match = new MatchState();
    if (!c->match->win) c->match->win = new WindowBufs();
    WindowBufs& W = *c->match->win;
    // per keyframe: Frame::AssignFeaturesToGrid (as window_run), everything concatenated; a keyframe's items index its own features
    std::vector<float> kx(NF), ky(NF), qr(n_mp);
    std::vector<int32_t> oct(NF), items(NF), cfirst(cell_first_off[n_kf]), none(n_mp, -1), qkf(n_mp);
    std::vector<uint8_t> fdesc((size_t)NF * 32);
    std::vector<GreedyKf> gk(n_kf);
    for (int k = 0; k < n_kf; k++) {
        const ccm_frame_grid& f = kfs[k];
        const int n = f.n, cells = f.grid_cols * f.grid_rows, f0 = feat_first[k];
        int* first = cfirst.data() + cell_first_off[k];
        std::vector<int> cell(n);
        for (int q = 0; q <= cells; q++) first[q] = 0;
        for (int i = 0; i < n; i++) {
            const int px = (int)std::round((f.kp_x[i] - f.min_x) * f.inv_w), py = (int)std::round((f.kp_y[i] - f.min_y) * f.inv_h);
            cell[i] = (px < 0 || px >= f.grid_cols || py < 0 || py >= f.grid_rows) ? -1 : px * f.grid_rows + py;
            if (cell[i] >= 0) first[cell[i] + 1]++;
            kx[f0 + i] = f.kp_x[i]; ky[f0 + i] = f.kp_y[i]; oct[f0 + i] = f.kp_octave[i];
        }
        if (n) memcpy(fdesc.data() + (size_t)f0 * 32, f.desc, (size_t)n * 32);
        for (int q = 0; q < cells; q++) first[q + 1] += first[q];
        { std::vector<int> fill(first, first + cells); for (int i = 0; i < n; i++) if (cell[i] >= 0) items[f0 + fill[cell[i]]++] = i; }
        for (int m = mp_first[k]; m < mp_first[k + 1]; m++) {
            qkf[m] = k;
            qr[m] = (valid[m] && n > 0) ? th * scale_factors[level[m]] : -1.f;                    // :380
        }
        gk[k] = GreedyKf{ mp_first[k], mp_first[k + 1] - mp_first[k], f0, n };
    }
    hipStream_t st = c->stream;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        CCM_RESERVE(c, b, std::max<size_t>(bytes, 16));
        if (bytes) CCM_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return CCM_OK;
    };
    int rc;
    if ((rc = up(W.kx, kx.data(), (size_t)NF * 4)) || (rc = up(W.ky, ky.data(), (size_t)NF * 4)) || (rc = up(W.oct, oct.data(), (size_t)NF * 4)) ||
        (rc = up(W.desc, fdesc.data(), (size_t)NF * 32)) || (rc = up(W.cfirst, cfirst.data(), cfirst.size() * 4)) || (rc = up(W.citems, items.data(), (size_t)NF * 4)) ||
        (rc = up(W.qx, u, (size_t)n_mp * 4)) || (rc = up(W.qy, v, (size_t)n_mp * 4)) || (rc = up(W.qr, qr.data(), (size_t)n_mp * 4)) ||
        (rc = up(W.minl, none.data(), (size_t)n_mp * 4)) || (rc = up(W.maxl, none.data(), (size_t)n_mp * 4)) || (rc = up(W.qdesc, mp_desc, (size_t)n_mp * 32)) ||
        (rc = up(W.qkf, qkf.data(), (size_t)n_mp * 4)) || (rc = up(W.act, valid, (size_t)n_mp)) || (rc = up(W.qflag, observed, (size_t)n_mp)) ||
        (rc = up(W.qlvl, level, (size_t)n_mp * 4)) || (rc = up(W.gkf, gk.data(), gk.size() * sizeof(GreedyKf))))
        return rc;
    std::vector<WinGrid> grids(n_kf);
    for (int k = 0; k < n_kf; k++) {
        const ccm_frame_grid& f = kfs[k];
        grids[k] = WinGrid{ f.n, f.grid_cols, f.grid_rows, f.min_x, f.min_y, f.inv_w, f.inv_h, W.kx.as<float>() + feat_first[k], W.ky.as<float>() + feat_first[k],
                            W.oct.as<int>() + feat_first[k], W.desc.as<uint8_t>() + (size_t)feat_first[k] * 32, W.cfirst.as<int>() + cell_first_off[k],
                            W.citems.as<int>() + feat_first[k] };
    }
    if ((rc = up(W.grids, grids.data(), grids.size() * sizeof(WinGrid)))) return rc;
    CCM_RESERVE(c, W.out, (size_t)n_mp * 4); CCM_RESERVE(c, W.status, (size_t)n_kf * 12 + 16); CCM_RESERVE(c, W.cn, (size_t)n_mp * 4);
    std::vector<int> status(3 * (size_t)n_kf);
    int cap = 64;
    for (int attempt = 0; attempt < 3; attempt++) {
        CCM_RESERVE(c, W.ci, (size_t)n_mp * cap * 4); CCM_RESERVE(c, W.cd, (size_t)n_mp * cap * 4);
        if ((rc = up(W.flag, matched, (size_t)NF))) return rc;                // (again on a retry: the first attempt may have set flags)
        match_launch_window_batch(st, W.grids.as<WinGrid>(), W.qkf.as<int>(), n_mp, W.qx.as<float>(), W.qy.as<float>(), W.qr.as<float>(), W.minl.as<int>(),
                                  W.maxl.as<int>(), W.qdesc.as<uint8_t>(), cap, W.ci.as<int>(), W.cd.as<int>(), W.cn.as<int>());
        CCM_HIP(c, hipMemsetAsync(W.out.p, 0xFF, (size_t)n_mp * 4, st));
        GreedyArgs A{ 0, 0, cap, W.ci.as<int>(), W.cd.as<int>(), W.cn.as<int>(), W.act.as<uint8_t>(), W.qlvl.as<int>(), W.oct.as<int>(), W.qflag.as<uint8_t>(),
                      W.flag.as<uint8_t>(), 0.f, W.out.as<int>(), W.status.as<int>(), 0, 0, nullptr, nullptr, nullptr, W.gkf.as<GreedyKf>() };
        if (match_launch_window_greedy_batch(st, A, n_kf, max_n)) return ccm_fail(c, CCM_E_DEVICE, "k_window_greedy: LDS request refused");
        CCM_HIP(c, hipGetLastError());
        CCM_HIP(c, hipMemcpyAsync(status.data(), W.status.p, status.size() * 4, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        int need = 0;
        for (int k = 0; k < n_kf; k++) if (status[3 * k] < 0) need = std::max(need, status[3 * k + 1]);
        if (need > 0) { cap = need; continue; }                               // rare: a denser window than expected (in any keyframe: all repeat)
        CCM_HIP(c, hipMemcpyAsync(best_idx, W.out.p, (size_t)n_mp * 4, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipMemcpyAsync(matched, W.flag.p, (size_t)NF, hipMemcpyDeviceToHost, st));
        CCM_HIP(c, hipStreamSynchronize(st));
        int total = 0;
        for (int k = 0; k < n_kf; k++) { n_matches[k] = status[3 * k]; total += status[3 * k]; }
        return total;
    }
    return ccm_fail(c, CCM_E_CAPACITY, "window candidate lists keep overflowing");
}

// ORBmatcher::CheckDistEpipolarLine, ORBmatcher.cpp:159-176
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* F12, float sigma2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float cc = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + cc;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

// ORBmatcher::SearchForTriangulation, ORBmatcher.cpp:700-852
int ccm_search_for_triangulation(ccm_ctx* c, const uint8_t* desc1, const int32_t* node1, const uint8_t* has_mp1, const float* x1, const float* y1,
                                 const float* angle1, int n1, const uint8_t* desc2, const int32_t* node2, const uint8_t* has_mp2,
                                 const float* x2, const float* y2, const float* angle2, const int32_t* octave2, int n2, const float* F12,
                                 float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int check_ori, int32_t* match12)
{
    if (!c) return CCM_E_ARG;
    if (n1 < 0 || n2 < 0 || (n1 > 0 && (!desc1 || !node1 || !has_mp1 || !x1 || !y1 || !match12 || (check_ori && !angle1))) ||
        (n2 > 0 && (!desc2 || !node2 || !has_mp2 || !x2 || !y2 || !octave2 || (check_ori && !angle2))) || !F12 || !scale_factors2 || !level_sigma2_2)
        return ccm_fail(c, CCM_E_ARG, "bad SearchForTriangulation arguments");
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    CCM_HIP(c, hipSetDevice(c->device));
    std::vector<uint8_t> free1(n1);
    for (int i = 0; i < n1; i++) free1[i] = !has_mp1[i];                                         // :744-746
    BowRanges R;
    int rc = bow_ranges(c, desc1, node1, free1.data(), n1, desc2, node2, n2, R);
    if (rc) return rc;
    const int HISTO = 30;
    std::vector<int> rot[HISTO];
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    for (int i1 : R.ord1) {
        if (node1[i1] < 0 || !free1[i1] || R.len[i1] == 0) continue;
        int bestDist = 50, bestIdx2 = -1;                                                       // TH_LOW
        const unsigned short* d = R.dist.data() + R.off[i1];
        for (int k = 0; k < R.len[i1]; k++) {
            const int idx2 = R.ord2[R.start[i1] + k];
            if (has_mp2[idx2]) continue;                                                         // :763; vbMatched2 is never set
            const int dist = d[k];
            if (dist > 50 || dist > bestDist) continue;
            const float distex = ex - x2[idx2], distey = ey - y2[idx2];
            if (distex * distex + distey * distey < 100 * scale_factors2[octave2[idx2]]) continue;
            if (check_dist_epipolar_line(x1[i1], y1[i1], x2[idx2], y2[idx2], F12, level_sigma2_2[octave2[idx2]])) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            match12[i1] = bestIdx2;
            nmatches++;
            if (check_ori) {
                float r = angle1[i1] - angle2[bestIdx2];
                if (r < 0.0) r += 360.0f;
                int bin = (int)std::round(r * factor);
                if (bin == HISTO) bin = 0;
                rot[bin].push_back(i1);
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(rot, HISTO, i1, i2, i3);
        for (int i = 0; i < HISTO; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx : rot[i]) { match12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

}  // extern "C"
