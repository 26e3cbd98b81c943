// Drop-in bodies of cslam::Optimizer (include/cslam/Optimizer.h:84-113): every static entry point.
// Replaces in src/Optimizer.cpp: BundleAdjustmentClient (:32-164 + write-back :166-212), PoseOptimizationClient (:215-347),
// LocalBundleAdjustmentClient (:349-644), MapFusionGBA (:646-865), OptimizeSim3 (:867-1062), OptimizeEssentialGraphLoopClosure
// (:1064-1331) and OptimizeEssentialGraphMapFusion (:1333-1574): the map is flattened into the arrays ccm_ba_solve / ccm_pose_optimize /
// ccm_optimize_sim3 / ccm_optimize_essential_graph take, the result is written back where the reference writes it.  No g2o optimizer,
// vertex or edge is created; g2o::Sim3 (a header-only value type, and part of two of the signatures) is used for the measurements'
// inverse and product so that they are the reference's own arithmetic.
#include <cslam/Optimizer.h>
#include <cslam/Converter.h>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/Map.h>
#include <cslam/MapPoint.h>
#include <unistd.h>
#include <set>
#include <utility>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include "ccm_shim.h"
#include "flat_graph.h"

namespace cslam {
namespace {

// the reads of the flattening that go through OpenCV types (shim/flat_graph.h)
struct CvAccess {
    static void pose(const Optimizer::kfptr& pKF, float T[16])
    {
        const cv::Mat Tcw = pKF->GetPose();                                  // 4x4 CV_32F
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[4 * i + j] = Tcw.at<float>(i, j);
    }
    static void intrinsics(const Optimizer::kfptr& pKF, double k[4]) { k[0] = pKF->fx; k[1] = pKF->fy; k[2] = pKF->cx; k[3] = pKF->cy; }
    static void keypoint(const Optimizer::kfptr& pKF, size_t idx, double xy[2], double* inv_sigma2)
    {
        const cv::KeyPoint& kp = pKF->mvKeysUn[idx];
        xy[0] = kp.pt.x; xy[1] = kp.pt.y;
        *inv_sigma2 = pKF->mvInvLevelSigma2[kp.octave];
    }
    static void world_pos(const Optimizer::mpptr& pMP, float X[3])
    {
        const cv::Mat Xw = pMP->GetWorldPos();                               // 3x1 CV_32F
        for (int i = 0; i < 3; i++) X[i] = Xw.at<float>(i);
    }
};
typedef ccm_shim::FlatGraph<Optimizer::kfptr, Optimizer::mpptr, CvAccess> FlatGraph;

cv::Mat pose_mat(const double* p7)
{
    cv::Mat T(4, 4, CV_32F);
    ccm_pose_to_mat4f(p7, T.ptr<float>());                                   // Converter::toCvMat(SE3Quat)
    return T;
}

}  // namespace

void Optimizer::MapFusionGBA(mapptr pMap, size_t /*ClientId*/, int nIterations, bool* pbStopFlag, idpair nLoopKF, const bool bRobust)
{
    const std::vector<kfptr> vpKFs = pMap->GetAllKeyFrames();
    const std::vector<mpptr> vpMP = pMap->GetAllMapPoints();
    const idpair zeropair = std::make_pair(0, pMap->mMapId);
    if (pMap->mvpKeyFrameOrigins.empty()) throw estd::infrastructure_ex();
    const idpair FixedId = (*(pMap->mvpKeyFrameOrigins.begin()))->mId;

    FlatGraph g;
    for (const kfptr& pKF : vpKFs) if (!pKF->isBad()) g.add_keyframe(pKF, pKF->mId == FixedId);
    for (const mpptr& pMP : vpMP) if (!pMP->isBad()) g.add_map_point(pMP, 2);      // :726-745

    ccm_ba_problem pb = g.problem();
    ccm_ba_options opt{nIterations, bRobust ? (double)(float)std::sqrt(5.99) : 0.0, 0, 5.991,
                       reinterpret_cast<const volatile uint8_t*>(pbStopFlag), /*pcg_tol=*/0.0};
    ccm_ba_result res{};
    if (ccm_ba_solve(ccm_shim::ctx(), &pb, &opt, &res)) throw estd::infrastructure_ex();

    for (size_t r = 0; r < g.kfs.size(); r++) {                               // :807-832
        const kfptr& pKF = g.kfs[r];
        if (pKF->isBad()) continue;                                           // culled by another thread while the GBA ran (:810-812)
        const cv::Mat Tcw = pose_mat(&g.poses[7 * r]);
        pKF->mTcwBefGBA = pKF->GetPose();
        if (nLoopKF == zeropair) pKF->SetPose(Tcw, true);
        else { pKF->mTcwGBA.create(4, 4, CV_32F); Tcw.copyTo(pKF->mTcwGBA); pKF->mBAGlobalForKF = nLoopKF; }
    }
    for (size_t r = 0; r < g.mps.size(); r++) {                               // :836-862
        const mpptr& pMP = g.mps[r];
        if (pMP->isBad()) continue;                                           // :840-843
        cv::Mat Xw(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) Xw.at<float>(i) = (float)g.points[3 * r + i];
        if (nLoopKF == zeropair) { pMP->SetWorldPos(Xw, true); pMP->UpdateNormalAndDepth(); }
        else { pMP->mPosGBA.create(3, 1, CV_32F); Xw.copyTo(pMP->mPosGBA); pMP->mBAGlobalForKF = nLoopKF; }
    }
}

void Optimizer::BundleAdjustmentClient(const std::vector<kfptr>& vpKFs, const std::vector<mpptr>& vpMP, size_t ClientId,
                                       int nIterations, bool* pbStopFlag, const idpair nLoopKF, const bool bRobust)
{
    FlatGraph g;
    for (const kfptr& pKF : vpKFs) if (!pKF->isBad()) g.add_keyframe(pKF, pKF->mId == std::make_pair((size_t)0, ClientId));   // :75
    for (const mpptr& pMP : vpMP) if (!pMP->isBad()) g.add_map_point(pMP, 1);      // nEdges == 0 removes the vertex (:133-141)
    ccm_ba_problem pb = g.problem();
    ccm_ba_options opt{nIterations, bRobust ? (double)(float)std::sqrt(5.99) : 0.0, 0, 5.991,
                       reinterpret_cast<const volatile uint8_t*>(pbStopFlag), 0.0};
    ccm_ba_result res{};
    if (ccm_ba_solve(ccm_shim::ctx(), &pb, &opt, &res)) throw estd::infrastructure_ex();
    const bool direct = nLoopKF == std::make_pair((size_t)0, ClientId);      // :172
    for (size_t r = 0; r < g.kfs.size(); r++) {
        const kfptr& pKF = g.kfs[r];
        if (pKF->isBad()) continue;                                           // :170-172
        const cv::Mat Tcw = pose_mat(&g.poses[7 * r]);
        if (direct) pKF->SetPose(Tcw, false);
        else { pKF->mTcwGBA.create(4, 4, CV_32F); Tcw.copyTo(pKF->mTcwGBA); pKF->mBAGlobalForKF = nLoopKF; }
    }
    for (size_t r = 0; r < g.mps.size(); r++) {
        const mpptr& pMP = g.mps[r];
        if (pMP->isBad()) continue;                                           // :195-196
        cv::Mat Xw(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) Xw.at<float>(i) = (float)g.points[3 * r + i];
        if (direct) { pMP->SetWorldPos(Xw, false); pMP->UpdateNormalAndDepth(); }
        else { pMP->mPosGBA.create(3, 1, CV_32F); Xw.copyTo(pMP->mPosGBA); pMP->mBAGlobalForKF = nLoopKF; }
    }
}

void Optimizer::GlobalBundleAdjustemntClient(mapptr pMap, size_t ClientId, int nIterations, bool* pbStopFlag,
                                             const idpair nLoopKF, const bool bRobust)
{
    BundleAdjustmentClient(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), ClientId, nIterations, pbStopFlag, nLoopKF, bRobust);   // :32-37
}

// src/Optimizer.cpp:349-644, called by LocalMapping (src/Mapping.cpp:97) after every keyframe insertion: BASELINE config 4.
void Optimizer::LocalBundleAdjustmentClient(kfptr pKF, bool* pbStopFlag, mapptr pMap, size_t ClientId, eSystemState SysState)
{
    // local keyframes (the current one + covisible neighbours), local map points, fixed keyframes: :351-406
    std::list<kfptr> lLocalKeyFrames, lFixedCameras;
    std::list<mpptr> lLocalMapPoints;
    ccm_shim::gather_local_ba(pKF, lLocalKeyFrames, lLocalMapPoints, lFixedCameras);

    FlatGraph g;
    for (const kfptr& pKFi : lLocalKeyFrames) {                                                   // :424-441
        if (pKFi->mId.first >= IDRANGE) throw estd::infrastructure_ex();
        g.add_keyframe(pKFi, pKFi->mId.first == 0 && pKFi->mId.second == ClientId);
    }
    const size_t n_local = g.kfs.size();
    for (const kfptr& pKFi : lFixedCameras) {                                                     // :443-460
        if (pKFi->mId.first >= IDRANGE) throw estd::infrastructure_ex();
        g.add_keyframe(pKFi, true);
    }
    for (const mpptr& pMP : lLocalMapPoints) {                                                    // :476-538: one edge per observation by a keyframe that is not bad
        if (pMP->mId.first >= IDRANGE) throw estd::infrastructure_ex();
        g.add_map_point(pMP, 0);
    }
    if (pbStopFlag && *pbStopFlag) return;                                                        // :540-542

    // optimize(5) with the Huber kernel, then chi2 > 5.991 or non-positive depth -> level 1, kernels dropped, optimize(10) (:544-575).
    // ccm_ba_solve runs that schedule in one call and skips the second stage when the stop flag came up during the first (bDoMore).
    // (An edge whose map point another thread has culled in between keeps level 0 and its kernel in the reference, :555-556; here
    // every edge is classified.)
    ccm_ba_problem pb = g.problem();
    std::vector<uint8_t> outlier(g.edge_pose.size() + 1, 0);
    ccm_ba_options opt{5, (double)(float)std::sqrt(5.991), 10, 5.991, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), /*pcg_tol=*/0.0};
    ccm_ba_result res{};
    res.edge_outlier = outlier.data();
    if (ccm_ba_solve(ccm_shim::ctx(), &pb, &opt, &res)) throw estd::infrastructure_ex();

    // observations to erase: the same test on the final state (:577-595)
    std::vector<std::pair<kfptr, mpptr> > vToErase;
    vToErase.reserve(g.edge_pose.size());
    for (size_t e = 0; e < g.edge_pose.size(); e++) {
        const mpptr& pMP = g.mps[g.edge_point[e]];
        if (pMP->isBad()) continue;
        if (outlier[e]) vToErase.push_back(std::make_pair(g.edge_kf[e], pMP));
    }

    if (SysState != eSystemState::SERVER)                                                         // :597-599
        while (!pMap->LockMapUpdate()) { usleep(params::timings::miLockSleep); }
    for (const auto& er : vToErase) {                                                             // :601-610
        er.first->EraseMapPointMatch(er.second);
        er.second->EraseObservation(er.first);
    }
    for (size_t r = 0; r < n_local; r++) {                                                        // :614-624 (local keyframes only: the fixed ones did not move)
        g.kfs[r]->SetPose(pose_mat(&g.poses[7 * r]), false);
        g.kfs[r]->mbUpdatedByServer = false;
    }
    for (size_t r = 0; r < g.mps.size(); r++) {                                                   // :626-644
        const mpptr& pMP = g.mps[r];
        if (pMP->isBad()) {
            // its observations may have been erased above; then the map must have dropped it
            if (pMap->GetMpPtr(pMP->mId)) throw estd::infrastructure_ex();
        } else {
            cv::Mat Xw(3, 1, CV_32F);
            for (int i = 0; i < 3; i++) Xw.at<float>(i) = (float)g.points[3 * r + i];
            pMP->SetWorldPos(Xw, false);
            pMP->UpdateNormalAndDepth();
        }
    }
    if (SysState != eSystemState::SERVER) pMap->UnLockMapUpdate();
}

int Optimizer::PoseOptimizationClient(Frame& F)
{
    const int N = F.N;
    std::vector<double> Xw, obs, info; std::vector<int> feat;
    for (int i = 0; i < N; i++) {                                             // :244-281
        const mpptr pMP = F.mvpMapPoints[i];
        if (!pMP) continue;
        F.mvbOutlier[i] = false;
        const cv::KeyPoint& kp = F.mvKeysUn[i];
        const cv::Mat P = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) Xw.push_back(P.at<float>(k));
        obs.push_back(kp.pt.x); obs.push_back(kp.pt.y);
        info.push_back(F.mvInvLevelSigma2[kp.octave]);
        feat.push_back(i);
    }
    const int n = (int)feat.size();
    if (n < 3) return 0;                                                      // :285-286
    double pose7[7]; ccm_pose_from_mat4f(F.mTcw.ptr<float>(), pose7);
    const double K4[4] = {F.fx, F.fy, F.cx, F.cy};
    const int32_t first[2] = {0, n};
    std::vector<uint8_t> outlier(n); int32_t nInliers = 0;
    ccm_pose_problem pp{1, pose7, K4, first, Xw.data(), obs.data(), info.data(), outlier.data(), &nInliers};
    if (ccm_pose_optimize(ccm_shim::ctx(), &pp)) throw estd::infrastructure_ex();
    for (int e = 0; e < n; e++) F.mvbOutlier[feat[e]] = outlier[e] != 0;      // :318-333
    F.SetPose(pose_mat(pose7));                                               // :341-344
    return nInliers;                                                          // nInitialCorrespondences - nBad
}


// ---- Sim3 problems of loop closing and map fusion
namespace {
void sim3_to8(const g2o::Sim3& S, double* o)
{
    const Eigen::Quaterniond& q = S.rotation();
    o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); o[3] = q.w();
    o[4] = S.translation()[0]; o[5] = S.translation()[1]; o[6] = S.translation()[2]; o[7] = S.scale();
}
g2o::Sim3 sim3_from8(const double* o)
{
    return g2o::Sim3(Eigen::Quaterniond(o[3], o[0], o[1], o[2]), Eigen::Vector3d(o[4], o[5], o[6]), o[7]);
}
}  // namespace

int Optimizer::OptimizeSim3(kfptr pKF1, kfptr pKF2, std::vector<mpptr>& vpMatches1, g2o::Sim3& g2oS12, const float th2, bool bFixScale)
{
    const cv::Mat& K1 = pKF1->mK;
    const cv::Mat& K2 = pKF2->mK;
    const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    const int N = (int)vpMatches1.size();
    const std::vector<mpptr> vpMapPoints1 = pKF1->GetMapPointMatches();
    std::vector<double> P1, P2, obs1, obs2, info1, info2;
    std::vector<size_t> vnIndexEdge;
    for (int i = 0; i < N; i++) {                                             // :918-990
        if (!vpMatches1[i]) continue;
        const mpptr pMP1 = vpMapPoints1[i], pMP2 = vpMatches1[i];
        const int i2 = pMP2->GetIndexInKeyFrame(pKF2);
        if (!pMP1 || !pMP2 || pMP1->isBad() || pMP2->isBad() || i2 < 0) continue;
        const cv::Mat P3D1c = R1w * pMP1->GetWorldPos() + t1w, P3D2c = R2w * pMP2->GetWorldPos() + t2w;
        for (int k = 0; k < 3; k++) { P1.push_back(P3D1c.at<float>(k)); P2.push_back(P3D2c.at<float>(k)); }
        const cv::KeyPoint& kpUn1 = pKF1->mvKeysUn[i];
        const cv::KeyPoint& kpUn2 = pKF2->mvKeysUn[i2];
        obs1.push_back(kpUn1.pt.x); obs1.push_back(kpUn1.pt.y); info1.push_back(pKF1->mvInvLevelSigma2[kpUn1.octave]);
        obs2.push_back(kpUn2.pt.x); obs2.push_back(kpUn2.pt.y); info2.push_back(pKF2->mvInvLevelSigma2[kpUn2.octave]);
        vnIndexEdge.push_back(i);
    }
    const int nC = (int)vnIndexEdge.size();
    double S[8], S0[8];
    sim3_to8(g2oS12, S);
    memcpy(S0, S, sizeof S);
    const int32_t fix = bFixScale ? 1 : 0, first[2] = { 0, nC };
    const double k1[4] = { K1.at<float>(0, 0), K1.at<float>(1, 1), K1.at<float>(0, 2), K1.at<float>(1, 2) };
    const double k2[4] = { K2.at<float>(0, 0), K2.at<float>(1, 1), K2.at<float>(0, 2), K2.at<float>(1, 2) };
    std::vector<uint8_t> inlier(std::max(nC, 1), 0);
    int32_t nIn = 0;
    ccm_sim3_problem pr{};
    pr.n_problems = 1; pr.sim3 = S; pr.fix_scale = &fix; pr.K1 = k1; pr.K2 = k2; pr.first = first;
    pr.P1 = P1.data(); pr.P2 = P2.data(); pr.obs1 = obs1.data(); pr.obs2 = obs2.data(); pr.info1 = info1.data(); pr.info2 = info2.data();
    pr.th2 = &th2; pr.inlier = inlier.data(); pr.n_inliers = &nIn;
    if (ccm_optimize_sim3(ccm_shim::ctx(), &pr)) throw estd::infrastructure_ex();
    // :1015-1058: an outlier of either round loses its match; with fewer than 10 survivors of the first round the reference returns 0
    // before the second (the first round's outliers are already nulled, g2oS12 untouched): inlier[] and sim3 come back that way
    for (int e = 0; e < nC; e++) if (!inlier[e]) vpMatches1[vnIndexEdge[e]] = static_cast<mpptr>(NULL);
    if (memcmp(S, S0, sizeof S) != 0) g2oS12 = sim3_from8(S);              // (left alone on the early return)
    return nIn;
}

namespace {
// both essential-graph entry points: the map-fusion one is the loop-closure one without Sim3 maps and with the _MM correction marks
void essential_graph(Optimizer::mapptr pMap, Optimizer::kfptr pLoopKF, Optimizer::kfptr pCurKF, const Optimizer::KeyFrameAndPose* NonCorrectedSim3,
                     const Optimizer::KeyFrameAndPose* CorrectedSim3, const std::map<Optimizer::kfptr, std::set<Optimizer::kfptr> >& LoopConnections,
                     bool bFixScale, bool map_fusion)
{
    typedef Optimizer::kfptr kfptr;
    typedef Optimizer::mpptr mpptr;
    typedef Optimizer::KeyFrameAndPose KeyFrameAndPose;
    const std::vector<kfptr> vpKFs = pMap->GetAllKeyFrames();
    const std::vector<mpptr> vpMPs = pMap->GetAllMapPoints();
    const unsigned int nMaxKFid = pMap->GetMaxKFidUnique();
    std::vector<g2o::Sim3, Eigen::aligned_allocator<g2o::Sim3> > vScw(nMaxKFid + 1);
    std::vector<int32_t> vertex_of(nMaxKFid + 1, -1);                           // mUniqueId -> vertex of the flat problem
    const int minFeat = params::opt::miEssGraphMinFeats;
    std::vector<double> sim3;                                                   // [n_vertices][8]
    std::vector<uint8_t> fixed;
    // ---- keyframe vertices (:1086-1121)
    for (size_t i = 0; i < vpKFs.size(); i++) {
        const kfptr pKF = vpKFs[i];
        if (pKF->isBad()) continue;
        const size_t nIDi = pKF->mUniqueId;
        KeyFrameAndPose::const_iterator it;
        if (CorrectedSim3 && (it = CorrectedSim3->find(pKF)) != CorrectedSim3->end()) vScw[nIDi] = it->second;
        else vScw[nIDi] = g2o::Sim3(Converter::toMatrix3d(pKF->GetRotation()), Converter::toVector3d(pKF->GetTranslation()), 1.0);
        vertex_of[nIDi] = (int32_t)fixed.size();
        sim3.resize(sim3.size() + 8);
        sim3_to8(vScw[nIDi], &sim3[sim3.size() - 8]);
        fixed.push_back(pKF == pLoopKF ? 1 : 0);
    }
    const std::vector<double> sim3_before = sim3;
    std::vector<int32_t> ei, ej;
    std::vector<double> meas;
    // an edge whose end has no vertex (a bad keyframe) is not added: g2o refuses an edge with a null vertex
    auto add_edge = [&](size_t nIDi, size_t nIDj, const g2o::Sim3& Sji) {
        if (vertex_of[nIDi] < 0 || vertex_of[nIDj] < 0) return;
        ei.push_back(vertex_of[nIDi]); ej.push_back(vertex_of[nIDj]);
        meas.resize(meas.size() + 8);
        sim3_to8(Sji, &meas[meas.size() - 8]);
    };
    auto non_corrected = [&](const kfptr& k) -> const g2o::Sim3* {
        if (!NonCorrectedSim3) return nullptr;
        KeyFrameAndPose::const_iterator it = NonCorrectedSim3->find(k);
        return it != NonCorrectedSim3->end() ? &it->second : nullptr;
    };
    std::set<std::pair<long unsigned int, long unsigned int> > sInsertedEdges;
    // ---- loop edges (:1127-1157)
    for (std::map<kfptr, std::set<kfptr> >::const_iterator mit = LoopConnections.begin(); mit != LoopConnections.end(); ++mit) {
        const kfptr pKF = mit->first;
        if (pKF->isBad()) continue;
        const size_t nIDi = pKF->mUniqueId;
        const g2o::Sim3 Swi = vScw[nIDi].inverse();
        for (std::set<kfptr>::const_iterator sit = mit->second.begin(); sit != mit->second.end(); ++sit) {
            if ((*sit)->isBad()) continue;
            const size_t nIDj = (*sit)->mUniqueId;
            if ((nIDi != pCurKF->mUniqueId || nIDj != pLoopKF->mUniqueId) && pKF->GetWeight(*sit) < minFeat) continue;
            add_edge(nIDi, nIDj, vScw[nIDj] * Swi);
            sInsertedEdges.insert(std::make_pair(std::min(nIDi, nIDj), std::max(nIDi, nIDj)));
        }
    }
    // ---- spanning tree, earlier loop edges, strong covisibility edges (:1159-1250)
    for (size_t i = 0; i < vpKFs.size(); i++) {
        const kfptr pKF = vpKFs[i];
        const size_t nIDi = pKF->mUniqueId;
        const g2o::Sim3* nc = non_corrected(pKF);
        const g2o::Sim3 Swi = nc ? nc->inverse() : vScw[nIDi].inverse();
        auto Sxw = [&](const kfptr& k) -> g2o::Sim3 { const g2o::Sim3* n = non_corrected(k); return n ? *n : vScw[k->mUniqueId]; };
        const kfptr pParentKF = pKF->GetParent();
        if (pParentKF) add_edge(nIDi, pParentKF->mUniqueId, Sxw(pParentKF) * Swi);
        const std::set<kfptr> sLoopEdges = pKF->GetLoopEdges();
        for (std::set<kfptr>::const_iterator sit = sLoopEdges.begin(); sit != sLoopEdges.end(); ++sit) {
            const size_t nIDj = (*sit)->mUniqueId;
            if (nIDj < nIDi) add_edge(nIDi, nIDj, Sxw(*sit) * Swi);
        }
        const std::vector<kfptr> vpConnectedKFs = pKF->GetCovisiblesByWeight(minFeat);
        for (std::vector<kfptr>::const_iterator vit = vpConnectedKFs.begin(); vit != vpConnectedKFs.end(); ++vit) {
            const kfptr pKFn = *vit;
            if (pKFn->isBad()) continue;
            if (pKFn && pKFn != pParentKF && !pKF->hasChild(pKFn) && !sLoopEdges.count(pKFn)) {
                const size_t nIDj = pKFn->mUniqueId;
                if (nIDj < nIDi) {
                    if (sInsertedEdges.count(std::make_pair(std::min(nIDi, nIDj), std::max(nIDi, nIDj)))) continue;
                    add_edge(nIDi, nIDj, Sxw(pKFn) * Swi);
                }
            }
        }
    }
    // ---- optimize(20), Levenberg with lambda 1e-16 (:1072-1076, :1253-1254)
    ccm_essential_graph eg{};
    eg.n_vertices = (int32_t)fixed.size(); eg.sim3 = sim3.data(); eg.fixed = fixed.data(); eg.fix_scale = bFixScale ? 1 : 0;
    eg.n_edges = (int32_t)ei.size(); eg.edge_i = ei.data(); eg.edge_j = ej.data(); eg.measurement = meas.data(); eg.iterations = 20;
    if (ccm_optimize_essential_graph(ccm_shim::ctx(), &eg)) throw estd::infrastructure_ex();
    // ---- SE3 pose recovering. Sim3:[sR t;0 1] -> SE3:[R t/s;0 1] (:1256-1276)
    for (size_t i = 0; i < vpKFs.size(); i++) {
        const kfptr pKFi = vpKFs[i];
        const int32_t vtx = vertex_of[pKFi->mUniqueId];
        if (vtx < 0) continue;
        const g2o::Sim3 CorrectedSiw = sim3_from8(&sim3[8 * (size_t)vtx]);
        const Eigen::Matrix3d eigR = CorrectedSiw.rotation().toRotationMatrix();
        Eigen::Vector3d eigt = CorrectedSiw.translation();
        eigt *= (1. / CorrectedSiw.scale());
        pKFi->SetPose(Converter::toCvSE3(eigR, eigt), true);
    }
    // ---- correct the map points through their reference keyframe (:1278-1330)
    const int nMP = (int)vpMPs.size();
    std::vector<double> pts((size_t)3 * std::max(nMP, 1), 0.0);
    std::vector<int32_t> ref(std::max(nMP, 1), -1);
    for (int i = 0; i < nMP; i++) {
        const mpptr pMP = vpMPs[i];
        if (pMP->isBad()) continue;
        size_t nIDr;
        if (map_fusion ? pMP->mCorrectedByKF_MM == pCurKF->mId : pMP->mCorrectedByKF_LC == pCurKF->mId)
            nIDr = map_fusion ? pMP->mCorrectedReference_MM : pMP->mCorrectedReference_LC;
        else nIDr = pMP->GetReferenceKeyFrame()->mUniqueId;
        if (nIDr > nMaxKFid) continue;
        ref[i] = vertex_of[nIDr];
        const cv::Mat P3Dw = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) pts[3 * (size_t)i + k] = P3Dw.at<float>(k);
    }
    if (ccm_correct_map_points(ccm_shim::ctx(), nMP, pts.data(), ref.data(), eg.n_vertices, sim3_before.data(), sim3.data()))
        throw estd::infrastructure_ex();
    for (int i = 0; i < nMP; i++) {
        if (ref[i] < 0) continue;
        const mpptr pMP = vpMPs[i];
        pMP->SetWorldPos(Converter::toCvMat(Eigen::Matrix<double, 3, 1>(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2])), true);
        pMP->UpdateNormalAndDepth();
    }
}
}  // namespace

void Optimizer::OptimizeEssentialGraphLoopClosure(mapptr pMap, kfptr pLoopKF, kfptr pCurKF, const KeyFrameAndPose& NonCorrectedSim3,
                                                  const KeyFrameAndPose& CorrectedSim3, const map<kfptr, set<kfptr> >& LoopConnections,
                                                  const bool& bFixScale)
{
    essential_graph(pMap, pLoopKF, pCurKF, &NonCorrectedSim3, &CorrectedSim3, LoopConnections, bFixScale, false);
}

void Optimizer::OptimizeEssentialGraphMapFusion(mapptr pMap, kfptr pLoopKF, kfptr pCurKF, const map<kfptr, set<kfptr> >& LoopConnections,
                                                const bool& bFixScale)
{
    essential_graph(pMap, pLoopKF, pCurKF, nullptr, nullptr, LoopConnections, bFixScale, true);
}

}  // namespace cslam
