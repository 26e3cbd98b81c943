// Drop-in bodies of cslam::Optimizer's bundle-adjustment entry points (include/cslam/Optimizer.h:84-97).
// Replaces in src/Optimizer.cpp: BundleAdjustmentClient (:32-164 + write-back :166-212), PoseOptimizationClient (:215-347),
// LocalBundleAdjustmentClient (:349-644) and MapFusionGBA (:646-865): the map is flattened into the arrays ccm_ba_solve /
// ccm_pose_optimize take, the result is written back where the reference writes it.  No g2o object is created.
#include <cslam/Optimizer.h>
#include <cslam/Converter.h>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/Map.h>
#include <cslam/MapPoint.h>
#include <unistd.h>
#include <cmath>
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

}  // namespace cslam
