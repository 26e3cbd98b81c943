// Drop-in bodies of cslam::Optimizer's bundle-adjustment entry points (include/cslam/Optimizer.h:84-97).
// Replaces in src/Optimizer.cpp: BundleAdjustmentClient (:32-164 + write-back :166-212), MapFusionGBA (:646-865) and
// PoseOptimizationClient (:215-347): the map is flattened into the arrays ccm_ba_solve / ccm_pose_optimize take, the result is
// written back where the reference writes it.  No g2o object is created.
#include <cslam/Optimizer.h>
#include <cslam/Converter.h>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/Map.h>
#include <cslam/MapPoint.h>
#include <cmath>
#include <map>
#include "ccm_shim.h"

namespace cslam {
namespace {

// keyframes / map points of one optimisation flattened the way ccm_ba_problem wants them
struct FlatGraph {
    std::vector<Optimizer::kfptr> kfs;                 // row -> keyframe
    std::vector<Optimizer::mpptr> mps;                 // point row -> map point
    std::map<size_t, int> kf_row;                      // KeyFrame::mUniqueId -> row
    std::vector<double> poses, intr, points, obs, info;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> edge_pose, edge_point;

    void add_keyframe(const Optimizer::kfptr& pKF, bool is_fixed)
    {
        double p7[7];
        const cv::Mat Tcw = pKF->GetPose();                                  // 4x4 CV_32F
        ccm_pose_from_mat4f(Tcw.ptr<float>(), p7);                           // Converter::toSE3Quat
        kf_row[pKF->mUniqueId] = (int)kfs.size();
        kfs.push_back(pKF);
        poses.insert(poses.end(), p7, p7 + 7);
        fixed.push_back(is_fixed ? 1 : 0);
        const double k[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
        intr.insert(intr.end(), k, k + 4);
    }

    // one landmark with the observations by keyframes of this graph; skipped below two of them (src/Optimizer.cpp:726-745)
    bool add_map_point(const Optimizer::mpptr& pMP)
    {
        const std::map<Optimizer::kfptr, size_t> observations = pMP->GetObservations();
        int usable = 0;
        for (const auto& ob : observations)
            if (ob.first && !ob.first->isBad() && kf_row.count(ob.first->mUniqueId)) usable++;
        if (usable < 2) return false;
        const cv::Mat Xw = pMP->GetWorldPos();                               // 3x1 CV_32F
        const int row = (int)mps.size();
        mps.push_back(pMP);
        for (int i = 0; i < 3; i++) points.push_back(Xw.at<float>(i));
        for (const auto& ob : observations) {
            const Optimizer::kfptr& pKF = ob.first;
            if (!pKF || pKF->isBad()) continue;
            const auto it = kf_row.find(pKF->mUniqueId);
            if (it == kf_row.end()) continue;
            const cv::KeyPoint& kp = pKF->mvKeysUn[ob.second];
            edge_pose.push_back(it->second); edge_point.push_back(row);
            obs.push_back(kp.pt.x); obs.push_back(kp.pt.y);
            info.push_back(pKF->mvInvLevelSigma2[kp.octave]);                // Identity * invSigma2 (:769-770)
        }
        return true;
    }

    ccm_ba_problem problem()
    {
        return ccm_ba_problem{(int)kfs.size(), poses.data(), fixed.data(), intr.data(), (int)mps.size(), points.data(),
                              (int)edge_pose.size(), edge_pose.data(), edge_point.data(), obs.data(), info.data()};
    }
};

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
    for (const mpptr& pMP : vpMP) if (!pMP->isBad()) g.add_map_point(pMP);

    ccm_ba_problem pb = g.problem();
    ccm_ba_options opt{nIterations, bRobust ? std::sqrt(5.99) : 0.0, 0, 5.991,
                       reinterpret_cast<const volatile uint8_t*>(pbStopFlag), /*pcg_tol=*/0.0};
    ccm_ba_result res{};
    if (ccm_ba_solve(ccm_shim::ctx(), &pb, &opt, &res)) throw estd::infrastructure_ex();

    for (size_t r = 0; r < g.kfs.size(); r++) {                               // :807-832
        const kfptr& pKF = g.kfs[r];
        const cv::Mat Tcw = pose_mat(&g.poses[7 * r]);
        pKF->mTcwBefGBA = pKF->GetPose();
        if (nLoopKF == zeropair) pKF->SetPose(Tcw, true);
        else { pKF->mTcwGBA.create(4, 4, CV_32F); Tcw.copyTo(pKF->mTcwGBA); pKF->mBAGlobalForKF = nLoopKF; }
    }
    for (size_t r = 0; r < g.mps.size(); r++) {                               // :836-862
        const mpptr& pMP = g.mps[r];
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
    for (const mpptr& pMP : vpMP) if (!pMP->isBad()) g.add_map_point(pMP);
    ccm_ba_problem pb = g.problem();
    ccm_ba_options opt{nIterations, bRobust ? std::sqrt(5.99) : 0.0, 0, 5.991,
                       reinterpret_cast<const volatile uint8_t*>(pbStopFlag), 0.0};
    ccm_ba_result res{};
    if (ccm_ba_solve(ccm_shim::ctx(), &pb, &opt, &res)) throw estd::infrastructure_ex();
    const bool direct = nLoopKF == std::make_pair((size_t)0, ClientId);      // :172
    for (size_t r = 0; r < g.kfs.size(); r++) {
        const kfptr& pKF = g.kfs[r];
        const cv::Mat Tcw = pose_mat(&g.poses[7 * r]);
        if (direct) pKF->SetPose(Tcw, false);
        else { pKF->mTcwGBA.create(4, 4, CV_32F); Tcw.copyTo(pKF->mTcwGBA); pKF->mBAGlobalForKF = nLoopKF; }
    }
    for (size_t r = 0; r < g.mps.size(); r++) {
        const mpptr& pMP = g.mps[r];
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
