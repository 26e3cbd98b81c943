// flat_graph.h -- the map-side logic of the Optimizer drop-ins, free of OpenCV / g2o / ROS so that it compiles and is tested
// on its own (tests/support/shim_flatten_check.cpp drives it with plain stand-in keyframe / map point classes):
//   * FlatGraph: keyframes, map points and observations flattened into the arrays ccm_ba_problem takes, with the reference's
//     per-entry-point rule for which map points become vertices (src/Optimizer.cpp:95-160, :468-538, :726-785);
//   * gather_local_ba: the covisibility walk of Optimizer::LocalBundleAdjustmentClient (src/Optimizer.cpp:351-406).
// Everything is a template over the pointer types; the map classes are used through the reference's own member names
// (isBad(), GetObservations(), mUniqueId, mBALocalForKF, ...).  `Access` supplies the three reads that go through OpenCV
// types in the reference:
//     static void pose(const KfPtr&, float T[16]);                          KeyFrame::GetPose(), 4x4 row-major
//     static void intrinsics(const KfPtr&, double k[4]);                    fx, fy, cx, cy
//     static void keypoint(const KfPtr&, size_t idx, double xy[2], double* inv_sigma2);   mvKeysUn[idx].pt, mvInvLevelSigma2[octave]
//     static void world_pos(const MpPtr&, float X[3]);                      MapPoint::GetWorldPos()
#pragma once
#include <ccm_hot.h>
#include <cstddef>
#include <cstdint>
#include <list>
#include <map>
#include <vector>

namespace ccm_shim {

template <class KfPtr, class MpPtr, class Access>
struct FlatGraph {
    std::vector<KfPtr> kfs;                            // row -> keyframe
    std::vector<MpPtr> mps;                            // point row -> map point
    std::map<size_t, int> kf_row;                      // KeyFrame::mUniqueId -> row
    std::vector<double> poses, intr, points, obs, info;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> edge_pose, edge_point;
    std::vector<KfPtr> edge_kf;                        // edge -> observing keyframe (vpEdgeKFMono of the local BA)

    void add_keyframe(const KfPtr& pKF, bool is_fixed)
    {
        float T[16];
        double p7[7], k[4];
        Access::pose(pKF, T);
        ccm_pose_from_mat4f(T, p7);                                          // Converter::toSE3Quat
        kf_row[pKF->mUniqueId] = (int)kfs.size();
        kfs.push_back(pKF);
        poses.insert(poses.end(), p7, p7 + 7);
        fixed.push_back(is_fixed ? 1 : 0);
        Access::intrinsics(pKF, k);
        intr.insert(intr.end(), k, k + 4);
    }

    // One landmark with its observations by the keyframes of this graph.  min_obs is the reference's rule for the entry point:
    //   MapFusionGBA                2  ("at least 2 proper observations", src/Optimizer.cpp:726-745: the point is left out and never written back)
    //   BundleAdjustmentClient      1  (every non-bad point whose nEdges >= 1 stays, :133-141; only nEdges == 0 removes the vertex)
    //   LocalBundleAdjustmentClient 0  (every local map point is a vertex and is written back, :468-520, :626-644)
    // An observation counts when its keyframe is set, not bad and a vertex of this graph.  Returns false when the point was
    // left out (vbNotIncludedMP of the reference).
    bool add_map_point(const MpPtr& pMP, int min_obs)
    {
        const auto observations = pMP->GetObservations();                    // std::map<kfptr, size_t>, by value like the reference
        int usable = 0;
        for (const auto& ob : observations)
            if (ob.first && !ob.first->isBad() && kf_row.count(ob.first->mUniqueId)) usable++;
        if (usable < min_obs) return false;
        float X[3];
        Access::world_pos(pMP, X);
        const int row = (int)mps.size();
        mps.push_back(pMP);
        for (int i = 0; i < 3; i++) points.push_back((double)X[i]);
        for (const auto& ob : observations) {
            const KfPtr& pKF = ob.first;
            if (!pKF || pKF->isBad()) continue;
            const auto it = kf_row.find(pKF->mUniqueId);
            if (it == kf_row.end()) continue;
            double xy[2], inv_sigma2;
            Access::keypoint(pKF, ob.second, xy, &inv_sigma2);
            edge_pose.push_back(it->second); edge_point.push_back(row);
            edge_kf.push_back(pKF);
            obs.push_back(xy[0]); obs.push_back(xy[1]);
            info.push_back(inv_sigma2);                                      // Identity * invSigma2 (:769-770)
        }
        return true;
    }

    ccm_ba_problem problem()
    {
        return ccm_ba_problem{(int)kfs.size(), poses.data(), fixed.data(), intr.data(), (int)mps.size(), points.data(),
                              (int)edge_pose.size(), edge_pose.data(), edge_point.data(), obs.data(), info.data()};
    }
};

// The three lists of Optimizer::LocalBundleAdjustmentClient (src/Optimizer.cpp:351-406), with its marks: mBALocalForKF on the
// current keyframe, on EVERY covisible neighbour (bad ones too: they are marked, not listed) and on the local map points;
// mBAFixedForKF on every other observer of a local map point (bad ones again marked, not listed).
template <class KfPtr, class MpPtr>
void gather_local_ba(const KfPtr& pKF, std::list<KfPtr>& lLocalKeyFrames, std::list<MpPtr>& lLocalMapPoints, std::list<KfPtr>& lFixedCameras)
{
    lLocalKeyFrames.push_back(pKF);
    pKF->mBALocalForKF = pKF->mId;
    const std::vector<KfPtr> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (const KfPtr& pKFi : vNeighKFs) {
        pKFi->mBALocalForKF = pKF->mId;
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    for (const KfPtr& pKFi : lLocalKeyFrames) {
        const std::vector<MpPtr> vpMPs = pKFi->GetMapPointMatches();
        for (const MpPtr& pMP : vpMPs)
            if (pMP && !pMP->isBad() && pMP->mBALocalForKF != pKF->mId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mBALocalForKF = pKF->mId;
            }
    }
    for (const MpPtr& pMP : lLocalMapPoints) {
        const auto observations = pMP->GetObservations();
        for (const auto& ob : observations) {
            const KfPtr& pKFi = ob.first;
            if (pKFi->mBALocalForKF != pKF->mId && pKFi->mBAFixedForKF != pKF->mId) {
                pKFi->mBAFixedForKF = pKF->mId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    }
}

}  // namespace ccm_shim
