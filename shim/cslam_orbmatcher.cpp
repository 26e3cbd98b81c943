// Drop-in bodies of the hot entry points of cslam::ORBmatcher (include/cslam/ORBmatcher.h:89-158).
// Replaces in cslam/src/ORBmatcher.cpp: DescriptorDistance (:1653-1669), SearchByBoW (:178-306, :565-698),
// SearchByProjection(Frame&, vector<mpptr>&, th) (:71-148).  The remaining overloads bind the same way.
#include <cslam/ORBmatcher.h>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/MapPoint.h>
#include "ccm_shim.h"

namespace cslam {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    return ccm_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>());
}

static cv::Mat contiguous(const cv::Mat& m) { return m.isContinuous() ? m : m.clone(); }

int ORBmatcher::SearchByBoW(kfptr pKF, Frame& F, std::vector<mpptr>& vpMapPointMatches)
{
    const std::vector<mpptr> mps = pKF->GetMapPointMatches();
    const int n1 = pKF->mDescriptors.rows, n2 = F.N;
    std::vector<int32_t> node1 = ccm_shim::nodes_of(pKF->mFeatVec, n1), node2 = ccm_shim::nodes_of(F.mFeatVec, n2), m12(n1, -1);
    std::vector<uint8_t> valid1(n1);
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) { valid1[i] = mps[i] && !mps[i]->isBad(); a1[i] = pKF->mvKeysUn[i].angle; }
    for (int i = 0; i < n2; i++) a2[i] = F.mvKeys[i].angle;
    const cv::Mat d1 = contiguous(pKF->mDescriptors), d2 = contiguous(F.mDescriptors);
    ccm_bow_options o{mfNNratio, mbCheckOrientation ? 1 : 0, TH_LOW, /*strict_th=*/0};
    const int n = ccm_match_bow(ccm_shim::ctx(), &o, d1.data, node1.data(), valid1.data(), a1.data(), n1,
                                d2.data, node2.data(), nullptr, a2.data(), n2, m12.data());
    if (n < 0) throw estd::infrastructure_ex();
    vpMapPointMatches.assign(n2, mpptr());
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) vpMapPointMatches[m12[i]] = mps[i];
    return n;
}

int ORBmatcher::SearchByBoW(kfptr pKF1, kfptr pKF2, std::vector<mpptr>& vpMatches12)
{
    const std::vector<mpptr> mps1 = pKF1->GetMapPointMatches(), mps2 = pKF2->GetMapPointMatches();
    const int n1 = pKF1->mDescriptors.rows, n2 = pKF2->mDescriptors.rows;
    std::vector<int32_t> node1 = ccm_shim::nodes_of(pKF1->mFeatVec, n1), node2 = ccm_shim::nodes_of(pKF2->mFeatVec, n2), m12(n1, -1);
    std::vector<uint8_t> valid1(n1), valid2(n2);
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) { valid1[i] = mps1[i] && !mps1[i]->isBad(); a1[i] = pKF1->mvKeysUn[i].angle; }
    for (int i = 0; i < n2; i++) { valid2[i] = mps2[i] && !mps2[i]->isBad(); a2[i] = pKF2->mvKeysUn[i].angle; }
    const cv::Mat d1 = contiguous(pKF1->mDescriptors), d2 = contiguous(pKF2->mDescriptors);
    ccm_bow_options o{mfNNratio, mbCheckOrientation ? 1 : 0, TH_LOW, /*strict_th=*/1};      // :629 compares with <, not <=
    const int n = ccm_match_bow(ccm_shim::ctx(), &o, d1.data, node1.data(), valid1.data(), a1.data(), n1,
                                d2.data, node2.data(), valid2.data(), a2.data(), n2, m12.data());
    if (n < 0) throw estd::infrastructure_ex();
    vpMatches12.assign(n1, mpptr());
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) vpMatches12[i] = mps2[m12[i]];
    return n;
}

int ORBmatcher::SearchByProjection(Frame& F, const std::vector<mpptr>& vpMapPoints, const float th)
{
    const int nMP = (int)vpMapPoints.size(), N = F.N;
    std::vector<float> kx(N), ky(N); std::vector<int32_t> koct(N);
    for (int i = 0; i < N; i++) { kx[i] = F.mvKeysUn[i].pt.x; ky[i] = F.mvKeysUn[i].pt.y; koct[i] = F.mvKeysUn[i].octave; }
    const cv::Mat fd = contiguous(F.mDescriptors);
    ccm_frame_grid fg{N, kx.data(), ky.data(), koct.data(), fd.data, F.mnMinX, F.mnMinY, F.mfGridElementWidthInv,
                      F.mfGridElementHeightInv, FRAME_GRID_COLS, FRAME_GRID_ROWS};
    std::vector<uint8_t> in_view(nMP), has_obs(nMP), mp_desc((size_t)nMP * 32), occupied(N);
    std::vector<int32_t> level(nMP), match(N, -1);
    std::vector<float> view_cos(nMP), px(nMP), py(nMP);
    for (int m = 0; m < nMP; m++) {
        const mpptr& pMP = vpMapPoints[m];
        in_view[m] = pMP->mbTrackInView && !pMP->isBad();                       // :79-83
        level[m] = pMP->mnTrackScaleLevel; view_cos[m] = pMP->mTrackViewCos;
        px[m] = pMP->mTrackProjX; py[m] = pMP->mTrackProjY;
        has_obs[m] = pMP->Observations() > 0;
        const cv::Mat d = pMP->GetDescriptor();
        if (!d.empty()) memcpy(&mp_desc[(size_t)m * 32], d.ptr<uint8_t>(), 32);
    }
    for (int i = 0; i < N; i++) occupied[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;   // :112-114
    const int n = ccm_search_by_projection(ccm_shim::ctx(), &fg, F.mvScaleFactors.data(), nMP, in_view.data(), level.data(), view_cos.data(),
                                           px.data(), py.data(), mp_desc.data(), has_obs.data(), occupied.data(), th, mfNNratio, match.data());
    if (n < 0) throw estd::infrastructure_ex();
    for (int i = 0; i < N; i++) if (match[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[match[i]];
    return n;
}

}  // namespace cslam
