// Drop-in bodies of the hot entry points of cslam::ORBmatcher (include/cslam/ORBmatcher.h:89-158).
// Replaces in cslam/src/ORBmatcher.cpp: DescriptorDistance (:1653-1669), SearchByBoW (:178-306, :565-698),
// SearchByProjection(Frame&, vector<mpptr>&, th) (:71-148), Fuse (:854-993, :995-1122).  The remaining overloads bind the same way
// (INTEGRATION.md).
#include <cslam/ORBmatcher.h>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/MapPoint.h>
#include <climits>
#include <set>
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

// ---- Fuse: the projection tests stay here (they walk the map's objects), the selection of the most similar feature in the window
// runs on the GPU (ccm_fuse_select), and the results are applied in map-point order with the reference's state checks repeated at
// application time -- a Replace of an earlier point can make a later one bad or put it into the keyframe, and the reference tests
// exactly that at the top of each iteration (:881-886, :1022-1025); the selection itself reads geometry and descriptors only.
namespace {
struct KfGrid {                                   // a keyframe's features as ccm_frame_grid wants them
    std::vector<float> kx, ky; std::vector<int32_t> oct; cv::Mat desc; ccm_frame_grid g;
    explicit KfGrid(const ORBmatcher::kfptr& pKF)
    {
        const int N = (int)pKF->mvKeysUn.size();
        kx.resize(N); ky.resize(N); oct.resize(N);
        for (int i = 0; i < N; i++) { kx[i] = pKF->mvKeysUn[i].pt.x; ky[i] = pKF->mvKeysUn[i].pt.y; oct[i] = pKF->mvKeysUn[i].octave; }
        desc = pKF->mDescriptors.isContinuous() ? pKF->mDescriptors : pKF->mDescriptors.clone();
        g = ccm_frame_grid{N, kx.data(), ky.data(), oct.data(), desc.data, (float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv,
                           pKF->mfGridElementHeightInv, pKF->mnGridCols, pKF->mnGridRows};
    }
};
// :888-927 / :1029-1066: projection, image bounds, distance range, viewing angle, predicted level
struct FuseQuery { std::vector<uint8_t> valid, desc; std::vector<float> u, v; std::vector<int32_t> level; };
FuseQuery project_for_fuse(const ORBmatcher::kfptr& pKF, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow,
                           const std::vector<ORBmatcher::mpptr>& pts, const std::vector<uint8_t>& candidate, bool invz_via_double)
{
    const int n = (int)pts.size();
    FuseQuery q; q.valid.assign(n, 0); q.desc.assign((size_t)n * 32, 0); q.u.assign(n, 0.f); q.v.assign(n, 0.f); q.level.assign(n, 0);
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    for (int i = 0; i < n; i++) {
        if (!candidate[i]) continue;
        const ORBmatcher::mpptr& pMP = pts[i];
        const cv::Mat p3Dw = pMP->GetWorldPos();
        const cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        // (:899 divides in float, `1/z`; :1041 in double, `1.0/z`, and rounds the quotient to float: kept apart, the two can differ in the last bit)
        const float invz = invz_via_double ? (float)(1.0 / p3Dc.at<float>(2)) : 1 / p3Dc.at<float>(2);
        const float u = fx * (p3Dc.at<float>(0) * invz) + cx, v = fy * (p3Dc.at<float>(1) * invz) + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (PO.dot(pMP->GetNormal()) < 0.5 * dist3D) continue;
        q.level[i] = pMP->PredictScale(dist3D, pKF);
        q.u[i] = u; q.v[i] = v; q.valid[i] = 1;
        const cv::Mat d = pMP->GetDescriptor();
        if (!d.empty()) memcpy(&q.desc[(size_t)i * 32], d.ptr<uint8_t>(), 32);
    }
    return q;
}
}  // namespace

int ORBmatcher::Fuse(kfptr pKF, const std::vector<mpptr>& vpMapPoints, const float th)
{
    const int nMPs = (int)vpMapPoints.size();
    std::vector<uint8_t> candidate(nMPs);
    for (int i = 0; i < nMPs; i++) {
        const mpptr& pMP = vpMapPoints[i];
        candidate[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF) && !pMP->mbDoNotReplace;      // :878-886
    }
    const KfGrid G(pKF);
    const FuseQuery q = project_for_fuse(pKF, pKF->GetRotation(), pKF->GetTranslation(), pKF->GetCameraCenter(), vpMapPoints, candidate, false);
    std::vector<int32_t> best(nMPs, -1), dist(nMPs, 256);
    if (ccm_fuse_select(ccm_shim::ctx(), &G.g, pKF->mvScaleFactors.data(), pKF->mvInvLevelSigma2.data(), nMPs, q.valid.data(), q.u.data(), q.v.data(),
                        q.level.data(), q.desc.data(), th, /*chi2_check=*/1, TH_LOW, best.data(), dist.data()))
        throw estd::infrastructure_ex();
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {                                              // :956-990, in map-point order
        if (best[i] < 0) continue;
        const mpptr& pMP = vpMapPoints[i];
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                     // an earlier Replace changed this point: the reference skips it at :881
        const mpptr pMPinKF = pKF->GetMapPoint(best[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad() && !pMPinKF->mbDoNotReplace) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, best[i]);
            pKF->AddMapPoint(pMP, best[i]);
        }
        nFused++;
    }
    return nFused;
}

int ORBmatcher::Fuse(kfptr pKF, cv::Mat Scw, const std::vector<mpptr>& vpPoints, float th, std::vector<mpptr>& vpReplacePoint)
{
    // Decompose Scw (:1003-1008)
    const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = std::sqrt(sRcw.row(0).dot(sRcw.row(0)));
    const cv::Mat Rcw = sRcw / scw;
    const cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<mpptr> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    std::vector<uint8_t> candidate(nPoints);
    for (int i = 0; i < nPoints; i++) candidate[i] = !vpPoints[i]->isBad() && !spAlreadyFound.count(vpPoints[i]);     // :1022-1025
    const KfGrid G(pKF);
    const FuseQuery q = project_for_fuse(pKF, Rcw, tcw, Ow, vpPoints, candidate, true);
    std::vector<int32_t> best(nPoints, -1), dist(nPoints, 256);
    if (ccm_fuse_select(ccm_shim::ctx(), &G.g, pKF->mvScaleFactors.data(), nullptr, nPoints, q.valid.data(), q.u.data(), q.v.data(), q.level.data(),
                        q.desc.data(), th, /*chi2_check=*/0, TH_LOW, best.data(), dist.data()))
        throw estd::infrastructure_ex();
    int nFused = 0;
    for (int i = 0; i < nPoints; i++) {                                           // :1100-1118
        if (best[i] < 0) continue;
        const mpptr& pMP = vpPoints[i];
        const mpptr pMPinKF = pKF->GetMapPoint(best[i]);
        if (pMPinKF) { if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF; }
        else { pMP->AddObservation(pKF, best[i]); pKF->AddMapPoint(pMP, best[i]); }
        nFused++;
    }
    return nFused;
}

}  // namespace cslam
