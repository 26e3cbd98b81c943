// Drop-in bodies of cslam::ORBmatcher (include/cslam/ORBmatcher.h:89-158): every public entry point.
// Replaces in cslam/src/ORBmatcher.cpp: DescriptorDistance (:1653-1669), SearchByBoW (:178-306, :565-698),
// SearchByProjection x4 (:71-148, :308-446, :1350-1476, :1478-1605), SearchForInitialization (:448-563), SearchForTriangulation
// (:700-852), SearchBySim3 (:1124-1348), Fuse x2 (:854-993, :995-1122).  The protected helpers (CheckDistEpipolarLine,
// RadiusByViewingCos, ComputeThreeMaxima) have no callers left: their arithmetic runs inside the library.
// Pattern of every body: the tests that walk the map's objects (isBad, projection with the object's own float cv::Mat arithmetic,
// distance range, PredictScale) stay here and fill flat arrays; the window search + Hamming selection + order-dependent acceptance is
// one ABI call; the results are applied to the objects in the reference's order.
#include <cslam/ORBmatcher.h>
#include <cslam/Frame.h>
#include <cslam/KeyFrame.h>
#include <cslam/MapPoint.h>
#include <algorithm>
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


// ---- the remaining windowed matchers
namespace {
struct FrGrid {                                   // a Frame's undistorted features as ccm_frame_grid wants them
    std::vector<float> kx, ky, angle; std::vector<int32_t> oct; cv::Mat desc; ccm_frame_grid g;
    explicit FrGrid(const Frame& F)
    {
        const int N = (int)F.mvKeysUn.size();
        kx.resize(N); ky.resize(N); oct.resize(N); angle.resize(N);
        for (int i = 0; i < N; i++) { kx[i] = F.mvKeysUn[i].pt.x; ky[i] = F.mvKeysUn[i].pt.y; oct[i] = F.mvKeysUn[i].octave; angle[i] = F.mvKeysUn[i].angle; }
        desc = F.mDescriptors.isContinuous() ? F.mDescriptors : F.mDescriptors.clone();
        g = ccm_frame_grid{N, kx.data(), ky.data(), oct.data(), desc.data, Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv,
                           Frame::mfGridElementHeightInv, FRAME_GRID_COLS, FRAME_GRID_ROWS};
    }
};
void copy_desc(const ORBmatcher::mpptr& pMP, uint8_t* dst)
{
    const cv::Mat d = pMP->GetDescriptor();
    if (!d.empty()) memcpy(dst, d.ptr<uint8_t>(), 32);
}
}  // namespace

// TrackWithMotionModel's matcher (:1350-1476).  A feature the loop assigns and the rotation histogram then removes ends as nullptr in
// the reference; here a feature without a match keeps what it held.  Such a feature held nullptr or a point without observations before
// the call (any other is skipped, :1419-1421), and Tracking clears the vector in front of both of its calls (src/Tracking.cpp:577, :587).
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const int nLast = LastFrame.N, N = CurrentFrame.N;
    std::vector<uint8_t> valid(nLast, 0), has_obs(nLast, 0), desc((size_t)nLast * 32, 0), occupied(N, 0);
    std::vector<float> u(nLast, 0.f), v(nLast, 0.f), last_angle(nLast, 0.f);
    std::vector<int32_t> last_oct(nLast, 0), match(std::max(N, 1), -1);
    for (int i = 0; i < nLast; i++) {
        const mpptr& pMP = LastFrame.mvpMapPoints[i];
        last_oct[i] = LastFrame.mvKeys[i].octave; last_angle[i] = LastFrame.mvKeysUn[i].angle;
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        const cv::Mat x3Dc = Rcw * pMP->GetWorldPos() + tcw;                       // :1382-1393, float like the reference (1.0/z in double, rounded)
        const float xc = x3Dc.at<float>(0), yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        const float uu = CurrentFrame.fx * xc * invzc + CurrentFrame.cx, vv = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (uu < CurrentFrame.mnMinX || uu > CurrentFrame.mnMaxX || vv < CurrentFrame.mnMinY || vv > CurrentFrame.mnMaxY) continue;
        valid[i] = 1; u[i] = uu; v[i] = vv; has_obs[i] = pMP->Observations() > 0;
        copy_desc(pMP, &desc[(size_t)i * 32]);
    }
    for (int i = 0; i < N; i++) occupied[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;   // :1419-1421
    const FrGrid G(CurrentFrame);
    const int n = ccm_search_by_projection_frame(ccm_shim::ctx(), &G.g, G.angle.data(), CurrentFrame.mvScaleFactors.data(), nLast, valid.data(), u.data(),
                                                 v.data(), last_oct.data(), last_angle.data(), desc.data(), has_obs.data(), occupied.data(), th,
                                                 mbCheckOrientation ? 1 : 0, TH_HIGH, match.data());
    if (n < 0) throw estd::infrastructure_ex();
    for (int i2 = 0; i2 < N; i2++) if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[match[i2]];
    return n;
}

// Relocalisation's matcher (:1478-1605): the same loop over a keyframe's map points; a candidate feature must hold no map point at all
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, kfptr pKF, const std::set<mpptr>& sAlreadyFound, const float th, const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<mpptr> vpMPs = pKF->GetMapPointMatches();
    const int nKF = (int)vpMPs.size(), N = CurrentFrame.N;
    std::vector<uint8_t> valid(nKF, 0), has_obs(nKF, 1), desc((size_t)nKF * 32, 0), occupied(N, 0);
    std::vector<float> u(nKF, 0.f), v(nKF, 0.f), kf_angle(nKF, 0.f);
    std::vector<int32_t> level(nKF, 0), match(std::max(N, 1), -1);
    for (int i = 0; i < nKF; i++) {
        const mpptr& pMP = vpMPs[i];
        kf_angle[i] = pKF->mvKeysUn[i].angle;
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;           // :1499-1502
        const cv::Mat x3Dw = pMP->GetWorldPos();
        const cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0), yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        const float uu = CurrentFrame.fx * xc * invzc + CurrentFrame.cx, vv = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (uu < CurrentFrame.mnMinX || uu > CurrentFrame.mnMaxX || vv < CurrentFrame.mnMinY || vv > CurrentFrame.mnMaxY) continue;
        const cv::Mat PO = x3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
        level[i] = pMP->PredictScale(dist3D, CurrentFrame.shared_from_this());    // :1531
        valid[i] = 1; u[i] = uu; v[i] = vv;
        copy_desc(pMP, &desc[(size_t)i * 32]);
    }
    for (int i = 0; i < N; i++) occupied[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;                 // :1547
    const FrGrid G(CurrentFrame);
    const int n = ccm_search_by_projection_frame(ccm_shim::ctx(), &G.g, G.angle.data(), CurrentFrame.mvScaleFactors.data(), nKF, valid.data(), u.data(),
                                                 v.data(), level.data(), kf_angle.data(), desc.data(), has_obs.data(), occupied.data(), th,
                                                 mbCheckOrientation ? 1 : 0, ORBdist, match.data());
    if (n < 0) throw estd::infrastructure_ex();
    for (int i2 = 0; i2 < N; i2++) if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[match[i2]];
    return n;
}

// Loop closing's matcher (:308-446).  A point the keyframe already observes is re-mapped to its best feature and leaves vpMatched alone
// (:414-436; the reference's bDoNotReplace compares the chosen feature's distance with itself and is never set).
int ORBmatcher::SearchByProjection(kfptr pKF, cv::Mat Scw, const std::vector<mpptr>& vpPoints, std::vector<mpptr>& vpMatched, int th)
{
    const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = std::sqrt(sRcw.row(0).dot(sRcw.row(0)));
    const cv::Mat Rcw = sRcw / scw;
    const cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    const cv::Mat Ow = -Rcw.t() * tcw;
    std::set<mpptr> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(nullptr);
    const int nPoints = (int)vpPoints.size(), N = (int)vpMatched.size();
    std::vector<uint8_t> candidate(nPoints), observed(nPoints, 0), matched(N, 0);
    for (int i = 0; i < nPoints; i++) {
        candidate[i] = !vpPoints[i]->isBad() && !spAlreadyFound.count(vpPoints[i]);                 // :333-334
        if (candidate[i]) observed[i] = vpPoints[i]->GetIndexInKeyFrame(pKF) != -1;
    }
    for (int i = 0; i < N; i++) matched[i] = vpMatched[i] ? 1 : 0;
    const KfGrid G(pKF);
    const FuseQuery q = project_for_fuse(pKF, Rcw, tcw, Ow, vpPoints, candidate, false);            // :336-377: the same tests as Fuse(pKF, Scw, ...), 1/z in float
    std::vector<int32_t> best(std::max(nPoints, 1), -1);
    const int n = ccm_search_by_projection_sim3(ccm_shim::ctx(), &G.g, pKF->mvScaleFactors.data(), nPoints, q.valid.data(), q.u.data(), q.v.data(),
                                                q.level.data(), q.desc.data(), observed.data(), matched.data(), (float)th, best.data());
    if (n < 0) throw estd::infrastructure_ex();
    for (int i = 0; i < nPoints; i++) {
        if (best[i] < 0) continue;
        const mpptr& pMP = vpPoints[i];
        if (observed[i]) pKF->RemapMapPointMatch(pMP, pMP->GetIndexInKeyFrame(pKF), best[i]);
        else vpMatched[best[i]] = pMP;
    }
    return n;
}

// Monocular initialisation (:448-563)
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize)
{
    const int n1 = (int)F1.mvKeysUn.size();
    std::vector<int32_t> oct1(n1), m12(std::max(n1, 1), -1);
    std::vector<float> a1(n1), prev((size_t)2 * n1);
    for (int i = 0; i < n1; i++) { oct1[i] = F1.mvKeysUn[i].octave; a1[i] = F1.mvKeysUn[i].angle; prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    const cv::Mat d1 = contiguous(F1.mDescriptors);
    const FrGrid G2(F2);
    const int n = ccm_search_for_initialization(ccm_shim::ctx(), n1, oct1.data(), d1.data, a1.data(), &G2.g, G2.angle.data(), prev.data(), windowSize,
                                                mfNNratio, mbCheckOrientation ? 1 : 0, m12.data());
    if (n < 0) throw estd::infrastructure_ex();
    vnMatches12.assign(m12.begin(), m12.begin() + n1);
    for (int i = 0; i < n1; i++) vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);       // :555-558 (only matched entries change)
    return n;
}

// Triangulation candidates of LocalMapping::CreateNewMapPoints (:700-852)
int ORBmatcher::SearchForTriangulation(kfptr pKF1, kfptr pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs)
{
    // epipole in the second image (:707-714)
    const cv::Mat C2 = pKF2->GetRotation() * pKF1->GetCameraCenter() + pKF2->GetTranslation();
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx, ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    const int n1 = pKF1->N, n2 = pKF2->N;
    const std::vector<int32_t> node1 = ccm_shim::nodes_of(pKF1->mFeatVec, n1), node2 = ccm_shim::nodes_of(pKF2->mFeatVec, n2);
    std::vector<uint8_t> has1(n1), has2(n2);
    std::vector<float> x1(n1), y1(n1), a1(n1), x2(n2), y2(n2), a2(n2);
    std::vector<int32_t> oct2(n2), m12(std::max(n1, 1), -1);
    for (int i = 0; i < n1; i++) { has1[i] = pKF1->GetMapPoint(i) ? 1 : 0; x1[i] = pKF1->mvKeysUn[i].pt.x; y1[i] = pKF1->mvKeysUn[i].pt.y; a1[i] = pKF1->mvKeysUn[i].angle; }
    for (int i = 0; i < n2; i++) {
        has2[i] = pKF2->GetMapPoint(i) ? 1 : 0; x2[i] = pKF2->mvKeysUn[i].pt.x; y2[i] = pKF2->mvKeysUn[i].pt.y; a2[i] = pKF2->mvKeysUn[i].angle;
        oct2[i] = pKF2->mvKeysUn[i].octave;
    }
    float F[9];
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) F[3 * r + cc] = F12.at<float>(r, cc);
    const cv::Mat d1 = contiguous(pKF1->mDescriptors), d2 = contiguous(pKF2->mDescriptors);
    const int n = ccm_search_for_triangulation(ccm_shim::ctx(), d1.data, node1.data(), has1.data(), x1.data(), y1.data(), a1.data(), n1, d2.data, node2.data(),
                                               has2.data(), x2.data(), y2.data(), a2.data(), oct2.data(), n2, F, ex, ey, pKF2->mvScaleFactors.data(),
                                               pKF2->mvLevelSigma2.data(), mbCheckOrientation ? 1 : 0, m12.data());
    if (n < 0) throw estd::infrastructure_ex();
    vMatchedPairs.clear();
    vMatchedPairs.reserve(n);
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));     // :843-849
    return n;
}

// Loop candidates' Sim3 refinement (:1124-1348): both directions are searched, a match must agree in both
int ORBmatcher::SearchBySim3(kfptr pKF1, kfptr pKF2, std::vector<mpptr>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th)
{
    const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;                          // (the reference uses KF1's intrinsics for both directions, :1127-1130)
    const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    const cv::Mat sR12 = s12 * R12;
    const cv::Mat sR21 = (1.0 / s12) * R12.t();
    const cv::Mat t21 = -sR21 * t12;
    const std::vector<mpptr> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<uint8_t> already1(N1, 0), already2(N2, 0);
    for (int i = 0; i < N1; i++) {                                                                  // :1154-1164
        const mpptr& pMP = vpMatches12[i];
        if (!pMP) continue;
        already1[i] = 1;
        const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
        if (idx2 >= 0 && idx2 < N2) already2[idx2] = 1;
    }
    // one direction: the points of `from` into `to` through (sR, t) after `from`'s own pose
    auto project = [&](const std::vector<mpptr>& pts, const std::vector<uint8_t>& already, const cv::Mat& Rw, const cv::Mat& tw, const cv::Mat& sR,
                       const cv::Mat& t, const kfptr& to, FuseQuery& q) {
        const int n = (int)pts.size();
        q.valid.assign(n, 0); q.desc.assign((size_t)n * 32, 0); q.u.assign(n, 0.f); q.v.assign(n, 0.f); q.level.assign(n, 0);
        for (int i = 0; i < n; i++) {
            const mpptr& pMP = pts[i];
            if (!pMP || already[i] || pMP->isBad()) continue;
            const cv::Mat pc_from = Rw * pMP->GetWorldPos() + tw;
            const cv::Mat pc = sR * pc_from + t;
            if (pc.at<float>(2) < 0.0) continue;
            const float invz = 1.0 / pc.at<float>(2);
            const float x = pc.at<float>(0) * invz, y = pc.at<float>(1) * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!to->IsInImage(u, v)) continue;
            const float dist3D = cv::norm(pc);
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            q.level[i] = pMP->PredictScale(dist3D, to);
            q.u[i] = u; q.v[i] = v; q.valid[i] = 1;
            copy_desc(pMP, &q.desc[(size_t)i * 32]);
        }
    };
    FuseQuery q1, q2;
    project(vpMapPoints1, already1, R1w, t1w, sR21, t21, pKF2, q1);                                // :1170-1208
    project(vpMapPoints2, already2, R2w, t2w, sR12, t12, pKF1, q2);                                // :1250-1288
    const KfGrid G1(pKF1), G2(pKF2);
    std::vector<int32_t> m12(std::max(N1, 1), -1);
    const int n = ccm_search_by_sim3(ccm_shim::ctx(), &G1.g, pKF1->mvScaleFactors.data(), &G2.g, pKF2->mvScaleFactors.data(), q1.valid.data(), q1.u.data(),
                                     q1.v.data(), q1.level.data(), q1.desc.data(), q2.valid.data(), q2.u.data(), q2.v.data(), q2.level.data(), q2.desc.data(),
                                     th, m12.data());
    if (n < 0) throw estd::infrastructure_ex();
    for (int i1 = 0; i1 < N1; i1++) if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];      // :1330-1345
    return n;
}

}  // namespace cslam
