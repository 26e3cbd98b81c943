// Drop-in body of cslam::ORBextractor (include/cslam/ORBextractor.h:97-164): constructor tables and operator().
// Replaces cslam/src/ORBextractor.cpp:579-639 and :1216-1278 (with everything they call).
#include <cslam/ORBextractor.h>
#include "ccm_shim.h"

namespace cslam {

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    ccm_orb_params p{_nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST};
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
    static_assert(sizeof(int) == sizeof(int32_t), "mnFeaturesPerLevel / umax are filled as int32");
    if (ccm_orb_tables(&p, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                       reinterpret_cast<int32_t*>(mnFeaturesPerLevel.data()), reinterpret_cast<int32_t*>(umax.data())))
        throw estd::infrastructure_ex();
    mvImagePyramid.resize(nlevels);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*mask: ignored, as in the reference*/,
                              std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
{
    if (_image.empty()) return;                                            // :1219-1220
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    static_assert(sizeof(cv::KeyPoint) == sizeof(ccm_keypoint), "ccm_keypoint mirrors cv::KeyPoint field by field");
    ccm_orb_params p{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    const int cap = nfeatures + 4 * nlevels + 64;                           // the quadtree may overshoot a level's quota by 3
    _keypoints.resize(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int32_t n = 0;
    const int rc = ccm_orb_extract(ccm_shim::ctx(), &p, image.data, image.cols, image.rows, (int)image.step, 0, 1,
                                   reinterpret_cast<ccm_keypoint*>(_keypoints.data()), desc.data, &n, cap);
    if (rc) throw estd::infrastructure_ex();
    _keypoints.resize(n);
    if (n == 0) _descriptors.release();                                     // :1236-1238
    else desc.rowRange(0, n).copyTo(_descriptors);
    // mvImagePyramid is a public member that only the Viewer-less reference never reads after extraction; callers that want
    // a level fetch it on demand:
    //   cv::Mat lv(h_l, w_l, CV_8U); ccm_orb_debug_level(ccm_shim::ctx(), 0, level, lv.data, (int)lv.step);
}

}  // namespace cslam
