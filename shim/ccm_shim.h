// ccm_shim.h -- shared by the three drop-in translation units of this directory.
//
// These files REPLACE cslam/src/ORBextractor.cpp, the hot entry points of cslam/src/ORBmatcher.cpp and the bundle-adjustment
// entry points of src/Optimizer.cpp in a CCM-SLAM checkout: same classes, same signatures (the reference's own headers
// include/cslam/{ORBextractor,ORBmatcher,Optimizer}.h stay untouched), bodies forward to the C ABI of libccm_hot.so
// (include/ccm_hot.h).  They need the reference's headers and its dependencies (OpenCV, Boost, ROS messages), so they are
// compiled by shim/CMakeLists.txt only where find_package(OpenCV) succeeds and CCM_SLAM_INCLUDE_DIR points at a checkout;
// in this repository tests/test_shim_cpu.py checks every ccm_* call in them against the prototypes of ccm_hot.h.
#pragma once
#include <ccm_hot.h>
#include <cstdint>
#include <vector>

namespace ccm_shim {

// One context (HIP stream + workspaces) per calling thread: Tracking, LocalMapping, LoopFinder and MapMatcher each run in
// their own thread in the reference (src/ClientHandler.cpp:140-176), and a ccm_ctx is not meant to be shared.
inline ccm_ctx* ctx()
{
    // a shim object built against another version of ccm_hot.h than the library would pass structures of the wrong size
    static thread_local ccm_ctx* c = ccm_abi_version() == CCM_ABI_VERSION ? ccm_create(0, 0) : nullptr;
    return c;                                          // nullptr: every ccm_* call returns CCM_E_ARG and the shim bodies throw
}

// per-feature vocabulary node of a DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>), -1 = none
template <class FeatVec>
inline std::vector<int32_t> nodes_of(const FeatVec& fv, int n)
{
    std::vector<int32_t> node(n, -1);
    for (const auto& kv : fv)
        for (unsigned i : kv.second) if ((int)i < n) node[i] = (int32_t)kv.first;
    return node;
}

}  // namespace ccm_shim
