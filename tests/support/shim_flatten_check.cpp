// shim_flatten_check.cpp -- drives shim/flat_graph.h (the map-side logic of the Optimizer drop-ins) with plain stand-in
// keyframe / map point classes.  The stand-ins carry only the members flat_graph.h touches, under the reference's names
// (include/cslam/KeyFrame.h, MapPoint.h); this is a test double for OUR header, not a build of the reference.
// Prints key=value pairs; tests/test_shim_cpu.py checks them against the reference's rules (src/Optimizer.cpp:95-160, :351-406,
// :468-538, :726-785).  Build: g++ -std=c++14 -I include -I shim tests/support/shim_flatten_check.cpp (no GPU, no library: the
// two ccm_pose_* helpers the header calls are defined below).
#include <cmath>
#include <cstdio>
#include <memory>
#include <utility>
#include "flat_graph.h"

extern "C" int ccm_pose_from_mat4f(const float* T, double* pose)
{
    // identity rotation is all the harness uses: keep the translation, unit quaternion
    pose[0] = pose[1] = pose[2] = 0.0; pose[3] = 1.0;
    for (int i = 0; i < 3; i++) pose[4 + i] = T[4 * i + 3];
    return 0;
}

typedef std::pair<size_t, size_t> idpair;
struct MP;
struct KF {
    size_t mUniqueId; idpair mId; bool bad = false;
    idpair mBALocalForKF{9999, 9999}, mBAFixedForKF{9999, 9999};
    float tx = 0;
    std::vector<std::shared_ptr<KF>> covis;
    std::vector<std::shared_ptr<MP>> matches;            // index = keypoint
    bool isBad() const { return bad; }
    std::vector<std::shared_ptr<KF>> GetVectorCovisibleKeyFrames() { return covis; }
    std::vector<std::shared_ptr<MP>> GetMapPointMatches() { return matches; }
};
struct MP {
    size_t mUniqueId; idpair mId; bool bad = false;
    idpair mBALocalForKF{9999, 9999};
    std::map<std::shared_ptr<KF>, size_t> obs;
    bool isBad() const { return bad; }
    std::map<std::shared_ptr<KF>, size_t> GetObservations() { return obs; }
};
typedef std::shared_ptr<KF> kfptr;
typedef std::shared_ptr<MP> mpptr;
struct Access {
    static void pose(const kfptr& k, float T[16]) { for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.f : 0.f; T[3] = k->tx; }
    static void intrinsics(const kfptr&, double k[4]) { k[0] = k[1] = 458.0; k[2] = 367.0; k[3] = 248.0; }
    static void keypoint(const kfptr& k, size_t idx, double xy[2], double* inv_sigma2) { xy[0] = 10.0 * k->mUniqueId + idx; xy[1] = 5.0; *inv_sigma2 = 1.0 / (1.0 + idx % 3); }
    static void world_pos(const mpptr& p, float X[3]) { X[0] = (float)p->mUniqueId; X[1] = 0; X[2] = 4; }
};
typedef ccm_shim::FlatGraph<kfptr, mpptr, Access> Flat;

static kfptr mkkf(size_t uid, size_t id, size_t client) { auto k = std::make_shared<KF>(); k->mUniqueId = uid; k->mId = idpair(id, client); k->tx = (float)uid; return k; }
static mpptr mkmp(size_t uid) { auto p = std::make_shared<MP>(); p->mUniqueId = uid; p->mId = idpair(uid, 0); return p; }
static void observe(const kfptr& k, const mpptr& p) { p->obs[k] = k->matches.size(); k->matches.push_back(p); }

int main()
{
    // Map: keyframes 0..5 (5 is bad), 6 exists but is NOT handed to the optimiser (another map).
    std::vector<kfptr> kf;
    for (size_t i = 0; i < 7; i++) kf.push_back(mkkf(100 + i, i, 0));
    kf[5]->bad = true;
    // points: p0 seen by 0,1,2 | p1 seen by 1 only | p2 seen by 2 and the bad 5 | p3 seen by 3 and the foreign 6 | p4 seen by nobody usable (5, 6)
    // p5 bad | p6 seen by 0,1,2,3,4
    std::vector<mpptr> mp;
    for (size_t i = 0; i < 7; i++) mp.push_back(mkmp(500 + i));
    observe(kf[0], mp[0]); observe(kf[1], mp[0]); observe(kf[2], mp[0]);
    observe(kf[1], mp[1]);
    observe(kf[2], mp[2]); observe(kf[5], mp[2]);
    observe(kf[3], mp[3]); observe(kf[6], mp[3]);
    observe(kf[5], mp[4]); observe(kf[6], mp[4]);
    mp[5]->bad = true; observe(kf[0], mp[5]); observe(kf[1], mp[5]);
    for (int i = 0; i < 5; i++) observe(kf[i], mp[6]);

    for (int min_obs = 0; min_obs <= 2; min_obs++) {
        Flat g;
        for (size_t i = 0; i < 6; i++) if (!kf[i]->isBad()) g.add_keyframe(kf[i], i == 0);
        int included = 0;
        for (const mpptr& p : mp) if (!p->isBad()) included += g.add_map_point(p, min_obs) ? 1 : 0;
        ccm_ba_problem pb = g.problem();
        bool consistent = pb.n_poses == (int)g.kfs.size() && pb.n_points == included && pb.n_edges == (int)g.edge_kf.size() && g.obs.size() == 2 * g.info.size();
        for (size_t e = 0; e < g.edge_pose.size(); e++)
            consistent = consistent && g.kfs[g.edge_pose[e]] == g.edge_kf[e] && g.edge_point[e] >= 0 && g.edge_point[e] < pb.n_points && !g.edge_kf[e]->isBad();
        printf("min_obs%d_poses=%d min_obs%d_points=%d min_obs%d_edges=%d min_obs%d_fixed=%d min_obs%d_consistent=%d\n", min_obs, pb.n_poses, min_obs, pb.n_points,
               min_obs, pb.n_edges, min_obs, (int)g.fixed[0] + 2 * (int)g.fixed[1], min_obs, consistent ? 1 : 0);
        if (min_obs == 1) printf("tx_of_row3=%g first_obs_x=%g\n", g.poses[7 * 3 + 4], g.obs[0]);
    }

    // local BA: current keyframe 2; covisible 1, 3 and the bad 5; local points = those seen in 2, 1, 3; fixed = other observers
    kf[2]->covis = {kf[1], kf[3], kf[5]};
    std::list<kfptr> lLocal, lFixed; std::list<mpptr> lPts;
    ccm_shim::gather_local_ba(kf[2], lLocal, lPts, lFixed);
    int fixed_has0 = 0, fixed_has4 = 0, fixed_has6 = 0, fixed_has5 = 0;
    for (const kfptr& k : lFixed) { fixed_has0 += k == kf[0]; fixed_has4 += k == kf[4]; fixed_has6 += k == kf[6]; fixed_has5 += k == kf[5]; }
    printf("local_kfs=%d local_first_is_current=%d local_points=%d fixed_kfs=%d fixed_has0=%d fixed_has4=%d fixed_has6=%d fixed_has5=%d\n", (int)lLocal.size(),
           lLocal.front() == kf[2] ? 1 : 0, (int)lPts.size(), (int)lFixed.size(), fixed_has0, fixed_has4, fixed_has6, fixed_has5);
    printf("bad_neighbour_marked_local=%d bad_observer_marked=%d point_marks=%d\n", kf[5]->mBALocalForKF == kf[2]->mId ? 1 : 0,
           (kf[5]->mBALocalForKF == kf[2]->mId || kf[5]->mBAFixedForKF == kf[2]->mId) ? 1 : 0, mp[0]->mBALocalForKF == kf[2]->mId ? 1 : 0);
    // flatten it the way LocalBundleAdjustmentClient does: every local point is a vertex (min_obs 0), edges to local + fixed keyframes
    Flat g;
    for (const kfptr& k : lLocal) g.add_keyframe(k, k->mId.first == 0 && k->mId.second == 0);
    const size_t n_local = g.kfs.size();
    for (const kfptr& k : lFixed) g.add_keyframe(k, true);
    for (const mpptr& p : lPts) g.add_map_point(p, 0);
    int fixed_flags = 0;
    for (size_t r = n_local; r < g.kfs.size(); r++) fixed_flags += g.fixed[r];
    printf("lba_poses=%d lba_points=%d lba_edges=%d lba_fixed_flags=%d lba_local_rows=%d\n", (int)g.kfs.size(), (int)g.mps.size(), (int)g.edge_pose.size(), fixed_flags, (int)n_local);
    return 0;
}
