/* ccm_hot.h -- C ABI of the MI355X hot path for CCM-SLAM
 * (ORB extraction, Hamming matching, reprojection bundle adjustment).
 *
 * This is the drop-in boundary: plain pointers and sizes,
 * caller-allocated HOST buffers unless a name says `_dev`, int return 0 = OK or a
 * negative CCM_E_* code, nothing throws.  One ccm_ctx per calling thread: it
 * owns a HIP stream, device workspaces and (optionally) an RCCL communicator,
 * so concurrent callers never share device state.
 *
 * All file:line citations are relative to the reference tree
 * (taiyaki-go/motioncheck_ccm_slam).  INTEGRATION.md shows the C++ shim a
 * maintainer adds on the reference side to forward the original classes here.
 */
#ifndef CCM_HOT_H
#define CCM_HOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct, an enum count or a prototype changes (3: ccm_ba_result.pcg_pipelined; round 2 had already changed
 * ccm_essential_graph, CCM_PROF_COUNT and removed ccm_comm_init_shm under version 1).  A caller compiled against another version
 * must not call further: ccm_abi_version() returns the library's value, compare it with this macro (the Python mirror and
 * shim/ccm_shim.h do). */
#define CCM_ABI_VERSION 3

enum {
    CCM_OK = 0,
    CCM_E_ARG = -1,       /* bad argument (null pointer, size <= 0, unsupported parameter) */
    CCM_E_DEVICE = -2,    /* HIP runtime error; ccm_last_error() has the text */
    CCM_E_NOMEM = -3,     /* host or device allocation failed */
    CCM_E_CAPACITY = -4,  /* caller buffer too small (max_per_image, topk ...) */
    CCM_E_NUMERIC = -5,   /* reduced system not positive definite in every LM trial */
    CCM_E_COMM = -6,      /* RCCL error */
    CCM_E_STATE = -7      /* call order violated (e.g. debug fetch before extract) */
};

typedef struct ccm_ctx ccm_ctx;

/* ------------------------------------------------------------------ context */
int ccm_abi_version(void);
/* device = HIP device ordinal.  flags: reserved, pass 0. */
ccm_ctx* ccm_create(int device, int flags);
void ccm_destroy(ccm_ctx*);
const char* ccm_last_error(const ccm_ctx*);
/* Blocks until everything queued on the context's stream has finished. */
int ccm_sync(ccm_ctx*);
/* The hipStream_t the context launches on (for callers that time with HIP events). */
void* ccm_stream(ccm_ctx*);

/* Per-kernel timing with HIP events on the context's stream (bench.py's roofline figures).
 * While enabled, every launch group is bracketed by two events.  ccm_profile_read synchronises,
 * returns for each label the summed milliseconds and the number of launches since the last read,
 * and resets the sums.  Labels: see CCM_PROF_*. */
enum { CCM_PROF_RESIZE = 0, CCM_PROF_FAST_SCORE, CCM_PROF_CELL_NMS, CCM_PROF_OCTREE, CCM_PROF_ORIENT_DESC,
       CCM_PROF_HAMMING_BF,
       /* bundle adjustment: linearisation (k_ba_lin_landmark + k_ba_lin_pose), landmark inverse and Y = Hpl Dinv
        * (k_sp_dinv + k_sp_edge_y), the Schur block GEMM (k_sp_schur_blocks), bschur, back-substitution */
       CCM_PROF_BA_LINEARIZE, CCM_PROF_BA_DINV_Y, CCM_PROF_BA_SCHUR_BLOCKS, CCM_PROF_BA_BSCHUR, CCM_PROF_BA_BACKSUB,
       CCM_PROF_COUNT };
/* Page-lock a host buffer the caller keeps (a frame pool, the cv::Mat a camera driver fills): uploads from it and downloads
 * into it then go by DMA at the full PCIe rate instead of through the runtime's staging of pageable memory.  Thin wrappers over
 * hipHostRegister / hipHostUnregister; optional -- every entry point accepts pageable pointers. */
int ccm_host_register(ccm_ctx*, void* ptr, size_t bytes);
int ccm_host_unregister(ccm_ctx*, void* ptr);

int ccm_profile_enable(ccm_ctx*, int on);
int ccm_profile_read(ccm_ctx*, float ms[CCM_PROF_COUNT], int32_t launches[CCM_PROF_COUNT]);

/* ---------------------------------------------------------------- extractor
 * Replaces ORBextractor::ORBextractor (cslam/src/ORBextractor.cpp:579-639) and
 * ORBextractor::operator() (:1216-1278) with its callees ComputePyramid
 * (:1280-1304), ComputeKeyPointsOctTree (:933-1024), DistributeOctTree
 * (:707-931), IC_Angle (:68-95), GaussianBlur 7x7 sigma 2 (:1259) and
 * computeOrbDescriptor (:100-316).                                           */
typedef struct {
    int   nfeatures;     /* ORBextractor.nFeatures   (cslam/conf/config.yaml:38-51) */
    float scale_factor;  /* ORBextractor.scaleFactor */
    int   nlevels;       /* ORBextractor.nLevels, 1..CCM_MAX_LEVELS */
    int   ini_th_fast;   /* ORBextractor.iniThFAST */
    int   min_th_fast;   /* ORBextractor.minThFAST */
} ccm_orb_params;

#define CCM_MAX_LEVELS 16

/* Layout-identical to cv::KeyPoint (pt.x, pt.y, size, angle, response, octave,
 * class_id): a std::vector<cv::KeyPoint>'s storage can be passed directly. */
typedef struct {
    float x, y;
    float size;
    float angle;     /* degrees [0,360) */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id; /* always -1 */
} ccm_keypoint;

/* Constructor tables (getters GetScaleFactors / GetInverseScaleFactors /
 * GetScaleSigmaSquares / GetInverseScaleSigmaSquares, include/cslam/ORBextractor.h:120-150).
 * Each output may be NULL; arrays hold nlevels entries (umax: 16). Host only, no GPU. */
int ccm_orb_tables(const ccm_orb_params*, float* scale, float* inv_scale, float* sigma2,
                   float* inv_sigma2, int32_t* features_per_level, int32_t* umax);
/* Pyramid level sizes for a w x h input (ORBextractor.cpp:1284-1285). Host only. */
int ccm_orb_level_sizes(const ccm_orb_params*, int w, int h, int32_t* level_w, int32_t* level_h);

/* Batched operator(): n_images 8-bit images of w x h (row stride `stride` bytes,
 * image i starts at img + i*image_stride) -> per image up to max_per_image
 * keypoints and 32-byte descriptors, rows in the reference's order (level-major,
 * then DistributeOctTree list order).  kps: [n_images][max_per_image],
 * desc: [n_images][max_per_image][32], counts: [n_images].
 * n_images == 0 or w*h == 0 -> CCM_OK with nothing written (the reference
 * returns silently on an empty image, ORBextractor.cpp:1219-1220).
 * Returns CCM_E_CAPACITY if an image yields more than max_per_image keypoints
 * (never happens with max_per_image >= nfeatures). */
int ccm_orb_extract(ccm_ctx*, const ccm_orb_params*, const uint8_t* img, int w, int h, int stride,
                    size_t image_stride, int n_images, ccm_keypoint* kps, uint8_t* desc,
                    int32_t* counts, int max_per_image);

/* Device-resident variant: img_dev is a device pointer; results stay on the
 * device (fetch with ccm_orb_fetch) so that ccm_hamming_match_dev can consume
 * the descriptors without a PCIe round trip.  The call is asynchronous on the
 * context's stream.  */
int ccm_orb_extract_dev(ccm_ctx*, const ccm_orb_params*, const uint8_t* img_dev, int w, int h,
                        int stride, size_t image_stride, int n_images, int max_per_image);
/* Copy the last ccm_orb_extract_dev results to host (any pointer may be NULL). */
int ccm_orb_fetch(ccm_ctx*, ccm_keypoint* kps, uint8_t* desc, int32_t* counts);
/* Device pointers of the last extract: desc_dev [n_images][max_per_image][32],
 * counts_dev [n_images] (int32).  Valid until the next extract on this ctx. */
int ccm_orb_result_dev(ccm_ctx*, const uint8_t** desc_dev, const int32_t** counts_dev,
                       int* max_per_image);

/* Test/debug taps of the last extract (host copies; synchronise internally):
 * pyramid level pixels (mvImagePyramid, include/cslam/ORBextractor.h:152) ... */
int ccm_orb_debug_level(ccm_ctx*, int image, int level, uint8_t* out, int out_stride);
/* ... and the per-level FAST candidates before DistributeOctTree, in the order
 * vToDistributeKeys is filled (ORBextractor.cpp:957-998): xy[2*i], xy[2*i+1]
 * relative to minBorderX/Y as there, score[i].  Returns the count (>= 0) or an error. */
int ccm_orb_debug_candidates(ccm_ctx*, int image, int level, int32_t* xy, int32_t* score, int max);

/* ------------------------------------------------------------------ matcher
 * ORBmatcher::DescriptorDistance (cslam/src/ORBmatcher.cpp:1653-1669): host helper. */
int ccm_descriptor_distance(const uint8_t* a, const uint8_t* b);

/* Brute-force best / second-best Hamming search, n_pairs independent problems
 * (the inner loop of SearchByBoW, ORBmatcher.cpp:224-245, over one vocabulary
 * node holding every feature).  q: [n_pairs][nq][32], t: [n_pairs][nt][32].
 * Optional per-pair live counts nq_n/nt_n (NULL = all nq/nt rows live).
 * Outputs per query: best_idx (lowest index wins ties, -1 if nt == 0),
 * best_dist and second_dist (256 when absent), exactly the running
 * (bestDist1,bestIdx,bestDist2) of the reference with strict `<`. */
int ccm_hamming_match(ccm_ctx*, const uint8_t* q, int nq, const uint8_t* t, int nt, int n_pairs,
                      const int32_t* nq_n, const int32_t* nt_n,
                      int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* Same on device pointers (descriptor strides in rows), asynchronous. */
int ccm_hamming_match_dev(ccm_ctx*, const uint8_t* q_dev, int nq, size_t q_pair_stride,
                          const uint8_t* t_dev, int nt, size_t t_pair_stride, int n_pairs,
                          const int32_t* nq_n_dev, const int32_t* nt_n_dev,
                          int32_t* best_idx_dev, int32_t* best_dist_dev, int32_t* second_dist_dev);

/* Acceptance test applied by the callers (ORBmatcher.cpp:247-249): host helper.
 * strict = 0 -> best <= th (Frame variant), 1 -> best < th (KF-KF variant, :641). */
int ccm_ratio_test(int best_dist, int second_dist, float nnratio, int th, int strict);

/* ORBmatcher::SearchByBoW, both overloads (ORBmatcher.cpp:178-306, 565-698), for one pair.
 * Side 1 = the keyframe whose features drive the outer loop, side 2 = the
 * frame/keyframe searched.  node1/node2: vocabulary node id of every feature
 * (the FeatureVector, as one id per feature; features with the same id are
 * visited in ascending feature index, as DBoW2 fills them).  valid1: feature
 * has a good MapPoint (pMP && !isBad).  valid2: NULL for the Frame overload
 * (every frame feature is a candidate), else the KF-KF validity mask.
 * angle1/angle2: keypoint angles (degrees) for the rotation histogram.
 * Output match12[n1]: index into side 2 or -1 -- the KF-KF overload's
 * vpMatches12; for the Frame overload the reference fills
 * vpMapPointMatches[idx2] = MP(idx1), the shim inverts it.  Returns the number
 * of matches (>= 0) or an error. */
typedef struct {
    float nnratio;        /* mfNNratio */
    int   check_ori;      /* mbCheckOrientation */
    int   th;             /* TH_LOW = 50 */
    int   strict_th;      /* 0: best <= th (Frame overload), 1: best < th (KF-KF) */
} ccm_bow_options;
int ccm_match_bow(ccm_ctx*, const ccm_bow_options*,
                  const uint8_t* desc1, const int32_t* node1, const uint8_t* valid1,
                  const float* angle1, int n1,
                  const uint8_t* desc2, const int32_t* node2, const uint8_t* valid2,
                  const float* angle2, int n2, int32_t* match12);

/* Windowed matchers.  A frame's undistorted keypoints with its 75x48 feature grid
 * (Frame::AssignFeaturesToGrid, src/Frame.cpp:103-118; FRAME_GRID_COLS/ROWS include/cslam/Frame.h:51-52). */
typedef struct {
    int n;                         /* Frame::N */
    const float* kp_x; const float* kp_y; const int32_t* kp_octave;   /* mvKeysUn */
    const uint8_t* desc;           /* mDescriptors [n][32] */
    float min_x, min_y;            /* mnMinX, mnMinY */
    float inv_w, inv_h;            /* mfGridElementWidthInv, mfGridElementHeightInv */
    int grid_cols, grid_rows;      /* 75, 48 */
} ccm_frame_grid;
/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cpp:200-253) for nq queries at once, with the
 * Hamming distance of each returned feature to the query's descriptor.  Per query up to `cap` entries, in
 * the order the reference returns them; cand_n[q] is the true count (CCM_E_CAPACITY if any exceeds cap).
 * r < 0 skips a query. */
int ccm_window_candidates(ccm_ctx*, const ccm_frame_grid*, int nq, const float* qx, const float* qy, const float* qr,
                          const int32_t* min_level, const int32_t* max_level, const uint8_t* qdesc, int cap,
                          int32_t* cand_idx, int32_t* cand_dist, int32_t* cand_n);
/* ORBmatcher::SearchByProjection(Frame&, const vector<mpptr>&, th) (cslam/src/ORBmatcher.cpp:71-148), the matcher
 * of TrackLocalMap.  Per map point: in_view = mbTrackInView && !isBad, level = mnTrackScaleLevel, view_cos =
 * mTrackViewCos, proj = mTrackProjX/Y, its descriptor, has_obs = Observations() > 0.  occupied[i] (in/out): feature i
 * already holds a map point with observations.  match[i] = index of the map point newly assigned to feature i or
 * -1.  Returns nmatches. */
int ccm_search_by_projection(ccm_ctx*, const ccm_frame_grid*, const float* scale_factors, int n_mp, const uint8_t* in_view,
                             const int32_t* level, const float* view_cos, const float* proj_x, const float* proj_y,
                             const uint8_t* mp_desc, const uint8_t* mp_has_obs, uint8_t* occupied, float th, float nnratio,
                             int32_t* match);

/* ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th) (ORBmatcher.cpp:1350-1476), the matcher of
 * TrackWithMotionModel.  Per last-frame feature: valid = has a map point, not an outlier, and its projection (u,v)
 * into the current frame (computed by the caller in float as :1382-1393) has positive depth and lies inside the
 * frame bounds; last_octave, last_angle = LastFrame.mvKeys[i].octave / mvKeysUn[i].angle.  match[i2] = last-frame
 * feature whose map point is assigned to current feature i2, or -1.  Returns nmatches.  orb_dist = TH_HIGH (100) here.
 * The relocalisation overload SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1478-1605) is the same
 * loop over the keyframe's map points: valid = not bad, not in sAlreadyFound and projected inside the frame and the
 * distance range (:1502-1531), last_octave = nPredictedLevel, last_angle = pKF->mvKeysUn[i].angle, mp_has_obs all 1
 * (any assigned feature is skipped, :1547), occupied = mvpMapPoints[i2] != null, orb_dist = ORBdist. */
int ccm_search_by_projection_frame(ccm_ctx*, const ccm_frame_grid* current, const float* cur_angle, const float* scale_factors,
                                   int n_last, const uint8_t* valid, const float* u, const float* v, const int32_t* last_octave,
                                   const float* last_angle, const uint8_t* mp_desc, const uint8_t* mp_has_obs, uint8_t* occupied,
                                   float th, int check_ori, int orb_dist, int32_t* match);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cpp:448-563).
 * prev_matched_xy [n1][2] is vbPrevMatched (updated for matched features as :556-558); matches12[i1] = index into
 * F2 or -1.  Returns nmatches. */
int ccm_search_for_initialization(ccm_ctx*, int n1, const int32_t* oct1, const uint8_t* desc1, const float* angle1,
                                  const ccm_frame_grid* f2, const float* angle2, float* prev_matched_xy, int window,
                                  float nnratio, int check_ori, int32_t* matches12);

/* Selection loop of ORBmatcher::Fuse, both overloads (ORBmatcher.cpp:914-955 with the chi2 gate, :1072-1100 without):
 * per map point that passed the caller's geometric checks (valid, projection u,v, predicted level) the most similar
 * keyframe feature in the window th * scaleFactor[level] at level-1..level.  best_idx[m] = feature or -1 (bestDist
 * > accept_th = TH_LOW); the caller applies Replace / AddObservation in map-point order as :958-990 / :1103-1118. */
int ccm_fuse_select(ccm_ctx*, const ccm_frame_grid* kf, const float* scale_factors, const float* inv_level_sigma2, int n_mp,
                    const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc, float th,
                    int chi2_check, int accept_th, int32_t* best_idx, int32_t* best_dist);

/* ccm_fuse_select for n_kf keyframes in ONE launch -- the server's fuse loops call Fuse once per keyframe (src/Mapping.cpp:515-546:
 * the current keyframe's map points into every neighbour; src/MapMerger.cpp:576-586: the loop map points into every corrected
 * keyframe).  Map points projected into keyframe k are rows mp_first[k] .. mp_first[k+1]-1 (mp_first[0] = 0) of valid / u / v /
 * level / mp_desc / best_idx / best_dist; best_idx is an index into keyframe k's own features.  Returns what n_kf sequential
 * ccm_fuse_select calls return: the selection reads projections, descriptors and features only; a point that an earlier
 * keyframe's Replace has made bad is skipped by the caller when it applies the results in keyframe order (ORBmatcher.cpp:884-886). */
int ccm_fuse_select_batch(ccm_ctx*, int n_kf, const ccm_frame_grid* kfs, const float* scale_factors, const float* inv_level_sigma2,
                          const int32_t* mp_first, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                          const uint8_t* mp_desc, float th, int chi2_check, int accept_th, int32_t* best_idx, int32_t* best_dist);

/* ORBmatcher::SearchBySim3 (ORBmatcher.cpp:1124-1348).  The caller projects every map point of KF1 into KF2 with the
 * Sim3 (valid1/u1/v1/level1 per feature of KF1, :1170-1208; mp_desc1 = GetDescriptor()) and vice versa; both
 * directions select the most similar feature (<= TH_HIGH) and match12[i1] = i2 where they agree (:1330-1345), else -1.
 * Returns nFound. */
int ccm_search_by_sim3(ccm_ctx*, const ccm_frame_grid* kf1, const float* scale_factors1, const ccm_frame_grid* kf2, const float* scale_factors2,
                       const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* mp_desc1,
                       const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* mp_desc2,
                       float th, int32_t* match12);

/* ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cpp:308-446), loop closing.  valid/u/v/
 * level = the caller's projection checks (:333-377); matched[idx] (in/out) = vpMatched[idx] != null; observed[m] = the
 * point already is an observation of pKF (GetIndexInKeyFrame != -1, :414: the caller re-maps it, vpMatched is not
 * touched).  best_idx[m] = chosen feature or -1.  Returns nmatches. */
int ccm_search_by_projection_sim3(ccm_ctx*, const ccm_frame_grid* kf, const float* scale_factors, int n_mp, const uint8_t* valid,
                                  const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc, const uint8_t* observed,
                                  uint8_t* matched, float th, int32_t* best_idx);
/* ccm_search_by_projection_sim3 for n_kf keyframes in ONE launch per kernel -- the loop closer projects the loop's map points into
 * every keyframe connected to the current one (src/LoopFinder.cpp, src/MapMatcher.cpp: one SearchByProjection(pKF, Scw, vpPoints,
 * vpMatched, th) per keyframe).  Keyframe k's map points are rows mp_first[k] .. mp_first[k+1]-1 (mp_first[0] = 0) of valid / u / v /
 * level / mp_desc / observed / best_idx; its vpMatched flags are bytes F_k .. F_k + kfs[k].n - 1 of `matched` with F_k = kfs[0].n + ...
 * + kfs[k-1].n (in/out); best_idx is an index into keyframe k's own features; n_matches[k] = what the single call returns for
 * keyframe k.  Returns the sum, i.e. exactly what n_kf sequential calls return and write. */
int ccm_search_by_projection_sim3_batch(ccm_ctx*, int n_kf, const ccm_frame_grid* kfs, const float* scale_factors, const int32_t* mp_first,
                                        const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc,
                                        const uint8_t* observed, uint8_t* matched, float th, int32_t* best_idx, int32_t* n_matches);

/* ORBmatcher::SearchForTriangulation (ORBmatcher.cpp:700-852): per feature of KF1 without a map point, the most
 * similar (<= TH_LOW, last one among equals) feature of KF2 in the same vocabulary node that has no map point, is
 * not near the epipole (ex, ey) (:775-777) and fulfils the epipolar constraint of F12 (row-major 3x3 float,
 * CheckDistEpipolarLine :159-176); rotation-histogram filter when check_ori.  match12[i1] = i2 or -1 (the caller
 * lists the pairs in i1 order, :843-849).  Returns nmatches. */
int ccm_search_for_triangulation(ccm_ctx*, const uint8_t* desc1, const int32_t* node1, const uint8_t* has_mp1, const float* x1, const float* y1,
                                 const float* angle1, int n1, const uint8_t* desc2, const int32_t* node2, const uint8_t* has_mp2,
                                 const float* x2, const float* y2, const float* angle2, const int32_t* octave2, int n2, const float* F12,
                                 float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int check_ori, int32_t* match12);

/* ---------------------------------------------------------------- vocabulary tree
 * DBoW2::TemplatedVocabulary (cslam/thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h).  A vocabulary is handed over as
 * the arrays loadFromTextFile (:1338-1423) fills: node 0 = root, parent[i] < i, descriptors [n_nodes][32], weights
 * [n_nodes]; children keep node-id order, a node without children is a word, words are numbered in node order.
 * The tree lives on the context's device until ccm_voc_destroy. */
typedef struct ccm_vocabulary ccm_vocabulary;
int  ccm_voc_create(ccm_ctx*, int k, int L, int n_nodes, const int32_t* parent, const uint8_t* descriptors, const double* weights,
                    ccm_vocabulary** out);
void ccm_voc_destroy(ccm_vocabulary*);
int  ccm_voc_words(const ccm_vocabulary*);

/* transform(feature, word_id, weight, nid, levelsup) (:1217-1258) for n features: the L-level k-way Hamming descent.
 * node_id = the node at level L - levelsup (0 = root when that is <= 0 or the branch ends above it).  The _dev form
 * takes descriptors resident on the device (ccm_orb_result_dev); results are host arrays. */
int ccm_voc_transform(ccm_vocabulary*, const uint8_t* features, int n, int levelsup, int32_t* word_id, double* weight, int32_t* node_id);
int ccm_voc_transform_dev(ccm_vocabulary*, const uint8_t* features_dev, int n, int levelsup, int32_t* word_id, double* weight,
                          int32_t* node_id);

/* transform(features, BowVector&, FeatureVector&, levelsup) (:1125-1193) from the per-feature results: out_id/out_val
 * (room for n entries) = the BowVector in key order after the weighting (0 TF_IDF, 1 TF, 2 IDF, 3 BINARY) and the
 * normalisation the scoring type asks for (0 L1_NORM ... 5 DOT_PRODUCT, BowVector.h:36-53); fv_node[i] = FeatureVector
 * node of feature i, -1 when its word is stopped -- the per-feature form ccm_match_bow takes.  Returns the size. */
int ccm_bow_vector(int n, const int32_t* word_id, const double* weight, const int32_t* node_id, int weighting, int scoring,
                   int32_t* out_id, double* out_val, int32_t* fv_node);
/* L1Scoring::score (ScoringObject.cpp:23-68) of two BowVectors in key order */
double ccm_bow_score_l1(int n1, const int32_t* id1, const double* v1, int n2, const int32_t* id2, const double* v2);

/* MapPoint::ComputeDistinctiveDescriptors (cslam/src/MapPoint.cpp:929-994) for n_points map points: the observed
 * descriptors of point p are desc[first[p] .. first[p]+count[p]); best[p] = index (within the point) of the descriptor
 * with the least median distance to the others, first among equals; -1 for count 0. */
int ccm_distinctive_descriptors(ccm_vocabulary*, const uint8_t* desc, const int64_t* first, const int32_t* count, int n_points,
                                int32_t* best);

/* ---------------------------------------------------------------- optimizer
 * The 6-DoF pose / 3-DoF point reprojection BA that Optimizer::BundleAdjustmentClient
 * (src/Optimizer.cpp:32-212), LocalBundleAdjustmentClient (:349-644) and
 * MapFusionGBA (:646-865) hand to g2o: EdgeSE3ProjectXYZ residual/Jacobian
 * (thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:103-147), Huber kernel,
 * BlockSolver_6_3 Schur complement (core/block_solver.hpp:354-486) and the
 * Levenberg loop (core/optimization_algorithm_levenberg.cpp:61-189).          */
typedef struct {
    int      n_poses;
    double*  poses;        /* [n_poses][7]: unit quaternion (x,y,z,w) then t (x,y,z) = SE3Quat of Tcw; in/out */
    const uint8_t* fixed;  /* [n_poses] 1 = setFixed(true) */
    const double*  intr;   /* [n_poses][4] fx,fy,cx,cy */
    int      n_points;
    double*  points;       /* [n_points][3] world xyz; in/out */
    int      n_edges;
    const int32_t* edge_pose;   /* [n_edges] */
    const int32_t* edge_point;  /* [n_edges] */
    const double*  obs;         /* [n_edges][2] u,v */
    const double*  info;        /* [n_edges] invSigma2 (information = info * I2) */
} ccm_ba_problem;

typedef struct {
    int    iterations;     /* optimize(n) */
    double huber_delta;    /* thHuber2D; <= 0 -> no robust kernel */
    /* Local BA's second stage (Optimizer.cpp:546-568): after `iterations`, mark
     * edges with chi2 > outlier_chi2 or non-positive depth as level 1, drop the
     * kernels and run `iterations2` more.  iterations2 <= 0 -> single stage. */
    int    iterations2;
    double outlier_chi2;   /* 5.991 */
    const volatile uint8_t* stop_flag; /* pbStopFlag, polled between LM iterations/trials; may be NULL */
    /* Relative residual |H dx - b| / |b| at which the iterative reduced-camera solve stops; 0 = default 1e-6.
     * The reference solves directly (sparse Cholesky); measured on config 5 the poses after 5 iterations differ from
     * a 1e-13 solve by 6e-13 (1e-11), 6e-11 (1e-9), 6e-9 (1e-7), 5e-7 (1e-5) against the 1e-5 contract; over the
     * reference's optimize(20) (9 iterations, LM corrects earlier inexactness) the final poses differ from the oracle's by
     * 2e-9 (1e-8), 9e-9 (1e-6), 2e-7 (1e-5), 1e-6 (1e-4), with identical iteration and trial counts and chi2 equal to 1e-11.
     * The default keeps three orders of magnitude to the contract and saves a sixth of the solve time against 1e-8. */
    double pcg_tol;
} ccm_ba_options;

typedef struct {
    int    iterations_done;   /* LM iterations run, both stages */
    int    trials;            /* linear solves */
    double chi2_initial;      /* activeRobustChi2 before the first iteration */
    double chi2_final;        /* activeRobustChi2 at the last accepted state */
    double lambda_final;
    int    stopped;           /* 1 if stop_flag ended the solve */
    /* per-stage wall seconds, mirroring G2OBatchStatistics (core/batch_stats.h).  They always add up to the LM loop's wall time;
     * below 200,000 edges the stream is not synchronised at every phase boundary (that cost a quarter of a local BA), so a phase's
     * GPU time is booked where the next necessary synchronisation falls (CCM_BA_TIMERS=1 forces exact phases). */
    double t_linearize, t_schur, t_solve, t_update;
    uint8_t* edge_outlier;    /* optional out [n_edges]: chi2 > outlier_chi2 || depth <= 0 at the end (:582) */
    /* structure of the reduced camera system and work of its solver */
    int32_t schur_blocks;     /* non-zero 6x6 blocks (upper triangle) */
    int64_t schur_pairs;      /* (landmark, pose-pair) contributions on this rank */
    int32_t pcg_iterations;   /* conjugate-gradient iterations, all trials (0 on the dense path) */
    int32_t pcg_fallbacks;    /* trials whose first solver gave up (pipelined -> classic PCG, PCG -> dense solve) */
    int32_t pcg_pipelined;    /* 1 = the two-kernel pipelined PCG iteration ran (pcg_tol >= 1e-7), 0 = the classic one or the dense solve */
} ccm_ba_result;

int ccm_ba_solve(ccm_ctx*, ccm_ba_problem*, const ccm_ba_options*, ccm_ba_result*);

/* Optimizer::PoseOptimizationClient (src/Optimizer.cpp:215-347), batched over frames: one free pose per
 * frame, one unary EdgeSE3ProjectXYZOnlyPose per frame feature with a MapPoint, Huber sqrt(5.991), four
 * rounds of optimize(10) restarted from the input pose, chi2 > 5.991 relabels outliers after each round.
 * Correspondences of frame f are rows first[f] .. first[f+1]-1 of points/obs/info.  On return poses holds
 * the optimised Tcw, outlier[] is Frame::mvbOutlier, n_inliers[f] the function's return value
 * (nInitialCorrespondences - nBad; 0 with fewer than 3 correspondences, pose untouched). */
typedef struct {
    int            n_frames;
    double*        poses;      /* [n_frames][7] in/out */
    const double*  intr;       /* [n_frames][4] fx fy cx cy */
    const int32_t* first;      /* [n_frames+1] */
    const double*  points;     /* [first[n_frames]][3] MapPoint world positions */
    const double*  obs;        /* [..][2] undistorted keypoints */
    const double*  info;       /* [..] invSigma2 of the keypoint's octave */
    uint8_t*       outlier;    /* [..] out */
    int32_t*       n_inliers;  /* [n_frames] out */
} ccm_pose_problem;
int ccm_pose_optimize(ccm_ctx*, ccm_pose_problem*);

/* Optimizer::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale) (src/Optimizer.cpp:867-1062), next row F4,
 * batched over candidate keyframe pairs: one VertexSim3Expmap, fixed points, EdgeSim3ProjectXYZ +
 * EdgeInverseSim3ProjectXYZ per correspondence with g2o's numeric Jacobians (delta 1e-9) and Huber(sqrt(th2));
 * optimize(5), prune chi2 > th2, optimize(5 or 10), classify.  Correspondence e of problem p (first[p] <= e <
 * first[p+1]) = one non-null vpMatches1[i] that passed :918-943: P1 = R1w*P3D1w + t1w, P2 = R2w*P3D2w + t2w (the
 * caller's float arithmetic, widened), obs1/info1 = pKF1->mvKeysUn[i] / mvInvLevelSigma2, obs2/info2 likewise for
 * i2.  sim3 = g2oS12 as qx,qy,qz,qw, tx,ty,tz, s (in/out; untouched when fewer than 10 correspondences survive the
 * first round, where the reference returns 0); inlier[e] = vpMatches1 entry kept; n_inliers[p] = the return value. */
typedef struct {
    int32_t        n_problems;
    double*        sim3;       /* [n_problems][8] in/out */
    const int32_t* fix_scale;  /* [n_problems] */
    const double*  K1;         /* [n_problems][4] fx, fy, cx, cy of pKF1 */
    const double*  K2;
    const int32_t* first;      /* [n_problems+1] */
    const double*  P1;         /* [..][3] */
    const double*  P2;
    const double*  obs1;       /* [..][2] */
    const double*  obs2;
    const double*  info1;      /* [..] */
    const double*  info2;
    const float*   th2;        /* [n_problems] */
    uint8_t*       inlier;     /* [..] out */
    int32_t*       n_inliers;  /* [n_problems] out */
} ccm_sim3_problem;
int ccm_optimize_sim3(ccm_ctx*, ccm_sim3_problem*);

/* The optimisation inside Optimizer::OptimizeEssentialGraphLoopClosure / OptimizeEssentialGraphMapFusion
 * (src/Optimizer.cpp:1064-1331, :1333-1574): one VertexSim3Expmap per keyframe (sim3 = Scw or the corrected Sim3,
 * :1094-1108; fixed = pLoopKF, :1110), one EdgeSim3 per loop / spanning-tree / covisibility edge built by the caller
 * exactly as :1123-1250 with edge_i = vertex 0, edge_j = vertex 1, measurement = Sji; identity information, numeric
 * Jacobians, Levenberg with lambda 1e-16, `iterations` = 20.  sim3 comes back as the CorrectedSiw of :1262.
 * Solved like :1072-1074 (BlockSolver_7_3 + sparse Cholesky): block-sparse normal equations, fill-reducing ordering once per call,
 * numeric factorisation per LM trial on the device; no dense matrix. */
typedef struct {
    int32_t        n_vertices;
    double*        sim3;          /* [n_vertices][8] in/out: qx,qy,qz,qw, tx,ty,tz, s */
    const uint8_t* fixed;         /* [n_vertices] */
    int32_t        fix_scale;     /* bFixScale */
    int32_t        n_edges;
    const int32_t* edge_i;        /* [n_edges] */
    const int32_t* edge_j;
    const double*  measurement;   /* [n_edges][8] */
    int32_t        iterations;
    int32_t        iterations_done;   /* out */
    double         chi2_initial, chi2_final;   /* out */
    /* out, the block-sparse solve: 7x7 blocks of the Cholesky factor (diagonal + lower, fill included), rounds of independent
     * columns (= launches per factorisation), device bytes of system + factor */
    int32_t        factor_blocks, factor_rounds;
    int64_t        solver_bytes;
} ccm_essential_graph;
int ccm_optimize_essential_graph(ccm_ctx*, ccm_essential_graph*);
/* Map point correction that follows it (:1300-1330): points[i] <- correctedSwr.map(Srw.map(points[i])) with r =
 * ref_vertex[i] (-1: leave the point), Srw = sim3_before[r] (vScw), correctedSwr = inverse(sim3_after[r]). */
int ccm_correct_map_points(ccm_ctx*, int n_points, double* points, const int32_t* ref_vertex, int n_vertices, const double* sim3_before,
                           const double* sim3_after);

/* Multi-GPU GBA: every rank calls ccm_ba_solve with the
 * SAME poses and ITS OWN landmark partition (points + their edges); the reduced
 * camera system is summed with one RCCL all-reduce per LM trial.  One rank
 * makes an id, the host program distributes it (e.g. torch.distributed
 * broadcast), every rank calls ccm_comm_init.  */
/* The landmark ranges ccm_ba_solve gives to the ranks (host only, no GPU): cuts[r] .. cuts[r+1]-1 -> rank r. */
int ccm_ba_landmark_cuts(const int32_t* edge_point, int n_edges, int n_points, int n_ranks, int32_t* cuts);
#define CCM_COMM_ID_BYTES 128
int ccm_comm_unique_id(uint8_t id[CCM_COMM_ID_BYTES]);
int ccm_comm_init(ccm_ctx*, const uint8_t id[CCM_COMM_ID_BYTES], int n_ranks, int rank);
/* Bring-your-own transport: a host that already has a collective layer (MPI, a test harness) attaches it instead of
 * RCCL.  The callbacks receive DEVICE pointers and the context's HIP stream; they must leave the reduced values in
 * place and return 0, or non-zero on failure (reported as CCM_E_COMM).  `destroy` runs when the context drops the
 * communicator.  The sharded solve is the same code path as with ccm_comm_init except for the all-reduce itself.
 * (tests/support/shm_transport.cpp, a shared-memory all-reduce between processes sharing ONE GPU, is such a transport:
 * it lets the sharded global BA be rehearsed end to end on a one-GPU box; it is not part of this library.) */
typedef struct {
    void* user;
    int  (*allreduce_f64)(void* user, double* dev, size_t n, int max_op, void* hip_stream);
    int  (*allreduce_u8_max)(void* user, uint8_t* dev, size_t n, void* hip_stream);
    void (*destroy)(void* user);
} ccm_comm_transport;
int ccm_comm_attach(ccm_ctx*, const ccm_comm_transport*, int n_ranks, int rank);
int ccm_comm_destroy(ccm_ctx*);

/* SE3Quat / Converter helpers (src/Converter.cc:40-56, 86-93): float32 4x4 row-major
 * Tcw <-> quaternion+translation double[7]. Host only. */
int ccm_pose_from_mat4f(const float* T16, double* pose7);
int ccm_pose_to_mat4f(const double* pose7, float* T16);

#ifdef __cplusplus
}
#endif
#endif
