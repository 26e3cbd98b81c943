/* oracle.h -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so.  Nothing under motioncheck_ccm_slam_amd/ includes, links or
 * calls it; the product path has no CPU fallback.
 *
 * PARITY STATUS:
 *  - The reference cannot be built here (needs OpenCV, Eigen3, Boost, ROS: absent
 *    and not installable) and ships no tests, fixtures or golden vectors.
 *  - FAST, resize, GaussianBlur, fastAtan2, cvRound are OpenCV's (un-vendored,
 *    un-pinned); SimplicialLDLT / Quaterniond are Eigen's.  Their published
 *    algorithms are restated from SURVEY.md section 12.  PARITY UNPINNED for
 *    these primitives.
 *  - Pinned by known-answer tests derived from the reference text
 *    (tests/test_oracle_kat.py): feature quotas {217,181,151,126,105,87,73,60},
 *    umax table, level sizes, Hamming bit-hack == popcount, descriptor bit
 *    packing, analytic vs numeric Jacobian, Schur == full solve.
 */
#ifndef CCM_ORACLE_H
#define CCM_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int nfeatures; float scale_factor; int nlevels; int ini_th; int min_th; } orc_orb_params;
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orc_keypoint;

/* ---- ORB ---- */
int  orc_orb_tables(const orc_orb_params*, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int32_t* nfeat, int32_t* umax);
int  orc_orb_level_sizes(const orc_orb_params*, int w, int h, int32_t* lw, int32_t* lh);
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
int  orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold,
                  int32_t* xy, int32_t* score, int max);
void orc_blur7_sigma2(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
float orc_fast_atan2(float y, float x);
int  orc_round_half_even(double v);
float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy, const int32_t* umax);
void orc_orb_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t* desc32);
void orc_set_sincos_libm(int on);   /* 1: libm cosf/sinf as ORBextractor.cpp:105 (measurement only), 0: include/ccm_sincos.h */
int  orc_distribute_octree(const int32_t* xy, const int32_t* score, int n, int minX, int maxX,
                           int minY, int maxY, int N, int32_t* out_idx);
/* Full operator().  level_out (optional) receives pointers-free copies of the pyramid:
 * level l at level_out + level_off[l], tight pitch lw[l].  cand_* (optional): FAST candidates of
 * `cand_level` before the quadtree.  Returns keypoint count or <0. */
int  orc_orb_extract(const orc_orb_params*, const uint8_t* img, int w, int h, int stride,
                     orc_keypoint* kps, uint8_t* desc, int max_kps,
                     uint8_t* level_out, size_t level_out_bytes,
                     int cand_level, int32_t* cand_xy, int32_t* cand_score, int cand_max, int32_t* cand_n);

/* all-core CPU baseline driver (pthreads over frames, then over consecutive pairs); seconds[0] extraction, [1] matching */
long orc_bench_extract_match_mt(const orc_orb_params*, const uint8_t* imgs, int n, int w, int h, int threads, double* seconds);

/* ---- matcher ---- */
int  orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
void orc_hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
void orc_three_maxima(const int32_t* hist_sizes, int L, int32_t* ind3);
int  orc_match_bow(float nnratio, int check_ori, int th, int strict_th,
                   const uint8_t* desc1, const int32_t* node1, const uint8_t* valid1, const float* angle1, int n1,
                   const uint8_t* desc2, const int32_t* node2, const uint8_t* valid2, const float* angle2, int n2,
                   int32_t* match12);

/* F1: Frame::GetFeaturesInArea and ORBmatcher::SearchByProjection(Frame&, vector<mpptr>&, th) */
int  orc_features_in_area(int n, const float* kx, const float* ky, const int32_t* oct, float min_x, float min_y, float inv_w, float inv_h,
                          int cols, int rows, float x, float y, float r, int minLevel, int maxLevel, int32_t* out, int cap);
int  orc_search_by_projection(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc,
                              float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                              int n_mp, const uint8_t* in_view, const int32_t* level, const float* view_cos, const float* proj_x,
                              const float* proj_y, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                              uint8_t* occupied, float th, float nnratio, int32_t* match);

int  orc_search_by_projection_frame(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc, const float* angle,
                                    float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                                    int n_last, const uint8_t* valid, const float* u, const float* v, const int32_t* last_octave,
                                    const float* last_angle, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                                    uint8_t* occupied, float th, int check_ori, int orb_dist, int32_t* match);

int  orc_search_for_initialization(int n1, const int32_t* oct1, const uint8_t* desc1, const float* angle1,
                                   int n2, const float* kx2, const float* ky2, const int32_t* oct2, const uint8_t* desc2, const float* angle2,
                                   float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                                   float* prev_matched_xy, int window, float nnratio, int check_ori, int32_t* matches12);

void orc_fuse_select(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc,
                     float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                     const float* inv_level_sigma2, int n_mp, const uint8_t* valid, const float* u, const float* v,
                     const int32_t* level, const uint8_t* mp_desc, float th, int chi2_check, int accept_th, int32_t* best_idx, int32_t* best_dist);

int orc_search_by_sim3(int n1, const float* kx1, const float* ky1, const int32_t* oct1, const uint8_t* desc1, const float* grid1, const float* sf1,
                       int n2, const float* kx2, const float* ky2, const int32_t* oct2, const uint8_t* desc2, const float* grid2, const float* sf2,
                       int cols, int rows,
                       const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* mpdesc1,
                       const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* mpdesc2,
                       float th, int32_t* match12);
int orc_search_by_projection_sim3(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc,
                                  float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                                  int n_mp, const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc,
                                  const uint8_t* observed, uint8_t* matched, float th, int32_t* best_idx);
int orc_search_for_triangulation(const uint8_t* desc1, const int32_t* node1, const uint8_t* has_mp1, const float* x1, const float* y1,
                                 const float* angle1, int n1,
                                 const uint8_t* desc2, const int32_t* node2, const uint8_t* has_mp2, const float* x2, const float* y2,
                                 const float* angle2, const int32_t* oct2, int n2,
                                 const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                 int check_ori, int32_t* match12);

/* ---- vocabulary tree (bow_oracle.c) ---- */
typedef struct orc_voc orc_voc;
orc_voc* orc_voc_create(int k, int L, int n_nodes, const int32_t* parent, const uint8_t* desc, const double* weight);
void orc_voc_destroy(orc_voc*);
void orc_voc_transform(const orc_voc*, const uint8_t* features, int n, int levelsup, int32_t* word_id, double* weight, int32_t* node_id);
int orc_bow_vector(int n, const int32_t* word_id, const double* weight, const int32_t* node_id, int weighting, int scoring,
                   int32_t* out_id, double* out_val, int32_t* fv_node);
double orc_bow_score_l1(int n1, const int32_t* id1, const double* v1, int n2, const int32_t* id2, const double* v2);
int orc_distinctive_descriptor(const uint8_t* desc, int n);

/* ---- bundle adjustment ---- */
typedef struct {
    int n_poses; double* poses; const uint8_t* fixed; const double* intr;
    int n_points; double* points;
    int n_edges; const int32_t* edge_pose; const int32_t* edge_point; const double* obs; const double* info;
} orc_ba_problem;
typedef struct { int iterations; double huber_delta; int iterations2; double outlier_chi2; } orc_ba_options;
typedef struct { int iterations_done; int trials; double chi2_initial, chi2_final, lambda_final; } orc_ba_result;
int  orc_ba_solve(orc_ba_problem*, const orc_ba_options*, orc_ba_result*, uint8_t* edge_outlier);
/* building blocks exposed for the known-answer tests */
void orc_ba_edge(const double* pose7, const double* intr4, const double* pt3, const double* obs2,
                 double* err2, double* Jpoint_2x3, double* Jpose_2x6);
void orc_se3_exp_mul(const double* delta6, const double* pose7_in, double* pose7_out);
void orc_pose_from_mat4f(const float* T16, double* pose7);
void orc_pose_to_mat4f(const double* pose7, float* T16);
/* F2: Optimizer::PoseOptimizationClient for one frame */
int  orc_pose_optimize(double* pose7, const double* intr4, int n, const double* pts, const double* obs,
                       const double* info, uint8_t* outlier, int* n_inliers);
/* One linearisation at the current state: dense reduced system for checks.
 * Hschur (6P x 6P, P = free poses, row-major, full symmetric) and bschur; returns P. */
int  orc_ba_reduced_system(const orc_ba_problem*, double huber_delta, double lambda,
                           double* Hschur, double* bschur, int32_t* free_index);
/* reduced solve selection (0 automatic: block-sparse Cholesky above 400 free keyframes, 1 dense Cholesky, 2 block-sparse) */
void orc_ba_set_solver(int mode);
int  orc_ba_solve_once(const orc_ba_problem*, double huber_delta, double lambda, int mode, double* xp, double* xl, double* stats);
/* block-sparse Cholesky of the reduced camera system (bchol_oracle.c; linear_solver_eigen.h:106-136,165-222) */
typedef struct orc_bchol orc_bchol;
orc_bchol* orc_bchol_new(int nb, const uint8_t* adj);
orc_bchol* orc_bchol_new_bs(int nb, const uint8_t* adj, int bs);   /* bs = 6 or 7 */
void orc_bchol_free(orc_bchol*);
long orc_bchol_nnz(const orc_bchol*);
double orc_bchol_flops(const orc_bchol*);
int  orc_bchol_factor(orc_bchol*, const int32_t* idx, const double* blk);
void orc_bchol_solve(const orc_bchol*, const double* b, double* x);

#ifdef __cplusplus
}
#endif

/* ---- F4: Optimizer::OptimizeSim3 (ba_oracle.c) ---- */
void orc_sim3_exp(const double* update7, double* sim3);
void orc_sim3_mul(const double* a, const double* b, double* o);
void orc_sim3_inverse(const double* a, double* o);
void orc_sim3_log(const double* sim3, double* log7);
void orc_ess_set_solver(int mode);      /* 0 automatic (block-sparse above 400 free vertices), 1 dense, 2 block-sparse */
int orc_essential_graph(int n, double* sim3, const uint8_t* fixed, int fix_scale, int ne, const int32_t* ei, const int32_t* ej,
                        const double* meas, int iterations, double* chi2_out);
int orc_optimize_sim3(double* sim3, int fix_scale, const double* K1, const double* K2, int n, const double* P1, const double* P2,
                      const double* obs1, const double* obs2, const double* info1, const double* info2, float th2, uint8_t* inlier);
#endif
