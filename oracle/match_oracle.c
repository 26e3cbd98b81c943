/* match_oracle.c -- CPU restatement of the ORBmatcher hot loop.  TEST INFRASTRUCTURE (see oracle.h).
 * Reference: cslam/src/ORBmatcher.cpp (both source trees are byte-identical). */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ORBmatcher::DescriptorDistance, ORBmatcher.cpp:1653-1669 -- the bit-hack as written there */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* running best / second best of ORBmatcher.cpp:220-245 over every train row */
void orc_hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    for (int i = 0; i < nq; i++) {
        int b1 = 256, bi = -1, b2 = 256;
        for (int j = 0; j < nt; j++) {
            const int d = orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi; best_dist[i] = b1; second_dist[i] = b2;
    }
}

/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cpp:1607-1648 */
void orc_three_maxima(const int32_t* hs, int L, int32_t* ind)
{
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = hs[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
        else if (s > max3) { max3 = s; i3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
    else if (max3 < 0.1f * (float)max1) { i3 = -1; }
    ind[0] = i1; ind[1] = i2; ind[2] = i3;
}

/* ORBmatcher::SearchByBoW, ORBmatcher.cpp:178-306 (valid2 == NULL) and :565-698 (valid2 != NULL).
 * FeatureVector = std::map<NodeId, vector<feature idx>>: nodes ascending, indices ascending inside. */
typedef struct { int32_t node; int32_t idx; } nf;
static int nf_cmp(const void* a, const void* b)
{
    const nf* x = (const nf*)a; const nf* y = (const nf*)b;
    if (x->node != y->node) return x->node < y->node ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int orc_match_bow(float nnratio, int check_ori, int th, int strict_th,
                  const uint8_t* desc1, const int32_t* node1, const uint8_t* valid1, const float* angle1, int n1,
                  const uint8_t* desc2, const int32_t* node2, const uint8_t* valid2, const float* angle2, int n2,
                  int32_t* match12)
{
    enum { HISTO = 30 };
    nf* f1 = (nf*)malloc(sizeof(nf) * (n1 > 0 ? n1 : 1));
    nf* f2 = (nf*)malloc(sizeof(nf) * (n2 > 0 ? n2 : 1));
    uint8_t* taken2 = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1);
    int32_t* hist = (int32_t*)malloc(sizeof(int32_t) * HISTO * (size_t)(n1 > 0 ? n1 : 1));
    int32_t hn[HISTO];
    memset(hn, 0, sizeof hn);
    int m1 = 0, m2 = 0;
    for (int i = 0; i < n1; i++) { match12[i] = -1; if (node1[i] >= 0) { f1[m1].node = node1[i]; f1[m1].idx = i; m1++; } }
    for (int i = 0; i < n2; i++) if (node2[i] >= 0) { f2[m2].node = node2[i]; f2[m2].idx = i; m2++; }
    qsort(f1, m1, sizeof(nf), nf_cmp);
    qsort(f2, m2, sizeof(nf), nf_cmp);
    const float factor = 1.0f / HISTO;                                  /* :191 */
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < m1 && b < m2) {
        if (f1[a].node == f2[b].node) {
            int ae = a, be = b;
            while (ae < m1 && f1[ae].node == f1[a].node) ae++;
            while (be < m2 && f2[be].node == f2[b].node) be++;
            for (int i = a; i < ae; i++) {
                const int idx1 = f1[i].idx;
                if (!valid1[idx1]) continue;                            /* :212-216 */
                int bd1 = 256, bi = -1, bd2 = 256;
                for (int j = b; j < be; j++) {
                    const int idx2 = f2[j].idx;
                    if (taken2[idx2]) continue;                         /* :228 / :619 */
                    if (valid2 && !valid2[idx2]) continue;              /* :619-623 */
                    const int d = orc_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (d < bd1) { bd2 = bd1; bd1 = d; bi = idx2; }
                    else if (d < bd2) bd2 = d;
                }
                const int pass = strict_th ? (bd1 < th) : (bd1 <= th);  /* :641 vs :247 */
                if (pass && (float)bd1 < nnratio * (float)bd2) {
                    match12[idx1] = bi;
                    taken2[bi] = 1;
                    if (check_ori) {
                        float rot = angle1[idx1] - angle2[bi];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);              /* :260: bins 0..12 only, as written */
                        if (bin == HISTO) bin = 0;
                        hist[bin * (size_t)n1 + hn[bin]++] = idx1;
                    }
                    nmatches++;
                }
            }
            a = ae; b = be;
        } else if (f1[a].node < f2[b].node) {
            while (a < m1 && f1[a].node < f2[b].node) a++;               /* lower_bound */
        } else {
            while (b < m2 && f2[b].node < f1[a].node) b++;
        }
    }
    if (check_ori) {
        int32_t ind[3];
        orc_three_maxima(hn, HISTO, ind);
        for (int i = 0; i < HISTO; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int j = 0; j < hn[i]; j++) { match12[hist[i * (size_t)n1 + j]] = -1; nmatches--; }
        }
    }
    free(f1); free(f2); free(taken2); free(hist);
    return nmatches;
}

/* ---------------------------------------------------------------- F1: windowed matching
 * Frame grid: AssignFeaturesToGrid / PosInGrid (src/Frame.cpp:103-118, 255-266), GetFeaturesInArea
 * (src/Frame.cpp:200-253), and ORBmatcher::SearchByProjection(Frame&, vector<mpptr>&, th)
 * (cslam/src/ORBmatcher.cpp:71-148). */
typedef struct { int n, cols, rows; float min_x, min_y, inv_w, inv_h; int* first; int* items; } orc_grid;

static orc_grid* grid_build(int n, const float* kx, const float* ky, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows)
{
    orc_grid* g = (orc_grid*)calloc(1, sizeof *g);
    g->n = n; g->cols = cols; g->rows = rows; g->min_x = min_x; g->min_y = min_y; g->inv_w = inv_w; g->inv_h = inv_h;
    g->first = (int*)calloc((size_t)cols * rows + 1, sizeof(int));
    g->items = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    int* cell = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        const int px = (int)roundf((kx[i] - min_x) * inv_w), py = (int)roundf((ky[i] - min_y) * inv_h);
        cell[i] = (px < 0 || px >= cols || py < 0 || py >= rows) ? -1 : px * rows + py;      /* mGrid[x][y] */
        if (cell[i] >= 0) g->first[cell[i] + 1]++;
    }
    for (int c = 0; c < cols * rows; c++) g->first[c + 1] += g->first[c];
    int* fill = (int*)malloc(sizeof(int) * ((size_t)cols * rows + 1));
    memcpy(fill, g->first, sizeof(int) * ((size_t)cols * rows + 1));
    for (int i = 0; i < n; i++) if (cell[i] >= 0) g->items[fill[cell[i]]++] = i;            /* push_back in index order */
    free(cell); free(fill);
    return g;
}
static void grid_free(orc_grid* g) { free(g->first); free(g->items); free(g); }

static int features_in_area(const orc_grid* g, const float* kx, const float* ky, const int32_t* oct,
                            float x, float y, float r, int minLevel, int maxLevel, int32_t* out, int cap)
{
    int n = 0;
    const int nMinCellX = (int)floorf((x - g->min_x - r) * g->inv_w) > 0 ? (int)floorf((x - g->min_x - r) * g->inv_w) : 0;
    if (nMinCellX >= g->cols) return 0;
    int nMaxCellX = (int)ceilf((x - g->min_x + r) * g->inv_w); if (nMaxCellX > g->cols - 1) nMaxCellX = g->cols - 1;
    if (nMaxCellX < 0) return 0;
    const int nMinCellY = (int)floorf((y - g->min_y - r) * g->inv_h) > 0 ? (int)floorf((y - g->min_y - r) * g->inv_h) : 0;
    if (nMinCellY >= g->rows) return 0;
    int nMaxCellY = (int)ceilf((y - g->min_y + r) * g->inv_h); if (nMaxCellY > g->rows - 1) nMaxCellY = g->rows - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * g->rows + iy;
            for (int k = g->first[c]; k < g->first[c + 1]; k++) {
                const int i = g->items[k];
                if (bCheckLevels) {
                    if (oct[i] < minLevel) continue;
                    if (maxLevel >= 0 && oct[i] > maxLevel) continue;
                }
                const float dx = kx[i] - x, dy = ky[i] - y;
                if (fabsf(dx) < r && fabsf(dy) < r) { if (n < cap) out[n] = i; n++; }
            }
        }
    return n;
}

int orc_features_in_area(int n, const float* kx, const float* ky, const int32_t* oct, float min_x, float min_y, float inv_w, float inv_h,
                         int cols, int rows, float x, float y, float r, int minLevel, int maxLevel, int32_t* out, int cap)
{
    orc_grid* g = grid_build(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows);
    const int m = features_in_area(g, kx, ky, oct, x, y, r, minLevel, maxLevel, out, cap);
    grid_free(g);
    return m;
}

int orc_search_by_projection(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc,
                             float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                             int n_mp, const uint8_t* in_view, const int32_t* level, const float* view_cos, const float* proj_x,
                             const float* proj_y, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                             uint8_t* occupied, float th, float nnratio, int32_t* match)
{
    orc_grid* g = grid_build(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) match[i] = -1;
    int nmatches = 0;
    const int bFactor = th != 1.0;
    for (int m = 0; m < n_mp; m++) {
        if (!in_view[m]) continue;                                            /* mbTrackInView, isBad */
        const int lvl = level[m];
        float r = view_cos[m] > 0.998 ? 2.5f : 4.0f;                          /* RadiusByViewingCos :150-156 */
        if (bFactor) r *= th;
        const int nc = features_in_area(g, kx, ky, oct, proj_x[m], proj_y[m], r * scale_factors[lvl], lvl - 1, lvl, cand, n);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            if (occupied[idx]) continue;                                       /* mvpMapPoints[idx] with Observations()>0 */
            const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)m, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct[idx]; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = oct[idx]; bestDist2 = dist; }
        }
        if (bestDist <= 100) {                                                 /* TH_HIGH */
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match[bestIdx] = m;                                                /* F.mvpMapPoints[bestIdx] = pMP */
            occupied[bestIdx] = mp_has_obs[m];
            nmatches++;
        }
    }
    free(cand); grid_free(g);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th) (ORBmatcher.cpp:1350-1476), the matcher of
 * TrackWithMotionModel.  valid[i]: last-frame feature i has a map point, is not an outlier, projects with
 * positive depth inside the current frame's bounds; (u,v)[i] is that projection (the caller computes it in
 * float exactly as :1382-1393). */
int orc_search_by_projection_frame(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc, const float* angle,
                                   float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                                   int n_last, const uint8_t* valid, const float* u, const float* v, const int32_t* last_octave,
                                   const float* last_angle, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                                   uint8_t* occupied, float th, int check_ori, int orb_dist, int32_t* match)
{
    enum { HISTO = 30 };
    orc_grid* g = grid_build(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int32_t* hist = (int32_t*)malloc(sizeof(int32_t) * HISTO * (size_t)(n_last > 0 ? n_last : 1));
    int32_t hn[HISTO]; memset(hn, 0, sizeof hn);
    for (int i = 0; i < n; i++) match[i] = -1;
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    for (int i = 0; i < n_last; i++) {
        if (!valid[i]) continue;
        const int lo = last_octave[i];
        const float radius = th * scale_factors[lo];
        const int nc = features_in_area(g, kx, ky, oct, u[i], v[i], radius, lo - 1, lo + 1, cand, n);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            if (occupied[i2]) continue;
            const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)i, desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {                        /* TH_HIGH (:1432) or the ORBdist argument (:1556) */
            match[bestIdx2] = i;
            occupied[bestIdx2] = mp_has_obs[i];
            nmatches++;
            if (check_ori) {
                float rot = last_angle[i] - angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO) bin = 0;
                hist[bin * (size_t)n_last + hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int32_t ind[3];
        orc_three_maxima(hn, HISTO, ind);
        for (int b = 0; b < HISTO; b++) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int j = 0; j < hn[b]; j++) { match[hist[b * (size_t)n_last + j]] = -1; nmatches--; }
        }
    }
    free(cand); free(hist); grid_free(g);
    return nmatches;
}

/* ORBmatcher::SearchForInitialization (ORBmatcher.cpp:448-563).  prev_matched (in/out) = vbPrevMatched. */
int orc_search_for_initialization(int n1, const int32_t* oct1, const uint8_t* desc1, const float* angle1,
                                  int n2, const float* kx2, const float* ky2, const int32_t* oct2, const uint8_t* desc2, const float* angle2,
                                  float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                                  float* prev_matched_xy, int window, float nnratio, int check_ori, int32_t* matches12)
{
    enum { HISTO = 30 };
    orc_grid* g = grid_build(n2, kx2, ky2, min_x, min_y, inv_w, inv_h, cols, rows);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    int* matched_dist = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* m21 = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int32_t* hist = (int32_t*)malloc(sizeof(int32_t) * HISTO * (size_t)(n1 > 0 ? n1 : 1));
    int32_t hn[HISTO]; memset(hn, 0, sizeof hn);
    for (int i = 0; i < n2; i++) { matched_dist[i] = 2147483647; m21[i] = -1; }
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = oct1[i1];
        if (level1 > 0) continue;
        const int nc = features_in_area(g, kx2, ky2, oct2, prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], (float)window, level1, level1, cand, n2);
        if (nc == 0) continue;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
            if (matched_dist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= 50) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { matches12[m21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2; m21[bestIdx2] = i1; matched_dist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    float rot = angle1[i1] - angle2[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO) bin = 0;
                    hist[bin * (size_t)n1 + hn[bin]++] = i1;
                }
            }
        }
    }
    if (check_ori) {
        int32_t ind[3];
        orc_three_maxima(hn, HISTO, ind);
        for (int b = 0; b < HISTO; b++) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int j = 0; j < hn[b]; j++) {
                const int idx1 = hist[b * (size_t)n1 + j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) { prev_matched_xy[2 * i1] = kx2[matches12[i1]]; prev_matched_xy[2 * i1 + 1] = ky2[matches12[i1]]; }
    free(cand); free(matched_dist); free(m21); free(hist); grid_free(g);
    return nmatches;
}

/* Selection loop of ORBmatcher::Fuse (ORBmatcher.cpp:914-955 with the chi2 test, :1072-1100 without): per map point
 * the most similar keyframe feature inside the window, at level nPredictedLevel-1 or nPredictedLevel.
 * best_idx = -1 unless bestDist <= TH_LOW. */
void orc_fuse_select(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc,
                     float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                     const float* inv_level_sigma2, int n_mp, const uint8_t* valid, const float* u, const float* v,
                     const int32_t* level, const uint8_t* mp_desc, float th, int chi2_check, int accept_th, int32_t* best_idx, int32_t* best_dist)
{
    orc_grid* g = grid_build(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int m = 0; m < n_mp; m++) {
        best_idx[m] = -1; best_dist[m] = 256;
        if (!valid[m]) continue;
        const int lvl = level[m];
        const float radius = th * scale_factors[lvl];
        const int nc = features_in_area(g, kx, ky, oct, u[m], v[m], radius, -1, -1, cand, n);
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            const int kpLevel = oct[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            if (chi2_check) {
                const float ex = u[m] - kx[idx], ey = v[m] - ky[idx];
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)m, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_dist[m] = bestDist;
        if (bestDist <= accept_th) best_idx[m] = bestIdx;   /* TH_LOW in Fuse, TH_HIGH in SearchBySim3 (:1226, :1306) */
    }
    free(cand); grid_free(g);
}

/* ORBmatcher::SearchBySim3 (ORBmatcher.cpp:1124-1348) after the caller's projections: both directions pick the most
 * similar feature (<= TH_HIGH), a pair is kept when the two directions agree (:1330-1345).  match12[i1] = i2 or -1. */
int orc_search_by_sim3(int n1, const float* kx1, const float* ky1, const int32_t* oct1, const uint8_t* desc1, const float* grid1, const float* sf1,
                       int n2, const float* kx2, const float* ky2, const int32_t* oct2, const uint8_t* desc2, const float* grid2, const float* sf2,
                       int cols, int rows,
                       const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* mpdesc1,
                       const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* mpdesc2,
                       float th, int32_t* match12)
{
    int32_t* m1 = (int32_t*)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1)), *d1 = (int32_t*)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1));
    int32_t* m2 = (int32_t*)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1)), *d2 = (int32_t*)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    /* map points of KF1 are searched in KF2 and vice versa */
    orc_fuse_select(n2, kx2, ky2, oct2, desc2, grid2[0], grid2[1], grid2[2], grid2[3], cols, rows, sf2, 0, n1, valid1, u1, v1, level1, mpdesc1, th, 0, 100, m1, d1);
    orc_fuse_select(n1, kx1, ky1, oct1, desc1, grid1[0], grid1[1], grid1[2], grid1[3], cols, rows, sf1, 0, n2, valid2, u2, v2, level2, mpdesc2, th, 0, 100, m2, d2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    free(m1); free(d1); free(m2); free(d2);
    return nFound;
}

/* ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cpp:308-446) after the caller's
 * projection (valid, u, v, predicted level).  matched (in/out) = vpMatched[idx] != null; observed[m] = the map point is
 * already an observation of pKF (GetIndexInKeyFrame != -1: it is re-mapped by the caller, vpMatched is not touched, :414-434).
 * best_idx[m] = chosen feature or -1.  Returns nmatches. */
int orc_search_by_projection_sim3(int n, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc,
                                  float min_x, float min_y, float inv_w, float inv_h, int cols, int rows, const float* scale_factors,
                                  int n_mp, const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* mp_desc,
                                  const uint8_t* observed, uint8_t* matched, float th, int32_t* best_idx)
{
    orc_grid* g = grid_build(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int nmatches = 0;
    for (int m = 0; m < n_mp; m++) {
        best_idx[m] = -1;
        if (!valid[m]) continue;
        const int lvl = level[m];
        const float radius = th * scale_factors[lvl];
        const int nc = features_in_area(g, kx, ky, oct, u[m], v[m], radius, -1, -1, cand, n);
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            if (matched[idx]) continue;
            if (oct[idx] < lvl - 1 || oct[idx] > lvl) continue;
            const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)m, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) {
            best_idx[m] = bestIdx;
            if (!observed[m]) { matched[bestIdx] = 1; nmatches++; }
        }
    }
    free(cand); grid_free(g);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (ORBmatcher.cpp:159-176); F12 row-major 3x3 float */
static int check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* F12, float sigma2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

/* ORBmatcher::SearchForTriangulation (ORBmatcher.cpp:700-852).  has_mp = the feature already has a map point;
 * (ex, ey) = the epipole in image 2 (:707-714, computed by the caller).  This version never sets vbMatched2, so
 * several features of image 1 may choose the same feature of image 2.  match12[i1] = i2 or -1. */
int orc_search_for_triangulation(const uint8_t* desc1, const int32_t* node1, const uint8_t* has_mp1, const float* x1, const float* y1,
                                 const float* angle1, int n1,
                                 const uint8_t* desc2, const int32_t* node2, const uint8_t* has_mp2, const float* x2, const float* y2,
                                 const float* angle2, const int32_t* oct2, int n2,
                                 const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                 int check_ori, int32_t* match12)
{
    enum { HISTO = 30 };
    nf* f1 = (nf*)malloc(sizeof(nf) * (n1 > 0 ? n1 : 1));
    nf* f2 = (nf*)malloc(sizeof(nf) * (n2 > 0 ? n2 : 1));
    int32_t* hist = (int32_t*)malloc(sizeof(int32_t) * HISTO * (size_t)(n1 > 0 ? n1 : 1));
    int32_t hn[HISTO];
    memset(hn, 0, sizeof hn);
    int m1 = 0, m2 = 0;
    for (int i = 0; i < n1; i++) { match12[i] = -1; if (node1[i] >= 0) { f1[m1].node = node1[i]; f1[m1].idx = i; m1++; } }
    for (int i = 0; i < n2; i++) if (node2[i] >= 0) { f2[m2].node = node2[i]; f2[m2].idx = i; m2++; }
    qsort(f1, m1, sizeof(nf), nf_cmp);
    qsort(f2, m2, sizeof(nf), nf_cmp);
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < m1 && b < m2) {
        if (f1[a].node == f2[b].node) {
            int ae = a, be = b;
            while (ae < m1 && f1[ae].node == f1[a].node) ae++;
            while (be < m2 && f2[be].node == f2[b].node) be++;
            for (int i = a; i < ae; i++) {
                const int idx1 = f1[i].idx;
                if (has_mp1[idx1]) continue;                                           /* :744-746 */
                int bestDist = 50, bestIdx2 = -1;                                      /* TH_LOW */
                for (int j = b; j < be; j++) {
                    const int idx2 = f2[j].idx;
                    if (has_mp2[idx2]) continue;                                       /* :763 (vbMatched2 stays false) */
                    const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist > 50 || dist > bestDist) continue;                        /* :770 */
                    const float distex = ex - x2[idx2], distey = ey - y2[idx2];
                    if (distex * distex + distey * distey < 100 * scale_factors2[oct2[idx2]]) continue;   /* :775-777 */
                    if (check_dist_epipolar_line(x1[idx1], y1[idx1], x2[idx2], y2[idx2], F12, level_sigma2_2[oct2[idx2]])) {
                        bestIdx2 = idx2; bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) {
                        float rot = angle1[idx1] - angle2[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO) bin = 0;
                        hist[bin * (size_t)n1 + hn[bin]++] = idx1;
                    }
                }
            }
            a = ae; b = be;
        } else if (f1[a].node < f2[b].node) {
            while (a < m1 && f1[a].node < f2[b].node) a++;
        } else {
            while (b < m2 && f2[b].node < f1[a].node) b++;
        }
    }
    if (check_ori) {
        int32_t ind[3];
        orc_three_maxima(hn, HISTO, ind);
        for (int i = 0; i < HISTO; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int j = 0; j < hn[i]; j++) { match12[hist[i * (size_t)n1 + j]] = -1; nmatches--; }
        }
    }
    free(f1); free(f2); free(hist);
    return nmatches;
}
