/* bow_oracle.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the vocabulary-tree path:
 *   DBoW2::TemplatedVocabulary::transform      cslam/thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1258
 *   DBoW2::BowVector::addWeight / normalize    cslam/thirdparty/DBoW2/DBoW2/BowVector.cpp:35-87
 *   DBoW2::L1Scoring::score                    cslam/thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68
 *   MapPoint::ComputeDistinctiveDescriptors    cslam/src/MapPoint.cpp:929-994
 * Parity pinning: ORBvoc.txt is not in the tree (.MISSING_LARGE_BLOBS), so the tests run on a
 * synthetic tree; the algorithm itself is integer Hamming + double sums in a fixed order, restated literally. */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* The vocabulary as TemplatedVocabulary::loadFromTextFile builds it (:1338-1423): node 0 is the root, node i > 0 has
 * parent[i] < i..., children in node-id order, a node without children is a word, words are numbered in node order. */
struct orc_voc { int k, L, n; int32_t* parent; int32_t* cfirst; int32_t* cn; int32_t* clist; const uint8_t* desc; const double* weight; int32_t* word_id; };

orc_voc* orc_voc_create(int k, int L, int n_nodes, const int32_t* parent, const uint8_t* desc, const double* weight)
{
    orc_voc* v = (orc_voc*)calloc(1, sizeof *v);
    v->k = k; v->L = L; v->n = n_nodes; v->desc = desc; v->weight = weight;
    v->parent = (int32_t*)malloc(sizeof(int32_t) * n_nodes);
    memcpy(v->parent, parent, sizeof(int32_t) * n_nodes);
    v->cfirst = (int32_t*)calloc(n_nodes + 1, sizeof(int32_t));
    v->cn = (int32_t*)calloc(n_nodes, sizeof(int32_t));
    v->clist = (int32_t*)malloc(sizeof(int32_t) * n_nodes);
    v->word_id = (int32_t*)malloc(sizeof(int32_t) * n_nodes);
    for (int i = 1; i < n_nodes; i++) v->cn[parent[i]]++;
    for (int i = 0; i < n_nodes; i++) v->cfirst[i + 1] = v->cfirst[i] + v->cn[i];
    int32_t* fill = (int32_t*)calloc(n_nodes, sizeof(int32_t));
    for (int i = 1; i < n_nodes; i++) v->clist[v->cfirst[parent[i]] + fill[parent[i]]++] = i;     /* push_back in node order */
    free(fill);
    int w = 0;
    for (int i = 0; i < n_nodes; i++) v->word_id[i] = (i > 0 && v->cn[i] == 0) ? w++ : 0;
    return v;
}
void orc_voc_destroy(orc_voc* v) { if (!v) return; free(v->parent); free(v->cfirst); free(v->cn); free(v->clist); free(v->word_id); free(v); }

/* transform(feature, word_id, weight, nid, levelsup), :1217-1258.  nid starts at 0 (the reference leaves it
 * uninitialised when the descent ends above nid_level). */
void orc_voc_transform(const orc_voc* v, const uint8_t* features, int n, int levelsup, int32_t* word_id, double* weight, int32_t* node_id)
{
    const int nid_level = v->L - levelsup;
    for (int f = 0; f < n; f++) {
        const uint8_t* feat = features + 32 * (size_t)f;
        int nid = 0, final_id = 0, current_level = 0;
        if (v->cn[0] == 0) { word_id[f] = 0; weight[f] = 0; node_id[f] = 0; continue; }
        do {
            ++current_level;
            const int32_t* nodes = v->clist + v->cfirst[final_id];
            const int nn = v->cn[final_id];
            final_id = nodes[0];
            double best_d = orc_descriptor_distance(feat, v->desc + 32 * (size_t)final_id);
            for (int j = 1; j < nn; j++) {
                const int id = nodes[j];
                const double d = orc_descriptor_distance(feat, v->desc + 32 * (size_t)id);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (v->cn[final_id] != 0);
        word_id[f] = v->word_id[final_id];
        weight[f] = v->weight[final_id];
        node_id[f] = nid;
    }
}

/* transform(features, BowVector, FeatureVector, levelsup), :1125-1193, for TF_IDF / TF weighting (addWeight) or IDF /
 * BINARY (addIfNotExist); scoring decides the normalisation (ScoringObject.h:74-89).  Output: the map's (id, value)
 * pairs in key order; fv_node[i] = node of feature i or -1 when its word is stopped (w <= 0).  Returns the map size. */
int orc_bow_vector(int n, const int32_t* word_id, const double* weight, const int32_t* node_id, int weighting, int scoring,
                   int32_t* out_id, double* out_val, int32_t* fv_node)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        fv_node[i] = -1;
        if (!(weight[i] > 0)) continue;
        fv_node[i] = node_id[i];
        int lo = 0, hi = m;                                    /* lower_bound */
        while (lo < hi) { const int mid = (lo + hi) / 2; if (out_id[mid] < word_id[i]) lo = mid + 1; else hi = mid; }
        if (lo < m && out_id[lo] == word_id[i]) {
            if (weighting == 0 || weighting == 1) out_val[lo] += weight[i];
        } else {
            memmove(out_id + lo + 1, out_id + lo, sizeof(int32_t) * (m - lo));
            memmove(out_val + lo + 1, out_val + lo, sizeof(double) * (m - lo));
            out_id[lo] = word_id[i]; out_val[lo] = weight[i]; m++;
        }
    }
    const int must = scoring != 5;                             /* DOT_PRODUCT does not normalise */
    const int l2 = scoring == 1;
    if ((weighting == 0 || weighting == 1) && m > 0 && !must) {
        const double nd = m;
        for (int i = 0; i < m; i++) out_val[i] /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (!l2) for (int i = 0; i < m; i++) norm += fabs(out_val[i]);
        else { for (int i = 0; i < m; i++) norm += out_val[i] * out_val[i]; norm = sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < m; i++) out_val[i] /= norm;
    }
    return m;
}

/* L1Scoring::score, ScoringObject.cpp:23-68 */
double orc_bow_score_l1(int n1, const int32_t* id1, const double* v1, int n2, const int32_t* id2, const double* v2)
{
    int a = 0, b = 0;
    double score = 0;
    while (a < n1 && b < n2) {
        if (id1[a] == id2[b]) { score += fabs(v1[a] - v2[b]) - fabs(v1[a]) - fabs(v2[b]); a++; b++; }
        else if (id1[a] < id2[b]) { while (a < n1 && id1[a] < id2[b]) a++; }
        else { while (b < n2 && id2[b] < id1[a]) b++; }
    }
    return -score / 2.0;
}

static int int_cmp(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cpp:957-988: index of the descriptor with the least median
 * distance to the others (first among equals) */
int orc_distinctive_descriptor(const uint8_t* desc, int n)
{
    if (n <= 0) return -1;
    int* d = (int*)malloc(sizeof(int) * n);
    int best_median = 0x7fffffff, best_idx = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) d[j] = i == j ? 0 : orc_descriptor_distance(desc + 32 * (size_t)i, desc + 32 * (size_t)j);
        qsort(d, n, sizeof(int), int_cmp);
        const int median = d[(int)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best_idx = i; }
    }
    free(d);
    return best_idx;
}
