// bow_host.cpp -- C ABI of the vocabulary-tree path (include/ccm_hot.h, row F3 of SURVEY.md section 8f):
// DBoW2::TemplatedVocabulary::transform -> Frame::ComputeBoW, BowVector assembly / L1 score, and
// MapPoint::ComputeDistinctiveDescriptors.
#include "ccm_internal.h"
#include <algorithm>
#include <cmath>
#include <map>

void bow_launch_transform(hipStream_t, const uint8_t* feat, int n, const int* node_first, const int* node_count, const uint8_t* slot_desc,
                          const int* slot_node, const int* node_word, int nid_level, int max_depth, int* word_id, int* leaf_node, int* node_id);
void bow_launch_distinctive(hipStream_t, const uint8_t* desc, const long long* first, const int* count, int n_points, int* best);

struct ccm_vocabulary {
    ccm_ctx* ctx = nullptr;
    int k = 0, L = 0, n = 0, n_words = 0, depth = 0;
    std::vector<double> weight;                 // per node (host): looked up for the word a feature lands in
    DevBuf node_first, node_count, slot_desc, slot_node, node_word;
    DevBuf feat, word, leaf, nid;               // staging of ccm_voc_transform
    DevBuf dd, dfirst, dcount, dbest;           // staging of ccm_distinctive_descriptors
};

extern "C" {

int ccm_voc_create(ccm_ctx* c, int k, int L, int n_nodes, const int32_t* parent, const uint8_t* descriptors, const double* weights,
                   ccm_vocabulary** out)
{
    if (!c || !out) return CCM_E_ARG;
    *out = nullptr;
    if (n_nodes < 1 || !parent || !descriptors || !weights || L < 0) return ccm_fail(c, CCM_E_ARG, "bad vocabulary arguments");
    for (int i = 1; i < n_nodes; i++)
        if (parent[i] < 0 || parent[i] >= i) return ccm_fail(c, CCM_E_ARG, "vocabulary node %d: parent %d must precede it", i, parent[i]);
    CCM_HIP(c, hipSetDevice(c->device));
    // children side by side, in the order loadFromTextFile appends them (TemplatedVocabulary.h:1385-1392)
    std::vector<int> cnt(n_nodes, 0), first(n_nodes + 1, 0), fill(n_nodes, 0), slot_node(std::max(n_nodes - 1, 1)), word(n_nodes, 0), depth(n_nodes, 0);
    for (int i = 1; i < n_nodes; i++) cnt[parent[i]]++;
    for (int i = 0; i < n_nodes; i++) first[i + 1] = first[i] + cnt[i];
    std::vector<uint8_t> slot_desc((size_t)std::max(n_nodes - 1, 1) * 32);
    int max_depth = 0;
    for (int i = 1; i < n_nodes; i++) {
        const int s = first[parent[i]] + fill[parent[i]]++;
        slot_node[s] = i;
        std::memcpy(&slot_desc[(size_t)s * 32], descriptors + (size_t)i * 32, 32);
        depth[i] = depth[parent[i]] + 1;
        max_depth = std::max(max_depth, depth[i]);
    }
    int nw = 0;
    for (int i = 1; i < n_nodes; i++) if (cnt[i] == 0) word[i] = nw++;            // words numbered in node order (:1408-1414)
    ccm_vocabulary* v = new ccm_vocabulary();
    v->ctx = c; v->k = k; v->L = L; v->n = n_nodes; v->n_words = nw; v->depth = max_depth;
    v->weight.assign(weights, weights + n_nodes);
    struct Up { DevBuf* b; const void* src; size_t bytes; } ups[] = {
        { &v->node_first, first.data(), (size_t)n_nodes * 4 }, { &v->node_count, cnt.data(), (size_t)n_nodes * 4 },
        { &v->slot_desc, slot_desc.data(), slot_desc.size() }, { &v->slot_node, slot_node.data(), slot_node.size() * 4 },
        { &v->node_word, word.data(), (size_t)n_nodes * 4 } };
    for (auto& u : ups) {
        if (u.b->reserve(u.bytes) || hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            ccm_voc_destroy(v);
            return ccm_fail(c, CCM_E_DEVICE, "vocabulary upload failed");
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { ccm_voc_destroy(v); return ccm_fail(c, CCM_E_DEVICE, "vocabulary upload failed"); }
    *out = v;
    return CCM_OK;
}

void ccm_voc_destroy(ccm_vocabulary* v)
{
    if (!v) return;
    DevBuf* all[] = { &v->node_first, &v->node_count, &v->slot_desc, &v->slot_node, &v->node_word, &v->feat, &v->word, &v->leaf, &v->nid,
                      &v->dd, &v->dfirst, &v->dcount, &v->dbest };
    for (DevBuf* b : all) b->release();
    delete v;
}

int ccm_voc_words(const ccm_vocabulary* v) { return v ? v->n_words : 0; }

// features_dev: n descriptors resident on the context's device (e.g. ccm_orb_result_dev); results on the host
int ccm_voc_transform_dev(ccm_vocabulary* v, const uint8_t* features_dev, int n, int levelsup, int32_t* word_id, double* weight, int32_t* node_id)
{
    if (!v) return CCM_E_ARG;
    ccm_ctx* c = v->ctx;
    if (n < 0 || (n > 0 && (!features_dev || !word_id || !weight || !node_id))) return ccm_fail(c, CCM_E_ARG, "bad transform arguments");
    if (n == 0) return CCM_OK;
    if (v->n_words == 0) {                                                        // empty(): transform() returns nothing (:1133)
        for (int i = 0; i < n; i++) { word_id[i] = 0; weight[i] = 0; node_id[i] = 0; }
        return CCM_OK;
    }
    CCM_HIP(c, hipSetDevice(c->device));
    CCM_RESERVE(c, v->word, (size_t)n * 4); CCM_RESERVE(c, v->leaf, (size_t)n * 4); CCM_RESERVE(c, v->nid, (size_t)n * 4);
    bow_launch_transform(c->stream, features_dev, n, v->node_first.as<int>(), v->node_count.as<int>(), v->slot_desc.as<uint8_t>(),
                         v->slot_node.as<int>(), v->node_word.as<int>(), v->L - levelsup, v->depth, v->word.as<int>(), v->leaf.as<int>(), v->nid.as<int>());
    CCM_HIP(c, hipGetLastError());
    std::vector<int32_t> leaf(n);
    CCM_HIP(c, hipMemcpyAsync(word_id, v->word.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipMemcpyAsync(leaf.data(), v->leaf.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipMemcpyAsync(node_id, v->nid.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) weight[i] = v->weight[leaf[i]];                   // m_nodes[final_id].weight (:1257)
    return CCM_OK;
}

int ccm_voc_transform(ccm_vocabulary* v, const uint8_t* features, int n, int levelsup, int32_t* word_id, double* weight, int32_t* node_id)
{
    if (!v) return CCM_E_ARG;
    ccm_ctx* c = v->ctx;
    if (n < 0 || (n > 0 && !features)) return ccm_fail(c, CCM_E_ARG, "bad transform arguments");
    if (n == 0) return CCM_OK;
    CCM_HIP(c, hipSetDevice(c->device));
    CCM_RESERVE(c, v->feat, (size_t)n * 32);
    CCM_HIP(c, hipMemcpyAsync(v->feat.p, features, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    return ccm_voc_transform_dev(v, v->feat.as<uint8_t>(), n, levelsup, word_id, weight, node_id);
}

// transform(features, BowVector&, FeatureVector&, levelsup), TemplatedVocabulary.h:1125-1193: the map arithmetic in the
// reference's order (addWeight in feature order, normalisation sums in key order)
int ccm_bow_vector(int n, const int32_t* word_id, const double* weight, const int32_t* node_id, int weighting, int scoring,
                   int32_t* out_id, double* out_val, int32_t* fv_node)
{
    if (n < 0 || (n > 0 && (!word_id || !weight || !node_id || !out_id || !out_val || !fv_node))) return CCM_E_ARG;
    std::map<int32_t, double> v;
    const bool tf = weighting == 0 || weighting == 1;                             // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    for (int i = 0; i < n; i++) {
        fv_node[i] = -1;
        if (!(weight[i] > 0)) continue;                                          // stopped word
        auto it = v.lower_bound(word_id[i]);
        if (it != v.end() && it->first == word_id[i]) { if (tf) it->second += weight[i]; }
        else v.insert(it, { word_id[i], weight[i] });
        fv_node[i] = node_id[i];
    }
    const bool must = scoring != 5;                                               // DotProductScoring (ScoringObject.h:89)
    if (tf && !v.empty() && !must) {
        const double nd = (double)v.size();
        for (auto& kv : v) kv.second /= nd;
    }
    if (must) {                                                                   // BowVector::normalize, BowVector.cpp:64-87
        double norm = 0.0;
        if (scoring != 1) for (auto& kv : v) norm += std::fabs(kv.second);
        else { for (auto& kv : v) norm += kv.second * kv.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
    }
    int m = 0;
    for (auto& kv : v) { out_id[m] = kv.first; out_val[m] = kv.second; m++; }
    return m;
}

// L1Scoring::score, ScoringObject.cpp:23-68
double ccm_bow_score_l1(int n1, const int32_t* id1, const double* v1, int n2, const int32_t* id2, const double* v2)
{
    int a = 0, b = 0;
    double score = 0;
    while (a < n1 && b < n2) {
        if (id1[a] == id2[b]) { score += std::fabs(v1[a] - v2[b]) - std::fabs(v1[a]) - std::fabs(v2[b]); a++; b++; }
        else if (id1[a] < id2[b]) a = (int)(std::lower_bound(id1 + a, id1 + n1, id2[b]) - id1);
        else b = (int)(std::lower_bound(id2 + b, id2 + n2, id1[a]) - id2);
    }
    return -score / 2.0;
}

// MapPoint::ComputeDistinctiveDescriptors for many map points at once (src/MapPoint.cpp:929-994)
int ccm_distinctive_descriptors(ccm_vocabulary* v, const uint8_t* desc, const int64_t* first, const int32_t* count, int n_points, int32_t* best)
{
    if (!v) return CCM_E_ARG;
    ccm_ctx* c = v->ctx;
    if (n_points < 0 || (n_points > 0 && (!first || !count || !best))) return ccm_fail(c, CCM_E_ARG, "bad ComputeDistinctiveDescriptors arguments");
    if (n_points == 0) return CCM_OK;
    long long total = 0;
    for (int p = 0; p < n_points; p++) {
        if (count[p] < 0 || count[p] > 65535 || first[p] < 0) return ccm_fail(c, CCM_E_ARG, "map point %d: bad descriptor range", p);
        total = std::max<long long>(total, first[p] + count[p]);
    }
    if (total > 0 && !desc) return ccm_fail(c, CCM_E_ARG, "bad ComputeDistinctiveDescriptors arguments");
    CCM_HIP(c, hipSetDevice(c->device));
    CCM_RESERVE(c, v->dd, (size_t)std::max<long long>(total, 1) * 32); CCM_RESERVE(c, v->dfirst, (size_t)n_points * 8);
    CCM_RESERVE(c, v->dcount, (size_t)n_points * 4); CCM_RESERVE(c, v->dbest, (size_t)n_points * 4);
    if (total) CCM_HIP(c, hipMemcpyAsync(v->dd.p, desc, (size_t)total * 32, hipMemcpyHostToDevice, c->stream));
    CCM_HIP(c, hipMemcpyAsync(v->dfirst.p, first, (size_t)n_points * 8, hipMemcpyHostToDevice, c->stream));
    CCM_HIP(c, hipMemcpyAsync(v->dcount.p, count, (size_t)n_points * 4, hipMemcpyHostToDevice, c->stream));
    bow_launch_distinctive(c->stream, v->dd.as<uint8_t>(), v->dfirst.as<long long>(), v->dcount.as<int>(), n_points, v->dbest.as<int>());
    CCM_HIP(c, hipGetLastError());
    CCM_HIP(c, hipMemcpyAsync(best, v->dbest.p, (size_t)n_points * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    return CCM_OK;
}

}  // extern "C"
