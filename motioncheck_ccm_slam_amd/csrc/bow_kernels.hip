// bow_kernels.hip -- vocabulary-tree kernels.
//   k_voc_transform   DBoW2::TemplatedVocabulary::transform(feature, word, weight, nid, levelsup)
//                     (cslam/thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1258): L levels of a k-way Hamming argmin
//   k_distinctive     MapPoint::ComputeDistinctiveDescriptors (cslam/src/MapPoint.cpp:957-988): least median distance
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ __forceinline__ int bow_ham256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    int d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y); d += __popc(a0.z ^ b0.z); d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x); d += __popc(a1.y ^ b1.y); d += __popc(a1.z ^ b1.z); d += __popc(a1.w ^ b1.w);
    return d;
}

// One thread per feature.  The children of a node are stored side by side (slot order = the reference's children
// order), so a level is one run of k 32-byte descriptors; the k loads of a level are independent.
// node_first[n] / node_count[n] index the slot arrays; slot_node[s] is the node id of a slot.
__global__ __launch_bounds__(256) void k_voc_transform(const uint8_t* __restrict__ feat, int n, const int* __restrict__ node_first,
                                                       const int* __restrict__ node_count, const uint4* __restrict__ slot_desc,
                                                       const int* __restrict__ slot_node, const int* __restrict__ node_word,
                                                       int nid_level, int max_depth, int* __restrict__ word_id,
                                                       int* __restrict__ leaf_node, int* __restrict__ node_id)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n) return;
    const uint4* fp = reinterpret_cast<const uint4*>(feat) + 2 * (long long)f;
    const uint4 a0 = fp[0], a1 = fp[1];
    int node = 0, nid = 0;
    // max_depth bounds the loop: every lane leaves it even on a malformed tree
    for (int level = 1; level <= max_depth; level++) {
        const int cnt = node_count[node];
        if (cnt == 0) break;
        const int s = node_first[node];
        int best = 0x7fffffff, bj = 0;
        for (int j = 0; j < cnt; j++) {
            const int d = bow_ham256(a0, a1, slot_desc[2 * (long long)(s + j)], slot_desc[2 * (long long)(s + j) + 1]);
            if (d < best) { best = d; bj = j; }                 // strict <: the first of equal children wins (:1241)
        }
        node = slot_node[s + bj];
        if (level == nid_level) nid = node;
    }
    word_id[f] = node_word[node];
    leaf_node[f] = node;
    node_id[f] = nid;
}

// One wave per map point; lane i owns row i of the distance matrix (rows beyond 64 in further rounds).  The median
// of a row = the value v with  #(d <= v) > k  and  #(d <= v-1) <= k,  k = (int)(0.5 * (N - 1)): found by bisection on
// v in [0, 256] with the distances recomputed each step (N is small, the descriptors stay in L1).
__global__ __launch_bounds__(256) void k_distinctive(const uint4* __restrict__ desc, const long long* __restrict__ first,
                                                     const int* __restrict__ count, int n_points, int* __restrict__ best_out)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n_points) return;
    const int N = count[p];
    if (N <= 0) { if (lane == 0) best_out[p] = -1; return; }
    const uint4* D = desc + 2 * first[p];
    const int k = (int)(0.5 * (N - 1));
    unsigned best_key = 0xFFFFFFFFu;                             // (median << 16 | row): min = least median, first row
    for (int i = lane; i < N; i += 64) {
        const uint4 a0 = D[2 * (long long)i], a1 = D[2 * (long long)i + 1];
        int lo = 0, hi = 256;                                    // smallest v with count(d <= v) > k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < N; j++) {
                const int d = j == i ? 0 : bow_ham256(a0, a1, D[2 * (long long)j], D[2 * (long long)j + 1]);
                c += d <= mid ? 1 : 0;
            }
            if (c > k) hi = mid; else lo = mid + 1;
        }
        const unsigned key = ((unsigned)lo << 16) | (unsigned)min(i, 0xFFFF);
        best_key = min(best_key, key);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best_key = min(best_key, (unsigned)__shfl_xor((int)best_key, d, 64));
    if (lane == 0) best_out[p] = (int)(best_key & 0xFFFFu);
}

void bow_launch_transform(hipStream_t s, const uint8_t* feat, int n, const int* node_first, const int* node_count, const uint8_t* slot_desc,
                          const int* slot_node, const int* node_word, int nid_level, int max_depth, int* word_id, int* leaf_node, int* node_id)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_voc_transform, dim3((n + 255) / 256), dim3(256), 0, s, feat, n, node_first, node_count,
                       reinterpret_cast<const uint4*>(slot_desc), slot_node, node_word, nid_level, max_depth, word_id, leaf_node, node_id);
}

void bow_launch_distinctive(hipStream_t s, const uint8_t* desc, const long long* first, const int* count, int n_points, int* best)
{
    if (n_points <= 0) return;
    hipLaunchKernelGGL(k_distinctive, dim3((n_points + 3) / 4), dim3(256), 0, s, reinterpret_cast<const uint4*>(desc), first, count, n_points, best);
}
