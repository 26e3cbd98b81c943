// match_kernels.hip -- 256-bit Hamming matching on CDNA4.
//   k_hamming_bf      brute-force best / second-best per query (inner loop of ORBmatcher::SearchByBoW,
//                     cslam/src/ORBmatcher.cpp:224-245, with every feature in one vocabulary node)
//   k_hamming_ranges  distances of each side-1 feature to its vocabulary node's side-2 features
//                     (the DescriptorDistance calls of SearchByBoW; the greedy acceptance stays on the host)
// DescriptorDistance (:1653-1669) is popcount(a^b) over 8 dwords: v_xor_b32 + v_bcnt_u32_b32 with accumulate.
#include <hip/hip_runtime.h>
#include <cstdint>

#define BF_THREADS 512
#define BF_TILE 1024          // train descriptors staged per pass: 32 KiB of LDS

__device__ __forceinline__ int ham256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    int d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y); d += __popc(a0.z ^ b0.z); d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x); d += __popc(a1.y ^ b1.y); d += __popc(a1.z ^ b1.z); d += __popc(a1.w ^ b1.w);
    return d;
}

// One workgroup per pair; a lane owns two queries (tid and tid+512) in 16 VGPRs; the train
// descriptors are staged in LDS once per 1024 and read back as wave-uniform (broadcast) b128 loads.
// Running state per query: best = (dist << 16 | index) so that v_min_u32 keeps the LOWEST index among
// equal distances, which is what the reference's strict `<` scan in index order keeps; second =
// min(second, max(dist, best_dist_before)), identical to its if / else-if.
__global__ __launch_bounds__(BF_THREADS) void k_hamming_bf(
    const uint8_t* __restrict__ q, long long q_pair_bytes, const uint8_t* __restrict__ t, long long t_pair_bytes,
    int nq, int nt, const int* __restrict__ nq_n, const int* __restrict__ nt_n,
    int* __restrict__ best_idx, int* __restrict__ best_dist, int* __restrict__ second_dist)
{
    __shared__ uint4 tile[BF_TILE * 2];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nqp = nq_n ? min(max(nq_n[pair], 0), nq) : nq;
    const int ntp = nt_n ? min(max(nt_n[pair], 0), nt) : nt;
    const uint4* qp = reinterpret_cast<const uint4*>(q + (long long)pair * q_pair_bytes);
    const uint4* tp = reinterpret_cast<const uint4*>(t + (long long)pair * t_pair_bytes);
    for (int qbase = 0; qbase < nq; qbase += 2 * BF_THREADS) {
        const int qi0 = qbase + tid, qi1 = qbase + BF_THREADS + tid;
        const bool live0 = qi0 < nqp, live1 = qi1 < nqp;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, c0 = a0, c1 = a0;
        if (live0) { a0 = qp[2 * qi0]; a1 = qp[2 * qi0 + 1]; }
        if (live1) { c0 = qp[2 * qi1]; c1 = qp[2 * qi1 + 1]; }
        unsigned bestA = (256u << 16) | 0xFFFFu, bestC = bestA;
        int secA = 256, secC = 256;
        for (int tbase = 0; tbase < ntp; tbase += BF_TILE) {
            const int cnt = min(BF_TILE, ntp - tbase);
            __syncthreads();
            for (int i = tid; i < 2 * cnt; i += BF_THREADS) tile[i] = tp[2 * tbase + i];
            __syncthreads();
#pragma unroll 4
            for (int j = 0; j < cnt; j++) {
                const uint4 b0 = tile[2 * j], b1 = tile[2 * j + 1];
                const unsigned idx = (unsigned)(tbase + j);
                const int dA = ham256(a0, a1, b0, b1);
                const int dC = ham256(c0, c1, b0, b1);
                secA = min(secA, max(dA, (int)(bestA >> 16)));
                secC = min(secC, max(dC, (int)(bestC >> 16)));
                bestA = min(bestA, ((unsigned)dA << 16) | idx);
                bestC = min(bestC, ((unsigned)dC << 16) | idx);
            }
        }
        const long long ob = (long long)pair * nq;
        if (qi0 < nq) {
            const bool has = live0 && (bestA >> 16) < 256u;
            best_idx[ob + qi0] = has ? (int)(bestA & 0xFFFFu) : -1;
            best_dist[ob + qi0] = live0 ? (int)(bestA >> 16) : 256;
            second_dist[ob + qi0] = live0 ? secA : 256;
        }
        if (qi1 < nq) {
            const bool has = live1 && (bestC >> 16) < 256u;
            best_idx[ob + qi1] = has ? (int)(bestC & 0xFFFFu) : -1;
            best_dist[ob + qi1] = live1 ? (int)(bestC >> 16) : 256;
            second_dist[ob + qi1] = live1 ? secC : 256;
        }
    }
}

// Distances of side-1 feature i to the side-2 features order2[start[i] .. start[i]+len[i]) (its vocabulary
// node, ascending feature index), written to dist[off[i] ..].  One wave per feature: lanes stride the range.
__global__ __launch_bounds__(256) void k_hamming_ranges(
    const uint8_t* __restrict__ d1, const uint8_t* __restrict__ d2, const int* __restrict__ order2,
    const int* __restrict__ start, const int* __restrict__ len, const long long* __restrict__ off, int n1,
    unsigned short* __restrict__ dist)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n1) return;
    const int L = len[i];
    if (L <= 0) return;
    const uint4* a = reinterpret_cast<const uint4*>(d1) + 2 * (long long)i;
    const uint4 a0 = a[0], a1 = a[1];
    const int s = start[i];
    unsigned short* o = dist + off[i];
    for (int k = lane; k < L; k += 64) {
        const uint4* b = reinterpret_cast<const uint4*>(d2) + 2 * (long long)order2[s + k];
        o[k] = (unsigned short)ham256(a0, a1, b[0], b[1]);
    }
}

void match_launch_bf(hipStream_t s, const uint8_t* q, long long q_pair_bytes, const uint8_t* t, long long t_pair_bytes,
                     int nq, int nt, int n_pairs, const int* nq_n, const int* nt_n, int* bi, int* bd, int* sd)
{
    hipLaunchKernelGGL(k_hamming_bf, dim3(n_pairs), dim3(BF_THREADS), 0, s, q, q_pair_bytes, t, t_pair_bytes,
                       nq, nt, nq_n, nt_n, bi, bd, sd);
}
void match_launch_ranges(hipStream_t s, const uint8_t* d1, const uint8_t* d2, const int* order2, const int* start,
                         const int* len, const long long* off, int n1, unsigned short* dist)
{
    hipLaunchKernelGGL(k_hamming_ranges, dim3((n1 + 3) / 4), dim3(256), 0, s, d1, d2, order2, start, len, off, n1, dist);
}
