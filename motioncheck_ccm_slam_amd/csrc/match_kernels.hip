// match_kernels.hip -- 256-bit Hamming matching on CDNA4.
//   k_hamming_mfma    brute-force best / second-best per query on the matrix cores (<= 2048 train rows per pair)
//   k_hamming_bf      the same on the vector ALU (any size): inner loop of ORBmatcher::SearchByBoW,
//                     cslam/src/ORBmatcher.cpp:224-245, with every feature in one vocabulary node
//   k_hamming_ranges  distances of each side-1 feature to its vocabulary node's side-2 features
//                     (the DescriptorDistance calls of SearchByBoW; the greedy acceptance stays on the host)
// DescriptorDistance (:1653-1669) is popcount(a^b) over 8 dwords: v_xor_b32 + v_bcnt_u32_b32 with accumulate.
#include <hip/hip_runtime.h>
#include <cstdint>

#define BF_TILE 1024          // train descriptors staged per pass: 32 KiB of LDS

__device__ __forceinline__ int ham256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    int d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y); d += __popc(a0.z ^ b0.z); d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x); d += __popc(a1.y ^ b1.y); d += __popc(a1.z ^ b1.z); d += __popc(a1.w ^ b1.w);
    return d;
}

// One workgroup per (pair, train split); a lane owns QPL queries (8 VGPRs each); the train
// descriptors are staged in LDS once per 1024 and read back as wave-uniform (broadcast) b128 loads.
// Running state per query: best = (dist << 16 | index) so that v_min_u32 keeps the LOWEST index among
// equal distances, which is what the reference's strict `<` scan in index order keeps; second =
// min(second, max(dist, best_dist_before)), identical to its if / else-if.
// With n_split > 1 the train rows of a pair are divided over n_split workgroups that write partial
// (best, second) to scratch; k_hamming_merge combines them exactly (see there).
template <int THREADS, int QPL>
__global__ __launch_bounds__(THREADS) void k_hamming_bf(
    const uint8_t* __restrict__ q, long long q_pair_bytes, const uint8_t* __restrict__ t, long long t_pair_bytes,
    int nq, int nt, const int* __restrict__ nq_n, const int* __restrict__ nt_n, int n_split,
    unsigned* __restrict__ part_best, int* __restrict__ part_second,
    int* __restrict__ best_idx, int* __restrict__ best_dist, int* __restrict__ second_dist)
{
    __shared__ uint4 tile[BF_TILE * 2];
    const int pair = blockIdx.x / n_split, split = blockIdx.x - pair * n_split, tid = threadIdx.x;
    const int nqp = nq_n ? min(max(nq_n[pair], 0), nq) : nq;
    const int ntp = nt_n ? min(max(nt_n[pair], 0), nt) : nt;
    // this workgroup's train rows [t_lo, t_hi): equal chunks rounded up to the wave-friendly 64
    const int chunk = ((ntp + n_split - 1) / n_split + 63) & ~63;
    const int t_lo = min(split * chunk, ntp), t_hi = min(t_lo + chunk, ntp);
    const uint4* qp = reinterpret_cast<const uint4*>(q + (long long)pair * q_pair_bytes);
    const uint4* tp = reinterpret_cast<const uint4*>(t + (long long)pair * t_pair_bytes);
    // rows past the live count only get the "no match" defaults (no train scan for them)
    const int q_done = ((nqp + QPL * THREADS - 1) / (QPL * THREADS)) * (QPL * THREADS);
    if (n_split == 1 || split == 0)
        for (int qi = q_done + tid; qi < nq; qi += THREADS) {
            if (n_split > 1) {
                for (int sp = 0; sp < n_split; sp++) {
                    const long long o = ((long long)pair * n_split + sp) * nq + qi;
                    part_best[o] = (256u << 16) | 0xFFFFu; part_second[o] = 256;
                }
            } else {
                const long long o = (long long)pair * nq + qi;
                best_idx[o] = -1; best_dist[o] = 256; second_dist[o] = 256;
            }
        }
    for (int qbase = 0; qbase < nqp; qbase += QPL * THREADS) {
        uint4 a0[QPL], a1[QPL];
        unsigned best[QPL]; int sec[QPL];
#pragma unroll
        for (int k = 0; k < QPL; k++) {
            const int qi = qbase + k * THREADS + tid;
            a0[k] = make_uint4(0, 0, 0, 0); a1[k] = a0[k];
            if (qi < nqp) { a0[k] = qp[2 * qi]; a1[k] = qp[2 * qi + 1]; }
            best[k] = (256u << 16) | 0xFFFFu; sec[k] = 256;
        }
        for (int tbase = t_lo; tbase < t_hi; tbase += BF_TILE) {
            const int cnt = min(BF_TILE, t_hi - tbase);
            __syncthreads();
            for (int i = tid; i < 2 * cnt; i += THREADS) tile[i] = tp[2 * tbase + i];
            __syncthreads();
#pragma unroll 2
            for (int j = 0; j < cnt; j++) {
                const uint4 b0 = tile[2 * j], b1 = tile[2 * j + 1];
                const unsigned idx = (unsigned)(tbase + j);
#pragma unroll
                for (int k = 0; k < QPL; k++) {
                    const int d = ham256(a0[k], a1[k], b0, b1);
                    sec[k] = min(sec[k], max(d, (int)(best[k] >> 16)));
                    best[k] = min(best[k], ((unsigned)d << 16) | idx);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < QPL; k++) {
            const int qi = qbase + k * THREADS + tid;
            if (qi >= nq) continue;
            if (n_split > 1) {
                const long long o = ((long long)pair * n_split + split) * nq + qi;
                part_best[o] = best[k]; part_second[o] = sec[k];
            } else {
                const bool live = qi < nqp;
                const long long o = (long long)pair * nq + qi;
                best_idx[o] = (live && (best[k] >> 16) < 256u) ? (int)(best[k] & 0xFFFFu) : -1;
                best_dist[o] = live ? (int)(best[k] >> 16) : 256;
                second_dist[o] = live ? sec[k] : 256;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_hamming_mfma: the same best / second-best search on the matrix cores.
//   hamming(a, b) = popc(a) + popc(b) - 2 <a, b>   with <a, b> the dot product of the 0/1 bit vectors,
// so a 32 x 32 block of distances is eight v_mfma_i32_32x32x32_i8 (K = 256 bits) on operands whose bits are
// expanded to bytes: train rows to 0/32, query columns to 0/-128, so that a common bit adds -4096 = -(2 << 11).
// Layout: A operand = 32 train rows, B operand = 32 query columns.  C/D puts column j = lane & 31 on the lane and
// rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) in its 16 registers, i.e. a lane sees ONE query and 16 trains per
// tile, so the running (best, second) of that query are two registers per lane.  The accumulator of the first k-step
// is preloaded (C operand) with the row word ((popc(train) + HM_BIAS) << 11) | train index, so the matrix cores
// deliver the finished signed key
//   ((popc(train) - 2 <a, b> + HM_BIAS) << 11) | train index
// -- popc(query) is constant per lane and added at the end -- and the fold is  second = med3(best, second, key);
// best = min(best, key): two VALU instructions per distance instead of ~18, the rest is MFMA.
// A/B fragments: lane (h = lane >> 5, r = lane & 31) supplies the 16 bytes of k-step s from bits [32 s + 16 h,
// 32 s + 16 h + 16) of its row / column; A and B use the same k assignment, which is all the dot product needs.
// One workgroup = 4 waves x 64 queries; the train fragments of a 32-train tile (8 KiB) are expanded once per
// workgroup into LDS (double-buffered) and read back as ds_read_b128.
typedef int hm_v4i __attribute__((ext_vector_type(4)));
typedef int hm_v16i __attribute__((ext_vector_type(16)));
#ifndef HM_CB
#define HM_CB 2                  // 32-query column blocks per wave; measured (CB, waves) at 10 k pairs: (2,4) 2.55 ms,
                                 // (1,8) 2.72, (1,4) 3.09, (2,8) 3.13, (4,4) 8.94 (spills)
#endif
#ifndef HM_WAVES
#define HM_WAVES 4
#endif
#ifndef HM_TT
#define HM_TT 1                  // 32-train tiles staged per barrier; measured at 10 k pairs: 1 -> 2.34 ms, 2 -> 2.32, 4 (2 WG/CU) -> 2.77
#endif
#ifndef HM_WG_PER_CU
#define HM_WG_PER_CU 4           // measured at 10 k pairs: 4 -> 2.32 ms, 3 -> 2.48 ms
#endif
#define HM_QW (32 * HM_CB)       // queries per wave
#define HM_TPB (64 * HM_WAVES)
#define HM_SENT 0x3FFFFFFF       // key of "no train": larger than every real key
#define HM_MAX_NT 2048           // train rows whose row words fit the LDS table (the launcher falls back beyond)

// median of three keys.  The compiler has no integer med3 pattern for variables, but every key is a positive normal
// float when read as one -- (hamming - popc(query) + HM_BIAS) << 11 | index with the distance field in [4096, 4864],
// HM_SENT = 0x3FFFFFFF -- and positive floats order like their bit patterns, so v_med3_f32 returns the same register.
#define HM_BIAS 4352
#define HM_IDX_BITS 11           // train index field of a key (HM_MAX_NT = 2048 rows)
__device__ __forceinline__ int hm_med3(int a, int b, int c)
{
    return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(a), __int_as_float(b), __int_as_float(c)));
}

__device__ __forceinline__ hm_v4i hm_expand16(unsigned hw, int shift)
{
    // 4 bits -> 4 bytes: n * 0x204081 puts bit i at bit 8 i (the 24-bit multiply is exact, n < 16); the 0/1 bytes are
    // then shifted to 0/32 (trains, shift 5) or 0/0x80 = -128 (queries, shift 7): a common bit contributes
    // 32 * -128 = -2 << HM_IDX_BITS to the accumulator, i.e. -2 in the distance field of the key
    // (the shift is applied to the nibble, not to the product: v_mul_u32_u24 stays a full-rate instruction)
    hm_v4i v;
    const unsigned hs = hw << shift, nm = 15u << shift, bm = 0x01010101u << shift;
    v.x = (int)(__umul24(hs & nm, 0x204081u) & bm); v.y = (int)(__umul24((hs >> 4) & nm, 0x204081u) & bm);
    v.z = (int)(__umul24((hs >> 8) & nm, 0x204081u) & bm); v.w = (int)(__umul24((hs >> 12) & nm, 0x204081u) & bm);
    return v;
}

// 4 workgroups per CU (<= 128 registers per lane): 255 pairs x 4 live query blocks = 1020 workgroups fit in ONE round
__global__ __launch_bounds__(HM_TPB, HM_WG_PER_CU * HM_TPB / 256) void k_hamming_mfma(
    const uint8_t* __restrict__ q, long long q_pair_bytes, const uint8_t* __restrict__ t, long long t_pair_bytes,
    int nq, int nt, const int* __restrict__ nq_n, const int* __restrict__ nt_n, int q_blocks,
    int* __restrict__ best_idx, int* __restrict__ best_dist, int* __restrict__ second_dist)
{
    __shared__ hm_v4i frag[2][HM_TT * 8][64];  // [buffer][sub-tile, k-step][lane]: 16 expanded bytes
    __shared__ __attribute__((aligned(16))) int wall[HM_MAX_NT];   // per train: (popc + HM_BIAS) << 11 | index, HM_SENT past the live count
    const int pair = blockIdx.x / q_blocks, qblk = blockIdx.x - pair * q_blocks;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5, r = lane & 31;
    const int nqp = nq_n ? min(max(nq_n[pair], 0), nq) : nq;
    const int ntp = nt_n ? min(max(nt_n[pair], 0), nt) : nt;
    const unsigned* qp = reinterpret_cast<const unsigned*>(q + (long long)pair * q_pair_bytes);
    const unsigned* tp = reinterpret_cast<const unsigned*>(t + (long long)pair * t_pair_bytes);
    const int q0 = qblk * (HM_QW * HM_WAVES) + wv * HM_QW;              // this wave's first query
    if (qblk * (HM_QW * HM_WAVES) >= nqp) {                              // whole block past the live queries: defaults only
        for (int i = tid; i < HM_QW * HM_WAVES; i += HM_TPB) {
            const int qi = qblk * (HM_QW * HM_WAVES) + i;
            if (qi < nq) { const long long o = (long long)pair * nq + qi; best_idx[o] = -1; best_dist[o] = 256; second_dist[o] = 256; }
        }
        return;
    }

    // ---- B fragments of the wave's 64 queries (0 / -1 bytes) and their popcounts, kept in registers
    hm_v4i bq[HM_CB][8];
    int pa[HM_CB];
#pragma unroll
    for (int cb = 0; cb < HM_CB; cb++) {
        const int qi = q0 + 32 * cb + r;
        unsigned d[8];
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) { d[s] = qi < nqp ? qp[8 * (long long)qi + s] : 0u; pc += __popc(d[s]); }
        pa[cb] = pc;
#pragma unroll
        for (int s = 0; s < 8; s++) bq[cb][s] = hm_expand16((d[s] >> (16 * h)) & 0xFFFFu, 7);
    }
    int m1[HM_CB], m2[HM_CB];
#pragma unroll
    for (int cb = 0; cb < HM_CB; cb++) { m1[cb] = HM_SENT; m2[cb] = HM_SENT; }

    // ---- staging of one 32-train tile: thread p expands dword s = p >> 5 of train r = p & 31 into the two
    //      lane-half fragments of k-step s; threads 0..31 also make the row words.  The raw words are fetched one
    //      tile ahead (issued before the MFMAs of the current tile) so that no wave waits on a global load.
    // fragment f of a tile (512 of them): lane half f >> 8, k-step (f >> 5) & 7, train f & 31; thread t owns
    // fragments t, t + HM_TPB, ... (with 256 threads: both halves of one dword)
    // HM_TT tiles are staged per barrier ("group")
    constexpr int HM_FPT = (512 * HM_TT + HM_TPB - 1) / HM_TPB;
    struct Raw { unsigned v[HM_FPT]; };
    auto fetch = [&](int group) {
        Raw x;
#pragma unroll
        for (int j = 0; j < HM_FPT; j++) {
            const int fi = tid + j * HM_TPB, sub = fi >> 9, fw = fi & 511;
            const int tr = (group * HM_TT + sub) * 32 + (fw & 31);
            x.v[j] = (fi < 512 * HM_TT && tr < ntp) ? tp[8 * (long long)tr + ((fw >> 5) & 7)] : 0u;
        }
        return x;
    };
    auto stage = [&](const Raw& x, int buf) {
#pragma unroll
        for (int j = 0; j < HM_FPT; j++) {
            const int fi = tid + j * HM_TPB, sub = fi >> 9, fw = fi & 511;
            if (fi < 512 * HM_TT) frag[buf][8 * sub + ((fw >> 5) & 7)][32 * (fw >> 8) + (fw & 31)] = hm_expand16((x.v[j] >> (16 * (fw >> 8))) & 0xFFFFu, 5);
        }
    };
    // row words of all trains, once per workgroup
    for (int tr = tid; tr < ((ntp + 31) & ~31); tr += HM_TPB) {
        int w = HM_SENT;
        if (tr < ntp) {
            const uint4* d = reinterpret_cast<const uint4*>(tp + 8 * (long long)tr);
            const uint4 d0 = d[0], d1 = d[1];
            const int pb = __popc(d0.x) + __popc(d0.y) + __popc(d0.z) + __popc(d0.w) + __popc(d1.x) + __popc(d1.y) + __popc(d1.z) + __popc(d1.w);
            w = ((pb + HM_BIAS) << HM_IDX_BITS) | tr;
        }
        wall[tr] = w;
    }
    const int ntiles = (ntp + 31) >> 5;
    const bool wave_live = q0 < nqp;                                    // waves past the live queries only help staging
    // the sixteen matrix instructions of one tile into `acc`.  The accumulators start from the row words of the tile
    // (C operand of the first k-step: row = train, the same word in every query column), so what comes out IS the key
    //   ((popc(train) + HM_BIAS) << 11 | train) - (2 << 11) <train, query>.
    auto mma = [&](int tile, int buf, int sub, hm_v16i (&acc)[HM_CB]) {
        hm_v16i wc;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const hm_v4i w4 = *reinterpret_cast<const hm_v4i*>(&wall[32 * tile + 8 * g + 4 * h]);
            wc[4 * g] = w4.x; wc[4 * g + 1] = w4.y; wc[4 * g + 2] = w4.z; wc[4 * g + 3] = w4.w;
        }
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const hm_v4i a = frag[buf][8 * sub + s][lane];
#pragma unroll
            for (int cb = 0; cb < HM_CB; cb++) acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[cb][s], s == 0 ? wc : acc[cb], 0, 0, 0);
        }
    };
    // (best, second) of the lane's query folded over the 16 keys it sees in `acc`: v_med3_f32 + v_min_i32 per key
    auto fold = [&](const hm_v16i (&acc)[HM_CB]) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int cb = 0; cb < HM_CB; cb++) {
                const int kx = acc[cb][i];
                m2[cb] = hm_med3(m1[cb], m2[cb], kx); m1[cb] = min(m1[cb], kx);
            }
    };
    const int ngroups = (ntiles + HM_TT - 1) / HM_TT;
    Raw nxt = fetch(0);
    if (ngroups > 0) stage(nxt, 0);
    nxt = fetch(1);
    __syncthreads();
    for (int grp = 0; grp < ngroups; grp++) {
        const int buf = grp & 1;
        if (grp + 1 < ngroups) stage(nxt, buf ^ 1);
        if (grp + 2 < ngroups) nxt = fetch(grp + 2);
        if (wave_live) {
#pragma unroll
            for (int sub = 0; sub < HM_TT; sub++) {
                const int tile = grp * HM_TT + sub;
                if (tile < ntiles) {
                    hm_v16i acc[HM_CB];
                    mma(tile, buf, sub, acc);
                    fold(acc);
                }
            }
        }
        __syncthreads();
    }
    // ---- the two lane halves saw disjoint trains of the same query: merge, add popc(query), write
#pragma unroll
    for (int cb = 0; cb < HM_CB; cb++) {
        const int o1 = __shfl_xor(m1[cb], 32, 64), o2 = __shfl_xor(m2[cb], 32, 64);
        const int b1 = min(m1[cb], o1);
        const int b2 = min(max(m1[cb], o1), min(m2[cb], o2));
        const int qi = q0 + 32 * cb + r;
        if (h == 0 && qi < nq) {
            const long long o = (long long)pair * nq + qi;
            const bool live = qi < nqp;
            const bool has1 = live && b1 < HM_SENT, has2 = live && b2 < HM_SENT;
            const int bd = has1 ? (b1 >> HM_IDX_BITS) - HM_BIAS + pa[cb] : 256;
            best_idx[o] = bd < 256 ? (b1 & ((1 << HM_IDX_BITS) - 1)) : -1;           // the reference starts at 256 and compares with <
            best_dist[o] = bd;
            second_dist[o] = has2 ? (b2 >> HM_IDX_BITS) - HM_BIAS + pa[cb] : 256;
        }
    }
}

// Exact merge of per-split partial results over disjoint, ascending index ranges: the overall best is the
// lexicographic minimum of (dist, index); the overall second-smallest distance is
// min(max(best_a, best_b), second_a, second_b) folded over the splits in index order.
__global__ __launch_bounds__(256) void k_hamming_merge(const unsigned* __restrict__ part_best, const int* __restrict__ part_second,
                                                       int n_pairs, int nq, int n_split, const int* __restrict__ nq_n,
                                                       int* __restrict__ best_idx, int* __restrict__ best_dist, int* __restrict__ second_dist)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)n_pairs * nq) return;
    const int pair = (int)(i / nq), qi = (int)(i - (long long)pair * nq);
    const int nqp = nq_n ? min(max(nq_n[pair], 0), nq) : nq;
    unsigned best = (256u << 16) | 0xFFFFu; int sec = 256;
    for (int s = 0; s < n_split; s++) {
        const long long o = ((long long)pair * n_split + s) * nq + qi;
        const unsigned b = part_best[o]; const int sc = part_second[o];
        sec = min(min(sec, sc), max((int)(best >> 16), (int)(b >> 16)));
        best = min(best, b);
    }
    const bool live = qi < nqp;
    best_idx[i] = (live && (best >> 16) < 256u) ? (int)(best & 0xFFFFu) : -1;
    best_dist[i] = live ? (int)(best >> 16) : 256;
    second_dist[i] = live ? sec : 256;
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

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW on the device (cslam/src/ORBmatcher.cpp:178-306 and :565-698).  A side-2 feature belongs to
// exactly one vocabulary node, so the reference's sequential "already matched" bookkeeping only couples features of the
// SAME node: one wave per node group walks its side-1 features in order (the reference's order), lanes scan the node's
// side-2 features, and the wave keeps (best, second) exactly as the if / else-if of :233-245 does -- best = first
// minimal distance in scan order, second = second smallest including duplicates.  Accepted matches mark their side-2
// feature taken, vote in the rotation histogram (integer counts), and k_bow_filter applies ComputeThreeMaxima
// (:1607-1648) and drops the matches of the other bins (:271-289).
struct BowDev {
    int n_groups, n1;
    const int* ga; const int* gae; const int* gb; const int* gbe;     // per group: ranges in ord1 / ord2
    const int* ord1; const int* ord2;
    const uint8_t* d1b; const uint8_t* d2b;                           // descriptors, 16-byte aligned
    const uint8_t* valid1; const uint8_t* valid2;                     // valid2 may be null
    const float* angle1; const float* angle2;
    uint8_t* taken; int* match12; int* bin_of; int* hist;             // hist[30] + [30] = kept count
    float nnratio; int th, strict_th, check_ori;
};
__global__ __launch_bounds__(256) void k_bow_greedy(BowDev B)
{
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grp >= B.n_groups) return;
    const int a0 = B.ga[grp], a1 = B.gae[grp], b0 = B.gb[grp], b1 = B.gbe[grp];
    for (int a = a0; a < a1; a++) {
        const int i1 = B.ord1[a];
        if (!B.valid1[i1]) continue;                                 // wave-uniform
        const uint4* D1 = reinterpret_cast<const uint4*>(B.d1b); const uint4* D2 = reinterpret_cast<const uint4*>(B.d2b);
        const uint4 q0 = D1[2 * (long long)i1], q1 = D1[2 * (long long)i1 + 1];
        int bd1 = 256, bd2 = 256, bi = -1;
        for (int base = b0; base < b1; base += 64) {
            const int j = base + lane;
            int d = 1 << 20, i2 = -1;                                // lanes without a candidate never win
            if (j < b1) {
                i2 = B.ord2[j];
                if (!B.taken[i2] && (!B.valid2 || B.valid2[i2])) d = ham256(q0, q1, D2[2 * (long long)i2], D2[2 * (long long)i2 + 1]);
            }
            // chunk best = first minimal distance in lane (= scan) order; chunk second = smallest of the other lanes
            unsigned key = ((unsigned)d << 6) | (unsigned)lane;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, s, 64));
            const int cb = (int)(key >> 6), cl = (int)(key & 63u);
            int o = lane == cl ? (1 << 20) : d;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) o = min(o, __shfl_xor(o, s, 64));
            const int ci2 = __shfl(i2, cl, 64);
            // fold the chunk into the running pair the way the sequential if / else-if would
            if (cb < bd1) { bd2 = min(bd1, o); bd1 = cb; bi = ci2; }
            else bd2 = min(bd2, cb);
        }
        const bool pass = B.strict_th ? (bd1 < B.th) : (bd1 <= B.th);
        if (bi >= 0 && pass && (float)bd1 < B.nnratio * (float)bd2) {
            if (lane == 0) {
                B.match12[i1] = bi;
                B.taken[bi] = 1;
                if (B.check_ori) {
                    float r = B.angle1[i1] - B.angle2[bi];
                    if (r < 0.0) r += 360.0f;
                    int bin = (int)roundf(r * (1.0f / 30));          // 1/HISTO_LENGTH as in the reference (:191, :260)
                    if (bin == 30) bin = 0;
                    B.bin_of[i1] = bin;
                    atomicAdd(&B.hist[bin], 1);
                }
            }
            __threadfence_block();                                   // the taken flag is read by this wave's next feature
        }
    }
}
// one workgroup: ComputeThreeMaxima on the 30 bin counts, then every match outside the kept bins is dropped
__global__ __launch_bounds__(256) void k_bow_filter(BowDev B)
{
    __shared__ int keep[3];
    __shared__ int cnt;
    if (threadIdx.x == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        if (B.check_ori) {
            for (int i = 0; i < 30; i++) {
                const int s = B.hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) ind3 = -1;
        }
        keep[0] = ind1; keep[1] = ind2; keep[2] = ind3; cnt = 0;
    }
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < B.n1; i += 256) {
        if (B.match12[i] < 0) continue;
        if (B.check_ori) {
            const int b = B.bin_of[i];
            if (b != keep[0] && b != keep[1] && b != keep[2]) { B.match12[i] = -1; continue; }
        }
        mine++;
    }
    atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) B.hist[30] = cnt;
}
void match_launch_bow(hipStream_t s, const BowDev& B)
{
    if (B.n_groups > 0) hipLaunchKernelGGL(k_bow_greedy, dim3((B.n_groups + 3) / 4), dim3(256), 0, s, B);
    hipLaunchKernelGGL(k_bow_filter, dim3(1), dim3(256), 0, s, B);
}

// Frame::GetFeaturesInArea (src/Frame.cpp:200-253) + the DescriptorDistance calls of the windowed matchers
// (SearchByProjection / Fuse / SearchBySim3 ..., cslam/src/ORBmatcher.cpp:71-148 and following): one wave per
// query walks the grid cells of its window in the reference's order (ix outer, iy inner, cell content in
// feature-index order), applies the level and |dx|,|dy| < r tests in float exactly as written there, and emits
// (feature index, Hamming distance) in that order through ballot compaction.
struct WinGrid {
    int n, cols, rows;
    float min_x, min_y, inv_w, inv_h;
    const float* kx; const float* ky; const int* oct; const uint8_t* desc;
    const int* cell_first; const int* cell_items;
};
// BATCH: the queries of many keyframes in one launch, query q against grids[q_kf[q]]; the candidate indices are the keyframe's own.
template <bool BATCH>
__global__ __launch_bounds__(256) void k_window_candidates(WinGrid G1, const WinGrid* __restrict__ grids, const int* __restrict__ q_kf,
                                                           int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                           const float* __restrict__ qr, const int* __restrict__ min_level,
                                                           const int* __restrict__ max_level, const uint8_t* __restrict__ qdesc, int cap,
                                                           int* __restrict__ cand_idx, int* __restrict__ cand_dist, int* __restrict__ cand_n)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const WinGrid G = BATCH ? grids[q_kf[q]] : G1;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minL = min_level[q], maxL = max_level[q];
    int n = 0;
    if (r >= 0.f) {                                        // r < 0 marks a query that is not searched
        const int nMinCellX = max(0, (int)floorf((x - G.min_x - r) * G.inv_w));
        const int nMaxCellX = min(G.cols - 1, (int)ceilf((x - G.min_x + r) * G.inv_w));
        const int nMinCellY = max(0, (int)floorf((y - G.min_y - r) * G.inv_h));
        const int nMaxCellY = min(G.rows - 1, (int)ceilf((y - G.min_y + r) * G.inv_h));
        if (nMinCellX < G.cols && nMaxCellX >= 0 && nMinCellY < G.rows && nMaxCellY >= 0) {
            const bool check = (minL > 0) || (maxL >= 0);
            const uint4* qd = reinterpret_cast<const uint4*>(qdesc) + 2 * (long long)q;
            const uint4 a0 = qd[0], a1 = qd[1];
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                // the cells (ix, nMinCellY..nMaxCellY) are adjacent in the CSR (column-major grid): one contiguous item range
                const int k0 = G.cell_first[ix * G.rows + nMinCellY], k1 = G.cell_first[ix * G.rows + nMaxCellY + 1];
                for (int base = k0; base < k1; base += 64) {
                    const int k = base + lane;
                    bool keep = false; int i = 0;
                    if (k < k1) {
                        i = G.cell_items[k];
                        const int o = G.oct[i];
                        keep = !(check && (o < minL || (maxL >= 0 && o > maxL)));
                        const float dx = G.kx[i] - x, dy = G.ky[i] - y;
                        keep = keep && fabsf(dx) < r && fabsf(dy) < r;
                    }
                    const unsigned long long m = __ballot(keep);
                    if (keep) {
                        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                        if (pos < cap) {
                            const uint4* b = reinterpret_cast<const uint4*>(G.desc) + 2 * (long long)i;
                            cand_idx[(long long)q * cap + pos] = i;
                            cand_dist[(long long)q * cap + pos] = ham256(a0, a1, b[0], b[1]);
                        }
                    }
                    n += __popcll(m);
                }
            }
        }
    }
    if (lane == 0) cand_n[q] = n;
}
// ---- device-side acceptance of the windowed matchers: no candidate list leaves the GPU.
// k_window_select: the selection loop of ORBmatcher::Fuse, both overloads (ORBmatcher.cpp:914-955, :1072-1100), and of
// SearchBySim3 (:1196-1245): the same window walk as k_window_candidates, but every lane keeps the best (distance, visiting
// position) of the candidates it filters and the wave reduces to ONE (index, distance) per query.  "First of equal distances
// wins" (the reference's strict `dist < bestDist` in visiting order) = minimum of distance * 2^32 + position.
// BATCH (ccm_fuse_select_batch): the queries of many keyframes in one launch; query q belongs to keyframe q_kf[q], whose grid is
// grids[q_kf[q]] (the server's fuse loops call Fuse once per neighbouring keyframe: src/Mapping.cpp:515-546, src/MapMerger.cpp:576-586).
template <bool BATCH>
__global__ __launch_bounds__(256) void k_window_select(WinGrid G1, const WinGrid* __restrict__ grids, const int* __restrict__ q_kf,
                                                       int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                       const float* __restrict__ qr, const int* __restrict__ min_level,
                                                       const int* __restrict__ max_level, const uint8_t* __restrict__ qdesc,
                                                       const float* __restrict__ inv_sigma2, int accept_th,
                                                       int* __restrict__ best_idx, int* __restrict__ best_dist)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const WinGrid G = BATCH ? grids[q_kf[q]] : G1;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minL = min_level[q], maxL = max_level[q];
    unsigned long long best = ~0ull; int bidx = -1;
    int n = 0;
    if (r >= 0.f) {
        const int nMinCellX = max(0, (int)floorf((x - G.min_x - r) * G.inv_w));
        const int nMaxCellX = min(G.cols - 1, (int)ceilf((x - G.min_x + r) * G.inv_w));
        const int nMinCellY = max(0, (int)floorf((y - G.min_y - r) * G.inv_h));
        const int nMaxCellY = min(G.rows - 1, (int)ceilf((y - G.min_y + r) * G.inv_h));
        if (nMinCellX < G.cols && nMaxCellX >= 0 && nMinCellY < G.rows && nMaxCellY >= 0) {
            const uint4* qd = reinterpret_cast<const uint4*>(qdesc) + 2 * (long long)q;
            const uint4 a0 = qd[0], a1 = qd[1];
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                const int k0 = G.cell_first[ix * G.rows + nMinCellY], k1 = G.cell_first[ix * G.rows + nMaxCellY + 1];
                for (int base = k0; base < k1; base += 64) {
                    const int k = base + lane;
                    bool in_window = false, keep = false; int i = 0;
                    if (k < k1) {
                        i = G.cell_items[k];
                        const float dx = G.kx[i] - x, dy = G.ky[i] - y;
                        in_window = fabsf(dx) < r && fabsf(dy) < r;           // GetFeaturesInArea is called without level limits here
                        const int o = G.oct[i];
                        keep = in_window && !(o < minL || o > maxL);           // :925-926 / :1081-1082 kpLevel in [level-1, level]
                        if (keep && inv_sigma2) {                              // :929-937: chi2 of the reprojection, Fuse(pKF, vpMapPoints) only
                            const float ex = x - G.kx[i], ey = y - G.ky[i];
                            const float e2 = ex * ex + ey * ey;
                            if ((double)(e2 * inv_sigma2[o]) > 5.99) keep = false;
                        }
                    }
                    const unsigned long long m = __ballot(in_window);
                    if (keep) {
                        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                        const uint4* b = reinterpret_cast<const uint4*>(G.desc) + 2 * (long long)i;
                        const unsigned long long key = ((unsigned long long)ham256(a0, a1, b[0], b[1]) << 32) | (unsigned)pos;
                        if (key < best) { best = key; bidx = i; }
                    }
                    n += __popcll(m);
                }
            }
        }
    }
    unsigned long long wbest = best;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(wbest, d, 64); wbest = o < wbest ? o : wbest; }
    const unsigned long long owner = __ballot(best == wbest && bidx >= 0);
    int idx = -1, dist = 256;
    if (owner) { idx = __shfl(bidx, __ffsll((long long)owner) - 1, 64); dist = (int)(wbest >> 32); }
    if (lane == 0) { best_dist[q] = dist; best_idx[q] = dist <= accept_th ? idx : -1; }
}
void match_launch_window_select(hipStream_t s, const WinGrid& G, int nq, const float* qx, const float* qy, const float* qr, const int* minl,
                                const int* maxl, const uint8_t* qdesc, const float* inv_sigma2, int accept_th, int* best_idx, int* best_dist)
{
    hipLaunchKernelGGL(k_window_select<false>, dim3((nq + 3) / 4), dim3(256), 0, s, G, (const WinGrid*)nullptr, (const int*)nullptr, nq, qx, qy, qr, minl, maxl, qdesc, inv_sigma2,
                       accept_th, best_idx, best_dist);
}
void match_launch_window_select_batch(hipStream_t s, const WinGrid* grids, const int* q_kf, int nq, const float* qx, const float* qy, const float* qr,
                                      const int* minl, const int* maxl, const uint8_t* qdesc, const float* inv_sigma2, int accept_th, int* best_idx, int* best_dist)
{
    if (nq > 0) hipLaunchKernelGGL(k_window_select<true>, dim3((nq + 3) / 4), dim3(256), 0, s, WinGrid{}, grids, q_kf, nq, qx, qy, qr, minl, maxl, qdesc, inv_sigma2,
                                   accept_th, best_idx, best_dist);
}

// k_window_greedy: the ORDER-DEPENDENT acceptance of SearchByProjection(Frame&, vpMapPoints) (ORBmatcher.cpp:71-148, MODE 0),
// SearchByProjection(pKF, Scw, vpPoints, vpMatched) (:308-446, MODE 1) and SearchByProjection(CurrentFrame, LastFrame / KeyFrame)
// (:1350-1476, :1478-1605, MODE 2) on the candidate lists k_window_candidates left in HBM.
// The reference visits the map points one after the other, and a point's choice depends on which features earlier points have
// taken (`occupied` / `vpMatched` / the frame's map-point slots).  A point takes the nearest of its still-free candidates (MODE 0
// also looks at the second nearest for the ratio test), and flags are only ever SET.  Hence the parallel form used here:
//   an unresolved point may decide as soon as NO EARLIER unresolved point has the feature it would take (MODE 0: the two it
//   looks at) among its own free candidates.  Flags that earlier points set later can then only fall on candidates this point
//   does not look at -- removing them changes neither its minimum nor the list order of the rest -- and the flag this point sets
//   falls on a feature no earlier unresolved point can choose.  "No acceptable candidate" is final at once: candidates only vanish.
// The earliest unresolved point always decides, so the rounds end.
// Layout: one workgroup of 16 waves; 1024 consecutive points per batch, thread = point, its first 16 candidates (feature | distance)
// in registers -- all global loads of a batch are issued by all threads at once; then the waves take turns in point order, and inside
// a wave the rounds need no workgroup barrier: claim[feature] = lowest lane among the unresolved lanes that have the feature free
// (LDS atomicMin), decide where the claim on the wanted feature(s) is one's own, reset the claims.  With many points per feature
// (20,000 points on a 1,000-feature keyframe) round 2's first version -- claims on ALL free candidates over the whole list, 48 rounds,
// then one wave visiting the rest in order with two global round trips per point -- took 9.1 ms; this takes the same decisions.
struct GreedyArgs {
    int nq, n, cap;
    const int* ci; const int* cd; const int* cn;      // candidate lists [nq][cap], counts
    const uint8_t* active;                            // mbTrackInView && !isBad  /  passed the projection tests
    const int* qlevel; const int* oct;                // predicted level per point, octave per feature
    const uint8_t* qflag;                             // Observations() > 0  /  already observed in the keyframe
    uint8_t* flag;                                    // in/out per feature: occupied / matched
    float nnratio;
    int* out;                                         // MODE 0 / 2: match[feature] = point; MODE 1: best_idx[point] = feature
    int* status;                                      // [0] matches (or -1: a list overflowed), [1] longest list, [2] most rounds a wave needed
    // MODE 2: acceptance threshold, rotation check and its inputs; ev[point] = accepted feature << 8 | rotation bin (or -1)
    int orb_dist, check_ori; const float* q_angle; const float* f_angle; int* ev;
    // batch (MODE 1, ccm_search_by_projection_sim3_batch): workgroup k works on keyframe k -- queries kfs[k].q0 .. +nq of the per-query
    // arrays, features kfs[k].f0 .. +n of the per-feature arrays, status words 3k .. 3k+2; nullptr = one problem, as described above
    const struct GreedyKf* kfs;
};
struct GreedyKf { int q0, nq, f0, n; };
#define WG_REG_CAND 16
#define WG_BATCH_ROUNDS 3
template <int MODE>
__global__ __launch_bounds__(1024) void k_window_greedy(GreedyArgs A)
{
    if (A.kfs) {
        const GreedyKf k = A.kfs[blockIdx.x];
        A.ci += (long long)k.q0 * A.cap; A.cd += (long long)k.q0 * A.cap; A.cn += k.q0; A.active += k.q0; A.qflag += k.q0;
        if (A.qlevel) A.qlevel += k.q0;
        A.oct += k.f0; A.flag += k.f0; A.out += MODE == 1 ? k.q0 : k.f0; A.status += 3 * blockIdx.x;
        A.nq = k.nq; A.n = k.n;
    }
    extern __shared__ int wg_lds[];
    int* claim = wg_lds;                                            // [n]
    uint8_t* flag = reinterpret_cast<uint8_t*>(claim + A.n);        // [n]
    uint8_t* octl = flag + ((A.n + 3) & ~3);                        // [n] octave per feature
    __shared__ int s_count, s_maxcn, s_rounds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_count = 0; s_maxcn = 0; s_rounds = 0; }
    for (int i = tid; i < A.n; i += 1024) { flag[i] = A.flag[i]; octl[i] = (uint8_t)A.oct[i]; claim[i] = 0x7FFFFFFF; }
    __syncthreads();
    int mx = 0;
    for (int m = tid; m < A.nq; m += 1024) mx = max(mx, A.cn[m]);
    atomicMax(&s_maxcn, mx);
    __syncthreads();
    if (s_maxcn > A.cap) { if (tid == 0) { A.status[0] = -1; A.status[1] = s_maxcn; A.status[2] = 0; } return; }   // the host retries with longer lists
    int mine = 0, most = 0;
    for (int base = 0; base < A.nq; base += 1024) {
        const int m = base + tid;
        int c = 0, lvl = 0, qf = 0;
        unsigned cand[WG_REG_CAND];                                  // feature | distance << 16
        if (m < A.nq && A.active[m]) c = A.cn[m];
        const int* ci = A.ci + (long long)min(m, A.nq - 1) * A.cap; const int* cd = A.cd + (long long)min(m, A.nq - 1) * A.cap;
#pragma unroll
        for (int k = 0; k < WG_REG_CAND; k++) cand[k] = k < c ? ((unsigned)ci[k] | ((unsigned)cd[k] << 16)) : 0u;
        if (c > 0) { qf = A.qflag[m]; if (MODE == 1) lvl = A.qlevel[m]; }
        if (MODE == 2 && m < A.nq) A.ev[m] = -1;
        // candidate k of this point is free and of a level the point may take
        auto usable = [&](int idx) -> bool {
            if (flag[idx]) return false;
            if (MODE == 1) { const int kl = octl[idx]; if (kl < lvl - 1 || kl > lvl) return false; }      // :394-400
            return true;
        };
        // one claim round for the unresolved points: `me` orders them (thread id inside the batch, or lane inside a wave); WG selects
        // the workgroup barrier (all 1024 points at once) or the wave barrier (one wave on its turn)
        bool unres = c > 0;
        auto sync = [&](bool wg) {
            if (wg) __syncthreads();
            else { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }
        };
        auto round = [&](int me, bool wg) {
            // nearest and second nearest usable candidate in (distance, list position) order: what the sequential scan's
            // `dist < bestDist` / `else if dist < bestDist2` keeps; and the claims
            unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
            if (unres) {
#pragma unroll
                for (int k = 0; k < WG_REG_CAND; k++) {
                    if (k < c) {
                        const int idx = (int)(cand[k] & 0xFFFFu);
                        if (usable(idx)) {
                            atomicMin(&claim[idx], me);
                            const unsigned key = (cand[k] & 0xFFFF0000u) | (unsigned)k;
                            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                        }
                    }
                }
                for (int k = WG_REG_CAND; k < c; k++) {
                    const int idx = ci[k];
                    if (usable(idx)) {
                        atomicMin(&claim[idx], me);
                        const unsigned key = ((unsigned)cd[k] << 16) | (unsigned)k;
                        if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                    }
                }
            }
            sync(wg);
            const int th = MODE == 0 ? 100 : (MODE == 1 ? 50 : A.orb_dist);       // TH_HIGH / TH_LOW / TH_HIGH or ORBdist
            const int bestDist = k1 == 0xFFFFFFFFu ? 256 : (int)(k1 >> 16);
            int idx1 = -1, idx2 = -1;
            bool safe = false;
            if (unres) {
                if (bestDist > th) safe = true;                                     // no acceptable candidate: final
                else {
                    const int p1 = (int)(k1 & 0xFFFFu), p2 = (int)(k2 & 0xFFFFu);
                    if (p1 >= WG_REG_CAND) idx1 = ci[p1];
                    if (MODE == 0 && k2 != 0xFFFFFFFFu && p2 >= WG_REG_CAND) idx2 = ci[p2];
#pragma unroll
                    for (int k = 0; k < WG_REG_CAND; k++) {
                        if (k == p1) idx1 = (int)(cand[k] & 0xFFFFu);
                        if (MODE == 0 && k2 != 0xFFFFFFFFu && k == p2) idx2 = (int)(cand[k] & 0xFFFFu);
                    }
                    safe = claim[idx1] == me;
                    if (MODE == 0 && idx2 >= 0) safe = safe && claim[idx2] == me;
                }
            }
            sync(wg);
            if (unres) {                                                            // claims back to "nobody"
#pragma unroll
                for (int k = 0; k < WG_REG_CAND; k++) if (k < c) claim[cand[k] & 0xFFFFu] = 0x7FFFFFFF;
                for (int k = WG_REG_CAND; k < c; k++) claim[ci[k]] = 0x7FFFFFFF;
            }
            if (unres && safe) {
                unres = false;
                if (bestDist <= th) {
                    if (MODE == 0) {
                        const int bestLevel = octl[idx1], bestLevel2 = idx2 >= 0 ? (int)octl[idx2] : -1;
                        const int bestDist2 = idx2 >= 0 ? (int)(k2 >> 16) : 256;
                        if (!(bestLevel == bestLevel2 && bestDist > A.nnratio * bestDist2)) { A.out[idx1] = m; flag[idx1] = (uint8_t)qf; mine++; }
                    } else if (MODE == 1) {
                        A.out[m] = idx1;
                        if (!qf) { flag[idx1] = 1; mine++; }                        // :436-440
                    } else {
                        A.out[idx1] = m;
                        flag[idx1] = (uint8_t)qf;
                        int bin = 255;
                        if (A.check_ori) {                                          // :1437-1447
                            float rot = A.q_angle[m] - A.f_angle[idx1];
                            if (rot < 0.0) rot += 360.0f;
                            bin = (int)roundf(rot * (1.0f / 30));
                            if (bin == 30) bin = 0;
                        }
                        A.ev[m] = (idx1 << 8) | bin;
                        mine++;
                    }
                }
            }
        };
        // (1) a few rounds over the whole batch: with few points per feature nearly every point decides here
        int left = 1, rounds = 0;
        for (int r = 0; r < WG_BATCH_ROUNDS && left; r++) {
            round(tid, true);
            left = __syncthreads_count(unres ? 1 : 0);                              // also: flags and claim resets visible to all
            rounds++;
        }
        // (2) what is left (many points wanting the same features) is finished wave by wave, in point order, without workgroup barriers
        //     inside a wave's turn
        if (left) {
            for (int turn = 0; turn < 16; turn++) {
                if (wv == turn) {
                    while (__ballot(unres) != 0ull) {
                        rounds++;
                        round(lane, false);
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __syncthreads();
            }
        }
        most = max(most, rounds);
    }
    if (mine) atomicAdd(&s_count, mine);
    atomicMax(&s_rounds, most);
    __syncthreads();
    if (MODE == 2 && A.check_ori) {
        // rotation consistency (:1453-1471): histogram of the accepted matches' angle differences, the three largest bins stay
        // (ComputeThreeMaxima :1607-1648), every match recorded in another bin is cleared -- also when a later point re-took the feature
        __shared__ int hist[30], keep3[3], s_removed;
        if (tid < 30) hist[tid] = 0;
        if (tid == 0) s_removed = 0;
        __syncthreads();
        for (int m = tid; m < A.nq; m += 1024) { const int e = (A.active[m] && A.cn[m] > 0) ? A.ev[m] : -1; if (e >= 0) atomicAdd(&hist[e & 255], 1); }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < 30; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
                else if (sz > max3) { max3 = sz; i3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if (max3 < 0.1f * (float)max1) { i3 = -1; }
            keep3[0] = i1; keep3[1] = i2; keep3[2] = i3;
        }
        __syncthreads();
        int removed = 0;
        for (int m = tid; m < A.nq; m += 1024) {
            const int e = (A.active[m] && A.cn[m] > 0) ? A.ev[m] : -1;
            if (e < 0) continue;
            const int b = e & 255;
            if (b != keep3[0] && b != keep3[1] && b != keep3[2]) { A.out[e >> 8] = -1; removed++; }
        }
        if (removed) atomicAdd(&s_removed, removed);
        __syncthreads();
        if (tid == 0) s_count -= s_removed;
        __syncthreads();
    }
    for (int i = tid; i < A.n; i += 1024) A.flag[i] = flag[i];
    if (tid == 0) { A.status[0] = s_count; A.status[1] = s_maxcn; A.status[2] = s_rounds; }
}
size_t match_window_greedy_lds(int n, int nq) { (void)nq; return (size_t)4 * n + 2 * (size_t)((n + 3) & ~3) + 16; }
int match_launch_window_greedy_batch(hipStream_t s, const GreedyArgs& A, int n_kf, int max_n)
{
    const size_t lds = match_window_greedy_lds(max_n, 0);
    if (hipFuncSetAttribute((const void*)k_window_greedy<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_window_greedy<1>, dim3(n_kf), dim3(1024), lds, s, A);
    return 0;
}
int match_launch_window_greedy(hipStream_t s, int mode, const GreedyArgs& A)
{
    const size_t lds = match_window_greedy_lds(A.n, A.nq);
    if (mode == 0) {
        if (hipFuncSetAttribute((const void*)k_window_greedy<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_window_greedy<0>, dim3(1), dim3(1024), lds, s, A);
    } else if (mode == 1) {
        if (hipFuncSetAttribute((const void*)k_window_greedy<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_window_greedy<1>, dim3(1), dim3(1024), lds, s, A);
    } else {
        if (hipFuncSetAttribute((const void*)k_window_greedy<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_window_greedy<2>, dim3(1), dim3(1024), lds, s, A);
    }
    return 0;
}

void match_launch_window(hipStream_t s, const WinGrid& G, int nq, const float* qx, const float* qy, const float* qr, const int* minl,
                         const int* maxl, const uint8_t* qdesc, int cap, int* ci, int* cd, int* cn)
{
    hipLaunchKernelGGL(k_window_candidates<false>, dim3((nq + 3) / 4), dim3(256), 0, s, G, (const WinGrid*)nullptr, (const int*)nullptr, nq, qx, qy, qr, minl, maxl, qdesc, cap,
                       ci, cd, cn);
}
void match_launch_window_batch(hipStream_t s, const WinGrid* grids, const int* q_kf, int nq, const float* qx, const float* qy, const float* qr, const int* minl,
                               const int* maxl, const uint8_t* qdesc, int cap, int* ci, int* cd, int* cn)
{
    if (nq > 0) hipLaunchKernelGGL(k_window_candidates<true>, dim3((nq + 3) / 4), dim3(256), 0, s, WinGrid{}, grids, q_kf, nq, qx, qy, qr, minl, maxl, qdesc, cap, ci, cd, cn);
}

// variant: 0 = 512 threads x 2 queries, 1 = 256 x 4, 2 = 1024 x 1 (tuning knob, CCM_BF_VARIANT)
void match_launch_bf(hipStream_t s, const uint8_t* q, long long q_pair_bytes, const uint8_t* t, long long t_pair_bytes,
                     int nq, int nt, int n_pairs, const int* nq_n, const int* nt_n, int n_split, int variant,
                     unsigned* part_best, int* part_second, int* bi, int* bd, int* sd)
{
    if (variant == 3 && nt <= HM_MAX_NT) {
        const int q_blocks = (nq + HM_QW * HM_WAVES - 1) / (HM_QW * HM_WAVES);
        hipLaunchKernelGGL(k_hamming_mfma, dim3(n_pairs * q_blocks), dim3(HM_TPB), 0, s, q, q_pair_bytes, t, t_pair_bytes, nq, nt, nq_n, nt_n,
                           q_blocks, bi, bd, sd);
        return;
    }
    const dim3 grid(n_pairs * n_split);
    if (variant == 1)
        hipLaunchKernelGGL((k_hamming_bf<256, 4>), grid, dim3(256), 0, s, q, q_pair_bytes, t, t_pair_bytes, nq, nt, nq_n, nt_n, n_split, part_best, part_second, bi, bd, sd);
    else if (variant == 2)
        hipLaunchKernelGGL((k_hamming_bf<1024, 1>), grid, dim3(1024), 0, s, q, q_pair_bytes, t, t_pair_bytes, nq, nt, nq_n, nt_n, n_split, part_best, part_second, bi, bd, sd);
    else
        hipLaunchKernelGGL((k_hamming_bf<512, 2>), grid, dim3(512), 0, s, q, q_pair_bytes, t, t_pair_bytes, nq, nt, nq_n, nt_n, n_split, part_best, part_second, bi, bd, sd);
    if (n_split > 1)
        hipLaunchKernelGGL(k_hamming_merge, dim3((unsigned)(((long long)n_pairs * nq + 255) / 256)), dim3(256), 0, s,
                           part_best, part_second, n_pairs, nq, n_split, nq_n, bi, bd, sd);
}
void match_launch_ranges(hipStream_t s, const uint8_t* d1, const uint8_t* d2, const int* order2, const int* start,
                         const int* len, const long long* off, int n1, unsigned short* dist)
{
    hipLaunchKernelGGL(k_hamming_ranges, dim3((n1 + 3) / 4), dim3(256), 0, s, d1, d2, order2, start, len, off, n1, dist);
}
