// ba_sparse.hip -- block-sparse reduced camera system (Schur complement) and its PCG solver.
//
// What BlockSolver<6,3>::solve does per landmark (cslam/thirdparty/g2o/g2o/core/block_solver.hpp:381-432),
//     Hschur(i,j) -= Hpl(i,l) Dinv(l) Hpl(j,l)^T   for every pair i <= j of poses observing landmark l,
// is reorganised for the GPU as a GATHER: all (landmark, pose-pair) contributions are enumerated once per
// problem, sorted by their target 6x6 block (rocPRIM radix sort, stable), and each block is then summed by
// one wave in that fixed order -- no atomics, bitwise reproducible, and the reduced system stays
// block-sparse (the covisibility pattern, ~130 blocks per keyframe row at BASELINE config 5 instead of 2000).
// The block pattern is the union over all ranks (byte map all-reduced with MAX), so every rank packs its
// partial blocks identically and one RCCL all-reduce over the packed nnz blocks completes the sum.
//
// The reduced system is solved by block-Jacobi preconditioned conjugate gradients on the packed blocks
// (stand-in for LinearSolverEigen's sparse LDLT, solvers/linear_solver_eigen.h:106-136; converged to a
// relative residual of 1e-13 it agrees with the exact solve far below the 1e-5 pose tolerance); small
// systems and non-converging ones are scattered into a dense array and go to the dense solve at the end of this file.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <cstdint>
#include "ba_types.h"
#include "ba_math.h"

// ---------------------------------------------------------------------------------------- structure
// number of (a <= b) pairs among a landmark's edges whose pose is free
__global__ __launch_bounds__(256) void k_sp_pair_count(BaDev D, int* __restrict__ cnt)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= D.L) return;
    int kf = 0;
    for (int e = D.pt_first[l]; e < D.pt_first[l + 1]; e++) kf += D.free_of[D.edge_pose[e]] >= 0;
    cnt[l] = kf * (kf + 1) / 2;
}

// enumerate the pairs: key = f_a * nfree + f_b (f_a <= f_b because edges are sorted by pose), value = (ea, eb), in the order of a walk
// through the landmark's triangle row by row.  SP_PF_LANES lanes per landmark: they collect the landmark's free edges in LDS (in edge
// order, by ballot) and write the triangle's entries i = g, g + SP_PF_LANES, ... -- neighbouring lanes, neighbouring entries.  (One
// thread per landmark left every lane of a wave storing into a region of its own, 46 entries apart at config 5: 260 us for 110 MB;
// a lane per ROW of the triangle, re-reading the edges from global memory: 153 us.)  A landmark with more than SP_PF_CAP free edges
// takes the row form.
#define SP_PF_LANES 8
#define SP_PF_CAP 48
__global__ __launch_bounds__(256) void k_sp_pair_fill(BaDev D, const int* __restrict__ off, unsigned* __restrict__ key,
                                                      unsigned long long* __restrict__ val)
{
    __shared__ int2 fl[256 / SP_PF_LANES][SP_PF_CAP];
    const int gt = blockIdx.x * 256 + threadIdx.x;
    const int l = gt / SP_PF_LANES, g = gt - l * SP_PF_LANES;
    if (l >= D.L) return;
    int2* mine = fl[threadIdx.x / SP_PF_LANES];
    const int e0 = D.pt_first[l], e1 = D.pt_first[l + 1];
    const int shift = (threadIdx.x & 63) - g;                        // first lane of the group inside its wave
    int kf = 0;
    for (int eb = e0; eb < e1; eb += SP_PF_LANES) {                  // (the same trip count for the lanes of a group)
        const int e = eb + g;
        const int fr = e < e1 ? D.free_of[D.edge_pose[e]] : -1;
        const unsigned gm = (unsigned)(__ballot(fr >= 0) >> shift) & ((1u << SP_PF_LANES) - 1u);
        if (fr >= 0) {
            const int pos = kf + __popc(gm & ((1u << g) - 1u));
            if (pos < SP_PF_CAP) mine[pos] = make_int2(e, fr);
        }
        kf += __popc(gm);
    }
    const int base = off[l];
    if (kf <= SP_PF_CAP) {
        const int T = kf * (kf + 1) / 2;
        int r = 0, row_start = 0, row_len = kf;
        for (int i = g; i < T; i += SP_PF_LANES) {
            while (i >= row_start + row_len) { row_start += row_len; row_len--; r++; }
            const int2 ea = mine[r], eb2 = mine[r + (i - row_start)];
            key[base + i] = (unsigned)ea.y * (unsigned)D.nfree + (unsigned)eb2.y;
            val[base + i] = ((unsigned long long)(unsigned)ea.x << 32) | (unsigned)eb2.x;
        }
        return;
    }
    int r = 0;
    for (int a = e0; a < e1; a++) {
        const int fa = D.free_of[D.edge_pose[a]];
        if (fa < 0) continue;
        if (r % SP_PF_LANES == g) {
            int p = base + r * kf - r * (r - 1) / 2;
            for (int b = a; b < e1; b++) {
                const int fb = D.free_of[D.edge_pose[b]];
                if (fb < 0) continue;
                key[p] = (unsigned)fa * (unsigned)D.nfree + (unsigned)fb;
                val[p] = ((unsigned long long)(unsigned)a << 32) | (unsigned)b;
                p++;
            }
        }
        r++;
    }
}

__global__ __launch_bounds__(256) void k_sp_mark(const unsigned* __restrict__ key, long long np, int nfree, uint8_t* __restrict__ map)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i < np) map[key[i]] = 1;
    if (i < nfree) map[(unsigned)i * (unsigned)nfree + (unsigned)i] = 1;      // every free pose owns its diagonal block
}

struct U8ToInt { __host__ __device__ int operator()(uint8_t v) const { return v ? 1 : 0; } };

__global__ __launch_bounds__(256) void k_sp_block_coords(const uint8_t* __restrict__ map, const int* __restrict__ id, long long n2, int nfree,
                                                         int* __restrict__ blk_row, int* __restrict__ blk_col, int* __restrict__ diag_id)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n2 || !map[i]) return;
    const int r = (int)(i / nfree), c = (int)(i - (long long)r * nfree);
    blk_row[id[i]] = r; blk_col[id[i]] = c;
    if (r == c) diag_id[r] = id[i];
}

__global__ __launch_bounds__(256) void k_sp_pair_block(const unsigned* __restrict__ key, const int* __restrict__ id, long long np, unsigned* __restrict__ out)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i < np) out[i] = (unsigned)id[key[i]];
}

__global__ __launch_bounds__(256) void k_sp_seg_bounds(const unsigned* __restrict__ sk, long long np, int* __restrict__ start, int* __restrict__ end)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= np) return;
    if (i == 0 || sk[i] != sk[i - 1]) start[sk[i]] = (int)i;
    if (i == np - 1 || sk[i] != sk[i + 1]) end[sk[i]] = (int)i + 1;
}

// symmetric row lists for the mat-vec: entry key = row * nfree + col, value = block id | transposed << 31
__global__ __launch_bounds__(256) void k_sp_row_entries(const int* __restrict__ blk_row, const int* __restrict__ blk_col, int nb, int nfree,
                                                        unsigned* __restrict__ key, unsigned* __restrict__ val)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    const unsigned r = blk_row[i], c = blk_col[i];
    key[2 * i] = r * (unsigned)nfree + c; val[2 * i] = (unsigned)i;
    key[2 * i + 1] = r == c ? 0xFFFFFFFFu : c * (unsigned)nfree + r;          // diagonal blocks appear once
    val[2 * i + 1] = (unsigned)i | 0x80000000u;
}
__global__ __launch_bounds__(256) void k_sp_row_ptr(const unsigned* __restrict__ skey, int n_ent, int nfree, int* __restrict__ row_ptr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ent) return;
    const unsigned k = skey[i];
    if (k == 0xFFFFFFFFu) return;
    const int r = (int)(k / (unsigned)nfree);
    const bool first = i == 0 || (int)(skey[i - 1] / (unsigned)nfree) != r;
    const bool last = i == n_ent - 1 || skey[i + 1] == 0xFFFFFFFFu || (int)(skey[i + 1] / (unsigned)nfree) != r;
    if (first) row_ptr[2 * r] = i;
    if (last) row_ptr[2 * r + 1] = i + 1;
}

// ---------------------------------------------------------------------------------------- per LM trial
// per landmark: Dinv = (Hll + lambda I)^-1, db = Dinv b_l, Y_e = Hpl_e Dinv for its edges
__global__ __launch_bounds__(256) void k_sp_dinv(BaDev D, double lambda)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= D.L) return;
    double Dm[9], Di[9];
    for (int i = 0; i < 9; i++) Dm[i] = D.Hll[9 * (long long)l + i];
    Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
    ba_inv3(Dm, Di);
    for (int i = 0; i < 9; i++) D.Dinv[9 * (long long)l + i] = Di[i];
    const double b0 = D.bl[3 * (long long)l], b1 = D.bl[3 * (long long)l + 1], b2 = D.bl[3 * (long long)l + 2];
    D.db[3 * (long long)l] = Di[0] * b0 + Di[1] * b1 + Di[2] * b2;
    D.db[3 * (long long)l + 1] = Di[3] * b0 + Di[4] * b1 + Di[5] * b2;
    D.db[3 * (long long)l + 2] = Di[6] * b0 + Di[7] * b1 + Di[8] * b2;
}
// per edge: Y_e = Hpl_e Dinv(landmark of e)  (one thread per edge: nine times the parallelism of a loop inside k_sp_dinv)
// Z_e = Hpl_e L^-T  (6 x 3) with Hll + lambda I = L L^T (Cholesky of the landmark's damped 3 x 3 block): Dinv = L^-T L^-1, so
// Hpl_a Dinv Hpl_b^T = Z_a Z_b^T, and BOTH operands of the Schur GEMM come from this one array.  With Y = Hpl Dinv on one side and
// Hpl on the other the kernel gathered from two arrays of 260 MB each at config 5 -- more than the 256 MB MALL holds; one array
// halves the working set.  (The factor is taken of the block itself, not of its computed inverse: three square roots of pivots that
// are positive whenever the block is.)
__global__ __launch_bounds__(256) void k_sp_edge_y(BaDev D, double lambda)
{
    // The workgroup's 256 blocks of Z (144 bytes each) and of ce (48) lie side by side: passed through LDS and stored as full lines
    // instead of 16 bytes per lane at a stride of 144 (182 -> 112 us at config 5; the same remedy as lm_store_group in ba_kernels.hip).
    __shared__ double2 stage[256 * 9];
    const long long e0 = blockIdx.x * 256LL;
    const long long e = e0 + threadIdx.x;
    double zz[18], cc[6];
    if (e < D.E) {
        const int l = D.edge_point[e];
        double f[6], Bx[18];
        ba_chol3(D.Hll + 9 * (long long)l, lambda, f);               // (pivot guard: see ba_math.h)
        const double2* B = reinterpret_cast<const double2*>(D.Hpl + 18 * e);
#pragma unroll
        for (int i = 0; i < 9; i++) { const double2 v = B[i]; Bx[2 * i] = v.x; Bx[2 * i + 1] = v.y; }
        const double* d = D.db + 3 * (long long)l;
        const double dd[3] = { d[0], d[1], d[2] };
        ba_edge_z_c(Bx, f, dd, zz, cc);
#pragma unroll
        for (int i = 0; i < 9; i++) stage[9 * threadIdx.x + i] = make_double2(zz[2 * i], zz[2 * i + 1]);
    }
    __syncthreads();
    const int ne = (int)min(256LL, D.E - e0);
    double2* zo = reinterpret_cast<double2*>(D.Z + 18 * e0);
#pragma unroll
    for (int j = 0; j < 9; j++) { const int q = j * 256 + threadIdx.x; if (q < 9 * ne) zo[q] = stage[q]; }
    __syncthreads();
    if (e < D.E) {
#pragma unroll
        for (int i = 0; i < 3; i++) stage[3 * threadIdx.x + i] = make_double2(cc[2 * i], cc[2 * i + 1]);
    }
    __syncthreads();
    double2* co = reinterpret_cast<double2*>(D.ce + 6 * e0);
#pragma unroll
    for (int j = 0; j < 3; j++) { const int q = j * 256 + threadIdx.x; if (q < 3 * ne) co[q] = stage[q]; }
}

// One workgroup per reduced-camera block: block = Hpp(diag) - sum over its sorted (landmark, pose pair) list of Z_a Z_b^T
// with Z_e = Hpl_e L^-T, Hll + lambda I = L L^T (6x3, k_sp_edge_y; the argument Y is that array).  The sum over pairs is one GEMM with K = 3 x pairs:
// [Y_a1 Y_a2 ...] (6 x K) times [W_b1 W_b2 ...]^T (K x 6), run on the f64 matrix cores as v_mfma_f64_16x16x4_f64
// (M = N = 16 of which 6 are used, K = 4 per instruction: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]).
// Four pairs = twelve k = three MFMAs per step (eight pairs per step measured slower); each lane gathers exactly the operand
// elements its (row, k) needs, so the operands never go through LDS.  The four waves of the workgroup take every fourth step
// (a keyframe's diagonal block has as many pairs as the keyframe has observations: one wave per block left a 0.5 ms tail),
// and a wave requests the pair indices two steps and the operands one step ahead of the MFMAs that use them (a step used to
// cost two dependent memory round trips).  The accumulation order -- the hardware's k order inside a wave, wave 0..3 at the
// end -- is fixed: reproducible run to run.
typedef double sp_v4d __attribute__((ext_vector_type(4)));
template <int NWV>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_sp_schur_blocks(BaDev D, const double* __restrict__ Y, const unsigned long long* __restrict__ pairs,
                                                         const int* __restrict__ seg_start, const int* __restrict__ seg_end,
                                                         const int* __restrict__ blk_row, const int* __restrict__ blk_col, int nb,
                                                         double* __restrict__ Hb, int per_xcd)
{
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The blocks are sorted by (row, column): in dispatch order
    // the ~57 blocks of a block row -- which all gather the row keyframe's Z blocks, and whose column keyframes are the next row's too --
    // are spread over all eight L2s, and every one of them fetches the same operands from the fabric.  per_xcd < 0 (shipped: -64): chunks
    // of -per_xcd consecutive block pairs go to one XCD, chunk c to XCD c % 8; per_xcd > 0: XCD x takes the contiguous range
    // [x per_xcd, (x + 1) per_xcd) (twice as slow as dispatch order: sp_launch_schur_blocks has the measurements).
    int wg = (int)blockIdx.x;
    if (per_xcd > 0) wg = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    else if (per_xcd < 0) {                                  // chunk-cyclic: -per_xcd consecutive block pairs to one XCD
        const int C = -per_xcd, xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        wg = ((j / C) * 8 + xcd) * C + j % C;
    }
    // TWO reduced blocks per workgroup share the 16 x 16 tile of the MFMA: operand rows 0..5 belong to block 2 g, rows 8..13 to block
    // 2 g + 1 (each with its own pair list), and the tile's two diagonal 6 x 6 corners are the two sums (the off-diagonal corners mix the
    // blocks and are dropped).  The same number of gather loads now serves two blocks: 48 of 64 lanes load instead of 24.
    __shared__ double part[NWV][72];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;              // operand row of the tile, k within an MFMA
    const int half = i16 >> 3, i = i16 & 7;                 // which of the two blocks, row inside it (< 6 used)
    const int b = 2 * wg + half;
    const bool live = i < 6 && b < nb;
    const int p0 = live ? seg_start[b] : 0, p1 = live ? seg_end[b] : 0;
    // the longer of the workgroup's two lists sets the trip count (wave-uniform)
    int pmax = p1 - p0;
    for (int st = 32; st >= 1; st >>= 1) pmax = max(pmax, __shfl_xor(pmax, st, 64));
    sp_v4d acc = { 0.0, 0.0, 0.0, 0.0 };
    // The twelve k of a step: k = 4 m + kq is column m of pair kq -- a lane's three operand elements are one row of ONE pair's block:
    // one pair index and 24 contiguous bytes per operand (a dwordx4 and a dwordx2), instead of three indices and three scattered
    // doubles.  The kernel's rate is set by its gather requests: 5 loads per step instead of 9.
    const unsigned long long NONE = ~0ull;
    typedef double sp_d2u __attribute__((ext_vector_type(2), aligned(8)));
    auto ld_idx = [&](int q) -> unsigned long long { return (live && p0 + q + kq < p1) ? pairs[p0 + q + kq] : NONE; };   // q = offset into the list
    auto ld_ops = [&](unsigned long long pr, double (&av)[3], double (&bv)[3]) {
        av[0] = av[1] = av[2] = 0.0; bv[0] = bv[1] = bv[2] = 0.0;
        if (pr != NONE) {
            const double* za = Y + 18 * (long long)(unsigned)(pr >> 32) + 3 * i;
            const double* zb = Y + 18 * (long long)(unsigned)(pr & 0xFFFFFFFFu) + 3 * i;
            const sp_d2u a01 = *reinterpret_cast<const sp_d2u*>(za), b01 = *reinterpret_cast<const sp_d2u*>(zb);
            av[0] = a01.x; av[1] = a01.y; av[2] = za[2];
            bv[0] = b01.x; bv[1] = b01.y; bv[2] = zb[2];
        }
    };
    unsigned long long pr_next, pr_far;
    double av[3], bv[3], av_next[3], bv_next[3];
    int q = 4 * wv;                                         // this wave's steps: 4 pairs each, 4 NWV pairs apart
    pr_next = ld_idx(q); ld_ops(pr_next, av, bv); pr_next = ld_idx(q + 4 * NWV);
    for (; q < pmax; q += 4 * NWV) {
        ld_ops(pr_next, av_next, bv_next);
        pr_far = ld_idx(q + 8 * NWV);
#pragma unroll
        for (int m = 0; m < 3; m++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[m], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; m++) { av[m] = av_next[m]; bv[m] = bv_next[m]; }
        pr_next = pr_far;
    }
    // C/D: column = lane & 15, row = (lane >> 4) + 4 * reg; block `half` sits in rows and columns 8 half .. 8 half + 5
    const int c = i16 & 7;
    if (c < 6) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int row = kq + 4 * reg;
            if ((row >> 3) == half && (row & 7) < 6) part[wv][36 * half + 6 * (row & 7) + c] = acc[reg];
        }
    }
    __syncthreads();
    if (threadIdx.x < 72) {
        const int h2 = threadIdx.x / 36, e = threadIdx.x - 36 * h2, bb = 2 * wg + h2;
        if (bb < nb) {
            const int rb = blk_row[bb], cb = blk_col[bb];
            const double base = rb == cb ? D.Hpp[36 * (long long)rb + e] : 0.0;
            double tot = part[0][threadIdx.x];
#pragma unroll
            for (int w2 = 1; w2 < NWV; w2++) tot += part[w2][threadIdx.x];                              // wave 0 .. NWV-1: fixed order
            Hb[36 * (long long)bb + e] = base - tot;
        }
    }
}

// one workgroup (4 waves) per free pose: bs = bp - sum over its edges of Hpl_e db(l_e); wave partials added 0..3 (fixed order)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_sp_bschur(BaDev D, double* __restrict__ bs)
{
    __shared__ double part[NW][6];
    const int f = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (f >= D.nfree) return;
    double c[6] = { 0, 0, 0, 0, 0, 0 };
    for (int k = D.pose_first[f] + threadIdx.x; k < D.pose_first[f + 1]; k += 64 * NW) {
        const double* ce = D.ce + 6 * (long long)D.pose_edges[k];         // Hpl_e db(l_e), left by k_sp_edge_y / the fused linearisation (0 for a dropped edge)
        for (int i = 0; i < 6; i++) c[i] += ce[i];
    }
    for (int i = 0; i < 6; i++)
        for (int s = 32; s >= 1; s >>= 1) c[i] += __shfl_xor(c[i], s, 64);
    if (lane == 0) for (int i = 0; i < 6; i++) part[wv][i] = c[i];
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
#pragma unroll
        for (int w = 4; w < NW; w++) v += part[w][threadIdx.x];                                          // wave 0 .. NW-1: fixed order
        bs[6 * (long long)f + threadIdx.x] = D.bp[6 * (long long)f + threadIdx.x] - v;
    }
}

// one wave per free pose: bs = bp - sum over its edges of Hpl_e db(l_e), fixed order
__global__ __launch_bounds__(256) void k_sp_bschur_wave(BaDev D, double* __restrict__ bs)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= D.nfree) return;
    double c[6] = { 0, 0, 0, 0, 0, 0 };
    // four edges in flight per lane: the 48-byte rows are gathered through the keyframe's edge list
    const int k1 = D.pose_first[f + 1];
    for (int k = D.pose_first[f] + lane; k < k1; k += 256) {
        typedef double bs_d2 __attribute__((ext_vector_type(2)));
        bs_d2 v[4][3];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kk = k + 64 * q;
#pragma unroll
            for (int i = 0; i < 3; i++) v[q][i] = bs_d2{0.0, 0.0};
            if (kk < k1) {
                const bs_d2* ce = reinterpret_cast<const bs_d2*>(D.ce + 6 * (long long)D.pose_edges[kk]);
                v[q][0] = ce[0]; v[q][1] = ce[1]; v[q][2] = ce[2];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { c[0] += v[q][0].x; c[1] += v[q][0].y; c[2] += v[q][1].x; c[3] += v[q][1].y; c[4] += v[q][2].x; c[5] += v[q][2].y; }
    }
    for (int i = 0; i < 6; i++)
        for (int s = 32; s >= 1; s >>= 1) c[i] += __shfl_xor(c[i], s, 64);
    if (lane < 6) bs[6 * (long long)f + lane] = D.bp[6 * (long long)f + lane] - c[lane];
}

__global__ __launch_bounds__(256) void k_sp_add_lambda(const int* __restrict__ diag_id, int nfree, double lambda, double* __restrict__ Hb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nfree * 6) return;
    const int f = i / 6, r = i - 6 * f;
    Hb[36 * (long long)diag_id[f] + 7 * r] += lambda;
}

// packed blocks -> dense row-major n x n (upper block triangle), for the dense path
__global__ __launch_bounds__(256) void k_sp_to_dense(const double* __restrict__ Hb, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                                     int nb, long long n, double* __restrict__ Hs)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= 36LL * nb) return;
    const int b = (int)(i / 36), e = (int)(i - 36LL * b), r = e / 6, c = e - 6 * r;
    Hs[(6LL * blk_row[b] + r) * n + 6 * blk_col[b] + c] = Hb[i];
}

// ---------------------------------------------------------------------------------------- PCG
// Cluster-Jacobi preconditioner: M = the diagonal blocks of PCG_CL consecutive free keyframes INCLUDING the coupling
// blocks between them (consecutive keyframes of a trajectory share most of their landmarks, so these are the strongest
// off-diagonal blocks of the reduced camera system); its inverse is a dense PCG_CN x PCG_CN matrix per cluster, applied
// as a mat-vec.  Minv layout: [cluster][PCG_CN][PCG_CN], symmetric.
#ifndef PCG_CL
#define PCG_CL 8               // keyframes per cluster (measured round 2: see DESIGN.md, preconditioner study on the GPU)
#endif
#ifndef PCG_XCDS
#define PCG_XCDS 8               // 1 = block rows in dispatch order (round-robin over the XCDs)
#endif
#define PCG_CN (6 * PCG_CL)
__global__ __launch_bounds__(256) void k_pcg_cl_gather(const double* __restrict__ Hb, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                                       int nb, double* __restrict__ Mc)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= 36LL * nb) return;
    const int b = (int)(i / 36), e = (int)(i - 36LL * b), r = e / 6, c = e - 6 * r;
    const int br = blk_row[b], bc = blk_col[b];
    if (br / PCG_CL != bc / PCG_CL) return;
    double* M = Mc + (long long)(br / PCG_CL) * PCG_CN * PCG_CN;
    const int rr = 6 * (br % PCG_CL) + r, cc = 6 * (bc % PCG_CL) + c;
    const double v = Hb[i];
    // The upper block triangle is stored once, so an off-diagonal block is mirrored.  Of a diagonal block only the upper
    // half is used, mirrored as well: taking both halves would let two not-quite-equal values race into one slot (seen
    // as 1e-15 run-to-run noise) and leave the preconditioner not exactly symmetric.
    if (br == bc && r > c) return;
    M[rr * PCG_CN + cc] = v;
    M[cc * PCG_CN + rr] = v;
}
// in-place inverse of every cluster matrix by Gauss-Jordan without pivoting (SPD); unused rows of the last cluster
// are made identity; a non-positive pivot raises `bad`.  The matrix lives in REGISTERS: thread (ty, tx) of the 16 x 16 workgroup owns
// the 3 x 3 elements (ty + 16 a, tx + 16 b); per pivot p the owners of column p and of row p publish them in LDS (two alternating
// buffers: one barrier per pivot) and every thread reads the three column and three row values its elements need -- the scheme of
// k_inv_diag below.  (Round 2 held the matrix in LDS with two barriers per pivot: 71 us per LM trial, on the critical path.)
static_assert(PCG_CN == 48, "k_pcg_cl_invert tiles a 48 x 48 cluster as 16 x 16 threads x 3 x 3 elements");
__global__ __launch_bounds__(256) void k_pcg_cl_invert(double* __restrict__ Mc, int nfree, int* __restrict__ bad)
{
    __shared__ double fcol[2][PCG_CN], prow[2][PCG_CN];
    __shared__ int s_bad;
    double* M = Mc + (long long)blockIdx.x * PCG_CN * PCG_CN;
    const int used = 6 * min(PCG_CL, nfree - (int)blockIdx.x * PCG_CL);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double a[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { const int r = ty + 16 * i, c = tx + 16 * j; a[i][j] = (r < used && c < used) ? M[r * PCG_CN + c] : (r == c ? 1.0 : 0.0); }
    if (threadIdx.x == 0) s_bad = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (tx + 16 * j == 0) fcol[0][ty + 16 * i] = a[i][j];
            if (ty + 16 * i == 0) prow[0][tx + 16 * j] = a[i][j];
        }
    __syncthreads();
    for (int p = 0; p < PCG_CN; p++) {
        const int cur = p & 1, nxt = cur ^ 1;
        const double piv = prow[cur][p];
        if (!(piv > 0.0) && threadIdx.x == 0) s_bad = 1;
        const double ip = 1.0 / piv;
        double fc[3], pr[3];
#pragma unroll
        for (int i = 0; i < 3; i++) fc[i] = fcol[cur][ty + 16 * i];
#pragma unroll
        for (int j = 0; j < 3; j++) { const int c = tx + 16 * j; pr[j] = (c == p ? 1.0 : prow[cur][c]) * ip; }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int r = ty + 16 * i, c = tx + 16 * j;
                a[i][j] = r == p ? pr[j] : ((c == p ? 0.0 : a[i][j]) - fc[i] * pr[j]);
                if (c == p + 1) fcol[nxt][r] = a[i][j];
                if (r == p + 1) prow[nxt][c] = a[i][j];
            }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[(ty + 16 * i) * PCG_CN + tx + 16 * j] = a[i][j];
    if (threadIdx.x == 0 && s_bad) atomicOr(bad, 1);
}

// Second level of the preconditioner (additive two-level Schwarz): the cluster inverses above damp the error inside a
// cluster, but the slowly varying error along the trajectory (many keyframes drifting together) converges only as fast as
// information travels from cluster to cluster.  Coarse space, 7 unknowns per aggregate of A = PCG_CL * pcg_agg_clusters()
// keyframes: a rigid increment (6) and a scale change about the aggregate's centre (a monocular map with one fixed keyframe
// has a free scale; its local version is the softest deformation of a trajectory piece), interpolated LINEARLY between the
// aggregate centres (hat functions: keyframe f takes 1 - a of aggregate I and a of I + 1, x = (f + 1/2) / A - 1/2 = I + a;
// the ends are clamped).  P = prolongation (n x 7 nagg), Ac = P^T H P dense and small, inverted once per LM trial (block
// Gauss-Jordan below), and      z = Minv r + P Ac^-1 P^T r.
// Iterations of a late LM trial of config 5 (lambda = 0.04, relative residual 1e-6; tools/gba_coarse_study.py reproduces the
// counts on the CPU): cluster level alone 1165, piecewise-constant rigid aggregates (round 2 until here) 517, hat functions 148,
// hat functions + scale 98.
#ifndef PCG_AGG
#define PCG_AGG 2                // clusters per aggregate up to PCG_COARSE_MAX coarse unknowns; doubled beyond (at most 8)
#endif
#define PCG_CDOF 7               // coarse unknowns per aggregate
#define PCG_COARSE_MAX 1792
// clusters per aggregate for a map of `nfree` free keyframes: the smallest of PCG_AGG, 2 PCG_AGG, ... (<= 8) that keeps the coarse
// system within PCG_COARSE_MAX unknowns (its inversion is cubic and has to fit inside one LM trial)
__host__ __device__ inline int pcg_agg_clusters(int nfree)
{
    int agg = PCG_AGG;
    while (2 * agg * PCG_CL <= 64 && PCG_CDOF * ((nfree + PCG_CL * agg - 1) / (PCG_CL * agg)) > PCG_COARSE_MAX) agg *= 2;
    return agg;
}
// the two aggregates keyframe f interpolates between, and its weights (w0 + w1 = 1)
struct PcgHat { int i0, i1; double w0, w1; };
__host__ __device__ inline PcgHat pcg_hat(int f, int A, int nagg)
{
    const double x = ((double)f + 0.5) * (1.0 / (double)A) - 0.5;  // A is a power of two: multiples of 1 / (2A), exact
    const int I = (int)(x + 1.0) - 1;                              // floor(x) for x >= -1
    const double al = x - (double)I;
    PcgHat h;
    h.i0 = min(max(I, 0), nagg - 1); h.i1 = min(I + 1, nagg - 1);
    h.w0 = 1.0 - al; h.w1 = al;
    if (h.i0 == h.i1) { h.w0 = 1.0; h.w1 = 0.0; }
    return h;
}
// weight of keyframe f in aggregate I
__host__ __device__ inline double pcg_hat_weight(int f, int I, int A, int nagg)
{
    const PcgHat h = pcg_hat(f, A, nagg);
    return (h.i0 == I ? h.w0 : 0.0) + (h.i1 == I ? h.w1 : 0.0);
}
// keyframes with a non-zero weight in aggregate I: [first, last)
__host__ __device__ inline void pcg_hat_support(int I, int A, int nfree, int& first, int& last)
{
    first = max(0, A * I - A / 2); last = min(nfree, A * I + A + A / 2);
}
// upper triangle of Ac = P^T H P, row-major with pitch ncp: one workgroup per aggregate pair I <= J, thread = keyframe pair
// (i, j) of the two supports, the 49 sums reduced over the workgroup in a fixed order (the ranks of a sharded solve must get
// the same bits).  Column 6 of keyframe i's 6 x 7 basis W_i is (0, 0, 0, t_i - c_I): svec holds t_i, cen the aggregate centres.
// (the grid runs over the aggregate pairs that hold at least one block -- `pairs`, made once per call by k_pcg_coarse_mark and the
// host: 1 in 8 of the upper triangle at config 5; as a full nagg x nagg grid the kernel flooded the chip from the side stream for
// 0.26 ms per trial, and the PCG's 1024-thread workgroups starved behind its small ones)
__global__ __launch_bounds__(256) void k_pcg_coarse_mark(const int* __restrict__ blk_row, const int* __restrict__ blk_col, int nb, int nfree, int nagg, uint8_t* __restrict__ aggmap)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    const int A = PCG_CL * pcg_agg_clusters(nfree);
    const PcgHat ha = pcg_hat(blk_row[i], A, nagg), hb = pcg_hat(blk_col[i], A, nagg);
    const int ia[2] = { ha.i0, ha.i1 }, ib[2] = { hb.i0, hb.i1 };
    for (int x = 0; x < 2; x++)
        for (int y = 0; y < 2; y++) aggmap[min(ia[x], ib[y]) * nagg + max(ia[x], ib[y])] = 1;
}
__global__ __launch_bounds__(256) void k_pcg_coarse_build(const double* __restrict__ Hb, const uint8_t* __restrict__ map, const int* __restrict__ id,
                                                          int nfree, int nagg, int ncp, const double* __restrict__ svec, const double* __restrict__ cen,
                                                          const int* __restrict__ pairs, double* __restrict__ Ac)
{
    __shared__ double red[4][PCG_CDOF * PCG_CDOF];
    const int I = pairs[2 * blockIdx.x], J = pairs[2 * blockIdx.x + 1];
    if (J < I) return;
    const int A = PCG_CL * pcg_agg_clusters(nfree);
    int i0, i1, j0, j1;
    pcg_hat_support(I, A, nfree, i0, i1); pcg_hat_support(J, A, nfree, j0, j1);
    const int ni = i1 - i0, nj = j1 - j0;
    double acc[PCG_CDOF * PCG_CDOF];
#pragma unroll
    for (int e = 0; e < PCG_CDOF * PCG_CDOF; e++) acc[e] = 0.0;
    for (int t = threadIdx.x; t < ni * nj; t += 256) {
        const int i = i0 + t / nj, j = j0 + t % nj;
        const int a = min(i, j), b = max(i, j);
        const long long idx = (long long)a * nfree + b;
        if (!map[idx]) continue;
        const double wij = pcg_hat_weight(i, I, A, nagg) * pcg_hat_weight(j, J, A, nagg);
        if (wij == 0.0) continue;
        const double* B = Hb + 36LL * id[idx];
        double si[3], sj[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { si[q] = svec[3 * i + q] - cen[3 * I + q]; sj[q] = svec[3 * j + q] - cen[3 * J + q]; }
        double col6[6], row6[6] = { 0, 0, 0, 0, 0, 0 }, corner = 0.0;
#pragma unroll
        for (int d = 0; d < 6; d++) {
            double c6 = 0.0;
#pragma unroll
            for (int e = 0; e < 6; e++) {
                // block (a, b) is stored for a <= b; (i, j) with i > j is its transpose; of a diagonal block the upper half counts
                const double v = i < j ? B[6 * d + e] : (i > j ? B[6 * e + d] : (d <= e ? B[6 * d + e] : B[6 * e + d]));
                acc[PCG_CDOF * d + e] += wij * v;
                if (e >= 3) c6 += v * sj[e - 3];
                if (d >= 3) row6[e] += si[d - 3] * v;
            }
            col6[d] = c6;
            if (d >= 3) corner += si[d - 3] * c6;
        }
#pragma unroll
        for (int d = 0; d < 6; d++) { acc[PCG_CDOF * d + 6] += wij * col6[d]; acc[PCG_CDOF * 6 + d] += wij * row6[d]; }
        acc[PCG_CDOF * 6 + 6] += wij * corner;
    }
#pragma unroll
    for (int e = 0; e < PCG_CDOF * PCG_CDOF; e++) {
        double v = acc[e];
        for (int st = 32; st >= 1; st >>= 1) v += __shfl_xor(v, st, 64);
        acc[e] = v;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < PCG_CDOF * PCG_CDOF; e++) red[threadIdx.x >> 6][e] = acc[e];
    }
    __syncthreads();
    if (threadIdx.x < PCG_CDOF * PCG_CDOF) {
        const int d = threadIdx.x / PCG_CDOF, e = threadIdx.x - PCG_CDOF * d;
        const int r = PCG_CDOF * I + d, c = PCG_CDOF * J + e;
        if (r <= c) Ac[(long long)r * ncp + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}
// Ac was built as an upper triangle: make it the full ncp x ncp matrix the inversion works on (lower from upper, identity
// in the padding beyond nc)
__global__ __launch_bounds__(256) void k_pcg_coarse_complete(double* __restrict__ A, int nc, int ncp)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)ncp * ncp) return;
    const int r = (int)(i / ncp), c = (int)(i - (long long)r * ncp);
    if (r >= nc || c >= nc) A[i] = r == c ? 1.0 : 0.0;
    else if (r > c) A[i] = A[(long long)c * ncp + r];
    else if (r == c && A[i] == 0.0) A[i] = 1.0;      // an aggregate whose keyframes all sit at one point has no scale column: the unknown stays inert
}
// after the inversion: the upper triangle is mirrored so that the preconditioner is exactly symmetric
__global__ __launch_bounds__(256) void k_pcg_coarse_mirror(double* __restrict__ A, int ncp)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)ncp * ncp) return;
    const int r = (int)(i / ncp), c = (int)(i - (long long)r * ncp);
    if (r > c) A[i] = A[(long long)c * ncp + r];
}

// ---- dense inverse of the (symmetric positive definite) coarse matrix: in-place block Gauss-Jordan without pivoting,
// 48 x 48 blocks, four small kernels per block step, every sum in a fixed order -- the result is the same bits on every
// run and every rank (rocSOLVER's potrf + potri, used here first, differed in the last bits from run to run).
//   D = A_kk^-1;  A_kj <- D A_kj (j != k);  A_ij <- A_ij - A_ik A_kj (i, j != k);  A_ik <- -A_ik D (i != k);  A_kk <- D
#define INV_B 48
#define INV_T 3                  // a thread of the 16 x 16 workgroup owns INV_T x INV_T outputs of a block
static_assert(INV_B == 16 * INV_T, "256 threads tile a block");
// D = A_kk^-1 by in-place Gauss-Jordan in LDS
__global__ __launch_bounds__(256) void k_inv_diag(const double* __restrict__ A, int lda, int k, double* __restrict__ D, int* __restrict__ bad)
{
    // Gauss-Jordan with the block in REGISTERS: thread (ty, tx) of the 16 x 16 workgroup owns the 3 x 3 elements (ty + 16 a, tx + 16 b);
    // per pivot p the owners of column p and of row p publish them in LDS (two alternating buffers: one barrier per pivot), every
    // thread reads the three column and three row values its elements need.  (With the block in LDS and two barriers per pivot the
    // 19 diagonal blocks of config 5's coarse matrix took 73 us each -- 60 % of the inversion, which shares the GPU with the PCG.)
    __shared__ double fcol[2][INV_B], prow[2][INV_B];
    __shared__ int s_bad;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const double* Akk = A + ((long long)k * INV_B) * lda + (long long)k * INV_B;
    double a[INV_T][INV_T];
#pragma unroll
    for (int i = 0; i < INV_T; i++)
#pragma unroll
        for (int j = 0; j < INV_T; j++) a[i][j] = Akk[(long long)(ty + 16 * i) * lda + tx + 16 * j];
    if (threadIdx.x == 0) s_bad = 0;
    // publish column 0 and row 0
#pragma unroll
    for (int i = 0; i < INV_T; i++)
#pragma unroll
        for (int j = 0; j < INV_T; j++) {
            if (tx + 16 * j == 0) fcol[0][ty + 16 * i] = a[i][j];
            if (ty + 16 * i == 0) prow[0][tx + 16 * j] = a[i][j];
        }
    __syncthreads();
    for (int p = 0; p < INV_B; p++) {
        const int cur = p & 1, nxt = cur ^ 1;
        const double piv = prow[cur][p];
        if (!(piv > 0.0) && threadIdx.x == 0) s_bad = 1;
        const double ip = 1.0 / piv;
        double fc[INV_T], pr[INV_T];
#pragma unroll
        for (int i = 0; i < INV_T; i++) fc[i] = fcol[cur][ty + 16 * i];
#pragma unroll
        for (int j = 0; j < INV_T; j++) { const int c = tx + 16 * j; pr[j] = (c == p ? 1.0 : prow[cur][c]) * ip; }
#pragma unroll
        for (int i = 0; i < INV_T; i++)
#pragma unroll
            for (int j = 0; j < INV_T; j++) {
                const int r = ty + 16 * i, c = tx + 16 * j;
                a[i][j] = r == p ? pr[j] : ((c == p ? 0.0 : a[i][j]) - fc[i] * pr[j]);
                if (c == p + 1) fcol[nxt][r] = a[i][j];
                if (r == p + 1) prow[nxt][c] = a[i][j];
            }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < INV_T; i++)
#pragma unroll
        for (int j = 0; j < INV_T; j++) D[(ty + 16 * i) * INV_B + tx + 16 * j] = a[i][j];
    if (threadIdx.x == 0 && s_bad) atomicOr(bad, 1);
}
// C (one block, in registers: thread = INV_T x INV_T outputs) = X Y with X, Y staged in LDS; k ascending
__device__ __forceinline__ void inv_mm(const double (*X)[INV_B], const double (*Y)[INV_B], double (&c)[INV_T][INV_T])
{
    const int tr = INV_T * (threadIdx.x >> 4), tc = INV_T * (threadIdx.x & 15);
#pragma unroll
    for (int i = 0; i < INV_T; i++)
#pragma unroll
        for (int j = 0; j < INV_T; j++) c[i][j] = 0.0;
    for (int kk = 0; kk < INV_B; kk++) {
        double x[INV_T], y[INV_T];
#pragma unroll
        for (int i = 0; i < INV_T; i++) { x[i] = X[tr + i][kk]; y[i] = Y[kk][tc + i]; }
#pragma unroll
        for (int i = 0; i < INV_T; i++)
#pragma unroll
            for (int j = 0; j < INV_T; j++) c[i][j] += x[i] * y[j];
    }
}
__device__ __forceinline__ void inv_load(double (*T)[INV_B], const double* __restrict__ src, int ld)
{
    for (int i = threadIdx.x; i < INV_B * INV_B; i += 256) { const int r = i / INV_B, c = i - r * INV_B; T[r][c] = src[(long long)r * ld + c]; }
}
// mode 0: A_kj <- D A_kj (block column j = blockIdx.x, skipping k);  mode 1: A_ik <- -A_ik D (block row i = blockIdx.x, skipping k; the
// extra last workgroup stores A_kk <- D);  mode 2: A_ij <- A_ij - A_ik A_kj (i = blockIdx.y, j = blockIdx.x, both skipping k)
__global__ __launch_bounds__(256) void k_inv_step(double* __restrict__ A, int lda, int nblk, int k, const double* __restrict__ D, int mode)
{
    __shared__ double X[INV_B][INV_B], Y[INV_B][INV_B];
    const int tr = INV_T * (threadIdx.x >> 4), tc = INV_T * (threadIdx.x & 15);
    auto blk = [&](int bi, int bj) { return A + ((long long)bi * INV_B) * lda + (long long)bj * INV_B; };
    double c[INV_T][INV_T];
    if (mode == 1 && (int)blockIdx.x == nblk - 1) {                    // A_kk <- D
        double* K = blk(k, k);
        for (int i = threadIdx.x; i < INV_B * INV_B; i += 256) K[(long long)(i / INV_B) * lda + i % INV_B] = D[i];
        return;
    }
    const int bx = (int)blockIdx.x + ((int)blockIdx.x >= k ? 1 : 0);
    if (mode == 0) {
        double* T = blk(k, bx);
        inv_load(X, D, INV_B); inv_load(Y, T, lda);
        __syncthreads();
        inv_mm(X, Y, c);
#pragma unroll
        for (int i = 0; i < INV_T; i++)
#pragma unroll
            for (int j = 0; j < INV_T; j++) T[(long long)(tr + i) * lda + tc + j] = c[i][j];
    } else if (mode == 1) {
        double* T = blk(bx, k);
        inv_load(X, T, lda); inv_load(Y, D, INV_B);
        __syncthreads();
        inv_mm(X, Y, c);
#pragma unroll
        for (int i = 0; i < INV_T; i++)
#pragma unroll
            for (int j = 0; j < INV_T; j++) T[(long long)(tr + i) * lda + tc + j] = -c[i][j];
    } else {
        const int by = (int)blockIdx.y + ((int)blockIdx.y >= k ? 1 : 0);
        double* T = blk(by, bx);
        inv_load(X, blk(by, k), lda); inv_load(Y, blk(k, bx), lda);
        __syncthreads();
        inv_mm(X, Y, c);
#pragma unroll
        for (int i = 0; i < INV_T; i++)
#pragma unroll
            for (int j = 0; j < INV_T; j++) T[(long long)(tr + i) * lda + tc + j] -= c[i][j];
    }
}
// y = A x for a dense row-major n x n matrix of pitch lda: a wave per row, lanes stride the columns, fixed butterfly
__global__ __launch_bounds__(256) void k_dense_matvec(const double* __restrict__ A, int n, int lda, const double* __restrict__ x, double* __restrict__ y)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const double* a = A + (long long)row * lda;
    double s = 0;
    for (int c = lane; c < n; c += 64) s += a[c] * x[c];
    for (int st = 32; st >= 1; st >>= 1) s += __shfl_xor(s, st, 64);
    if (lane == 0) y[row] = s;
}
// in-place inverse of the ncp x ncp matrix A (ncp a multiple of INV_B); D = one block of scratch; *bad is raised on a non-positive pivot
void pcg_launch_coarse_invert(hipStream_t s, double* A, int ncp, double* D, int* bad)
{
    const int nb = ncp / INV_B;
    for (int k = 0; k < nb; k++) {
        hipLaunchKernelGGL(k_inv_diag, dim3(1), dim3(256), 0, s, A, ncp, k, D, bad);
        if (nb > 1) hipLaunchKernelGGL(k_inv_step, dim3(nb - 1), dim3(256), 0, s, A, ncp, nb, k, D, 0);
        if (nb > 1) hipLaunchKernelGGL(k_inv_step, dim3(nb - 1, nb - 1), dim3(256), 0, s, A, ncp, nb, k, D, 2);
        hipLaunchKernelGGL(k_inv_step, dim3(nb), dim3(256), 0, s, A, ncp, nb, k, D, 1);
    }
}

// P^T r comes from the kernels that make r (k_pcg_init, k_pcg_update): a block of PCG_UPD_TPB scalars is PCG_UPD_KF consecutive
// keyframes, which touch at most PCG_RSLOTS consecutive aggregates (the first is pcg_hat(first keyframe).i0); thread (slot, d) walks
// the block's keyframes in order and leaves its partial sum in rpart[block][slot][d].  k_pcg_coarse adds the two or three blocks
// of an aggregate in block order: fixed summation order, no extra launch (a kernel of its own took 15 us per PCG iteration).
#define PCG_UPD_TPB 192       // 4 clusters of PCG_CN scalars: a cluster never straddles two blocks
#define PCG_UPD_KF (PCG_UPD_TPB / 6)
#define PCG_RSLOTS 4          // PCG_UPD_KF / (aggregate >= 16 keyframes) + 2
__device__ __forceinline__ void pcg_block_restrict(const double* rs, int nfree, int nagg, const double* __restrict__ svec, const double* __restrict__ cen,
                                                   double* __restrict__ rpart)
{
    if (!rpart) return;                                            // (uniform: no coarse level in this solve)
    // every thread (keyframe k of the block, component d) weights its residual for the keyframe's two aggregates; the thread of
    // component 3 also forms the scale products; then thread (slot, d) adds the block's keyframes in order
    __shared__ double cw[2][PCG_UPD_KF][PCG_CDOF];
    __shared__ int ci0[PCG_UPD_KF], ci1[PCG_UPD_KF];
    const int A = PCG_CL * pcg_agg_clusters(nfree);
    const int f0 = blockIdx.x * PCG_UPD_KF;
    {
        const int k = threadIdx.x / 6, d = threadIdx.x - 6 * k, f = f0 + k;
        if (f < nfree) {
            const PcgHat h = pcg_hat(f, A, nagg);
            const double r = rs[threadIdx.x];
            cw[0][k][d] = h.w0 * r; cw[1][k][d] = h.w1 * r;
            if (d == 3) {
                const double* t = svec + 3LL * f; const double* c0 = cen + 3 * h.i0; const double* c1 = cen + 3 * h.i1;
                const double* rf = rs + 6 * k + 3;
                cw[0][k][6] = h.w0 * (((t[0] - c0[0]) * rf[0] + (t[1] - c0[1]) * rf[1]) + (t[2] - c0[2]) * rf[2]);
                cw[1][k][6] = h.w1 * (((t[0] - c1[0]) * rf[0] + (t[1] - c1[1]) * rf[1]) + (t[2] - c1[2]) * rf[2]);
                ci0[k] = h.i0; ci1[k] = h.i1;
            }
        } else if (d == 3) { ci0[k] = -1; ci1[k] = -1; }
    }
    __syncthreads();
    if (threadIdx.x < PCG_RSLOTS * PCG_CDOF) {
        const int slot = threadIdx.x / PCG_CDOF, d = threadIdx.x - PCG_CDOF * slot;
        const int I = pcg_hat(f0, A, nagg).i0 + slot;
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < PCG_UPD_KF; k++) {                       // branch-free, so that the LDS reads of several keyframes are in flight (x + 0.0 == x)
            const int j0 = ci0[k], j1 = ci1[k];
            const double a0 = cw[0][k][d], a1 = cw[1][k][d];
            s += j0 == I ? a0 : 0.0;
            s += (j1 == I && j1 != j0) ? a1 : 0.0;
        }
        rpart[((long long)blockIdx.x * PCG_RSLOTS + slot) * PCG_CDOF + d] = s;
    }
}
// yc = Ac^-1 (P^T r), one wave per row; cpart = the workgroup's share of (P^T r) . yc  (= r . (P yc), the coarse part of r.z)
__global__ __launch_bounds__(256) void k_pcg_coarse(const double* __restrict__ Aci, int nc, int ncp, const double* __restrict__ rpart, int nfree,
                                                    double* __restrict__ yc, double* __restrict__ cpart)
{
    extern __shared__ double rc[];
    __shared__ double dots[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, row = blockIdx.x * 4 + wv;
    // the row of the inverse is requested first: its latency passes under the copy of P^T r
    constexpr int PRE = 24;                                   // 64 * 24 = 1536 columns in registers, the rest (larger maps) afterwards
    double av[PRE];
    const double* A = Aci + (long long)min(row, nc - 1) * ncp;
#pragma unroll
    for (int q = 0; q < PRE; q++) { const int c = lane + 64 * q; av[q] = c < nc ? A[c] : 0.0; }
    {
        const int A = PCG_CL * pcg_agg_clusters(nfree), nagg = nc / PCG_CDOF;
        for (int i = threadIdx.x; i < nc; i += 256) {
            const int I = i / PCG_CDOF, d = i - PCG_CDOF * I;
            int f0, f1;
            pcg_hat_support(I, A, nfree, f0, f1);
            double sum = 0.0;
            for (int b = f0 / PCG_UPD_KF; b <= (f1 - 1) / PCG_UPD_KF; b++) {            // the blocks of k_pcg_update that hold keyframes of I
                const int slot = I - pcg_hat(b * PCG_UPD_KF, A, nagg).i0;
                if (slot >= 0 && slot < PCG_RSLOTS) sum += rpart[((long long)b * PCG_RSLOTS + slot) * PCG_CDOF + d];
            }
            rc[i] = sum;
        }
    }
    __syncthreads();
    double s = 0;
    if (row < nc) {
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int c = lane + 64 * q; if (c < nc) s += av[q] * rc[c]; }
        for (int c = lane + 64 * PRE; c < nc; c += 64) s += A[c] * rc[c];
    }
    for (int st = 32; st >= 1; st >>= 1) s += __shfl_xor(s, st, 64);
    if (lane == 0) { if (row < nc) yc[row] = s; dots[wv] = row < nc ? s * rc[row] : 0.0; }
    __syncthreads();
    if (threadIdx.x == 0) cpart[blockIdx.x] = ((dots[0] + dots[1]) + dots[2]) + dots[3];
}
// The new search direction, one thread per scalar unknown:  p = (z + P yc) + beta p,  beta = r.z (now) / r.z (previous iteration).
// Every workgroup re-reduces the partial sums of r.z (cluster part from k_pcg_init / k_pcg_update, coarse part from k_pcg_coarse) in
// the same fixed order; the previous value comes from the alternating slot sc[8 + (parity ^ 1)], and block 0 leaves the current
// one in sc[8 + parity] for k_pcg_update and the next iteration, together with the scalars the host looks at.  yc == nullptr: no
// coarse level in this solve.
__global__ __launch_bounds__(256) void k_pcg_direction(const double* __restrict__ yc, const double* __restrict__ svec, const double* __restrict__ cen,
                                                       int nfree, int nagg, double* __restrict__ w, int nblk_part, const double* __restrict__ part,
                                                       const double* __restrict__ cpart, int ncpart, double* __restrict__ sc, int parity)
{
    __shared__ double s_beta;
    if (threadIdx.x < 64) {
        double rz = 0, rr = 0;
        for (int i = threadIdx.x; i < nblk_part; i += 64) { rz += part[3 * i]; rr += part[3 * i + 1]; }
        double rzc = 0;
        for (int i = threadIdx.x; i < ncpart; i += 64) rzc += cpart[i];
        for (int st = 32; st >= 1; st >>= 1) { rz += __shfl_xor(rz, st, 64); rr += __shfl_xor(rr, st, 64); rzc += __shfl_xor(rzc, st, 64); }
        rz += rzc;
        if (threadIdx.x == 0) {
            const double rz_prev = sc[8 + (parity ^ 1)];
            s_beta = rz_prev > 0.0 ? rz / rz_prev : 0.0;
            if (blockIdx.x == 0) {                                 // scalars for k_pcg_update, the next iteration and the host
                const double pap = part[2];
                sc[8 + parity] = rz;
                sc[0] = rz; sc[2] = rr; if (pap < sc[3]) sc[3] = pap; sc[4] += 1.0;
            }
        }
    }
    __syncthreads();
    const long long n = 6LL * nfree;
    const long long o = blockIdx.x * 256LL + threadIdx.x;
    if (o >= n) return;
    double zf = w[2 * n + o];
    if (yc) {
        const int f = (int)(o / 6), d = (int)(o - 6LL * f);
        const PcgHat h = pcg_hat(f, PCG_CL * pcg_agg_clusters(nfree), nagg);
        const double* y0 = yc + PCG_CDOF * h.i0; const double* y1 = yc + PCG_CDOF * h.i1;
        double v = h.w0 * y0[d] + h.w1 * y1[d];
        if (d >= 3) {
            const double t = svec[3LL * f + (d - 3)];
            v += h.w0 * ((t - cen[3 * h.i0 + (d - 3)]) * y0[6]) + h.w1 * ((t - cen[3 * h.i1 + (d - 3)]) * y1[6]);
        }
        zf += v;
    }
    w[3 * n + o] = zf + s_beta * w[3 * n + o];
}

// state vector layout in `w`: x | r | z (cluster level only) | p | Ap  (each n doubles); scalars in sc[]:
//   sc[0] rz, sc[1] |b|^2, sc[2] |r|^2, sc[3] min p.Ap seen, sc[4] iterations
static_assert(PCG_UPD_TPB % PCG_CN == 0, "a block must hold whole clusters");
__global__ __launch_bounds__(PCG_UPD_TPB) void k_pcg_init(const double* __restrict__ b, const double* __restrict__ Minv, int nfree,
                                                          double* __restrict__ w, double* __restrict__ part, int nagg, const double* __restrict__ svec,
                                                          const double* __restrict__ cen, double* __restrict__ rpart)
{
    __shared__ double rs[PCG_UPD_TPB];
    __shared__ double red[2][3];
    const long long n = 6LL * nfree;
    const long long o = (long long)blockIdx.x * PCG_UPD_TPB + threadIdx.x;
    const double ri = o < n ? b[o] : 0.0;
    rs[threadIdx.x] = ri;
    __syncthreads();
    pcg_block_restrict(rs, nfree, nagg, svec, cen, rpart);
    double rz = 0, bb = 0;
    if (o < n) {
        const int cl = (int)(o / PCG_CN), li = (int)(o - (long long)cl * PCG_CN), base = (threadIdx.x / PCG_CN) * PCG_CN;
        const double* M = Minv + (long long)cl * PCG_CN * PCG_CN + li;
        double z = 0;
#pragma unroll 8
        for (int k = 0; k < PCG_CN; k++) z += M[k * PCG_CN] * rs[base + k];       // symmetric: column li read with unit stride across lanes
        w[o] = 0.0; w[n + o] = ri; w[2 * n + o] = z; w[3 * n + o] = 0.0;                         // p = z + beta * 0 in the first direction
        rz = ri * z; bb = ri * ri;
    }
    for (int s = 32; s >= 1; s >>= 1) { rz += __shfl_xor(rz, s, 64); bb += __shfl_xor(bb, s, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = rz; red[1][threadIdx.x >> 6] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {                   // same layout as k_pcg_update: r.z, |r|^2, p.Ap
        part[3 * blockIdx.x] = (red[0][0] + red[0][1]) + red[0][2];
        part[3 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + red[1][2];
        part[3 * blockIdx.x + 2] = 1e300;
    }
}
__global__ __launch_bounds__(64) void k_pcg_init_fin(const double* __restrict__ part, int nblk, const double* __restrict__ cpart, int ncpart,
                                                     double* __restrict__ sc)
{
    if (threadIdx.x != 0) return;
    double rz = 0, bb = 0, rzc = 0;
    for (int i = 0; i < nblk; i++) { rz += part[3 * i]; bb += part[3 * i + 1]; }
    for (int i = 0; i < ncpart; i++) rzc += cpart[i];
    rz += rzc;
    sc[0] = rz; sc[1] = bb; sc[2] = bb; sc[3] = 1e300; sc[4] = -1.0; sc[8] = rz; sc[9] = rz;      // (the direction call of the start-up adds 1)
}

// Ap = A p, one workgroup per block row (p comes from k_pcg_direction).
// Thread = (entry slot 0..41, row component 0..5); the 42 slot sums of a component are added in slot order by one lane.
__global__ __launch_bounds__(256) void k_pcg_spmv(const double* __restrict__ Hb, const int* __restrict__ row_ptr, const unsigned* __restrict__ ent_key,
                                                  const unsigned* __restrict__ ent_val, int nfree, double* __restrict__ w, double* __restrict__ pap_part)
{
    __shared__ double red[42][6];
    // Workgroups are dealt round-robin to the 8 XCDs, so b and b + 8 share an L2.  Giving each XCD a contiguous range of
    // block rows means that block (i, j) of the band, needed by row i and (transposed) by row j a few rows later, is
    // fetched from the fabric once and found in that XCD's L2 the second time.
    const int rows_per_xcd = (nfree + PCG_XCDS - 1) / PCG_XCDS;
    const int row = ((int)blockIdx.x % PCG_XCDS) * rows_per_xcd + (int)blockIdx.x / PCG_XCDS;
    if (row >= nfree) return;
    const long long n = 6LL * nfree;
    const double* p = w + 3 * n;
    const int slot = threadIdx.x / 6, r = threadIdx.x - 6 * slot;
    if (slot < 42) {
        double acc = 0;
        const int k_end = row_ptr[2 * row + 1];
        // four entries per step: all index loads are issued first, then all block / vector loads, so a row of
        // up to 168 blocks costs two dependent memory round trips instead of eight
        for (int k0 = row_ptr[2 * row] + slot; k0 < k_end; k0 += 4 * 42) {
            unsigned v[4]; int col[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int k = k0 + 42 * q;
                v[q] = 0xFFFFFFFFu; col[q] = 0;
                if (k < k_end) { v[q] = ent_val[k]; col[q] = (int)(ent_key[k] - (unsigned)row * (unsigned)nfree); }
            }
            double bv[4][6], xv[4][6];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (v[q] == 0xFFFFFFFFu) {
#pragma unroll
                    for (int c = 0; c < 6; c++) { bv[q][c] = 0; xv[q][c] = 0; }
                    continue;
                }
                const double* B = Hb + 36 * (long long)(v[q] & 0x7FFFFFFFu);
                const double* pc = p + 6 * (long long)col[q];
                const bool tr = (v[q] & 0x80000000u) != 0u;
#pragma unroll
                for (int c = 0; c < 6; c++) { bv[q][c] = tr ? B[c * 6 + r] : B[r * 6 + c]; xv[q][c] = pc[c]; }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 6; c++) acc += bv[q][c] * xv[q][c];
        }
        red[slot][r] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double tot = 0, pap = 0;
        if (threadIdx.x < 6) {
            for (int s2 = 0; s2 < 42; s2++) tot += red[s2][threadIdx.x];
            const long long o = 6LL * row + threadIdx.x;
            w[4 * n + o] = tot;
            pap = tot * p[o];
        }
        for (int st = 4; st >= 1; st >>= 1) pap += __shfl_xor(pap, st, 64);     // lanes 0..7 (6,7 hold 0)
        if (threadIdx.x == 0) pap_part[row] = pap;
    }
}

// x += alpha p; r -= alpha Ap; z = Minv r (cluster mat-vec); partial r.z and r.r.  One thread per scalar unknown, a
// block holds 4 whole clusters whose new residuals are shared through LDS.  Every block re-reduces p.Ap itself.
__global__ __launch_bounds__(PCG_UPD_TPB) void k_pcg_update(const double* __restrict__ Minv, int nfree, double* __restrict__ w,
                                                            const double* __restrict__ pap_part, const double* __restrict__ sc, double* __restrict__ part,
                                                            int parity, int nagg, const double* __restrict__ svec, const double* __restrict__ cen,
                                                            double* __restrict__ rpart)
{
    __shared__ double red[4];
    __shared__ double red2[2][3];
    __shared__ double rs[PCG_UPD_TPB];
    const long long n = 6LL * nfree;
    const long long o = (long long)blockIdx.x * PCG_UPD_TPB + threadIdx.x;
    // everything this thread needs is requested before the reduction, so that all global loads overlap
    double r_old = 0, ap = 0, x_old = 0, p_old = 0;
    double mv[PCG_CN];                                              // this unknown's column of its cluster inverse: 48 loads that wait for nothing
    if (o < n) {
        r_old = w[n + o]; ap = w[4 * n + o]; x_old = w[o]; p_old = w[3 * n + o];   // this iteration's direction
        const int cl = (int)(o / PCG_CN), li = (int)(o - (long long)cl * PCG_CN);
        const double* M = Minv + (long long)cl * PCG_CN * PCG_CN + li;
#pragma unroll
        for (int k = 0; k < PCG_CN; k++) mv[k] = M[k * PCG_CN];    // symmetric: column li read with unit stride across lanes
    }
    const double rz_old = sc[8 + (parity ^ 1)];                     // this iteration's r.z, left there by k_pcg_direction
    double s = 0;
#pragma unroll 4
    for (int k = threadIdx.x; k < nfree; k += PCG_UPD_TPB) s += pap_part[k];
    for (int st = 32; st >= 1; st >>= 1) s += __shfl_xor(s, st, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double pap = (red[0] + red[1]) + red[2];
    const double alpha = pap > 0.0 ? rz_old / pap : 0.0;
    const double ri = r_old - alpha * ap;
    rs[threadIdx.x] = ri;
    __syncthreads();
    pcg_block_restrict(rs, nfree, nagg, svec, cen, rpart);
    double rz = 0, rr = 0;
    if (o < n) {
        const int base = (threadIdx.x / PCG_CN) * PCG_CN;
        double z = 0;
#pragma unroll
        for (int k = 0; k < PCG_CN; k++) z += mv[k] * rs[base + k];
        w[o] = x_old + alpha * p_old; w[n + o] = ri; w[2 * n + o] = z;
        rz = ri * z; rr = ri * ri;
    }
    for (int st = 32; st >= 1; st >>= 1) { rz += __shfl_xor(rz, st, 64); rr += __shfl_xor(rr, st, 64); }
    if ((threadIdx.x & 63) == 0) { red2[0][threadIdx.x >> 6] = rz; red2[1][threadIdx.x >> 6] = rr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = (red2[0][0] + red2[0][1]) + red2[0][2];
        part[3 * blockIdx.x + 1] = (red2[1][0] + red2[1][1]) + red2[1][2];
        part[3 * blockIdx.x + 2] = pap;
    }
}

__global__ __launch_bounds__(64) void k_pcg_scalars(int nblk, const double* __restrict__ part, const double* __restrict__ cpart, int ncpart,
                                                    double* __restrict__ sc)
{
    // the same sums in the same order as a workgroup of k_pcg_spmv forms them (one thread walking the ~300 partials took 20 us)
    double rz = 0, rr = 0, rzc = 0;
    for (int i = threadIdx.x; i < nblk; i += 64) { rz += part[3 * i]; rr += part[3 * i + 1]; }
    for (int i = threadIdx.x; i < ncpart; i += 64) rzc += cpart[i];
    for (int st = 32; st >= 1; st >>= 1) { rz += __shfl_xor(rz, st, 64); rr += __shfl_xor(rr, st, 64); rzc += __shfl_xor(rzc, st, 64); }
    if (threadIdx.x != 0) return;
    rz += rzc;
    const double pap = part[2];
    sc[0] = rz; sc[2] = rr; if (pap < sc[3]) sc[3] = pap; sc[4] += 1.0;
}

// ---------------------------------------------------------------------------------------- pipelined PCG (round 3)
// The classic iteration above is four dependent grid-wide steps (mat-vec | p.Ap -> update, P^T r | coarse mat-vec | prolongation,
// direction): the two dot products and the two halves of the coarse level each force a kernel boundary, and at 2000 keyframes
// every one of the four kernels is a latency floor (5-15 us), not a bandwidth problem.  The pipelined form of Ghysels & Vanroose
// (Parallel Computing 40, 2014, Alg. 4) applies the preconditioner and the matrix to w = A u instead of r, which decouples them
// from the dot products:
//     gamma = (r, u), delta = (w, u)                      (partials left by the previous iteration's row kernel)
//     m = M^-1 w;  nn = A m
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)          (first iteration: beta = 0, alpha = gamma / delta)
//     z = nn + beta z;  q = m + beta q;  s = w + beta s;  p = u + beta p
//     x += alpha p;  r -= alpha s;  u -= alpha q;  w -= alpha z
// which is TWO kernels per iteration:
//   k_ppcg_prec  one workgroup per 32 keyframes: m = Minv_cluster w + P Aci (P^T w).  Every workgroup sums the per-keyframe
//                contributions to P^T w (left by the row kernel in `CA`, one coalesced pass) and multiplies the rows of Aci that its
//                own <= PCG_PSLOTS aggregates need -- redundant work out of L2 instead of two more grid-wide steps.  One extra
//                workgroup reduces the dot-product partials and publishes alpha, beta for the row kernel.
//   k_ppcg_row   one workgroup per block row: nn = A m from the row-contiguous copy of the matrix (`Hf`: both triangles, entries of
//                a row side by side, transposition done once per LM trial by k_ppcg_expand instead of in every iteration), then
//                the eight vector updates of its six unknowns, the partial dot products and the contributions to P^T w_new.
// Measured on config 5 (optimize(20), 518 iterations): 27.3 us per iteration inside a graph (row kernel 14.5: it streams the 66 MB of
// Hf at 5.3 TB/s out of the Infinity Cache; k_ppcg_prec 12.8: 0.5 MB per workgroup through ONE CU's path to L2) against 37 us for the
// four classic kernels.  A variant with one workgroup per CLUSTER of 8 keyframes (mat-vec, updates, cluster level of M^-1 and the
// cluster's pre-summed share of P^T w in one kernel, which shrinks what the second kernel reads from 0.5 MB to 0.18 MB per
// workgroup) was built and measured slower: 23.3 + 8.2 us -- 250 workgroups of 1024 threads leave one workgroup per CU, and
// nothing overlaps its barriers and its serial tail, where four 256-thread workgroups per CU overlap each other.  Two rows per
// 512-thread workgroup with their contributions to P^T w added up before they are stored (half the CA rows for k_ppcg_prec to
// read) lost for the same reason: 19.4 against 17.7 ms of solve time per optimize(20).  A stop flag on the device (kernels return at
// once when |r|^2 has reached the tolerance, the host keeps one chunk of iterations enqueued ahead of what it has seen: no GPU idle
// time in the host's round trips, exact iteration counts 498 instead of 518) measured the same 17.6-18.2 ms: the round trips are
// not what costs, as round 2 had found for the classic iteration.
// In exact arithmetic the iterates are those of the classic method; in floating point the recurrences for u and w drift, and the
// true residual stalls near 1e-9 |b| (tools/gba_pipelined_study.py: identical iteration counts at 1e-6 and 1e-9 on a late trial
// of config 5, floor 2e-9) -- so solves asked for more than 1e-7 keep the classic kernels (ba_host.cpp decides).
// Vector slots of the state buffer (n doubles each):
enum { PV_X = 0, PV_R, PV_U, PV_W, PV_P, PV_S, PV_Q, PV_Z, PV_M, PV_COUNT };
// scalars: sc[0] gamma, sc[1] |b|^2, sc[2] |r|^2, sc[3] min (p, A p) seen, sc[4] iterations, sc[5] alpha, sc[6] beta

// Hf[k][36] = block of entry k of the symmetric row lists, as the row sees it (transposed where the stored block is the mirror
// image); ecol[k] = its block column.  One thread per element; padding entries (sorted to the end) are skipped.
__global__ __launch_bounds__(256) void k_ppcg_expand(const double* __restrict__ Hb, const unsigned* __restrict__ ent_key, const unsigned* __restrict__ ent_val,
                                                     int n_ent, int nfree, double* __restrict__ Hf, int* __restrict__ ecol)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= 36LL * n_ent) return;
    const int k = (int)(i / 36), e = (int)(i - 36LL * k), r = e / 6, c = e - 6 * r;
    const unsigned key = ent_key[k];
    if (key == 0xFFFFFFFFu) return;
    const unsigned v = ent_val[k];
    const double* B = Hb + 36LL * (v & 0x7FFFFFFFu);
    Hf[i] = (v & 0x80000000u) ? B[c * 6 + r] : B[r * 6 + c];
    if (e == 0) ecol[k] = (int)(key % (unsigned)nfree);
}

// contribution of keyframe f's six values v[0..5] (lanes d = 0..5 hold v[d], lane 6 forms the scale product) to the restricted
// vector: CA[s][7 I + d], s = f - first keyframe of aggregate I's support; called by 7 consecutive lanes with d = 0..6
__device__ __forceinline__ void ppcg_contribute(int f, int d, double vd, double v3, double v4, double v5, int nfree, int nagg, int ncp,
                                                const double* __restrict__ svec, const double* __restrict__ cen, double* __restrict__ CA)
{
    const int A = PCG_CL * pcg_agg_clusters(nfree);
    const PcgHat h = pcg_hat(f, A, nagg);
    double t0 = 0, t1 = 0, t2 = 0;
    if (d == 6) { t0 = svec[3LL * f]; t1 = svec[3LL * f + 1]; t2 = svec[3LL * f + 2]; }
    {
        const int first = max(0, A * h.i0 - A / 2);
        double val = vd;
        if (d == 6) { const double* c = cen + 3 * h.i0; val = ((t0 - c[0]) * v3 + (t1 - c[1]) * v4) + (t2 - c[2]) * v5; }
        CA[(long long)(f - first) * ncp + PCG_CDOF * h.i0 + d] = h.w0 * val;
    }
    if (h.i1 != h.i0) {
        const int first = max(0, A * h.i1 - A / 2);
        double val = vd;
        if (d == 6) { const double* c = cen + 3 * h.i1; val = ((t0 - c[0]) * v3 + (t1 - c[1]) * v4) + (t2 - c[2]) * v5; }
        CA[(long long)(f - first) * ncp + PCG_CDOF * h.i1 + d] = h.w1 * val;
    }
}

// start of a solve: x = 0, r = b, the direction vectors zero, P^T r contributions; the last workgroup sums |b|^2 and sets the scalars
__global__ __launch_bounds__(256) void k_ppcg_init(const double* __restrict__ b, int nfree, double* __restrict__ wb, int nagg, int ncp,
                                                   const double* __restrict__ svec, const double* __restrict__ cen, double* __restrict__ CA, double* __restrict__ sc)
{
    const long long n = 6LL * nfree;
    const int nb32 = (nfree + PCG_UPD_KF - 1) / PCG_UPD_KF;
    if ((int)blockIdx.x == nb32) {
        __shared__ double red[4];
        double s = 0;
#pragma unroll 8
        for (long long i = threadIdx.x; i < n; i += 256) s += b[i] * b[i];
        for (int st = 32; st >= 1; st >>= 1) s += __shfl_xor(s, st, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double bb = ((red[0] + red[1]) + red[2]) + red[3];
            sc[0] = 0.0; sc[1] = bb; sc[2] = bb; sc[3] = 1e300; sc[4] = 0.0; sc[5] = 1.0; sc[6] = 0.0;
        }
        return;
    }
    __shared__ double rs[PCG_UPD_TPB];
    const long long o = (long long)blockIdx.x * PCG_UPD_TPB + threadIdx.x;
    if (threadIdx.x < PCG_UPD_TPB) {
        const double ri = o < n ? b[o] : 0.0;
        rs[threadIdx.x] = ri;
        if (o < n) {
            wb[PV_X * n + o] = 0.0; wb[PV_R * n + o] = ri; wb[PV_P * n + o] = 0.0; wb[PV_S * n + o] = 0.0; wb[PV_Q * n + o] = 0.0; wb[PV_Z * n + o] = 0.0;
        }
    }
    __syncthreads();
    if (CA && threadIdx.x < PCG_UPD_KF * PCG_CDOF) {
        const int k = threadIdx.x / PCG_CDOF, d = threadIdx.x - PCG_CDOF * k, f = blockIdx.x * PCG_UPD_KF + k;
        if (f < nfree) ppcg_contribute(f, d, d < 6 ? rs[6 * k + d] : 0.0, rs[6 * k + 3], rs[6 * k + 4], rs[6 * k + 5], nfree, nagg, ncp, svec, cen, CA);
    }
}

// out = M^-1 in  (vector slots of wb);  Aci == nullptr: cluster level alone.  The extra last workgroup (do_scalars) turns the
// row kernel's partials into gamma, delta, |r|^2 and publishes alpha, beta.
// The kernel is a latency problem (64 workgroups, ~0.5 MB each out of L2): a workgroup has 1024 threads and EVERY global load of a
// thread is requested before the first is used -- the rows of Aci (they do not depend on P^T in), the columns of the cluster
// inverses and the CA rows -- so the whole preconditioner costs about two L2 round trips plus the time 0.5 MB take through one CU's
// path to L2.  Partial sums meet in LDS and are added in a fixed order.
#ifndef PP_KF
#define PP_KF 8                                                // keyframes per workgroup of k_ppcg_prec: one cluster.  Measured at config 5 (solve phase of optimize(20),
                                                               // same-box A/B): 32 keyframes (63 workgroups, 28 rows of the inverse each) 17.72 ms, 16: 17.72, 8 (250, 14 rows): 17.35
#endif
#define PP_UTPB (6 * PP_KF)
#define PCG_PSLOTS (PP_KF / (PCG_CL * PCG_AGG) + 2)            // aggregates a block of PP_KF keyframes can touch
static_assert(PP_UTPB % PCG_CN == 0, "a block must hold whole clusters");
#define PP_TPB 1024
#define PP_NW (PP_TPB / 64)
#define PP_CQ 4                                                // a column of CA is summed by PP_CQ threads (a quarter of the support rows each)
#define PP_MQ 4                                                // an unknown's cluster product by PP_MQ threads (PCG_CN / PP_MQ terms each)
#define PP_COLS 4                                              // columns of CA per thread: PP_COLS * (PP_TPB / PP_CQ) >= coarse pitch
#define PP_D2 8                                                // 16-byte pieces of an Aci row per lane: 128 * PP_D2 >= coarse pitch
#define PP_ROWS_PER_WAVE ((PCG_PSLOTS * PCG_CDOF + PP_NW - 1) / PP_NW)
static_assert(PCG_CN % PP_MQ == 0 && PP_MQ * PP_UTPB <= PP_TPB, "cluster product split");
__host__ __device__ inline bool ppcg_prec_fits(int ncp, int agg_keyframes)
{ return ncp <= PP_COLS * (PP_TPB / PP_CQ) && ncp <= 128 * PP_D2 && (2 * agg_keyframes) % PP_CQ == 0 && PP_KF / agg_keyframes + 2 <= PCG_PSLOTS; }
__global__ __launch_bounds__(PP_TPB) void k_ppcg_prec(const double* __restrict__ Minv, int nfree, double* __restrict__ wb, int in_slot, int out_slot,
                                                      const double* __restrict__ Aci, int nc, int ncp, const double* __restrict__ CA, int nagg,
                                                      const double* __restrict__ svec, const double* __restrict__ cen,
                                                      const double* __restrict__ part, double* __restrict__ sc, int do_scalars)
{
    const long long n = 6LL * nfree;
    const int nb32 = (nfree + PP_KF - 1) / PP_KF;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x == nb32) {
        if (!do_scalars) return;
        __shared__ double red[3][PP_NW];
        double g = 0, dl = 0, rr = 0;
        for (int i = threadIdx.x; i < nfree; i += PP_TPB) { g += part[3 * i]; dl += part[3 * i + 1]; rr += part[3 * i + 2]; }
        for (int st = 32; st >= 1; st >>= 1) { g += __shfl_xor(g, st, 64); dl += __shfl_xor(dl, st, 64); rr += __shfl_xor(rr, st, 64); }
        if (lane == 0) { red[0][wv] = g; red[1][wv] = dl; red[2][wv] = rr; }
        __syncthreads();
        if (threadIdx.x == 0) {
            g = dl = rr = 0;
            for (int w2 = 0; w2 < PP_NW; w2++) { g += red[0][w2]; dl += red[1][w2]; rr += red[2][w2]; }          // wave order: fixed
            const double it = sc[4], g_old = sc[0], a_old = sc[5];
            double beta = 0.0, den = dl;
            if (it > 0.0) { beta = g_old != 0.0 ? g / g_old : 0.0; den = dl - beta * g / a_old; }
            // den = (p, A p) in exact arithmetic: not positive = the system (or the preconditioner) is not positive definite, or the
            // recurrences have broken down; the host looks at sc[3] and abandons the solve
            double alpha = den > 0.0 ? g / den : 0.0;
            if (!(g > 0.0) && rr > 0.0) den = -1.0;
            if (!(den >= sc[3])) sc[3] = den;                    // (also catches NaN)
            if (alpha == 0.0) alpha = 1e-300;                     // keeps the next iteration's division finite; the solve is abandoned anyway
            sc[0] = g; sc[2] = rr; sc[4] = it + 1.0; sc[5] = alpha; sc[6] = beta;
        }
        return;
    }
    extern __shared__ double pp_lds[];                     // rc[ncp] (P^T in), cpart[PP_CQ][ncp]
    __shared__ double ws[PP_UTPB], mpart[PP_MQ][PP_UTPB], yl[PCG_PSLOTS * PCG_CDOF];
    double* rc = pp_lds;
    double* cpart = pp_lds + ncp;
    const int f0 = blockIdx.x * PP_KF;
    const int A = PCG_CL * pcg_agg_clusters(nfree);
    typedef double pp_d2 __attribute__((ext_vector_type(2)));
    // ---- every load of this thread, requested back to back
    // (1) rows of Aci: wave wv takes rows wv, wv + PP_NW, ... of the block's 7 * nsl
    int ibase = 0, nrows = 0;
    pp_d2 arow[PP_ROWS_PER_WAVE][PP_D2];
    if (Aci) {
        ibase = pcg_hat(f0, A, nagg).i0;
        nrows = PCG_CDOF * min(PCG_PSLOTS, nagg - ibase);
#pragma unroll
        for (int q = 0; q < PP_ROWS_PER_WAVE; q++) {
            const int row = min(wv + PP_NW * q, nrows - 1);
            const pp_d2* R = reinterpret_cast<const pp_d2*>(Aci + (long long)(PCG_CDOF * ibase + row) * ncp);
#pragma unroll
            for (int k = 0; k < PP_D2; k++) { const int c2 = lane + 64 * k; arow[q][k] = 2 * c2 < ncp ? R[c2] : pp_d2{0.0, 0.0}; }
        }
    }
    // (2) cluster inverse: thread (unknown u, quarter mq) takes PCG_CN / PP_MQ terms of the unknown's column
    const int mu = threadIdx.x % PP_UTPB, mq = threadIdx.x / PP_UTPB;
    const long long mo = (long long)f0 * 6 + mu;
    const bool mlive = mq < PP_MQ && mo < n;
    double mv[PCG_CN / PP_MQ];
    if (mlive) {
        const int cl = (int)(mo / PCG_CN), li = (int)(mo - (long long)cl * PCG_CN);
        const double* M = Minv + (long long)cl * PCG_CN * PCG_CN + li + (long long)(mq * (PCG_CN / PP_MQ)) * PCG_CN;
#pragma unroll
        for (int k = 0; k < PCG_CN / PP_MQ; k++) mv[k] = M[k * PCG_CN];          // symmetric: column li read with unit stride across lanes
    }
    if (threadIdx.x < PP_UTPB) ws[threadIdx.x] = mo < n ? wb[(long long)in_slot * n + mo] : 0.0;
    // (3) P^T in: thread (column group, quarter cq) sums its quarter of the 2 A support rows of up to PP_COLS columns
    if (Aci) {
        const int cq = threadIdx.x / (PP_TPB / PP_CQ), cj = threadIdx.x % (PP_TPB / PP_CQ);
        const int rq = (2 * A) / PP_CQ;                                           // rows per quarter
        double cs[PP_COLS];
#pragma unroll
        for (int k = 0; k < PP_COLS; k++) {
            const int j = cj + (PP_TPB / PP_CQ) * k;
            double s = 0.0;
            if (j < ncp) {
                const double* col = CA + (long long)(cq * rq) * ncp + j;
#pragma unroll 8
                for (int q = 0; q < rq; q++) s += col[(long long)q * ncp];
            }
            cs[k] = s;
        }
#pragma unroll
        for (int k = 0; k < PP_COLS; k++) { const int j = cj + (PP_TPB / PP_CQ) * k; if (j < ncp) cpart[cq * ncp + j] = cs[k]; }
    }
    __syncthreads();
    if (Aci)
        for (int j = threadIdx.x; j < ncp; j += PP_TPB) rc[j] = j < nc ? ((cpart[j] + cpart[ncp + j]) + cpart[2 * ncp + j]) + cpart[3 * ncp + j] : 0.0;      // quarters in row order: ascending keyframe
    if (mlive) {
        const int base = (mu / PCG_CN) * PCG_CN + mq * (PCG_CN / PP_MQ);
        double z = 0.0;
#pragma unroll
        for (int k = 0; k < PCG_CN / PP_MQ; k++) z += mv[k] * ws[base + k];
        mpart[mq][mu] = z;
    }
    __syncthreads();
    if (Aci) {
#pragma unroll
        for (int q = 0; q < PP_ROWS_PER_WAVE; q++) {
            const int row = wv + PP_NW * q;
            double sa = 0.0;
#pragma unroll
            for (int k = 0; k < PP_D2; k++) { const int c2 = lane + 64 * k; if (2 * c2 < ncp) sa += arow[q][k].x * rc[2 * c2] + arow[q][k].y * rc[2 * c2 + 1]; }
            for (int st = 32; st >= 1; st >>= 1) sa += __shfl_xor(sa, st, 64);
            if (lane == 0 && row < nrows) yl[row] = sa;
        }
        __syncthreads();
    }
    if (threadIdx.x < PP_UTPB && mo < n) {
        double z = ((mpart[0][mu] + mpart[1][mu]) + mpart[2][mu]) + mpart[3][mu];
        if (Aci) {
            const int f = (int)(mo / 6), d = (int)(mo - 6LL * f);
            const PcgHat h = pcg_hat(f, A, nagg);
            const double* y0 = yl + PCG_CDOF * (h.i0 - ibase); const double* y1 = yl + PCG_CDOF * (h.i1 - ibase);
            double v = h.w0 * y0[d] + h.w1 * y1[d];
            if (d >= 3) {
                const double t = svec[3LL * f + (d - 3)];
                v += h.w0 * ((t - cen[3 * h.i0 + (d - 3)]) * y0[6]) + h.w1 * ((t - cen[3 * h.i1 + (d - 3)]) * y1[6]);
            }
            z += v;
        }
        wb[(long long)out_slot * n + mo] = z;
    }
}

// One workgroup per block row.  MODE 0 (start of a solve): w = A u, partials of (r, u), (w, u), (r, r), contributions of w.
// MODE 1: nn = A m, then the recurrences above for the row's six unknowns, the same partials and contributions for the new r, u, w.
// EPS = entries per thread and step: 4 (<= 128 registers, 4 workgroups per CU: two rounds of workgroups at 2000 keyframes) or
// 2 (<= 64 registers: every workgroup of a 2000-keyframe map resident at once, two steps for the average row)
template <int MODE, int EPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EPS == 2 ? 8 : 4, EPS == 2 ? 8 : 4))) void k_ppcg_row(
    const double* __restrict__ Hf, const int* __restrict__ ecol, const int* __restrict__ row_ptr, int nfree, double* __restrict__ wb, const double* __restrict__ sc,
    double* __restrict__ part, double* __restrict__ CA, int nagg, int ncp, const double* __restrict__ svec, const double* __restrict__ cen)
{
    __shared__ double red[42][6];
    // block rows are dealt to the XCDs in contiguous ranges (workgroups go round-robin over the 8 XCDs)
    const int rows_per_xcd = (nfree + PCG_XCDS - 1) / PCG_XCDS;
    const int row = ((int)blockIdx.x % PCG_XCDS) * rows_per_xcd + (int)blockIdx.x / PCG_XCDS;
    if (row >= nfree) return;
    const long long n = 6LL * nfree;
    const double* xin = wb + (long long)(MODE == 0 ? PV_U : PV_M) * n;
    const int slot = threadIdx.x / 6, r = threadIdx.x - 6 * slot;
    typedef double pr_d2 __attribute__((ext_vector_type(2)));
    const long long o = 6LL * row + threadIdx.x;
    double acc = 0;
    if (slot < 42) {
        const int k_end = row_ptr[2 * row + 1];
        for (int k0 = row_ptr[2 * row] + slot; k0 < k_end; k0 += EPS * 42) {
            pr_d2 bv[EPS][3]; int col[EPS];
#pragma unroll
            for (int q = 0; q < EPS; q++) {
                const int k = k0 + 42 * q;
                col[q] = -1;
#pragma unroll
                for (int c = 0; c < 3; c++) bv[q][c] = pr_d2{0.0, 0.0};
                if (k < k_end) {
                    col[q] = ecol[k];
                    const pr_d2* B = reinterpret_cast<const pr_d2*>(Hf + 36LL * k + 6 * r);
                    bv[q][0] = B[0]; bv[q][1] = B[1]; bv[q][2] = B[2];
                }
            }
            pr_d2 xv[EPS][3];
#pragma unroll
            for (int q = 0; q < EPS; q++) {
#pragma unroll
                for (int c = 0; c < 3; c++) xv[q][c] = pr_d2{0.0, 0.0};
                if (col[q] >= 0) {
                    const pr_d2* X = reinterpret_cast<const pr_d2*>(xin + 6LL * col[q]);
                    xv[q][0] = X[0]; xv[q][1] = X[1]; xv[q][2] = X[2];
                }
            }
#pragma unroll
            for (int q = 0; q < EPS; q++)
#pragma unroll
                for (int c = 0; c < 3; c++) { acc += bv[q][c].x * xv[q][c].x; acc += bv[q][c].y * xv[q][c].y; }
        }
    }
    // the epilogue's operands (lanes 0..5 of wave 0 own the row's six unknowns) are requested here, behind the mat-vec's loads -- held
    // across the loop they cost 20 registers and a wave per SIMD -- and arrive while the workgroup meets at the barrier
    double alpha = 0, beta = 0, vz = 0, vq = 0, vs = 0, vp = 0, vx = 0, vr = 0, vu = 0, vw = 0, vm = 0;
    if (threadIdx.x < 6) {
        vr = wb[PV_R * n + o]; vu = wb[PV_U * n + o];
        if (MODE == 1) {
            alpha = sc[5]; beta = sc[6];
            vz = wb[PV_Z * n + o]; vq = wb[PV_Q * n + o]; vs = wb[PV_S * n + o]; vp = wb[PV_P * n + o];
            vx = wb[PV_X * n + o]; vw = wb[PV_W * n + o]; vm = wb[PV_M * n + o];
        }
    }
    if (slot < 42) red[slot][r] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        double g = 0, dl = 0, rr = 0, wn = 0;
        if (threadIdx.x < 6) {
            double tot = 0;
            for (int s2 = 0; s2 < 42; s2++) tot += red[s2][threadIdx.x];
            if (MODE == 0) {
                wn = tot;
                wb[PV_W * n + o] = wn;
            } else {
                vz = tot + beta * vz; vq = vm + beta * vq; vs = vw + beta * vs; vp = vu + beta * vp;
                vx += alpha * vp; vr -= alpha * vs; vu -= alpha * vq; wn = vw - alpha * vz;
                wb[PV_Z * n + o] = vz; wb[PV_Q * n + o] = vq; wb[PV_S * n + o] = vs; wb[PV_P * n + o] = vp;
                wb[PV_X * n + o] = vx; wb[PV_R * n + o] = vr; wb[PV_U * n + o] = vu; wb[PV_W * n + o] = wn;
            }
            g = vr * vu; dl = wn * vu; rr = vr * vr;
        }
        for (int st = 4; st >= 1; st >>= 1) { g += __shfl_xor(g, st, 64); dl += __shfl_xor(dl, st, 64); rr += __shfl_xor(rr, st, 64); }   // lanes 0..7 (6, 7 hold 0)
        if (threadIdx.x == 0) { part[3 * row] = g; part[3 * row + 1] = dl; part[3 * row + 2] = rr; }
        if (CA) {
            const double w3 = __shfl(wn, 3, 64), w4 = __shfl(wn, 4, 64), w5 = __shfl(wn, 5, 64);
            if (threadIdx.x < PCG_CDOF) ppcg_contribute(row, threadIdx.x, wn, w3, w4, w5, nfree, nagg, ncp, svec, cen, CA);
        }
    }
}

// |r|^2 of the current iterate for the host (the row kernel leaves partials; the next k_ppcg_prec would reduce them one iteration late)
__global__ __launch_bounds__(256) void k_ppcg_publish(const double* __restrict__ part, int nfree, double* __restrict__ sc)
{
    __shared__ double red[4];
    double rr = 0;
    for (int i = threadIdx.x; i < nfree; i += 256) rr += part[3 * i + 2];
    for (int st = 32; st >= 1; st >>= 1) rr += __shfl_xor(rr, st, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = rr;
    __syncthreads();
    if (threadIdx.x == 0) sc[2] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---------------------------------------------------------------------------------------- host wrappers
static inline int nblk(long long n, int b) { return (int)((n + b - 1) / b); }

size_t sp_scan_temp_bytes(size_t n)
{
    size_t a = 0, b = 0;
    (void)rocprim::exclusive_scan(nullptr, a, (const int*)nullptr, (int*)nullptr, 0, n, rocprim::plus<int>());
    auto it = rocprim::make_transform_iterator((const uint8_t*)nullptr, U8ToInt());
    (void)rocprim::exclusive_scan(nullptr, b, it, (int*)nullptr, 0, n, rocprim::plus<int>());
    return a > b ? a : b;
}
hipError_t sp_scan_int(hipStream_t s, void* tmp, size_t tmp_bytes, const int* in, int* out, size_t n)
{
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0, n, rocprim::plus<int>(), s);
}
hipError_t sp_scan_flags(hipStream_t s, void* tmp, size_t tmp_bytes, const uint8_t* in, int* out, size_t n)
{
    auto it = rocprim::make_transform_iterator(in, U8ToInt());
    return rocprim::exclusive_scan(tmp, tmp_bytes, it, out, 0, n, rocprim::plus<int>(), s);
}
size_t sp_sort_temp_bytes(size_t n)
{
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned long long*)nullptr,
                                    (unsigned long long*)nullptr, n, 0, 32);
    (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, n, 0, 32);
    return a > b ? a : b;
}
hipError_t sp_sort_u64(hipStream_t s, void* tmp, size_t tmp_bytes, const unsigned* kin, unsigned* kout, const unsigned long long* vin,
                       unsigned long long* vout, size_t n, int bits)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, bits, s);
}
hipError_t sp_sort_u32(hipStream_t s, void* tmp, size_t tmp_bytes, const unsigned* kin, unsigned* kout, const unsigned* vin, unsigned* vout, size_t n)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, 32, s);
}

void sp_launch_pair_count(hipStream_t s, const BaDev& D, int* cnt) { hipLaunchKernelGGL(k_sp_pair_count, dim3(nblk(D.L, 256)), dim3(256), 0, s, D, cnt); }
void sp_launch_pair_fill(hipStream_t s, const BaDev& D, const int* off, unsigned* key, unsigned long long* val)
{ hipLaunchKernelGGL(k_sp_pair_fill, dim3(nblk((long long)SP_PF_LANES * D.L, 256)), dim3(256), 0, s, D, off, key, val); }
void sp_launch_mark(hipStream_t s, const unsigned* key, long long np, int nfree, uint8_t* map)
{ const long long m = np > nfree ? np : nfree; hipLaunchKernelGGL(k_sp_mark, dim3(nblk(m, 256)), dim3(256), 0, s, key, np, nfree, map); }
void sp_launch_block_coords(hipStream_t s, const uint8_t* map, const int* id, long long n2, int nfree, int* br, int* bc, int* diag)
{ hipLaunchKernelGGL(k_sp_block_coords, dim3(nblk(n2, 256)), dim3(256), 0, s, map, id, n2, nfree, br, bc, diag); }
void sp_launch_pair_block(hipStream_t s, const unsigned* key, const int* id, long long np, unsigned* out)
{ if (np > 0) hipLaunchKernelGGL(k_sp_pair_block, dim3(nblk(np, 256)), dim3(256), 0, s, key, id, np, out); }
void sp_launch_seg_bounds(hipStream_t s, const unsigned* sk, long long np, int* st, int* en)
{ if (np > 0) hipLaunchKernelGGL(k_sp_seg_bounds, dim3(nblk(np, 256)), dim3(256), 0, s, sk, np, st, en); }
void sp_launch_row_entries(hipStream_t s, const int* br, const int* bc, int nb, int nfree, unsigned* key, unsigned* val)
{ hipLaunchKernelGGL(k_sp_row_entries, dim3(nblk(nb, 256)), dim3(256), 0, s, br, bc, nb, nfree, key, val); }
void sp_launch_row_ptr(hipStream_t s, const unsigned* skey, int n_ent, int nfree, int* row_ptr)
{ hipLaunchKernelGGL(k_sp_row_ptr, dim3(nblk(n_ent, 256)), dim3(256), 0, s, skey, n_ent, nfree, row_ptr); }
void sp_launch_dinv(hipStream_t s, const BaDev& D, double lambda)
{
    if (D.L > 0) hipLaunchKernelGGL(k_sp_dinv, dim3(nblk(D.L, 256)), dim3(256), 0, s, D, lambda);
    if (D.E > 0) hipLaunchKernelGGL(k_sp_edge_y, dim3(nblk(D.E, 256)), dim3(256), 0, s, D, lambda);
}
void sp_launch_schur_blocks(hipStream_t s, const BaDev& D, const double* Y, const unsigned long long* pairs, const int* st, const int* en,
                            const int* br, const int* bc, int nb, double* Hb)
{
    // few blocks with long pair lists (a local BA: 210 blocks of ~1000 pairs) get 16 waves per block, maps with many blocks 4
    if (nb <= 0) return;
    // measured (config 5, Schur phase of optimize(20), tools/bench_gba.py): dispatch order 4.67 ms, contiguous eighths 8.73 (each XCD then
    // works on a handful of rows at a time and their operands sit in a few L2 channels), chunks of 4 / 16 / 64 / 512 block pairs
    // 4.53 / 4.14 / 4.00 / 4.36
    static const int xcd = getenv("CCM_SP_XCD") ? atoi(getenv("CCM_SP_XCD")) : 64;     // 0 dispatch order, 1 contiguous eighths, C > 1 chunks of C
    const int nwg = (nb + 1) / 2, per = (nwg + 7) / 8;
    if (nb < 2048) hipLaunchKernelGGL(k_sp_schur_blocks<16>, dim3(nwg), dim3(1024), 0, s, D, Y, pairs, st, en, br, bc, nb, Hb, 0);
    else if (xcd == 1) hipLaunchKernelGGL(k_sp_schur_blocks<4>, dim3(8 * per), dim3(256), 0, s, D, Y, pairs, st, en, br, bc, nb, Hb, per);
    else if (xcd > 1) hipLaunchKernelGGL(k_sp_schur_blocks<4>, dim3((nwg + 8 * xcd - 1) / (8 * xcd) * (8 * xcd)), dim3(256), 0, s, D, Y, pairs, st, en, br, bc, nb, Hb, -xcd);
    else hipLaunchKernelGGL(k_sp_schur_blocks<4>, dim3(nwg), dim3(256), 0, s, D, Y, pairs, st, en, br, bc, nb, Hb, 0);
}
void sp_launch_bschur(hipStream_t s, const BaDev& D, double* bs)
{
    if (D.nfree > 0 && D.nfree < 64) hipLaunchKernelGGL(k_sp_bschur<16>, dim3(D.nfree), dim3(1024), 0, s, D, bs);
    else if (D.nfree > 0 && D.nfree < 512) hipLaunchKernelGGL(k_sp_bschur<4>, dim3(D.nfree), dim3(256), 0, s, D, bs);
    else if (D.nfree > 0) hipLaunchKernelGGL(k_sp_bschur_wave, dim3(nblk(D.nfree, 4)), dim3(256), 0, s, D, bs);
}
void sp_launch_add_lambda(hipStream_t s, const int* diag, int nfree, double lambda, double* Hb)
{ hipLaunchKernelGGL(k_sp_add_lambda, dim3(nblk(6LL * nfree, 256)), dim3(256), 0, s, diag, nfree, lambda, Hb); }
void sp_launch_to_dense(hipStream_t s, const double* Hb, const int* br, const int* bc, int nb, long long n, double* Hs)
{ hipLaunchKernelGGL(k_sp_to_dense, dim3(nblk(36LL * nb, 256)), dim3(256), 0, s, Hb, br, bc, nb, n, Hs); }
size_t pcg_minv_bytes(int nfree) { return (size_t)nblk(nfree, PCG_CL) * PCG_CN * PCG_CN * 8; }
hipError_t pcg_launch_minv(hipStream_t s, const double* Hb, const int* blk_row, const int* blk_col, int nb, int nfree, double* Minv, int* bad)
{
    const int ncl = nblk(nfree, PCG_CL);
    hipError_t e = hipMemsetAsync(Minv, 0, pcg_minv_bytes(nfree), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pcg_cl_gather, dim3(nblk(36LL * nb, 256)), dim3(256), 0, s, Hb, blk_row, blk_col, nb, Minv);
    hipLaunchKernelGGL(k_pcg_cl_invert, dim3(ncl), dim3(256), 0, s, Minv, nfree, bad);
    return hipSuccess;
}
// coarse level: sizes, set-up (Ac into `Ac`, upper triangle row-major; the caller factors and inverts it, then mirrors)
int pcg_coarse_aggregates(int nfree) { return nblk(nfree, PCG_CL * pcg_agg_clusters(nfree)); }
int pcg_coarse_agg_keyframes(int nfree) { return PCG_CL * pcg_agg_clusters(nfree); }
int pcg_coarse_dim(int nfree) { return PCG_CDOF * pcg_coarse_aggregates(nfree); }
int pcg_coarse_pitch(int nfree) { return nblk(pcg_coarse_dim(nfree), INV_B) * INV_B; }
int pcg_coarse_parts(int nfree) { return nblk(pcg_coarse_dim(nfree), 4); }
// Ac = P^T H P as a full (padded) matrix in `Ac`
void pcg_launch_coarse_mark(hipStream_t s, const int* blk_row, const int* blk_col, int nb, int nfree, uint8_t* aggmap)
{
    if (nb > 0) hipLaunchKernelGGL(k_pcg_coarse_mark, dim3(nblk(nb, 256)), dim3(256), 0, s, blk_row, blk_col, nb, nfree, pcg_coarse_aggregates(nfree), aggmap);
}
hipError_t pcg_launch_coarse_build(hipStream_t s, const double* Hb, const uint8_t* map, const int* id, int nfree, const double* svec, const double* cen,
                                   const int* pairs, int npairs, double* Ac)
{
    const int nagg = pcg_coarse_aggregates(nfree), nc = PCG_CDOF * nagg, ncp = pcg_coarse_pitch(nfree);
    hipError_t e = hipMemsetAsync(Ac, 0, (size_t)ncp * ncp * sizeof(double), s);          // aggregate pairs without a block stay zero
    if (e != hipSuccess) return e;
    if (npairs > 0) hipLaunchKernelGGL(k_pcg_coarse_build, dim3(npairs), dim3(256), 0, s, Hb, map, id, nfree, nagg, ncp, svec, cen, pairs, Ac);
    hipLaunchKernelGGL(k_pcg_coarse_complete, dim3(nblk((long long)ncp * ncp, 256)), dim3(256), 0, s, Ac, nc, ncp);
    return hipSuccess;
}
void pcg_launch_coarse_mirror(hipStream_t s, double* A, int ncp)
{
    hipLaunchKernelGGL(k_pcg_coarse_mirror, dim3(nblk((long long)ncp * ncp, 256)), dim3(256), 0, s, A, ncp);
}
// yc = Ac^-1 P^T r  (P^T r as block partials in C.rc, left there by k_pcg_init / k_pcg_update), and the coarse share of r.z into C.cpart
static void pcg_launch_coarse(hipStream_t s, const PcgCoarse& C, int nfree, double* w)
{
    const int nagg = pcg_coarse_aggregates(nfree), nc = PCG_CDOF * nagg;
    (void)w;
    hipLaunchKernelGGL(k_pcg_coarse, dim3(nblk(nc, 4)), dim3(256), (size_t)nc * 8, s, C.Aci, nc, pcg_coarse_pitch(nfree), C.rc, nfree, C.yc, C.cpart);
}
// p = (z + P yc) + beta p and the scalars of the iteration; parity = the r.z slot this call writes
static void pcg_launch_direction(hipStream_t s, const PcgCoarse& C, int nfree, double* w, double* part, double* sc, int parity)
{
    const long long n = 6LL * nfree;
    hipLaunchKernelGGL(k_pcg_direction, dim3(nblk(n, 256)), dim3(256), 0, s, C.Aci ? C.yc : nullptr, C.svec, C.cen, nfree, C.Aci ? pcg_coarse_aggregates(nfree) : 0, w,
                       nblk(n, PCG_UPD_TPB), part, C.cpart, C.Aci ? pcg_coarse_parts(nfree) : 0, sc, parity);
}
void pcg_launch_init(hipStream_t s, const double* b, const double* Minv, int nfree, double* w, double* part, double* sc, const PcgCoarse& C)
{
    const int nb = nblk(6LL * nfree, PCG_UPD_TPB);
    hipLaunchKernelGGL(k_pcg_init, dim3(nb), dim3(PCG_UPD_TPB), 0, s, b, Minv, nfree, w, part, C.Aci ? pcg_coarse_aggregates(nfree) : 0, C.svec, C.cen,
                       C.Aci ? C.rc : nullptr);
    if (C.Aci) pcg_launch_coarse(s, C, nfree, w);
    hipLaunchKernelGGL(k_pcg_init_fin, dim3(1), dim3(64), 0, s, part, nb, C.cpart, C.Aci ? pcg_coarse_parts(nfree) : 0, sc);
    pcg_launch_direction(s, C, nfree, w, part, sc, 1);          // "iteration -1": beta = r.z / r.z with p = 0, i.e. p = z + P yc
}
// one PCG iteration = mat-vec, the vector updates with the cluster level, the coarse level, the next direction
void pcg_launch_iter(hipStream_t s, const double* Hb, const int* row_ptr, const unsigned* ekey, const unsigned* eval, const double* Minv,
                     int nfree, double* w, double* pap_part, double* part, double* sc, int parity, const PcgCoarse& C)
{
    const int nb = nblk(6LL * nfree, PCG_UPD_TPB);
    hipLaunchKernelGGL(k_pcg_spmv, dim3(PCG_XCDS * nblk(nfree, PCG_XCDS)), dim3(256), 0, s, Hb, row_ptr, ekey, eval, nfree, w, pap_part);
    hipLaunchKernelGGL(k_pcg_update, dim3(nb), dim3(PCG_UPD_TPB), 0, s, Minv, nfree, w, pap_part, sc, part, parity, C.Aci ? pcg_coarse_aggregates(nfree) : 0,
                       C.svec, C.cen, C.Aci ? C.rc : nullptr);
    if (C.Aci) pcg_launch_coarse(s, C, nfree, w);
    pcg_launch_direction(s, C, nfree, w, part, sc, parity);
}
// publish the scalars of the last iteration (before the host reads them)
void pcg_launch_publish(hipStream_t s, int nfree, double* part, double* sc, const PcgCoarse& C)
{
    hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(64), 0, s, nblk(6LL * nfree, PCG_UPD_TPB), part, C.cpart, C.Aci ? pcg_coarse_parts(nfree) : 0, sc);
}

// ---- pipelined PCG: host wrappers
bool ppcg_supported(int nfree) { return ppcg_prec_fits(pcg_coarse_pitch(nfree), pcg_coarse_agg_keyframes(nfree)); }
size_t ppcg_state_doubles(int nfree) { return (size_t)PV_COUNT * 6 * (size_t)nfree; }
size_t ppcg_ca_doubles(int nfree) { return (size_t)2 * pcg_coarse_agg_keyframes(nfree) * (size_t)pcg_coarse_pitch(nfree); }
void ppcg_launch_expand(hipStream_t s, const double* Hb, const unsigned* ekey, const unsigned* eval, int n_ent, int nfree, double* Hf, int* ecol)
{
    if (n_ent > 0) hipLaunchKernelGGL(k_ppcg_expand, dim3(nblk(36LL * n_ent, 256)), dim3(256), 0, s, Hb, ekey, eval, n_ent, nfree, Hf, ecol);
}
static void ppcg_launch_prec(hipStream_t s, const double* Minv, int nfree, double* wb, int in_slot, int out_slot, const PcgCoarse& C, const PpcgBufs& B,
                             double* part, double* sc, int do_scalars)
{
    const int nb32 = nblk(nfree, PP_KF);
    const int nagg = C.Aci ? pcg_coarse_aggregates(nfree) : 0, nc = PCG_CDOF * nagg, ncp = C.Aci ? pcg_coarse_pitch(nfree) : 0;
    const size_t lds = (size_t)((1 + PP_CQ) * ncp + 2) * sizeof(double);
    hipLaunchKernelGGL(k_ppcg_prec, dim3(nb32 + 1), dim3(PP_TPB), lds, s, Minv, nfree, wb, in_slot, out_slot, (const double*)C.Aci, nc, ncp, (const double*)(C.Aci ? B.CA : nullptr), nagg,
                       C.svec, C.cen, (const double*)part, sc, do_scalars);
}
template <int MODE>
static void ppcg_launch_row(hipStream_t s, const int* row_ptr, int nfree, double* wb, double* part, double* sc, const PcgCoarse& C, const PpcgBufs& B)
{
    const int nagg = C.Aci ? pcg_coarse_aggregates(nfree) : 0, ncp = C.Aci ? pcg_coarse_pitch(nfree) : 0;
    static const int eps = getenv("CCM_PPCG_EPS") ? atoi(getenv("CCM_PPCG_EPS")) : 4;        // measured: see DESIGN.md
    if (eps == 2) hipLaunchKernelGGL((k_ppcg_row<MODE, 2>), dim3(PCG_XCDS * nblk(nfree, PCG_XCDS)), dim3(256), 0, s, (const double*)B.Hf, (const int*)B.ecol, row_ptr, nfree, wb, (const double*)sc, part,
                                     C.Aci ? B.CA : nullptr, nagg, ncp, C.svec, C.cen);
    else hipLaunchKernelGGL((k_ppcg_row<MODE, 4>), dim3(PCG_XCDS * nblk(nfree, PCG_XCDS)), dim3(256), 0, s, (const double*)B.Hf, (const int*)B.ecol, row_ptr, nfree, wb, (const double*)sc, part,
                            C.Aci ? B.CA : nullptr, nagg, ncp, C.svec, C.cen);
}
// x = 0, r = b, u = M^-1 r, w = A u and the first partials
hipError_t ppcg_launch_init(hipStream_t s, const double* b, const double* Minv, const int* row_ptr, int nfree, double* wb, double* part, double* sc,
                            const PcgCoarse& C, const PpcgBufs& B)
{
    const int nagg = C.Aci ? pcg_coarse_aggregates(nfree) : 0, ncp = C.Aci ? pcg_coarse_pitch(nfree) : 0;
    double* CA = C.Aci ? B.CA : nullptr;
    if (CA) { hipError_t e = hipMemsetAsync(CA, 0, ppcg_ca_doubles(nfree) * sizeof(double), s); if (e != hipSuccess) return e; }   // support rows past a map's ends stay zero
    hipLaunchKernelGGL(k_ppcg_init, dim3(nblk(nfree, PCG_UPD_KF) + 1), dim3(256), 0, s, b, nfree, wb, nagg, ncp, C.svec, C.cen, CA, sc);
    ppcg_launch_prec(s, Minv, nfree, wb, PV_R, PV_U, C, B, part, sc, 0);
    ppcg_launch_row<0>(s, row_ptr, nfree, wb, part, sc, C, B);
    return hipSuccess;
}
void ppcg_launch_iter(hipStream_t s, const double* Minv, const int* row_ptr, int nfree, double* wb, double* part, double* sc, const PcgCoarse& C, const PpcgBufs& B)
{
    ppcg_launch_prec(s, Minv, nfree, wb, PV_W, PV_M, C, B, part, sc, 1);
    ppcg_launch_row<1>(s, row_ptr, nfree, wb, part, sc, C, B);
}
void ppcg_launch_publish(hipStream_t s, const double* part, int nfree, double* sc)
{
    hipLaunchKernelGGL(k_ppcg_publish, dim3(1), dim3(256), 0, s, part, nfree, sc);
}

// Reduced systems of a local BA (config 4: 20 free keyframes, n = 120 unknowns) are solved by ONE workgroup in one launch: L L^T
// factorisation and both triangular solves.  The block Gauss-Jordan path below needs ~14 launches for such a system and took
// 0.5 ms per LM trial -- 70 % of the whole local BA.  n <= DENSE_SMALL_MAX so that every tile of the lower triangle has its thread.
#define DENSE_SMALL_MAX 138
#define DS_TPB 576
// Every 6x6 tile of the lower triangle lives in the REGISTERS of TWO neighbouring threads for the whole solve, three rows each
// (n <= 138: at most 276 tiles, 552 threads): per block column the diagonal tile's owners factor it and publish it in LDS, the owners
// of the tiles below solve against it (multiplying by the reciprocals of its diagonal: a double-precision division is ~12
// instructions, 36 of them per tile and column were a fifth of the kernel) and publish the panel, every remaining tile subtracts
// panel_I panel_K^T from its registers -- two barriers per block column, and the only LDS traffic is the panel and the right-hand
// side.  The forward substitution rides along with the factorisation and the backward one reads the factor from the registers it
// already is in, so the factor is never written anywhere.  (History at n = 120: matrix in LDS column by column 256 us, blocked in LDS
// 133 us, register tiles + the factor copied to LDS for two separate substitution loops 116 us, one thread per tile 88 us: the
// kernel is bound by the double-precision instructions its busiest thread issues per block column -- 216 multiply-adds of a trailing
// tile, now 108 -- not by anything a second workgroup could share.)  The two owners of a diagonal tile exchange their rows by
// shuffles and BOTH run the serial 6 x 6 factorisation, so neither waits for the other's result.
// The damping is added to the diagonal while the tiles are loaded and the verdict is WRITTEN (0 / 1) rather than or-ed in: the
// launches of k_sp_add_lambda and of the memset of `bad` in front of this kernel were two of a local BA trial's thirteen.
__global__ __launch_bounds__(DS_TPB) void k_dense_small_solve(const double* __restrict__ Hb, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                                              int nb, int n, const double* __restrict__ b, double* __restrict__ x, int* __restrict__ bad, double lambda)
{
    extern __shared__ double ds_lds[];
    double* v = ds_lds;                      // [n] right-hand side -> y -> solution
    __shared__ double s_ljj[36], s_linv[6], s_y[6];
    __shared__ double s_panel[DENSE_SMALL_MAX / 6][36];
    __shared__ int s_tile_blk[DENSE_SMALL_MAX / 6 * (DENSE_SMALL_MAX / 6 + 1) / 2];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int nbk = n / 6, ntiles = nbk * (nbk + 1) / 2;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < ntiles; i += DS_TPB) s_tile_blk[i] = -1;
    for (int i = tid; i < n; i += DS_TPB) v[i] = b[i];
    __syncthreads();
    for (int k = tid; k < nb; k += DS_TPB) { const int r = blk_row[k], c = blk_col[k]; if (r <= c && c < nbk) s_tile_blk[c * (c + 1) / 2 + r] = k; }
    __syncthreads();
    // rows 3 p .. 3 p + 2 of tile (I, K), K <= I
    const int tile = tid >> 1, p = tid & 1;
    int I = 0, K = 0;
    const bool have = tile < ntiles;
    // tiles in COLUMN-major order (block column K, then block row I): the tiles still at work in step J -- K >= J -- are the tail of
    // the thread range, so the waves in front of it skip a step's phases altogether instead of issuing them for one or two live lanes
    // (row-major order kept every wave busy until the last columns: 79.6 against 68.2 us at n = 120)
    if (have) { int t = tile; while (t >= nbk - K) { t -= nbk - K; K++; } I = K + t; }
    const int tile_rm = I * (I + 1) / 2 + K;                                 // its index in s_tile_blk (row-major)
    double T[3][6];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 6; k++) T[i][k] = 0.0;
    if (have && s_tile_blk[tile_rm] >= 0) {
        const double* B = Hb + 36LL * s_tile_blk[tile_rm];                   // stored block (row K, col I): tile(i, k) = B[k][i]
#pragma unroll
        for (int ii = 0; ii < 3; ii++) {
            const int i = 3 * p + ii;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                // a diagonal block is stored whole; keep it exactly symmetric (upper half wins)
                T[ii][k] = (I == K) ? B[max(i, k) * 6 + min(i, k)] : B[k * 6 + i];
                if (I == K && k == i) T[ii][k] += lambda;
            }
        }
    }
    // the full diagonal tile out of its two owners' rows (both owners call this together: they are neighbouring lanes of one wave)
    auto gather_diag = [&](double (&F)[6][6]) {
#pragma unroll
        for (int ii = 0; ii < 3; ii++)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const double mine = T[ii][k], other = __shfl_xor(mine, 1, 64);
                F[ii][k] = p ? other : mine;
                F[3 + ii][k] = p ? mine : other;
            }
    };
    double dinv[6] = { 0, 0, 0, 0, 0, 0 };                                   // a diagonal tile's owners: 1 / L_cc
    // factorisation, with the forward substitution L y = b riding along: the diagonal tile's owners solve its six unknowns as soon
    // as the tile is factored, and the owner of a panel tile's row takes its product with y_J off b_I when the row is final
    // (one writer per unknown and step).  Two barriers per block column.
    for (int J = 0; J < nbk; J++) {
        const int j0 = 6 * J;
        if (have && I == J && K == J) {                                      // 6x6 Cholesky of the diagonal tile, by both owners alike
            double F[6][6];
            gather_diag(F);
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double d = F[c][c];
#pragma unroll
                for (int k = 0; k < c; k++) d -= F[c][k] * F[c][k];
                if (!(d > 0.0)) { s_bad = 1; d = 1.0; }
                // 1 / sqrt(d) once, sqrt(d) = d / sqrt(d): a square root AND a division per pivot were two ~25-instruction sequences on the
                // threads every other thread waits for (a third of the kernel)
                const double id = rsqrt(d);
                F[c][c] = d * id;
                dinv[c] = id;
#pragma unroll
                for (int r = c + 1; r < 6; r++) {
                    double w = F[r][c];
#pragma unroll
                    for (int k = 0; k < c; k++) w -= F[r][k] * F[c][k];
                    F[r][c] = w * id;
                }
            }
            double y[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double w = v[j0 + c];
#pragma unroll
                for (int k = 0; k < c; k++) w -= F[c][k] * y[k];
                y[c] = w * dinv[c];
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int k = 0; k < 6; k++) if (k > i) F[i][k] = 0.0;
#pragma unroll
            for (int ii = 0; ii < 3; ii++)
#pragma unroll
                for (int k = 0; k < 6; k++) T[ii][k] = p ? F[3 + ii][k] : F[ii][k];
            if (p == 0) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
#pragma unroll
                    for (int k = 0; k < 6; k++) s_ljj[i * 6 + k] = F[i][k];
                    s_linv[i] = dinv[i]; s_y[i] = y[i]; v[j0 + i] = y[i];
                }
            }
        }
        __syncthreads();
        if (have && K == J && I > J) {                                       // panel tile: X L_JJ^T = T, row by row
            double lj[6][6], li[6], yj[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                li[i] = s_linv[i]; yj[i] = s_y[i];
#pragma unroll
                for (int k = 0; k < i; k++) lj[i][k] = s_ljj[i * 6 + k];
            }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double w = T[r][c];
#pragma unroll
                    for (int k = 0; k < c; k++) w -= T[r][k] * lj[c][k];
                    T[r][c] = w * li[c];
                }
#pragma unroll
            for (int ii = 0; ii < 3; ii++) {
                const int i = 3 * p + ii;
#pragma unroll
                for (int k = 0; k < 6; k++) s_panel[I][i * 6 + k] = T[ii][k];
                v[6 * I + i] -= ((T[ii][0] * yj[0] + T[ii][1] * yj[1]) + (T[ii][2] * yj[2] + T[ii][3] * yj[3])) + (T[ii][4] * yj[4] + T[ii][5] * yj[5]);
            }
        }
        __syncthreads();
        if (have && K > J) {                                                 // trailing tile: T -= panel_I panel_K^T
            double pi[3][6], pk[6][6];
#pragma unroll
            for (int ii = 0; ii < 3; ii++)
#pragma unroll
                for (int k = 0; k < 6; k++) pi[ii][k] = s_panel[I][(3 * p + ii) * 6 + k];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int k = 0; k < 6; k++) pk[i][k] = s_panel[K][i * 6 + k];
#pragma unroll
            for (int ii = 0; ii < 3; ii++)
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    double w = 0;
#pragma unroll
                    for (int q = 0; q < 6; q++) w += pi[ii][q] * pk[k][q];
                    T[ii][k] -= w;
                }
        }
    }
    __syncthreads();
    // L^T x = y from the tiles where they are (registers): block row J of L is what column J of L^T needs
    for (int J = nbk - 1; J >= 0; J--) {
        const int j0 = 6 * J;
        if (have && I == J && K == J) {
            double F[6][6], xx[6];
            gather_diag(F);
#pragma unroll
            for (int c = 5; c >= 0; c--) {
                double w = v[j0 + c];
#pragma unroll
                for (int k = c + 1; k < 6; k++) w -= F[k][c] * xx[k];
                xx[c] = w * dinv[c];
            }
            if (p == 0) {
#pragma unroll
                for (int c = 0; c < 6; c++) { v[j0 + c] = xx[c]; s_y[c] = xx[c]; }
            }
        }
        __syncthreads();
        if (have && I == J && K < J) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double w = 0;
#pragma unroll
                for (int rr = 0; rr < 3; rr++) w += T[rr][c] * s_y[3 * p + rr];
                w += __shfl_xor(w, 1, 64);                                   // the tile's other three rows
                if (p == 0) v[6 * K + c] -= w;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += DS_TPB) x[i] = v[i];
    if (tid == 0) *bad = s_bad ? 1 : 0;
}
int dense_small_max() { return DENSE_SMALL_MAX; }
int dense_launch_small_solve(hipStream_t s, const double* Hb, const int* blk_row, const int* blk_col, int nb, int n, const double* b, double* x, int* bad, double lambda)
{
    const size_t lds = (size_t)(n + 8) * sizeof(double);
    if (hipFuncSetAttribute((const void*)k_dense_small_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_dense_small_solve, dim3(1), dim3(DS_TPB), lds, s, Hb, blk_row, blk_col, nb, n, b, x, bad, lambda);
    return 0;
}

int dense_pitch(long long n) { return (int)((n + INV_B - 1) / INV_B) * INV_B; }
// Dense solve of the reduced system (small maps; fallback of the PCG): A (upper block triangle, row-major, pitch lda = dense_pitch(n),
// followed by one INV_B x INV_B block of scratch) is completed, inverted in place by the block Gauss-Jordan above and applied
// to b.  No library call: rocSOLVER's potrf / potrs returned wrong solutions (relative errors up to 1e-2, tools/dbg_potrf.py)
// whenever a second process factored on the same GPU at the same time, and differed in the last bits from run to run next to
// this library's own side stream; they were exact and repeatable only when they ran alone.
void dense_launch_solve(hipStream_t s, double* A, int n, int lda, const double* b, double* x, int* bad)
{
    hipLaunchKernelGGL(k_pcg_coarse_complete, dim3(nblk((long long)lda * lda, 256)), dim3(256), 0, s, A, n, lda);
    pcg_launch_coarse_invert(s, A, lda, A + (size_t)lda * lda, bad);
    hipLaunchKernelGGL(k_pcg_coarse_mirror, dim3(nblk((long long)lda * lda, 256)), dim3(256), 0, s, A, lda);
    hipLaunchKernelGGL(k_dense_matvec, dim3(nblk(n, 4)), dim3(256), 0, s, A, n, lda, b, x);
}
