// ba_kernels.hip -- CDNA4 kernels of the 6-DoF pose / 3-DoF point reprojection bundle adjustment
// (what src/Optimizer.cpp hands to g2o's BlockSolver_6_3 + Levenberg).  All arithmetic is float64.
//
// Edge storage is sorted by (landmark, pose): a landmark's observations are contiguous, which is the
// order BlockSolver::solve walks them in (block_solver.hpp:381-432).
//   k_ba_pose_rt      quaternion+t -> R|t per pose
//   k_ba_errors       computeActiveErrors + activeRobustChi2            (sparse_optimizer.cpp:61-114)
//   k_ba_lin_landmark linearizeOplus + constructQuadraticForm, landmark side: Hll, b_l, Hpl per edge
//   k_ba_lin_pose     the same, pose side: Hpp, b_p (one workgroup per free pose, fixed summation order)
//   (Schur complement and reduced solve: ba_sparse.hip)
//   k_ba_backsub      x_l = Dinv (b_l - Hpl^T x_p)                        (block_solver.hpp:461-481)
//   k_ba_update       oplus on poses (exp map) and points                 (sparse_optimizer.cpp:422-435)
#include <hip/hip_runtime.h>
#include <cstdint>
#include "ba_math.h"
#include "ba_types.h"

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_ba_pose_rt(BaDev D)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= D.P) return;
    double R[9];
    ba_quat_to_R(D.poses + 7 * (long long)p, R);
    double* o = D.Rt + 12 * (long long)p;
    for (int i = 0; i < 9; i++) o[i] = R[i];
    o[9] = D.poses[7 * (long long)p + 4]; o[10] = D.poses[7 * (long long)p + 5]; o[11] = D.poses[7 * (long long)p + 6];
}

// One thread per edge; block partial sums of rho(chi2) in a fixed tree -> partial[blockIdx.x]
__global__ __launch_bounds__(256) void k_ba_errors(BaDev D, double huber_delta, double* __restrict__ partial)
{
    __shared__ double red[4];
    const int e = blockIdx.x * 256 + threadIdx.x;
    double r = 0;
    if (e < D.E && D.active[e]) {
        const int pi = D.edge_pose[e], li = D.edge_point[e];
        double er[2];
        ba_edge_eval(D.Rt + 12 * (long long)pi, D.intr + 4 * (long long)pi, D.points + 3 * (long long)li,
                     D.obs + 2 * (long long)e, er, nullptr, nullptr, nullptr);
        D.err[2 * (long long)e] = er[0]; D.err[2 * (long long)e + 1] = er[1];
        const double c2 = D.info[e] * (er[0] * er[0] + er[1] * er[1]);
        if (huber_delta > 0) { double r1; ba_huber(c2, huber_delta, &r, &r1); }
        else r = c2;
    }
    r = wave_sum_d(r);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Sum n doubles with one block in a fixed order; optional max instead of sum.  out[0] = result.
__global__ __launch_bounds__(256) void k_ba_reduce(const double* __restrict__ in, int n, double* __restrict__ out, int take_max)
{
    __shared__ double red[256];
    double acc = 0.0;
    // eight loads in flight, added in the same order as one by one (the loop was 28 dependent memory round trips at config 5: 20 us)
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = in[i + 256 * q];
#pragma unroll
        for (int q = 0; q < 8; q++) acc = take_max ? fmax(acc, fabs(v[q])) : acc + v[q];
    }
    for (; i < n; i += 256) acc = take_max ? fmax(acc, fabs(in[i])) : acc + in[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = take_max ? fmax(red[threadIdx.x], red[threadIdx.x + s]) : red[threadIdx.x] + red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// the sums of TWO arrays in one launch (block 0: in0 -> out0, block 1: in1 -> out1), each in k_ba_reduce's order: a trial's chi2 and
// its gain denominator are needed together, and a launch of their own each was 5 us of a launch-bound local BA's trial
__global__ __launch_bounds__(256) void k_ba_reduce2(const double* __restrict__ in0, int n0, double* __restrict__ out0,
                                                    const double* __restrict__ in1, int n1, double* __restrict__ out1)
{
    __shared__ double red[256];
    const double* __restrict__ in = blockIdx.x ? in1 : in0;
    const int n = blockIdx.x ? n1 : n0;
    double acc = 0.0;
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = in[i + 256 * q];
#pragma unroll
        for (int q = 0; q < 8; q++) acc = acc + v[q];
    }
    for (; i < n; i += 256) acc = acc + in[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) (blockIdx.x ? out1 : out0)[0] = red[0];
}

// G lanes per landmark (G = BA_LM_LANES = 8 on every map size): accumulate Hll (full 3x3) and b_l over its edges, store Hpl = w B^T A per edge.
// Lane g walks every G-th edge of the landmark; the G partial sums are added pairwise by xor-shuffles (a fixed tree: reproducible).
// One thread per landmark was eight dependent rounds of f64 latency on a landmark of eight observations, and a local BA's 5000
// landmarks were 79 waves on a chip with room for 8192; large maps gain as well (config 5: 3.2 -> 2.6 ms per nine trials).
template <int G>
__device__ __forceinline__ double ba_group_sum(double v)
{
#pragma unroll
    for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// MODE 0: Hll, b_l and Hpl per edge (the first LM iteration, whose lambda comes out of this very linearisation).
// MODE 1 (lambda known before the linearisation: every later iteration): the landmark's share of the Schur step rides along --
// Dinv = (Hll + lambda I)^-1, db = Dinv b_l, and per edge Z = Hpl L^-T and ce = Hpl db, from the Hpl blocks the group has just
// written -- instead of k_sp_dinv + k_sp_edge_y re-reading Hll and the 260 MB of Hpl (config 5) in launches of their own.
// MODE 2: the same share WITHOUT Hpl.  After the first iteration nothing but this kernel's own tail read the array (the Schur product
// and the back-substitution work from Z, the reduced right-hand side from ce), and its 144 bytes per edge were 36 % of the bytes the
// kernel writes -- a kernel that runs at the rate its scattered 8-byte stores drain (2.2 TB/s, profiles/r03f_gba_counters.txt).  The
// second pass evaluates the edge again instead (its operands are the first pass's, still in the L2) and forms Hpl in registers.  A
// trial that is REJECTED needs Z for another lambda: the host then rebuilds Hpl with MODE 0 (rare: no iteration of config 5).
// The G lanes of a landmark hold one block of NV 16-byte values each (one edge's Z, ce or Hpl) for the edges eb .. eb + G - 1, which lie
// one after the other in the output array.  Stored lane by lane, one store instruction touches 64 different lines, 16 bytes each, and
// the kernel runs at the rate those partial lines drain; passed through LDS the group writes the same bytes as runs of G x 16 = 128.
// (Lanes of one wave: LDS operations of a wave complete in program order, no barrier.)
template <int G, int NV>
__device__ __forceinline__ void lm_store_group(double2* stage_group, int g, bool have, const double* v, double2* out_group, int n_edges)
{
    if (have) {
#pragma unroll
        for (int i = 0; i < NV; i++) stage_group[g * NV + i] = make_double2(v[2 * i], v[2 * i + 1]);
    }
    const int nq = n_edges * NV;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int q = j * G + g;
        if (q < nq) out_group[q] = stage_group[q];
    }
}
template <int G, int MODE>
__global__ __launch_bounds__(256) void k_ba_lin_landmark(BaDev D, double huber_delta, double lambda)
{
    __shared__ double2 stage[256 * 9];
    const int gt = blockIdx.x * 256 + threadIdx.x;
    const int l = gt / G, g = gt - l * G;
    const bool live = l < D.L;
    double2* stage_group = stage + 9 * (threadIdx.x - g);
    double H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, b[3] = { 0, 0, 0 };
    const double* pt = D.points + 3 * (long long)(live ? l : 0);
    const double p3[3] = { pt[0], pt[1], pt[2] };
    const int e0 = live ? D.pt_first[l] : 0, e1 = live ? D.pt_first[l + 1] : 0;
    for (int eb = e0; eb < e1; eb += G) {                            // (the same trip count for the G lanes of a landmark)
        const int e = eb + g;
        const bool have = e < e1;
        double Hx[18];
        for (int i = 0; i < 18; i++) Hx[i] = 0;
        if (have && D.active[e]) {
            const int pi = D.edge_pose[e];
            double er[2], A[6], B[12];
            ba_edge_eval(D.Rt + 12 * (long long)pi, D.intr + 4 * (long long)pi, p3, D.obs + 2 * (long long)e, er, A, B, nullptr);
            const double om = D.info[e];
            double r0, r1 = 1.;
            if (huber_delta > 0) ba_huber(om * (er[0] * er[0] + er[1] * er[1]), huber_delta, &r0, &r1);
            const double w = r1 * om;
            const double g0 = -om * er[0] * r1, g1 = -om * er[1] * r1;
            for (int i = 0; i < 3; i++) {
                b[i] += A[i] * g0 + A[3 + i] * g1;
                for (int j = 0; j < 3; j++) H[i * 3 + j] += w * (A[i] * A[j] + A[3 + i] * A[3 + j]);
            }
            if (MODE != 2 && D.free_of[pi] >= 0) {
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 3; j++) Hx[i * 3 + j] = w * (B[i] * A[j] + B[6 + i] * A[3 + j]);
            }
        }
        if (MODE != 2) lm_store_group<G, 9>(stage_group, g, have, Hx, reinterpret_cast<double2*>(D.Hpl + 18 * (long long)eb), min(G, e1 - eb));
    }
    if (G > 1) {
#pragma unroll
        for (int i = 0; i < 9; i++) H[i] = ba_group_sum<G>(H[i]);
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = ba_group_sum<G>(b[i]);
    }
    if (live && g == 0) {
        for (int i = 0; i < 9; i++) D.Hll[9 * (long long)l + i] = H[i];
        for (int i = 0; i < 3; i++) D.bl[3 * (long long)l + i] = b[i];
    }
    if (MODE != 0 && live) {
        double Dm[9], Di[9], f[6], d[3];
        for (int i = 0; i < 9; i++) Dm[i] = H[i];
        Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
        ba_inv3(Dm, Di);
        d[0] = Di[0] * b[0] + Di[1] * b[1] + Di[2] * b[2];
        d[1] = Di[3] * b[0] + Di[4] * b[1] + Di[5] * b[2];
        d[2] = Di[6] * b[0] + Di[7] * b[1] + Di[8] * b[2];
        ba_chol3(H, lambda, f);
        if (g == 0) {
            for (int i = 0; i < 9; i++) D.Dinv[9 * (long long)l + i] = Di[i];
            for (int i = 0; i < 3; i++) D.db[3 * (long long)l + i] = d[i];
        }
        for (int eb = e0; eb < e1; eb += G) {                        // this lane's edges again
            const int e = eb + g;
            const bool have = e < e1;
            double Bx[18];
            for (int i = 0; i < 18; i++) Bx[i] = 0;
            if (MODE == 1) {                                         // their Hpl blocks are the group's own stores
                if (have) {
                    const double2* Hx = reinterpret_cast<const double2*>(D.Hpl + 18 * (long long)e);
                    for (int i = 0; i < 9; i++) { const double2 v = Hx[i]; Bx[2 * i] = v.x; Bx[2 * i + 1] = v.y; }
                }
            } else if (have) {
                const int pi = D.edge_pose[e];
                if (D.active[e] && D.free_of[pi] >= 0) {
                    double er[2], A[6], B[12];
                    ba_edge_eval(D.Rt + 12 * (long long)pi, D.intr + 4 * (long long)pi, p3, D.obs + 2 * (long long)e, er, A, B, nullptr);
                    const double om = D.info[e];
                    double r0, r1 = 1.;
                    if (huber_delta > 0) ba_huber(om * (er[0] * er[0] + er[1] * er[1]), huber_delta, &r0, &r1);
                    const double w = r1 * om;
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 3; j++) Bx[i * 3 + j] = w * (B[i] * A[j] + B[6 + i] * A[3 + j]);
                }
            }
            double zz[18], cc[6];
            ba_edge_z_c(Bx, f, d, zz, cc);
            const int ne = min(G, e1 - eb);
            lm_store_group<G, 9>(stage_group, g, have, zz, reinterpret_cast<double2*>(D.Z + 18 * (long long)eb), ne);
            lm_store_group<G, 3>(stage_group, g, have, cc, reinterpret_cast<double2*>(D.ce + 6 * (long long)eb), ne);
        }
    }
}

// One workgroup (4 waves) per free pose: the waves take every fourth 64-edge run of the pose's list, reduce inside the wave by a fixed
// butterfly and are added wave 0..3 by wave 0.  (One WAVE per pose, as in round 1, left a 20-keyframe local BA with 20 waves on a chip
// with room for 8192, and config 5 with 2000: 56 us resp. 65 us per call.)
// NW waves per keyframe: 4, or 16 when a map has few keyframes (a local BA's 20 workgroups leave the chip empty either way, and
// with 1300 edges per keyframe a thread of 256 walked five of them one after the other)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_ba_lin_pose(BaDev D, double huber_delta)
{
    __shared__ double part[NW][28];
    const int f = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (f >= D.nfree) return;
    const int pi = D.pose_of_free[f];
    double Rt[12], K[4];
    for (int i = 0; i < 12; i++) Rt[i] = D.Rt[12 * (long long)pi + i];
    for (int i = 0; i < 4; i++) K[i] = D.intr[4 * (long long)pi + i];
    double H[21], b[6];
    for (int i = 0; i < 21; i++) H[i] = 0;
    for (int i = 0; i < 6; i++) b[i] = 0;
    for (int k = D.pose_first[f] + threadIdx.x; k < D.pose_first[f + 1]; k += 64 * NW) {
        const int e = D.pose_edges[k];
        if (!D.active[e]) continue;
        double er[2], A[6], B[12];
        ba_edge_eval(Rt, K, D.points + 3 * (long long)D.edge_point[e], D.obs + 2 * (long long)e, er, A, B, nullptr);
        const double om = D.info[e];
        double r0, r1 = 1.;
        if (huber_delta > 0) ba_huber(om * (er[0] * er[0] + er[1] * er[1]), huber_delta, &r0, &r1);
        const double w = r1 * om;
        const double g0 = -om * er[0] * r1, g1 = -om * er[1] * r1;
        int m = 0;
        for (int i = 0; i < 6; i++) {
            b[i] += B[i] * g0 + B[6 + i] * g1;
            for (int j = i; j < 6; j++) H[m++] += w * (B[i] * B[j] + B[6 + i] * B[6 + j]);
        }
    }
    for (int i = 0; i < 21; i++) H[i] = wave_sum_d(H[i]);
    for (int i = 0; i < 6; i++) b[i] = wave_sum_d(b[i]);
    if (lane == 0) {
        for (int i = 0; i < 21; i++) part[wv][i] = H[i];
        for (int i = 0; i < 6; i++) part[wv][21 + i] = b[i];
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        double v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
#pragma unroll
        for (int w = 4; w < NW; w++) v += part[w][threadIdx.x];                                          // wave 0 .. NW-1: fixed order
        if (threadIdx.x >= 21) D.bp[6 * (long long)f + (threadIdx.x - 21)] = v;
        else {
            int i = 0, m = threadIdx.x;                       // packed upper-triangle index -> (i, j)
            while (m >= 6 - i) { m -= 6 - i; i++; }
            const int j = i + m;
            double* o = D.Hpp + 36 * (long long)f;
            o[i * 6 + j] = v; o[j * 6 + i] = v;
        }
    }
}

// One wave per free pose: lanes stride the pose's edge list, then a fixed butterfly reduction.
__global__ __launch_bounds__(256) void k_ba_lin_pose_wave(BaDev D, double huber_delta)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= D.nfree) return;
    const int pi = D.pose_of_free[f];
    double Rt[12], K[4];
    for (int i = 0; i < 12; i++) Rt[i] = D.Rt[12 * (long long)pi + i];
    for (int i = 0; i < 4; i++) K[i] = D.intr[4 * (long long)pi + i];
    double H[21], b[6];
    for (int i = 0; i < 21; i++) H[i] = 0;
    for (int i = 0; i < 6; i++) b[i] = 0;
    for (int k = D.pose_first[f] + lane; k < D.pose_first[f + 1]; k += 64) {
        const int e = D.pose_edges[k];
        if (!D.active[e]) continue;
        double er[2], A[6], B[12];
        ba_edge_eval(Rt, K, D.points + 3 * (long long)D.edge_point[e], D.obs + 2 * (long long)e, er, A, B, nullptr);
        const double om = D.info[e];
        double r0, r1 = 1.;
        if (huber_delta > 0) ba_huber(om * (er[0] * er[0] + er[1] * er[1]), huber_delta, &r0, &r1);
        const double w = r1 * om;
        const double g0 = -om * er[0] * r1, g1 = -om * er[1] * r1;
        int m = 0;
        for (int i = 0; i < 6; i++) {
            b[i] += B[i] * g0 + B[6 + i] * g1;
            for (int j = i; j < 6; j++) H[m++] += w * (B[i] * B[j] + B[6 + i] * B[6 + j]);
        }
    }
    for (int i = 0; i < 21; i++) H[i] = wave_sum_d(H[i]);
    for (int i = 0; i < 6; i++) b[i] = wave_sum_d(b[i]);
    if (lane == 0) {
        double* o = D.Hpp + 36 * (long long)f;
        int m = 0;
        for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) { o[i * 6 + j] = H[m]; o[j * 6 + i] = H[m]; m++; }
        for (int i = 0; i < 6; i++) D.bp[6 * (long long)f + i] = b[i];
    }
}

// x_l = Dinv (b_l - sum_e Hpl_e^T x_p(e)) with Hpl_e = Z_e L^T (L L^T = Hll + lambda I, Z the trial's Schur operand): Dinv b_l = db and
// Dinv L = L^-T, so x_l = db - L^-T sum_e Z_e^T x_p(e) -- read from the array the Schur product has just used, which lets the
// linearisation drop Hpl (k_ba_lin_landmark MODE 2).
// (The loads of Z stay lane by lane, 16 bytes each at a stride of 144: fetched as runs of G x 16 bytes and handed out through LDS -- the
// mirror image of lm_store_group, which pays for the stores -- the kernel measured 99 instead of 90 us at config 5: a line that a lane's
// nine loads touch is still in the L1 for the next of them.)
template <int G>
__global__ __launch_bounds__(256) void k_ba_backsub(BaDev D, double lambda)
{
    const int gt = blockIdx.x * 256 + threadIdx.x;
    const int l = gt / G, g = gt - l * G;
    const bool live = l < D.L;
    const long long ll = live ? l : 0;
    double t0 = 0, t1 = 0, t2 = 0;
    const int e0 = live ? D.pt_first[l] : 0, e1 = live ? D.pt_first[l + 1] : 0;
    for (int e = e0 + g; e < e1; e += G) {
        const int f = D.free_of[D.edge_pose[e]];
        if (f < 0 || !D.active[e]) continue;
        const double2* Zv = reinterpret_cast<const double2*>(D.Z + 18 * (long long)e);
        double Zi[18];
#pragma unroll
        for (int i = 0; i < 9; i++) { const double2 v = Zv[i]; Zi[2 * i] = v.x; Zi[2 * i + 1] = v.y; }
        const double* xp = D.x + 6 * f;
        for (int i = 0; i < 6; i++) { t0 += Zi[i * 3] * xp[i]; t1 += Zi[i * 3 + 1] * xp[i]; t2 += Zi[i * 3 + 2] * xp[i]; }
    }
    if (G > 1) { t0 = ba_group_sum<G>(t0); t1 = ba_group_sum<G>(t1); t2 = ba_group_sum<G>(t2); }
    if (!live || g != 0) return;
    double f[6];
    ba_chol3(D.Hll + 9 * ll, lambda, f);
    const double y2 = t2 * f[5];                                      // L^T y = t
    const double y1 = (t1 - f[4] * y2) * f[3];
    const double y0 = (t0 - f[1] * y1 - f[2] * y2) * f[0];
    const double* db = D.db + 3 * ll;
    double* xl = D.x + 6LL * D.nfree + 3 * ll;
    xl[0] = db[0] - y0; xl[1] = db[1] - y1; xl[2] = db[2] - y2;
}

// The state before the step goes to save_poses / save_points (the LM loop's push: a rejected step copies it back), and the
// new poses' rotation matrices are written where k_ba_pose_rt would write them (same arithmetic) -- a local BA is bound by its
// launches, and these were three of the sixteen of a trial.
__global__ __launch_bounds__(256) void k_ba_update(BaDev D, double* __restrict__ save_poses, double* __restrict__ save_points)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < D.nfree) {
        const int p = D.pose_of_free[i];
        double o[7], old[7];
        for (int k = 0; k < 7; k++) old[k] = D.poses[7 * (long long)p + k];
        for (int k = 0; k < 7; k++) save_poses[7 * (long long)p + k] = old[k];
        ba_se3_exp_mul(D.x + 6 * i, old, o);
        for (int k = 0; k < 7; k++) D.poses[7 * (long long)p + k] = o[k];
        double R[9];
        ba_quat_to_R(o, R);
        double* rt = D.Rt + 12 * (long long)p;
        for (int k = 0; k < 9; k++) rt[k] = R[k];
        rt[9] = o[4]; rt[10] = o[5]; rt[11] = o[6];
    }
    if (i < D.L)
        for (int k = 0; k < 3; k++) {
            const double v = D.points[3 * (long long)i + k];
            save_points[3 * (long long)i + k] = v;
            D.points[3 * (long long)i + k] = v + D.x[6LL * D.nfree + 3 * (long long)i + k];
        }
}

// computeScale partial (optimization_algorithm_levenberg.cpp:182-189):
//   sum_poses x*b_p (+ lambda x^2 if add_pose_lambda) + sum_landmarks x*(lambda x + b_l) -> partial[block]
__global__ __launch_bounds__(256) void k_ba_scale(BaDev D, double lambda, int add_pose_lambda, double* __restrict__ partial)
{
    __shared__ double red[4];
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    const long long np = 6LL * D.nfree, nx = np + 3LL * D.L;
    double v = 0;
    if (i < np) v = D.x[i] * ((add_pose_lambda ? lambda * D.x[i] : 0.0) + D.bp[i]);
    else if (i < nx) v = D.x[i] * (lambda * D.x[i] + D.bl[i - np]);
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// max |diagonal| of Hll (local landmarks) and of Hpp -> partial arrays (computeLambdaInit :166-180)
__global__ __launch_bounds__(256) void k_ba_diag(BaDev D, double* __restrict__ out_ll, double* __restrict__ out_pp_diag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < D.L) {
        const double* H = D.Hll + 9 * (long long)i;
        out_ll[i] = fmax(fabs(H[0]), fmax(fabs(H[4]), fabs(H[8])));
    }
    if (i < D.nfree * 6) out_pp_diag[i] = D.Hpp[36 * (long long)(i / 6) + 7 * (i % 6)];
}

// outlier test of src/Optimizer.cpp:556 / :582 from the LAST COMPUTED errors (g2o does not recompute
// them after a rejected step or for level-1 edges) and the current depth.
__global__ __launch_bounds__(256) void k_ba_outliers(BaDev D, double chi2_th, uint8_t* __restrict__ flag)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= D.E) return;
    const int pi = D.edge_pose[e], li = D.edge_point[e];
    const double* Rt = D.Rt + 12 * (long long)pi; const double* p = D.points + 3 * (long long)li;
    const double z = Rt[6] * p[0] + Rt[7] * p[1] + Rt[8] * p[2] + Rt[11];
    const double e0 = D.err[2 * (long long)e], e1 = D.err[2 * (long long)e + 1];
    const double c2 = D.info[e] * (e0 * e0 + e1 * e1);
    flag[e] = (c2 > chi2_th || !(z > 0.0)) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_ba_deactivate(BaDev D, const uint8_t* __restrict__ flag)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < D.E && flag[e]) D.active[e] = 0;
}

// ---- launchers
static inline int nblk(long long n, int b) { return (int)((n + b - 1) / b); }
void ba_launch_pose_rt(hipStream_t s, const BaDev& D) { if (D.P > 0) hipLaunchKernelGGL(k_ba_pose_rt, dim3(nblk(D.P, 256)), dim3(256), 0, s, D); }
int ba_errors_blocks(const BaDev& D) { return nblk(D.E, 256); }
void ba_launch_errors(hipStream_t s, const BaDev& D, double hd, double* partial, double* out)
{
    const int nb = nblk(D.E, 256);
    if (nb > 0) hipLaunchKernelGGL(k_ba_errors, dim3(nb), dim3(256), 0, s, D, hd, partial);
    hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(256), 0, s, partial, nb, out, 0);
}
#ifndef BA_LM_LANES
#define BA_LM_LANES 8                     // lanes that share a landmark in k_ba_lin_landmark / k_ba_backsub (measured: see DESIGN.md)
#endif
// lambda > 0: the landmarks' share of the Schur step for that lambda is computed along (MODE 1, or MODE 2 when the caller can do
// without Hpl); landmarks_only: the rebuild of Hpl after a rejected trial of a MODE 2 iteration
void ba_launch_lin_pose(hipStream_t s, const BaDev& D, double hd);
void ba_launch_linearize(hipStream_t s, const BaDev& D, double hd, double lambda, bool keep_hpl, bool landmarks_only)
{
    const dim3 grid(nblk((long long)BA_LM_LANES * D.L, 256));                                       // a zero-size grid is a launch error
    if (D.L > 0 && lambda > 0 && keep_hpl) hipLaunchKernelGGL((k_ba_lin_landmark<BA_LM_LANES, 1>), grid, dim3(256), 0, s, D, hd, lambda);
    else if (D.L > 0 && lambda > 0) hipLaunchKernelGGL((k_ba_lin_landmark<BA_LM_LANES, 2>), grid, dim3(256), 0, s, D, hd, lambda);
    else if (D.L > 0) hipLaunchKernelGGL((k_ba_lin_landmark<BA_LM_LANES, 0>), grid, dim3(256), 0, s, D, hd, 0.0);
    if (landmarks_only) return;
    ba_launch_lin_pose(s, D, hd);
}
// the keyframes' side alone: Hpp, bp of the state as it is (also launched ahead of time by the LM loop: ba_host.cpp)
void ba_launch_lin_pose(hipStream_t s, const BaDev& D, double hd)
{
    // few keyframes with long edge lists (local BA): a workgroup per keyframe; maps with thousands of keyframes: a wave each (measured:
    // the workgroup form costs config 5 another 60 us per linearisation, the wave form costs config 4 40 us)
    // (16 waves per keyframe measured no faster than 4 at 20 keyframes x 1300 edges: 22.1 against 21.1 us -- the keyframe's edges are
    //  a CU's worth of f64 work either way; more CUs per keyframe would need a second reduction stage)
    if (D.nfree > 0 && D.nfree < 512) hipLaunchKernelGGL(k_ba_lin_pose<4>, dim3(D.nfree), dim3(256), 0, s, D, hd);
    else if (D.nfree > 0) hipLaunchKernelGGL(k_ba_lin_pose_wave, dim3(nblk(D.nfree, 4)), dim3(256), 0, s, D, hd);
}
// ---- the edge list's index structures made on the device (large unsharded maps: the host passes over 1.8 M edges were 2 ms of a 39 ms call)
// flags[0] |= 1 when the list is not sorted by (landmark, keyframe); flags[1] = lowest edge with a vertex index out of range (or INT_MAX)
__global__ __launch_bounds__(256) void k_ix_check(const int* __restrict__ edge_pose, const int* __restrict__ edge_point, int E, int P, int L, int* __restrict__ flags)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int l = edge_point[e], p = edge_pose[e];
    if (p < 0 || p >= P || l < 0 || l >= L) { atomicMin(flags + 1, e); return; }
    if (e > 0) {
        const int lp = edge_point[e - 1], pp = edge_pose[e - 1];
        if (l < lp || (l == lp && p < pp)) atomicOr(flags, 1);
    }
}
// pt_first[q] = first edge of landmark q (edges sorted by landmark): where the landmark index changes; landmarks without edges start
// where the next one does
__global__ __launch_bounds__(256) void k_ix_pt_first(const int* __restrict__ edge_point, int E, int L, int* __restrict__ pt_first)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e > E) return;
    const int l = e < E ? min(max(edge_point[e], 0), L - 1) : L;          // (clamped: an out-of-range list is rejected by the caller anyway)
    const int lp = e > 0 ? min(max(edge_point[e - 1], 0), L - 1) : -1;
    for (int q = lp + 1; q <= l; q++) pt_first[q] = e;
}
__global__ __launch_bounds__(256) void k_ix_pose_keys(const int* __restrict__ edge_pose, const int* __restrict__ free_of, int E, int P, int nfree,
                                                      unsigned* __restrict__ key, unsigned* __restrict__ val)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int p = edge_pose[e];
    const int f = (p >= 0 && p < P) ? free_of[p] : -1;
    key[e] = f >= 0 ? (unsigned)f : (unsigned)nfree;                      // fixed keyframes' edges sort behind the free ones
    val[e] = (unsigned)e;
}
// pose_first[f] = first position of key f in the sorted keys; pose_first[nfree] = number of edges of free keyframes
__global__ __launch_bounds__(256) void k_ix_pose_first(const unsigned* __restrict__ skey, int E, int nfree, int* __restrict__ pose_first)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > E) return;
    const int f = k < E ? (int)skey[k] : nfree + 1;
    const int fp = k > 0 ? (int)skey[k - 1] : -1;
    for (int q = fp + 1; q <= min(f, nfree); q++) pose_first[q] = k;
}
void ba_launch_index_check(hipStream_t s, const int* edge_pose, const int* edge_point, int E, int P, int L, int* flags, int* pt_first)
{
    hipLaunchKernelGGL(k_ix_check, dim3(nblk(E, 256)), dim3(256), 0, s, edge_pose, edge_point, E, P, L, flags);
    hipLaunchKernelGGL(k_ix_pt_first, dim3(nblk((long long)E + 1, 256)), dim3(256), 0, s, edge_point, E, L, pt_first);
}
void ba_launch_index_pose_keys(hipStream_t s, const int* edge_pose, const int* free_of, int E, int P, int nfree, unsigned* key, unsigned* val)
{
    hipLaunchKernelGGL(k_ix_pose_keys, dim3(nblk(E, 256)), dim3(256), 0, s, edge_pose, free_of, E, P, nfree, key, val);
}
void ba_launch_index_pose_first(hipStream_t s, const unsigned* skey, int E, int nfree, int* pose_first)
{
    hipLaunchKernelGGL(k_ix_pose_first, dim3(nblk((long long)E + 1, 256)), dim3(256), 0, s, skey, E, nfree, pose_first);
}
void ba_launch_backsub(hipStream_t s, const BaDev& D, double lambda)
{
    if (D.L > 0) hipLaunchKernelGGL(k_ba_backsub<BA_LM_LANES>, dim3(nblk((long long)BA_LM_LANES * D.L, 256)), dim3(256), 0, s, D, lambda);
}
void ba_launch_update(hipStream_t s, const BaDev& D, double* save_poses, double* save_points)
{
    const int n = D.L > D.nfree ? D.L : D.nfree;
    if (n > 0) hipLaunchKernelGGL(k_ba_update, dim3(nblk(n, 256)), dim3(256), 0, s, D, save_poses, save_points);
}
int ba_scale_blocks(const BaDev& D) { return nblk(6LL * D.nfree + 3LL * D.L, 256); }
// chi2 of the state after the update (-> out[0]) and the trial's gain denominator (-> out[1]); partial: room for both kernels' blocks
void ba_launch_errors_scale(hipStream_t s, const BaDev& D, double hd, double lambda, int add_pose_lambda, double* partial, double* out)
{
    const int nbe = nblk(D.E, 256), nbs = ba_scale_blocks(D);
    if (nbe > 0) hipLaunchKernelGGL(k_ba_errors, dim3(nbe), dim3(256), 0, s, D, hd, partial);
    if (nbs > 0) hipLaunchKernelGGL(k_ba_scale, dim3(nbs), dim3(256), 0, s, D, lambda, add_pose_lambda, partial + nbe);
    hipLaunchKernelGGL(k_ba_reduce2, dim3(2), dim3(256), 0, s, (const double*)partial, nbe, out, (const double*)(partial + nbe), nbs, out + 1);
}
void ba_launch_scale(hipStream_t s, const BaDev& D, double lambda, int add_pose_lambda, double* partial, double* out)
{
    const int nb = ba_scale_blocks(D);
    if (nb > 0) hipLaunchKernelGGL(k_ba_scale, dim3(nb), dim3(256), 0, s, D, lambda, add_pose_lambda, partial);
    hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(256), 0, s, partial, nb, out, 0);
}
void ba_launch_diag(hipStream_t s, const BaDev& D, double* tmp_ll, double* pp_diag, double* out_ll_max)
{
    const int n = D.L > D.nfree * 6 ? D.L : D.nfree * 6;
    if (n > 0) hipLaunchKernelGGL(k_ba_diag, dim3(nblk(n, 256)), dim3(256), 0, s, D, tmp_ll, pp_diag);
    hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(256), 0, s, tmp_ll, D.L, out_ll_max, 1);
}
void ba_launch_outliers(hipStream_t s, const BaDev& D, double th, uint8_t* flag) { if (D.E > 0) hipLaunchKernelGGL(k_ba_outliers, dim3(nblk(D.E, 256)), dim3(256), 0, s, D, th, flag); }
void ba_launch_deactivate(hipStream_t s, const BaDev& D, const uint8_t* flag) { if (D.E > 0) hipLaunchKernelGGL(k_ba_deactivate, dim3(nblk(D.E, 256)), dim3(256), 0, s, D, flag); }
