// ess_kernels.hip -- the optimisation inside Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion
// (cslam/src/Optimizer.cpp:1064-1331, :1333-1574): 7-DoF pose graph of VertexSim3Expmap / EdgeSim3 with g2o's numeric
// Jacobians (delta 1e-9, core/base_binary_edge.hpp:147-196) on both vertices.
//   k_ess_errors     thread = (edge, variant): variant 0 = the error, 1..14 = vertex 0 perturbed by +-delta along d,
//                    15..28 = vertex 1 likewise; error = log(Sji * Siw * Sjw^-1)   (types_seven_dof_expmap.h:119-127)
//   k_ess_blocks     thread = edge: Jacobians from the 29 errors, J^T J blocks and J^T e
//   k_essp_*         block-sparse normal equations on the graph's pattern and their Cholesky factorisation / solves
//   k_ess_update     VertexSim3Expmap::oplusImpl per free vertex
//   k_ess_chi2       sum |e|^2 in a fixed order
#include <hip/hip_runtime.h>
#include <cstdint>
#include "sim3_math.h"

__device__ void ess_error(const double* C, const double* Si, const double* Sj, double* e)
{
    double t1[8], inv[8], t2[8];
    s3_mul(C, Si, t1);
    s3_inverse(Sj, inv);
    s3_mul(t1, inv, t2);
    s3_log(t2, e);
}

__global__ __launch_bounds__(256) void k_ess_errors(int ne, const int* __restrict__ ei, const int* __restrict__ ej, const double* __restrict__ meas,
                                                    const double* __restrict__ sim3, const uint8_t* __restrict__ fixed, int fix_scale, int variants,
                                                    double* __restrict__ err /* [ne][variants][7] */)
{
    const long long t = blockIdx.x * 256LL + threadIdx.x;
    if (t >= (long long)ne * variants) return;
    const int k = (int)(t / variants), v = (int)(t - (long long)k * variants);
    double C[8], Si[8], Sj[8];
    for (int i = 0; i < 8; i++) { C[i] = meas[8 * (long long)k + i]; Si[i] = sim3[8 * (long long)ei[k] + i]; Sj[i] = sim3[8 * (long long)ej[k] + i]; }
    double e[7] = { 0, 0, 0, 0, 0, 0, 0 };
    if (v == 0) ess_error(C, Si, Sj, e);
    else {
        const int which = (v - 1) / 14, d = ((v - 1) % 14) >> 1, neg = (v - 1) & 1;
        if (!fixed[which == 0 ? ei[k] : ej[k]]) {
            double up[7] = { 0, 0, 0, 0, 0, 0, 0 }, Sp[8];
            up[d] = neg ? -1e-9 : 1e-9;
            s3_oplus(which == 0 ? Si : Sj, up, fix_scale, Sp);
            ess_error(C, which == 0 ? Sp : Si, which == 0 ? Sj : Sp, e);
        }
    }
    for (int r = 0; r < 7; r++) err[(t * 7) + r] = e[r];
}

// per edge: blocks [k][0] = Ji^T Ji, [1] = Jj^T Jj, [2] = Ji^T Jj (row-major 7x7), grad [k][0..6] = -Ji^T e, [7..13] = -Jj^T e
__global__ __launch_bounds__(64) void k_ess_blocks(int ne, const double* __restrict__ err, double* __restrict__ blocks, double* __restrict__ grad)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= ne) return;
    const double* E = err + 29LL * 7 * k;
    double J[2][49];
    const double scalar = 1.0 / (2 * 1e-9);
    for (int v = 0; v < 2; v++)
        for (int d = 0; d < 7; d++)
            for (int r = 0; r < 7; r++) J[v][r * 7 + d] = scalar * (E[7 * (1 + 14 * v + 2 * d) + r] - E[7 * (2 + 14 * v + 2 * d) + r]);
    double* B = blocks + 147LL * k;
    for (int p = 0; p < 7; p++)
        for (int q = 0; q < 7; q++) {
            double a = 0, b = 0, c = 0;
            for (int r = 0; r < 7; r++) { a += J[0][r * 7 + p] * J[0][r * 7 + q]; b += J[1][r * 7 + p] * J[1][r * 7 + q]; c += J[0][r * 7 + p] * J[1][r * 7 + q]; }
            B[p * 7 + q] = a; B[49 + p * 7 + q] = b; B[98 + p * 7 + q] = c;
        }
    for (int p = 0; p < 7; p++) {
        double gi = 0, gj = 0;
        for (int r = 0; r < 7; r++) { gi += J[0][r * 7 + p] * (-E[r]); gj += J[1][r * 7 + p] * (-E[r]); }
        grad[14LL * k + p] = gi; grad[14LL * k + 7 + p] = gj;
    }
}

// ------------------------------------------------------------------------------------------------
// Block-sparse normal equations and their Cholesky factorisation (7x7 blocks on the essential graph's pattern), the
// counterpart of g2o's BlockSolver_7_3 + sparse LinearSolverEigen (src/Optimizer.cpp:1072-1074).  The host orders the free
// keyframes by rounds of independent minimum-degree eliminations (ess_host.cpp) and hands over, for every block of the
// factor, the list of (L_ik, L_jk) products it needs; every sum below runs over such a list in its stored order, so the
// result is bit-reproducible.  Storage: D[c][49] diagonal blocks, Lb[slot][49] strictly-lower blocks, row-major, permuted
// order.  Columns of one round are independent: one workgroup per column, one launch per round.
//
// b: per free vertex (original free index) the gradient sum, in edge order
__global__ __launch_bounds__(64) void k_essp_grad(int nv, const int* __restrict__ fidx, const int* __restrict__ inc_ptr, const int* __restrict__ inc_list,
                                                  const double* __restrict__ grad, double* __restrict__ b)
{
    const int v = blockIdx.x, t = threadIdx.x;
    const int f = fidx[v];
    if (f < 0 || t >= 7) return;
    double g = 0;
    for (int q = inc_ptr[v]; q < inc_ptr[v + 1]; q++) { const int en = inc_list[q]; g += grad[14LL * (en >> 1) + 7 * (en & 1) + t]; }
    b[7 * f + t] = g;
}

// A: target t < ncol is the diagonal block of column t, target ncol + s the lower block in slot s.  alist entry = edge * 4 + code,
// code 0: Ji^T Ji, 1: Jj^T Jj, 2: Ji^T Jj, 3: (Ji^T Jj)^T.  Targets the pattern adds by fill-in have empty lists (zero blocks).
__global__ __launch_bounds__(64) void k_essp_assemble(int ntargets, int ncol, const int* __restrict__ aptr, const int* __restrict__ alist,
                                                      const double* __restrict__ blocks, double* __restrict__ D, double* __restrict__ Lb)
{
    const int t = blockIdx.x, e = threadIdx.x;
    if (t >= ntargets || e >= 49) return;
    const int p = e / 7, q = e - 7 * p;
    double a = 0;
    for (int k = aptr[t]; k < aptr[t + 1]; k++) {
        const int en = alist[k], code = en & 3;
        const double* B = blocks + 147LL * (en >> 2);
        a += code == 3 ? B[98 + q * 7 + p] : B[49 * code + e];
    }
    if (t < ncol) D[49LL * t + e] = a; else Lb[49LL * (t - ncol) + e] = a;
}

// One round of the left-looking factorisation.  Column j (= cols[blockIdx.x]):
//   D_j  <- chol( A_jj + lambda I - sum_k L_jk L_jk^T )
//   L_ij <- ( A_ij - sum_k L_ik L_jk^T ) D_j^-T            for every block (i, j) of the column
// tptr / tpa / tpb: per target (numbered as in k_essp_assemble) the slots of the two factors of each product, ascending k.
__global__ __launch_bounds__(256) void k_essp_factor(const int* __restrict__ cols, const int* __restrict__ colptr, int ncol, const int* __restrict__ tptr,
                                                     const int* __restrict__ tpa, const int* __restrict__ tpb, double lambda,
                                                     double* __restrict__ D, double* __restrict__ Lb, int* __restrict__ bad)
{
    __shared__ double Ljj[49];
    const int j = cols[blockIdx.x], tid = threadIdx.x;
    if (tid < 49) {
        const int r = tid / 7, c2 = tid - 7 * r;
        double a = D[49LL * j + tid] + (r == c2 ? lambda : 0.0);
        // the update terms in list order, four at a time: their index loads are issued together, then their block rows (a late round's
        // column has 60-100 terms; one after the other they were 60-100 dependent round trips, 70 us per round)
        const int k1 = tptr[j + 1];
        for (int k = tptr[j]; k < k1; k += 4) {
            int ia[4], ib[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int kk = min(k + u, k1 - 1); ia[u] = tpa[kk]; ib[u] = tpb[kk]; }
            double va[4][7], vb[4][7];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double* A = Lb + 49LL * ia[u] + 7 * r; const double* B = Lb + 49LL * ib[u] + 7 * c2;
#pragma unroll
                for (int q = 0; q < 7; q++) { va[u][q] = A[q]; vb[u][q] = B[q]; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (k + u < k1) {
                    double v = 0;
#pragma unroll
                    for (int q = 0; q < 7; q++) v += va[u][q] * vb[u][q];
                    a -= v;
                }
            }
        }
        Ljj[tid] = a;
    }
    __syncthreads();
    if (tid == 0) {                                       // 7x7 Cholesky, lower, in place
        for (int c2 = 0; c2 < 7; c2++) {
            double d = Ljj[c2 * 7 + c2];
            for (int k = 0; k < c2; k++) d -= Ljj[c2 * 7 + k] * Ljj[c2 * 7 + k];
            if (!(d > 0.0)) { atomicOr(bad, 1); d = 1.0; }
            d = sqrt(d); Ljj[c2 * 7 + c2] = d;
            const double id = 1.0 / d;
            for (int r = c2 + 1; r < 7; r++) {
                double v = Ljj[r * 7 + c2];
                for (int k = 0; k < c2; k++) v -= Ljj[r * 7 + k] * Ljj[c2 * 7 + k];
                Ljj[r * 7 + c2] = v * id;
            }
            for (int r = 0; r < c2; r++) Ljj[r * 7 + c2] = 0.0;
        }
    }
    __syncthreads();
    if (tid < 49) D[49LL * j + tid] = Ljj[tid];
    // thread = (block of the column, row of the block): the row's 7 sums, then its forward substitution against Ljj
    const int c0 = colptr[j], nb = colptr[j + 1] - c0;
    for (int w = tid; w < nb * 7; w += 256) {
        const int s = c0 + w / 7, r = w % 7;
        double row[7];
#pragma unroll
        for (int q = 0; q < 7; q++) row[q] = Lb[49LL * s + 7 * r + q];
        const int t = ncol + s;
        const int k1 = tptr[t + 1];
        for (int k = tptr[t]; k < k1; k += 2) {                       // two terms at a time (index loads, then rows, then sums; list order kept)
            int ia[2], ib[2];
#pragma unroll
            for (int u = 0; u < 2; u++) { const int kk = min(k + u, k1 - 1); ia[u] = tpa[kk]; ib[u] = tpb[kk]; }
            double av[2][7], bv[2][49];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const double* A = Lb + 49LL * ia[u] + 7 * r; const double* B = Lb + 49LL * ib[u];
#pragma unroll
                for (int q = 0; q < 7; q++) av[u][q] = A[q];
#pragma unroll
                for (int q = 0; q < 49; q++) bv[u][q] = B[q];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (k + u < k1) {
#pragma unroll
                    for (int c2 = 0; c2 < 7; c2++) {
                        double v = 0;
#pragma unroll
                        for (int q = 0; q < 7; q++) v += av[u][q] * bv[u][7 * c2 + q];
                        row[c2] -= v;
                    }
                }
            }
        }
#pragma unroll
        for (int c2 = 0; c2 < 7; c2++) {
            double v = row[c2];
            for (int q = 0; q < c2; q++) v -= row[q] * Ljj[c2 * 7 + q];
            row[c2] = v / Ljj[c2 * 7 + c2];
        }
#pragma unroll
        for (int q = 0; q < 7; q++) Lb[49LL * s + 7 * r + q] = row[q];
    }
}

// forward round: y_j = D_j^-1 ( b_perm(j) - sum_{k < j} L_jk y_k ); rptr / rslot / rcol: row j's blocks, ascending k
__global__ __launch_bounds__(64) void k_essp_forward(const int* __restrict__ cols, const int* __restrict__ perm, const int* __restrict__ rptr,
                                                     const int* __restrict__ rslot, const int* __restrict__ rcol, const double* __restrict__ D,
                                                     const double* __restrict__ Lb, const double* __restrict__ b, double* __restrict__ y)
{
    __shared__ double acc[7];
    const int j = cols[blockIdx.x], t = threadIdx.x;
    if (t < 7) {
        double a = b[7LL * perm[j] + t];
        for (int k = rptr[j]; k < rptr[j + 1]; k++) {
            const double* L = Lb + 49LL * rslot[k] + 7 * t; const double* yk = y + 7LL * rcol[k];
            double v = 0;
#pragma unroll
            for (int q = 0; q < 7; q++) v += L[q] * yk[q];
            a -= v;
        }
        acc[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        const double* Ljj = D + 49LL * j;
        double o[7];
        for (int r = 0; r < 7; r++) { double v = acc[r]; for (int q = 0; q < r; q++) v -= Ljj[r * 7 + q] * o[q]; o[r] = v / Ljj[r * 7 + r]; }
        for (int r = 0; r < 7; r++) y[7LL * j + r] = o[r];
    }
}
// backward round: x_j = D_j^-T ( y_j - sum_{i > j} L_ij^T x_i ); the column's blocks in row order; x leaves in ORIGINAL free order too
__global__ __launch_bounds__(64) void k_essp_backward(const int* __restrict__ cols, const int* __restrict__ perm, const int* __restrict__ colptr,
                                                      const int* __restrict__ rowidx, const double* __restrict__ D, const double* __restrict__ Lb,
                                                      const double* __restrict__ y, double* __restrict__ xp, double* __restrict__ x)
{
    __shared__ double acc[7];
    const int j = cols[blockIdx.x], t = threadIdx.x;
    if (t < 7) {
        double a = y[7LL * j + t];
        for (int s = colptr[j]; s < colptr[j + 1]; s++) {
            const double* L = Lb + 49LL * s; const double* xi = xp + 7LL * rowidx[s];
            double v = 0;
#pragma unroll
            for (int r = 0; r < 7; r++) v += L[7 * r + t] * xi[r];
            a -= v;
        }
        acc[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        const double* Ljj = D + 49LL * j;
        double o[7];
        for (int r = 6; r >= 0; r--) { double v = acc[r]; for (int q = r + 1; q < 7; q++) v -= Ljj[q * 7 + r] * o[q]; o[r] = v / Ljj[r * 7 + r]; }
        for (int r = 0; r < 7; r++) { xp[7LL * j + r] = o[r]; x[7LL * perm[j] + r] = o[r]; }
    }
}

__global__ __launch_bounds__(256) void k_ess_update(int nv, const int* __restrict__ fidx, const double* __restrict__ x, int fix_scale, double* __restrict__ sim3)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const int f = fidx[v];
    if (f < 0) return;
    double S[8], u[7], o[8];
    for (int i = 0; i < 8; i++) S[i] = sim3[8LL * v + i];
    for (int i = 0; i < 7; i++) u[i] = x[7LL * f + i];
    s3_oplus(S, u, fix_scale, o);
    for (int i = 0; i < 8; i++) sim3[8LL * v + i] = o[i];
}

// chi2 = sum over edges of |e|^2 (information = identity): per-block partials, then one thread adds them in order
__global__ __launch_bounds__(256) void k_ess_chi2(int ne, const double* __restrict__ err, int stride, double* __restrict__ part)
{
    __shared__ double red[256];
    const int k = blockIdx.x * 256 + threadIdx.x;
    double c = 0;
    if (k < ne) for (int r = 0; r < 7; r++) { const double e = err[(long long)stride * k + r]; c += e * e; }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_ess_chi2_fin(int nb, const double* __restrict__ part, double* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0;
    for (int i = 0; i < nb; i++) s += part[i];
    *out = s;
}
void ess_launch_errors(hipStream_t s, int ne, const int* ei, const int* ej, const double* meas, const double* sim3, const uint8_t* fixed, int fix_scale,
                       int variants, double* err)
{
    const long long n = (long long)ne * variants;
    if (n > 0) hipLaunchKernelGGL(k_ess_errors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ne, ei, ej, meas, sim3, fixed, fix_scale, variants, err);
}
void ess_launch_blocks(hipStream_t s, int ne, int nv, const int* fidx, const int* inc_ptr, const int* inc_list, const double* err, double* blocks,
                       double* grad, double* b)
{
    hipLaunchKernelGGL(k_ess_blocks, dim3((ne + 63) / 64), dim3(64), 0, s, ne, err, blocks, grad);
    hipLaunchKernelGGL(k_essp_grad, dim3(nv), dim3(64), 0, s, nv, fidx, inc_ptr, inc_list, grad, b);
}
void essp_launch_assemble(hipStream_t s, int ntargets, int ncol, const int* aptr, const int* alist, const double* blocks, double* D, double* Lb)
{ if (ntargets > 0) hipLaunchKernelGGL(k_essp_assemble, dim3(ntargets), dim3(64), 0, s, ntargets, ncol, aptr, alist, blocks, D, Lb); }
void essp_launch_factor_round(hipStream_t s, const int* cols, int n, const int* colptr, int ncol, const int* tptr, const int* tpa, const int* tpb,
                              double lambda, double* D, double* Lb, int* bad)
{ if (n > 0) hipLaunchKernelGGL(k_essp_factor, dim3(n), dim3(256), 0, s, cols, colptr, ncol, tptr, tpa, tpb, lambda, D, Lb, bad); }
void essp_launch_forward_round(hipStream_t s, const int* cols, int n, const int* perm, const int* rptr, const int* rslot, const int* rcol,
                               const double* D, const double* Lb, const double* b, double* y)
{ if (n > 0) hipLaunchKernelGGL(k_essp_forward, dim3(n), dim3(64), 0, s, cols, perm, rptr, rslot, rcol, D, Lb, b, y); }
void essp_launch_backward_round(hipStream_t s, const int* cols, int n, const int* perm, const int* colptr, const int* rowidx, const double* D,
                                const double* Lb, const double* y, double* xp, double* x)
{ if (n > 0) hipLaunchKernelGGL(k_essp_backward, dim3(n), dim3(64), 0, s, cols, perm, colptr, rowidx, D, Lb, y, xp, x); }
void ess_launch_update(hipStream_t s, int nv, const int* fidx, const double* x, int fix_scale, double* sim3)
{ hipLaunchKernelGGL(k_ess_update, dim3((nv + 255) / 256), dim3(256), 0, s, nv, fidx, x, fix_scale, sim3); }
void ess_launch_chi2(hipStream_t s, int ne, const double* err, int stride, double* part, double* out)
{
    const int nb = (ne + 255) / 256;
    hipLaunchKernelGGL(k_ess_chi2, dim3(nb), dim3(256), 0, s, ne, err, stride, part);
    hipLaunchKernelGGL(k_ess_chi2_fin, dim3(1), dim3(64), 0, s, nb, part, out);
}

// Map point correction after the pose graph (src/Optimizer.cpp:1300-1330): P' = correctedSwr.map(Srw.map(P))
__global__ __launch_bounds__(256) void k_ess_correct_points(int np, const int* __restrict__ ref, const double* __restrict__ s_old, const double* __restrict__ s_new,
                                                            double* __restrict__ pts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const int r = ref[i];
    if (r < 0) return;
    double So[8], Sn[8], Si[8], P[3], Q[3], O[3];
    for (int k = 0; k < 8; k++) { So[k] = s_old[8LL * r + k]; Sn[k] = s_new[8LL * r + k]; }
    for (int k = 0; k < 3; k++) P[k] = pts[3LL * i + k];
    s3_rotv(So, P, Q);
    for (int k = 0; k < 3; k++) Q[k] = So[7] * Q[k] + So[4 + k];
    s3_inverse(Sn, Si);
    s3_rotv(Si, Q, O);
    for (int k = 0; k < 3; k++) pts[3LL * i + k] = Si[7] * O[k] + Si[4 + k];
}
void ess_launch_correct(hipStream_t s, int np, const int* ref, const double* s_old, const double* s_new, double* pts)
{ if (np > 0) hipLaunchKernelGGL(k_ess_correct_points, dim3((np + 255) / 256), dim3(256), 0, s, np, ref, s_old, s_new, pts); }
