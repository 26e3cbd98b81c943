// ess_kernels.hip -- the optimisation inside Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion
// (cslam/src/Optimizer.cpp:1064-1331, :1333-1574): 7-DoF pose graph of VertexSim3Expmap / EdgeSim3 with g2o's numeric
// Jacobians (delta 1e-9, core/base_binary_edge.hpp:147-196) on both vertices.
//   k_ess_errors     thread = (edge, variant): variant 0 = the error, 1..14 = vertex 0 perturbed by +-delta along d,
//                    15..28 = vertex 1 likewise; error = log(Sji * Siw * Sjw^-1)   (types_seven_dof_expmap.h:119-127)
//   k_ess_blocks     thread = edge: Jacobians from the 29 errors, J^T J blocks and J^T e
//   k_ess_assemble   thread = free vertex: diagonal block and gradient = sums over its incident edges in edge order;
//                    off-diagonal blocks are added per edge (one edge per vertex pair: a single, exact addition)
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

// inc_ptr/inc_list: per vertex its incident (edge << 1 | role) entries in edge order; fidx[v] = free index or -1
__global__ __launch_bounds__(64) void k_ess_assemble(int nv, const int* __restrict__ fidx, const int* __restrict__ inc_ptr, const int* __restrict__ inc_list,
                                                     const double* __restrict__ blocks, const double* __restrict__ grad, int N,
                                                     double* __restrict__ H, double* __restrict__ b)
{
    const int v = blockIdx.x, t = threadIdx.x;
    const int f = fidx[v];
    if (f < 0) return;
    if (t < 49) {
        double a = 0;
        for (int q = inc_ptr[v]; q < inc_ptr[v + 1]; q++) { const int en = inc_list[q]; a += blocks[147LL * (en >> 1) + 49 * (en & 1) + t]; }
        H[(long long)(7 * f + t / 7) * N + 7 * f + t % 7] = a;
    } else if (t < 56) {
        double g = 0;
        for (int q = inc_ptr[v]; q < inc_ptr[v + 1]; q++) { const int en = inc_list[q]; g += grad[14LL * (en >> 1) + 7 * (en & 1) + (t - 49)]; }
        b[7 * f + (t - 49)] = g;
    }
}
__global__ __launch_bounds__(64) void k_ess_offdiag(int ne, const int* __restrict__ ei, const int* __restrict__ ej, const int* __restrict__ fidx,
                                                    const double* __restrict__ blocks, int N, double* __restrict__ H)
{
    const int k = blockIdx.x, t = threadIdx.x;
    if (k >= ne || t >= 49) return;
    const int fi = fidx[ei[k]], fj = fidx[ej[k]];
    if (fi < 0 || fj < 0 || fi == fj) return;
    const int p = t / 7, q = t % 7;
    const double c = blocks[147LL * k + 98 + t];                      // (Ji^T Jj)[p][q]
    unsafeAtomicAdd(&H[(long long)(7 * fi + p) * N + 7 * fj + q], c);
    unsafeAtomicAdd(&H[(long long)(7 * fj + q) * N + 7 * fi + p], c);
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
__global__ __launch_bounds__(256) void k_ess_add_lambda(int N, double lambda, double* __restrict__ H)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < N) H[(long long)j * N + j] += lambda;
}

void ess_launch_errors(hipStream_t s, int ne, const int* ei, const int* ej, const double* meas, const double* sim3, const uint8_t* fixed, int fix_scale,
                       int variants, double* err)
{
    const long long n = (long long)ne * variants;
    if (n > 0) hipLaunchKernelGGL(k_ess_errors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ne, ei, ej, meas, sim3, fixed, fix_scale, variants, err);
}
void ess_launch_system(hipStream_t s, int ne, int nv, const int* ei, const int* ej, const int* fidx, const int* inc_ptr, const int* inc_list,
                       const double* err, double* blocks, double* grad, int N, double* H, double* b)
{
    hipLaunchKernelGGL(k_ess_blocks, dim3((ne + 63) / 64), dim3(64), 0, s, ne, err, blocks, grad);
    hipLaunchKernelGGL(k_ess_assemble, dim3(nv), dim3(64), 0, s, nv, fidx, inc_ptr, inc_list, blocks, grad, N, H, b);
    hipLaunchKernelGGL(k_ess_offdiag, dim3(ne), dim3(64), 0, s, ne, ei, ej, fidx, blocks, N, H);
}
void ess_launch_update(hipStream_t s, int nv, const int* fidx, const double* x, int fix_scale, double* sim3)
{ hipLaunchKernelGGL(k_ess_update, dim3((nv + 255) / 256), dim3(256), 0, s, nv, fidx, x, fix_scale, sim3); }
void ess_launch_chi2(hipStream_t s, int ne, const double* err, int stride, double* part, double* out)
{
    const int nb = (ne + 255) / 256;
    hipLaunchKernelGGL(k_ess_chi2, dim3(nb), dim3(256), 0, s, ne, err, stride, part);
    hipLaunchKernelGGL(k_ess_chi2_fin, dim3(1), dim3(64), 0, s, nb, part, out);
}
void ess_launch_add_lambda(hipStream_t s, int N, double lambda, double* H)
{ hipLaunchKernelGGL(k_ess_add_lambda, dim3((N + 255) / 256), dim3(256), 0, s, N, lambda, H); }

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
