// sim3_kernels.hip -- Optimizer::OptimizeSim3 (cslam/src/Optimizer.cpp:867-1062) on the GPU, batched.
//
// One workgroup owns one (KF1, KF2) candidate pair and runs the whole schedule in one launch: optimize(5), the
// chi2 > th2 pruning, optimize(5 or 10), the final classification.  The graph is one VertexSim3Expmap
// (thirdparty/g2o/g2o/types/types_seven_dof_expmap.h:52-107) and two projection edges per correspondence
// (EdgeSim3ProjectXYZ :146-166, EdgeInverseSim3ProjectXYZ :169-189) whose Jacobians g2o takes NUMERICALLY (central
// differences, delta 1e-9, core/base_binary_edge.hpp:147-196).  The 14 perturbed estimates exp(+-delta e_d) * S and
// their inverses are the same for every edge: they are made once per iteration by 14 lanes and kept in LDS.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cfloat>
#include "sim3_math.h"

#define S3_TPB 256

struct Sim3Dev {
    int n_problems;
    double* sim3;             // [n][8] in/out: qx,qy,qz,qw, tx,ty,tz, s
    const int* fix_scale;     // [n]
    const double* K1; const double* K2;     // [n][4] fx, fy, cx, cy
    const int* first;         // [n+1]
    const double* P1; const double* P2;     // [total][3] map points in their own camera frame
    const double* obs1; const double* obs2; // [total][2]
    const double* info1; const double* info2;
    const float* th2;         // [n]
    double* err;              // [total][4] scratch: last computed e12, e21
    uint8_t* inlier;          // [total] out
    int* n_in;                // [n] out
};

// projection of x through (R|t|s) given as rotation matrix: e = obs - K(project(s R x + t))
__device__ __forceinline__ void s3_proj_err(const double* R, const double* t, double s, const double* K, const double* x, const double* obs, double* e)
{
    double p[3];
    for (int i = 0; i < 3; i++) p[i] = s * (R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2]) + t[i];
    e[0] = obs[0] - ((p[0] / p[2]) * K[0] + K[2]);
    e[1] = obs[1] - ((p[1] / p[2]) * K[1] + K[3]);
}

template <int NV>
__device__ __forceinline__ void s3_block_sum(double* v, double* lds /* [4][NV] */)
{
#pragma unroll
    for (int i = 0; i < NV; i++)
        for (int s = 32; s >= 1; s >>= 1) v[i] += __shfl_xor(v[i], s, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < NV; i++) lds[(threadIdx.x >> 6) * NV + i] = v[i];
    __syncthreads();
    for (int i = 0; i < NV; i++) v[i] = (lds[i] + lds[NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
}

__device__ bool s3_chol7(const double* H, const double* b, double* x)
{
    double L[49];
    for (int i = 0; i < 49; i++) L[i] = H[i];
    for (int j = 0; j < 7; j++) {
        double d = L[j * 7 + j];
        for (int k = 0; k < j; k++) d -= L[j * 7 + k] * L[j * 7 + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        L[j * 7 + j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < 7; i++) {
            double v = L[i * 7 + j];
            for (int k = 0; k < j; k++) v -= L[i * 7 + k] * L[j * 7 + k];
            L[i * 7 + j] = v * id;
        }
    }
    for (int i = 0; i < 7; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= L[i * 7 + k] * x[k];
        x[i] = v / L[i * 7 + i];
    }
    for (int i = 6; i >= 0; i--) {
        double v = x[i];
        for (int k = i + 1; k < 7; k++) v -= L[k * 7 + i] * x[k];
        x[i] = v / L[i * 7 + i];
    }
    return true;
}

// forward (S) and inverse (S^-1) transforms as rotation matrix + t + s: 13 doubles each
struct S3Xf { double R[9], t[3], s; };
__device__ __forceinline__ void s3_xf(const double* S, S3Xf* f)
{
    ba_quat_to_R(S, f->R);
    f->t[0] = S[4]; f->t[1] = S[5]; f->t[2] = S[6]; f->s = S[7];
}

__global__ __launch_bounds__(S3_TPB) void k_sim3_opt(Sim3Dev D)
{
    __shared__ double red[4 * 36];
    __shared__ double s_S[8], s_x[7];
    __shared__ S3Xf s_f[15], s_i[15];            // [0] = current estimate, [1 + 2d], [2 + 2d] = +delta, -delta along d
    __shared__ int s_ok;
    const int pb = blockIdx.x, tid = threadIdx.x;
    const int e0 = D.first[pb], e1 = D.first[pb + 1], n = e1 - e0;
    const int fix_scale = D.fix_scale[pb];
    const float th2 = D.th2[pb];
    const double delta = (double)sqrtf(th2);                               // const float deltaHuber = sqrt(th2), :903
    double K1[4], K2[4];
    for (int i = 0; i < 4; i++) { K1[i] = D.K1[4 * (long long)pb + i]; K2[i] = D.K2[4 * (long long)pb + i]; }
    for (int e = e0 + tid; e < e1; e += S3_TPB) D.inlier[e] = 1;
    if (tid < 8) s_S[tid] = D.sim3[8 * (long long)pb + tid];
    __syncthreads();

    // errors of edge pair e at transform slot k (0 = estimate)
    auto pair_err = [&](int k, int e, double* e12, double* e21) {
        s3_proj_err(s_f[k].R, s_f[k].t, s_f[k].s, K1, D.P2 + 3 * (long long)e, D.obs1 + 2 * (long long)e, e12);
        s3_proj_err(s_i[k].R, s_i[k].t, s_i[k].s, K2, D.P1 + 3 * (long long)e, D.obs2 + 2 * (long long)e, e21);
    };
    auto load_estimate = [&]() {                                           // slot 0 from s_S (call by all, sync inside)
        if (tid == 0) { double Si[8]; s3_xf(s_S, &s_f[0]); s3_inverse(s_S, Si); s3_xf(Si, &s_i[0]); }
        __syncthreads();
    };

    int n_bad = 0;
    for (int round = 0; round < 2; round++) {
        const int iterations = round == 0 ? 5 : (n_bad > 0 ? 10 : 5);
        double lambda = 0, ni = 2;
        int nBad = 0;
        for (int it = 0; it < iterations; it++) {
            // estimate and its 14 perturbations
            if (tid < 15) {
                double Sk[8], Si[8];
                if (tid == 0) for (int i = 0; i < 8; i++) Sk[i] = s_S[i];
                else {
                    double up[7] = { 0, 0, 0, 0, 0, 0, 0 };
                    up[(tid - 1) >> 1] = ((tid - 1) & 1) ? -1e-9 : 1e-9;
                    s3_oplus(s_S, up, fix_scale, Sk);
                }
                s3_xf(Sk, &s_f[tid]); s3_inverse(Sk, Si); s3_xf(Si, &s_i[tid]);
            }
            __syncthreads();
            double acc[36];
            for (int i = 0; i < 36; i++) acc[i] = 0;
            int nact = 0;
            for (int e = e0 + tid; e < e1; e += S3_TPB) {
                if (!D.inlier[e]) continue;
                nact++;
                double er[4];
                pair_err(0, e, er, er + 2);
                for (int i = 0; i < 4; i++) D.err[4 * (long long)e + i] = er[i];
                double J[4][7];
                const double scalar = 1.0 / (2 * 1e-9);
                for (int d = 0; d < 7; d++) {
                    double a[4], b[4];
                    pair_err(1 + 2 * d, e, a, a + 2);
                    pair_err(2 + 2 * d, e, b, b + 2);
                    for (int r = 0; r < 4; r++) J[r][d] = scalar * (a[r] - b[r]);
                }
                for (int k = 0; k < 2; k++) {
                    const double om = k == 0 ? D.info1[e] : D.info2[e];
                    const double r0e = er[2 * k], r1e = er[2 * k + 1];
                    const double c2 = om * (r0e * r0e + r1e * r1e);
                    double r0, r1;
                    ba_huber(c2, delta, &r0, &r1);
                    acc[35] += r0;
                    const double w = r1 * om, g0 = -om * r0e * r1, g1 = -om * r1e * r1;
                    int m = 0;
                    for (int i = 0; i < 7; i++) {
                        acc[28 + i] += J[2 * k][i] * g0 + J[2 * k + 1][i] * g1;
                        for (int j = i; j < 7; j++) acc[m++] += w * (J[2 * k][i] * J[2 * k][j] + J[2 * k + 1][i] * J[2 * k + 1][j]);
                    }
                }
            }
            s3_block_sum<36>(acc, red);
            const int any_active = __syncthreads_or(nact);
            if (!any_active) break;
            double H[49], b[7];
            {
                int m = 0;
                for (int i = 0; i < 7; i++) { b[i] = acc[28 + i]; for (int j = i; j < 7; j++) { H[i * 7 + j] = acc[m]; H[j * 7 + i] = acc[m]; m++; } }
            }
            double cur = acc[35];
            const double ini = cur;
            if (it == 0) {
                double md = 0;
                for (int j = 0; j < 7; j++) md = fmax(md, fabs(H[8 * j]));
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                double save[8];
                for (int i = 0; i < 8; i++) save[i] = s_S[i];
                __syncthreads();
                if (tid == 0) {
                    double Hl[49], x[7] = { 0, 0, 0, 0, 0, 0, 0 };
                    for (int i = 0; i < 49; i++) Hl[i] = H[i];
                    for (int j = 0; j < 7; j++) Hl[8 * j] += lambda;
                    const bool ok = s3_chol7(Hl, b, x);
                    s_ok = ok ? 1 : 0;
                    for (int i = 0; i < 7; i++) s_x[i] = x[i];
                    if (ok) {
                        double o[8];
                        s3_oplus(s_S, x, fix_scale, o);
                        for (int i = 0; i < 8; i++) s_S[i] = o[i];
                    }
                }
                __syncthreads();
                const int ok2 = s_ok;
                double x[7];
                for (int i = 0; i < 7; i++) x[i] = s_x[i];
                double temp = DBL_MAX;
                if (ok2) {
                    load_estimate();
                    double t[1] = { 0 };
                    for (int e = e0 + tid; e < e1; e += S3_TPB) {
                        if (!D.inlier[e]) continue;
                        double er[4];
                        pair_err(0, e, er, er + 2);
                        for (int i = 0; i < 4; i++) D.err[4 * (long long)e + i] = er[i];
                        double r0, r1;
                        ba_huber(D.info1[e] * (er[0] * er[0] + er[1] * er[1]), delta, &r0, &r1); t[0] += r0;
                        ba_huber(D.info2[e] * (er[2] * er[2] + er[3] * er[3]), delta, &r0, &r1); t[0] += r0;
                    }
                    s3_block_sum<1>(t, red);
                    temp = t[0];
                }
                double scale = 1e-3;
                for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + b[j]);
                rho = ok2 ? (cur - temp) / scale : -1.0;
                __syncthreads();
                if (rho > 0 && isfinite(temp)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha); ni = 2; cur = temp;
                } else {
                    lambda *= ni; ni *= 2;
                    if (tid < 8) s_S[tid] = save[tid];                       // pop(): the estimate goes back, the edge errors stay
                }
                __syncthreads();
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;
            if ((ini - cur) * 1e3 < ini) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        // classification on the last computed errors (:989-1006, :1033-1049)
        __syncthreads();
        int bad = 0, good = 0;
        for (int e = e0 + tid; e < e1; e += S3_TPB) {
            if (!D.inlier[e]) continue;
            const double* er = D.err + 4 * (long long)e;
            const double c12 = D.info1[e] * (er[0] * er[0] + er[1] * er[1]);
            const double c21 = D.info2[e] * (er[2] * er[2] + er[3] * er[3]);
            if (c12 > th2 || c21 > th2) { D.inlier[e] = 0; bad++; } else good++;
        }
        double t[2] = { (double)bad, (double)good };
        s3_block_sum<2>(t, red);
        __syncthreads();
        if (round == 0) {
            n_bad = (int)t[0];
            if (n - n_bad < 10) {                                          // :1022-1023: return 0, g2oS12 untouched
                if (tid == 0) D.n_in[pb] = 0;
                return;
            }
        } else {
            if (tid == 0) D.n_in[pb] = (int)t[1];
        }
    }
    if (tid < 8) D.sim3[8 * (long long)pb + tid] = s_S[tid];
}

void sim3_launch(hipStream_t s, const Sim3Dev& D) { hipLaunchKernelGGL(k_sim3_opt, dim3(D.n_problems), dim3(S3_TPB), 0, s, D); }
