// pose_kernels.hip -- Optimizer::PoseOptimizationClient (src/Optimizer.cpp:215-347) on the GPU.
//
// One workgroup owns one frame and runs the WHOLE schedule inside a single launch: 4 rounds x up to 10
// Levenberg iterations x up to 10 trials, each a pass over the frame's <= ~1000 unary edges
// (EdgeSE3ProjectXYZOnlyPose, cslam/thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:266-296) with a
// fixed-order block reduction of the 6x6 normal equations, a 6x6 Cholesky on one lane, the exp-map update
// and the chi2 of the trial.  A batch of frames (e.g. all clients' current frames on the server, or the
// frames of a benchmark) fills the chip; there is no host round trip inside the optimisation.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cfloat>
#include "ba_math.h"

#define PO_TPB 256

struct PoseDev {
    int n_frames;
    double* poses;            // [n_frames][7] in/out
    const double* intr;       // [n_frames][4]
    const int* first;         // [n_frames+1]
    const double* pts;        // [total][3]
    const double* obs;        // [total][2]
    const double* info;       // [total]
    double* err;              // [total][2] scratch: last computed error per edge
    uint8_t* outlier;         // [total] out
    int* n_inliers;           // [n_frames] out
};

// fixed-order block sum of NV doubles per thread -> result broadcast to all threads via LDS
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* lds /* [4][NV] + [NV] */)
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

__device__ __forceinline__ bool chol6(const double* H, const double* b, double* x)
{
    double L[36];
    for (int i = 0; i < 36; i++) L[i] = H[i];
    for (int j = 0; j < 6; j++) {
        double d = L[j * 6 + j];
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        L[j * 6 + j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < 6; i++) {
            double v = L[i * 6 + j];
            for (int k = 0; k < j; k++) v -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = v * id;
        }
    }
    for (int i = 0; i < 6; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= L[i * 6 + k] * x[k];
        x[i] = v / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double v = x[i];
        for (int k = i + 1; k < 6; k++) v -= L[k * 6 + i] * x[k];
        x[i] = v / L[i * 6 + i];
    }
    return true;
}

__global__ __launch_bounds__(PO_TPB) void k_pose_opt(PoseDev D)
{
    __shared__ double red[5 * 28];
    __shared__ double s_pose[7], s_x[6];
    __shared__ int s_ok;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int e0 = D.first[f], e1 = D.first[f + 1], n = e1 - e0;
    for (int e = e0 + tid; e < e1; e += PO_TPB) D.outlier[e] = 0;
    if (n < 3) { if (tid == 0) D.n_inliers[f] = 0; return; }            // Optimizer.cpp:296-297
    double K[4], pose0[7];
    for (int i = 0; i < 4; i++) K[i] = D.intr[4 * (long long)f + i];
    for (int i = 0; i < 7; i++) pose0[i] = D.poses[7 * (long long)f + i];
    const double delta = sqrt(5.991);
    double pose[7], Rt[12];
    int nbad_edges = 0;

    auto set_pose = [&](const double* p) {
        for (int i = 0; i < 7; i++) pose[i] = p[i];
        double R[9];
        ba_quat_to_R(pose, R);
        for (int i = 0; i < 9; i++) Rt[i] = R[i];
        Rt[9] = pose[4]; Rt[10] = pose[5]; Rt[11] = pose[6];
    };
    __syncthreads();
    for (int round = 0; round < 4; round++) {
        const bool robust = round <= 2;
        set_pose(pose0);                                                   // :309 restart from Frame.mTcw
        double lambda = 0, ni = 2;
        int nBad = 0;
        for (int it = 0; it < 10; it++) {
            // computeActiveErrors + buildSystem over the active (level 0) edges
            double acc[28];
            for (int i = 0; i < 28; i++) acc[i] = 0;
            int nact = 0;
            for (int e = e0 + tid; e < e1; e += PO_TPB) {
                if (D.outlier[e]) continue;
                nact++;
                double er[2], A[6], B[12];
                ba_edge_eval(Rt, K, D.pts + 3 * (long long)e, D.obs + 2 * (long long)e, er, A, B, nullptr);
                D.err[2 * (long long)e] = er[0]; D.err[2 * (long long)e + 1] = er[1];
                const double om = D.info[e];
                const double c2 = om * (er[0] * er[0] + er[1] * er[1]);
                double r0 = c2, r1 = 1.;
                if (robust) ba_huber(c2, delta, &r0, &r1);
                acc[27] += r0;
                const double w = r1 * om, g0 = -om * er[0] * r1, g1 = -om * er[1] * r1;
                int m = 0;
                for (int i = 0; i < 6; i++) {
                    acc[21 + i] += B[i] * g0 + B[6 + i] * g1;
                    for (int j = i; j < 6; j++) acc[m++] += w * (B[i] * B[j] + B[6 + i] * B[6 + j]);
                }
            }
            block_sum<28>(acc, red);
            const int any_active = __syncthreads_or(nact);
            if (!any_active) break;
            double H[36], b[6];
            {
                int m = 0;
                for (int i = 0; i < 6; i++) { b[i] = acc[21 + i]; for (int j = i; j < 6; j++) { H[i * 6 + j] = acc[m]; H[j * 6 + i] = acc[m]; m++; } }
            }
            double cur = acc[27];
            const double ini = cur;
            if (it == 0) {                                                 // computeLambdaInit
                double md = 0;
                for (int j = 0; j < 6; j++) md = fmax(md, fabs(H[7 * j]));
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                double save[7];
                for (int i = 0; i < 7; i++) save[i] = pose[i];
                // one lane solves (H + lambda I) x = b and applies the exp map; everybody reads the result from LDS
                if (tid == 0) {
                    double Hl[36], x[6] = { 0, 0, 0, 0, 0, 0 };
                    for (int i = 0; i < 36; i++) Hl[i] = H[i];
                    for (int j = 0; j < 6; j++) Hl[7 * j] += lambda;
                    const bool ok = chol6(Hl, b, x);
                    s_ok = ok ? 1 : 0;
                    for (int i = 0; i < 6; i++) s_x[i] = x[i];
                    if (ok) {
                        double o[7];
                        ba_se3_exp_mul(x, pose, o);
                        for (int i = 0; i < 7; i++) s_pose[i] = o[i];
                    }
                }
                __syncthreads();
                const int ok2 = s_ok;
                double x[6];
                for (int i = 0; i < 6; i++) x[i] = s_x[i];
                double temp = DBL_MAX;
                if (ok2) {
                    double np[7];
                    for (int i = 0; i < 7; i++) np[i] = s_pose[i];
                    set_pose(np);
                    double t[1] = { 0 };
                    for (int e = e0 + tid; e < e1; e += PO_TPB) {
                        if (D.outlier[e]) continue;
                        double er[2];
                        ba_edge_eval(Rt, K, D.pts + 3 * (long long)e, D.obs + 2 * (long long)e, er, nullptr, nullptr, nullptr);
                        D.err[2 * (long long)e] = er[0]; D.err[2 * (long long)e + 1] = er[1];
                        const double c2 = D.info[e] * (er[0] * er[0] + er[1] * er[1]);
                        double r0 = c2, r1;
                        if (robust) ba_huber(c2, delta, &r0, &r1);
                        t[0] += r0;
                    }
                    block_sum<1>(t, red);
                    temp = t[0];
                }
                __syncthreads();                                           // s_ok / s_pose consumed before the next trial rewrites them
                double scale = 1e-3;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                rho = ok2 ? (cur - temp) / scale : -1.0;
                if (rho > 0 && isfinite(temp)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha); ni = 2; cur = temp;
                } else {
                    lambda *= ni; ni *= 2;
                    set_pose(save);
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;
            if ((ini - cur) * 1e3 < ini) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        // classification (:313-338): level-1 edges get a fresh error, the others keep the last computed one
        int bad = 0;
        __syncthreads();
        for (int e = e0 + tid; e < e1; e += PO_TPB) {
            double er0, er1;
            if (D.outlier[e]) {
                double er[2];
                ba_edge_eval(Rt, K, D.pts + 3 * (long long)e, D.obs + 2 * (long long)e, er, nullptr, nullptr, nullptr);
                D.err[2 * (long long)e] = er[0]; D.err[2 * (long long)e + 1] = er[1];
                er0 = er[0]; er1 = er[1];
            } else { er0 = D.err[2 * (long long)e]; er1 = D.err[2 * (long long)e + 1]; }
            const float chi2 = (float)(D.info[e] * (er0 * er0 + er1 * er1));
            const bool out = chi2 > 5.991f;
            D.outlier[e] = out ? 1 : 0;
            bad += out;
        }
        {
            double t[1] = { (double)bad };
            block_sum<1>(t, red);
            nbad_edges = (int)t[0];
        }
        __syncthreads();
        if (n < 10) break;                                                 // :340-341
    }
    if (tid == 0) {
        for (int i = 0; i < 7; i++) D.poses[7 * (long long)f + i] = pose[i];
        D.n_inliers[f] = n - nbad_edges;
    }
}

void pose_launch(hipStream_t s, const PoseDev& D) { hipLaunchKernelGGL(k_pose_opt, dim3(D.n_frames), dim3(PO_TPB), 0, s, D); }
