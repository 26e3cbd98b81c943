// sim3_math.h -- g2o::Sim3 (cslam/thirdparty/g2o/g2o/types/sim3.h) on the device: exp (:62-131), product (:277-283),
// inverse (:245-248), log (:137-213), and VertexSim3Expmap::oplusImpl (types_seven_dof_expmap.h:68-77).
// sim3 = qx,qy,qz,qw, tx,ty,tz, s.
#pragma once
#include "ba_math.h"

__device__ __forceinline__ void s3_quat_mul(const double* a, const double* b, double* o)
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void s3_rotv(const double* q, const double* v, double* o)
{
    double R[9]; ba_quat_to_R(q, R);
    for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
// Sim3(const Vector7d&), sim3.h:62-131
__device__ void s3_exp(const double* u, double* S)
{
    const double* omega = u; const double* upsilon = u + 3;
    const double sigma = u[6];
    const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    const double Om[9] = { 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0 };
    double Om2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Om2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    const double s = exp(sigma);
    const double eps = 0.00001;
    double A, B, C, R[9];
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) { A = 1. / 2.; B = 1. / 6.; for (int i = 0; i < 9; i++) R[i] = ((i & 3) == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; }
        else {
            const double theta2 = theta * theta;
            A = (1 - cos(theta)) / theta2;
            B = (theta - sin(theta)) / (theta2 * theta);
            for (int i = 0; i < 9; i++) R[i] = ((i & 3) == 0 ? 1.0 : 0.0) + sin(theta) / theta * Om[i] + (1 - cos(theta)) / (theta * theta) * Om2[i];
        }
    } else {
        C = (s - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * s + 1) / sigma2;
            B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
            for (int i = 0; i < 9; i++) R[i] = ((i & 3) == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
        } else {
            for (int i = 0; i < 9; i++) R[i] = ((i & 3) == 0 ? 1.0 : 0.0) + sin(theta) / theta * Om[i] + (1 - cos(theta)) / (theta * theta) * Om2[i];
            const double a = s * sin(theta), b = s * cos(theta);
            const double theta2 = theta * theta, sigma2 = sigma * sigma;
            const double c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    ba_R_to_quat(R, S);
    for (int i = 0; i < 3; i++) {
        S[4 + i] = 0;
        for (int j = 0; j < 3; j++) S[4 + i] += (A * Om[3 * i + j] + B * Om2[3 * i + j] + C * (i == j ? 1.0 : 0.0)) * upsilon[j];
    }
    S[7] = s;
}
__device__ void s3_mul(const double* a, const double* b, double* o)      // sim3.h:277-283
{
    double q[4], t[3];
    s3_quat_mul(a, b, q);
    s3_rotv(a, b + 4, t);
    for (int i = 0; i < 4; i++) o[i] = q[i];
    for (int i = 0; i < 3; i++) o[4 + i] = a[7] * t[i] + a[4 + i];
    o[7] = a[7] * b[7];
}
__device__ void s3_inverse(const double* a, double* o)                    // sim3.h:245-248
{
    const double qc[4] = { -a[0], -a[1], -a[2], a[3] };
    const double ts[3] = { (-1. / a[7]) * a[4], (-1. / a[7]) * a[5], (-1. / a[7]) * a[6] };
    double t[3]; s3_rotv(qc, ts, t);
    for (int i = 0; i < 4; i++) o[i] = qc[i];
    for (int i = 0; i < 3; i++) o[4 + i] = t[i];
    o[7] = 1. / a[7];
}
__device__ void s3_oplus(const double* S, const double* upd, int fix_scale, double* o)   // VertexSim3Expmap::oplusImpl
{
    double u[7];
    for (int i = 0; i < 7; i++) u[i] = upd[i];
    if (fix_scale) u[6] = 0;
    double E[8]; s3_exp(u, E);
    s3_mul(E, S, o);
}
// Sim3::log, sim3.h:137-213 (W.lu().solve(t) by Gaussian elimination with partial pivoting)
__device__ void s3_log(const double* S, double* res)
{
    const double s = S[7];
    const double sigma = log(s);
    double R[9]; ba_quat_to_R(S, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };
    double omega[3], A, B, C;
    const double eps = 0.00001;
    if (fabs(sigma) < eps) {
        C = 1;
        if (d > 1 - eps) { for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i]; A = 1. / 2.; B = 1. / 6.; }
        else {
            const double theta = acos(d), theta2 = theta * theta;
            for (int i = 0; i < 3; i++) omega[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
            A = (1 - cos(theta)) / theta2;
            B = (theta - sin(theta)) / (theta2 * theta);
        }
    } else {
        C = (s - 1) / sigma;
        if (d > 1 - eps) {
            const double sigma2 = sigma * sigma;
            for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
            A = ((sigma - 1) * s + 1) / sigma2;
            B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
        } else {
            const double theta = acos(d);
            for (int i = 0; i < 3; i++) omega[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
            const double theta2 = theta * theta;
            const double a = s * sin(theta), b = s * cos(theta);
            const double c = theta2 + sigma * sigma;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    const double Om[9] = { 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0 };
    double M[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            const double o2 = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
            M[4 * i + j] = A * Om[3 * i + j] + B * o2 + C * (i == j ? 1.0 : 0.0);
        }
        M[4 * i + 3] = S[4 + i];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int p = c;
        for (int r = c + 1; r < 3; r++) if (fabs(M[4 * r + c]) > fabs(M[4 * p + c])) p = r;
        if (p != c) for (int j = 0; j < 4; j++) { const double t = M[4 * c + j]; M[4 * c + j] = M[4 * p + j]; M[4 * p + j] = t; }
        for (int r = c + 1; r < 3; r++) {
            const double f = M[4 * r + c] / M[4 * c + c];
            for (int j = c; j < 4; j++) M[4 * r + j] -= f * M[4 * c + j];
        }
    }
    double ups[3];
    for (int r = 2; r >= 0; r--) {
        double v = M[4 * r + 3];
        for (int j = r + 1; j < 3; j++) v -= M[4 * r + j] * ups[j];
        ups[r] = v / M[4 * r + r];
    }
    for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[3 + i] = ups[i]; }
    res[6] = sigma;
}
