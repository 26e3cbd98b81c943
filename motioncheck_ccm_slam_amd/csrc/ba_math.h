// ba_math.h -- float64 geometry of the reprojection BA, usable from host and device code.
// Restates (not copies) the arithmetic of g2o's SE3Quat / EdgeSE3ProjectXYZ in the reference tree:
//   cslam/thirdparty/g2o/g2o/types/se3quat.h:104-110,217-257,280-285
//   cslam/thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:90-101, cpp:103-147}
//   cslam/thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-91
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define BA_HD __host__ __device__ inline
#else
#define BA_HD inline
#endif

// unit quaternion (x,y,z,w) -> row-major 3x3
BA_HD void ba_quat_to_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen's Quaterniond(Matrix3d) branch structure
BA_HD void ba_R_to_quat(const double* m, double* q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

BA_HD void ba_quat_normalize(double* q)      // w >= 0, unit norm
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// pose_out = exp(delta) * pose_in; delta = (omega, upsilon); pose = (qx,qy,qz,qw,tx,ty,tz)
BA_HD void ba_se3_exp_mul(const double* d, const double* Tin, double* Tout)
{
    const double o0 = d[0], o1 = d[1], o2 = d[2];
    const double theta = sqrt(o0 * o0 + o1 * o1 + o2 * o2);
    const double O[9] = { 0, -o2, o1, o2, 0, -o0, -o1, o0, 0 };
    double O2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double qe[4], te[3], Re[9];
    ba_R_to_quat(R, qe);
    for (int i = 0; i < 3; i++) te[i] = V[i * 3] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5];
    ba_quat_to_R(qe, Re);
    double q[4];
    const double* b4 = Tin;
    q[3] = qe[3] * b4[3] - qe[0] * b4[0] - qe[1] * b4[1] - qe[2] * b4[2];
    q[0] = qe[3] * b4[0] + qe[0] * b4[3] + qe[1] * b4[2] - qe[2] * b4[1];
    q[1] = qe[3] * b4[1] + qe[1] * b4[3] + qe[2] * b4[0] - qe[0] * b4[2];
    q[2] = qe[3] * b4[2] + qe[2] * b4[3] + qe[0] * b4[1] - qe[1] * b4[0];
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = te[i] + Re[i * 3] * Tin[4] + Re[i * 3 + 1] * Tin[5] + Re[i * 3 + 2] * Tin[6];
    ba_quat_normalize(q);
    for (int i = 0; i < 4; i++) Tout[i] = q[i];
    for (int i = 0; i < 3; i++) Tout[4 + i] = t[i];
}

// Residual and (optionally) Jacobians of one observation.  Rt = R (9, row-major) then t (3).
// A = d e / d point (2x3), B = d e / d pose (2x6, columns omega then upsilon).
BA_HD void ba_edge_eval(const double* Rt, const double* K, const double* p, const double* obs,
                        double* e, double* A, double* B, double* zout)
{
    const double x = Rt[0] * p[0] + Rt[1] * p[1] + Rt[2] * p[2] + Rt[9];
    const double y = Rt[3] * p[0] + Rt[4] * p[1] + Rt[5] * p[2] + Rt[10];
    const double z = Rt[6] * p[0] + Rt[7] * p[1] + Rt[8] * p[2] + Rt[11];
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    e[0] = obs[0] - (x / z * fx + cx);
    e[1] = obs[1] - (y / z * fy + cy);
    if (zout) *zout = z;
    if (!A) return;
    const double z2 = z * z;
    const double t0 = fx, t2 = -x / z * fx, t4 = fy, t5 = -y / z * fy;
    const double iz = -1. / z;
    for (int j = 0; j < 3; j++) {
        A[j] = iz * (t0 * Rt[j] + t2 * Rt[6 + j]);
        A[3 + j] = iz * (t4 * Rt[3 + j] + t5 * Rt[6 + j]);
    }
    B[0] = x * y / z2 * fx;        B[1] = -(1 + (x * x / z2)) * fx;  B[2] = y / z * fx;
    B[3] = -1. / z * fx;           B[4] = 0;                          B[5] = x / z2 * fx;
    B[6] = (1 + y * y / z2) * fy;  B[7] = -x * y / z2 * fy;           B[8] = -x / z * fy;
    B[9] = 0;                      B[10] = -1. / z * fy;              B[11] = y / z2 * fy;
}

BA_HD void ba_huber(double e, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
    else { const double s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

// inverse of a symmetric-or-not 3x3 by cofactors
BA_HD void ba_inv3(const double* m, double* o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// ---- per-landmark pieces of the Schur step, shared by ba_sparse.hip (k_sp_dinv, k_sp_edge_y) and the fused tail of k_ba_lin_landmark:
// the same expressions in the same order, so both paths produce the same bits.
// (ba_inv3 above = Eigen's 3 x 3 inverse by cofactors, what BlockSolver::solve uses for Hll, block_solver.hpp:396)
// lower Cholesky factor of H + lambda I (upper half of H read): f = { 1/l00, l10, l20, 1/l11, l21, 1/l22 }.
// Pivots: in exact arithmetic each is >= lambda > 0 (H is a sum of J^T J terms).  For a landmark block that is rank-deficient (one
// observation, or no parallax) and a late, tiny lambda the computed second or third pivot can round to <= 0, and its square root would
// poison every reduced block the landmark touches.  A pivot below the rounding level of its own diagonal entry is noise either way, so
// it is held at that level (2^-52 of the entry): the step stays finite like the reference's, whose 3 x 3 inverse takes no square roots.
BA_HD void ba_chol3(const double* H, double lambda, double* f)
{
    const double d1 = H[4] + lambda, d2 = H[8] + lambda;
    const double l00 = sqrt(H[0] + lambda), i00 = 1.0 / l00;
    const double l10 = H[1] * i00, l20 = H[2] * i00;
    const double l11 = sqrt(fmax(d1 - l10 * l10, 0x1p-52 * d1)), i11 = 1.0 / l11;
    const double l21 = (H[5] - l20 * l10) * i11;
    const double i22 = 1.0 / sqrt(fmax(d2 - l20 * l20 - l21 * l21, 0x1p-52 * d2));
    f[0] = i00; f[1] = l10; f[2] = l20; f[3] = i11; f[4] = l21; f[5] = i22;
}
// Z = B L^-T (6 x 3, row z of Z solves z L^T = row of B) and c = B d (6): the per-edge operand of the Schur product and the edge's
// share of Hpl Dinv b_l
BA_HD void ba_edge_z_c(const double* B, const double* f, const double* d, double* z, double* c)
{
    for (int i = 0; i < 6; i++) {
        const double z0 = B[i * 3] * f[0];
        const double z1 = (B[i * 3 + 1] - z0 * f[1]) * f[3];
        const double z2 = (B[i * 3 + 2] - z0 * f[2] - z1 * f[4]) * f[5];
        z[i * 3] = z0; z[i * 3 + 1] = z1; z[i * 3 + 2] = z2;
        c[i] = B[i * 3] * d[0] + B[i * 3 + 1] * d[1] + B[i * 3 + 2] * d[2];
    }
}
