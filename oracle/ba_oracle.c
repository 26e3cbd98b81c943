#define _GNU_SOURCE
/* ba_oracle.c -- CPU restatement (float64) of the 6-DoF pose / 3-DoF point
 * reprojection bundle adjustment that Optimizer.cpp hands to g2o.
 * TEST INFRASTRUCTURE (see oracle.h).
 *
 * Follows, in the reference tree:
 *   cslam/thirdparty/g2o/g2o/types/types_six_dof_expmap.{h,cpp}   (edge error + Jacobians)
 *   cslam/thirdparty/g2o/g2o/types/se3quat.h, se3_ops.hpp          (exp map, composition)
 *   cslam/thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-120      (quadratic form)
 *   cslam/thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-91     (Huber)
 *   cslam/thirdparty/g2o/g2o/core/block_solver.hpp:354-486,564-604 (Schur, lambda)
 *   cslam/thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189
 *   cslam/thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-435
 *   src/Optimizer.cpp:536-602 (local BA's two-stage schedule), src/Converter.cc:40-119
 * Eigen (Quaterniond(R), 3x3 inverse, SimplicialLDLT) is not in the tree: its
 * arithmetic is restated; the reduced system is solved by dense Cholesky on small
 * graphs and by the block-sparse Cholesky of bchol_oracle.c (minimum-degree ordering
 * once, exact factorisation per trial: linear_solver_eigen.h:106-136,165-222) above
 * ORC_SPARSE_MIN_FREE free keyframes, any exact SPD solve being equivalent up to
 * rounding.  Edge
 * summation order here is edge-array order; g2o's depends on pointer-ordered
 * maps, so bitwise equality with g2o is impossible by construction and the
 * contract is a tolerance (1e-5 on pose updates).  PARITY UNPINNED vs g2o.
 */
#include "oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- rotations */
static void quat_to_R(const double* q, double* R)   /* q = x,y,z,w ; Eigen toRotationMatrix */
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
static void R_to_quat(const double* m, double* q)   /* Eigen Quaterniond(Matrix3d) */
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}
static void quat_normalize(double* q)                /* SE3Quat::normalizeRotation, se3quat.h:280-285 */
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_mul(const double* a, const double* b, double* o)
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void mat3_mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

/* T_out = exp(delta) * T_in : VertexSE3Expmap::oplusImpl (types_six_dof_expmap.h:73-76),
 * SE3Quat::exp (se3quat.h:223-257), operator* (:104-110). delta = (omega, upsilon). */
void orc_se3_exp_mul(const double* d, const double* Tin, double* Tout)
{
    const double om[3] = { d[0], d[1], d[2] }, up[3] = { d[3], d[4], d[5] };
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double O2[9], R[9], V[9];
    mat3_mul(O, O, O2);
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double qe[4], te[3];
    R_to_quat(R, qe);
    for (int i = 0; i < 3; i++) te[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    /* result = exp * Tin */
    double Re[9], q[4];
    quat_to_R(qe, Re);
    quat_mul(qe, Tin, q);
    double t[3];
    for (int i = 0; i < 3; i++)
        t[i] = te[i] + Re[i * 3] * Tin[4] + Re[i * 3 + 1] * Tin[5] + Re[i * 3 + 2] * Tin[6];
    quat_normalize(q);
    for (int i = 0; i < 4; i++) Tout[i] = q[i];
    for (int i = 0; i < 3; i++) Tout[4 + i] = t[i];
}

/* Converter::toSE3Quat (src/Converter.cc:40-56): float 4x4 -> double R,t -> SE3Quat(R,t) */
void orc_pose_from_mat4f(const float* T, double* pose)
{
    double R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = (double)T[i * 4 + j];
    R_to_quat(R, pose);
    quat_normalize(pose);                                  /* SE3Quat ctor normalises */
    for (int i = 0; i < 3; i++) pose[4 + i] = (double)T[i * 4 + 3];
}
/* Converter::toCvMat(SE3Quat) (src/Converter.cc:86-93) */
void orc_pose_to_mat4f(const double* pose, float* T)
{
    double R[9];
    quat_to_R(pose, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)pose[4 + i];
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

/* ---------------------------------------------------------------- one edge
 * computeError (types_six_dof_expmap.h:90-101) + linearizeOplus (.cpp:103-139) */
static void edge_eval(const double* R, const double* t, const double* K, const double* p,
                      const double* obs, double* e, double* A, double* B, double* zout)
{
    const double x = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    const double y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    const double z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    e[0] = obs[0] - (x / z * fx + cx);
    e[1] = obs[1] - (y / z * fy + cy);
    if (zout) *zout = z;
    if (!A) return;
    const double z2 = z * z;
    const double tm[6] = { fx, 0, -x / z * fx, 0, fy, -y / z * fy };
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
        A[i * 3 + j] = -1. / z * (tm[i * 3] * R[j] + tm[i * 3 + 1] * R[3 + j] + tm[i * 3 + 2] * R[6 + j]);
    B[0] = x * y / z2 * fx;  B[1] = -(1 + (x * x / z2)) * fx;  B[2] = y / z * fx;
    B[3] = -1. / z * fx;     B[4] = 0;                          B[5] = x / z2 * fx;
    B[6] = (1 + y * y / z2) * fy;  B[7] = -x * y / z2 * fy;     B[8] = -x / z * fy;
    B[9] = 0;                B[10] = -1. / z * fy;              B[11] = y / z2 * fy;
}

void orc_ba_edge(const double* pose7, const double* intr4, const double* pt3, const double* obs2,
                 double* err2, double* A, double* B)
{
    double R[9];
    quat_to_R(pose7, R);
    edge_eval(R, pose7 + 4, intr4, pt3, obs2, err2, A, B, 0);
}

/* RobustKernelHuber::robustify, robust_kernel_impl.cpp:78-91 */
static void huber(double e, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
    else { const double s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

/* ---------------------------------------------------------------- solver state */
typedef struct {
    const orc_ba_problem* pb;
    int P, L, E;
    int nfree; int* free_of;          /* pose -> free index or -1 */
    double* R;                         /* [P][9] */
    uint8_t* active;                   /* [E] level 0 */
    double* err;                       /* [E][2] last computed (stale for inactive edges, like g2o) */
    double huber_delta;                /* <= 0: none */
    /* system */
    double* Hpp; double* bp;           /* [nfree][36], [nfree][6] */
    double* Hll; double* bl;           /* [L][9], [L][3] */
    double* Hpl;                       /* [E][18] (6x3), zero for fixed / inactive */
    double* Hs; double* bs;            /* dense 6nfree x 6nfree, 6nfree */
    int sparse;                        /* reduced system held block-sparse + bchol_oracle.c instead of dense Hs */
    int32_t* sidx; double* sblk; int nsblk;   /* [nfree*nfree] (f1 <= f2) -> slot, [nsblk][36] upper blocks of Hschur */
    orc_bchol* chol;
    double* x;                         /* [6nfree + 3L] */
    double* Dinv; double* coeff;
    int* pt_first; int* pt_edges;      /* CSR landmark -> edges (sorted by pose) */
} ba_t;

static void refresh_R(ba_t* s)
{
    for (int i = 0; i < s->P; i++) quat_to_R(s->pb->poses + 7 * i, s->R + 9 * i);
}

/* computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:61-114) */
static double compute_errors(ba_t* s)
{
    const orc_ba_problem* pb = s->pb;
    refresh_R(s);
    double chi = 0;
    for (int e = 0; e < s->E; e++) {
        if (!s->active[e]) continue;
        const int pi = pb->edge_pose[e], li = pb->edge_point[e];
        edge_eval(s->R + 9 * pi, pb->poses + 7 * pi + 4, pb->intr + 4 * pi, pb->points + 3 * li,
                  pb->obs + 2 * e, s->err + 2 * e, 0, 0, 0);
        const double c2 = pb->info[e] * (s->err[2 * e] * s->err[2 * e] + s->err[2 * e + 1] * s->err[2 * e + 1]);
        if (s->huber_delta > 0) { double r0, r1; huber(c2, s->huber_delta, &r0, &r1); chi += r0; }
        else chi += c2;
    }
    return chi;
}

/* BlockSolver::buildSystem (block_solver.hpp:502-560) */
static void build_system(ba_t* s)
{
    const orc_ba_problem* pb = s->pb;
    memset(s->Hpp, 0, sizeof(double) * 36 * (size_t)s->nfree);
    memset(s->bp, 0, sizeof(double) * 6 * (size_t)s->nfree);
    memset(s->Hll, 0, sizeof(double) * 9 * (size_t)s->L);
    memset(s->bl, 0, sizeof(double) * 3 * (size_t)s->L);
    memset(s->Hpl, 0, sizeof(double) * 18 * (size_t)s->E);
    for (int e = 0; e < s->E; e++) {
        if (!s->active[e]) continue;
        const int pi = pb->edge_pose[e], li = pb->edge_point[e];
        double er[2], A[6], B[12];
        edge_eval(s->R + 9 * pi, pb->poses + 7 * pi + 4, pb->intr + 4 * pi, pb->points + 3 * li,
                  pb->obs + 2 * e, er, A, B, 0);
        const double om = pb->info[e];
        double r1 = 1.;
        if (s->huber_delta > 0) { double r0; huber(om * (er[0] * er[0] + er[1] * er[1]), s->huber_delta, &r0, &r1); }
        const double w = r1 * om;
        const double g0 = -om * er[0] * r1, g1 = -om * er[1] * r1;   /* omega_r * rho[1] */
        double* Hl = s->Hll + 9 * (size_t)li; double* b_l = s->bl + 3 * (size_t)li;
        for (int i = 0; i < 3; i++) {
            b_l[i] += A[i] * g0 + A[3 + i] * g1;
            for (int j = 0; j < 3; j++) Hl[i * 3 + j] += w * (A[i] * A[j] + A[3 + i] * A[3 + j]);
        }
        const int f = s->free_of[pi];
        if (f >= 0) {
            double* Hp = s->Hpp + 36 * (size_t)f; double* b_p = s->bp + 6 * (size_t)f;
            double* Hx = s->Hpl + 18 * (size_t)e;
            for (int i = 0; i < 6; i++) {
                b_p[i] += B[i] * g0 + B[6 + i] * g1;
                for (int j = 0; j < 6; j++) Hp[i * 6 + j] += w * (B[i] * B[j] + B[6 + i] * B[6 + j]);
                for (int j = 0; j < 3; j++) Hx[i * 3 + j] = w * (B[i] * A[j] + B[6 + i] * A[3 + j]);
            }
        }
    }
}

static int inv3(const double* m, double* o)      /* cofactor inverse, as Eigen does for 3x3 */
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return isfinite(id);
}

/* dense Cholesky A = L L^T in the lower triangle, row-major n x n; returns 0 if not SPD */
static int chol_factor(double* A, int n)
{
    for (int j = 0; j < n; j++) {
        double* Aj = A + (size_t)j * n;
        double d = Aj[j];
        for (int k = 0; k < j; k++) d -= Aj[k] * Aj[k];
        if (!(d > 0)) return 0;
        d = sqrt(d);
        Aj[j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < n; i++) {
            double* Ai = A + (size_t)i * n;
            double v = Ai[j];
            for (int k = 0; k < j; k++) v -= Ai[k] * Aj[k];
            Ai[j] = v * id;
        }
    }
    return 1;
}
static void chol_solve(const double* A, int n, double* b)
{
    for (int i = 0; i < n; i++) {
        double v = b[i];
        const double* Ai = A + (size_t)i * n;
        for (int k = 0; k < i; k++) v -= Ai[k] * b[k];
        b[i] = v / Ai[i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double v = b[i];
        for (int k = i + 1; k < n; k++) v -= A[(size_t)k * n + i] * b[k];
        b[i] = v / A[(size_t)i * n + i];
    }
}

/* Schur reduction of BlockSolver::solve (block_solver.hpp:370-439) with lambda added to every
 * diagonal element (setLambda :564-589).  Fills Hs (full symmetric) and bs. */
static void schur_reduce(ba_t* s, double lambda)
{
    const int n = 6 * s->nfree;
    memset(s->Hs, 0, sizeof(double) * (size_t)n * n);
    memset(s->coeff, 0, sizeof(double) * n);
    for (int f = 0; f < s->nfree; f++)
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++)
            s->Hs[(size_t)(6 * f + i) * n + 6 * f + j] = s->Hpp[36 * (size_t)f + i * 6 + j] + (i == j ? lambda : 0.0);
    for (int l = 0; l < s->L; l++) {
        double D[9], db[3];
        memcpy(D, s->Hll + 9 * (size_t)l, sizeof D);
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = s->Dinv + 9 * (size_t)l;
        inv3(D, Di);
        const double* b_l = s->bl + 3 * (size_t)l;
        for (int i = 0; i < 3; i++) db[i] = Di[i * 3] * b_l[0] + Di[i * 3 + 1] * b_l[1] + Di[i * 3 + 2] * b_l[2];
        for (int a = s->pt_first[l]; a < s->pt_first[l + 1]; a++) {
            const int e1 = s->pt_edges[a];
            const int f1 = s->free_of[s->pb->edge_pose[e1]];
            if (f1 < 0 || !s->active[e1]) continue;
            const double* Bi = s->Hpl + 18 * (size_t)e1;
            double BD[18];
            for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++)
                BD[i * 3 + j] = Bi[i * 3] * Di[j] + Bi[i * 3 + 1] * Di[3 + j] + Bi[i * 3 + 2] * Di[6 + j];
            for (int i = 0; i < 6; i++)
                s->coeff[6 * f1 + i] += Bi[i * 3] * db[0] + Bi[i * 3 + 1] * db[1] + Bi[i * 3 + 2] * db[2];
            for (int b = a; b < s->pt_first[l + 1]; b++) {
                const int e2 = s->pt_edges[b];
                const int f2 = s->free_of[s->pb->edge_pose[e2]];
                if (f2 < 0 || !s->active[e2]) continue;
                const double* Bj = s->Hpl + 18 * (size_t)e2;
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
                    const double v = BD[i * 3] * Bj[j * 3] + BD[i * 3 + 1] * Bj[j * 3 + 1] + BD[i * 3 + 2] * Bj[j * 3 + 2];
                    s->Hs[(size_t)(6 * f1 + i) * n + 6 * f2 + j] -= v;
                    if (f1 != f2) s->Hs[(size_t)(6 * f2 + j) * n + 6 * f1 + i] -= v;
                }
            }
        }
    }
    for (int i = 0; i < n; i++) s->bs[i] = s->bp[i] - s->coeff[i];
}

/* The same reduction into the block-sparse upper triangle (f1 <= f2) of Hschur: identical products in the identical
 * order, only the destination differs. */
static void schur_reduce_sparse(ba_t* s, double lambda)
{
    const int n = 6 * s->nfree, nf = s->nfree;
    memset(s->sblk, 0, sizeof(double) * 36 * (size_t)s->nsblk);
    memset(s->coeff, 0, sizeof(double) * n);
    for (int f = 0; f < nf; f++) {
        double* Hd = s->sblk + 36 * (size_t)s->sidx[(size_t)f * nf + f];
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++)
            Hd[i * 6 + j] = s->Hpp[36 * (size_t)f + i * 6 + j] + (i == j ? lambda : 0.0);
    }
    for (int l = 0; l < s->L; l++) {
        double D[9], db[3];
        memcpy(D, s->Hll + 9 * (size_t)l, sizeof D);
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = s->Dinv + 9 * (size_t)l;
        inv3(D, Di);
        const double* b_l = s->bl + 3 * (size_t)l;
        for (int i = 0; i < 3; i++) db[i] = Di[i * 3] * b_l[0] + Di[i * 3 + 1] * b_l[1] + Di[i * 3 + 2] * b_l[2];
        for (int a = s->pt_first[l]; a < s->pt_first[l + 1]; a++) {
            const int e1 = s->pt_edges[a];
            const int f1 = s->free_of[s->pb->edge_pose[e1]];
            if (f1 < 0 || !s->active[e1]) continue;
            const double* Bi = s->Hpl + 18 * (size_t)e1;
            double BD[18];
            for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++)
                BD[i * 3 + j] = Bi[i * 3] * Di[j] + Bi[i * 3 + 1] * Di[3 + j] + Bi[i * 3 + 2] * Di[6 + j];
            for (int i = 0; i < 6; i++)
                s->coeff[6 * f1 + i] += Bi[i * 3] * db[0] + Bi[i * 3 + 1] * db[1] + Bi[i * 3 + 2] * db[2];
            for (int b = a; b < s->pt_first[l + 1]; b++) {
                const int e2 = s->pt_edges[b];
                const int f2 = s->free_of[s->pb->edge_pose[e2]];
                if (f2 < 0 || !s->active[e2]) continue;
                const double* Bj = s->Hpl + 18 * (size_t)e2;
                /* edges of a landmark are sorted by pose, so f1 <= f2; a pose repeated inside one landmark (f1 == f2,
                 * a != b; the synthetic graphs have none) adds the product and its transpose to the diagonal block */
                double* T = s->sblk + 36 * (size_t)s->sidx[(size_t)f1 * nf + f2];
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
                    const double v = BD[i * 3] * Bj[j * 3] + BD[i * 3 + 1] * Bj[j * 3 + 1] + BD[i * 3 + 2] * Bj[j * 3 + 2];
                    T[i * 6 + j] -= v;
                    if (f1 == f2 && a != b) T[j * 6 + i] -= v;
                }
            }
        }
    }
    for (int i = 0; i < n; i++) s->bs[i] = s->bp[i] - s->coeff[i];
}

/* solve with the current lambda; x = [poses | landmarks]; returns 0 if the reduced system is not SPD */
static int schur_solve(ba_t* s, double lambda)
{
    const int n = 6 * s->nfree;
    if (s->sparse) {
        schur_reduce_sparse(s, lambda);
        if (n > 0) {
            if (orc_bchol_factor(s->chol, s->sidx, s->sblk) <= 0) return 0;
            orc_bchol_solve(s->chol, s->bs, s->x);
        }
    } else {
    schur_reduce(s, lambda);
    if (n > 0) {
        if (!chol_factor(s->Hs, n)) return 0;
        memcpy(s->x, s->bs, sizeof(double) * n);
        chol_solve(s->Hs, n, s->x);
    }
    }
    /* landmarks: xl = Dinv (bl - Hpl^T xp)  (block_solver.hpp:461-481) */
    for (int l = 0; l < s->L; l++) {
        double c[3] = { s->bl[3 * l], s->bl[3 * l + 1], s->bl[3 * l + 2] };
        for (int a = s->pt_first[l]; a < s->pt_first[l + 1]; a++) {
            const int e = s->pt_edges[a];
            const int f = s->free_of[s->pb->edge_pose[e]];
            if (f < 0 || !s->active[e]) continue;
            const double* Bi = s->Hpl + 18 * (size_t)e;
            const double* xp = s->x + 6 * f;
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 6; i++) c[j] -= Bi[i * 3 + j] * xp[i];
        }
        const double* Di = s->Dinv + 9 * (size_t)l;
        double* xl = s->x + n + 3 * l;
        for (int i = 0; i < 3; i++) xl[i] = Di[i * 3] * c[0] + Di[i * 3 + 1] * c[1] + Di[i * 3 + 2] * c[2];
    }
    return 1;
}

static int cmp_edge_pose(const void* a, const void* b, void* ctx)
{
    const int32_t* ep = (const int32_t*)ctx;
    const int x = *(const int*)a, y = *(const int*)b;
    if (ep[x] != ep[y]) return ep[x] < ep[y] ? -1 : 1;
    return x < y ? -1 : (x > y);
}

/* 0 = automatic (block-sparse above ORC_SPARSE_MIN_FREE free keyframes), 1 = dense, 2 = block-sparse.  Test hook. */
#define ORC_SPARSE_MIN_FREE 400
static int g_solver_mode = 0;
void orc_ba_set_solver(int mode) { g_solver_mode = mode; }

static ba_t* ba_new_mode(orc_ba_problem* pb, int mode);
static ba_t* ba_new(orc_ba_problem* pb) { return ba_new_mode(pb, g_solver_mode); }

static ba_t* ba_new_mode(orc_ba_problem* pb, int mode)
{
    ba_t* s = (ba_t*)calloc(1, sizeof(ba_t));
    s->pb = pb; s->P = pb->n_poses; s->L = pb->n_points; s->E = pb->n_edges;
    s->free_of = (int*)malloc(sizeof(int) * (s->P + 1));
    s->nfree = 0;
    for (int i = 0; i < s->P; i++) s->free_of[i] = (pb->fixed && pb->fixed[i]) ? -1 : s->nfree++;
    const int n = 6 * s->nfree;
    s->R = (double*)malloc(sizeof(double) * 9 * (size_t)(s->P + 1));
    s->active = (uint8_t*)malloc(s->E + 1); memset(s->active, 1, s->E + 1);
    s->err = (double*)calloc(2 * (size_t)s->E + 2, sizeof(double));
    s->Hpp = (double*)malloc(sizeof(double) * 36 * (size_t)(s->nfree + 1));
    s->bp = (double*)malloc(sizeof(double) * 6 * (size_t)(s->nfree + 1));
    s->Hll = (double*)malloc(sizeof(double) * 9 * (size_t)(s->L + 1));
    s->bl = (double*)malloc(sizeof(double) * 3 * (size_t)(s->L + 1));
    s->Hpl = (double*)malloc(sizeof(double) * 18 * (size_t)(s->E + 1));
    s->sparse = mode == 2 || (mode == 0 && s->nfree > ORC_SPARSE_MIN_FREE);
    s->Hs = (double*)malloc(sizeof(double) * (s->sparse ? 1 : (size_t)n * n + 1));
    s->bs = (double*)malloc(sizeof(double) * (n + 1));
    s->x = (double*)calloc((size_t)n + 3 * (size_t)s->L + 1, sizeof(double));
    s->Dinv = (double*)malloc(sizeof(double) * 9 * (size_t)(s->L + 1));
    s->coeff = (double*)malloc(sizeof(double) * (n + 1));
    s->pt_first = (int*)calloc(s->L + 2, sizeof(int));
    s->pt_edges = (int*)malloc(sizeof(int) * (s->E + 1));
    for (int e = 0; e < s->E; e++) s->pt_first[pb->edge_point[e] + 1]++;
    for (int l = 0; l < s->L; l++) s->pt_first[l + 1] += s->pt_first[l];
    int* fill = (int*)malloc(sizeof(int) * (s->L + 1));
    memcpy(fill, s->pt_first, sizeof(int) * (s->L + 1));
    for (int e = 0; e < s->E; e++) s->pt_edges[fill[pb->edge_point[e]]++] = e;
    for (int l = 0; l < s->L; l++)
        qsort_r(s->pt_edges + s->pt_first[l], s->pt_first[l + 1] - s->pt_first[l], sizeof(int),
                cmp_edge_pose, (void*)pb->edge_pose);
    free(fill);
    if (s->sparse && s->nfree > 0) {
        /* BlockSolver::buildStructure (block_solver.hpp:143-295): the pattern of Hschur is the diagonal plus one block per
         * pair of free keyframes sharing a landmark, whatever the edges' activity */
        const int nf = s->nfree;
        uint8_t* adj = (uint8_t*)calloc((size_t)nf * nf + 1, 1);
        for (int f = 0; f < nf; f++) adj[(size_t)f * nf + f] = 1;
        for (int l = 0; l < s->L; l++)
            for (int a = s->pt_first[l]; a < s->pt_first[l + 1]; a++) {
                const int f1 = s->free_of[pb->edge_pose[s->pt_edges[a]]];
                if (f1 < 0) continue;
                for (int b = a; b < s->pt_first[l + 1]; b++) {
                    const int f2 = s->free_of[pb->edge_pose[s->pt_edges[b]]];
                    if (f2 >= 0) adj[(size_t)f1 * nf + f2] = 1;
                }
            }
        s->sidx = (int32_t*)malloc(sizeof(int32_t) * ((size_t)nf * nf + 1));
        int m = 0;
        for (int f1 = 0; f1 < nf; f1++) for (int f2 = 0; f2 < nf; f2++)
            s->sidx[(size_t)f1 * nf + f2] = (f2 >= f1 && adj[(size_t)f1 * nf + f2]) ? m++ : -1;
        s->nsblk = m;
        s->sblk = (double*)malloc(sizeof(double) * 36 * (size_t)(m + 1));
        s->chol = orc_bchol_new(nf, adj);
        free(adj);
    }
    return s;
}
static void ba_free(ba_t* s)
{
    free(s->free_of); free(s->R); free(s->active); free(s->err); free(s->Hpp); free(s->bp);
    free(s->Hll); free(s->bl); free(s->Hpl); free(s->Hs); free(s->bs); free(s->x); free(s->Dinv);
    free(s->coeff); free(s->pt_first); free(s->pt_edges); free(s->sidx); free(s->sblk); orc_bchol_free(s->chol); free(s);
}

/* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg::solve per iteration */
static int lm_optimize(ba_t* s, int iterations, orc_ba_result* res, double* lambda_io)
{
    orc_ba_problem* pb = (orc_ba_problem*)s->pb;
    const int n = 6 * s->nfree, nx = n + 3 * s->L;
    double lambda = 0, ni = 2;
    int nBad = 0, done = 0;
    double* save_pose = (double*)malloc(sizeof(double) * 7 * (size_t)(s->P + 1));
    double* save_pt = (double*)malloc(sizeof(double) * 3 * (size_t)(s->L + 1));
    for (int it = 0; it < iterations; it++) {
        double currentChi = compute_errors(s);
        const double iniChi = currentChi;
        if (res && res->iterations_done == 0 && it == 0 && res->chi2_initial < 0) res->chi2_initial = currentChi;
        build_system(s);
        if (it == 0) {                                             /* computeLambdaInit :166-180 */
            double md = 0;
            for (int f = 0; f < s->nfree; f++) for (int j = 0; j < 6; j++) md = fmax(fabs(s->Hpp[36 * (size_t)f + 7 * j]), md);
            for (int l = 0; l < s->L; l++) for (int j = 0; j < 3; j++) md = fmax(fabs(s->Hll[9 * (size_t)l + 4 * j]), md);
            lambda = 1e-5 * md; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(save_pose, pb->poses, sizeof(double) * 7 * (size_t)s->P);      /* push */
            memcpy(save_pt, pb->points, sizeof(double) * 3 * (size_t)s->L);
            const int ok2 = schur_solve(s, lambda);
            if (res) res->trials++;
            double tempChi;
            if (ok2) {
                for (int p = 0; p < s->P; p++) {                                     /* update :422-435 */
                    const int f = s->free_of[p];
                    if (f < 0) continue;
                    double o[7];
                    orc_se3_exp_mul(s->x + 6 * f, pb->poses + 7 * p, o);
                    memcpy(pb->poses + 7 * p, o, sizeof o);
                }
                for (int l = 0; l < s->L; l++) for (int j = 0; j < 3; j++) pb->points[3 * l + j] += s->x[n + 3 * l + j];
                tempChi = compute_errors(s);
            } else tempChi = DBL_MAX;
            double scale = 0;                                                         /* computeScale :182-189 */
            for (int j = 0; j < nx; j++) {
                const double bj = j < n ? s->bp[j] : s->bl[j - n];
                scale += s->x[j] * (lambda * s->x[j] + bj);
            }
            scale += 1e-3;
            rho = ok2 ? (currentChi - tempChi) / scale : -1.0;   /* failed factorisation = rejected step */
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                const double sf = fmax(1. / 3., alpha);
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(pb->poses, save_pose, sizeof(double) * 7 * (size_t)s->P);   /* pop */
                memcpy(pb->points, save_pt, sizeof(double) * 3 * (size_t)s->L);
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        done++;
        if (res) { res->iterations_done++; res->chi2_final = currentChi; res->lambda_final = lambda; }
        if (qmax == 10 || rho == 0) break;                                            /* Terminate :151 */
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;              /* :154-161 */
        if (nBad >= 3) break;
    }
    free(save_pose); free(save_pt);
    if (lambda_io) *lambda_io = lambda;
    return done;
}

int orc_ba_solve(orc_ba_problem* pb, const orc_ba_options* opt, orc_ba_result* res, uint8_t* edge_outlier)
{
    if (!pb || !opt) return -1;
    ba_t* s = ba_new(pb);
    orc_ba_result local; if (!res) res = &local;
    memset(res, 0, sizeof *res);
    res->chi2_initial = -1;
    s->huber_delta = opt->huber_delta;
    lm_optimize(s, opt->iterations, res, 0);
    if (opt->iterations2 > 0) {                                    /* src/Optimizer.cpp:546-568 */
        refresh_R(s);
        for (int e = 0; e < s->E; e++) {
            const int pi = pb->edge_pose[e], li = pb->edge_point[e];
            const double* R = s->R + 9 * pi; const double* p = pb->points + 3 * li;
            const double z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + pb->poses[7 * pi + 6];
            const double c2 = pb->info[e] * (s->err[2 * e] * s->err[2 * e] + s->err[2 * e + 1] * s->err[2 * e + 1]);
            if (c2 > opt->outlier_chi2 || !(z > 0.0)) s->active[e] = 0;
        }
        s->huber_delta = 0;
        lm_optimize(s, opt->iterations2, res, 0);
    }
    if (edge_outlier) {                                            /* src/Optimizer.cpp:574-588 */
        refresh_R(s);
        for (int e = 0; e < s->E; e++) {
            const int pi = pb->edge_pose[e], li = pb->edge_point[e];
            const double* R = s->R + 9 * pi; const double* p = pb->points + 3 * li;
            const double z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + pb->poses[7 * pi + 6];
            const double c2 = pb->info[e] * (s->err[2 * e] * s->err[2 * e] + s->err[2 * e + 1] * s->err[2 * e + 1]);
            edge_outlier[e] = (c2 > opt->outlier_chi2 || !(z > 0.0)) ? 1 : 0;
        }
    }
    ba_free(s);
    return 0;
}

int orc_ba_reduced_system(const orc_ba_problem* pb, double huber_delta, double lambda,
                          double* Hschur, double* bschur, int32_t* free_index)
{
    ba_t* s = ba_new_mode((orc_ba_problem*)pb, 1);            /* dense by definition */
    s->huber_delta = huber_delta;
    compute_errors(s);
    build_system(s);
    schur_reduce(s, lambda);
    const int n = 6 * s->nfree;
    if (Hschur) memcpy(Hschur, s->Hs, sizeof(double) * (size_t)n * n);
    if (bschur) memcpy(bschur, s->bs, sizeof(double) * n);
    if (free_index) for (int i = 0; i < s->P; i++) free_index[i] = s->free_of[i];
    const int P = s->nfree;
    ba_free(s);
    return P;
}

/* One linearisation at the current state solved with the selected solver (mode as orc_ba_set_solver): the keyframe
 * increment xp [6 nfree] and the landmark increment xl [3 L]; returns nfree, or -1 when the factorisation fails.
 * stats (optional): [0] non-zero blocks of the factor, [1] flops of one factorisation, [2] upper blocks of Hschur. */
int orc_ba_solve_once(const orc_ba_problem* pb, double huber_delta, double lambda, int mode, double* xp, double* xl, double* stats)
{
    ba_t* s = ba_new_mode((orc_ba_problem*)pb, mode);
    s->huber_delta = huber_delta;
    compute_errors(s);
    build_system(s);
    const int ok = schur_solve(s, lambda);
    const int n = 6 * s->nfree, P = s->nfree;
    if (ok && xp) memcpy(xp, s->x, sizeof(double) * n);
    if (ok && xl) memcpy(xl, s->x + n, sizeof(double) * 3 * (size_t)s->L);
    if (stats) { stats[0] = s->chol ? (double)orc_bchol_nnz(s->chol) : 0; stats[1] = s->chol ? orc_bchol_flops(s->chol) : 0; stats[2] = s->nsblk; }
    ba_free(s);
    return ok ? P : -1;
}

/* ---------------------------------------------------------------- F2: pose-only optimisation
 * Optimizer::PoseOptimizationClient (src/Optimizer.cpp:215-347): one free pose, unary
 * EdgeSE3ProjectXYZOnlyPose edges (types_six_dof_expmap.cpp:266-296), Huber sqrt(5.991), four rounds of
 * optimize(10) each restarted from the frame's initial pose, chi2 > 5.991 (compared in float) relabels
 * outliers after every round, the kernel is dropped after the third round.  No marginalised vertex, so
 * BlockSolver solves the 6x6 system directly (LinearSolverDense, Eigen LDLT). */
static int chol6_solve(const double* H, const double* b, double* x)
{
    double L[36];
    memcpy(L, H, sizeof L);
    if (!chol_factor(L, 6)) return 0;
    memcpy(x, b, 6 * sizeof(double));
    chol_solve(L, 6, x);
    return 1;
}

int orc_pose_optimize(double* pose7, const double* intr4, int n, const double* pts, const double* obs,
                      const double* info, uint8_t* outlier, int* n_inliers)
{
    *n_inliers = 0;
    for (int i = 0; i < n; i++) outlier[i] = 0;
    if (n < 3) return 0;                                           /* :296-297 */
    double pose0[7]; memcpy(pose0, pose7, sizeof pose0);
    double* err = (double*)calloc(2 * (size_t)n, sizeof(double));
    const double delta = sqrt(5.991);
    int nBadEdges = 0;
    for (int round = 0; round < 4; round++) {
        const int robust = round <= 2;                             /* kernel removed at the end of it == 2 */
        memcpy(pose7, pose0, sizeof pose0);                        /* :309 */
        double lambda = 0, ni = 2; int nBad = 0;
        for (int it = 0; it < 10; it++) {
            double R[9]; quat_to_R(pose7, R);
            double H[36], b[6], cur = 0;
            memset(H, 0, sizeof H); memset(b, 0, sizeof b);
            int nact = 0;
            for (int e = 0; e < n; e++) {
                if (outlier[e]) continue;
                nact++;
                double A[6], B[12];
                edge_eval(R, pose7 + 4, intr4, pts + 3 * e, obs + 2 * e, err + 2 * e, A, B, 0);
                const double c2 = info[e] * (err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1]);
                double r0 = c2, r1 = 1.;
                if (robust) huber(c2, delta, &r0, &r1);
                cur += r0;
                const double w = r1 * info[e], g0 = -info[e] * err[2 * e] * r1, g1 = -info[e] * err[2 * e + 1] * r1;
                for (int i = 0; i < 6; i++) {
                    b[i] += B[i] * g0 + B[6 + i] * g1;
                    for (int j = 0; j < 6; j++) H[i * 6 + j] += w * (B[i] * B[j] + B[6 + i] * B[6 + j]);
                }
            }
            if (nact == 0) break;                                  /* nothing active: g2o optimises nothing */
            const double ini = cur;
            if (it == 0) {
                double md = 0;
                for (int j = 0; j < 6; j++) md = fmax(md, fabs(H[7 * j]));
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            double rho = 0; int qmax = 0;
            do {
                double save[7]; memcpy(save, pose7, sizeof save);
                double Hl[36], x[6] = { 0, 0, 0, 0, 0, 0 };
                memcpy(Hl, H, sizeof Hl);
                for (int j = 0; j < 6; j++) Hl[7 * j] += lambda;
                const int ok2 = chol6_solve(Hl, b, x);
                double temp = DBL_MAX;
                if (ok2) {
                    double o[7]; orc_se3_exp_mul(x, pose7, o); memcpy(pose7, o, sizeof o);
                    double R2[9]; quat_to_R(pose7, R2);
                    temp = 0;
                    for (int e = 0; e < n; e++) {
                        if (outlier[e]) continue;
                        edge_eval(R2, pose7 + 4, intr4, pts + 3 * e, obs + 2 * e, err + 2 * e, 0, 0, 0);
                        const double c2 = info[e] * (err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1]);
                        double r0 = c2, r1;
                        if (robust) huber(c2, delta, &r0, &r1);
                        temp += r0;
                    }
                }
                double scale = 1e-3;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                rho = ok2 ? (cur - temp) / scale : -1.0;
                if (rho > 0 && isfinite(temp)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha); ni = 2; cur = temp;
                } else { lambda *= ni; ni *= 2; memcpy(pose7, save, sizeof save); }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;
            if ((ini - cur) * 1e3 < ini) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        /* classification (:313-338): inactive edges get a fresh error, active ones keep the last computed one */
        double R[9]; quat_to_R(pose7, R);
        nBadEdges = 0;
        for (int e = 0; e < n; e++) {
            if (outlier[e]) edge_eval(R, pose7 + 4, intr4, pts + 3 * e, obs + 2 * e, err + 2 * e, 0, 0, 0);
            const float chi2 = (float)(info[e] * (err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1]));
            if (chi2 > 5.991f) { outlier[e] = 1; nBadEdges++; } else outlier[e] = 0;
        }
        if (n < 10) break;                                         /* :340-341 */
    }
    free(err);
    *n_inliers = n - nBadEdges;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * F4 (first half): Optimizer::OptimizeSim3 (cslam/src/Optimizer.cpp:867-1062).
 * One VertexSim3Expmap (types_seven_dof_expmap.h:52-107, Sim3 = sim3.h), fixed points, two edges per
 * correspondence: EdgeSim3ProjectXYZ (:146-166) and EdgeInverseSim3ProjectXYZ (:169-189), whose Jacobians g2o takes
 * numerically (base_binary_edge.hpp:147-196: central differences, delta 1e-9); Huber(sqrt(th2)) on every edge;
 * BlockSolverX + LinearSolverDense (7x7 LDLT) + Levenberg: optimize(5), drop the pairs with chi2 > th2, optimize(5
 * or 10), classify again.  sim3 = qx,qy,qz,qw, tx,ty,tz, s. */
static void rotv(const double* q, const double* v, double* o)
{
    double R[9]; quat_to_R(q, R);
    for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
/* Sim3(const Vector7d& update), sim3.h:62-131 */
void orc_sim3_exp(const double* u, double* S)
{
    const double* omega = u; const double* upsilon = u + 3;
    const double sigma = u[6];
    const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    const double Om[9] = { 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0 };
    double Om2[9]; mat3_mul(Om, Om, Om2);
    const double s = exp(sigma);
    const double eps = 0.00001;
    double A, B, C, R[9];
    const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) { A = 1. / 2.; B = 1. / 6.; for (int i = 0; i < 9; i++) R[i] = I[i] + Om[i] + Om2[i]; }
        else {
            const double theta2 = theta * theta;
            A = (1 - cos(theta)) / theta2;
            B = (theta - sin(theta)) / (theta2 * theta);
            for (int i = 0; i < 9; i++) R[i] = I[i] + sin(theta) / theta * Om[i] + (1 - cos(theta)) / (theta * theta) * Om2[i];
        }
    } else {
        C = (s - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * s + 1) / sigma2;
            B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
            for (int i = 0; i < 9; i++) R[i] = I[i] + Om[i] + Om2[i];
        } else {
            for (int i = 0; i < 9; i++) R[i] = I[i] + sin(theta) / theta * Om[i] + (1 - cos(theta)) / (theta * theta) * Om2[i];
            const double a = s * sin(theta), b = s * cos(theta);
            const double theta2 = theta * theta, sigma2 = sigma * sigma;
            const double c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    R_to_quat(R, S);
    for (int i = 0; i < 3; i++) {
        S[4 + i] = 0;
        for (int j = 0; j < 3; j++) S[4 + i] += (A * Om[3 * i + j] + B * Om2[3 * i + j] + C * I[3 * i + j]) * upsilon[j];
    }
    S[7] = s;
}
/* Sim3::operator*, sim3.h:277-283 */
void orc_sim3_mul(const double* a, const double* b, double* o)
{
    double q[4], t[3];
    quat_mul(a, b, q);
    rotv(a, b + 4, t);
    for (int i = 0; i < 4; i++) o[i] = q[i];
    for (int i = 0; i < 3; i++) o[4 + i] = a[7] * t[i] + a[4 + i];
    o[7] = a[7] * b[7];
}
/* Sim3::inverse, sim3.h:245-248 */
void orc_sim3_inverse(const double* a, double* o)
{
    const double qc[4] = { -a[0], -a[1], -a[2], a[3] };
    const double ts[3] = { (-1. / a[7]) * a[4], (-1. / a[7]) * a[5], (-1. / a[7]) * a[6] };
    double t[3]; rotv(qc, ts, t);
    for (int i = 0; i < 4; i++) o[i] = qc[i];
    for (int i = 0; i < 3; i++) o[4 + i] = t[i];
    o[7] = 1. / a[7];
}
static void sim3_map(const double* S, const double* x, double* o)
{
    double r[3]; rotv(S, x, r);
    for (int i = 0; i < 3; i++) o[i] = S[7] * r[i] + S[4 + i];
}
/* the two edge errors of correspondence e at estimate S (types_seven_dof_expmap.h:155-163, :178-186) */
static void sim3_errors(const double* S, const double* K1, const double* K2, const double* P1, const double* P2,
                        const double* obs1, const double* obs2, double* e12, double* e21)
{
    double x[3], Si[8];
    sim3_map(S, P2, x);
    e12[0] = obs1[0] - ((x[0] / x[2]) * K1[0] + K1[2]);
    e12[1] = obs1[1] - ((x[1] / x[2]) * K1[1] + K1[3]);
    orc_sim3_inverse(S, Si);
    sim3_map(Si, P1, x);
    e21[0] = obs2[0] - ((x[0] / x[2]) * K2[0] + K2[2]);
    e21[1] = obs2[1] - ((x[1] / x[2]) * K2[1] + K2[3]);
}
static void sim3_oplus(const double* S, const double* upd, int fix_scale, double* o)
{
    double u[7]; memcpy(u, upd, sizeof u);
    if (fix_scale) u[6] = 0;
    double E[8]; orc_sim3_exp(u, E);
    orc_sim3_mul(E, S, o);
}

/* One optimize(iterations) call on the active correspondences; err12/err21 keep the last computed edge errors. */
static void sim3_lm(double* S, int fix_scale, const double* K1, const double* K2, int n, const uint8_t* active,
                    const double* P1, const double* P2, const double* obs1, const double* obs2, const double* info1,
                    const double* info2, double delta, int iterations, double* err12, double* err21)
{
    double lambda = 0, ni = 2; int nBad = 0;
    for (int it = 0; it < iterations; it++) {
        double H[49], b[7], cur = 0;
        memset(H, 0, sizeof H); memset(b, 0, sizeof b);
        int nact = 0;
        for (int e = 0; e < n; e++) {
            if (!active[e]) continue;
            nact++;
            sim3_errors(S, K1, K2, P1 + 3 * e, P2 + 3 * e, obs1 + 2 * e, obs2 + 2 * e, err12 + 2 * e, err21 + 2 * e);
            double J12[14], J21[14];                                   /* [row][d] */
            for (int d = 0; d < 7; d++) {
                double up[7] = { 0, 0, 0, 0, 0, 0, 0 }, Sp[8], Sm[8], a12[2], a21[2], b12[2], b21[2];
                up[d] = 1e-9; sim3_oplus(S, up, fix_scale, Sp);
                sim3_errors(Sp, K1, K2, P1 + 3 * e, P2 + 3 * e, obs1 + 2 * e, obs2 + 2 * e, a12, a21);
                up[d] = -1e-9; sim3_oplus(S, up, fix_scale, Sm);
                sim3_errors(Sm, K1, K2, P1 + 3 * e, P2 + 3 * e, obs1 + 2 * e, obs2 + 2 * e, b12, b21);
                const double scalar = 1.0 / (2 * 1e-9);
                J12[d] = scalar * (a12[0] - b12[0]); J12[7 + d] = scalar * (a12[1] - b12[1]);
                J21[d] = scalar * (a21[0] - b21[0]); J21[7 + d] = scalar * (a21[1] - b21[1]);
            }
            for (int k = 0; k < 2; k++) {                              /* edge order of the graph: e12 then e21 */
                const double* er = k == 0 ? err12 + 2 * e : err21 + 2 * e;
                const double* J = k == 0 ? J12 : J21;
                const double inf = k == 0 ? info1[e] : info2[e];
                const double c2 = inf * (er[0] * er[0] + er[1] * er[1]);
                double r0, r1; huber(c2, delta, &r0, &r1);
                cur += r0;
                const double w = r1 * inf, g0 = -inf * er[0] * r1, g1 = -inf * er[1] * r1;
                for (int i = 0; i < 7; i++) {
                    b[i] += J[i] * g0 + J[7 + i] * g1;
                    for (int j = 0; j < 7; j++) H[i * 7 + j] += w * (J[i] * J[j] + J[7 + i] * J[7 + j]);
                }
            }
        }
        if (nact == 0) break;
        const double ini = cur;
        if (it == 0) {
            double md = 0;
            for (int j = 0; j < 7; j++) md = fmax(md, fabs(H[8 * j]));
            lambda = 1e-5 * md; ni = 2; nBad = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            double save[8]; memcpy(save, S, sizeof save);
            double Hl[49], x[7] = { 0, 0, 0, 0, 0, 0, 0 };
            memcpy(Hl, H, sizeof Hl);
            for (int j = 0; j < 7; j++) Hl[8 * j] += lambda;
            int ok2 = chol_factor(Hl, 7);
            if (ok2) { memcpy(x, b, sizeof x); chol_solve(Hl, 7, x); }
            double temp = DBL_MAX;
            if (ok2) {
                double o[8]; sim3_oplus(S, x, fix_scale, o); memcpy(S, o, sizeof o);
                temp = 0;
                for (int e = 0; e < n; e++) {
                    if (!active[e]) continue;
                    sim3_errors(S, K1, K2, P1 + 3 * e, P2 + 3 * e, obs1 + 2 * e, obs2 + 2 * e, err12 + 2 * e, err21 + 2 * e);
                    double r0, r1;
                    huber(info1[e] * (err12[2 * e] * err12[2 * e] + err12[2 * e + 1] * err12[2 * e + 1]), delta, &r0, &r1); temp += r0;
                    huber(info2[e] * (err21[2 * e] * err21[2 * e] + err21[2 * e + 1] * err21[2 * e + 1]), delta, &r0, &r1); temp += r0;
                }
            }
            double scale = 1e-3;
            for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + b[j]);
            rho = ok2 ? (cur - temp) / scale : -1.0;
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; cur = temp;
            } else { lambda *= ni; ni *= 2; memcpy(S, save, sizeof save); }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;
        if ((ini - cur) * 1e3 < ini) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
}

/* Returns nIn; inlier[e] = the correspondence survives (vpMatches1 entry kept); sim3 is updated unless fewer than 10
 * correspondences survive the first round (:1022-1023: return 0 before the estimate is read back). */
int orc_optimize_sim3(double* sim3, int fix_scale, const double* K1, const double* K2, int n, const double* P1, const double* P2,
                      const double* obs1, const double* obs2, const double* info1, const double* info2, float th2, uint8_t* inlier)
{
    double S[8]; memcpy(S, sim3, sizeof S);
    double* err12 = (double*)calloc(2 * (size_t)(n > 0 ? n : 1), sizeof(double));
    double* err21 = (double*)calloc(2 * (size_t)(n > 0 ? n : 1), sizeof(double));
    const double delta = (double)sqrtf(th2);                           /* const float deltaHuber = sqrt(th2), :903 */
    for (int e = 0; e < n; e++) inlier[e] = 1;
    sim3_lm(S, fix_scale, K1, K2, n, inlier, P1, P2, obs1, obs2, info1, info2, delta, 5, err12, err21);
    int nBad = 0;
    for (int e = 0; e < n; e++) {
        const double c12 = info1[e] * (err12[2 * e] * err12[2 * e] + err12[2 * e + 1] * err12[2 * e + 1]);
        const double c21 = info2[e] * (err21[2 * e] * err21[2 * e] + err21[2 * e + 1] * err21[2 * e + 1]);
        if (c12 > th2 || c21 > th2) { inlier[e] = 0; nBad++; }
    }
    const int more = nBad > 0 ? 10 : 5;
    if (n - nBad < 10) { free(err12); free(err21); return 0; }
    sim3_lm(S, fix_scale, K1, K2, n, inlier, P1, P2, obs1, obs2, info1, info2, delta, more, err12, err21);
    int nIn = 0;
    for (int e = 0; e < n; e++) {
        if (!inlier[e]) continue;
        const double c12 = info1[e] * (err12[2 * e] * err12[2 * e] + err12[2 * e + 1] * err12[2 * e + 1]);
        const double c21 = info2[e] * (err21[2 * e] * err21[2 * e] + err21[2 * e + 1] * err21[2 * e + 1]);
        if (c12 > th2 || c21 > th2) inlier[e] = 0; else nIn++;
    }
    memcpy(sim3, S, sizeof S);
    free(err12); free(err21);
    return nIn;
}

/* ------------------------------------------------------------------------------------------------
 * F4 (second half): the optimisation inside Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion
 * (cslam/src/Optimizer.cpp:1064-1331, :1333-1574): VertexSim3Expmap per keyframe, EdgeSim3 (types_seven_dof_expmap.h:
 * 112-140) with identity information and NUMERIC Jacobians on both vertices, BlockSolver_7_3 + LinearSolverEigen,
 * Levenberg with setUserLambdaInit(1e-16), optimize(20).  The graph (which edges, their measurements Sji) is built by
 * the caller exactly as :1123-1250. */
/* Sim3::log, sim3.h:137-213 */
void orc_sim3_log(const double* S, double* res)
{
    const double s = S[7];
    const double sigma = log(s);
    double R[9]; quat_to_R(S, R);
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
    double Om2[9]; mat3_mul(Om, Om, Om2);
    double W[9];
    for (int i = 0; i < 9; i++) W[i] = A * Om[i] + B * Om2[i] + C * ((i % 4) == 0 ? 1.0 : 0.0);
    /* upsilon = W.lu().solve(t): Gaussian elimination with partial pivoting */
    double M[12];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[4 * i + j] = W[3 * i + j]; M[4 * i + 3] = S[4 + i]; }
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

/* EdgeSim3::computeError: log(C * Si * Sj^-1), types_seven_dof_expmap.h:119-127 */
static void ess_error(const double* C, const double* Si, const double* Sj, double* e)
{
    double t1[8], inv[8], t2[8];
    orc_sim3_mul(C, Si, t1);
    orc_sim3_inverse(Sj, inv);
    orc_sim3_mul(t1, inv, t2);
    orc_sim3_log(t2, e);
}

/* Solver of the pose graph's normal equations: 0 automatic (block-sparse Cholesky above 400 free vertices), 1 dense, 2 block-sparse.
 * The reference uses g2o::BlockSolver_7_3 + LinearSolverEigen (sparse LDL^T, src/Optimizer.cpp:1072-1074); bchol_oracle.c restates that
 * method on 7x7 blocks, the dense Cholesky is the same system factored without reordering (results agree to rounding). */
static int g_ess_solver = 0;
void orc_ess_set_solver(int mode) { g_ess_solver = mode; }

/* sim3 [n][8] in/out; fixed[n]; edges (ei[k] = vertex 0, ej[k] = vertex 1, meas[k] = Sji).  Returns the iterations done; chi2 in out[0..1]. */
int orc_essential_graph(int n, double* sim3, const uint8_t* fixed, int fix_scale, int ne, const int32_t* ei, const int32_t* ej,
                        const double* meas, int iterations, double* chi2_out)
{
    int* fidx = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    int nf = 0;
    for (int i = 0; i < n; i++) fidx[i] = fixed[i] ? -1 : nf++;
    const int N = 7 * nf;
    const int sparse = g_ess_solver == 2 || (g_ess_solver == 0 && nf > 400);
    /* block-sparse form: slot[ia * nf + ib] (ia <= ib) -> 7x7 block of H in Hb, pattern = the edges */
    int32_t* slot = NULL; double* Hb = NULL; double* Hbl = NULL; orc_bchol* chol = NULL; int nslot = 0;
    if (sparse && nf > 0) {
        slot = (int32_t*)malloc(sizeof(int32_t) * (size_t)nf * nf);
        memset(slot, 0xff, sizeof(int32_t) * (size_t)nf * nf);
        uint8_t* adj = (uint8_t*)calloc((size_t)nf * nf, 1);
        for (int i = 0; i < nf; i++) slot[(size_t)i * nf + i] = nslot++;
        for (int k = 0; k < ne; k++) {
            const int ia = fidx[ei[k]], ib = fidx[ej[k]];
            if (ia < 0 || ib < 0 || ia == ib) continue;
            const int lo = ia < ib ? ia : ib, hi = ia < ib ? ib : ia;
            if (slot[(size_t)lo * nf + hi] < 0) { slot[(size_t)lo * nf + hi] = nslot++; adj[(size_t)lo * nf + hi] = 1; }
        }
        chol = orc_bchol_new_bs(nf, adj, 7);
        free(adj);
        Hb = (double*)malloc(sizeof(double) * 49 * (size_t)nslot);
        Hbl = (double*)malloc(sizeof(double) * 49 * (size_t)nslot);
    }
    const size_t NN = sparse ? 1 : (size_t)(N > 0 ? N : 1) * (N > 0 ? N : 1);
    double* H = (double*)calloc(NN, sizeof(double));
    double* Hl = (double*)malloc(sizeof(double) * NN);
    double* b = (double*)calloc(N > 0 ? N : 1, sizeof(double));
    double* x = (double*)calloc(N > 0 ? N : 1, sizeof(double));
    double* save = (double*)malloc(sizeof(double) * 8 * (size_t)(n > 0 ? n : 1));
    double lambda = 0, ni = 2;
    int nBad = 0, done = 0;
    double first_chi = -1, cur = 0;
    for (int it = 0; it < iterations && N > 0 && ne > 0; it++) {
        if (sparse) memset(Hb, 0, sizeof(double) * 49 * (size_t)nslot); else memset(H, 0, sizeof(double) * (size_t)N * N);
        memset(b, 0, sizeof(double) * N);
        cur = 0;
        for (int k = 0; k < ne; k++) {
            const double* C = meas + 8 * (size_t)k;
            double* Si = sim3 + 8 * (size_t)ei[k]; double* Sj = sim3 + 8 * (size_t)ej[k];
            double e[7]; ess_error(C, Si, Sj, e);
            for (int r = 0; r < 7; r++) cur += e[r] * e[r];
            double J[2][49];                                    /* [vertex][row * 7 + d] */
            for (int v = 0; v < 2; v++) {
                const int vid = v == 0 ? ei[k] : ej[k];
                if (fixed[vid]) continue;
                for (int d = 0; d < 7; d++) {
                    double up[7] = { 0, 0, 0, 0, 0, 0, 0 }, Sp[8], Sm[8], ea[7], eb[7];
                    up[d] = 1e-9; sim3_oplus(v == 0 ? Si : Sj, up, fix_scale, Sp);
                    ess_error(C, v == 0 ? Sp : Si, v == 0 ? Sj : Sp, ea);
                    up[d] = -1e-9; sim3_oplus(v == 0 ? Si : Sj, up, fix_scale, Sm);
                    ess_error(C, v == 0 ? Sm : Si, v == 0 ? Sj : Sm, eb);
                    for (int r = 0; r < 7; r++) J[v][r * 7 + d] = (1.0 / (2 * 1e-9)) * (ea[r] - eb[r]);
                }
            }
            for (int va = 0; va < 2; va++) {
                const int ia = fidx[va == 0 ? ei[k] : ej[k]];
                if (ia < 0) continue;
                for (int p = 0; p < 7; p++) {
                    double g = 0;
                    for (int r = 0; r < 7; r++) g += J[va][r * 7 + p] * (-e[r]);
                    b[7 * ia + p] += g;
                }
                for (int vb = 0; vb < 2; vb++) {
                    const int ib = fidx[vb == 0 ? ei[k] : ej[k]];
                    if (ib < 0) continue;
                    if (sparse && ia > ib) continue;                        /* upper blocks only */
                    double* Hs = sparse ? Hb + 49 * (size_t)slot[(size_t)ia * nf + ib] : NULL;
                    for (int p = 0; p < 7; p++) for (int q = 0; q < 7; q++) {
                        double h = 0;
                        for (int r = 0; r < 7; r++) h += J[va][r * 7 + p] * J[vb][r * 7 + q];
                        if (sparse) Hs[p * 7 + q] += h; else H[(size_t)(7 * ia + p) * N + 7 * ib + q] += h;
                    }
                }
            }
        }
        if (first_chi < 0) first_chi = cur;
        const double ini = cur;
        if (it == 0) { lambda = 1e-16; ni = 2; nBad = 0; }      /* setUserLambdaInit(1e-16), :1073 */
        double rho = 0; int qmax = 0;
        do {
            memcpy(save, sim3, sizeof(double) * 8 * (size_t)n);
            int ok2;
            if (sparse) {
                memcpy(Hbl, Hb, sizeof(double) * 49 * (size_t)nslot);
                for (int i = 0; i < nf; i++) for (int d = 0; d < 7; d++) Hbl[49 * (size_t)slot[(size_t)i * nf + i] + 8 * d] += lambda;
                ok2 = orc_bchol_factor(chol, slot, Hbl);
            } else {
                memcpy(Hl, H, sizeof(double) * (size_t)N * N);
                for (int j = 0; j < N; j++) Hl[(size_t)j * N + j] += lambda;
                ok2 = chol_factor(Hl, N);
            }
            double temp = DBL_MAX;
            if (ok2) {
                if (sparse) orc_bchol_solve(chol, b, x);
                else { memcpy(x, b, sizeof(double) * N); chol_solve(Hl, N, x); }
                for (int v = 0; v < n; v++) {
                    if (fidx[v] < 0) continue;
                    double o[8]; sim3_oplus(sim3 + 8 * (size_t)v, x + 7 * fidx[v], fix_scale, o);
                    memcpy(sim3 + 8 * (size_t)v, o, sizeof o);
                }
                temp = 0;
                for (int k = 0; k < ne; k++) {
                    double e[7]; ess_error(meas + 8 * (size_t)k, sim3 + 8 * (size_t)ei[k], sim3 + 8 * (size_t)ej[k], e);
                    for (int r = 0; r < 7; r++) temp += e[r] * e[r];
                }
            } else memset(x, 0, sizeof(double) * N);
            double scale = 1e-3;
            for (int j = 0; j < N; j++) scale += x[j] * (lambda * x[j] + b[j]);
            rho = ok2 ? (cur - temp) / scale : -1.0;
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; cur = temp;
            } else { lambda *= ni; ni *= 2; memcpy(sim3, save, sizeof(double) * 8 * (size_t)n); }
            qmax++;
        } while (rho < 0 && qmax < 10);
        done = it + 1;
        if (qmax == 10 || rho == 0) break;
        if ((ini - cur) * 1e3 < ini) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    if (chi2_out) { chi2_out[0] = first_chi < 0 ? 0 : first_chi; chi2_out[1] = cur; }
    free(fidx); free(H); free(Hl); free(b); free(x); free(save); free(slot); free(Hb); free(Hbl); orc_bchol_free(chol);
    return done;
}
