// tests/test_abi_cpu.py::test_landmark_cholesky_pivots -- ba_math.h's 3 x 3 Cholesky of a landmark block (the operand of the Schur
// product, k_sp_edge_y / the fused linearisation) on the host: well-conditioned blocks give the plain factor, a rank-deficient block
// (one observation: H = J^T J of a 2 x 3 Jacobian) with a damping far below the rounding level of its entries stays finite.
#include <cmath>
#include <cstdio>
#include <initializer_list>
#include "../../motioncheck_ccm_slam_amd/csrc/ba_math.h"

int main()
{
    // 1. SPD block: L L^T reproduces H + lambda I
    {
        const double H[9] = { 4, 1, 0.5, 1, 3, 0.2, 0.5, 0.2, 2 };
        const double lambda = 1e-3;
        double f[6];
        ba_chol3(H, lambda, f);
        const double l00 = 1 / f[0], l10 = f[1], l20 = f[2], l11 = 1 / f[3], l21 = f[4], l22 = 1 / f[5];
        const double R[6] = { l00 * l00, l10 * l00, l20 * l00, l10 * l10 + l11 * l11, l20 * l10 + l21 * l11, l20 * l20 + l21 * l21 + l22 * l22 };
        const double E[6] = { H[0] + lambda, H[1], H[2], H[4] + lambda, H[5], H[8] + lambda };
        double err = 0;
        for (int i = 0; i < 6; i++) err = std::fmax(err, std::fabs(R[i] - E[i]));
        std::printf("spd_err=%.3e\n", err);
    }
    // 2. one observation: rank 2, lambda = 1e-12 of the diagonal and smaller
    int finite = 1;
    for (double rel : { 1e-12, 1e-16, 1e-20, 0.0 }) {
        const double J[6] = { 400.0, 0.0, -37.0, 0.0, 400.0, 21.0 };               // 2 x 3, pixels per metre at depth 1
        double H[9];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) H[3 * a + b] = J[a] * J[b] + J[3 + a] * J[3 + b];
        double f[6], z[18], c[6];
        ba_chol3(H, rel * H[0], f);
        const double B[18] = { 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18 };
        const double d[3] = { 0.1, 0.2, 0.3 };
        ba_edge_z_c(B, f, d, z, c);
        for (int i = 0; i < 6; i++) finite = finite && std::isfinite(f[i]) && f[i] == f[i];
        for (int i = 0; i < 18; i++) finite = finite && std::isfinite(z[i]);
    }
    std::printf("rank_deficient_finite=%d\n", finite);
    return 0;
}
