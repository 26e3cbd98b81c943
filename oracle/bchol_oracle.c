/* bchol_oracle.c -- block-sparse Cholesky (6x6 blocks) of the reduced camera system.
 * TEST INFRASTRUCTURE (see oracle.h).
 *
 * The reference solves Hschur x = bschur with g2o::LinearSolverEigen
 * (cslam/thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:106-136, selected at src/Optimizer.cpp:679):
 * Eigen::SimplicialLDLT on the scalar sparse matrix with an approximate-minimum-degree ordering computed
 * once (computeSymbolicDecomposition, :165-222), numeric factorisation per LM trial, "not positive
 * definite" reported as a failed solve (:116-123).  Eigen is not in the tree; this file restates the same
 * method -- fill-reducing ordering once, exact sparse factorisation per solve -- on the 6x6 block
 * pattern: greedy minimum-degree elimination ordering of the keyframe graph (exact degrees, lowest index
 * on ties), symbolic fill from the same elimination, right-looking block L L^T.  LDL^T vs L L^T and a
 * different elimination order change rounding only, so results agree with the
 * dense Cholesky of ba_oracle.c to ~1e-13 relative (tests/test_oracle_kat.py checks it) and the method
 * scales to the 2000-keyframe graph of BASELINE config 5 (the dense O(n^3) factorisation does not).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_bchol {
    int nb;
    int bs;         /* block size: 6 (reduced camera system) or 7 (Sim3 pose graph) */
    int* perm;      /* step -> original block */
    int* iperm;     /* original block -> step */
    int* colptr;    /* [nb+1] into rowidx / L, strictly-lower blocks of permuted column k */
    int* rowidx;    /* permuted row of each strictly-lower block, ascending per column */
    int* pos;       /* [nb*nb] permuted (i,j), i > j -> slot in L, or -1 */
    double* D;      /* [nb][36] diagonal blocks (row-major), lower Cholesky factor after factorisation */
    double* L;      /* [nnz][36] block (i,k) row-major 6x6 */
    long nnz;
    double flops;   /* multiply-adds x2 of one numeric factorisation (for the record) */
};

static inline int popc64(unsigned long long v) { return __builtin_popcountll(v); }

orc_bchol* orc_bchol_new_bs(int nb, const uint8_t* adj /* nb*nb, non-zero where a block exists (any triangle) */, int bs)
{
    orc_bchol* c = (orc_bchol*)calloc(1, sizeof *c);
    c->nb = nb; c->bs = bs;
    const int BS = bs, BB = bs * bs;
    const int W = (nb + 63) / 64;
    unsigned long long* A = (unsigned long long*)calloc((size_t)nb * W + 1, 8);
    for (int i = 0; i < nb; i++)
        for (int j = 0; j < nb; j++)
            if (i != j && (adj[(size_t)i * nb + j] || adj[(size_t)j * nb + i])) A[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63);
    c->perm = (int*)malloc(sizeof(int) * (nb + 1));
    c->iperm = (int*)malloc(sizeof(int) * (nb + 1));
    uint8_t* alive = (uint8_t*)malloc(nb + 1);
    memset(alive, 1, nb + 1);
    int* deg = (int*)malloc(sizeof(int) * (nb + 1));
    for (int i = 0; i < nb; i++) { int d = 0; for (int w = 0; w < W; w++) d += popc64(A[(size_t)i * W + w]); deg[i] = d; }
    /* column patterns in original ids, converted once the order is known */
    int** colrows = (int**)calloc(nb + 1, sizeof(int*));
    int* colcnt = (int*)calloc(nb + 1, sizeof(int));
    unsigned long long* nk = (unsigned long long*)malloc(8 * (size_t)W + 8);
    long nnz = 0;
    for (int step = 0; step < nb; step++) {
        int k = -1, best = 1 << 30;
        for (int i = 0; i < nb; i++) if (alive[i] && deg[i] < best) { best = deg[i]; k = i; }
        c->perm[step] = k; c->iperm[k] = step; alive[k] = 0;
        memcpy(nk, A + (size_t)k * W, 8 * (size_t)W);
        colrows[step] = (int*)malloc(sizeof(int) * (best + 1));
        int m = 0;
        for (int w = 0; w < W; w++) {
            unsigned long long v = nk[w];
            while (v) { const int b = __builtin_ctzll(v); v &= v - 1; colrows[step][m++] = w * 64 + b; }
        }
        colcnt[step] = m; nnz += m;
        for (int a = 0; a < m; a++) {                      /* neighbours become a clique; k leaves the graph */
            const int i = colrows[step][a];
            unsigned long long* Ai = A + (size_t)i * W;
            int d = 0;
            for (int w = 0; w < W; w++) { Ai[w] |= nk[w]; }
            Ai[i >> 6] &= ~(1ull << (i & 63));
            Ai[k >> 6] &= ~(1ull << (k & 63));
            for (int w = 0; w < W; w++) d += popc64(Ai[w]);
            deg[i] = d;
        }
    }
    c->nnz = nnz;
    c->colptr = (int*)malloc(sizeof(int) * (nb + 2));
    c->rowidx = (int*)malloc(sizeof(int) * (nnz + 1));
    c->pos = (int*)malloc(sizeof(int) * ((size_t)nb * nb + 1));
    memset(c->pos, 0xff, sizeof(int) * ((size_t)nb * nb + 1));
    c->colptr[0] = 0;
    double fl = 0;
    for (int k = 0; k < nb; k++) {
        const int m = colcnt[k];
        int* r = c->rowidx + c->colptr[k];
        for (int a = 0; a < m; a++) r[a] = c->iperm[colrows[k][a]];
        for (int a = 1; a < m; a++) { const int v = r[a]; int b = a - 1; while (b >= 0 && r[b] > v) { r[b + 1] = r[b]; b--; } r[b + 1] = v; }
        for (int a = 0; a < m; a++) c->pos[(size_t)r[a] * nb + k] = c->colptr[k] + a;
        c->colptr[k + 1] = c->colptr[k] + m;
        fl += 2.0 * ((double)BB * BS / 6 + (double)m * BS * (BS * (BS + 1) / 2) + (double)m * (m + 1) / 2 * BB * BS);
        free(colrows[k]);
    }
    c->flops = fl;
    c->D = (double*)malloc(sizeof(double) * BB * (size_t)(nb + 1));
    c->L = (double*)malloc(sizeof(double) * BB * (size_t)(nnz + 1));
    free(colrows); free(colcnt); free(nk); free(deg); free(alive); free(A);
    return c;
}

orc_bchol* orc_bchol_new(int nb, const uint8_t* adj) { return orc_bchol_new_bs(nb, adj, 6); }

void orc_bchol_free(orc_bchol* c)
{
    if (!c) return;
    free(c->perm); free(c->iperm); free(c->colptr); free(c->rowidx); free(c->pos); free(c->D); free(c->L); free(c);
}

long orc_bchol_nnz(const orc_bchol* c) { return c->nnz; }
double orc_bchol_flops(const orc_bchol* c) { return c->flops; }

/* BS x BS lower Cholesky in place (row-major, upper part left untouched); 0 if not positive definite */
static int cholb(double* A, int BS)
{
    for (int j = 0; j < BS; j++) {
        double d = A[j * BS + j];
        for (int k = 0; k < j; k++) d -= A[j * BS + k] * A[j * BS + k];
        if (!(d > 0)) return 0;
        d = sqrt(d); A[j * BS + j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < BS; i++) {
            double v = A[i * BS + j];
            for (int k = 0; k < j; k++) v -= A[i * BS + k] * A[j * BS + k];
            A[i * BS + j] = v * id;
        }
    }
    return 1;
}

/* Numeric factorisation.  idx[f1*nb+f2] (f1 <= f2, original order) -> slot of the upper block in blk, or -1;
 * blk[slot] = Hschur(f1, f2) row-major 6x6.  Returns 0 when a pivot block is not positive definite. */
int orc_bchol_factor(orc_bchol* c, const int32_t* idx, const double* blk)
{
    const int nb = c->nb, BS = c->bs, BB = BS * BS;
    memset(c->D, 0, sizeof(double) * BB * (size_t)nb);
    memset(c->L, 0, sizeof(double) * BB * (size_t)c->nnz);
    for (int f1 = 0; f1 < nb; f1++)
        for (int f2 = f1; f2 < nb; f2++) {
            const int s = idx[(size_t)f1 * nb + f2];
            if (s < 0) continue;
            const double* B = blk + BB * (size_t)s;            /* block (f1, f2) */
            const int p1 = c->iperm[f1], p2 = c->iperm[f2];
            if (f1 == f2) { memcpy(c->D + BB * (size_t)p1, B, BB * sizeof(double)); continue; }
            if (p1 > p2) {                                      /* lower block (p1, p2) = B */
                const int t = c->pos[(size_t)p1 * nb + p2];
                if (t < 0) return -1;
                memcpy(c->L + BB * (size_t)t, B, BB * sizeof(double));
            } else {                                            /* lower block (p2, p1) = B^T */
                const int t = c->pos[(size_t)p2 * nb + p1];
                if (t < 0) return -1;
                double* T = c->L + BB * (size_t)t;
                for (int i = 0; i < BS; i++) for (int j = 0; j < BS; j++) T[i * BS + j] = B[j * BS + i];
            }
        }
    for (int k = 0; k < nb; k++) {
        double* Lkk = c->D + BB * (size_t)k;
        if (!cholb(Lkk, BS)) return 0;
        const int c0 = c->colptr[k], c1 = c->colptr[k + 1];
        for (int a = c0; a < c1; a++) {                         /* L_ik = A_ik Lkk^-T : forward substitution per row */
            double* X = c->L + BB * (size_t)a;
            for (int r = 0; r < BS; r++)
                for (int j = 0; j < BS; j++) {
                    double v = X[r * BS + j];
                    for (int q = 0; q < j; q++) v -= X[r * BS + q] * Lkk[j * BS + q];
                    X[r * BS + j] = v / Lkk[j * BS + j];
                }
        }
        for (int a = c0; a < c1; a++) {
            const int i = c->rowidx[a];
            const double* Li = c->L + BB * (size_t)a;
            double* Dii = c->D + BB * (size_t)i;
            for (int r = 0; r < BS; r++)
                for (int s2 = 0; s2 <= r; s2++) {
                    double v = 0;
                    for (int q = 0; q < BS; q++) v += Li[r * BS + q] * Li[s2 * BS + q];
                    Dii[r * BS + s2] -= v;
                }
            const int* prow = c->pos + (size_t)i * nb;
            for (int b = c0; b < a; b++) {                      /* rows ascend: rowidx[b] < i */
                const double* Lj = c->L + BB * (size_t)b;
                double* T = c->L + BB * (size_t)prow[c->rowidx[b]];
                for (int r = 0; r < BS; r++)
                    for (int s2 = 0; s2 < BS; s2++) {
                        double v = 0;
                        for (int q = 0; q < BS; q++) v += Li[r * BS + q] * Lj[s2 * BS + q];
                        T[r * BS + s2] -= v;
                    }
            }
        }
    }
    return 1;
}

/* x = Hschur^-1 b (original order, BS nb doubles each) */
void orc_bchol_solve(const orc_bchol* c, const double* b, double* x)
{
    const int nb = c->nb, BS = c->bs, BB = BS * BS;
    double* y = (double*)malloc(sizeof(double) * BS * (size_t)(nb + 1));
    for (int k = 0; k < nb; k++) memcpy(y + BS * k, b + BS * (size_t)c->perm[k], BS * sizeof(double));
    for (int k = 0; k < nb; k++) {                              /* L y' = y */
        const double* Lkk = c->D + BB * (size_t)k;
        double* yk = y + BS * k;
        for (int j = 0; j < BS; j++) { double v = yk[j]; for (int q = 0; q < j; q++) v -= Lkk[j * BS + q] * yk[q]; yk[j] = v / Lkk[j * BS + j]; }
        for (int a = c->colptr[k]; a < c->colptr[k + 1]; a++) {
            const double* Li = c->L + BB * (size_t)a; double* yi = y + BS * c->rowidx[a];
            for (int r = 0; r < BS; r++) { double v = 0; for (int q = 0; q < BS; q++) v += Li[r * BS + q] * yk[q]; yi[r] -= v; }
        }
    }
    for (int k = nb - 1; k >= 0; k--) {                         /* L^T x' = y' */
        const double* Lkk = c->D + BB * (size_t)k;
        double* yk = y + BS * k;
        for (int a = c->colptr[k]; a < c->colptr[k + 1]; a++) {
            const double* Li = c->L + BB * (size_t)a; const double* yi = y + BS * c->rowidx[a];
            for (int q = 0; q < BS; q++) { double v = 0; for (int r = 0; r < BS; r++) v += Li[r * BS + q] * yi[r]; yk[q] -= v; }
        }
        for (int j = BS - 1; j >= 0; j--) { double v = yk[j]; for (int q = j + 1; q < BS; q++) v -= Lkk[q * BS + j] * yk[q]; yk[j] = v / Lkk[j * BS + j]; }
    }
    for (int k = 0; k < nb; k++) memcpy(x + BS * (size_t)c->perm[k], y + BS * k, BS * sizeof(double));
    free(y);
}
