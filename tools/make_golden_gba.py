#!/usr/bin/env python3
"""tests/golden/gba_config5.npz: BASELINE config 5 (2000 keyframes / 200k points / 3 agents, seed 0xBA000005)
after 5 Levenberg iterations of the CPU oracle, whose reduced solve at this size is the block-sparse Cholesky
of oracle/bchol_oracle.c (sparse + exact like linear_solver_eigen.h:106-136; the dense Cholesky cannot run it).

Before writing, the oracle's first linear solve at full size is checked against an independent statement of
the same normal equations in numpy/scipy (vectorised Jacobians, scipy.sparse products): the relative residual
of (J^T W J + lambda I) x = J^T W r with the oracle's x is stored in the file and must be < 1e-9.

Nothing from the reference tree is read; the graph is motioncheck_ccm_slam_amd/synth.py's deterministic generator.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from motioncheck_ccm_slam_amd import synth
from oracle import oracle_py as O

HUBER = float(np.float32(np.sqrt(5.99)))            # src/Optimizer.cpp:712
ITERS = 5
POINT_STRIDE = 40


def quat_to_R(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def normal_equations(g, huber):
    """J^T W J and J^T W r of the robustified problem at g's state (types_six_dof_expmap.cpp:103-139,
    base_binary_edge.hpp:55-120 with Huber rho'), unknowns = [6 per free keyframe | 3 per landmark]."""
    ep, el = g["edge_pose"], g["edge_point"]
    R = quat_to_R(g["poses"][:, :4])[ep]; t = g["poses"][ep, 4:]
    K = g["intr"][ep]; p = g["points"][el]
    pc = np.einsum("eij,ej->ei", R, p) + t
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    fx, fy, cx, cy = K[:, 0], K[:, 1], K[:, 2], K[:, 3]
    err = np.stack([g["obs"][:, 0] - (x / z * fx + cx), g["obs"][:, 1] - (y / z * fy + cy)], 1)
    E = len(ep)
    tm = np.zeros((E, 2, 3))
    tm[:, 0, 0] = fx; tm[:, 0, 2] = -x / z * fx; tm[:, 1, 1] = fy; tm[:, 1, 2] = -y / z * fy
    A = -(1.0 / z)[:, None, None] * np.einsum("eij,ejk->eik", tm, R)              # d err / d point
    z2 = z * z
    B = np.zeros((E, 2, 6))
    B[:, 0, 0] = x * y / z2 * fx; B[:, 0, 1] = -(1 + x * x / z2) * fx; B[:, 0, 2] = y / z * fx
    B[:, 0, 3] = -1.0 / z * fx; B[:, 0, 5] = x / z2 * fx
    B[:, 1, 0] = (1 + y * y / z2) * fy; B[:, 1, 1] = -x * y / z2 * fy; B[:, 1, 2] = -x / z * fy
    B[:, 1, 4] = -1.0 / z * fy; B[:, 1, 5] = y / z2 * fy
    om = g["info"]
    c2 = om * (err ** 2).sum(1)
    rho1 = np.where(c2 <= huber * huber, 1.0, huber / np.sqrt(np.maximum(c2, 1e-300)))
    w = om * rho1
    free_of = np.cumsum(g["fixed"] == 0) - 1
    free_of[g["fixed"] != 0] = -1
    nf = int((g["fixed"] == 0).sum()); L = len(g["points"])
    fe = free_of[ep]
    rows = np.repeat(np.arange(2 * E).reshape(E, 2), 3, axis=1).reshape(E, 2, 3)
    colsA = 6 * nf + 3 * el[:, None, None] + np.arange(3)[None, None, :] + np.zeros((E, 2, 1), np.int64)
    m = fe >= 0
    rowsB = np.repeat(np.arange(2 * E).reshape(E, 2), 6, axis=1).reshape(E, 2, 6)[m]
    colsB = (6 * fe[m, None, None] + np.arange(6)[None, None, :] + np.zeros((m.sum(), 2, 1), np.int64))
    J = sp.csr_matrix((np.concatenate([A.ravel(), B[m].ravel()]),
                       (np.concatenate([rows.ravel(), rowsB.ravel()]), np.concatenate([colsA.ravel(), colsB.ravel()]))),
                      shape=(2 * E, 6 * nf + 3 * L))
    W = sp.diags(np.repeat(w, 2))
    H = (J.T @ W @ J).tocsr()
    b = -(J.T @ (np.repeat(w, 2) * err.ravel()))
    return H, b, nf


def main():
    t0 = time.time()
    g = synth.gba_graph()
    print("graph: %d edges, %.1f s" % (len(g["edge_pose"]), time.time() - t0))
    lam = 10.0
    t0 = time.time()
    xp, xl, st = O.ba_solve_once(g, HUBER, lam, 0)
    print("oracle linear solve: %.1f s" % (time.time() - t0), st)
    H, b, nf = normal_equations(g, HUBER)
    x = np.concatenate([xp.ravel(), xl.ravel()])
    res = H @ x + lam * x - b
    rel = float(np.linalg.norm(res) / np.linalg.norm(b))
    print("independent normal equations: relative residual of the oracle's increment = %.3e" % rel)
    assert rel < 1e-9, rel
    t0 = time.time()
    r = O.ba_solve(g, ITERS, HUBER)
    dt = time.time() - t0
    print("oracle %d LM iterations: %.1f s (%.3f it/s)" % (ITERS, dt, r["iterations_done"] / dt),
          {k: v for k, v in r.items() if not hasattr(v, "shape")})
    # the reference's own call: optimize(20) (cslam/conf/config.yaml:129); g2o's stop criterion (three iterations in a row gaining
    # less than 0.1 % of chi2, optimization_algorithm_levenberg.cpp:154-161) ends it earlier on this graph
    t0 = time.time()
    r20 = O.ba_solve(g, 20, HUBER)
    print("oracle optimize(20): %.1f s" % (time.time() - t0), {k: v for k, v in r20.items() if not hasattr(v, "shape")})
    out = os.path.join(ROOT, "tests", "golden", "gba_config5.npz")
    np.savez_compressed(out, poses20=r20["poses"], points20_sub=r20["points"][::POINT_STRIDE], chi2_20=np.array([r20["chi2_initial"], r20["chi2_final"]]),
                        iterations20=np.array([r20["iterations_done"], r20["trials"]]), lambda20=np.array(r20["lambda_final"]),
                        poses=r["poses"], points_sub=r["points"][::POINT_STRIDE], point_stride=np.array(POINT_STRIDE),
                        points_sum=r["points"].sum(0), chi2=np.array([r["chi2_initial"], r["chi2_final"]]),
                        iterations=np.array([r["iterations_done"], r["trials"]]), lambda_final=np.array(r["lambda_final"]),
                        lin_check=np.array([lam, rel]), factor_blocks=np.array(st["factor_blocks"]),
                        n_edges=np.array(len(g["edge_pose"])))
    print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
