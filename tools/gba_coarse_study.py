#!/usr/bin/env python3
"""CPU study of the PCG preconditioner of the global BA (config 5) at a late LM trial (small lambda): iteration counts of
  (a) the shipped two-level preconditioner (48x48 cluster inverses of 8 keyframes + 6 rigid unknowns per 16-keyframe aggregate),
  (b) the same plus ONE global scale mode in the coarse space (a monocular map with one fixed keyframe has a free scale:
      increment (0, eps * t_i) per keyframe leaves every reprojection unchanged),
  (c) the same plus one scale unknown per aggregate (7 per aggregate).
Runs the oracle for `--iters` LM iterations first to reach the state.  usage: python3 tools/gba_coarse_study.py [--iters 8] [--lam 0.042]"""
import argparse, os, sys, time
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from motioncheck_ccm_slam_amd import synth
from oracle import oracle_py as O

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=8); ap.add_argument("--lam", type=float, default=0.042)
ap.add_argument("--kf", type=int, default=2000); ap.add_argument("--tol", type=float, default=1e-6); a = ap.parse_args()
g = synth.gba_graph() if a.kf == 2000 else synth.gba_graph(n_kf=a.kf, n_points=100 * a.kf)
t = time.time()
if a.iters > 0:
    r = O.ba_solve(g, a.iters, 0.0)          # MapFusionGBA: no robust kernel
    g = dict(g); g["poses"] = r["poses"]; g["points"] = r["points"]
    print("oracle: %d iterations, chi2 %.1f -> %.1f, lambda %.4g, %.1f s" % (r["iterations_done"], r["chi2_initial"], r["chi2_final"], r["lambda_final"], time.time() - t))
H, b, fi = O.ba_reduced_system(g, 0.0, a.lam)
n = len(b); nf = n // 6
Hs = sp.csr_matrix(H); del H
Hs = sp.triu(Hs) + sp.triu(Hs, 1).T          # the oracle fills the upper block triangle
Hs = Hs.tocsr()
print("reduced system: n = %d, nnz = %d" % (n, Hs.nnz))
poses = np.asarray(g["poses"]); free = np.where(np.asarray(g["fixed"]) == 0)[0]
CL, AGG = int(os.environ.get("STUDY_CL", "8")), 2
ncl = (nf + CL - 1) // CL; nagg = (ncl + AGG - 1) // AGG
# cluster level
Minv = []
for c in range(ncl):
    i0, i1 = 6 * c * CL, min(n, 6 * (c + 1) * CL)
    Minv.append(np.linalg.inv(Hs[i0:i1, i0:i1].toarray()))
def cluster_apply(r):
    z = np.empty_like(r)
    for c in range(ncl):
        i0, i1 = 6 * c * CL, min(n, 6 * (c + 1) * CL)
        z[i0:i1] = Minv[c] @ r[i0:i1]
    return z
# coarse spaces: columns of R^T
def rigid_cols():
    rows, cols = [], []
    for f in range(nf):
        ag = f // (CL * AGG)
        for d in range(6): rows.append(6 * f + d); cols.append(6 * ag + d)
    return sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, 6 * nagg))
def scale_vec():
    w = np.zeros(n)
    for f in range(nf): w[6 * f + 3:6 * f + 6] = poses[free[f], 4:7]      # increment (omega, upsilon) = (0, t_i): t -> (1 + eps) t
    return w
def make_coarse(Rt):
    Ac = (Rt.T @ (Hs @ Rt))
    Ac = Ac.toarray() if sp.issparse(Ac) else np.asarray(Ac)
    Aci = np.linalg.inv(Ac)
    return lambda r: Rt @ (Aci @ (Rt.T @ r))
Rt6 = rigid_cols()
w = scale_vec()
Rt_glob = sp.hstack([Rt6, sp.csr_matrix(w[:, None])]).tocsr()
rows, cols, vals = [], [], []
for f in range(nf):
    ag = f // (CL * AGG)
    for d in range(3): rows.append(6 * f + 3 + d); cols.append(ag); vals.append(w[6 * f + 3 + d])
Rt7 = sp.hstack([Rt6, sp.csr_matrix((vals, (rows, cols)), shape=(n, nagg))]).tocsr()
def pcg(prec, tol=a.tol, maxit=3000):
    x = np.zeros(n); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; bb = b @ b
    for it in range(1, maxit + 1):
        Ap = Hs @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        if r @ r <= tol * tol * bb: return it, x
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit, x
def agg_cols(per, with_scale):
    na = (nf + per - 1) // per
    rows, cols, vals = [], [], []
    for f in range(nf):
        ag = f // per
        for d in range(6): rows.append(6 * f + d); cols.append(6 * ag + d); vals.append(1.0)
    R = sp.csr_matrix((vals, (rows, cols)), shape=(n, 6 * na))
    if with_scale:
        rows, cols, vals = [], [], []
        for f in range(nf):
            for d in range(3): rows.append(6 * f + 3 + d); cols.append(f // per); vals.append(w[6 * f + 3 + d])
        R = sp.hstack([R, sp.csr_matrix((vals, (rows, cols)), shape=(n, na))]).tocsr()
    return R
def hat_cols(per, with_scale):
    """piecewise-linear (hat) interpolation between aggregate centres instead of piecewise-constant aggregates"""
    na = (nf + per - 1) // per
    rows, cols, vals = [], [], []
    srows, scols, svals = [], [], []
    for f in range(nf):
        x = (f + 0.5) / per - 0.5                      # position in units of aggregates, centres at integers
        I = int(np.floor(x)); al = x - I
        for (ag, wt) in ((I, 1 - al), (I + 1, al)):
            ag = min(max(ag, 0), na - 1)
            if wt == 0: continue
            for d in range(6): rows.append(6 * f + d); cols.append(6 * ag + d); vals.append(wt)
            for d in range(3): srows.append(6 * f + 3 + d); scols.append(ag); svals.append(wt * w[6 * f + 3 + d])
    R = sp.csr_matrix((vals, (rows, cols)), shape=(n, 6 * na))
    if with_scale: R = sp.hstack([R, sp.csr_matrix((svals, (srows, scols)), shape=(n, na))]).tocsr()
    return R
def hat_cols_centred(per):
    """hat functions + a scale unknown per aggregate whose column is (t_i - c_I): c_I = mean translation of the aggregate's own keyframes"""
    na = (nf + per - 1) // per
    tt = poses[free, 4:7]
    cen = np.array([tt[I * per:min(nf, (I + 1) * per)].mean(0) for I in range(na)])
    rows, cols, vals = [], [], []
    for f in range(nf):
        x = (f + 0.5) / per - 0.5
        I = int(np.floor(x)); al = x - I
        i0, i1, w0, w1 = max(I, 0), min(I + 1, na - 1), 1 - al, al
        i0 = min(i0, na - 1)
        if i0 == i1: w0, w1 = 1.0, 0.0
        for (ag, wt) in ((i0, w0), (i1, w1)):
            if wt == 0: continue
            for d in range(6): rows.append(6 * f + d); cols.append(7 * ag + d); vals.append(wt)
            for d in range(3): rows.append(6 * f + 3 + d); cols.append(7 * ag + 6); vals.append(wt * (tt[f, d] - cen[ag, d]))
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, 7 * na))
extra = [("hat functions over 32 keyframes + scale, centred", hat_cols_centred(32)), ("hat functions over 64 keyframes + scale, centred", hat_cols_centred(64)),
         ("hat functions over 16 keyframes", hat_cols(16, False)), ("  + scale", hat_cols(16, True)), ("  + scale, centred per aggregate", hat_cols_centred(16)),
         ("hat functions over 8 keyframes", hat_cols(8, False)), ("  + scale", hat_cols(8, True))]
for per in (8, 4, 2):
    extra.append(("rigid aggregates of %d keyframes (%d coarse unknowns)" % (per, 6 * ((nf + per - 1) // per)), agg_cols(per, False)))
    extra.append(("  + scale per aggregate", agg_cols(per, True)))
for name, Rt in [("cluster only", None), ("cluster + rigid aggregates (shipped)", Rt6), ("+ one global scale mode", Rt_glob), ("+ a scale unknown per aggregate", Rt7)] + extra:
    if Rt is None: prec = cluster_apply
    else:
        co = make_coarse(Rt); prec = (lambda co: (lambda r: cluster_apply(r) + co(r)))(co)
    t = time.time(); it, x = pcg(prec); print("%-62s %5d iterations  (%.1f s)" % (name, it, time.time() - t))
