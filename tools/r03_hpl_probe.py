"""Perturbed graphs, to be run with CCM_BA_TEST_REJECT_AT=N (a trial rejected after the first iteration: the path that rebuilds Hpl): prints iterations / trials / chi2 and a
digest of the result, for a run with CCM_BA_KEEP_HPL=1 and one without to be compared."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer
ctx = _lib.Context(0)
for kf, pts, scale in ((60, 3000, 1.0), (60, 3000, 8.0), (60, 3000, 20.0), (240, 20000, 10.0), (240, 20000, 25.0)):
    g = dict(synth.gba_graph(n_kf=kf, n_points=pts, n_agents=3, seed=kf))
    rng = np.random.default_rng(kf)
    p = np.array(g["poses"], dtype=np.float64)
    p[:, 4:7] += 0.03 * (scale - 1.0) * rng.standard_normal((len(p), 3))
    g["poses"] = p
    x = np.array(g["points"], dtype=np.float64) + 0.03 * (scale - 1.0) * rng.standard_normal(np.shape(g["points"]))
    g["points"] = x
    r = Optimizer.MapFusionGBA(g, 12, ctx=ctx)
    h = hashlib.sha1(np.ascontiguousarray(r["poses"]).tobytes() + np.ascontiguousarray(r["points"]).tobytes()).hexdigest()[:12]
    print(kf, scale, r["iterations_done"], r["trials"], "%.6f" % r["chi2_final"], h, " ".join("%.12g" % v for v in np.asarray(r["poses"])[kf // 2]))
