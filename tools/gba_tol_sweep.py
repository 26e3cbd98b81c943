#!/usr/bin/env python3
"""GBA at config 5 with the PCG tolerance given by CCM_PCG_TOL: prints iterations, time and the result's checksum so that
runs with different tolerances can be compared (tools only; not part of the product)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer

ctx = _lib.Context(0)
g = synth.gba_graph()
Optimizer.MapFusionGBA(synth.gba_graph(n_kf=300, n_points=30000, seed=8), 2, ctx=ctx)      # warm-up
t0 = time.perf_counter()
r = Optimizer.MapFusionGBA(g, 5, ctx=ctx)
dt = time.perf_counter() - t0
out = os.environ.get("GBA_OUT")
if out:
    np.save(out, r["poses"])
ref = os.environ.get("GBA_REF")
diff = None
if ref and os.path.exists(ref):
    diff = float(np.abs(np.load(ref) - r["poses"]).max())
print(json.dumps({"tol": os.environ.get("CCM_PCG_TOL", "1e-13"), "pcg_iterations": r["pcg_iterations"], "trials": r["trials"],
                  "iterations": r["iterations_done"], "t_solve": r["t_solve"], "call_s": dt, "chi2_final": r["chi2_final"], "max_pose_diff_vs_ref": diff}))
