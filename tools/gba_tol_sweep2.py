import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer, pose_delta
z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/gba_config5.npz"))
ctx = _lib.Context(0); g = synth.gba_graph()
Optimizer.MapFusionGBA(g, 1, ctx=ctx)
for tol in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
    t = time.perf_counter(); r = Optimizer.MapFusionGBA(g, 20, ctx=ctx, pcg_tol=tol); t = time.perf_counter() - t
    d = pose_delta(r["poses"], z["poses20"]).max(); dp = np.abs(r["points"][::40] - z["points20_sub"]).max()
    print("tol %.0e: its %d trials %d pcg %d solve %.1f ms call %.1f ms | pose dev %.2e point dev %.2e chi2 rel %.2e" % (
        tol, r["iterations_done"], r["trials"], r["pcg_iterations"], r["t_solve"] * 1e3, t * 1e3, d, dp, abs(r["chi2_final"] / z["chi2_20"][1] - 1)))
