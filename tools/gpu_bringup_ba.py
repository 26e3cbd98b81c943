#!/usr/bin/env python3
"""BA on the GPU vs the CPU oracle (run on a GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer, pose_delta
from oracle import oracle_py as O

def show(tag, r):
    print(tag, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items() if k not in ("poses", "points", "outlier")})

ok = True
# config 4: local BA, two-stage schedule
g = synth.local_ba_graph()
t = time.time(); r = Optimizer.LocalBundleAdjustmentClient(g); t1 = time.time() - t
t = time.time(); r = Optimizer.LocalBundleAdjustmentClient(g); t2 = time.time() - t
ref = O.ba_solve(g, 5, np.sqrt(5.991), 10)
show("gpu", r); show("cpu", {k: v for k, v in ref.items()})
d = pose_delta(r["poses"], ref["poses"]).max(); dp = np.abs(r["points"] - ref["points"]).max()
print("local BA: first call %.3fs second %.3fs  max pose delta %.3e  max point delta %.3e  outliers gpu %d cpu %d same %s" %
      (t1, t2, d, dp, r["outlier"].sum(), ref["outlier"].sum(), (r["outlier"] == ref["outlier"]).all()))
ok &= d < 1e-5
# single stage, robust, 20 its (client GBA shape)
g = synth.gba_graph(n_kf=60, n_points=3000, n_agents=3, seed=7)
r = Optimizer.MapFusionGBA(g, 20); ref = O.ba_solve(g, 20, np.sqrt(5.99))
show("gpu", r); show("cpu", ref)
d = pose_delta(r["poses"], ref["poses"]).max()
print("small GBA max pose delta %.3e" % d); ok &= d < 1e-5
# medium
g = synth.gba_graph(n_kf=300, n_points=30000, n_agents=3, seed=8)
t = time.time(); r = Optimizer.MapFusionGBA(g, 5); tg = time.time() - t
t = time.time(); ref = O.ba_solve(g, 5, np.sqrt(5.99)); tc = time.time() - t
show("gpu", r); show("cpu", ref)
d = pose_delta(r["poses"], ref["poses"]).max()
print("medium GBA (300 KF) gpu %.3fs cpu %.3fs max pose delta %.3e" % (tg, tc, d)); ok &= d < 1e-5
if len(sys.argv) > 1:
    g = synth.gba_graph()
    t = time.time(); r = Optimizer.MapFusionGBA(g, 3); tg = time.time() - t
    show("full GBA gpu %.3fs" % tg, r)
print("BA", "OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
