import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer
ctx = _lib.Context(0)
gl = synth.local_ba_graph()
Optimizer.LocalBundleAdjustmentClient(gl, ctx=ctx)
for rep in range(3):
    ts = []
    for _ in range(20):
        t = time.perf_counter(); r = Optimizer.LocalBundleAdjustmentClient(gl, ctx=ctx); ts.append(time.perf_counter() - t)
    print("local BA ms: median %.3f min %.3f" % (np.median(ts) * 1e3, min(ts) * 1e3))
