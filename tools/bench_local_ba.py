#!/usr/bin/env python3
"""BASELINE config 4 (local BA: 20 free + 10 fixed keyframes, 5000 points) alone, for rocprofv3 / latency work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer
ctx = _lib.Context(0); g = synth.local_ba_graph()
Optimizer.LocalBundleAdjustmentClient(g, ctx=ctx)
t = []
for _ in range(10):
    t0 = time.perf_counter(); r = Optimizer.LocalBundleAdjustmentClient(g, ctx=ctx); t.append(time.perf_counter() - t0)
print("local BA: %.3f ms per call (median of 10), %d iterations, %d trials; timers lin %.2f schur %.2f solve %.2f update %.2f ms" % (
    np.median(t) * 1e3, r["iterations_done"], r["trials"], r["t_linearize"] * 1e3, r["t_schur"] * 1e3, r["t_solve"] * 1e3, r["t_update"] * 1e3))
