#!/usr/bin/env python3
"""One 752x480 frame through ccm_orb_extract with host buffers (what Tracking calls per frame), for rocprofv3 / latency work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.orb import ORBextractor
ctx = _lib.Context(0); ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
img = synth.frames(0, 1)
ex.extract_batch(img)
t = []
for _ in range(200):
    t0 = time.perf_counter(); k, d, c = ex.extract_batch(img); t.append(time.perf_counter() - t0)
t = np.asarray(t) * 1e3
print("single frame: median %.4f ms, p10 %.4f, p90 %.4f, %d keypoints" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90), int(c[0])))
