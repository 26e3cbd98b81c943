#!/usr/bin/env python3
"""The windowed matchers with device-side acceptance alone, for rocprofv3: SearchByProjection(KF, Scw) with 20,000 map points against one
keyframe, SearchByProjection(Frame, map points) with 2000.   usage: python3 tools/bench_window.py [--reps 5]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.matcher import FrameGridView, ORBmatcher
from motioncheck_ccm_slam_amd.orb import ORBextractor

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=5); a = ap.parse_args()
ctx = _lib.Context(0); ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
kps, desc = ex(synth.frame(0)); fr = FrameGridView(kps["x"], kps["y"], kps["octave"], desc); sf = ex.GetScaleFactors(); n = len(fr.kx)
m = ORBmatcher(0.8, ctx=ctx)

def case(nmp, seed):
    rng = np.random.default_rng(seed); src = rng.integers(0, n, nmp)
    mp = desc[src] ^ np.packbits(rng.random((nmp, 256)) < 0.05, axis=1, bitorder="little")
    px = (fr.kx[src] + rng.normal(0, 2.0, nmp)).astype("f4"); py = (fr.ky[src] + rng.normal(0, 2.0, nmp)).astype("f4")
    lvl = np.clip(fr.oct[src] + rng.integers(0, 2, nmp), 0, 7)
    return rng, mp, px, py, lvl

rng, mp, px, py, lvl = case(20000, 5)
v = np.ones(20000, np.uint8); ob = (rng.random(20000) < 0.1).astype(np.uint8); mt = (rng.random(n) < 0.1).astype(np.uint8)
f1 = lambda: m.SearchByProjectionSim3(fr, sf, v, px, py, lvl, mp, ob, mt, 8.0)
rng2, mp2, px2, py2, lvl2 = case(2000, 0)
ones = np.ones(2000, np.uint8); occ = np.zeros(n, np.uint8); vc = rng2.uniform(0.99, 1, 2000).astype("f4")
f2 = lambda: m.SearchByProjection(fr, sf, ones, lvl2, vc, px2, py2, mp2, ones, occ, 3.0)
for name, f in (("sim3_20k", f1), ("projection_2k", f2)):
    f(); t = time.perf_counter()
    for _ in range(a.reps): f()
    print("%s: %.3f ms per call" % (name, (time.perf_counter() - t) / a.reps * 1e3))
ctx.close()
