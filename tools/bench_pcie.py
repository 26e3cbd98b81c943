#!/usr/bin/env python3
"""The drop-in call with host buffers (ccm_orb_extract: 256 frames of 752x480 in, keypoints + descriptors out), median of `--calls`:
pageable buffers, page-locked frames, page-locked frames and result pools.  Environment: CCM_ORB_CHUNK (frames per chunk),
CCM_ORB_UPLOAD_2D=1 (the strided upload of rounds 1-2).   usage: python3 tools/bench_pcie.py [--calls 7]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.orb import ORBextractor

ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=7); a = ap.parse_args()
ctx = _lib.Context(0)
ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
frames = synth.frames(0, 256)
def med(fn):
    fn(); ts = []
    for _ in range(a.calls):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(float(np.median(ts)) * 1e3, 3)
r = {"chunk": os.environ.get("CCM_ORB_CHUNK", "64"), "upload_2d": os.environ.get("CCM_ORB_UPLOAD_2D", "0")}
r["pageable_ms"] = med(lambda: ex.extract_batch(frames))
ctx.host_register(frames)
r["registered_input_ms"] = med(lambda: ex.extract_batch(frames))
outs = ex.extract_batch(frames)
for x in outs: ctx.host_register(x)
r["registered_input_and_output_ms"] = med(lambda: ex.extract_batch(frames, out=outs))
one = frames[0:1].copy()
r["single_frame_ms"] = med(lambda: ex.extract_batch(one))
print(json.dumps(r))
ctx.close()
