#!/usr/bin/env python3
"""Config 5 (2000 KF / 200k points) global BA alone, for rocprofv3: one warm-up call, then `--calls` timed calls.
usage: python3 tools/bench_gba.py [--iters 5] [--calls 1] [--pcg-tol 0]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--calls", type=int, default=1)
ap.add_argument("--pcg-tol", type=float, default=0.0)
a = ap.parse_args()
ctx = _lib.Context(0)
g = synth.gba_graph()
Optimizer.MapFusionGBA(g, 1, ctx=ctx, pcg_tol=a.pcg_tol)
for _ in range(a.calls):
    t = time.perf_counter()
    r = Optimizer.MapFusionGBA(g, a.iters, ctx=ctx, pcg_tol=a.pcg_tol)
    t = time.perf_counter() - t
    lm = r["t_linearize"] + r["t_schur"] + r["t_solve"] + r["t_update"]
    print(json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items() if not hasattr(v, "shape")}
                     | {"call_s": round(t, 4), "lm_s": round(lm, 4), "it_per_s_lm": round(r["iterations_done"] / lm, 2),
                        "it_per_s_call": round(r["iterations_done"] / t, 2), "edges": len(g["edge_pose"])}))
ctx.close()
