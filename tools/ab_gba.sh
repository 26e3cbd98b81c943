#!/bin/bash
# Run on the GPU box from the repo root: tools/ab_gba.sh ROUNDS ab/libA.so ab/libB.so ...
# A/B timing of builds of libccm_hot.so on ONE box: config 5, optimize(20), page-locked edge arrays (tools/gba_laps.py without the debug
# output) and a BaWorkspace (AB_NO_WORKSPACE=1: without), five timed calls per library and round, the libraries interleaved.
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    CCM_HOT_LIB=$PWD/$lib timeout -k 10 120 python3 - <<PY || exit 1
import os, sys, time
sys.path.insert(0, os.getcwd())
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer, BaWorkspace
ctx = _lib.Context(0)
g = synth.gba_graph()
for k in ("edge_pose", "edge_point", "obs", "info"):
    ctx.host_register(g[k])
ws = BaWorkspace(ctx, len(g["poses"]), len(g["points"])) if os.environ.get("AB_NO_WORKSPACE") != "1" else None
Optimizer.MapFusionGBA(g, 1, ctx=ctx, workspace=ws); Optimizer.MapFusionGBA(g, 20, ctx=ctx, workspace=ws)
ts = []
for _ in range(5):
    t = time.perf_counter(); r = Optimizer.MapFusionGBA(g, 20, ctx=ctx, workspace=ws); ts.append(time.perf_counter() - t)
ts.sort()
print("$lib round $r: call median %.3f ms min %.3f  lin %.2f schur %.2f solve %.2f update %.2f  chi2 %.5f its %d" % (
    1e3 * ts[2], 1e3 * ts[0], 1e3 * r["t_linearize"], 1e3 * r["t_schur"], 1e3 * r["t_solve"], 1e3 * r["t_update"], r["chi2_final"], r["pcg_iterations"]))
PY
  done
done
