import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer
ctx = _lib.Context(0)
g = synth.gba_graph()
for k in ("edge_pose", "edge_point", "obs", "info"):
    ctx.host_register(g[k])
Optimizer.MapFusionGBA(g, 1, ctx=ctx)
Optimizer.MapFusionGBA(g, 20, ctx=ctx)
os.environ["CCM_DEBUG"] = "1"
t = time.perf_counter(); r = Optimizer.MapFusionGBA(g, 20, ctx=ctx); print("call", time.perf_counter() - t)
