#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_pp_stamps.sh -- phases of one workgroup of k_ppcg_prec (diagnostic build -DPP_STAMPS)
O=gpurun_out; mkdir -p $O
touch motioncheck_ccm_slam_amd/csrc/ba_sparse.hip
make -s -C motioncheck_ccm_slam_amd/csrc EXTRA="-DPP_STAMPS" > $O/pp_stamps.build 2>&1 || { tail -5 $O/pp_stamps.build; exit 1; }
timeout -k 10 200 python3 - <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer
ctx = _lib.Context(0); g = synth.gba_graph()
Optimizer.MapFusionGBA(g, 20, ctx=ctx)
lib = _lib.load(); lib.ccm_debug_pp_stamps.argtypes = [C.c_void_p]
b = np.zeros(32, np.uint64); assert lib.ccm_debug_pp_stamps(b.ctypes.data) == 0
t = b.astype(np.int64)
print("workgroup 5 of the last k_ppcg_prec: start -> own loads consumed / partial sums stored %d; -> barrier 1 %d; -> quarters summed, cluster products, barrier 2 %d; -> coarse rows times P^T w, barrier 3 %d; -> result written %d; total %d"
      % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
print("scalar workgroup: %d" % (t[17] - t[16]))
PY
touch motioncheck_ccm_slam_amd/csrc/ba_sparse.hip; make -s -C motioncheck_ccm_slam_amd/csrc > /dev/null 2>&1
