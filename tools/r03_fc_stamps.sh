#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_fc_stamps.sh TAG -- where does a band of k_fast_cells spend its ~14 us?
# Diagnostic build (-DFC_STAMPS): s_memtime of wave 0 at the phase boundaries of every workgroup of one 256-frame launch.
tag=$1; O=gpurun_out; mkdir -p $O
touch motioncheck_ccm_slam_amd/csrc/orb_kernels.hip
make -s -C motioncheck_ccm_slam_amd/csrc EXTRA="-DFC_STAMPS" > $O/${tag}_stamps.build 2>&1 || { tail -5 $O/${tag}_stamps.build; exit 1; }
timeout -k 10 200 python3 - > $O/${tag}_fc_stamps.txt 2>&1 <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import torch
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.orb import ORBextractor
ctx = _lib.Context(0); ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
frames = torch.from_numpy(synth.frames(0, 256)).cuda()
for _ in range(3):
    ex.extract_dev(frames.data_ptr(), 752, 480, 752, 752 * 480, 256); ctx.sync()
lib = _lib.load()
n = 1 << 17
buf = np.zeros((n, 8), np.uint64)
lib.ccm_debug_fc_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.ccm_debug_fc_stamps(buf.ctypes.data, n) == 0
ok = buf[:, 4] > 0
b = buf[ok].astype(np.int64)
print("workgroups with stamps:", int(ok.sum()))
names = ["start -> staged (barrier 1)", "staged -> every pixel scored (wave-local rejection + scoring, barrier 2)",
         "scored -> local maxima listed", "local maxima -> written (end)"]
seq = [(0, 1), (1, 2), (2, 3), (3, 4)]
print("  start -> band descriptor arrived: median %.0f; -> level record etc. arrived, pixel loads can be issued: median %.0f; -> this wave's pixels arrived and stored: median %.0f; -> rest of the set-up + barrier: median %.0f" % (
      np.median(b[:, 6] - b[:, 0]), np.median(b[:, 5] - b[:, 6]), np.median(b[:, 7] - b[:, 5]), np.median(b[:, 1] - b[:, 7])))
tot = (b[:, 4] - b[:, 0])
print("s_memtime ticks (100 MHz constant clock? or shader clock): total median %.0f  mean %.0f" % (np.median(tot), tot.mean()))
for nm, (a, c) in zip(names, seq):
    d = b[:, c] - b[:, a]
    print("%-46s median %7.0f  mean %7.0f  (%.1f %% of the mean total)" % (nm, np.median(d), d.mean(), 100 * d.mean() / tot.mean()))
PY
cat $O/${tag}_fc_stamps.txt
touch motioncheck_ccm_slam_amd/csrc/orb_kernels.hip; make -s -C motioncheck_ccm_slam_amd/csrc > /dev/null 2>&1
