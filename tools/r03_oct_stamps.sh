#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_oct_stamps.sh -- where does the level-0 quadtree of one frame spend its ~100 us?
# Diagnostic build (-DOCT_STAMPS): s_memtime of thread 0 of the (frame 0, level 0) workgroup at its phase boundaries, one 256-frame launch.
O=gpurun_out; mkdir -p $O
touch motioncheck_ccm_slam_amd/csrc/orb_kernels.hip
make -s -C motioncheck_ccm_slam_amd/csrc EXTRA="-DOCT_STAMPS" > $O/oct_stamps.build 2>&1 || { tail -5 $O/oct_stamps.build; exit 1; }
timeout -k 10 200 python3 - <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import torch
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.orb import ORBextractor
ctx = _lib.Context(0); ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
frames = torch.from_numpy(synth.frames(0, 256)).cuda()
lib = _lib.load()
lib.ccm_debug_oct_stamps.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(64, np.uint64); n = np.zeros(1, np.int32)
for rep in range(3):
    ex.extract_dev(frames.data_ptr(), 752, 480, 752, 752 * 480, 256); ctx.sync()
    assert lib.ccm_debug_oct_stamps(buf.ctypes.data, n.ctypes.data) == 0
k = int(n[0]); t = buf[:k].astype(np.int64)
print("stamps:", k, " total %d ticks" % (t[-1] - t[0]))
# stamps 1, 2: counts known in gather rounds 0 and 1; 3: wave 0's keys stored; 4: gathered (barrier); 5: roots
print("start -> round 0 counts %d -> round 1 counts %d -> wave 0 stored %d -> barrier %d; -> roots made %d" % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]))
t = t[3:]; k -= 3
i = 3; p = 0
while i + 2 < k:
    print("pass %d: partition %d, order %d, new list %d" % (p, t[i] - t[i - 1], t[i + 1] - t[i], t[i + 2] - t[i + 1])); i += 3; p += 1
print("final selection %d" % (t[k - 1] - t[k - 2]))
PY
touch motioncheck_ccm_slam_amd/csrc/orb_kernels.hip; make -s -C motioncheck_ccm_slam_amd/csrc > /dev/null 2>&1
