#!/usr/bin/env python3
"""Count, on the 256 frames of BASELINE config 2, how many descriptor bits / keypoints change when the oracle calls this
machine's libm cosf/sinf (what ORBextractor.cpp:105 calls) instead of include/ccm_sincos.h (what oracle and kernel share).
CPU only.  Prints one JSON line; the figures are quoted in DESIGN.md section 2."""
import json, os, platform, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import synth
from oracle import oracle_py as O

par = O.default_params()
n_kp = n_diff = bits = frames_diff = 0
worst = 0
for f in range(256):
    img = synth.frame(f)
    O.lib().orc_set_sincos_libm(0); a = O.orb_extract(par, img)
    O.lib().orc_set_sincos_libm(1); b = O.orb_extract(par, img)
    O.lib().orc_set_sincos_libm(0)
    assert (a["kps"] == b["kps"]).all()
    x = np.unpackbits(a["desc"] ^ b["desc"], axis=1).sum(1)
    n_kp += len(x); n_diff += int((x > 0).sum()); bits += int(x.sum()); frames_diff += int((x > 0).any()); worst = max(worst, int(x.max()))
print(json.dumps({"frames": 256, "keypoints": n_kp, "keypoints_with_a_different_descriptor": n_diff, "differing_bits": bits,
                  "of_bits": n_kp * 256, "frames_affected": frames_diff, "max_bits_in_one_descriptor": worst,
                  "libc": " ".join(platform.libc_ver())}))
