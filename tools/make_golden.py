#!/usr/bin/env python3
"""Write the committed fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden vectors and cannot be built or imported here,
so these fixtures pin the ORACLE (drift guard) and give the GPU tests a data-only expectation that
travels to the GPU box.  Inputs are the deterministic synthetic generators of
motioncheck_ccm_slam_amd/synth.py; nothing from the reference tree is read.
"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from motioncheck_ccm_slam_amd import synth
from oracle import oracle_py as O

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)
par = O.default_params()
r = O.orb_extract(par, synth.frame(0), cand_level=0, want_levels=True)
np.savez_compressed(os.path.join(out, "orb_frame0.npz"),
                    kps=r["kps"], desc=r["desc"], cand0_xy=r["cand_xy"].astype(np.int16), cand0_score=r["cand_score"].astype(np.uint8),
                    level_sha=np.array([hashlib.sha256(l.tobytes()).hexdigest() for l in r["levels"]]))
small = synth.frame(3, 200, 160)
p2 = O.default_params(300, 1.2, 4, 20, 7)
r2 = O.orb_extract(p2, small)
np.savez_compressed(os.path.join(out, "orb_small.npz"), kps=r2["kps"], desc=r2["desc"])
a, b = synth.descriptor_pair(0)
bi, bd, sd = O.hamming_match(a, b)
np.savez_compressed(os.path.join(out, "match_pair0.npz"), best_idx=bi.astype(np.int16), best_dist=bd.astype(np.int16), second_dist=sd.astype(np.int16))
g = synth.local_ba_graph()
res = O.ba_solve(g, 5, float(np.float32(np.sqrt(5.991))), 10)
np.savez_compressed(os.path.join(out, "ba_local.npz"), poses=res["poses"], points_head=res["points"][:64],
                    chi2=np.array([res["chi2_initial"], res["chi2_final"]]), outliers=np.flatnonzero(res["outlier"]).astype(np.int32),
                    iterations=np.array([res["iterations_done"], res["trials"]]))
for f in sorted(os.listdir(out)):
    print(f, os.path.getsize(os.path.join(out, f)))
