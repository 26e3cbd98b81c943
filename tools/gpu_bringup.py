#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP extractor / matcher with the CPU oracle (run on a GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motioncheck_ccm_slam_amd import synth
from motioncheck_ccm_slam_amd.orb import ORBextractor
from motioncheck_ccm_slam_amd.matcher import ORBmatcher
from oracle import oracle_py as O

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
imgs = synth.frames(0, nf)
ex = ORBextractor(1000, 1.2, 8, 20, 7)
t = time.time(); kps, desc, counts = ex.extract_batch(imgs); print("gpu extract", time.time() - t, "counts", counts)
par = O.default_params()
ok = True
for f in range(nf):
    ref = O.orb_extract(par, imgs[f], want_levels=True)
    for l in range(8):
        g = ex.image_pyramid_level(f, l)
        d = (g != ref["levels"][l])
        if d.any():
            ok = False
            print("frame", f, "level", l, "pyramid mismatch", int(d.sum()), "of", d.size, "first", np.argwhere(d)[:3].tolist())
    for l in range(8):
        r = O.orb_extract(par, imgs[f], cand_level=l)
        xy, sc = ex.fast_candidates(f, l)
        same = len(xy) == len(r["cand_xy"]) and (xy == r["cand_xy"]).all() and (sc == r["cand_score"]).all()
        if not same:
            ok = False
            print("frame", f, "level", l, "candidates differ: gpu", len(xy), "cpu", len(r["cand_xy"]))
            a = set(map(tuple, np.c_[xy, sc].tolist())); b = set(map(tuple, np.c_[r["cand_xy"], r["cand_score"]].tolist()))
            print("   only gpu", sorted(a - b)[:5], "only cpu", sorted(b - a)[:5])
    n = int(counts[f])
    rk, rd = ref["kps"], ref["desc"]
    if n != len(rk):
        ok = False; print("frame", f, "count gpu", n, "cpu", len(rk))
    m = min(n, len(rk))
    for name in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        bad = np.flatnonzero(kps[f, :m][name] != rk[:m][name])
        if len(bad):
            ok = False; print("frame", f, name, "mismatch at", bad[:5].tolist(), kps[f, bad[:3]][name], rk[bad[:3]][name])
    bad = np.flatnonzero((desc[f, :m] != rd[:m]).any(1))
    if len(bad):
        ok = False; print("frame", f, "descriptor mismatch rows", len(bad), bad[:5].tolist())
print("ORB", "OK" if ok else "MISMATCH")

mt = ORBmatcher(0.7)
q, tt = synth.descriptor_pairs(0, 4)
bi, bd, sd = mt.BruteForce(q, tt)
okm = True
for p in range(4):
    rbi, rbd, rsd = O.hamming_match(q[p], tt[p])
    if not ((bi[p] == rbi).all() and (bd[p] == rbd).all() and (sd[p] == rsd).all()):
        okm = False; print("pair", p, "mismatch", int((bi[p] != rbi).sum()), int((bd[p] != rbd).sum()), int((sd[p] != rsd).sum()))
print("MATCH", "OK" if okm else "MISMATCH")
# ragged
bi, bd, sd = mt.BruteForce(q[:2, :700], tt[:2, :1500 if tt.shape[1] >= 1500 else tt.shape[1]], nq_n=[700, 3], nt_n=[5, 0])
rbi, rbd, rsd = O.hamming_match(q[0, :700], tt[0, :5])
print("ragged", (bi[0] == rbi).all() and (sd[0] == rsd).all(), bi[1, :5], bd[1, :5])
# BoW
rng = np.random.default_rng(0)
n1 = n2 = 600
d1, d2 = q[0, :n1], tt[0, :n2]
node1 = rng.integers(0, 40, n1); node2 = rng.integers(0, 40, n2)
v1 = rng.random(n1) < 0.8; a1 = rng.random(n1) * 360; a2 = rng.random(n2) * 360
for v2 in (None, rng.random(n2) < 0.9):
    n, m = mt.SearchByBoW(d1, node1, v1, a1, d2, node2, a2, valid2=v2)
    rn, rm = O.match_bow(0.7, 1, 50, 0 if v2 is None else 1, d1, node1, v1, a1, d2, node2, v2, a2)
    print("bow", n, rn, (m == rm).all())
sys.exit(0 if ok and okm else 1)
