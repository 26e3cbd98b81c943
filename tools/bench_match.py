#!/usr/bin/env python3
"""Time k_hamming_bf through the C ABI (config 3: 1000x1000 per pair).  Env: CCM_BF_VARIANT, CCM_BF_SPLIT."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from motioncheck_ccm_slam_amd import _lib, synth
from oracle import oracle_py as O
lib = _lib.load(); ctx = _lib.Context(0)
for n_pairs in (255, 10000):
    base = 64
    q, t = synth.descriptor_pairs(0, base)
    reps = (n_pairs + base - 1) // base
    qd = torch.from_numpy(np.tile(q, (reps, 1, 1))[:n_pairs]).cuda(); td = torch.from_numpy(np.tile(t, (reps, 1, 1))[:n_pairs]).cuda()
    bi = torch.empty((n_pairs, 1000), dtype=torch.int32, device="cuda"); bd = torch.empty_like(bi); sd = torch.empty_like(bi)
    def run():
        ctx.check(lib.ccm_hamming_match_dev(ctx.handle, C.c_void_p(qd.data_ptr()), 1000, C.c_size_t(1000), C.c_void_p(td.data_ptr()), 1000,
                                            C.c_size_t(1000), n_pairs, None, None, C.c_void_p(bi.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(sd.data_ptr())))
    run(); ctx.sync()
    ok = True
    for p in (0, 63, n_pairs - 1):
        rbi, rbd, rsd = O.hamming_match(q[p % base], t[p % base])
        ok &= bool((bi[p].cpu().numpy() == rbi).all() and (bd[p].cpu().numpy() == rbd).all() and (sd[p].cpu().numpy() == rsd).all())
    ctx.profile(True)
    n = 20 if n_pairs < 1000 else 5
    for _ in range(n): run()
    ms, cnt = ctx.profile_read()["k_hamming_bf"]
    ctx.profile(False)
    per = ms / n
    print("variant=%s split=%s pairs=%d  %.3f ms/launch  %.2f Gdist/s  %.1f Mqueries/s  parity=%s" %
          (os.environ.get("CCM_BF_VARIANT", "3 (matrix cores)"), os.environ.get("CCM_BF_SPLIT", "auto"), n_pairs, per, n_pairs * 1e6 / per / 1e6, n_pairs * 1000 / per / 1e3, ok))
