"""world_size-2 gloo tests of the sharding rules (no GPU): frame ranges cover the batch exactly, and
the landmark-sharded reduced camera systems sum to the unsharded one (what the RCCL all-reduce in
csrc/ba_host.cpp computes), checked with the oracle's reduced-system builder."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from motioncheck_ccm_slam_amd import synth
    from motioncheck_ccm_slam_amd import dist as D
    from oracle import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # frames
    lo, hi = D.shard_range(257, rank, world)
    cover = torch.zeros(257, dtype=torch.int32); cover[lo:hi] = 1
    dist.all_reduce(cover)
    ok_frames = bool((cover == 1).all())
    # landmark-sharded Schur
    g = synth.gba_graph(n_kf=24, n_points=900, n_agents=3, seed=5)
    lam = 0.37
    sub = D.subgraph_for_rank(g, rank, world)
    Hs, bs, _ = O.ba_reduced_system(sub, float(np.float32(np.sqrt(5.99))), 0.0)      # lambda only on the landmark blocks below
    Hs, bs, _ = O.ba_reduced_system(sub, float(np.float32(np.sqrt(5.99))), lam)
    n = Hs.shape[0]
    Hs = Hs - lam * np.eye(n)                                      # the pose-diagonal lambda is added once, after the sum
    t = torch.from_numpy(np.concatenate([Hs.ravel(), bs]))
    dist.all_reduce(t)
    Hsum = t[:n * n].numpy().reshape(n, n) + lam * np.eye(n); bsum = t[n * n:].numpy()
    Hfull, bfull, _ = O.ba_reduced_system(g, float(np.float32(np.sqrt(5.99))), lam)
    err = max(np.abs(Hsum - Hfull).max() / np.abs(Hfull).max(), np.abs(bsum - bfull).max() / np.abs(bfull).max())
    # the ranks' landmark ranges partition the landmarks
    l0, l1 = sub["landmark_range"]
    owned = torch.zeros(len(g["points"]), dtype=torch.int32); owned[l0:l1] = 1
    dist.all_reduce(owned)
    q.put((rank, ok_frames, float(err), bool((owned == 1).all())))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps: p.join(60)
    for rank, ok_frames, err, ok_lm in res:
        assert ok_frames and ok_lm
        assert err < 1e-12


def test_landmark_cuts_balance():
    from motioncheck_ccm_slam_amd import synth, dist as D
    g = synth.gba_graph(n_kf=60, n_points=4000, n_agents=3, seed=6)
    for ranks in (1, 2, 4, 8):
        cuts = D.landmark_cuts(g["edge_point"], len(g["points"]), ranks)
        assert cuts[0] == 0 and cuts[-1] == len(g["points"]) and (np.diff(cuts) >= 0).all()
        deg = np.bincount(g["edge_point"], minlength=len(g["points"])).astype(float)
        cost = 0.5 * deg * (deg + 1) + deg + 1
        per = [cost[cuts[r]:cuts[r + 1]].sum() for r in range(ranks)]
        assert max(per) < 1.1 * cost.sum() / ranks + cost.max()


def test_cxx_landmark_cuts_equal_python_rule():
    """The C++ partition used inside ccm_ba_solve (host-only entry point) is the rule the gloo test sums over."""
    import ctypes as C
    from motioncheck_ccm_slam_amd import _lib, synth, dist as D
    lib = _lib.load()
    g = synth.gba_graph(n_kf=90, n_points=7000, n_agents=3, seed=9)
    ep = np.ascontiguousarray(g["edge_point"], "i4")
    for ranks in (1, 2, 3, 4, 8):
        cuts = np.zeros(ranks + 1, "i4")
        assert lib.ccm_ba_landmark_cuts(_lib.ptr(ep), len(ep), len(g["points"]), ranks, _lib.ptr(cuts)) == 0
        assert (cuts == D.landmark_cuts(ep, len(g["points"]), ranks)).all()
    bad = ep.copy(); bad[0] = 10 ** 6
    assert lib.ccm_ba_landmark_cuts(_lib.ptr(bad), len(bad), len(g["points"]), 2, _lib.ptr(np.zeros(3, "i4"))) == -1
