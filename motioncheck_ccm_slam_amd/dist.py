"""One process per GPU: how the hot path is sharded.

* Extraction and matching shard by frames / frame pairs: contiguous ranges, no collective.
* Global BA shards LANDMARKS: every rank holds all poses and the landmarks of one contiguous range,
  balanced by the Schur cost k(k+1)/2 + k + 1 of a k-observation landmark; each LM trial sums the
  partial reduced camera systems with one RCCL all-reduce (csrc/ba_host.cpp, csrc/comm.cpp).
  `landmark_cuts` restates the C++ rule so that the CPU tests can check the sharded sum.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) of n_items for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def landmark_cuts(edge_point: np.ndarray, n_points: int, ranks: int) -> np.ndarray:
    deg = np.bincount(np.asarray(edge_point, np.int64), minlength=n_points).astype(np.float64)
    cost = 0.5 * deg * (deg + 1) + deg + 1
    total = cost.sum()
    acc = np.cumsum(cost)
    cuts = np.full(ranks + 1, n_points, np.int64)
    cuts[0] = 0
    r = 0
    for l in range(n_points):
        while r + 1 < ranks and acc[l] >= total * (r + 1) / ranks:
            r += 1
            cuts[r] = l + 1
    return cuts


def subgraph_for_rank(graph: dict, rank: int, ranks: int) -> dict:
    """The landmarks / edges rank `rank` owns (poses replicated)."""
    cuts = landmark_cuts(graph["edge_point"], len(graph["points"]), ranks)
    l0, l1 = int(cuts[rank]), int(cuts[rank + 1])
    sel = (graph["edge_point"] >= l0) & (graph["edge_point"] < l1)
    out = dict(graph)
    out["points"] = graph["points"][l0:l1]
    out["edge_pose"] = graph["edge_pose"][sel]
    out["edge_point"] = graph["edge_point"][sel] - l0
    out["obs"] = graph["obs"][sel]
    out["info"] = graph["info"][sel]
    out["landmark_range"] = (l0, l1)
    return out


def init_comm(ctx, rank: int, world: int):
    """Attach an RCCL communicator to a ccm context; the unique id travels over torch.distributed."""
    import torch
    import torch.distributed as dist
    from . import _lib
    lib = _lib.load()
    # the id travels with a status byte: if rank 0 cannot make one, EVERY rank raises instead of waiting in the broadcast
    msg = np.zeros(_lib.COMM_ID_BYTES + 1, np.uint8)
    if rank == 0:
        ident = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
        rc = lib.ccm_comm_unique_id(_lib.ptr(ident))
        msg[:_lib.COMM_ID_BYTES] = ident
        msg[-1] = 1 if rc else 0
    t = torch.from_numpy(msg.copy())
    if world > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
        t = t.cpu()
    msg = t.numpy().astype(np.uint8)
    if msg[-1]:
        raise _lib.CcmError(-6, "ccm_comm_unique_id failed on rank 0")
    ident = np.ascontiguousarray(msg[:_lib.COMM_ID_BYTES])
    ctx.check(lib.ccm_comm_init(ctx.handle, _lib.ptr(ident), world, rank))
