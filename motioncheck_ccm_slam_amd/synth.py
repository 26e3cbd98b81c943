"""Deterministic synthetic inputs for the hot path.

There is no dataset access, so every test and benchmark input is generated
here from a SplitMix64 stream (counter based, so it vectorises in numpy).
Shapes follow the reference's defaults: 752x480 EuRoC frames
(cslam/conf/vi_euroc.yaml:9-12), 1000 features, 8 levels
(cslam/conf/config.yaml:38-51).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

EUROC_W, EUROC_H = 752, 480
EUROC_K = (458.654, 457.296, 367.215, 248.375)


class SplitMix64:
    """SplitMix64; `take(n)` returns the next n outputs as uint64."""

    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def take(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * _GOLDEN
            self.state = self.state + np.uint64(n) * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        return z

    def below(self, n: int, m: int) -> np.ndarray:
        return (self.take(n) % np.uint64(m)).astype(np.int64)

    def uniform(self, n: int) -> np.ndarray:
        """doubles in [0,1)"""
        return (self.take(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n: int) -> np.ndarray:
        """Box-Muller on two uniform streams (2n draws)."""
        u = self.uniform(2 * n)
        u1 = np.maximum(u[:n], 1e-300)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u[n:])


def frame(f: int, w: int = EUROC_W, h: int = EUROC_H, n_rect: int = 600) -> np.ndarray:
    """Synthetic frame f (configs 1 and 2): grey 128, n_rect rectangles, +-4 noise."""
    rng = SplitMix64(0x0C0FFEE0 + f)
    img = np.full((h, w), 128, dtype=np.int32)
    r = rng.take(5 * n_rect).reshape(n_rect, 5)
    x0 = (r[:, 0] % np.uint64(w)).astype(np.int64)
    y0 = (r[:, 1] % np.uint64(h)).astype(np.int64)
    rw = (r[:, 2] % np.uint64(89)).astype(np.int64) + 8
    rh = (r[:, 3] % np.uint64(89)).astype(np.int64) + 8
    val = (r[:, 4] % np.uint64(256)).astype(np.int64)
    for i in range(n_rect):
        img[y0[i]:min(h, y0[i] + rh[i]), x0[i]:min(w, x0[i] + rw[i])] = val[i]
    noise = (rng.take(w * h) % np.uint64(9)).astype(np.int32).reshape(h, w) - 4
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def frames(first: int, count: int, w: int = EUROC_W, h: int = EUROC_H) -> np.ndarray:
    return np.stack([frame(first + i, w, h) for i in range(count)])


def descriptor_pair(p: int, n: int = 1000, flip: float = 0.06, inlier: float = 0.7):
    """Config 3, pair p: A random; B[i] = A[perm[i]] with bit noise for 70 % of rows, fresh for 30 %."""
    rng = SplitMix64(0xDE5C0000 + p)
    a = rng.take(n * 4).view(np.uint8).reshape(n, 32).copy()
    perm = np.argsort(rng.take(n), kind="stable")
    is_in = rng.uniform(n) < inlier
    flips = (rng.uniform(n * 256) < flip).reshape(n, 256)
    mask = np.packbits(flips, axis=1, bitorder="little")
    fresh = rng.take(n * 4).view(np.uint8).reshape(n, 32)
    b = np.where(is_in[:, None], a[perm] ^ mask, fresh).astype(np.uint8)
    return a, b


def descriptor_pairs(first: int, count: int, n: int = 1000):
    qa = np.empty((count, n, 32), np.uint8)
    tb = np.empty((count, n, 32), np.uint8)
    for i in range(count):
        qa[i], tb[i] = descriptor_pair(first + i, n)
    return qa, tb


def descriptor_pairs_torch(first: int, count: int, n: int = 1000, flip: float = 0.06, inlier: float = 0.7,
                           device="cuda", batch: int = 50):
    """descriptor_pairs() evaluated with torch on `device`: the same SplitMix64 streams, bit for bit (checked against the
    numpy version in tests/test_synth_cpu.py), so that the 10,000 pairs of config 3 (640 MB) are made where they are used
    instead of taking a minute of numpy time and a PCIe trip.  Returns (A, B) uint8 tensors [count, n, 32]."""
    import torch

    def s64(v):                                   # a uint64 constant as the int64 with the same bits
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    G, M1, M2 = s64(0x9E3779B97F4A7C15), s64(0xBF58476D1CE4E5B9), s64(0x94D049BB133111EB)

    def lsr(z, k):                                # logical shift right of int64 bit patterns
        return (z >> k) & ((1 << (64 - k)) - 1)

    def mix(z):
        z = (z ^ lsr(z, 30)) * M1
        z = (z ^ lsr(z, 27)) * M2
        return z ^ lsr(z, 31)

    per = 266 * n                                 # draws of one pair: 4n + n + n + 256n + 4n
    idx = torch.arange(1, per + 1, dtype=torch.int64, device=device)
    A = torch.empty((count, n, 32), dtype=torch.uint8, device=device)
    B = torch.empty_like(A)
    bitw = (1 << torch.arange(8, device=device, dtype=torch.int32))
    for b0 in range(0, count, batch):
        b1 = min(count, b0 + batch)
        seeds = torch.tensor([s64(0xDE5C0000 + first + p) for p in range(b0, b1)], dtype=torch.int64, device=device)
        z = mix(seeds[:, None] + idx[None, :] * G)                                   # [pairs, per]
        m = b1 - b0
        a = z[:, :4 * n].contiguous().view(torch.uint8).reshape(m, n, 32)
        keys = z[:, 4 * n:5 * n] ^ (-(1 << 63))                                      # unsigned order as signed order
        perm = torch.sort(keys, dim=1, stable=True).indices
        uni = lambda t: lsr(t, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
        is_in = uni(z[:, 5 * n:6 * n]) < inlier
        flips = (uni(z[:, 6 * n:262 * n]) < flip).reshape(m, n, 32, 8)
        mask = (flips.to(torch.int32) * bitw).sum(-1).to(torch.uint8)
        fresh = z[:, 262 * n:266 * n].contiguous().view(torch.uint8).reshape(m, n, 32)
        ap = torch.gather(a, 1, perm[:, :, None].expand(m, n, 32))
        A[b0:b1] = a
        B[b0:b1] = torch.where(is_in[:, :, None], ap ^ mask, fresh)
    return A, B


# --------------------------------------------------------------------------- BA graphs
def _look_at(cam_pos: np.ndarray, target: np.ndarray) -> np.ndarray:
    """World->camera rotation with +z toward target, y roughly down."""
    z = target - cam_pos
    z /= np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    if np.linalg.norm(x) < 1e-6:
        x = np.array([1.0, 0.0, 0.0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])


def _rot_to_quat(R: np.ndarray) -> np.ndarray:
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _quat_to_rot(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _exp_so3(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def _finish_graph(rng, Rs, ts, pts, cand, fixed, max_obs, noise=True,
                  rot_sigma=0.01, trans_sigma=0.03, pt_sigma=0.03, anchor=None):
    """Project, keep in-image observations (nearest `max_obs` keyframes, at least 2), add pixel
    noise, perturb the estimate.  cand: int array [n_points, C] of candidate keyframes, -1 = none."""
    fx, fy, cx, cy = EUROC_K
    Rs = np.asarray(Rs); ts = np.asarray(ts); cand = np.asarray(cand, np.int64)
    P = len(Rs)
    n, C = cand.shape
    ci = np.where(cand >= 0, cand, 0)
    pc = np.einsum("ncij,nj->nci", Rs[ci], pts) + ts[ci]
    z = pc[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = fx * pc[..., 0] / z + cx
        v = fy * pc[..., 1] / z + cy
    ok = (cand >= 0) & (z >= 0.3) & (u >= 0) & (u < EUROC_W) & (v >= 0) & (v < EUROC_H)
    # drop duplicate candidates of one landmark (keep the first)
    srt = np.sort(np.where(ok, cand, -1 - np.arange(C)[None, :]), axis=1)
    order = np.argsort(np.where(ok, cand, -1 - np.arange(C)[None, :]), axis=1, kind="stable")
    dup_sorted = np.zeros_like(ok)
    dup_sorted[:, 1:] = srt[:, 1:] == srt[:, :-1]
    dup = np.zeros_like(ok)
    np.put_along_axis(dup, order, dup_sorted, axis=1)
    ok &= ~dup
    key = z if anchor is None else np.abs(cand - np.asarray(anchor)[:, None]).astype(np.float64)
    depth_rank = np.argsort(np.argsort(np.where(ok, key, np.inf), axis=1, kind="stable"), axis=1, kind="stable")
    ok &= depth_rank < max_obs
    keep_pt = ok.sum(1) >= 2
    ok &= keep_pt[:, None]
    new_index = np.cumsum(keep_pt) - 1
    li, cj = np.nonzero(ok)
    e_pose = cand[li, cj]
    o = np.lexsort((e_pose, li))
    li, cj, e_pose = li[o], cj[o], e_pose[o]
    e_pt = new_index[li].astype(np.int32)
    e_obs = np.stack([u[li, cj], v[li, cj]], 1)
    e_pose = e_pose.astype(np.int32)
    pts = pts[keep_pt]
    E = len(e_pose)
    octave = rng.below(E, 8)
    sigma = 1.2 ** octave
    if noise:
        e_obs = e_obs + rng.normal(2 * E).reshape(E, 2) * sigma[:, None]
    # the reference holds observations and inverse sigmas as float32 (src/Optimizer.cpp:133-134)
    e_obs = e_obs.astype(np.float32).astype(np.float64)
    scale = np.float32(1.0)
    inv_s2 = []
    for _ in range(8):
        inv_s2.append(np.float32(1.0) / (scale * scale))
        scale = np.float32(scale * np.float32(1.2))
    e_info = np.asarray(inv_s2, np.float32)[octave].astype(np.float64)
    poses = np.zeros((P, 7))
    gt = np.zeros((P, 7))
    nrm = rng.normal(6 * P).reshape(P, 6)
    for i in range(P):
        R, t = Rs[i], ts[i]
        gt[i, :4] = _rot_to_quat(R); gt[i, 4:] = t
        if not fixed[i]:
            dR = _exp_so3(nrm[i, :3] * rot_sigma)
            R = dR @ R
            t = dR @ t + nrm[i, 3:] * trans_sigma
        poses[i, :4] = _rot_to_quat(R)
        poses[i, 4:] = t
    pts_est = pts + rng.normal(3 * len(pts)).reshape(-1, 3) * pt_sigma
    intr = np.tile(np.asarray(EUROC_K, np.float64), (P, 1))
    return dict(poses=poses, fixed=np.asarray(fixed, np.uint8), intr=intr, points=pts_est,
                edge_pose=e_pose, edge_point=e_pt, obs=e_obs, info=e_info,
                gt_poses=gt, gt_points=pts)


def local_ba_graph(n_free: int = 20, n_fixed: int = 10, n_points: int = 5000, seed: int = 0xBA000004,
                   max_obs: int = 8, noise: bool = True):
    """Config 4: keyframes on a 3 m arc looking inward, landmarks in a 4x4x2 m box in front."""
    rng = SplitMix64(seed)
    P = n_free + n_fixed
    ang = np.linspace(-0.6, 0.6, P)
    cams = np.stack([3.0 * np.sin(ang), -3.0 * np.cos(ang), 0.15 * np.sin(3 * ang)], 1)
    Rs, ts = [], []
    for i in range(P):
        R = _look_at(cams[i], np.array([0.0, 2.5, 0.0]))
        Rs.append(R); ts.append(-R @ cams[i])
    u = rng.uniform(3 * n_points).reshape(n_points, 3)
    pts = np.stack([(u[:, 0] - 0.5) * 4.0, 1.0 + u[:, 1] * 4.0, (u[:, 2] - 0.5) * 2.0], 1)
    fixed = np.zeros(P, bool)
    if n_fixed:
        fixed[np.unique(np.linspace(0, P - 1, n_fixed).round().astype(int))] = True
        while fixed.sum() < n_fixed:
            fixed[np.flatnonzero(~fixed)[0]] = True
    cand = np.tile(np.arange(P), (n_points, 1))
    anchor = rng.below(n_points, P)
    return _finish_graph(rng, Rs, ts, pts, cand, fixed, max_obs, noise, anchor=anchor)


def gba_graph(n_kf: int = 2000, n_points: int = 200000, n_agents: int = 3, seed: int = 0xBA000005,
              max_obs: int = 8, cross_frac: float = 0.05, noise: bool = True):
    """Config 5: agents on interleaved Lissajous loops, 0.15 m keyframe spacing, landmarks 3-8 m in
    front of an anchor keyframe and seen by its along-track neighbours; a fraction of landmarks is
    also seen by 4 keyframes of another agent (loop / map-merge edges).  One fixed keyframe."""
    rng = SplitMix64(seed)
    per = [n_kf // n_agents + (1 if a < n_kf % n_agents else 0) for a in range(n_agents)]
    cams, heads, agent_of = [], [], []
    for a, m in enumerate(per):
        length = 0.15 * m
        rad = max(length / (2 * np.pi * 1.3), 1.0)
        s = np.linspace(0, 2 * np.pi, m, endpoint=False)
        ph = 2 * np.pi * a / n_agents
        pos = np.stack([rad * 1.3 * np.sin(s + ph), rad * 0.8 * np.sin(2 * (s + ph)) + 0.4 * a,
                        0.3 * np.sin(3 * s + ph)], 1)
        d = np.roll(pos, -1, 0) - pos
        cams.append(pos); heads.append(d / np.linalg.norm(d, axis=1, keepdims=True))
        agent_of += [a] * m
    cams = np.concatenate(cams); heads = np.concatenate(heads); agent_of = np.asarray(agent_of)
    P = len(cams)
    Rs = np.zeros((P, 3, 3)); ts = np.zeros((P, 3))
    for i in range(P):
        side = np.cross(heads[i], np.array([0, 0, 1.0]))
        side /= np.linalg.norm(side)
        Rs[i] = _look_at(cams[i], cams[i] + 0.6 * heads[i] + side)
        ts[i] = -Rs[i] @ cams[i]
    anchor = rng.below(n_points, P)
    depth = 3.0 + 5.0 * rng.uniform(n_points)
    uu = (rng.uniform(n_points) - 0.5) * 1.2
    vv = (rng.uniform(n_points) - 0.5) * 0.8
    pc = np.stack([uu * depth, vv * depth, depth], 1)
    pts = np.einsum("nji,nj->ni", Rs[anchor], pc - ts[anchor])
    starts = np.cumsum([0] + per)
    per_a = np.asarray(per)
    half = max_obs // 2
    a_of = agent_of[anchor]
    k = anchor - starts[a_of]
    offs = np.arange(-half, half + 1)
    cand = starts[a_of][:, None] + (k[:, None] + offs[None, :]) % per_a[a_of][:, None]
    cross = np.full((n_points, 4), -1, np.int64)
    if n_agents > 1:
        is_cross = rng.uniform(n_points) < cross_frac
        b = (a_of + 1 + rng.below(n_points, n_agents - 1)) % n_agents
        for ag in range(n_agents):
            sel = np.flatnonzero(is_cross & (b == ag))
            if len(sel) == 0:
                continue
            seg = cams[starts[ag]:starts[ag] + per[ag]]
            sub = seg[::max(1, per[ag] // 128)]
            j = np.argmin(np.linalg.norm(sub[None, :, :] - pts[sel, None, :], axis=2), axis=1) * max(1, per[ag] // 128)
            cross[sel] = starts[ag] + (j[:, None] + np.arange(-2, 2)[None, :]) % per[ag]
    cand = np.concatenate([cand, cross], 1)
    fixed = np.zeros(P, bool); fixed[0] = True
    return _finish_graph(rng, Rs, ts, pts, cand, fixed, max_obs + 4, noise)
