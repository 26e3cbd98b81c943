"""Synthetic OptimizeSim3 problems shared by the CPU and GPU tests: two keyframes see the same points, the map of
keyframe 2 is a similarity (drifted scale) away from the map of keyframe 1."""
import numpy as np


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rand_sim3(rng, rot=0.15, trans=0.5, scale=0.2):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = rng.uniform(-rot, rot)
    q = np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]])
    return np.concatenate([q, rng.uniform(-trans, trans, 3), [np.exp(rng.uniform(-scale, scale))]])


def sim3_map(S, X):
    return S[7] * (X @ quat_R(S[:4]).T) + S[4:7]


def make_problem(rng, n, outlier_frac=0.1, noise=0.5, start_err=0.05):
    K1 = np.array([458.654, 457.296, 367.215, 248.375]); K2 = np.array([435.2, 435.2, 367.4, 252.2])
    S_true = rand_sim3(rng)
    P2 = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2.5, 9, n)], 1)
    P1 = sim3_map(S_true, P2) + rng.normal(0, 0.002, (n, 3))
    proj = lambda K, P: np.stack([P[:, 0] / P[:, 2] * K[0] + K[2], P[:, 1] / P[:, 2] * K[1] + K[3]], 1)
    obs1 = proj(K1, P1) + rng.normal(0, noise, (n, 2)); obs2 = proj(K2, P2) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outlier_frac
    obs1[bad] += rng.uniform(15, 60, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))
    lvl1 = rng.integers(0, 8, n); lvl2 = rng.integers(0, 8, n)
    info1 = (1.0 / 1.2 ** (2 * lvl1)).astype(np.float32).astype(np.float64)      # mvInvLevelSigma2 is float
    info2 = (1.0 / 1.2 ** (2 * lvl2)).astype(np.float32).astype(np.float64)
    S0 = S_true.copy()
    S0[:4] += rng.normal(0, start_err * 0.3, 4); S0[:4] /= np.linalg.norm(S0[:4])
    S0[4:7] += rng.normal(0, start_err, 3); S0[7] *= np.exp(rng.normal(0, start_err))
    return dict(S_true=S_true, S0=S0, K1=K1, K2=K2, P1=P1, P2=P2, obs1=obs1, obs2=obs2, info1=info1, info2=info2, bad=bad)


def make_pose_graph(oracle, rng, n=60, drift=0.01, scale_drift=0.004, covis=2):
    """A closed trajectory of n keyframes with accumulated Sim3 drift, as the loop closer sees it: vertex estimates are
    the drifted Siw except the current keyframe (last), which carries its loop-corrected Sim3; spanning-tree and
    covisibility edges are measured from the drifted estimates (NonCorrectedSim3), the loop edge from the truth.
    Returns sim3 [n][8], fixed, edge_i, edge_j, meas [ne][8], truth [n][8]."""
    truth = []
    for i in range(n):
        ang = 2 * np.pi * i / n
        c = np.array([3 * np.cos(ang), 0.2 * np.sin(3 * ang), 3 * np.sin(ang)])
        yaw = -ang
        q = np.array([0, np.sin(yaw / 2), 0, np.cos(yaw / 2)])
        t = -quat_R(q) @ c
        truth.append(np.concatenate([q, t, [1.0]]))
    truth = np.array(truth)
    drifted = [truth[0].copy()]
    for i in range(1, n):
        rel = oracle.sim3_mul(truth[i], oracle.sim3_inverse(truth[i - 1]))                      # S_i,i-1
        noise = oracle.sim3_exp(np.concatenate([rng.normal(0, drift, 3), rng.normal(0, drift, 3), [rng.normal(0, scale_drift)]]))
        drifted.append(oracle.sim3_mul(oracle.sim3_mul(noise, rel), drifted[i - 1]))
    drifted = np.array(drifted)
    ei, ej, meas = [], [], []

    def edge(i, j, Si, Sj):                                                                   # vertex 0 = i, vertex 1 = j, Sji = Sjw * Swi
        ei.append(i); ej.append(j); meas.append(oracle.sim3_mul(Sj, oracle.sim3_inverse(Si)))
    for i in range(1, n):
        edge(i, i - 1, drifted[i], drifted[i - 1])                                             # spanning tree
        for k in range(2, covis + 2):
            if i - k >= 0:
                edge(i, i - k, drifted[i], drifted[i - k])                                     # covisibility, nIDj < nIDi
    edge(n - 1, 0, truth[n - 1], truth[0])                                                     # the loop connection
    sim3 = drifted.copy()
    sim3[n - 1] = truth[n - 1]                                                                 # CorrectedSim3 of the current keyframe
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    return sim3, fixed, np.array(ei, "i4"), np.array(ej, "i4"), np.array(meas), truth
