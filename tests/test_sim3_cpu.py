"""Sim3 oracle known answers (g2o sim3.h exp / product / inverse) and OptimizeSim3 on synthetic keyframe pairs."""
import numpy as np

from sim3_problems import make_problem, quat_R, sim3_map


def test_sim3_exp_known_answers(oracle):
    I = oracle.sim3_exp(np.zeros(7))
    assert np.allclose(I, [0, 0, 0, 1, 0, 0, 0, 1], atol=0)
    # pure scale: s = e^sigma, t = (s-1)/sigma * upsilon (C of sim3.h:96), no rotation
    S = oracle.sim3_exp([0, 0, 0, 1.0, -2.0, 0.5, 0.3])
    assert abs(S[7] - np.exp(0.3)) < 1e-15 and np.allclose(S[4:7], (np.exp(0.3) - 1) / 0.3 * np.array([1.0, -2.0, 0.5]), atol=1e-14)
    assert np.allclose(S[:4], [0, 0, 0, 1])
    # rotation about z by theta, sigma = 0: R = Rz(theta), t = V * upsilon with the SE3 V matrix
    th = 0.7
    S = oracle.sim3_exp([0, 0, th, 1.0, 0.0, 0.0, 0.0])
    assert np.allclose(quat_R(S[:4]), [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], atol=1e-15)
    assert np.allclose(S[4:7], [np.sin(th) / th, (1 - np.cos(th)) / th, 0], atol=1e-15) and S[7] == 1.0
    # group laws
    rng = np.random.default_rng(0)
    a = oracle.sim3_exp(rng.normal(0, 0.4, 7)); b = oracle.sim3_exp(rng.normal(0, 0.4, 7))
    ab = oracle.sim3_mul(a, b)
    X = rng.normal(size=(5, 3))
    assert np.allclose(sim3_map(ab, X), sim3_map(a, sim3_map(b, X)), atol=1e-13)
    e = oracle.sim3_mul(a, oracle.sim3_inverse(a))
    assert np.allclose(sim3_map(e, X), X, atol=1e-13) and abs(e[7] - 1) < 1e-15


def test_optimize_sim3_recovers_the_similarity(oracle):
    rng = np.random.default_rng(1)
    p = make_problem(rng, 120, outlier_frac=0.15, noise=0.3)
    S, inl, nin = oracle.optimize_sim3(p["S0"], 0, p["K1"], p["K2"], p["P1"], p["P2"], p["obs1"], p["obs2"], p["info1"], p["info2"], 10.0)
    X = p["P2"]
    assert np.abs(sim3_map(S, X) - sim3_map(p["S_true"], X)).max() < 0.08          # depth (scale) is weakly observable: 0.5 % at 7 m
    assert abs(S[7] / p["S_true"][7] - 1) < 1e-2
    assert not inl[p["bad"]].any() and inl[~p["bad"]].mean() > 0.9 and nin == inl.sum()
    # fixed scale: the scale of the start is kept exactly
    S2, _, _ = oracle.optimize_sim3(p["S0"], 1, p["K1"], p["K2"], p["P1"], p["P2"], p["obs1"], p["obs2"], p["info1"], p["info2"], 10.0)
    assert S2[7] == p["S0"][7]
    # fewer than 10 survivors: returns 0 and leaves the estimate alone
    q = make_problem(rng, 12, outlier_frac=0.5)
    S3, inl3, n3 = oracle.optimize_sim3(q["S0"], 0, q["K1"], q["K2"], q["P1"], q["P2"], q["obs1"], q["obs2"], q["info1"], q["info2"], 10.0)
    if n3 == 0:
        assert (S3 == q["S0"]).all()
    S4, inl4, n4 = oracle.optimize_sim3(q["S0"], 0, q["K1"], q["K2"], q["P1"][:5], q["P2"][:5], q["obs1"][:5], q["obs2"][:5], q["info1"][:5], q["info2"][:5], 10.0)
    assert n4 == 0 and (S4 == q["S0"]).all()
