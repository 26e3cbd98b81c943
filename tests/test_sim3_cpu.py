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


def test_sim3_log_inverts_exp(oracle):
    rng = np.random.default_rng(2)
    cases = [rng.normal(0, 0.5, 7) for _ in range(20)]
    cases += [np.array([1e-7, -2e-7, 1e-7, 0.3, -0.2, 0.1, 0.2]),          # theta below eps, sigma not
              np.array([0.4, -0.2, 0.3, 0.3, -0.2, 0.1, 1e-7]),            # sigma below eps, theta not
              np.array([1e-7, 0, 0, 0.3, -0.2, 0.1, 1e-8]), np.zeros(7)]
    for u in cases:
        back = oracle.sim3_log(oracle.sim3_exp(u))
        assert np.abs(back - u).max() < 1e-9, (u, back)
    # known answer: pure scale 2 with translation t: log = (0, t * sigma / (s - 1), ln 2)
    S = np.array([0, 0, 0, 1, 1.0, 2.0, -1.0, 2.0])
    assert np.allclose(oracle.sim3_log(S), np.concatenate([[0, 0, 0], np.array([1.0, 2.0, -1.0]) * np.log(2.0), [np.log(2.0)]]), atol=1e-14)


def test_essential_graph_closes_the_loop(oracle):
    from sim3_problems import make_pose_graph
    rng = np.random.default_rng(3)
    sim3, fixed, ei, ej, meas, truth = make_pose_graph(oracle, rng, n=40)
    out, info = oracle.essential_graph(sim3, fixed, ei, ej, meas, False, 20)
    assert info["chi2_final"] < 0.05 * info["chi2_initial"] and info["iterations_done"] >= 3
    assert (out[0] == sim3[0]).all()                                        # the loop keyframe is fixed
    centre = lambda S: -quat_R(S[:4]).T @ S[4:7] / S[7]
    before = max(np.linalg.norm(centre(sim3[i]) - centre(truth[i])) for i in range(40))
    after = max(np.linalg.norm(centre(out[i]) - centre(truth[i])) for i in range(40))
    assert after < 0.6 * before                                             # the accumulated drift is spread over the loop
    # a consistent graph (all measurements from the estimates) has zero error and does not move
    sim3b = sim3.copy(); sim3b[-1] = sim3[-2]                               # irrelevant vertex value; rebuild consistent edges
    meas_c = np.array([oracle.sim3_mul(sim3b[j], oracle.sim3_inverse(sim3b[i])) for i, j in zip(ei, ej)])
    out2, info2 = oracle.essential_graph(sim3b, fixed, ei, ej, meas_c, False, 20)
    assert info2["chi2_initial"] < 1e-20 and np.abs(out2 - sim3b).max() < 1e-9
