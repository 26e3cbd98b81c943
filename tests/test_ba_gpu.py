"""HIP bundle adjustment vs the CPU oracle through the C ABI (BASELINE configs 4 and 5).
Tolerance: max |log(T_gpu T_cpu^-1)| <= 1e-5 per keyframe, the bound BASELINE.json states ("pose delta
within 1e-5"); summation order differs (atomics), so this is a tolerance, not bit equality."""
import os

import numpy as np
import pytest

from motioncheck_ccm_slam_amd import synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer, pose_delta

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def test_local_ba_config4(ctx, oracle):
    g = synth.local_ba_graph()
    r = Optimizer.LocalBundleAdjustmentClient(g, ctx=ctx)
    ref = oracle.ba_solve(g, 5, float(np.float32(np.sqrt(5.991))), 10)
    assert pose_delta(r["poses"], ref["poses"]).max() <= TOL
    assert np.abs(r["points"] - ref["points"]).max() <= 1e-6
    assert r["iterations_done"] == ref["iterations_done"] and r["trials"] == ref["trials"]
    assert np.isclose(r["chi2_initial"], ref["chi2_initial"], rtol=1e-10) and np.isclose(r["chi2_final"], ref["chi2_final"], rtol=1e-9)
    assert (r["outlier"] == ref["outlier"]).all()
    z = np.load(os.path.join(G, "ba_local.npz"))
    assert pose_delta(r["poses"], z["poses"]).max() <= TOL
    # stage results: the first stage alone (5 robust iterations) also agrees
    r1 = Optimizer.BundleAdjustmentClient({**g}, 5, ctx=ctx)
    ref1 = oracle.ba_solve(g, 5, float(np.float32(np.sqrt(5.99))))
    assert pose_delta(r1["poses"], ref1["poses"]).max() <= TOL
    # fixed keyframes never move
    fx = g["fixed"].astype(bool)
    assert (r["poses"][fx] == g["poses"][fx]).all()


def test_gba_small_and_medium(ctx, oracle):
    for kf, pts, its in ((60, 3000, 20), (240, 20000, 6)):
        g = synth.gba_graph(n_kf=kf, n_points=pts, n_agents=3, seed=kf)
        r = Optimizer.MapFusionGBA(g, its, ctx=ctx)
        ref = oracle.ba_solve(g, its, float(np.float32(np.sqrt(5.99))))
        assert pose_delta(r["poses"], ref["poses"]).max() <= TOL
        assert r["iterations_done"] == ref["iterations_done"]
        assert np.isclose(r["chi2_final"], ref["chi2_final"], rtol=1e-8)


def test_landmarks_with_long_observation_lists(ctx, oracle):
    """Landmarks seen by up to 64 keyframes (60 of them free): the pair enumeration of the block structure keeps a landmark's
    free edges in LDS up to 48 and takes another path beyond; the 8-lane landmark kernels make several trips per landmark."""
    g = synth.local_ba_graph(n_free=60, n_fixed=4, n_points=400, max_obs=64, seed=11)
    assert np.bincount(g["edge_point"]).max() > 56
    r = Optimizer.LocalBundleAdjustmentClient(g, ctx=ctx)
    ref = oracle.ba_solve(g, 5, float(np.float32(np.sqrt(5.991))), 10)
    assert pose_delta(r["poses"], ref["poses"]).max() <= TOL
    assert np.abs(r["points"] - ref["points"]).max() <= 1e-6
    assert r["iterations_done"] == ref["iterations_done"] and r["trials"] == ref["trials"]
    assert (r["outlier"] == ref["outlier"]).all()


def test_workspace_and_optional_outlier_output(ctx):
    """BaWorkspace (page-locked pose / point arrays a caller re-uses) changes where the results land, not what they are; the
    global entry points leave the per-edge outlier test to LocalBundleAdjustmentClient, the one the reference runs it in."""
    from motioncheck_ccm_slam_amd.optimizer import BaWorkspace
    g = synth.gba_graph(n_kf=60, n_points=3000, n_agents=3, seed=60)
    a = Optimizer.MapFusionGBA(g, 4, ctx=ctx)
    assert a["outlier"] is None
    ws = BaWorkspace(ctx, 80, 4000)                      # larger than the graph: views of the first rows come back
    try:
        b = Optimizer.MapFusionGBA(g, 4, ctx=ctx, workspace=ws)
        assert b["poses"].shape == a["poses"].shape and (b["poses"] == a["poses"]).all() and (b["points"] == a["points"]).all()
        assert np.shares_memory(b["poses"], ws.poses) and np.shares_memory(b["points"], ws.points)
        gl = synth.local_ba_graph()
        c1 = Optimizer.LocalBundleAdjustmentClient(gl, ctx=ctx)
        assert c1["outlier"] is not None and c1["outlier"].sum() > 0
        with pytest.raises(ValueError):
            Optimizer.MapFusionGBA(synth.gba_graph(n_kf=90, n_points=3000, n_agents=3, seed=1), 1, ctx=ctx, workspace=ws)
    finally:
        ws.close()


def test_edge_order_does_not_matter(ctx):
    """The library orders the observations by (landmark, keyframe) itself (sorted input is detected in one pass, anything else goes
    through a linear-time counting sort): a shuffled edge list must give bit-identical poses, points and the outlier flags of the
    SAME observations."""
    g = synth.local_ba_graph()
    r = Optimizer.LocalBundleAdjustmentClient(g, ctx=ctx)
    rng = np.random.default_rng(3)
    o = rng.permutation(len(g["edge_pose"]))
    g2 = dict(g, edge_pose=g["edge_pose"][o], edge_point=g["edge_point"][o], obs=g["obs"][o], info=g["info"][o])
    r2 = Optimizer.LocalBundleAdjustmentClient(g2, ctx=ctx)
    assert (r2["poses"] == r["poses"]).all() and (r2["points"] == r["points"]).all()
    assert (r2["outlier"] == r["outlier"][o]).all() and r["outlier"].sum() > 0
    assert r2["iterations_done"] == r["iterations_done"] and r2["chi2_final"] == r["chi2_final"]


def test_large_edge_list_indexed_on_the_device(ctx):
    """Above 400,000 edges an unsharded map's edge list is checked and indexed on the device (k_ix_*).  The device path (sorted list),
    the host path forced by CCM_BA_HOST_INDEX=1 (child process) and the fallback for a shuffled list must give bit-identical results."""
    import subprocess, sys
    g = synth.gba_graph(n_kf=600, n_points=60000, n_agents=3, seed=11)
    assert len(g["edge_pose"]) >= 400000
    r = Optimizer.MapFusionGBA(g, 3, ctx=ctx)
    rng = np.random.default_rng(4)
    o = rng.permutation(len(g["edge_pose"]))
    g2 = dict(g, edge_pose=g["edge_pose"][o], edge_point=g["edge_point"][o], obs=g["obs"][o], info=g["info"][o])
    r2 = Optimizer.MapFusionGBA(g2, 3, ctx=ctx)
    assert (r2["poses"] == r["poses"]).all() and (r2["points"] == r["points"]).all() and r2["chi2_final"] == r["chi2_final"]
    code = ("import numpy as np\nfrom motioncheck_ccm_slam_amd import _lib, synth\nfrom motioncheck_ccm_slam_amd.optimizer import Optimizer\n"
            "g = synth.gba_graph(n_kf=600, n_points=60000, n_agents=3, seed=11)\nr = Optimizer.MapFusionGBA(g, 3, ctx=_lib.Context(0))\n"
            "print(repr(float(r['chi2_final'])), repr(float(np.abs(r['poses']).sum())))")
    env = dict(os.environ, CCM_BA_HOST_INDEX="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    chi, ps = out.stdout.strip().splitlines()[-1].split()
    assert float(chi) == float(r["chi2_final"]) and float(ps) == float(np.abs(r["poses"]).sum())


def test_noise_free_graph_is_recovered(ctx):
    g = synth.local_ba_graph(n_free=8, n_fixed=3, n_points=600, seed=31, noise=False)
    r = Optimizer.BundleAdjustmentClient(g, 25, bRobust=False, ctx=ctx)
    assert r["chi2_final"] < 1e-6 * r["chi2_initial"]
    assert pose_delta(r["poses"], g["gt_poses"]).max() < 1e-5
    assert np.abs(r["points"] - g["gt_points"]).max() < 1e-4


def test_edge_cases(ctx, oracle):
    g = synth.local_ba_graph(n_free=5, n_fixed=2, n_points=300, seed=41)
    # stop flag already set: nothing moves (Optimizer.cpp:531-533 returns before optimising)
    r = Optimizer.LocalBundleAdjustmentClient(g, pbStopFlag=np.ones(1, np.uint8), ctx=ctx)
    assert r["stopped"] and (r["poses"] == g["poses"]).all() and r["iterations_done"] == 0
    # every keyframe fixed: only the landmarks are refined
    g2 = dict(g); g2["fixed"] = np.ones_like(g["fixed"])
    r = Optimizer.BundleAdjustmentClient(g2, 5, ctx=ctx)
    ref = oracle.ba_solve(g2, 5, float(np.float32(np.sqrt(5.99))))
    assert (r["poses"] == g["poses"]).all() and np.abs(r["points"] - ref["points"]).max() < 1e-8
    # a landmark seen once and a keyframe without observations
    keep = np.ones(len(g["edge_pose"]), bool)
    first = np.flatnonzero(g["edge_point"] == 0)
    keep[first[1:]] = False
    keep[g["edge_pose"] == int(np.flatnonzero(g["fixed"] == 0)[0])] = False
    g3 = {k: (v[keep] if k in ("edge_pose", "edge_point", "obs", "info") else v) for k, v in g.items()}
    r = Optimizer.BundleAdjustmentClient(g3, 5, ctx=ctx)
    ref = oracle.ba_solve(g3, 5, float(np.float32(np.sqrt(5.99))))
    assert pose_delta(r["poses"], ref["poses"]).max() <= TOL
    with pytest.raises(Exception):
        bad = dict(g); bad["edge_pose"] = g["edge_pose"].copy(); bad["edge_pose"][0] = 999
        Optimizer.BundleAdjustmentClient(bad, 1, ctx=ctx)


@pytest.mark.parametrize("pcg_tol", [0.0, 1e-13])
def test_gba_config5_matches_the_oracle_at_full_size(ctx, pcg_tol):
    """BASELINE config 5 itself (2000 KF / 200k points / 1.82 M edges, 5 LM iterations) against the committed result of
    the CPU oracle at the same size (tests/golden/gba_config5.npz, made by tools/make_golden_gba.py: block-sparse exact
    Cholesky per trial, its first solve checked against scipy's normal equations to 4e-15).  Default solver (PCG with
    the 125-aggregate coarse level, relative residual 1e-8) and the PCG converged to 1e-13: keyframes within the
    contract's 1e-5, landmarks within 1e-5 m, identical iteration and trial counts, chi2 to 1e-9 relative."""
    z = np.load(os.path.join(G, "gba_config5.npz"))
    g = synth.gba_graph()
    assert len(g["edge_pose"]) == int(z["n_edges"])
    r = Optimizer.MapFusionGBA(g, 5, ctx=ctx, pcg_tol=pcg_tol)
    assert r["pcg_iterations"] > 0 and r["pcg_fallbacks"] == 0
    d = pose_delta(r["poses"], z["poses"]).max()
    assert d <= (TOL if pcg_tol == 0.0 else 1e-8), d
    st = int(z["point_stride"])
    dp = np.abs(r["points"][::st] - z["points_sub"]).max()
    assert dp <= (1e-5 if pcg_tol == 0.0 else 1e-7), dp
    assert np.abs(r["points"].sum(0) - z["points_sum"]).max() <= 1e-5 * len(r["points"]) ** 0.5
    assert [r["iterations_done"], r["trials"]] == list(z["iterations"])
    assert np.isclose(r["chi2_initial"], z["chi2"][0], rtol=1e-11) and np.isclose(r["chi2_final"], z["chi2"][1], rtol=1e-9)
    assert np.isclose(r["lambda_final"], float(z["lambda_final"]), rtol=1e-6)


def test_gba_config5_optimize20_matches_the_oracle(ctx):
    """The reference's own call on config 5: optimize(20) (cslam/conf/config.yaml:129).  g2o's stop criterion (three iterations
    in a row gaining < 0.1 % of chi2, optimization_algorithm_levenberg.cpp:154-161) ends it after 9 iterations in the oracle;
    the HIP path must stop at the same iteration with the same state (lambda has fallen to 0.04 by then, the hardest reduced
    systems of the run for the PCG)."""
    z = np.load(os.path.join(G, "gba_config5.npz"))
    g = synth.gba_graph()
    r = Optimizer.MapFusionGBA(g, 20, ctx=ctx)
    assert [r["iterations_done"], r["trials"]] == list(z["iterations20"]) and r["pcg_fallbacks"] == 0
    d = pose_delta(r["poses"], z["poses20"]).max()
    dp = np.abs(r["points"][::int(z["point_stride"])] - z["points20_sub"]).max()
    assert d <= TOL and dp <= 1e-5, (d, dp)
    assert np.isclose(r["chi2_final"], z["chi2_20"][1], rtol=1e-9) and np.isclose(r["lambda_final"], float(z["lambda20"]), rtol=1e-6)


def test_full_gba_properties(ctx):
    """BASELINE config 5 at full size (2000 KF / 200k points), beyond the 5 iterations of the golden comparison above:
    size-independent properties: chi2 decreases monotonically over accepted iterations, the fixed
    keyframe is untouched, the result is reproducible to rounding, and poses move toward ground truth."""
    g = synth.gba_graph()
    r3 = Optimizer.MapFusionGBA(g, 3, ctx=ctx)
    r6 = Optimizer.MapFusionGBA(g, 6, ctx=ctx)
    assert r3["chi2_final"] < 0.2 * r3["chi2_initial"] and r6["chi2_final"] <= r3["chi2_final"] * (1 + 1e-12)
    assert (r6["poses"][0] == g["poses"][0]).all()
    again = Optimizer.MapFusionGBA(g, 3, ctx=ctx)
    assert (again["poses"] == r3["poses"]).all() and (again["points"] == r3["points"]).all()      # fixed summation orders: bit-reproducible
    e0 = pose_delta(g["poses"], g["gt_poses"]); e1 = pose_delta(r6["poses"], g["gt_poses"])
    assert np.median(e1) < 0.25 * np.median(e0)


def test_stop_flag_raised_mid_solve(ctx):
    """*pbStopFlag set by another thread WHILE config 5 is being optimised (sparse_optimizer.cpp:376 tests it before every
    iteration, optimization_algorithm_levenberg.cpp:149 inside the trial loop): the solve ends early with `stopped`, and its
    result is the result of a run asked for exactly the iterations it completed."""
    import threading, time
    g = synth.gba_graph()
    Optimizer.MapFusionGBA(g, 1, ctx=ctx)                                   # allocations and graph capture out of the way
    flag = np.zeros(1, np.uint8)
    N = 8                                                                   # (g2o's own stop criterion ends this graph after 9)
    t_full = time.perf_counter(); full = Optimizer.MapFusionGBA(g, N, ctx=ctx); t_full = time.perf_counter() - t_full
    assert not full["stopped"] and full["iterations_done"] == N

    def raiser():
        time.sleep(0.45 * t_full)
        flag[0] = 1
    th = threading.Thread(target=raiser); th.start()
    r = Optimizer.MapFusionGBA(g, N, pbStopFlag=flag, ctx=ctx)
    th.join()
    assert r["stopped"] and 0 < r["iterations_done"] < N, (r["stopped"], r["iterations_done"])
    k = r["iterations_done"]
    ref = Optimizer.MapFusionGBA(g, k, ctx=ctx)
    if r["trials"] == ref["trials"]:                                       # the flag did not cut a retry loop short
        assert (r["poses"] == ref["poses"]).all() and (r["points"] == ref["points"]).all() and r["chi2_final"] == ref["chi2_final"]
    else:
        assert r["chi2_final"] <= r["chi2_initial"]
    with pytest.raises(TypeError):
        Optimizer.MapFusionGBA(g, 1, pbStopFlag=True, ctx=ctx)              # a Python bool can never be raised by another thread
    with pytest.raises(TypeError):
        Optimizer.MapFusionGBA(g, 1, pbStopFlag=np.zeros(1, np.int32), ctx=ctx)


def test_larger_map_widens_the_coarse_aggregates(ctx):
    """5000 keyframes / 300k points (2.7 M edges), beyond every BASELINE config: the coarse level of the preconditioner
    switches to 32-keyframe aggregates (1099 coarse unknowns instead of 2191) so that its inversion still fits inside an LM
    trial.  No oracle at this size: chi2 must drop, no trial may fall back to the dense solver, and the run must repeat bit
    for bit."""
    g = synth.gba_graph(n_kf=5000, n_points=300000, n_agents=3, seed=5)
    a = Optimizer.MapFusionGBA(g, 3, ctx=ctx)
    b = Optimizer.MapFusionGBA(g, 3, ctx=ctx)
    assert a["chi2_final"] < 0.2 * a["chi2_initial"] and a["pcg_iterations"] > 0 and a["pcg_fallbacks"] == 0
    assert (a["poses"] == b["poses"]).all() and (a["points"] == b["points"]).all()
    e0 = pose_delta(g["poses"], g["gt_poses"]); e1 = pose_delta(a["poses"], g["gt_poses"])
    assert np.median(e1) < 0.5 * np.median(e0)


@pytest.mark.parametrize("coarse", ["1", "0"])
def test_iterative_solver_forced_on_small_graphs(coarse):
    """The reduced-camera PCG (cluster-Jacobi, HIP graph) normally starts above 256 free keyframes, where the oracle's
    dense solve is too slow to compare with.  A child process with CCM_BA_DENSE_MAX=0 runs the oracle-sized graphs
    through it: odd cluster sizes (13, 60 and 239 free keyframes), robust and two-stage schedules.  coarse = 1: the
    239-keyframe graph also runs the second preconditioner level (coarse inverse pipelined on the side stream); 0: the
    cluster level alone."""
    import subprocess, sys
    code = r'''
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer, pose_delta
from oracle import oracle_py as O
ctx = _lib.Context(0)
g = synth.local_ba_graph()
r = Optimizer.LocalBundleAdjustmentClient(g, ctx=ctx); ref = O.ba_solve(g, 5, float(np.float32(np.sqrt(5.991))), 10)
assert r["pcg_iterations"] > 0, "the dense path ran"
assert pose_delta(r["poses"], ref["poses"]).max() <= 1e-5 and (r["outlier"] == ref["outlier"]).all()
assert r["iterations_done"] == ref["iterations_done"] and r["trials"] == ref["trials"]
for kf, pts, its in ((60, 3000, 8), (240, 20000, 4)):
    g = synth.gba_graph(n_kf=kf, n_points=pts, n_agents=3, seed=kf)
    r = Optimizer.MapFusionGBA(g, its, ctx=ctx); ref = O.ba_solve(g, its, float(np.float32(np.sqrt(5.99))))
    assert r["pcg_iterations"] > 0 and r["pcg_fallbacks"] == 0
    d = pose_delta(r["poses"], ref["poses"]).max()
    assert d <= 1e-5 and r["iterations_done"] == ref["iterations_done"], (kf, d)
    strict = Optimizer.MapFusionGBA(g, its, ctx=ctx, pcg_tol=1e-13)
    assert pose_delta(strict["poses"], ref["poses"]).max() <= 1e-9
# a stretch of keyframes that starts from ONE position: the scale columns of their aggregates vanish (the coarse unknowns stay inert),
# and a map whose keyframes all start from one position: no scale column at all
for lo, hi in ((100, 160), (0, 240)):
    g = synth.gba_graph(n_kf=240, n_points=20000, n_agents=3, seed=7)
    p = np.array(g["poses"], dtype=np.float64); p[lo:hi, 4:7] = p[lo, 4:7]; g = dict(g); g["poses"] = p
    r = Optimizer.MapFusionGBA(g, 3, ctx=ctx, pcg_tol=1e-13); ref = O.ba_solve(g, 3, float(np.float32(np.sqrt(5.99))))
    assert r["pcg_iterations"] > 0 and r["pcg_fallbacks"] == 0, (lo, hi, r["pcg_fallbacks"])
    assert r["trials"] == ref["trials"] and pose_delta(r["poses"], ref["poses"]).max() <= 1e-7, (lo, hi, pose_delta(r["poses"], ref["poses"]).max())
print("ok")
'''
    env = dict(os.environ, CCM_BA_DENSE_MAX="0", CCM_PCG_COARSE=coarse, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]


def test_rejected_trial_after_a_linearisation_without_hpl():
    """From the second LM iteration on the landmark linearisation leaves Hpl unwritten (k_ba_lin_landmark MODE 2): the Schur
    product, the reduced right-hand side and the back-substitution work from Z and ce.  A REJECTED trial needs Z for another
    lambda, and the host rebuilds Hpl for it.  The synthetic graphs accept every trial, so a test switch rejects the first
    trial of iteration 2; the run must repeat bit for bit what a build that always keeps Hpl (CCM_BA_KEEP_HPL=1) computes,
    on a dense-solver graph and on a PCG graph."""
    import subprocess, sys
    code = r'''
import hashlib, numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.optimizer import Optimizer
ctx = _lib.Context(0)
for kf, pts in ((60, 3000), (400, 30000)):
    g = synth.gba_graph(n_kf=kf, n_points=pts, n_agents=3, seed=kf)
    r = Optimizer.MapFusionGBA(g, 6, ctx=ctx)
    assert r["trials"] == r["iterations_done"] + 1, (r["trials"], r["iterations_done"])
    assert r["chi2_final"] < 0.1 * r["chi2_initial"]
    print(hashlib.sha1(np.ascontiguousarray(r["poses"]).tobytes() + np.ascontiguousarray(r["points"]).tobytes()).hexdigest())
'''
    outs = []
    # ... and the LM loop's look-ahead (the keyframes' side of the next linearisation enqueued behind a trial's chi2, dropped when the
    # trial is rejected) must not change a bit either: third run without it
    for extra in ({}, {"CCM_BA_KEEP_HPL": "1"}, {"CCM_BA_NO_LOOKAHEAD": "1"}):
        env = dict(os.environ, CCM_BA_TEST_REJECT_AT="2", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        outs.append(out.stdout.split())
    assert len(outs[0]) == 2 and outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("dense_max", ["0", "100000"])
def test_sharded_gba_two_ranks_on_one_gpu(dense_max, tmp_path):
    """Multi-GPU global BA rehearsed on one GPU: two processes = two ranks, each keeps the landmarks of its range,
    the partial reduced camera systems are summed through the shared-memory transport (RCCL's place).  Both ranks must
    return the same poses and points, and they must agree with the CPU oracle.  dense_max 0 = PCG, large = the dense
    solve; the solve is replicated and rank 0's increment is the one every rank applies (one small all-reduce per
    trial), so the ranks are bit-identical with either solver."""
    import subprocess, sys, uuid
    code = r'''
import sys
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth, dist
from motioncheck_ccm_slam_amd.optimizer import Optimizer, pose_delta
from tests.support import shm_transport
rank, name, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
ctx = _lib.Context(0)
shm_transport.attach(ctx, name, rank, 2)
g = synth.gba_graph(n_kf=240, n_points=20000, n_agents=3, seed=240)
r = Optimizer.MapFusionGBA(g, 4, ctx=ctx)
if rank == 0:
    from oracle import oracle_py as O
    ref = O.ba_solve(g, 4, float(np.float32(np.sqrt(5.99))))
    d = pose_delta(r["poses"], ref["poses"]).max()
    dp = np.abs(r["points"] - ref["points"]).max()
    assert d <= 1e-5 and r["iterations_done"] == ref["iterations_done"] and r["trials"] == ref["trials"], (d, r["iterations_done"], ref["iterations_done"], r["trials"], ref["trials"])
    assert dp <= 1e-5, (dp, d, int(np.abs(r["points"] - ref["points"]).max(axis=1).argmax()), r["chi2_final"], ref["chi2_final"])
np.savez(out, poses=r["poses"], points=r["points"], pairs=r["schur_pairs"])
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "/ccm_test_" + uuid.uuid4().hex[:12]
    env = dict(os.environ, PYTHONPATH=root, CCM_BA_DENSE_MAX=dense_max)
    files = [str(tmp_path / ("rank%d.npz" % rk)) for rk in (0, 1)]
    procs = [subprocess.Popen([sys.executable, "-c", code, str(rk), name, files[rk]], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rk in (0, 1)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=600)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, o, e in outs:
        assert rc == 0 and o.strip().endswith("ok"), o[-1500:] + e[-1500:]
    a, b = np.load(files[0]), np.load(files[1])
    assert (a["poses"] == b["poses"]).all() and (a["points"] == b["points"]).all(), \
        "the ranks disagree on the result: max |d poses| %.3e, max |d points| %.3e" % (np.abs(a["poses"] - b["poses"]).max(), np.abs(a["points"] - b["points"]).max())
    assert int(a["pairs"]) > 0 and int(b["pairs"]) > 0 and int(a["pairs"]) != int(b["pairs"])      # each rank enumerated its own share


def test_sharded_gba_stop_flag_on_one_rank_only(tmp_path):
    """Two ranks on one GPU (shared-memory transport); only rank 1's stop flag is set.  The decision is collective (the flag
    rides on the chi2 all-reduce), so BOTH ranks stop before the first iteration instead of rank 0 waiting for ever in the
    next all-reduce."""
    import subprocess, sys, uuid
    code = r'''
import sys
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth, dist
from motioncheck_ccm_slam_amd.optimizer import Optimizer
from tests.support import shm_transport
rank, name = int(sys.argv[1]), sys.argv[2]
ctx = _lib.Context(0)
shm_transport.attach(ctx, name, rank, 2)
g = synth.gba_graph(n_kf=120, n_points=8000, n_agents=3, seed=120)
flag = np.full(1, 1 if rank == 1 else 0, np.uint8)
r = Optimizer.MapFusionGBA(g, 4, pbStopFlag=flag, ctx=ctx)
assert r["stopped"] and r["iterations_done"] == 0 and (r["poses"] == g["poses"]).all(), (rank, r["stopped"], r["iterations_done"])
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "/ccm_test_" + uuid.uuid4().hex[:12]
    env = dict(os.environ, PYTHONPATH=root)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(rk), name], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rk in (0, 1)]
    try:
        outs = [p.communicate(timeout=300) + (p.returncode,) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for o, e, rc in outs:
        assert rc == 0 and o.strip().endswith("ok"), o[-1500:] + e[-1500:]


def test_rccl_calls_with_one_rank_communicator():
    """The RCCL leg itself (ncclGetUniqueId, ncclCommInitRank, in-place ncclAllReduce f64 sum/max and u8 max on the
    context's stream) with a communicator of ONE rank, in a process that has torch loaded like bench.py: the reduced
    system goes through RCCL and the result must equal the solve without a communicator, bit for bit."""
    import subprocess, sys
    code = r'''
import numpy as np, torch
torch.cuda.init()
from motioncheck_ccm_slam_amd import _lib, synth, dist
from motioncheck_ccm_slam_amd.optimizer import Optimizer
g = synth.gba_graph(n_kf=240, n_points=20000, n_agents=3, seed=240)
ctx0 = _lib.Context(0)
r0 = Optimizer.MapFusionGBA(g, 3, ctx=ctx0)
ctx = _lib.Context(0)
dist.init_comm(ctx, 0, 1)
r = Optimizer.MapFusionGBA(g, 3, ctx=ctx)
assert (r["poses"] == r0["poses"]).all() and (r["points"] == r0["points"]).all()
assert r["iterations_done"] == r0["iterations_done"] and r["chi2_final"] == r0["chi2_final"]
ctx.close(); ctx0.close()
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, CCM_BA_DENSE_MAX="0", CCM_COMM_RCCL_SINGLE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]
