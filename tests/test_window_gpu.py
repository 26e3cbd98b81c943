"""Windowed matching: batched Frame::GetFeaturesInArea with distances and
ORBmatcher::SearchByProjection(Frame&, map points, th) on the GPU vs the CPU oracle.  The frame is a real
extraction of a synthetic image; the map points are its own features re-projected with noise."""
import numpy as np
import pytest

from motioncheck_ccm_slam_amd import synth
from motioncheck_ccm_slam_amd.matcher import FrameGridView, ORBmatcher
from motioncheck_ccm_slam_amd.orb import ORBextractor

pytestmark = pytest.mark.gpu


def _frame(ctx, f=0):
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    kps, desc = ex(synth.frame(f))
    return FrameGridView(kps["x"], kps["y"], kps["octave"], desc), ex.GetScaleFactors(), kps, desc


def test_features_in_area(ctx, oracle):
    fr, sf, kps, desc = _frame(ctx)
    m = ORBmatcher(0.8, ctx=ctx)
    rng = np.random.default_rng(0)
    nq = 400
    x = rng.uniform(-30, 780, nq).astype("f4"); y = rng.uniform(-30, 510, nq).astype("f4")
    r = rng.choice([2.5, 4.0, 10.0, 25.0, 60.0], nq).astype("f4")
    mn = rng.integers(-1, 6, nq); mx = np.where(rng.random(nq) < 0.3, -1, mn + rng.integers(0, 3, nq))
    qd = desc[rng.integers(0, len(desc), nq)]
    r[5] = -1.0                                                    # skipped query
    ci, cd, cn = m.FeaturesInArea(fr, x, y, r, mn, mx, qd, cap=1100)
    assert cn[5] == 0
    for q in range(nq):
        if r[q] < 0:
            continue
        ref = oracle.features_in_area(fr.kx, fr.ky, fr.oct, fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, x[q], y[q], r[q], mn[q], mx[q])
        assert cn[q] == len(ref) and (ci[q, :cn[q]] == ref).all(), q
        for k in range(0, cn[q], 7):
            assert cd[q, k] == oracle.distance(qd[q], desc[ci[q, k]])
    assert cn.max() > 50 and (cn == 0).sum() > 0


@pytest.mark.parametrize("th", [1.0, 3.0])
def test_search_by_projection(ctx, oracle, th):
    fr, sf, kps, desc = _frame(ctx, 1)
    n = len(fr.kx)
    rng = np.random.default_rng(1)
    # map points: 1500 = features of the frame (noisy projection, noisy descriptor) + 300 unrelated
    src = rng.integers(0, n, 1200)
    flips = np.packbits(rng.random((1200, 256)) < 0.05, axis=1, bitorder="little")
    mp_desc = np.concatenate([desc[src] ^ flips, rng.integers(0, 256, (300, 32), dtype=np.uint8)])
    px = np.concatenate([fr.kx[src] + rng.normal(0, 1.5, 1200), rng.uniform(0, 752, 300)]).astype("f4")
    py = np.concatenate([fr.ky[src] + rng.normal(0, 1.5, 1200), rng.uniform(0, 480, 300)]).astype("f4")
    level = np.concatenate([np.clip(fr.oct[src] + rng.integers(0, 2, 1200), 0, 7), rng.integers(0, 8, 300)])
    view_cos = rng.uniform(0.99, 1.0, 1500).astype("f4")
    in_view = rng.random(1500) < 0.9
    has_obs = rng.random(1500) < 0.95
    occupied = rng.random(n) < 0.2                               # features already matched by the previous tracking stage
    m = ORBmatcher(0.8, ctx=ctx)
    nm, match, occ = m.SearchByProjection(fr, sf, in_view, level, view_cos, px, py, mp_desc, has_obs, occupied, th)
    rn, rmatch, rocc = oracle.search_by_projection(fr.kx, fr.ky, fr.oct, desc, fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf, in_view, level,
                                                   view_cos, px, py, mp_desc, has_obs, occupied, th, 0.8)
    assert nm == rn and (match == rmatch).all() and (occ == rocc).all()
    assert nm > 300
    good = match >= 0
    assert (match[good] < 1200).mean() > 0.95                   # matches go to the true correspondences
    # nothing in view -> nothing matched
    nm, match, _ = m.SearchByProjection(fr, sf, np.zeros(1500, bool), level, view_cos, px, py, mp_desc, has_obs, occupied, th)
    assert nm == 0 and (match == -1).all()


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_projection_frame_to_frame(ctx, oracle, check_ori):
    """TrackWithMotionModel's matcher: last frame = frame 2, current = the same image shifted by a few pixels."""
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    img = synth.frame(2)
    cur_img = np.roll(img, (3, -5), axis=(0, 1))
    k1, d1 = ex(img); k2, d2 = ex(cur_img)
    cur = FrameGridView(k2["x"], k2["y"], k2["octave"], d2)
    sf = ex.GetScaleFactors()
    rng = np.random.default_rng(4)
    n_last = len(k1)
    valid = rng.random(n_last) < 0.85
    u = (k1["x"] - 5 + rng.normal(0, 1.0, n_last)).astype("f4"); v = (k1["y"] + 3 + rng.normal(0, 1.0, n_last)).astype("f4")
    valid &= (u >= 0) & (u <= 752) & (v >= 0) & (v <= 480)
    has_obs = rng.random(n_last) < 0.9                             # a few map points without observations: overwrites happen
    occupied = np.zeros(len(k2), bool)
    m = ORBmatcher(0.9, check_ori, ctx=ctx)
    for th in (7.0, 15.0, 40.0):                                   # 40: far more candidates per point than the 16 a thread keeps in registers
        nm, match, occ = m.SearchByProjectionFrame(cur, k2["angle"], sf, valid, u, v, k1["octave"], k1["angle"], d1, has_obs, occupied, th)
        rn, rmatch, rocc = oracle.search_by_projection_frame(cur.kx, cur.ky, cur.oct, d2, k2["angle"], cur.min_x, cur.min_y, cur.inv_w, cur.inv_h,
                                                             sf, valid, u, v, k1["octave"], k1["angle"], d1, has_obs, occupied, th, check_ori)
        assert nm == rn and (match == rmatch).all() and (occ == rocc).all()
        assert nm > 200


def test_search_for_initialization(ctx, oracle):
    """Monocular initialisation matcher: frame 3 against a shifted copy, window 100 px, level-0 features only."""
    ex = ORBextractor(2000, 1.2, 8, 20, 7, ctx=ctx)                # the initialisation extractor has 2x features
    img = synth.frame(3)
    k1, d1 = ex(img); k2, d2 = ex(np.roll(img, (6, 9), axis=(0, 1)))
    f2 = FrameGridView(k2["x"], k2["y"], k2["octave"], d2)
    prev = np.stack([k1["x"], k1["y"]], 1)
    for ori in (True, False):
        m = ORBmatcher(0.9, ori, ctx=ctx)
        nm, m12, pm = m.SearchForInitialization(k1["octave"], d1, k1["angle"], f2, k2["angle"], prev, 100)
        rn, rm12, rpm = oracle.search_for_initialization(k1["octave"], d1, k1["angle"], f2.kx, f2.ky, f2.oct, d2, k2["angle"], f2.min_x, f2.min_y,
                                                         f2.inv_w, f2.inv_h, prev, 100, 0.9, ori)
        assert nm == rn and (m12 == rm12).all() and (pm == rpm).all()
        assert nm > 100 and (m12[k1["octave"] > 0] == -1).all()


@pytest.mark.parametrize("chi2", [True, False])
def test_fuse_select(ctx, oracle, chi2):
    fr, sf, kps, desc = _frame(ctx, 4)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    is2 = ex.GetInverseScaleSigmaSquares()
    n = len(fr.kx)
    rng = np.random.default_rng(8)
    src = rng.integers(0, n, 900)
    flips = np.packbits(rng.random((900, 256)) < 0.04, axis=1, bitorder="little")
    mp_desc = np.concatenate([desc[src] ^ flips, rng.integers(0, 256, (200, 32), dtype=np.uint8)])
    u = np.concatenate([fr.kx[src] + rng.normal(0, 1.2, 900), rng.uniform(0, 752, 200)]).astype("f4")
    v = np.concatenate([fr.ky[src] + rng.normal(0, 1.2, 900), rng.uniform(0, 480, 200)]).astype("f4")
    level = np.concatenate([np.clip(fr.oct[src] + rng.integers(0, 2, 900), 0, 7), rng.integers(0, 8, 200)])
    valid = rng.random(1100) < 0.9
    m = ORBmatcher(ctx=ctx)
    for th in (3.0, 4.0):
        bi, bd = m.FuseSelect(fr, sf, is2, valid, u, v, level, mp_desc, th, chi2)
        rbi, rbd = oracle.fuse_select(fr.kx, fr.ky, fr.oct, desc, fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf, is2, valid, u, v, level, mp_desc, th, chi2)
        assert (bi == rbi).all() and (bd == rbd).all()
        assert (bi[:900] >= 0).sum() > 400 and (bi[~valid] == -1).all()


def test_fuse_select_batch_equals_keyframe_by_keyframe(ctx, oracle):
    """ccm_fuse_select_batch (round 3): the map points of one keyframe projected into K neighbours, all in one launch, against K
    sequential calls and against the oracle per keyframe; an empty keyframe and a keyframe without map points ride along."""
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    is2 = ex.GetInverseScaleSigmaSquares()
    rng = np.random.default_rng(21)
    kfs, per_kf, descs = [], [], []
    sf = None
    for k, seed in enumerate((4, 5, 6, 7, 9)):
        fr, sf, kps, desc = _frame(ctx, seed)
        n = len(fr.kx)
        nm = (700, 0, 1500, 300, 40)[k]
        src = rng.integers(0, n, nm)
        flips = np.packbits(rng.random((nm, 256)) < 0.05, axis=1, bitorder="little")
        mp_desc = (desc[src] ^ flips).astype(np.uint8)
        u = (fr.kx[src] + rng.normal(0, 1.5, nm)).astype("f4"); v = (fr.ky[src] + rng.normal(0, 1.5, nm)).astype("f4")
        level = np.clip(fr.oct[src] + rng.integers(0, 2, nm), 0, 7).astype("i4")
        valid = rng.random(nm) < 0.85
        kfs.append(fr); descs.append(desc); per_kf.append((valid, u, v, level, mp_desc))
    empty = FrameGridView(np.zeros(0, "f4"), np.zeros(0, "f4"), np.zeros(0, "i4"), np.zeros((0, 32), np.uint8))
    kfs.insert(2, empty); descs.insert(2, np.zeros((0, 32), np.uint8))
    per_kf.insert(2, (np.ones(25, bool), rng.uniform(0, 752, 25).astype("f4"), rng.uniform(0, 480, 25).astype("f4"), rng.integers(0, 8, 25).astype("i4"),
                      rng.integers(0, 256, (25, 32), dtype=np.uint8)))
    m = ORBmatcher(ctx=ctx)
    for chi2 in (True, False):
        got = m.FuseSelectBatch(kfs, sf, is2, per_kf, 3.0, chi2)
        assert len(got) == len(kfs)
        for k, (fr, t) in enumerate(zip(kfs, per_kf)):
            bi, bd = m.FuseSelect(fr, sf, is2, *t, 3.0, chi2)
            assert (got[k][0] == bi).all() and (got[k][1] == bd).all(), k
            if len(fr.kx):
                rbi, rbd = oracle.fuse_select(fr.kx, fr.ky, fr.oct, descs[k], fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf, is2, *t, 3.0, chi2)
                assert (got[k][0] == rbi).all() and (got[k][1] == rbd).all(), k
            else:
                assert (got[k][0] == -1).all()
        assert sum(int((g[0] >= 0).sum()) for g in got) > 800


def _noisy_points(fr, desc, rng, n_rel, n_rand, sigma=1.2, flip=0.04):
    """Map points = features of frame `fr` re-projected with noise (+ unrelated ones): desc, u, v, level."""
    n = len(fr.kx)
    src = rng.integers(0, n, n_rel)
    flips = np.packbits(rng.random((n_rel, 256)) < flip, axis=1, bitorder="little")
    mp_desc = np.concatenate([desc[src] ^ flips, rng.integers(0, 256, (n_rand, 32), dtype=np.uint8)])
    u = np.concatenate([fr.kx[src] + rng.normal(0, sigma, n_rel), rng.uniform(0, 752, n_rand)]).astype("f4")
    v = np.concatenate([fr.ky[src] + rng.normal(0, sigma, n_rel), rng.uniform(0, 480, n_rand)]).astype("f4")
    level = np.concatenate([np.clip(fr.oct[src] + rng.integers(0, 2, n_rel), 0, 7), rng.integers(0, 8, n_rand)])
    return mp_desc, u, v, level, src


def test_search_by_sim3(ctx, oracle):
    """Two keyframes = two extractions; the 'map points' of each are noisy copies of the OTHER keyframe's features so
    that the two directions agree on many pairs and disagree on some."""
    f1, sf, k1, d1 = _frame(ctx, 5)
    f2, _, k2, d2 = _frame(ctx, 6)
    rng = np.random.default_rng(11)
    n1, n2 = len(f1.kx), len(f2.kx)
    # feature i1 of KF1 carries a map point that projects near feature p12[i1] of KF2, and the reverse for KF2
    p12 = rng.integers(0, n2, n1)
    fl = np.packbits(rng.random((n1, 256)) < 0.05, axis=1, bitorder="little")
    mp1 = d2[p12] ^ fl
    u1 = (f2.kx[p12] + rng.normal(0, 1.0, n1)).astype("f4"); v1 = (f2.ky[p12] + rng.normal(0, 1.0, n1)).astype("f4")
    l1 = np.clip(f2.oct[p12] + rng.integers(0, 2, n1), 0, 7)
    p21 = rng.integers(0, n1, n2)
    back = rng.random(n2) < 0.7                                   # 70 %: KF2's point is the partner chosen by KF1's feature
    inv = np.full(n2, -1); inv[p12] = np.arange(n1)
    p21 = np.where(back & (inv >= 0), inv, p21)
    fl2 = np.packbits(rng.random((n2, 256)) < 0.05, axis=1, bitorder="little")
    mp2 = d1[p21] ^ fl2
    u2 = (f1.kx[p21] + rng.normal(0, 1.0, n2)).astype("f4"); v2 = (f1.ky[p21] + rng.normal(0, 1.0, n2)).astype("f4")
    l2 = np.clip(f1.oct[p21] + rng.integers(0, 2, n2), 0, 7)
    val1 = rng.random(n1) < 0.9; val2 = rng.random(n2) < 0.9
    m = ORBmatcher(ctx=ctx)
    for th in (7.5, 3.0):
        n, m12 = m.SearchBySim3(f1, sf, f2, sf, val1, u1, v1, l1, mp1, val2, u2, v2, l2, mp2, th)
        rn, r12 = oracle.search_by_sim3(f1, sf, f2, sf, val1, u1, v1, l1, mp1, val2, u2, v2, l2, mp2, th)
        assert n == rn and (m12 == r12).all()
        assert n > 100 and (m12 < 0).sum() > 100


def test_search_by_projection_sim3(ctx, oracle):
    fr, sf, kps, desc = _frame(ctx, 7)
    rng = np.random.default_rng(12)
    mp_desc, u, v, level, src = _noisy_points(fr, desc, rng, 1400, 200)      # several points compete for the same feature
    valid = rng.random(1600) < 0.9
    observed = rng.random(1600) < 0.15
    matched = rng.random(len(fr.kx)) < 0.2
    m = ORBmatcher(ctx=ctx)
    for th in (10.0, 4.0):
        n, bi, mt = m.SearchByProjectionSim3(fr, sf, valid, u, v, level, mp_desc, observed, matched, th)
        rn, rbi, rmt = oracle.search_by_projection_sim3(fr, sf, valid, u, v, level, mp_desc, observed, matched, th)
        assert n == rn and (bi == rbi).all() and (mt == rmt).all()
        assert n > 200 and (mt.sum() - matched.sum()) == n


def test_search_by_projection_sim3_batch_equals_keyframe_by_keyframe(ctx, oracle):
    """ccm_search_by_projection_sim3_batch (round 3): the order-dependent acceptance (k_window_greedy) for K keyframes in one launch,
    one workgroup per keyframe, against K sequential calls and the oracle; a keyframe without features and one without map points
    ride along, and one keyframe has far more points than features (many points competing for a feature: the wave-by-wave rounds)."""
    rng = np.random.default_rng(31)
    kfs, per_kf = [], []
    sf = None
    for k, (seed, n_rel, n_rand) in enumerate(((4, 1400, 200), (5, 0, 0), (6, 6000, 500), (7, 300, 50), (9, 900, 0))):
        fr, sf, kps, desc = _frame(ctx, seed)
        nm = n_rel + n_rand
        if nm:
            mp_desc, u, v, level, src = _noisy_points(fr, desc, rng, n_rel, n_rand)
        else:
            mp_desc, u, v, level = np.zeros((0, 32), np.uint8), np.zeros(0, "f4"), np.zeros(0, "f4"), np.zeros(0, "i4")
        valid = rng.random(nm) < 0.9
        observed = rng.random(nm) < 0.15
        matched = rng.random(len(fr.kx)) < 0.2
        kfs.append(fr); per_kf.append((valid, u, v, level, mp_desc, observed, matched))
    empty = FrameGridView(np.zeros(0, "f4"), np.zeros(0, "f4"), np.zeros(0, "i4"), np.zeros((0, 32), np.uint8))
    kfs.insert(3, empty)
    per_kf.insert(3, (np.ones(25, bool), rng.uniform(0, 752, 25).astype("f4"), rng.uniform(0, 480, 25).astype("f4"), rng.integers(0, 8, 25).astype("i4"),
                      rng.integers(0, 256, (25, 32), dtype=np.uint8), np.zeros(25, bool), np.zeros(0, bool)))
    m = ORBmatcher(ctx=ctx)
    for th in (10.0, 4.0):
        got = m.SearchByProjectionSim3Batch(kfs, sf, per_kf, th)
        assert len(got) == len(kfs)
        for k, (fr, t) in enumerate(zip(kfs, per_kf)):
            n, bi, mt = m.SearchByProjectionSim3(fr, sf, *t, th)
            assert got[k][0] == n and (got[k][1] == bi).all() and (got[k][2] == mt[:len(fr.kx)]).all(), k
            if len(fr.kx) and len(t[0]):
                rn, rbi, rmt = oracle.search_by_projection_sim3(fr, sf, *t, th)
                assert got[k][0] == rn and (got[k][1] == rbi).all() and (got[k][2] == rmt).all(), k
            else:
                assert got[k][0] == 0 and (got[k][1] == -1).all()
        assert sum(g[0] for g in got) > 1000


def test_search_by_projection_keyframe_overload(ctx, oracle):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): has_obs all ones, ORBdist 64."""
    fr, sf, kps, desc = _frame(ctx, 8)
    rng = np.random.default_rng(13)
    mp_desc, u, v, level, src = _noisy_points(fr, desc, rng, 800, 100, flip=0.12)
    nmp = 900
    valid = rng.random(nmp) < 0.9
    kf_angle = rng.uniform(0, 360, nmp).astype("f4")
    kf_angle[:800] = (kps["angle"][src] + rng.normal(0, 3, 800)) % 360
    occ = rng.random(len(fr.kx)) < 0.3
    ones = np.ones(nmp, np.uint8)
    m = ORBmatcher(0.9, True, ctx=ctx)
    for dist in (64, 100):
        n, match, o2 = m.SearchByProjectionFrame(fr, kps["angle"], sf, valid, u, v, level, kf_angle, mp_desc, ones, occ, 10.0, orb_dist=dist)
        rn, rmatch, ro2 = oracle.search_by_projection_frame(fr.kx, fr.ky, fr.oct, desc, kps["angle"], fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf,
                                                            valid, u, v, level, kf_angle, mp_desc, ones, occ, 10.0, True, orb_dist=dist)
        assert n == rn and (match == rmatch).all() and (o2 == ro2).all()
    assert n > 100


def test_search_for_triangulation(ctx, oracle):
    f1, sf, k1, d1 = _frame(ctx, 9)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    sig2 = ex.GetScaleSigmaSquares()
    rng = np.random.default_rng(14)
    n1 = len(f1.kx)
    # image 2 = image 1 shifted by a pure x translation of the camera: the epipolar lines are the image rows
    n2 = n1 + 150
    perm = rng.permutation(n1)
    disp = rng.uniform(2, 40, n1).astype("f4")
    x2 = np.concatenate([f1.kx[perm] - disp, rng.uniform(0, 752, 150)]).astype("f4")
    y2 = np.concatenate([f1.ky[perm] + rng.normal(0, 0.8, n1), rng.uniform(0, 480, 150)]).astype("f4")
    fl = np.packbits(rng.random((n1, 256)) < 0.06, axis=1, bitorder="little")
    d2 = np.concatenate([d1[perm] ^ fl, rng.integers(0, 256, (150, 32), dtype=np.uint8)])
    a2 = np.concatenate([(k1["angle"][perm] + rng.normal(0, 4, n1)) % 360, rng.uniform(0, 360, 150)]).astype("f4")
    o2 = np.concatenate([f1.oct[perm], rng.integers(0, 8, 150)])
    node1 = rng.integers(0, 60, n1); node1[rng.random(n1) < 0.03] = -1
    node2 = np.concatenate([np.where(rng.random(n1) < 0.9, node1[perm], rng.integers(0, 60, n1)), rng.integers(0, 60, 150)])
    has1 = rng.random(n1) < 0.4; has2 = rng.random(n2) < 0.3
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], "f4") + rng.normal(0, 1e-4, (3, 3)).astype("f4")   # [t]x for t = (1,0,0), perturbed
    ex_, ey_ = 300.0, 240.0                                        # an epipole inside the image: features near it are rejected
    for ori in (True, False):
        m = ORBmatcher(0.6, ori, ctx=ctx)
        n, m12 = m.SearchForTriangulation(d1, node1, has1, f1.kx, f1.ky, k1["angle"], d2, node2, has2, x2, y2, a2, o2, F12, ex_, ey_, sf, sig2)
        rn, r12 = oracle.search_for_triangulation(d1, node1, has1, f1.kx, f1.ky, k1["angle"], d2, node2, has2, x2, y2, a2, o2, F12, ex_, ey_, sf, sig2, ori)
        assert n == rn and (m12 == r12).all()
        assert n > 100 and (m12[has1] == -1).all()


def test_device_acceptance_chains_and_large_batches(ctx, oracle):
    """The order-dependent acceptance on the device (k_window_greedy) at its corners:
    (a) 3000 map points projected into ONE small region: every point competes with every earlier one, so the workgroup-wide claim
        rounds decide little and the waves finish their points in order, round by round;
    (b) 20,000 map points against one keyframe (server-side map matching): no candidate list leaves the GPU;
    (c) wide windows: more candidates per point than the 16 a thread keeps in registers.
    All must equal the reference's sequential loop (the oracle) exactly, for SearchByProjection(Frame, points) and for
    SearchByProjection(KF, Scw)."""
    fr, sf, kps, desc = _frame(ctx, 10)
    n = len(fr.kx)
    rng = np.random.default_rng(21)
    m = ORBmatcher(0.8, ctx=ctx)
    # (a) a chain
    nmp = 3000
    centre = np.array([fr.kx[n // 2], fr.ky[n // 2]])
    px = (centre[0] + rng.normal(0, 6, nmp)).astype("f4"); py = (centre[1] + rng.normal(0, 6, nmp)).astype("f4")
    near = np.argsort((fr.kx - centre[0]) ** 2 + (fr.ky - centre[1]) ** 2)[:40]
    src = near[rng.integers(0, len(near), nmp)]
    mp_desc = desc[src] ^ np.packbits(rng.random((nmp, 256)) < 0.04, axis=1, bitorder="little")
    level = np.clip(fr.oct[src] + rng.integers(0, 2, nmp), 0, 7)
    ones = np.ones(nmp, bool)
    has_obs = rng.random(nmp) < 0.7
    occ0 = np.zeros(n, bool)
    nm, match, occ = m.SearchByProjection(fr, sf, ones, level, np.full(nmp, 0.9, "f4"), px, py, mp_desc, has_obs, occ0, 3.0)
    rn, rmatch, rocc = oracle.search_by_projection(fr.kx, fr.ky, fr.oct, desc, fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf, ones, level,
                                                   np.full(nmp, 0.9, "f4"), px, py, mp_desc, has_obs, occ0, 3.0, 0.8)
    assert nm == rn and (match == rmatch).all() and (occ == rocc).all() and nm > 20
    observed = rng.random(nmp) < 0.2
    n2, bi, mt = m.SearchByProjectionSim3(fr, sf, ones, px, py, level, mp_desc, observed, occ0, 6.0)
    r2, rbi, rmt = oracle.search_by_projection_sim3(fr, sf, ones, px, py, level, mp_desc, observed, occ0, 6.0)
    assert n2 == r2 and (bi == rbi).all() and (mt == rmt).all() and n2 > 3
    # (b) a large batch
    nmp = 20000
    src = rng.integers(0, n, nmp)
    mp_desc = desc[src] ^ np.packbits(rng.random((nmp, 256)) < 0.05, axis=1, bitorder="little")
    px = (fr.kx[src] + rng.normal(0, 2.0, nmp)).astype("f4"); py = (fr.ky[src] + rng.normal(0, 2.0, nmp)).astype("f4")
    level = np.clip(fr.oct[src] + rng.integers(0, 2, nmp), 0, 7)
    valid = rng.random(nmp) < 0.95
    observed = rng.random(nmp) < 0.1
    matched = rng.random(n) < 0.1
    n2, bi, mt = m.SearchByProjectionSim3(fr, sf, valid, px, py, level, mp_desc, observed, matched, 8.0)
    r2, rbi, rmt = oracle.search_by_projection_sim3(fr, sf, valid, px, py, level, mp_desc, observed, matched, 8.0)
    assert n2 == r2 and (bi == rbi).all() and (mt == rmt).all() and n2 > 300
    is2 = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx).GetInverseScaleSigmaSquares()
    bi, bd = m.FuseSelect(fr, sf, is2, valid, px, py, level, mp_desc, 3.0, True)
    rbi, rbd = oracle.fuse_select(fr.kx, fr.ky, fr.oct, desc, fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf, is2, valid, px, py, level, mp_desc, 3.0, True)
    assert (bi == rbi).all() and (bd == rbd).all() and (bi >= 0).sum() > 5000
    # (c) wide windows
    nmp = 1500
    src = rng.integers(0, n, nmp)
    mp_desc = desc[src] ^ np.packbits(rng.random((nmp, 256)) < 0.08, axis=1, bitorder="little")
    px = (fr.kx[src] + rng.normal(0, 4.0, nmp)).astype("f4"); py = (fr.ky[src] + rng.normal(0, 4.0, nmp)).astype("f4")
    level = np.clip(fr.oct[src] + rng.integers(0, 2, nmp), 0, 7)
    ones = np.ones(nmp, bool); has_obs = rng.random(nmp) < 0.8; observed = rng.random(nmp) < 0.2
    cn = m.FeaturesInArea(fr, px, py, (25.0 * np.asarray(sf)[level]).astype("f4"), (level - 1).astype("i4"), level.astype("i4"), mp_desc, cap=256)[2]
    assert (cn > 16).sum() > 100, (cn > 16).sum()
    nm, match, occ = m.SearchByProjection(fr, sf, ones, level, np.full(nmp, 0.9, "f4"), px, py, mp_desc, has_obs, occ0, 25.0 / 4.0)
    rn, rmatch, rocc = oracle.search_by_projection(fr.kx, fr.ky, fr.oct, desc, fr.min_x, fr.min_y, fr.inv_w, fr.inv_h, sf, ones, level,
                                                   np.full(nmp, 0.9, "f4"), px, py, mp_desc, has_obs, occ0, 25.0 / 4.0, 0.8)
    assert nm == rn and (match == rmatch).all() and (occ == rocc).all() and nm > 100
    n2, bi, mt = m.SearchByProjectionSim3(fr, sf, ones, px, py, level, mp_desc, observed, occ0, 25.0)
    r2, rbi, rmt = oracle.search_by_projection_sim3(fr, sf, ones, px, py, level, mp_desc, observed, occ0, 25.0)
    assert n2 == r2 and (bi == rbi).all() and (mt == rmt).all() and n2 > 100


def test_host_acceptance_paths_stay_exact():
    """The acceptance loops on the host (the round-1 path, still used when a problem exceeds the single workgroup's LDS) stay covered:
    a child process with CCM_WINDOW_HOST_ACCEPT=1 reruns the matcher tests of this file against the oracle."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CCM_WINDOW_HOST_ACCEPT="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "search_by_projection or frame_to_frame or keyframe_overload or sim3 or device_acceptance"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
