"""HIP extractor vs the CPU oracle through the C ABI: bit-exact pyramid pixels, FAST corner lists,
keypoints and descriptors (BASELINE configs 1 and 2), edge cases, and size-independent properties on
the full 256-frame batch."""
import os

import numpy as np
import pytest

from motioncheck_ccm_slam_amd import synth
from motioncheck_ccm_slam_amd.orb import ORBextractor

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(kps, desc, ref):
    assert len(kps) == len(ref["kps"])
    for name in kps.dtype.names:
        assert (kps[name] == ref["kps"][name]).all(), name
    assert (desc == ref["desc"]).all()


def test_single_frame_matches_oracle_and_fixture(ctx, oracle):
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    img = synth.frame(0)
    kps, desc = ex(img)
    ref = oracle.orb_extract(oracle.default_params(), img, want_levels=True)
    _same(kps, desc, ref)
    z = np.load(os.path.join(G, "orb_frame0.npz"))
    assert (kps == z["kps"]).all() and (desc == z["desc"]).all()
    for l in range(8):
        assert (ex.image_pyramid_level(0, l) == ref["levels"][l]).all(), "pyramid level %d" % l
    xy, sc = ex.fast_candidates(0, 0)
    assert (xy == z["cand0_xy"]).all() and (sc == z["cand0_score"]).all()


def test_batch_every_stage(ctx, oracle):
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    imgs = synth.frames(10, 5)
    kps, desc, counts = ex.extract_batch(imgs)
    par = oracle.default_params()
    for f in range(5):
        for l in range(8):
            r = oracle.orb_extract(par, imgs[f], cand_level=l)
            xy, sc = ex.fast_candidates(f, l)
            assert len(xy) == len(r["cand_xy"]) and (xy == r["cand_xy"]).all() and (sc == r["cand_score"]).all(), (f, l)
        _same(kps[f, :counts[f]], desc[f, :counts[f]], r)


@pytest.mark.parametrize("w,h,nf,sf,nl,ini,mn", [
    (200, 160, 300, 1.2, 4, 20, 7),        # small image, few levels
    (641, 479, 2000, 1.2, 8, 20, 7),       # odd size, the 2x-features initialisation extractor (Tracking.cpp:74-77)
    (752, 480, 500, 1.5, 5, 30, 10),       # other scale factor / thresholds
    (1241, 376, 1500, 1.2, 8, 20, 7),      # KITTI aspect ratio: 4 quadtree roots
    (752, 480, 1000, 1.2, 8, 7, 20),       # iniTh < minTh (degenerate but legal)
    (1920, 1080, 3000, 1.2, 8, 20, 7),     # full HD: 64 x 36 cells on level 0, 625-feature quota
])
def test_parameter_sweep(ctx, oracle, w, h, nf, sf, nl, ini, mn):
    ex = ORBextractor(nf, sf, nl, ini, mn, ctx=ctx)
    img = synth.frame(42, w, h, n_rect=max(60, 600 * w * h // (752 * 480)))
    kps, desc = ex(img)
    ref = oracle.orb_extract(oracle.default_params(nf, sf, nl, ini, mn), img)
    _same(kps, desc, ref)
    if (w, h) == (200, 160):
        z = np.load(os.path.join(G, "orb_small.npz"))
        img3 = synth.frame(3, 200, 160)
        k3, d3 = ex(img3)
        assert (k3 == z["kps"]).all() and (d3 == z["desc"]).all()


def test_texture_extremes(ctx, oracle):
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    par = oracle.default_params()
    flat = np.full((480, 752), 90, np.uint8)
    kps, desc = ex(flat)
    assert len(kps) == 0 and desc.shape == (0, 32)
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (480, 752), dtype=np.uint8)         # tens of thousands of candidates per level
    _same(*ex(noise), oracle.orb_extract(par, noise))
    checker = ((np.indices((480, 752)).sum(0) // 5) % 2 * 200 + 20).astype(np.uint8)   # massive score ties
    _same(*ex(checker), oracle.orb_extract(par, checker))
    sparse = np.full((480, 752), 30, np.uint8); sparse[200:260, 300:380] = 220       # a handful of corners: threshold fallback cells
    sparse = (sparse.astype(int) + (np.arange(752)[None, :] * 5 + np.arange(480)[:, None] * 3) % 7).astype(np.uint8)
    _same(*ex(sparse), oracle.orb_extract(par, sparse))


def test_empty_and_strided_inputs(ctx, oracle):
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    kps, desc = ex(np.zeros((0, 0), np.uint8))            # reference returns silently on an empty image
    assert len(kps) == 0
    big = np.zeros((480, 800), np.uint8)
    big[:, :752] = synth.frame(1)
    view = big[:, :752]                                   # non-contiguous rows are copied by row, like cv::Mat ROIs
    _same(*ex(np.ascontiguousarray(view)), oracle.orb_extract(oracle.default_params(), np.ascontiguousarray(view)))


def test_full_batch_properties(ctx, oracle):
    """BASELINE config 2: 256 frames.  Spot-check frames against the oracle and check the
    size-independent properties on all of them."""
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    imgs = synth.frames(0, 256)
    kps, desc, counts = ex.extract_batch(imgs)
    kps2, desc2, counts2 = ex.extract_batch(imgs)
    assert (counts == counts2).all() and (kps == kps2).all() and (desc == desc2).all()      # deterministic
    quota = ex.mnFeaturesPerLevel
    lw, lh = ex.level_sizes(752, 480)
    sc = ex.GetScaleFactors()
    for f in range(256):
        n = counts[f]
        k = kps[f, :n]
        assert 900 <= n <= 1000 + 3 * 8
        assert (np.diff(k["octave"]) >= 0).all()                         # level-major order
        per = np.bincount(k["octave"], minlength=8)
        assert (per <= quota + 3).all()
        x = k["x"] / sc[k["octave"]]; y = k["y"] / sc[k["octave"]]
        assert (x >= 18.99).all() and (y >= 18.99).all()
        assert (x <= lw[k["octave"]] - 19 + 0.01).all() and (y <= lh[k["octave"]] - 19 + 0.01).all()
        assert (k["angle"] >= 0).all() and (k["angle"] < 360).all()
        assert (kps[f, n:]["size"] == 0).all()
    par = oracle.default_params()
    for f in (0, 37, 101, 200, 255):
        _same(kps[f, :counts[f]], desc[f, :counts[f]], oracle.orb_extract(par, imgs[f]))
    # page-locked frame and result pools: every chunk's results go down on a stream of their own while the next chunk is
    # extracted (round 3) -- the same bytes must arrive, the pad rows included
    pool = (np.zeros_like(kps), np.zeros_like(desc), np.zeros_like(counts))
    for a in pool + (imgs,):
        ctx.host_register(a)
    try:
        k3, d3, c3 = ex.extract_batch(imgs, out=pool)
        assert k3 is pool[0] and (c3 == counts).all() and (k3 == kps).all() and (d3 == desc).all()
    finally:
        for a in pool + (imgs,):
            ctx.host_unregister(a)


@pytest.mark.parametrize("env", [{"CCM_ORB_FUSED": "0"}, {"CCM_FC_PACKED": "0"}, {"CCM_ORB_CHUNK": "2"}, {"CCM_BF_VARIANT": "0"}, {"CCM_OCT_REG_KEYS": "0"}])
def test_alternative_kernel_paths(env):
    """The paths the defaults do not take -- two-kernel FAST through a score map (cells wider than one LDS tile), the
    scalar rejection test, tiny upload chunks, the vector-ALU matcher (more than 2048 train rows) -- selected by their
    environment switches in a child process (the switches are read once per process) and checked against the oracle."""
    import subprocess, sys
    code = r'''
import numpy as np
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.orb import ORBextractor
from motioncheck_ccm_slam_amd.matcher import ORBmatcher
from oracle import oracle_py as O
ctx = _lib.Context(0)
ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
imgs = synth.frames(20, 5)
kps, desc, counts = ex.extract_batch(imgs)
for f in range(5):
    r = O.orb_extract(O.default_params(), imgs[f])
    n = counts[f]
    assert n == len(r["kps"]) and (desc[f, :n] == r["desc"]).all()
    for name in r["kps"].dtype.names:
        assert (kps[f, :n][name] == r["kps"][name]).all(), name
noise = np.random.default_rng(0).integers(0, 256, (480, 752), dtype=np.uint8)
k2, d2 = ex(noise); r = O.orb_extract(O.default_params(), noise)
assert len(k2) == len(r["kps"]) and (d2 == r["desc"]).all()
m = ORBmatcher(ctx=ctx)
bi, bd, sd = m.BruteForce(desc[0, :counts[0]], desc[1, :counts[1]])
rbi, rbd, rsd = O.hamming_match(desc[0, :counts[0]], desc[1, :counts[1]])
assert (bi[0] == rbi).all() and (bd[0] == rbd).all() and (sd[0] == rsd).all()
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=root, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]
