"""How far the project's shared sine/cosine departs from the reference's libm call, counted in descriptor bits.

ORBextractor.cpp:104-105 steers the rBRIEF pattern with libm cosf/sinf; oracle and HIP kernel share include/ccm_sincos.h
instead (bit-identical on both compilers).  The oracle can be switched to this machine's glibc cosf/sinf, so the departure
is measurable on the CPU: same frames, both ways, count what differs.  Measured on the 256 frames of BASELINE config 2
(tools/count_sincos_deviation.py, glibc 2.35): see DESIGN.md section 2.  This test repeats it on 12 frames and pins the
order of magnitude: a handful of bits per ten thousand keypoints, never a keypoint position, angle or count.
"""
import numpy as np

from motioncheck_ccm_slam_amd import pattern, synth


def _both_ways(oracle, frames):
    par = oracle.default_params()
    out = []
    for on in (0, 1):
        oracle.lib().orc_set_sincos_libm(on)
        try:
            out.append([oracle.orb_extract(par, f) for f in frames])
        finally:
            oracle.lib().orc_set_sincos_libm(0)
    return out


def test_libm_vs_shared_sincos_descriptor_bits(oracle):
    frames = [synth.frame(f) for f in range(0, 256, 22)]
    shared, libm = _both_ways(oracle, frames)
    n_kp = n_kp_diff = bits = 0
    for a, b in zip(shared, libm):
        assert len(a["kps"]) == len(b["kps"]) and (a["kps"] == b["kps"]).all()          # keypoints do not depend on it
        x = np.unpackbits(a["desc"] ^ b["desc"], axis=1).sum(1)
        n_kp += len(x); n_kp_diff += int((x > 0).sum()); bits += int(x.sum())
    assert n_kp > 10000
    # the deviation exists in principle (1 ulp on ~2.6 % of angles) but rarely crosses a cvRound boundary:
    assert n_kp_diff <= 0.01 * n_kp and bits <= 4 * max(n_kp_diff, 1), (n_kp, n_kp_diff, bits)


def test_rotated_pattern_never_leaves_the_level_image():
    """The largest rBRIEF pattern radius is sqrt(13^2 + 13^2) = 18.38 (e.g. (-13,-13)); a rotation keeps the radius, cvRound
    moves a coordinate by at most 0.5, so a sample lies at most 19 pixels from the keypoint in x and in y.  Keypoints come
    out of cv::FAST on cell ROIs that start EDGE_THRESHOLD - 3 = 16 pixels inside the level (ORBextractor.cpp:941-944) and
    FAST itself never reports within 3 pixels of its ROI's edge, so every keypoint is >= 19 pixels from the level's border:
    computeOrbDescriptor (:100-316) reads the blurred clone strictly inside its buffer.  (A keypoint AT distance 16 would read
    outside the tight clone -- undefined in the reference -- but none exists; this test pins the argument.)"""
    pts = pattern.PATTERN.reshape(-1, 2).astype(np.float64)
    r = np.sqrt((pts ** 2).sum(1)).max()
    assert r < 18.5
    ang = np.deg2rad(np.arange(0, 360, 0.05))
    a, b = np.cos(ang)[:, None], np.sin(ang)[:, None]
    rows = np.rint(pts[None, :, 0] * b + pts[None, :, 1] * a)
    cols = np.rint(pts[None, :, 0] * a - pts[None, :, 1] * b)
    assert max(np.abs(rows).max(), np.abs(cols).max()) <= 19 - 1 + 0       # 18: one pixel of slack to the 19-pixel margin
    EDGE_THRESHOLD, FAST_MARGIN = 19, 3
    assert (EDGE_THRESHOLD - 3) + FAST_MARGIN >= 19
