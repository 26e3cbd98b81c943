"""Host-side mirror of `cslam::ORBmatcher` (include/cslam/ORBmatcher.h:89-158) over the C ABI.

Frames / keyframes are passed as plain arrays: descriptors [N,32] uint8, the
FeatureVector as one vocabulary-node id per feature, MapPoint validity as a
0/1 mask, keypoint angles in degrees.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BowOptions


class ORBmatcher:
    TH_HIGH = 100          # cslam/src/ORBmatcher.cpp:63-65
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, ctx: _lib.Context | None = None):
        self.lib = _lib.load()
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._ctx = ctx

    @property
    def ctx(self) -> _lib.Context:
        if self._ctx is None:
            self._ctx = _lib.default_context(0)
        return self._ctx

    @staticmethod
    def DescriptorDistance(a: np.ndarray, b: np.ndarray) -> int:
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return _lib.load().ccm_descriptor_distance(_lib.ptr(a), _lib.ptr(b))

    def BruteForce(self, q: np.ndarray, t: np.ndarray, nq_n=None, nt_n=None):
        """q [P,NQ,32], t [P,NT,32] -> best_idx, best_dist, second_dist, each [P,NQ] int32."""
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        if q.ndim == 2:
            q = q[None]; t = t[None]
        p, nq, _ = q.shape
        nt = t.shape[1]
        bi = np.full((p, nq), -1, "i4"); bd = np.full((p, nq), 256, "i4"); sd = np.full((p, nq), 256, "i4")
        nq_n = None if nq_n is None else np.ascontiguousarray(nq_n, "i4")
        nt_n = None if nt_n is None else np.ascontiguousarray(nt_n, "i4")
        self.ctx.check(self.lib.ccm_hamming_match(self.ctx.handle, _lib.ptr(q), nq, _lib.ptr(t), nt, p,
                                                  _lib.ptr(nq_n), _lib.ptr(nt_n), _lib.ptr(bi), _lib.ptr(bd), _lib.ptr(sd)))
        return bi, bd, sd

    def RatioTest(self, best, second, th=None, strict=False):
        th = self.TH_LOW if th is None else th
        f = self.lib.ccm_ratio_test
        return np.array([f(int(b), int(s), C.c_float(self.mfNNratio), int(th), int(strict)) for b, s in zip(best, second)], bool)

    def SearchByBoW(self, desc1, node1, valid1, angle1, desc2, node2, angle2, valid2=None):
        """Both overloads of SearchByBoW: valid2=None is (KeyFrame, Frame) -- accept best <= TH_LOW;
        valid2 given is (KeyFrame, KeyFrame) -- accept best < TH_LOW.  Returns (nmatches, match12)
        with match12[i1] = index into side 2 or -1."""
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        n1, n2 = len(d1), len(d2)
        node1 = np.ascontiguousarray(node1, "i4"); node2 = np.ascontiguousarray(node2, "i4")
        valid1 = np.ascontiguousarray(valid1, np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        a1 = np.ascontiguousarray(angle1, "f4"); a2 = np.ascontiguousarray(angle2, "f4")
        opt = BowOptions(self.mfNNratio, int(self.mbCheckOrientation), self.TH_LOW, 0 if valid2 is None else 1)
        m = np.full(n1, -1, "i4")
        n = self.ctx.check(self.lib.ccm_match_bow(self.ctx.handle, C.byref(opt), _lib.ptr(d1), _lib.ptr(node1),
                                                  _lib.ptr(valid1), _lib.ptr(a1), n1, _lib.ptr(d2), _lib.ptr(node2),
                                                  _lib.ptr(v2), _lib.ptr(a2), n2, _lib.ptr(m)))
        return n, m


FRAME_GRID_COLS, FRAME_GRID_ROWS = 75, 48        # include/cslam/Frame.h:51-52


class FrameGridView:
    """What the windowed matchers need from a `Frame`: undistorted keypoints, descriptors and the feature grid
    parameters computed in the Frame constructor (src/Frame.cpp:80-88)."""

    def __init__(self, kp_x, kp_y, kp_octave, desc, min_x=0.0, max_x=752.0, min_y=0.0, max_y=480.0):
        self.kx = np.ascontiguousarray(kp_x, "f4"); self.ky = np.ascontiguousarray(kp_y, "f4")
        self.oct = np.ascontiguousarray(kp_octave, "i4"); self.desc = np.ascontiguousarray(desc, np.uint8)
        self.min_x = np.float32(min_x); self.min_y = np.float32(min_y)
        self.inv_w = np.float32(FRAME_GRID_COLS) / np.float32(np.float32(max_x) - np.float32(min_x))
        self.inv_h = np.float32(FRAME_GRID_ROWS) / np.float32(np.float32(max_y) - np.float32(min_y))

    def struct(self):
        p = _lib.ptr
        return _lib.FrameGrid(len(self.kx), p(self.kx), p(self.ky), p(self.oct), p(self.desc), float(self.min_x), float(self.min_y),
                              float(self.inv_w), float(self.inv_h), FRAME_GRID_COLS, FRAME_GRID_ROWS)


def _features_in_area(self, frame: FrameGridView, x, y, r, min_level, max_level, qdesc, cap=256):
    """Frame::GetFeaturesInArea for many queries + Hamming distance to each query descriptor."""
    x = np.ascontiguousarray(x, "f4"); y = np.ascontiguousarray(y, "f4"); r = np.ascontiguousarray(r, "f4")
    mn = np.ascontiguousarray(min_level, "i4"); mx = np.ascontiguousarray(max_level, "i4")
    qd = np.ascontiguousarray(qdesc, np.uint8)
    nq = len(x)
    ci = np.zeros((nq, cap), "i4"); cd = np.zeros((nq, cap), "i4"); cn = np.zeros(nq, "i4")
    g = frame.struct()
    self.ctx.check(self.lib.ccm_window_candidates(self.ctx.handle, C.byref(g), nq, _lib.ptr(x), _lib.ptr(y), _lib.ptr(r), _lib.ptr(mn),
                                                  _lib.ptr(mx), _lib.ptr(qd), cap, _lib.ptr(ci), _lib.ptr(cd), _lib.ptr(cn)))
    return ci, cd, cn


def _search_by_projection(self, frame: FrameGridView, scale_factors, in_view, level, view_cos, proj_x, proj_y, mp_desc,
                          mp_has_obs, occupied, th: float = 1.0):
    """ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th) (ORBmatcher.cpp:71-148).
    Returns (nmatches, match[feature] = map point index or -1, occupied after the call)."""
    sf = np.ascontiguousarray(scale_factors, "f4"); iv = np.ascontiguousarray(in_view, np.uint8)
    lv = np.ascontiguousarray(level, "i4"); vc = np.ascontiguousarray(view_cos, "f4")
    px = np.ascontiguousarray(proj_x, "f4"); py = np.ascontiguousarray(proj_y, "f4")
    md = np.ascontiguousarray(mp_desc, np.uint8); ho = np.ascontiguousarray(mp_has_obs, np.uint8)
    occ = np.ascontiguousarray(occupied, np.uint8).copy()
    match = np.full(max(len(frame.kx), 1), -1, "i4")
    g = frame.struct()
    n = self.ctx.check(self.lib.ccm_search_by_projection(self.ctx.handle, C.byref(g), _lib.ptr(sf), len(iv), _lib.ptr(iv), _lib.ptr(lv),
                                                         _lib.ptr(vc), _lib.ptr(px), _lib.ptr(py), _lib.ptr(md), _lib.ptr(ho), _lib.ptr(occ),
                                                         C.c_float(th), C.c_float(self.mfNNratio), _lib.ptr(match)))
    return n, match[:len(frame.kx)], occ


ORBmatcher.FeaturesInArea = _features_in_area
ORBmatcher.SearchByProjection = _search_by_projection


def _search_by_projection_frame(self, cur: FrameGridView, cur_angle, scale_factors, valid, u, v, last_octave, last_angle, mp_desc,
                                mp_has_obs, occupied, th: float, orb_dist: int = 100):
    """ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th) (ORBmatcher.cpp:1350-1476); with
    mp_has_obs all ones and orb_dist = ORBdist also SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1478-1605)."""
    a = np.ascontiguousarray
    ca = a(cur_angle, "f4"); sf = a(scale_factors, "f4"); va = a(valid, np.uint8); uu = a(u, "f4"); vv = a(v, "f4")
    lo = a(last_octave, "i4"); la = a(last_angle, "f4"); md = a(mp_desc, np.uint8); ho = a(mp_has_obs, np.uint8)
    occ = a(occupied, np.uint8).copy()
    match = np.full(max(len(cur.kx), 1), -1, "i4")
    g = cur.struct()
    p = _lib.ptr
    n = self.ctx.check(self.lib.ccm_search_by_projection_frame(self.ctx.handle, C.byref(g), p(ca), p(sf), len(va), p(va), p(uu), p(vv), p(lo),
                                                               p(la), p(md), p(ho), p(occ), C.c_float(th), int(self.mbCheckOrientation), int(orb_dist), p(match)))
    return n, match[:len(cur.kx)], occ


ORBmatcher.SearchByProjectionFrame = _search_by_projection_frame


def _search_for_initialization(self, oct1, desc1, angle1, f2: FrameGridView, angle2, prev_matched, window: int = 100):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cpp:448-563).  Returns (nmatches, matches12, prev_matched)."""
    a = np.ascontiguousarray
    o1 = a(oct1, "i4"); d1 = a(desc1, np.uint8); a1 = a(angle1, "f4"); a2 = a(angle2, "f4")
    pm = a(prev_matched, "f4").copy()
    m12 = np.full(max(len(o1), 1), -1, "i4")
    g = f2.struct()
    p = _lib.ptr
    n = self.ctx.check(self.lib.ccm_search_for_initialization(self.ctx.handle, len(o1), p(o1), p(d1), p(a1), C.byref(g), p(a2), p(pm), int(window),
                                                              C.c_float(self.mfNNratio), int(self.mbCheckOrientation), p(m12)))
    return n, m12[:len(o1)], pm


ORBmatcher.SearchForInitialization = _search_for_initialization


def _fuse_select(self, kf: FrameGridView, scale_factors, inv_level_sigma2, valid, u, v, level, mp_desc, th: float, chi2_check: bool,
                 accept_th: int = 50):
    """Selection loop of ORBmatcher::Fuse (both overloads).  Returns (best_idx, best_dist) per map point."""
    a = np.ascontiguousarray
    sf = a(scale_factors, "f4"); s2 = a(inv_level_sigma2, "f4"); va = a(valid, np.uint8); uu = a(u, "f4"); vv = a(v, "f4")
    lv = a(level, "i4"); md = a(mp_desc, np.uint8)
    n = len(va)
    bi = np.full(max(n, 1), -1, "i4"); bd = np.full(max(n, 1), 256, "i4")
    g = kf.struct()
    p = _lib.ptr
    self.ctx.check(self.lib.ccm_fuse_select(self.ctx.handle, C.byref(g), p(sf), p(s2), n, p(va), p(uu), p(vv), p(lv), p(md), C.c_float(th),
                                            int(chi2_check), int(accept_th), p(bi), p(bd)))
    return bi[:n], bd[:n]


ORBmatcher.FuseSelect = _fuse_select


def _fuse_select_batch(self, kfs, scale_factors, inv_level_sigma2, per_kf, th: float, chi2_check: bool, accept_th: int = 50):
    """FuseSelect for many keyframes in one launch (ccm_fuse_select_batch; the server's fuse loops, src/Mapping.cpp:515-546,
    src/MapMerger.cpp:576-586).  kfs: list of FrameGridView; per_kf: list of (valid, u, v, level, mp_desc) per keyframe.
    Returns a list of (best_idx, best_dist) per keyframe -- what FuseSelect returns keyframe by keyframe."""
    a = np.ascontiguousarray
    sf = a(scale_factors, "f4"); s2 = a(inv_level_sigma2, "f4")
    first = np.zeros(len(kfs) + 1, "i4")
    for k, t in enumerate(per_kf):
        first[k + 1] = first[k] + len(t[0])
    cat = lambda i, dt: a(np.concatenate([np.asarray(t[i]) for t in per_kf]) if len(per_kf) else np.zeros(0), dt)
    va, uu, vv, lv = cat(0, np.uint8), cat(1, "f4"), cat(2, "f4"), cat(3, "i4")
    md = a(np.concatenate([np.asarray(t[4], np.uint8).reshape(-1, 32) for t in per_kf]) if len(per_kf) else np.zeros((0, 32)), np.uint8)
    n = int(first[-1])
    bi = np.full(max(n, 1), -1, "i4"); bd = np.full(max(n, 1), 256, "i4")
    grids = (_lib.FrameGrid * max(len(kfs), 1))(*[kf.struct() for kf in kfs])
    p = _lib.ptr
    self.ctx.check(self.lib.ccm_fuse_select_batch(self.ctx.handle, len(kfs), C.cast(grids, C.c_void_p), p(sf), p(s2), p(first), p(va), p(uu), p(vv), p(lv), p(md),
                                                  C.c_float(th), int(chi2_check), int(accept_th), p(bi), p(bd)))
    return [(bi[first[k]:first[k + 1]].copy(), bd[first[k]:first[k + 1]].copy()) for k in range(len(kfs))]


ORBmatcher.FuseSelectBatch = _fuse_select_batch


def _search_by_sim3(self, kf1: FrameGridView, sf1, kf2: FrameGridView, sf2, valid1, u1, v1, level1, mp_desc1, valid2, u2, v2, level2, mp_desc2,
                    th: float):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cpp:1124-1348) after the caller's projections.  Returns (nFound, match12)."""
    a = np.ascontiguousarray
    arrs = [a(sf1, "f4"), a(sf2, "f4"), a(valid1, np.uint8), a(u1, "f4"), a(v1, "f4"), a(level1, "i4"), a(mp_desc1, np.uint8),
            a(valid2, np.uint8), a(u2, "f4"), a(v2, "f4"), a(level2, "i4"), a(mp_desc2, np.uint8)]
    m12 = np.full(max(len(kf1.kx), 1), -1, "i4")
    g1, g2 = kf1.struct(), kf2.struct()
    p = _lib.ptr
    n = self.ctx.check(self.lib.ccm_search_by_sim3(self.ctx.handle, C.byref(g1), p(arrs[0]), C.byref(g2), p(arrs[1]), *[p(x) for x in arrs[2:]],
                                                   C.c_float(th), p(m12)))
    return n, m12[:len(kf1.kx)]


def _search_by_projection_sim3(self, kf: FrameGridView, scale_factors, valid, u, v, level, mp_desc, observed, matched, th: float):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cpp:308-446).
    Returns (nmatches, best_idx per map point, matched after the call)."""
    a = np.ascontiguousarray
    sf = a(scale_factors, "f4"); va = a(valid, np.uint8); uu = a(u, "f4"); vv = a(v, "f4"); lv = a(level, "i4"); md = a(mp_desc, np.uint8)
    ob = a(observed, np.uint8); mt = a(matched, np.uint8).copy()
    bi = np.full(max(len(va), 1), -1, "i4")
    g = kf.struct()
    p = _lib.ptr
    n = self.ctx.check(self.lib.ccm_search_by_projection_sim3(self.ctx.handle, C.byref(g), p(sf), len(va), p(va), p(uu), p(vv), p(lv), p(md), p(ob),
                                                              p(mt), C.c_float(th), p(bi)))
    return n, bi[:len(va)], mt


def _search_by_projection_sim3_batch(self, kfs, scale_factors, per_kf, th: float):
    """SearchByProjection(pKF, Scw, ...) for many keyframes in one launch per kernel (ccm_search_by_projection_sim3_batch).
    kfs: list of FrameGridView; per_kf: list of (valid, u, v, level, mp_desc, observed, matched) per keyframe.
    Returns a list of (nmatches, best_idx, matched after the call) per keyframe -- what SearchByProjectionSim3 returns keyframe by keyframe."""
    a = np.ascontiguousarray
    sf = a(scale_factors, "f4")
    K = len(kfs)
    first = np.zeros(K + 1, "i4"); ffirst = np.zeros(K + 1, "i8")
    for k, t in enumerate(per_kf):
        first[k + 1] = first[k] + len(t[0]); ffirst[k + 1] = ffirst[k] + len(kfs[k].kx)
    cat = lambda i, dt: a(np.concatenate([np.asarray(t[i]) for t in per_kf]) if K else np.zeros(0), dt)
    va, uu, vv, lv, ob = cat(0, np.uint8), cat(1, "f4"), cat(2, "f4"), cat(3, "i4"), cat(5, np.uint8)
    md = a(np.concatenate([np.asarray(t[4], np.uint8).reshape(-1, 32) for t in per_kf]) if K else np.zeros((0, 32)), np.uint8)
    mt = cat(6, np.uint8).copy()
    assert len(mt) == ffirst[-1]
    n = int(first[-1])
    bi = np.full(max(n, 1), -1, "i4"); nm = np.zeros(max(K, 1), "i4")
    if len(mt) == 0:
        mt = np.zeros(1, np.uint8)
    grids = (_lib.FrameGrid * max(K, 1))(*[kf.struct() for kf in kfs])
    p = _lib.ptr
    tot = self.ctx.check(self.lib.ccm_search_by_projection_sim3_batch(self.ctx.handle, K, C.cast(grids, C.c_void_p), p(sf), p(first), p(va), p(uu), p(vv), p(lv),
                                                                      p(md), p(ob), p(mt), C.c_float(th), p(bi), p(nm)))
    out = [(int(nm[k]), bi[first[k]:first[k + 1]].copy(), mt[ffirst[k]:ffirst[k + 1]].copy()) for k in range(K)]
    assert tot == sum(o[0] for o in out)
    return out


ORBmatcher.SearchByProjectionSim3Batch = _search_by_projection_sim3_batch


def _search_for_triangulation(self, desc1, node1, has_mp1, x1, y1, angle1, desc2, node2, has_mp2, x2, y2, angle2, octave2, F12, ex, ey,
                              scale_factors2, level_sigma2_2):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cpp:700-852).  Returns (nmatches, match12)."""
    a = np.ascontiguousarray
    d1 = a(desc1, np.uint8); n1 = a(node1, "i4"); h1 = a(has_mp1, np.uint8); xx1 = a(x1, "f4"); yy1 = a(y1, "f4"); a1 = a(angle1, "f4")
    d2 = a(desc2, np.uint8); n2 = a(node2, "i4"); h2 = a(has_mp2, np.uint8); xx2 = a(x2, "f4"); yy2 = a(y2, "f4"); a2 = a(angle2, "f4")
    o2 = a(octave2, "i4"); F = a(F12, "f4").reshape(9); sf = a(scale_factors2, "f4"); s2 = a(level_sigma2_2, "f4")
    m12 = np.full(max(len(d1), 1), -1, "i4")
    p = _lib.ptr
    n = self.ctx.check(self.lib.ccm_search_for_triangulation(self.ctx.handle, p(d1), p(n1), p(h1), p(xx1), p(yy1), p(a1), len(d1), p(d2), p(n2),
                                                             p(h2), p(xx2), p(yy2), p(a2), p(o2), len(d2), p(F), C.c_float(ex), C.c_float(ey),
                                                             p(sf), p(s2), int(self.mbCheckOrientation), p(m12)))
    return n, m12[:len(d1)]


ORBmatcher.SearchBySim3 = _search_by_sim3
ORBmatcher.SearchByProjectionSim3 = _search_by_projection_sim3
ORBmatcher.SearchForTriangulation = _search_for_triangulation
