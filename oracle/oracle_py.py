"""ctypes front end of liboracle.so -- TEST INFRASTRUCTURE.

Import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (motioncheck_ccm_slam_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th", C.c_int), ("min_th", C.c_int)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("poses", C.c_void_p), ("fixed", C.c_void_p), ("intr", C.c_void_p),
                ("n_points", C.c_int), ("points", C.c_void_p),
                ("n_edges", C.c_int), ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p),
                ("obs", C.c_void_p), ("info", C.c_void_p)]


class BaOptions(C.Structure):
    _fields_ = [("iterations", C.c_int), ("huber_delta", C.c_double), ("iterations2", C.c_int),
                ("outlier_chi2", C.c_double)]


class BaResult(C.Structure):
    _fields_ = [("iterations_done", C.c_int), ("trials", C.c_int), ("chi2_initial", C.c_double),
                ("chi2_final", C.c_double), ("lambda_final", C.c_double)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.c", "match_oracle.c", "ba_oracle.c", "bchol_oracle.c", "bow_oracle.c", "oracle.h")]
    if force or not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_round_half_even.argtypes = [C.c_double]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params(nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7) -> OrbParams:
    return OrbParams(nfeatures, scale, nlevels, ini, mn)


def orb_tables(par: OrbParams):
    n = par.nlevels
    sc = np.zeros(n, "f4"); inv = np.zeros(n, "f4"); s2 = np.zeros(n, "f4"); is2 = np.zeros(n, "f4")
    nf = np.zeros(n, "i4"); um = np.zeros(16, "i4")
    rc = lib().orc_orb_tables(C.byref(par), _p(sc), _p(inv), _p(s2), _p(is2), _p(nf), _p(um))
    assert rc == 0
    return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, nfeat=nf, umax=um)


def level_sizes(par: OrbParams, w: int, h: int):
    lw = np.zeros(par.nlevels, "i4"); lh = np.zeros(par.nlevels, "i4")
    lib().orc_orb_level_sizes(C.byref(par), w, h, _p(lw), _p(lh))
    return lw, lh


def resize(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def fast(img: np.ndarray, th: int, cap: int = 1 << 16):
    img = np.ascontiguousarray(img, np.uint8)
    xy = np.zeros((cap, 2), "i4"); sc = np.zeros(cap, "i4")
    n = lib().orc_fast9_16(_p(img), img.shape[1], img.shape[0], img.shape[1], th, _p(xy), _p(sc), cap)
    assert n <= cap
    return xy[:n].copy(), sc[:n].copy()


def blur(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orc_blur7_sigma2(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(out), img.shape[1])
    return out


def orb_extract(par: OrbParams, img: np.ndarray, max_kps: int = 4096, cand_level: int = -1, want_levels=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    kps = np.zeros(max_kps, KP_DTYPE); desc = np.zeros((max_kps, 32), np.uint8)
    lw, lh = level_sizes(par, w, h)
    tot = int((lw.astype(np.int64) * lh).sum())
    levels = np.zeros(tot, np.uint8) if want_levels else None
    cap = 1 << 18
    cxy = np.zeros((cap, 2), "i4"); csc = np.zeros(cap, "i4"); cn = C.c_int32(0)
    n = lib().orc_orb_extract(C.byref(par), _p(img), w, h, w, _p(kps), _p(desc), max_kps,
                              _p(levels), C.c_size_t(tot if want_levels else 0),
                              cand_level, _p(cxy), _p(csc), cap, C.byref(cn))
    assert n >= 0, n
    out = dict(kps=kps[:n].copy(), desc=desc[:n].copy())
    if cand_level >= 0:
        out["cand_xy"] = cxy[:cn.value].copy(); out["cand_score"] = csc[:cn.value].copy()
    if want_levels:
        lv, off = [], 0
        for l in range(par.nlevels):
            lv.append(levels[off:off + lw[l] * lh[l]].reshape(lh[l], lw[l])); off += lw[l] * lh[l]
        out["levels"] = lv
    return out


def distance(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def hamming_match(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    bi = np.zeros(nq, "i4"); bd = np.zeros(nq, "i4"); sd = np.zeros(nq, "i4")
    lib().orc_hamming_match(_p(q), nq, _p(t), nt, _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def match_bow(nnratio, check_ori, th, strict, d1, node1, valid1, ang1, d2, node2, valid2, ang2):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    node1 = np.ascontiguousarray(node1, "i4"); node2 = np.ascontiguousarray(node2, "i4")
    valid1 = np.ascontiguousarray(valid1, np.uint8)
    valid2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    ang1 = np.ascontiguousarray(ang1, "f4"); ang2 = np.ascontiguousarray(ang2, "f4")
    m = np.zeros(len(d1), "i4")
    n = lib().orc_match_bow(C.c_float(nnratio), int(check_ori), int(th), int(strict),
                            _p(d1), _p(node1), _p(valid1), _p(ang1), len(d1),
                            _p(d2), _p(node2), _p(valid2), _p(ang2), len(d2), _p(m))
    return n, m


def _ba_problem(g):
    keep = dict(
        poses=np.ascontiguousarray(g["poses"], "f8").copy(), fixed=np.ascontiguousarray(g["fixed"], np.uint8),
        intr=np.ascontiguousarray(g["intr"], "f8"), points=np.ascontiguousarray(g["points"], "f8").copy(),
        edge_pose=np.ascontiguousarray(g["edge_pose"], "i4"), edge_point=np.ascontiguousarray(g["edge_point"], "i4"),
        obs=np.ascontiguousarray(g["obs"], "f8"), info=np.ascontiguousarray(g["info"], "f8"))
    pb = BaProblem(len(keep["poses"]), _p(keep["poses"]), _p(keep["fixed"]), _p(keep["intr"]),
                   len(keep["points"]), _p(keep["points"]), len(keep["edge_pose"]),
                   _p(keep["edge_pose"]), _p(keep["edge_point"]), _p(keep["obs"]), _p(keep["info"]))
    return pb, keep


def ba_solve(g, iterations, huber_delta, iterations2=0, outlier_chi2=5.991):
    pb, keep = _ba_problem(g)
    opt = BaOptions(iterations, huber_delta, iterations2, outlier_chi2)
    res = BaResult()
    outl = np.zeros(len(keep["edge_pose"]), np.uint8)
    rc = lib().orc_ba_solve(C.byref(pb), C.byref(opt), C.byref(res), _p(outl))
    assert rc == 0
    return dict(poses=keep["poses"], points=keep["points"], outlier=outl,
                iterations_done=res.iterations_done, trials=res.trials, chi2_initial=res.chi2_initial,
                chi2_final=res.chi2_final, lambda_final=res.lambda_final)


def ba_edge(pose7, intr4, pt3, obs2):
    e = np.zeros(2); A = np.zeros((2, 3)); B = np.zeros((2, 6))
    a = [np.ascontiguousarray(v, "f8") for v in (pose7, intr4, pt3, obs2)]
    lib().orc_ba_edge(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(e), _p(A), _p(B))
    return e, A, B


def se3_exp_mul(delta6, pose7):
    d = np.ascontiguousarray(delta6, "f8"); p = np.ascontiguousarray(pose7, "f8"); o = np.zeros(7)
    lib().orc_se3_exp_mul(_p(d), _p(p), _p(o))
    return o


def ba_reduced_system(g, huber_delta, lam):
    pb, keep = _ba_problem(g)
    nfree = int((keep["fixed"] == 0).sum())
    H = np.zeros((6 * nfree, 6 * nfree)); b = np.zeros(6 * nfree); fi = np.zeros(len(keep["poses"]), "i4")
    P = lib().orc_ba_reduced_system(C.byref(pb), C.c_double(huber_delta), C.c_double(lam), _p(H), _p(b), _p(fi))
    assert P == nfree
    return H, b, fi


def pose_optimize(pose7, intr4, pts, obs, info):
    pose = np.ascontiguousarray(pose7, "f8").copy()
    a = [np.ascontiguousarray(v, "f8") for v in (intr4, pts, obs, info)]
    n = len(a[3])
    outl = np.zeros(max(n, 1), np.uint8); ninl = C.c_int(0)
    lib().orc_pose_optimize(_p(pose), _p(a[0]), n, _p(a[1]), _p(a[2]), _p(a[3]), _p(outl), C.byref(ninl))
    return pose, outl[:n].copy(), ninl.value


def features_in_area(kx, ky, octv, min_x, min_y, inv_w, inv_h, x, y, r, min_level, max_level, cols=75, rows=48):
    kx = np.ascontiguousarray(kx, "f4"); ky = np.ascontiguousarray(ky, "f4"); octv = np.ascontiguousarray(octv, "i4")
    out = np.zeros(max(len(kx), 1), "i4")
    n = lib().orc_features_in_area(len(kx), _p(kx), _p(ky), _p(octv), C.c_float(min_x), C.c_float(min_y), C.c_float(inv_w), C.c_float(inv_h),
                                   cols, rows, C.c_float(x), C.c_float(y), C.c_float(r), int(min_level), int(max_level), _p(out), len(out))
    return out[:n].copy()


def search_by_projection(kx, ky, octv, desc, min_x, min_y, inv_w, inv_h, scale_factors, in_view, level, view_cos, proj_x, proj_y,
                         mp_desc, mp_has_obs, occupied, th, nnratio, cols=75, rows=48):
    a = lambda v, t: np.ascontiguousarray(v, t)
    kx = a(kx, "f4"); ky = a(ky, "f4"); octv = a(octv, "i4"); desc = a(desc, np.uint8); sf = a(scale_factors, "f4")
    iv = a(in_view, np.uint8); lv = a(level, "i4"); vc = a(view_cos, "f4"); px = a(proj_x, "f4"); py = a(proj_y, "f4")
    md = a(mp_desc, np.uint8); ho = a(mp_has_obs, np.uint8); occ = a(occupied, np.uint8).copy()
    match = np.full(max(len(kx), 1), -1, "i4")
    n = lib().orc_search_by_projection(len(kx), _p(kx), _p(ky), _p(octv), _p(desc), C.c_float(min_x), C.c_float(min_y), C.c_float(inv_w),
                                       C.c_float(inv_h), cols, rows, _p(sf), len(iv), _p(iv), _p(lv), _p(vc), _p(px), _p(py), _p(md), _p(ho),
                                       _p(occ), C.c_float(th), C.c_float(nnratio), _p(match))
    return n, match[:len(kx)].copy(), occ


def search_by_projection_frame(kx, ky, octv, desc, angle, min_x, min_y, inv_w, inv_h, scale_factors, valid, u, v, last_octave, last_angle,
                               mp_desc, mp_has_obs, occupied, th, check_ori, cols=75, rows=48, orb_dist=100):
    a = lambda x, t: np.ascontiguousarray(x, t)
    kx = a(kx, "f4"); ky = a(ky, "f4"); octv = a(octv, "i4"); desc = a(desc, np.uint8); angle = a(angle, "f4"); sf = a(scale_factors, "f4")
    va = a(valid, np.uint8); u = a(u, "f4"); v = a(v, "f4"); lo = a(last_octave, "i4"); la = a(last_angle, "f4")
    md = a(mp_desc, np.uint8); ho = a(mp_has_obs, np.uint8); occ = a(occupied, np.uint8).copy()
    match = np.full(max(len(kx), 1), -1, "i4")
    n = lib().orc_search_by_projection_frame(len(kx), _p(kx), _p(ky), _p(octv), _p(desc), _p(angle), C.c_float(min_x), C.c_float(min_y),
                                             C.c_float(inv_w), C.c_float(inv_h), cols, rows, _p(sf), len(va), _p(va), _p(u), _p(v), _p(lo),
                                             _p(la), _p(md), _p(ho), _p(occ), C.c_float(th), int(check_ori), int(orb_dist), _p(match))
    return n, match[:len(kx)].copy(), occ


def search_for_initialization(oct1, desc1, angle1, kx2, ky2, oct2, desc2, angle2, min_x, min_y, inv_w, inv_h, prev_matched, window,
                              nnratio, check_ori, cols=75, rows=48):
    a = lambda x, t: np.ascontiguousarray(x, t)
    o1 = a(oct1, "i4"); d1 = a(desc1, np.uint8); a1 = a(angle1, "f4")
    kx2 = a(kx2, "f4"); ky2 = a(ky2, "f4"); o2 = a(oct2, "i4"); d2 = a(desc2, np.uint8); a2 = a(angle2, "f4")
    pm = a(prev_matched, "f4").copy(); m12 = np.full(max(len(o1), 1), -1, "i4")
    n = lib().orc_search_for_initialization(len(o1), _p(o1), _p(d1), _p(a1), len(kx2), _p(kx2), _p(ky2), _p(o2), _p(d2), _p(a2),
                                            C.c_float(min_x), C.c_float(min_y), C.c_float(inv_w), C.c_float(inv_h), cols, rows,
                                            _p(pm), int(window), C.c_float(nnratio), int(check_ori), _p(m12))
    return n, m12[:len(o1)].copy(), pm


def fuse_select(kx, ky, octv, desc, min_x, min_y, inv_w, inv_h, scale_factors, inv_level_sigma2, valid, u, v, level, mp_desc, th, chi2_check,
                cols=75, rows=48, accept_th=50):
    a = lambda x, t: np.ascontiguousarray(x, t)
    kx = a(kx, "f4"); ky = a(ky, "f4"); octv = a(octv, "i4"); desc = a(desc, np.uint8); sf = a(scale_factors, "f4"); s2 = a(inv_level_sigma2, "f4")
    va = a(valid, np.uint8); u = a(u, "f4"); v = a(v, "f4"); lv = a(level, "i4"); md = a(mp_desc, np.uint8)
    n = len(va); bi = np.full(max(n, 1), -1, "i4"); bd = np.full(max(n, 1), 256, "i4")
    lib().orc_fuse_select(len(kx), _p(kx), _p(ky), _p(octv), _p(desc), C.c_float(min_x), C.c_float(min_y), C.c_float(inv_w), C.c_float(inv_h),
                          cols, rows, _p(sf), _p(s2), n, _p(va), _p(u), _p(v), _p(lv), _p(md), C.c_float(th), int(chi2_check), int(accept_th), _p(bi), _p(bd))
    return bi[:n].copy(), bd[:n].copy()


def search_by_sim3(f1, sf1, f2, sf2, valid1, u1, v1, level1, mpdesc1, valid2, u2, v2, level2, mpdesc2, th, cols=75, rows=48):
    """f1, f2: objects with kx, ky, oct, desc, min_x, min_y, inv_w, inv_h (matcher.FrameGridView)."""
    a = lambda x, t: np.ascontiguousarray(x, t)
    g1 = np.array([f1.min_x, f1.min_y, f1.inv_w, f1.inv_h], "f4"); g2 = np.array([f2.min_x, f2.min_y, f2.inv_w, f2.inv_h], "f4")
    sf1 = a(sf1, "f4"); sf2 = a(sf2, "f4")
    A = [a(valid1, np.uint8), a(u1, "f4"), a(v1, "f4"), a(level1, "i4"), a(mpdesc1, np.uint8),
         a(valid2, np.uint8), a(u2, "f4"), a(v2, "f4"), a(level2, "i4"), a(mpdesc2, np.uint8)]
    m12 = np.full(max(len(f1.kx), 1), -1, "i4")
    n = lib().orc_search_by_sim3(len(f1.kx), _p(f1.kx), _p(f1.ky), _p(f1.oct), _p(f1.desc), _p(g1), _p(sf1),
                                 len(f2.kx), _p(f2.kx), _p(f2.ky), _p(f2.oct), _p(f2.desc), _p(g2), _p(sf2), cols, rows,
                                 *[_p(x) for x in A], C.c_float(th), _p(m12))
    return n, m12[:len(f1.kx)].copy()


def search_by_projection_sim3(f, scale_factors, valid, u, v, level, mp_desc, observed, matched, th, cols=75, rows=48):
    a = lambda x, t: np.ascontiguousarray(x, t)
    sf = a(scale_factors, "f4"); va = a(valid, np.uint8); u = a(u, "f4"); v = a(v, "f4"); lv = a(level, "i4"); md = a(mp_desc, np.uint8)
    ob = a(observed, np.uint8); mt = a(matched, np.uint8).copy()
    bi = np.full(max(len(va), 1), -1, "i4")
    n = lib().orc_search_by_projection_sim3(len(f.kx), _p(f.kx), _p(f.ky), _p(f.oct), _p(f.desc), C.c_float(f.min_x), C.c_float(f.min_y),
                                            C.c_float(f.inv_w), C.c_float(f.inv_h), cols, rows, _p(sf), len(va), _p(va), _p(u), _p(v), _p(lv),
                                            _p(md), _p(ob), _p(mt), C.c_float(th), _p(bi))
    return n, bi[:len(va)].copy(), mt


def search_for_triangulation(desc1, node1, has_mp1, x1, y1, angle1, desc2, node2, has_mp2, x2, y2, angle2, octave2, F12, ex, ey,
                             scale_factors2, level_sigma2_2, check_ori):
    a = lambda x, t: np.ascontiguousarray(x, t)
    d1 = a(desc1, np.uint8); n1 = a(node1, "i4"); h1 = a(has_mp1, np.uint8); xx1 = a(x1, "f4"); yy1 = a(y1, "f4"); a1 = a(angle1, "f4")
    d2 = a(desc2, np.uint8); n2 = a(node2, "i4"); h2 = a(has_mp2, np.uint8); xx2 = a(x2, "f4"); yy2 = a(y2, "f4"); a2 = a(angle2, "f4")
    o2 = a(octave2, "i4"); F = a(F12, "f4").reshape(9); sf = a(scale_factors2, "f4"); s2 = a(level_sigma2_2, "f4")
    m12 = np.full(max(len(d1), 1), -1, "i4")
    n = lib().orc_search_for_triangulation(_p(d1), _p(n1), _p(h1), _p(xx1), _p(yy1), _p(a1), len(d1), _p(d2), _p(n2), _p(h2), _p(xx2), _p(yy2),
                                           _p(a2), _p(o2), len(d2), _p(F), C.c_float(ex), C.c_float(ey), _p(sf), _p(s2), int(check_ori), _p(m12))
    return n, m12[:len(d1)].copy()


class Voc:
    """orc_voc_* wrappers (bow_oracle.c)."""

    def __init__(self, k, L, parent, desc, weight):
        self.par = np.ascontiguousarray(parent, "i4"); self.desc = np.ascontiguousarray(desc, np.uint8); self.w = np.ascontiguousarray(weight, "f8")
        lib().orc_voc_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_voc_create(int(k), int(L), len(self.par), _p(self.par), _p(self.desc), _p(self.w)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_voc_destroy(self.h); self.h = None

    def transform_features(self, feat, levelsup=4):
        f = np.ascontiguousarray(feat, np.uint8); n = len(f)
        wid = np.zeros(max(n, 1), "i4"); w = np.zeros(max(n, 1), "f8"); nid = np.zeros(max(n, 1), "i4")
        lib().orc_voc_transform(self.h, _p(f), n, int(levelsup), _p(wid), _p(w), _p(nid))
        return wid[:n], w[:n], nid[:n]


def bow_vector(word_id, weight, node_id, weighting=0, scoring=0):
    wid = np.ascontiguousarray(word_id, "i4"); w = np.ascontiguousarray(weight, "f8"); nid = np.ascontiguousarray(node_id, "i4")
    n = len(wid)
    oid = np.zeros(max(n, 1), "i4"); oval = np.zeros(max(n, 1), "f8"); fv = np.full(max(n, 1), -1, "i4")
    m = lib().orc_bow_vector(n, _p(wid), _p(w), _p(nid), int(weighting), int(scoring), _p(oid), _p(oval), _p(fv))
    return oid[:m].copy(), oval[:m].copy(), fv[:n].copy()


def bow_score_l1(a, b):
    (i1, v1), (i2, v2) = a, b
    i1 = np.ascontiguousarray(i1, "i4"); v1 = np.ascontiguousarray(v1, "f8"); i2 = np.ascontiguousarray(i2, "i4"); v2 = np.ascontiguousarray(v2, "f8")
    lib().orc_bow_score_l1.restype = C.c_double
    return float(lib().orc_bow_score_l1(len(i1), _p(i1), _p(v1), len(i2), _p(i2), _p(v2)))


def distinctive_descriptor(desc):
    d = np.ascontiguousarray(desc, np.uint8)
    return int(lib().orc_distinctive_descriptor(_p(d), len(d)))


def optimize_sim3(sim3, fix_scale, K1, K2, P1, P2, obs1, obs2, info1, info2, th2):
    a = lambda x, t: np.ascontiguousarray(x, t)
    s = a(sim3, "f8").copy(); k1 = a(K1, "f8"); k2 = a(K2, "f8")
    A = [a(P1, "f8"), a(P2, "f8"), a(obs1, "f8"), a(obs2, "f8"), a(info1, "f8"), a(info2, "f8")]
    n = len(A[4]); inl = np.zeros(max(n, 1), np.uint8)
    nin = lib().orc_optimize_sim3(_p(s), int(fix_scale), _p(k1), _p(k2), n, *[_p(x) for x in A], C.c_float(th2), _p(inl))
    return s, inl[:n].copy(), nin


def sim3_exp(u):
    u = np.ascontiguousarray(u, "f8"); o = np.zeros(8)
    lib().orc_sim3_exp(_p(u), _p(o))
    return o


def sim3_mul(a, b):
    a = np.ascontiguousarray(a, "f8"); b = np.ascontiguousarray(b, "f8"); o = np.zeros(8)
    lib().orc_sim3_mul(_p(a), _p(b), _p(o))
    return o


def sim3_inverse(a):
    a = np.ascontiguousarray(a, "f8"); o = np.zeros(8)
    lib().orc_sim3_inverse(_p(a), _p(o))
    return o


def sim3_log(s):
    s = np.ascontiguousarray(s, "f8"); o = np.zeros(7)
    lib().orc_sim3_log(_p(s), _p(o))
    return o


def essential_graph(sim3, fixed, edge_i, edge_j, meas, fix_scale=False, iterations=20):
    a = lambda x, t: np.ascontiguousarray(x, t)
    s = a(sim3, "f8").copy(); fx = a(fixed, np.uint8); ei = a(edge_i, "i4"); ej = a(edge_j, "i4"); ms = a(meas, "f8"); chi = np.zeros(2)
    done = lib().orc_essential_graph(len(s), _p(s), _p(fx), int(fix_scale), len(ei), _p(ei), _p(ej), _p(ms), int(iterations), _p(chi))
    return s, dict(iterations_done=done, chi2_initial=chi[0], chi2_final=chi[1])


def ess_set_solver(mode: int):
    """Essential graph: 0 automatic (block-sparse Cholesky above 400 free vertices), 1 dense Cholesky, 2 block-sparse Cholesky."""
    lib().orc_ess_set_solver(int(mode))


def ba_set_solver(mode: int):
    """0 automatic (block-sparse Cholesky above 400 free keyframes), 1 dense Cholesky, 2 block-sparse Cholesky."""
    lib().orc_ba_set_solver(int(mode))


def ba_solve_once(g, huber_delta, lam, mode=0):
    """One linearisation + reduced solve with the chosen solver: (xp [nfree,6], xl [L,3], stats) or None if not SPD."""
    pb, keep = _ba_problem(g)
    nfree = int((keep["fixed"] == 0).sum())
    xp = np.zeros((nfree, 6)); xl = np.zeros((len(keep["points"]), 3)); st = np.zeros(3)
    P = lib().orc_ba_solve_once(C.byref(pb), C.c_double(huber_delta), C.c_double(lam), int(mode), _p(xp), _p(xl), _p(st))
    if P < 0:
        return None
    assert P == nfree
    return xp, xl, dict(factor_blocks=int(st[0]), factor_flops=float(st[1]), schur_upper_blocks=int(st[2]))


def bench_extract_match_mt(par: OrbParams, frames: np.ndarray, threads: int):
    """All-core CPU baseline: frames [n,h,w] uint8 dealt to `threads` POSIX threads inside the oracle (no GIL involved), then the
    n - 1 consecutive pairs.  Returns (keypoints, extraction seconds, matching seconds)."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    sec = np.zeros(2)
    L = lib()
    L.orc_bench_extract_match_mt.restype = C.c_long
    tot = L.orc_bench_extract_match_mt(C.byref(par), _p(frames), n, w, h, int(threads), _p(sec))
    assert tot >= 0
    return int(tot), float(sec[0]), float(sec[1])
