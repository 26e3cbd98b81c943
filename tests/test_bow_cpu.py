"""Vocabulary-tree oracle known answers and the host-side BowVector arithmetic of the library (no GPU needed):
DBoW2 transform / addWeight / normalize / L1 score and MapPoint::ComputeDistinctiveDescriptors."""
import ctypes as C

import numpy as np

from motioncheck_ccm_slam_amd import _lib


def _desc(*bits):
    d = np.zeros(256, np.uint8); d[list(bits)] = 1
    return np.packbits(d, bitorder="little")


def test_transform_known_answer(oracle):
    # root -> A(1), B(2); A -> A0(3), A1(4); B is a word.  Words in node order: B=0, A0=1, A1=2
    parent = [0, 0, 0, 1, 1]
    desc = np.stack([_desc(), _desc(0, 1, 2, 3), _desc(100, 101), _desc(0, 1, 2, 3, 4), _desc(0, 1, 2, 3, 50, 51)])
    w = np.array([0.0, 0.0, 2.0, 3.0, 5.0])
    voc = oracle.Voc(2, 2, parent, desc, w)
    feats = np.stack([_desc(0, 1, 2, 3, 4), _desc(100), _desc(0, 1, 2, 3, 50), _desc(0, 1)])
    wid, ww, nid = voc.transform_features(feats, levelsup=1)            # nid_level = 1
    assert wid.tolist() == [1, 0, 2, 1]                                  # last: A (dist 2 < 4), then A0 (3) vs A1 (4) -> A0
    assert ww.tolist() == [3.0, 2.0, 5.0, 3.0]
    assert nid.tolist() == [1, 2, 1, 1]
    _, _, nid0 = voc.transform_features(feats, levelsup=2)               # nid_level 0 -> root
    assert nid0.tolist() == [0, 0, 0, 0]
    # a tie between the two children of the root keeps the first (strict <)
    assert voc.transform_features(_desc(0, 1, 100)[None], 1)[2].tolist() == [1]       # 3 vs 3


def test_bow_vector_and_score(oracle):
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n = 500
    wid = rng.integers(0, 120, n).astype("i4"); w = rng.uniform(0.1, 9, n); w[rng.random(n) < 0.05] = 0.0
    nid = rng.integers(1, 40, n).astype("i4")
    # hand-checked small case: TF_IDF + L1
    oid, oval, fv = oracle.bow_vector([7, 3, 7, 9], [1.0, 2.0, 0.5, 0.0], [11, 12, 11, 13])
    assert oid.tolist() == [3, 7] and oval.tolist() == [2.0 / 3.5, 1.5 / 3.5] and fv.tolist() == [11, 12, 11, -1]
    vecs = {}
    for weighting in range(4):
        for scoring in (0, 1, 5):
            ref = oracle.bow_vector(wid, w, nid, weighting, scoring)
            oi = np.zeros(n, "i4"); ov = np.zeros(n, "f8"); f = np.zeros(n, "i4")
            m = lib.ccm_bow_vector(n, _lib.ptr(wid), _lib.ptr(w), _lib.ptr(nid), weighting, scoring, _lib.ptr(oi), _lib.ptr(ov), _lib.ptr(f))
            assert m == len(ref[0]) and (oi[:m] == ref[0]).all() and (ov[:m] == ref[1]).all() and (f == ref[2]).all()   # bit-exact doubles
            vecs[(weighting, scoring)] = (oi[:m].copy(), ov[:m].copy())
            if scoring == 0:
                assert abs(ov[:m].sum() - 1.0) < 1e-12
    a = vecs[(0, 0)]
    wid2 = wid.copy(); wid2[:200] = rng.integers(100, 300, 200)
    b = oracle.bow_vector(wid2, w, nid)[:2]
    s = lib.ccm_bow_score_l1(len(a[0]), _lib.ptr(a[0]), _lib.ptr(a[1]), len(b[0]), _lib.ptr(b[0]), _lib.ptr(b[1]))
    assert s == oracle.bow_score_l1(a, b) and 0.0 < s < 1.0
    assert abs(lib.ccm_bow_score_l1(len(a[0]), _lib.ptr(a[0]), _lib.ptr(a[1]), len(a[0]), _lib.ptr(a[0]), _lib.ptr(a[1])) - 1.0) < 1e-12


def test_distinctive_descriptor_known_answer(oracle):
    # 3 descriptors: distances d01 = 2, d02 = 6, d12 = 4; sorted rows {0,2,6} {0,2,4} {0,4,6}; median index (int)(0.5*2) = 1
    d = np.stack([_desc(0, 1), _desc(), _desc(2, 3, 4, 5)])
    assert oracle.distance(d[0], d[1]) == 2 and oracle.distance(d[0], d[2]) == 6 and oracle.distance(d[1], d[2]) == 4
    assert oracle.distinctive_descriptor(d) == 0                         # medians 2, 2, 4 -> first of the two
    assert oracle.distinctive_descriptor(d[[2, 1, 0]]) == 1
    assert oracle.distinctive_descriptor(d[:1]) == 0
    # even N: the lower middle element, (int)(0.5 * 3) = 1
    d4 = np.stack([_desc(), _desc(0), _desc(0, 1), _desc(0, 1, 2, 3, 4, 5)])
    # rows sorted: {0,1,2,6} {0,1,1,5} {0,1,2,4} {0,4,5,6}: medians 1,1,1,4
    assert oracle.distinctive_descriptor(d4) == 0
