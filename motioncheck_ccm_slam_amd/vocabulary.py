"""Host-side mirror of `DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>` (ORBVocabulary) for the calls the SLAM
makes: `transform` (Frame::ComputeBoW / KeyFrame::ComputeBoW, src/Frame.cpp:268-275), `score`, and the batched
`MapPoint::ComputeDistinctiveDescriptors` (src/MapPoint.cpp:929-994)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

TF_IDF, TF, IDF, BINARY = range(4)                      # BowVector.h:36-42
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)


class ORBVocabulary:
    def __init__(self, k: int, L: int, parent, descriptors, weights, weighting: int = TF_IDF, scoring: int = L1_NORM,
                 ctx: _lib.Context | None = None):
        self.lib = _lib.load()
        self.ctx = ctx if ctx is not None else _lib.default_context(0)
        self.k, self.L, self.weighting, self.scoring = int(k), int(L), int(weighting), int(scoring)
        par = np.ascontiguousarray(parent, "i4"); d = np.ascontiguousarray(descriptors, np.uint8); w = np.ascontiguousarray(weights, "f8")
        assert d.shape == (len(par), 32) and len(w) == len(par)
        h = C.c_void_p()
        self.ctx.check(self.lib.ccm_voc_create(self.ctx.handle, self.k, self.L, len(par), _lib.ptr(par), _lib.ptr(d), _lib.ptr(w), C.byref(h)))
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.ccm_voc_destroy(h); self.handle = None

    def size(self) -> int:
        return self.lib.ccm_voc_words(self.handle)

    def transform_features(self, desc, levelsup: int = 4):
        """Per feature: (word id, word weight, node id at level L - levelsup)."""
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        wid = np.zeros(max(n, 1), "i4"); w = np.zeros(max(n, 1), "f8"); nid = np.zeros(max(n, 1), "i4")
        self.ctx.check(self.lib.ccm_voc_transform(self.handle, _lib.ptr(d), n, int(levelsup), _lib.ptr(wid), _lib.ptr(w), _lib.ptr(nid)))
        return wid[:n], w[:n], nid[:n]

    def transform(self, desc, levelsup: int = 4):
        """transform(features, BowVector, FeatureVector, levelsup): returns (word ids, values, node per feature)."""
        wid, w, nid = self.transform_features(desc, levelsup)
        n = len(wid)
        oid = np.zeros(max(n, 1), "i4"); oval = np.zeros(max(n, 1), "f8"); fv = np.full(max(n, 1), -1, "i4")
        m = self.lib.ccm_bow_vector(n, _lib.ptr(wid), _lib.ptr(w), _lib.ptr(nid), self.weighting, self.scoring, _lib.ptr(oid), _lib.ptr(oval), _lib.ptr(fv))
        if m < 0:
            raise ValueError("ccm_bow_vector: bad arguments")
        return oid[:m].copy(), oval[:m].copy(), fv[:n].copy()

    def score(self, a, b) -> float:
        (i1, v1), (i2, v2) = a, b
        i1 = np.ascontiguousarray(i1, "i4"); v1 = np.ascontiguousarray(v1, "f8"); i2 = np.ascontiguousarray(i2, "i4"); v2 = np.ascontiguousarray(v2, "f8")
        return float(self.lib.ccm_bow_score_l1(len(i1), _lib.ptr(i1), _lib.ptr(v1), len(i2), _lib.ptr(i2), _lib.ptr(v2)))

    def distinctive_descriptors(self, desc, first, count):
        d = np.ascontiguousarray(desc, np.uint8); f = np.ascontiguousarray(first, "i8"); c = np.ascontiguousarray(count, "i4")
        best = np.full(max(len(c), 1), -1, "i4")
        self.ctx.check(self.lib.ccm_distinctive_descriptors(self.handle, _lib.ptr(d), _lib.ptr(f), _lib.ptr(c), len(c), _lib.ptr(best)))
        return best[:len(c)]


def synthetic_tree(k: int, L: int, seed: int = 0, ragged: bool = True):
    """A vocabulary of the shape of ORBvoc (k-way, depth L) with random centres: children are noisy copies of their
    parent so that the descent is non-trivial; with `ragged` some branches end early and some weights are 0 (stopped
    words).  Returns (parent, descriptors, weights) in loadFromTextFile's node order (depth first)."""
    rng = np.random.default_rng(seed)
    parent = [0]; desc = [np.zeros(32, np.uint8)]; depth = [0]
    stack = [0]
    while stack:
        p = stack.pop()
        if depth[p] >= L or (ragged and depth[p] >= 2 and rng.random() < 0.08):
            continue
        kids = []
        for _ in range(k if not ragged else int(rng.integers(max(2, k - 3), k + 1))):
            flips = np.packbits(rng.random(256) < (0.5 if p == 0 else 0.18), bitorder="little")
            parent.append(p); desc.append(desc[p] ^ flips); depth.append(depth[p] + 1)
            kids.append(len(parent) - 1)
        stack.extend(reversed(kids))
    n = len(parent)
    w = rng.uniform(0.5, 9.0, n)
    if ragged:
        w[rng.random(n) < 0.02] = 0.0
    return np.array(parent, "i4"), np.stack(desc), w
