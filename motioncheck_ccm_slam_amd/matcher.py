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
