"""Host-side mirror of `cslam::ORBextractor` (include/cslam/ORBextractor.h:97-164) over the C ABI.

Same constructor arguments, same getters, `__call__(image, mask)` returns
(keypoints, descriptors) the way `operator()(image, mask, keypoints, descriptors)`
fills its outputs.  Batched and device-resident variants are extensions for the
GPU (a batch of frames per launch).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbParams


class ORBextractor:
    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 ctx: _lib.Context | None = None):
        self.lib = _lib.load()
        self.par = OrbParams(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))
        n = self.par.nlevels
        self._sc = np.zeros(n, "f4"); self._isc = np.zeros(n, "f4")
        self._s2 = np.zeros(n, "f4"); self._is2 = np.zeros(n, "f4")
        self.mnFeaturesPerLevel = np.zeros(n, "i4"); self.umax = np.zeros(16, "i4")
        rc = self.lib.ccm_orb_tables(C.byref(self.par), _lib.ptr(self._sc), _lib.ptr(self._isc), _lib.ptr(self._s2),
                                     _lib.ptr(self._is2), _lib.ptr(self.mnFeaturesPerLevel), _lib.ptr(self.umax))
        if rc:
            raise _lib.CcmError(rc, "bad ORBextractor parameters")
        self._ctx = ctx
        self.max_per_image = int(nfeatures) + 4 * n + 64      # the quadtree may overshoot each quota by up to 3
        self._last = None

    # ---- getters (ORBextractor.h:120-150)
    def GetLevels(self): return self.par.nlevels
    def GetScaleFactor(self): return self.par.scale_factor
    def GetScaleFactors(self): return self._sc.copy()
    def GetInverseScaleFactors(self): return self._isc.copy()
    def GetScaleSigmaSquares(self): return self._s2.copy()
    def GetInverseScaleSigmaSquares(self): return self._is2.copy()

    @property
    def ctx(self) -> _lib.Context:
        if self._ctx is None:
            self._ctx = _lib.default_context(0)
        return self._ctx

    def level_sizes(self, w: int, h: int):
        lw = np.zeros(self.par.nlevels, "i4"); lh = np.zeros(self.par.nlevels, "i4")
        self.lib.ccm_orb_level_sizes(C.byref(self.par), w, h, _lib.ptr(lw), _lib.ptr(lh))
        return lw, lh

    # ---- operator()
    def __call__(self, image: np.ndarray, mask=None):
        """One CV_8UC1 image -> (keypoints[KP_DTYPE], descriptors[N,32] uint8).  The mask is ignored,
        as in the reference (ORBextractor.cpp:1216-1278).  An empty image returns empty outputs."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "image must be CV_8UC1"
        kps, desc, counts = self.extract_batch(image[None])
        n = int(counts[0])
        return kps[0, :n].copy(), desc[0, :n].copy()

    def extract_batch(self, images: np.ndarray, out=None):
        """images [B,H,W] uint8 (host) -> kps [B,max], desc [B,max,32], counts [B].  `out` = (kps, desc, counts) of those shapes
        re-uses the caller's buffers (a server keeps its frame and result pools page-locked with Context.host_register: the results
        of a chunk of frames then go down while the next chunk is extracted)."""
        images = np.ascontiguousarray(images, np.uint8)
        b, h, w = images.shape
        m = self.max_per_image
        if out is not None:
            kps, desc, counts = out
            assert kps.shape == (b, m) and kps.dtype == KP_DTYPE and desc.shape == (b, m, 32) and desc.dtype == np.uint8 and counts.shape == (b,) and counts.dtype == np.int32
            assert kps.flags.c_contiguous and desc.flags.c_contiguous and counts.flags.c_contiguous
        else:
            kps = np.zeros((b, m), KP_DTYPE); desc = np.zeros((b, m, 32), np.uint8); counts = np.zeros(b, "i4")
        self.ctx.check(self.lib.ccm_orb_extract(self.ctx.handle, C.byref(self.par), _lib.ptr(images), w, h, w,
                                                C.c_size_t(w * h), b, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(counts), m))
        self._last = (b, w, h)
        return kps, desc, counts

    def extract_dev(self, img_ptr: int, w: int, h: int, stride: int, image_stride: int, n_images: int):
        """Device-resident batch (img_ptr = device address); asynchronous on the context's stream."""
        self.ctx.check(self.lib.ccm_orb_extract_dev(self.ctx.handle, C.byref(self.par), C.c_void_p(img_ptr), w, h, stride,
                                                    C.c_size_t(image_stride), n_images, self.max_per_image))
        self._last = (n_images, w, h)

    def fetch(self):
        b = self._last[0]; m = self.max_per_image
        kps = np.zeros((b, m), KP_DTYPE); desc = np.zeros((b, m, 32), np.uint8); counts = np.zeros(b, "i4")
        self.ctx.check(self.lib.ccm_orb_fetch(self.ctx.handle, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(counts)))
        return kps, desc, counts

    def result_dev(self):
        d = C.c_void_p(); c = C.c_void_p(); m = C.c_int()
        self.ctx.check(self.lib.ccm_orb_result_dev(self.ctx.handle, C.byref(d), C.byref(c), C.byref(m)))
        return d.value, c.value, m.value

    # ---- debug taps (mvImagePyramid and the pre-quadtree FAST corners)
    def image_pyramid_level(self, image: int, level: int) -> np.ndarray:
        lw, lh = self.level_sizes(self._last[1], self._last[2])
        out = np.zeros((lh[level], lw[level]), np.uint8)
        self.ctx.check(self.lib.ccm_orb_debug_level(self.ctx.handle, image, level, _lib.ptr(out), int(lw[level])))
        return out

    def fast_candidates(self, image: int, level: int, cap: int = 1 << 18):
        xy = np.zeros((cap, 2), "i4"); sc = np.zeros(cap, "i4")
        n = self.ctx.check(self.lib.ccm_orb_debug_candidates(self.ctx.handle, image, level, _lib.ptr(xy), _lib.ptr(sc), cap))
        assert n <= cap
        return xy[:n].copy(), sc[:n].copy()
