"""ctypes binding of libccm_hot.so (the C ABI declared in include/ccm_hot.h).

There is no CPU fallback: if the shared library is missing or a GPU call fails,
the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCM_HOT_LIB: another build of the same ABI (A/B timing of kernel variants, tools/ab_orb.sh); the ABI version is checked either way
LIB_PATH = os.environ.get("CCM_HOT_LIB") or os.path.join(_HERE, "libccm_hot.so")

CCM_OK = 0
ERRORS = {-1: "CCM_E_ARG", -2: "CCM_E_DEVICE", -3: "CCM_E_NOMEM", -4: "CCM_E_CAPACITY",
          -5: "CCM_E_NUMERIC", -6: "CCM_E_COMM", -7: "CCM_E_STATE"}
COMM_ID_BYTES = 128

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])

# every symbol include/ccm_hot.h declares (tests check the library exports all of them)
SYMBOLS = [
    "ccm_abi_version", "ccm_create", "ccm_destroy", "ccm_last_error", "ccm_sync", "ccm_stream",
    "ccm_host_register", "ccm_host_unregister", "ccm_profile_enable", "ccm_profile_read",
    "ccm_orb_tables", "ccm_orb_level_sizes", "ccm_orb_extract", "ccm_orb_extract_dev", "ccm_orb_fetch",
    "ccm_orb_result_dev", "ccm_orb_debug_level", "ccm_orb_debug_candidates",
    "ccm_descriptor_distance", "ccm_hamming_match", "ccm_hamming_match_dev", "ccm_ratio_test", "ccm_match_bow",
    "ccm_window_candidates", "ccm_search_by_projection", "ccm_search_by_projection_frame", "ccm_search_for_initialization", "ccm_fuse_select", "ccm_fuse_select_batch", "ccm_search_by_sim3", "ccm_search_by_projection_sim3", "ccm_search_by_projection_sim3_batch",
    "ccm_search_for_triangulation", "ccm_voc_create", "ccm_voc_destroy", "ccm_voc_words", "ccm_voc_transform", "ccm_voc_transform_dev",
    "ccm_bow_vector", "ccm_bow_score_l1", "ccm_distinctive_descriptors", "ccm_optimize_sim3", "ccm_optimize_essential_graph", "ccm_correct_map_points",
    "ccm_ba_solve", "ccm_ba_landmark_cuts", "ccm_pose_optimize", "ccm_comm_unique_id", "ccm_comm_init", "ccm_comm_attach", "ccm_comm_destroy",
    "ccm_pose_from_mat4f", "ccm_pose_to_mat4f",
]


class CcmError(RuntimeError):
    def __init__(self, code, text=""):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "CCM_E_?"), code, text))
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int)]


class BowOptions(C.Structure):
    _fields_ = [("nnratio", C.c_float), ("check_ori", C.c_int), ("th", C.c_int), ("strict_th", C.c_int)]


class FrameGrid(C.Structure):
    _fields_ = [("n", C.c_int), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p), ("kp_octave", C.c_void_p), ("desc", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("inv_w", C.c_float), ("inv_h", C.c_float),
                ("grid_cols", C.c_int), ("grid_rows", C.c_int)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("poses", C.c_void_p), ("fixed", C.c_void_p), ("intr", C.c_void_p),
                ("n_points", C.c_int), ("points", C.c_void_p),
                ("n_edges", C.c_int), ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p),
                ("obs", C.c_void_p), ("info", C.c_void_p)]


class PoseProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("poses", C.c_void_p), ("intr", C.c_void_p), ("first", C.c_void_p),
                ("points", C.c_void_p), ("obs", C.c_void_p), ("info", C.c_void_p), ("outlier", C.c_void_p),
                ("n_inliers", C.c_void_p)]


class Sim3Problem(C.Structure):
    _fields_ = [("n_problems", C.c_int), ("sim3", C.c_void_p), ("fix_scale", C.c_void_p), ("K1", C.c_void_p), ("K2", C.c_void_p),
                ("first", C.c_void_p), ("P1", C.c_void_p), ("P2", C.c_void_p), ("obs1", C.c_void_p), ("obs2", C.c_void_p),
                ("info1", C.c_void_p), ("info2", C.c_void_p), ("th2", C.c_void_p), ("inlier", C.c_void_p), ("n_inliers", C.c_void_p)]


class EssentialGraph(C.Structure):
    _fields_ = [("n_vertices", C.c_int), ("sim3", C.c_void_p), ("fixed", C.c_void_p), ("fix_scale", C.c_int), ("n_edges", C.c_int),
                ("edge_i", C.c_void_p), ("edge_j", C.c_void_p), ("measurement", C.c_void_p), ("iterations", C.c_int),
                ("iterations_done", C.c_int), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("factor_blocks", C.c_int), ("factor_rounds", C.c_int), ("solver_bytes", C.c_int64)]


class BaOptions(C.Structure):
    _fields_ = [("iterations", C.c_int), ("huber_delta", C.c_double), ("iterations2", C.c_int),
                ("outlier_chi2", C.c_double), ("stop_flag", C.c_void_p), ("pcg_tol", C.c_double)]


class BaResult(C.Structure):
    _fields_ = [("iterations_done", C.c_int), ("trials", C.c_int), ("chi2_initial", C.c_double),
                ("chi2_final", C.c_double), ("lambda_final", C.c_double), ("stopped", C.c_int),
                ("t_linearize", C.c_double), ("t_schur", C.c_double), ("t_solve", C.c_double),
                ("t_update", C.c_double), ("edge_outlier", C.c_void_p),
                ("schur_blocks", C.c_int32), ("schur_pairs", C.c_int64), ("pcg_iterations", C.c_int32),
                ("pcg_fallbacks", C.c_int32), ("pcg_pipelined", C.c_int32)]


ABI_VERSION = 3        # CCM_ABI_VERSION of the include/ccm_hot.h these ctypes structures were written against
_lib = None


def load():
    """Load libccm_hot.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C motioncheck_ccm_slam_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    # the structures below mirror include/ccm_hot.h at this version; a stale library would read and write past them
    if lib.ccm_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, this package mirrors version %d of include/ccm_hot.h: rebuild it"
                          % (LIB_PATH, lib.ccm_abi_version(), ABI_VERSION))
    lib.ccm_create.restype = C.c_void_p
    lib.ccm_create.argtypes = [C.c_int, C.c_int]
    lib.ccm_destroy.argtypes = [C.c_void_p]
    lib.ccm_destroy.restype = None
    lib.ccm_last_error.restype = C.c_char_p
    lib.ccm_last_error.argtypes = [C.c_void_p]
    lib.ccm_sync.argtypes = [C.c_void_p]
    lib.ccm_stream.restype = C.c_void_p
    lib.ccm_stream.argtypes = [C.c_void_p]
    vp = C.c_void_p
    lib.ccm_host_register.argtypes = [vp, vp, C.c_size_t]
    lib.ccm_host_unregister.argtypes = [vp, vp]
    lib.ccm_profile_enable.argtypes = [vp, C.c_int]
    lib.ccm_profile_read.argtypes = [vp, vp, vp]
    lib.ccm_orb_tables.argtypes = [C.POINTER(OrbParams)] + [vp] * 6
    lib.ccm_orb_level_sizes.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int, vp, vp]
    lib.ccm_orb_extract.argtypes = [vp, C.POINTER(OrbParams), vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                    vp, vp, vp, C.c_int]
    lib.ccm_orb_extract_dev.argtypes = [vp, C.POINTER(OrbParams), vp, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                        C.c_int, C.c_int]
    lib.ccm_orb_fetch.argtypes = [vp, vp, vp, vp]
    lib.ccm_orb_result_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]
    lib.ccm_orb_debug_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    lib.ccm_orb_debug_candidates.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int]
    lib.ccm_descriptor_distance.argtypes = [vp, vp]
    lib.ccm_hamming_match.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.ccm_hamming_match_dev.argtypes = [vp, vp, C.c_int, C.c_size_t, vp, C.c_int, C.c_size_t, C.c_int,
                                          vp, vp, vp, vp, vp]
    lib.ccm_ratio_test.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
    lib.ccm_match_bow.argtypes = [vp, C.POINTER(BowOptions), vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.ccm_window_candidates.argtypes = [vp, C.POINTER(FrameGrid), C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    lib.ccm_search_by_projection.argtypes = [vp, C.POINTER(FrameGrid), vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp]
    lib.ccm_search_by_projection_frame.argtypes = [vp, C.POINTER(FrameGrid), vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp]
    lib.ccm_search_for_initialization.argtypes = [vp, C.c_int, vp, vp, vp, C.POINTER(FrameGrid), vp, vp, C.c_int, C.c_float, C.c_int, vp]
    lib.ccm_fuse_select.argtypes = [vp, C.POINTER(FrameGrid), vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp]
    lib.ccm_fuse_select_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp]
    lib.ccm_search_by_sim3.argtypes = [vp, C.POINTER(FrameGrid), vp, C.POINTER(FrameGrid), vp] + [vp] * 10 + [C.c_float, vp]
    lib.ccm_search_by_projection_sim3.argtypes = [vp, C.POINTER(FrameGrid), vp, C.c_int] + [vp] * 7 + [C.c_float, vp]
    lib.ccm_search_by_projection_sim3_batch.argtypes = [vp, C.c_int, vp, vp, vp] + [vp] * 7 + [C.c_float, vp, vp]
    lib.ccm_search_for_triangulation.argtypes = [vp] * 7 + [C.c_int] + [vp] * 7 + [C.c_int, vp, C.c_float, C.c_float, vp, vp, C.c_int, vp]
    lib.ccm_voc_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)]
    lib.ccm_voc_destroy.argtypes = [vp]; lib.ccm_voc_destroy.restype = None
    lib.ccm_voc_words.argtypes = [vp]
    lib.ccm_voc_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.ccm_voc_transform_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.ccm_bow_vector.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.ccm_bow_score_l1.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp]; lib.ccm_bow_score_l1.restype = C.c_double
    lib.ccm_distinctive_descriptors.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    lib.ccm_ba_solve.argtypes = [vp, C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(BaResult)]
    lib.ccm_ba_landmark_cuts.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.ccm_comm_attach.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.ccm_pose_optimize.argtypes = [vp, C.POINTER(PoseProblem)]
    lib.ccm_optimize_sim3.argtypes = [vp, C.POINTER(Sim3Problem)]
    lib.ccm_optimize_essential_graph.argtypes = [vp, C.POINTER(EssentialGraph)]
    lib.ccm_correct_map_points.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp]
    lib.ccm_comm_unique_id.argtypes = [vp]
    lib.ccm_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.ccm_comm_destroy.argtypes = [vp]
    lib.ccm_pose_from_mat4f.argtypes = [vp, vp]
    lib.ccm_pose_to_mat4f.argtypes = [vp, vp]
    _lib = lib
    return lib


def ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One ccm_ctx: a HIP stream plus device workspaces.  Not thread-safe; one per calling thread."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.handle = self.lib.ccm_create(int(device), 0)
        if not self.handle:
            raise CcmError(-2, "ccm_create(%d) failed: no such HIP device" % device)
        self.device = device

    def check(self, rc: int) -> int:
        if rc < 0:
            raise CcmError(rc, self.lib.ccm_last_error(self.handle).decode(errors="replace"))
        return rc

    PROF_LABELS = ("k_pyr_resize", "k_fast_score", "k_cell_nms", "k_octree", "k_orient_desc", "k_hamming_bf",
                   "ba_linearize", "ba_dinv_y", "k_sp_schur_blocks", "k_sp_bschur", "k_ba_backsub")

    def host_register(self, arr):
        """Page-lock a numpy array the caller keeps re-using as an input / output buffer (ccm_host_register)."""
        self.check(self.lib.ccm_host_register(self.handle, ptr(arr), C.c_size_t(arr.nbytes)))

    def host_unregister(self, arr):
        self.check(self.lib.ccm_host_unregister(self.handle, ptr(arr)))

    def profile(self, on: bool):
        self.check(self.lib.ccm_profile_enable(self.handle, int(on)))

    def profile_read(self):
        """{kernel: (total_ms, launches)} since the last read (HIP events on this context's stream)."""
        ms = np.zeros(len(self.PROF_LABELS), "f4"); n = np.zeros(len(self.PROF_LABELS), "i4")
        self.check(self.lib.ccm_profile_read(self.handle, ptr(ms), ptr(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.PROF_LABELS)}

    def sync(self):
        self.check(self.lib.ccm_sync(self.handle))

    @property
    def stream(self) -> int:
        return self.lib.ccm_stream(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
