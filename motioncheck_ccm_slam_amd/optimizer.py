"""Host-side mirror of `cslam::Optimizer` (include/cslam/Optimizer.h:84-97) over the C ABI.

The reference's entry points take Map / KeyFrame / MapPoint objects and build a g2o graph
(src/Optimizer.cpp:48-164, 406-530, 676-791).  Here the graph is passed as arrays -- exactly what
those functions extract: SE3Quat poses of the keyframes, fixed flags, intrinsics, world points,
one edge per observation with its pixel measurement and invSigma2 -- and the results come back
where the reference writes them (poses / points), plus the outlier edges that
LocalBundleAdjustmentClient erases (src/Optimizer.cpp:570-602).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BaOptions, BaProblem, BaResult

TH_HUBER_2D_GLOBAL = float(np.float32(np.sqrt(5.99)))    # src/Optimizer.cpp:81, :712 (`const float thHuber2D = sqrt(5.99)`: rounded to float)
TH_HUBER_2D_LOCAL = float(np.float32(np.sqrt(5.991)))    # src/Optimizer.cpp:468 (`const float thHuberMono`)
CHI2_MONO = 5.991                            # src/Optimizer.cpp:556, :582


def _stop_flag_view(stop_flag):
    """pbStopFlag is a `bool*` another thread writes while the solve runs (src/Optimizer.cpp:160, :531): the library must
    poll the caller's own byte, so only a writable one-byte array (uint8 / bool) is accepted -- a Python bool, a list or
    an array of another dtype would be copied and could never stop the solve."""
    if stop_flag is None:
        return None
    if not isinstance(stop_flag, np.ndarray) or stop_flag.dtype.itemsize != 1 or stop_flag.size < 1 \
            or not stop_flag.flags.c_contiguous or not stop_flag.flags.writeable:
        raise TypeError("pbStopFlag must be a writable numpy array of uint8 or bool (the library polls its first byte in place)")
    return stop_flag.view(np.uint8)


class BaWorkspace:
    """Page-locked in/out arrays for the keyframe poses and the map points, kept by a caller that optimises maps again and again (a
    server's GBA thread): ccm_ba_solve reads the start values from and writes the results to the caller's own arrays, and a copy
    between the GPU and freshly allocated pageable memory costs several times what the PCIe transfer does (config 5: 0.9 ms for the
    4.9 MB of results).  A solve that is given a workspace copies the graph's start values into it and returns VIEWS of it: they are
    overwritten by the next solve with the same workspace."""

    def __init__(self, ctx, n_poses: int, n_points: int):
        self.ctx = ctx
        self.poses = np.empty((max(int(n_poses), 1), 7), "f8")
        self.points = np.empty((max(int(n_points), 1), 3), "f8")
        ctx.host_register(self.poses)
        ctx.host_register(self.points)
        self._open = True

    def close(self):
        if self._open:
            self._open = False
            self.ctx.host_unregister(self.poses)
            self.ctx.host_unregister(self.points)

    def take(self, poses, points):
        poses = np.asarray(poses, "f8").reshape(-1, 7); points = np.asarray(points, "f8").reshape(-1, 3)
        if len(poses) > len(self.poses) or len(points) > len(self.points):
            raise ValueError("BaWorkspace holds %d poses / %d points, the graph has %d / %d"
                             % (len(self.poses), len(self.points), len(poses), len(points)))
        a = self.poses[:len(poses)]; b = self.points[:len(points)]
        np.copyto(a, poses); np.copyto(b, points)
        return a, b


def _solve(ctx, graph, iterations, huber, iterations2=0, stop_flag=None, pcg_tol=0.0, want_outliers=False, workspace=None):
    """want_outliers: the per-edge chi2 / depth test on the final state, which only LocalBundleAdjustmentClient uses
    (src/Optimizer.cpp:577-595); the global entry points leave `edge_outlier` NULL like the shim (shim/cslam_optimizer.cpp) -- the
    test is another pass over the edges and 1 byte per edge back to the host."""
    lib = _lib.load()
    if workspace is not None:
        poses0, points0 = workspace.take(graph["poses"], graph["points"])
    else:
        poses0 = np.ascontiguousarray(graph["poses"], "f8").copy(); points0 = np.ascontiguousarray(graph["points"], "f8").copy()
    keep = dict(
        poses=poses0, fixed=np.ascontiguousarray(graph["fixed"], np.uint8),
        intr=np.ascontiguousarray(graph["intr"], "f8"), points=points0,
        edge_pose=np.ascontiguousarray(graph["edge_pose"], "i4"), edge_point=np.ascontiguousarray(graph["edge_point"], "i4"),
        obs=np.ascontiguousarray(graph["obs"], "f8"), info=np.ascontiguousarray(graph["info"], "f8"))
    p = _lib.ptr
    pb = BaProblem(len(keep["poses"]), p(keep["poses"]), p(keep["fixed"]), p(keep["intr"]),
                   len(keep["points"]), p(keep["points"]), len(keep["edge_pose"]),
                   p(keep["edge_pose"]), p(keep["edge_point"]), p(keep["obs"]), p(keep["info"]))
    outl = np.zeros(max(len(keep["edge_pose"]), 1), np.uint8) if want_outliers else None
    flag = _stop_flag_view(stop_flag)
    opt = BaOptions(int(iterations), float(huber), int(iterations2), CHI2_MONO, p(flag), float(pcg_tol))
    res = BaResult()
    res.edge_outlier = p(outl) if want_outliers else None
    ctx.check(lib.ccm_ba_solve(ctx.handle, C.byref(pb), C.byref(opt), C.byref(res)))
    return dict(poses=keep["poses"], points=keep["points"], outlier=outl[:len(keep["edge_pose"])] if want_outliers else None,
                iterations_done=res.iterations_done, trials=res.trials, chi2_initial=res.chi2_initial,
                chi2_final=res.chi2_final, lambda_final=res.lambda_final, stopped=bool(res.stopped),
                t_linearize=res.t_linearize, t_schur=res.t_schur, t_solve=res.t_solve, t_update=res.t_update,
                schur_blocks=res.schur_blocks, schur_pairs=res.schur_pairs, pcg_iterations=res.pcg_iterations,
                pcg_fallbacks=res.pcg_fallbacks, pcg_pipelined=res.pcg_pipelined)


class Optimizer:
    """Static-method style like the reference; `ctx` selects the GPU context (default: device 0)."""

    @staticmethod
    def BundleAdjustmentClient(graph, nIterations: int = 5, pbStopFlag=None, bRobust: bool = True, ctx=None, workspace=None):
        ctx = ctx or _lib.default_context(0)
        return _solve(ctx, graph, nIterations, TH_HUBER_2D_GLOBAL if bRobust else 0.0, 0, pbStopFlag, workspace=workspace)

    GlobalBundleAdjustemntClient = BundleAdjustmentClient      # [sic] the reference's spelling, Optimizer.h:86

    @staticmethod
    def LocalBundleAdjustmentClient(graph, pbStopFlag=None, ctx=None):
        """5 robust iterations, outlier relabelling, 10 more without kernels (src/Optimizer.cpp:536-568)."""
        ctx = ctx or _lib.default_context(0)
        return _solve(ctx, graph, 5, TH_HUBER_2D_LOCAL, 10, pbStopFlag, want_outliers=True)

    @staticmethod
    def MapFusionGBA(graph, nIterations: int = 5, pbStopFlag=None, bRobust: bool = True, ctx=None, pcg_tol: float = 0.0, workspace=None):
        ctx = ctx or _lib.default_context(0)
        return _solve(ctx, graph, nIterations, TH_HUBER_2D_GLOBAL if bRobust else 0.0, 0, pbStopFlag, pcg_tol, workspace=workspace)


    @staticmethod
    def PoseOptimizationClient(poses, intr, first, points, obs, info, ctx=None):
        """Batched Optimizer::PoseOptimizationClient (src/Optimizer.cpp:215-347).  Frame f's correspondences are
        rows first[f]..first[f+1]-1.  Returns (poses, outlier flags, n_inliers per frame)."""
        ctx = ctx or _lib.default_context(0)
        lib = _lib.load()
        poses = np.ascontiguousarray(poses, "f8").copy(); intr = np.ascontiguousarray(intr, "f8")
        first = np.ascontiguousarray(first, "i4"); points = np.ascontiguousarray(points, "f8")
        obs = np.ascontiguousarray(obs, "f8"); info = np.ascontiguousarray(info, "f8")
        nf = len(poses)
        outl = np.zeros(max(len(info), 1), np.uint8); ninl = np.zeros(nf, "i4")
        p = _lib.ptr
        pb = _lib.PoseProblem(nf, p(poses), p(intr), p(first), p(points), p(obs), p(info), p(outl), p(ninl))
        ctx.check(lib.ccm_pose_optimize(ctx.handle, C.byref(pb)))
        return poses, outl[:len(info)], ninl


    @staticmethod
    def OptimizeSim3(sim3, fix_scale, K1, K2, first, P1, P2, obs1, obs2, info1, info2, th2, ctx=None):
        """Batched Optimizer::OptimizeSim3 (src/Optimizer.cpp:867-1062).  sim3 [n][8] = qx,qy,qz,qw,tx,ty,tz,s.
        Returns (sim3, inlier flags per correspondence, nIn per problem)."""
        ctx = ctx or _lib.default_context(0)
        lib = _lib.load()
        a = np.ascontiguousarray
        s3 = a(sim3, "f8").copy().reshape(-1, 8); n = len(s3)
        fs = a(np.broadcast_to(fix_scale, n), "i4"); k1 = a(K1, "f8"); k2 = a(K2, "f8"); fr = a(first, "i4")
        A = [a(P1, "f8"), a(P2, "f8"), a(obs1, "f8"), a(obs2, "f8"), a(info1, "f8"), a(info2, "f8")]
        t2 = a(np.broadcast_to(th2, n), "f4")
        inl = np.zeros(max(len(A[4]), 1), np.uint8); nin = np.zeros(n, "i4")
        p = _lib.ptr
        pb = _lib.Sim3Problem(n, p(s3), p(fs), p(k1), p(k2), p(fr), *[p(x) for x in A], p(t2), p(inl), p(nin))
        ctx.check(lib.ccm_optimize_sim3(ctx.handle, C.byref(pb)))
        return s3, inl[:len(A[4])], nin


    @staticmethod
    def OptimizeEssentialGraph(sim3, fixed, edge_i, edge_j, measurement, bFixScale: bool = False, iterations: int = 20, ctx=None):
        """The pose-graph optimisation of OptimizeEssentialGraphLoopClosure / MapFusion (src/Optimizer.cpp:1064-1574).
        Returns (corrected Siw per keyframe, dict with iterations_done / chi2_initial / chi2_final)."""
        ctx = ctx or _lib.default_context(0)
        lib = _lib.load()
        a = np.ascontiguousarray
        s3 = a(sim3, "f8").copy().reshape(-1, 8); fx = a(fixed, np.uint8); ei = a(edge_i, "i4"); ej = a(edge_j, "i4"); ms = a(measurement, "f8")
        p = _lib.ptr
        g = _lib.EssentialGraph(len(s3), p(s3), p(fx), int(bFixScale), len(ei), p(ei), p(ej), p(ms), int(iterations), 0, 0.0, 0.0, 0, 0, 0)
        ctx.check(lib.ccm_optimize_essential_graph(ctx.handle, C.byref(g)))
        return s3, dict(iterations_done=g.iterations_done, chi2_initial=g.chi2_initial, chi2_final=g.chi2_final,
                        factor_blocks=g.factor_blocks, factor_rounds=g.factor_rounds, solver_bytes=g.solver_bytes)

    @staticmethod
    def CorrectMapPoints(points, ref_vertex, sim3_before, sim3_after, ctx=None):
        """P' = correctedSwr.map(Srw.map(P)) per map point (src/Optimizer.cpp:1300-1330)."""
        ctx = ctx or _lib.default_context(0)
        a = np.ascontiguousarray
        pts = a(points, "f8").copy(); rv = a(ref_vertex, "i4"); sb = a(sim3_before, "f8"); sa = a(sim3_after, "f8")
        p = _lib.ptr
        ctx.check(_lib.load().ccm_correct_map_points(ctx.handle, len(pts), p(pts), p(rv), len(sb), p(sb), p(sa)))
        return pts


def pose_from_mat4f(T: np.ndarray) -> np.ndarray:
    T = np.ascontiguousarray(T, np.float32); out = np.zeros(7)
    _lib.load().ccm_pose_from_mat4f(_lib.ptr(T), _lib.ptr(out))
    return out


def pose_to_mat4f(pose: np.ndarray) -> np.ndarray:
    pose = np.ascontiguousarray(pose, np.float64); out = np.zeros((4, 4), np.float32)
    _lib.load().ccm_pose_to_mat4f(_lib.ptr(pose), _lib.ptr(out))
    return out


def pose_delta(p_a: np.ndarray, p_b: np.ndarray) -> np.ndarray:
    """|log(T_a T_b^-1)| per pose as a 6-vector magnitude bound: max(|rotation vector|, |translation diff|)."""
    def rot(q):
        x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                         np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                         np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
    Ra, Rb = rot(p_a[:, :4]), rot(p_b[:, :4])
    dR = np.einsum("nij,nkj->nik", Ra, Rb)
    v = np.stack([dR[:, 2, 1] - dR[:, 1, 2], dR[:, 0, 2] - dR[:, 2, 0], dR[:, 1, 0] - dR[:, 0, 1]], 1)
    ang = np.arctan2(0.5 * np.linalg.norm(v, axis=1), (np.trace(dR, axis1=1, axis2=2) - 1) / 2)   # exact near 0
    dt = np.linalg.norm(p_a[:, 4:] - np.einsum("nij,nj->ni", dR, p_b[:, 4:]), axis=1)
    return np.maximum(ang, dt)
