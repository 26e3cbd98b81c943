#!/usr/bin/env python3
"""Is rocSOLVER's dpotrf/dpotrs repeatable?  Factor and solve the same SPD system many times (optionally while a second
process does the same on the same GPU, or, with procs = 0, while a second stream of this process runs matrix products) and
report the spread of the solutions.  usage: dbg_potrf.py [n] [reps] [procs]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1434
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
procs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if procs > 1:
    ps = [subprocess.Popen([sys.executable, __file__, str(n), str(reps), "1"]) for _ in range(procs)]
    sys.exit(max(p.wait() for p in ps))
rs = C.CDLL("/opt/rocm/lib/librocsolver.so"); rb = C.CDLL("/opt/rocm/lib/librocblas.so")
h = C.c_void_p(); assert rb.rocblas_create_handle(C.byref(h)) == 0
rng = np.random.default_rng(1)
M = rng.standard_normal((n, n)); A = M @ M.T / n + np.diag(rng.uniform(1e-4, 1.0, n)); b = rng.standard_normal(n)
Ad = torch.from_numpy(A).cuda(); bd = torch.from_numpy(b).cuda(); info = torch.zeros(1, dtype=torch.int32, device="cuda")
LOWER = 122   # rocblas_fill_lower
sols = []
side = torch.cuda.Stream() if procs == 0 else None
busy_a = torch.randn(2048, 2048, device="cuda", dtype=torch.float64) if side else None
for r in range(reps):
    if side:
        with torch.cuda.stream(side):
            for _ in range(4): busy_b = busy_a @ busy_a
    W = Ad.clone(); x = bd.clone().reshape(n, 1).contiguous()
    assert rs.rocsolver_dpotrf(h, LOWER, n, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr())) == 0
    assert rs.rocsolver_dpotrs(h, LOWER, n, 1, C.c_void_p(W.data_ptr()), n, C.c_void_p(x.data_ptr()), n) == 0
    torch.cuda.synchronize()
    sols.append(x.cpu().numpy().ravel().copy())
S = np.array(sols); ref = np.linalg.solve(A, b)
dev = np.abs(S - S[0]).max(axis=1); err = np.abs(S - ref).max(axis=1) / np.abs(ref).max()
print("pid %d n=%d reps=%d: distinct results %d, max |x_r - x_0| %.3e, max rel err vs numpy %.3e (min %.3e)" %
      (os.getpid(), n, reps, len({s.tobytes() for s in sols}), dev.max(), err.max(), err.min()), flush=True)
