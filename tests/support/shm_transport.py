"""TEST SCAFFOLDING: attach the shared-memory all-reduce transport (shm_transport.cpp) to a ccm context.

Lets two or more processes on ONE GPU act as the ranks of the sharded global BA: the library's own code path (landmark
partition, pattern union, partial sums, collective stop flag) with the all-reduce going through a POSIX shared-memory
segment instead of RCCL.  Used by tests/ and by bench.py's rehearsal switch (CCM_BENCH_COMM=shm); never by the product."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allreduce_f64", C.c_void_p), ("allreduce_u8_max", C.c_void_p), ("destroy", C.c_void_p)]


def build():
    subprocess.check_call(["make", "-C", _HERE, "libccm_shm_transport.so"], stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "libccm_shm_transport.so")


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libccm_shm_transport.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.shm_transport_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
    return _LIB


def attach(ctx, name: str, rank: int, world: int, capacity_bytes: int = 64 << 20):
    """capacity_bytes per rank must hold the largest all-reduce: 36 doubles per reduced-camera block + 6 per keyframe."""
    from motioncheck_ccm_slam_amd import _lib as L
    t = Transport()
    rc = _lib().shm_transport_create(name.encode(), int(world), int(rank), int(capacity_bytes), C.byref(t))
    if rc:
        raise RuntimeError("shm transport %s rank %d/%d: setup failed (%d)" % (name, rank, world, rc))
    ctx.check(L.load().ccm_comm_attach(ctx.handle, C.byref(t), int(world), int(rank)))
