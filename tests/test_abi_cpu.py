"""CPU-side checks of the C ABI: the library loads, exports every declared symbol, and its
host-only entry points agree with the oracle.  No GPU work here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from motioncheck_ccm_slam_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ccm_hot.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ccm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert sorted(_lib.SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ccm_abi_version() == 3 == _lib.ABI_VERSION


def test_tables_match_oracle(oracle):
    lib = _lib.load()
    for nf, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (500, 1.5, 5), (1200, 1.1, 12), (7, 2.0, 3)):
        par = _lib.OrbParams(nf, sf, nl, 20, 7)
        sc = np.zeros(nl, "f4"); isc = np.zeros(nl, "f4"); s2 = np.zeros(nl, "f4"); is2 = np.zeros(nl, "f4")
        nfl = np.zeros(nl, "i4"); um = np.zeros(16, "i4")
        assert lib.ccm_orb_tables(C.byref(par), _lib.ptr(sc), _lib.ptr(isc), _lib.ptr(s2), _lib.ptr(is2), _lib.ptr(nfl), _lib.ptr(um)) == 0
        t = oracle.orb_tables(oracle.default_params(nf, sf, nl))
        for a, b in ((sc, t["scale"]), (isc, t["inv_scale"]), (s2, t["sigma2"]), (is2, t["inv_sigma2"]), (nfl, t["nfeat"]), (um, t["umax"])):
            assert (a == b).all()
        for w, h in ((752, 480), (1241, 376), (640, 480), (321, 243)):
            lw = np.zeros(nl, "i4"); lh = np.zeros(nl, "i4")
            assert lib.ccm_orb_level_sizes(C.byref(par), w, h, _lib.ptr(lw), _lib.ptr(lh)) == 0
            rw, rh = oracle.level_sizes(oracle.default_params(nf, sf, nl), w, h)
            assert (lw == rw).all() and (lh == rh).all()
    bad = _lib.OrbParams(1000, 1.2, 99, 20, 7)
    assert lib.ccm_orb_tables(C.byref(bad), None, None, None, None, None, None) == -1


def test_descriptor_distance_and_ratio(oracle):
    lib = _lib.load()
    a, b = synth.descriptor_pair(3, n=64)
    for i in range(64):
        assert lib.ccm_descriptor_distance(_lib.ptr(a[i]), _lib.ptr(b[i])) == oracle.distance(a[i], b[i])
    f = lib.ccm_ratio_test
    assert f(50, 100, C.c_float(0.7), 50, 0) == 1 and f(50, 100, C.c_float(0.7), 50, 1) == 0     # <= vs <
    assert f(49, 70, C.c_float(0.7), 50, 1) == 0        # 49 < 0.7*70 = 49.0 is false
    assert f(48, 70, C.c_float(0.7), 50, 1) == 1
    assert f(256, 256, C.c_float(0.9), 50, 0) == 0


def test_pose_helpers_match_oracle(oracle):
    lib = _lib.load()
    rng = np.random.default_rng(2)
    for _ in range(20):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = synth._quat_to_rot(q); T[:3, 3] = rng.normal(size=3)
        p1 = np.zeros(7); p2 = np.zeros(7)
        assert lib.ccm_pose_from_mat4f(_lib.ptr(T), _lib.ptr(p1)) == 0
        oracle.lib().orc_pose_from_mat4f(T.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p))
        assert (p1 == p2).all()
        T1 = np.zeros((4, 4), np.float32); T2 = np.zeros((4, 4), np.float32)
        lib.ccm_pose_to_mat4f(_lib.ptr(p1), _lib.ptr(T1))
        oracle.lib().orc_pose_to_mat4f(p2.ctypes.data_as(C.c_void_p), T2.ctypes.data_as(C.c_void_p))
        assert (T1 == T2).all()


def test_pattern_table_is_the_reference_one():
    import hashlib
    from motioncheck_ccm_slam_amd.pattern import PATTERN, SHA256
    assert hashlib.sha256(PATTERN.astype(np.int8).tobytes()).hexdigest() == SHA256
    assert SHA256 == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    assert PATTERN[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]          # first two pairs, ORBextractor.cpp:321-322


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.CcmError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "motioncheck_ccm_slam_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(d, f), errors="replace").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "orc_" not in txt, os.path.join(d, f)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (and as C++11), and a C program must link against the
    library and call a host-only entry point without any C++ or HIP at the call site."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "ccm_hot.h"\n'
                   '#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    unsigned char a[32] = {0}, b[32] = {0}; b[0] = 0x0F; b[31] = 0x80;\n'
                   '    ccm_orb_params p = {1000, 1.2f, 8, 20, 7};\n'
                   '    float sf[8], isf[8], s2[8], is2[8]; int nf[8], um[16];\n'
                   '    if (ccm_orb_tables(&p, sf, isf, s2, is2, nf, um) != 0) return 2;\n'
                   '    printf("%d %d %d\\n", ccm_descriptor_distance(a, b), nf[0], (int)sizeof(ccm_keypoint));\n'
                   '    return 0;\n}\n')
    inc = os.path.join(root, "include")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        if shutil.which(cc) is None:
            pytest.skip("no host compiler")
        subprocess.check_call([cc, std, "-Wall", "-Wextra", "-pedantic", "-I", inc, "-fsyntax-only"] + (["-x", "c++"] if cc == "g++" else []) + [str(src)])
    exe = tmp_path / "abi"
    libdir = os.path.join(root, "motioncheck_ccm_slam_amd")
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lccm_hot", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split() == ["5", "217", "28"], out.stdout + out.stderr


def test_library_links_no_dense_solver_library():
    """The product library carries its own solvers (block Gauss-Jordan, PCG, block-sparse Cholesky): rocSOLVER / rocBLAS, whose
    dpotrf the builder measured to return wrong factors when the GPU is shared, are not even linked."""
    import subprocess
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", out)
    assert needed and not [n for n in needed if "rocsolver" in n or "rocblas" in n], needed
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rocsolver_" not in syms and "rocblas_" not in syms


def test_landmark_cholesky_pivots(tmp_path):
    """ba_math.h's Cholesky of a landmark's 3 x 3 block (operand of the Schur product): exact on a well-conditioned block, finite on
    a rank-deficient one (a landmark with ONE observation) at dampings down to 0 -- ADVICE r2: an unguarded pivot gave NaN there."""
    import re
    exe = str(tmp_path / "chol3_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "support", "chol3_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    r = {k: float(v) for k, v in re.findall(r"(\w+)=([-+.\de]+)", out.stdout)}
    assert r["spd_err"] < 1e-14 and r["rank_deficient_finite"] == 1, out.stdout
