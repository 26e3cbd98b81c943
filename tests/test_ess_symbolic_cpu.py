"""Host logic of the essential graph's block-sparse Cholesky (csrc/ess_symbolic.h), checked without a GPU.

tests/support/ess_symbolic_check.cpp includes the product's symbolic phase (ordering by rounds of independent low-degree
eliminations, fill, assembly / product / row lists) and replays the index walks of the k_essp_* kernels on the CPU with random
SPD blocks: the solution must equal a dense Cholesky solve (small graphs) or leave a 1e-9 residual (2000 keyframes), the
permutation and the lists must be well formed, and a regular band graph (every keyframe linked to its 4 predecessors, the
shape of an essential graph's covisibility edges) must need tens of rounds, not one per keyframe."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ess") / "ess_symbolic_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "support", "ess_symbolic_check.cpp"), "-o", exe])
    return exe


def _run(exe, *args):
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return {k: float(v) for k, v in re.findall(r"(\w+)=([-+.\de]+)", out.stdout)}


@pytest.mark.parametrize("n,covis,loops,seed", [(50, 2, 2, 1), (200, 3, 5, 2), (300, 6, 20, 3)])
def test_replayed_factorisation_equals_dense_cholesky(harness, n, covis, loops, seed):
    r = _run(harness, n, covis, loops, seed)
    assert r["rel_err"] < 1e-11 and r["structure_ok"] == 1 and r["bad"] == 0
    assert r["factor_blocks"] >= n - 1 + r["edges"] - loops - 2          # at least the pattern itself (duplicates aside)


def test_band_graph_of_2000_keyframes_needs_few_rounds(harness):
    r = _run(harness, 2000, 4, 1, 2, 1, 1)                                 # regular band: 7992 edges
    assert r["rel_err"] < 1e-9 and r["structure_ok"] == 1
    assert r["rounds"] < 100 and r["factor_blocks"] < 4 * (1999 + r["edges"])
    strict = _run(harness, 2000, 4, 1, 2, 0, 1)                            # near-minimum-degree candidates only: peels the chain from its ends
    assert strict["rounds"] > 900
