"""bench.py's contract on the GPU box: one JSON line with the required keys at N=1, and the N=2 flow (torchrun, ranks,
max-over-ranks timing, sharded global BA) rehearsed on one GPU with the gloo control plane and the shared-memory
all-reduce in RCCL's place."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"}


def _line(out):
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--gba-iters", "2", "--no-cpu"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _line(out.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 1.0 and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and "workload" in d["config"]
    assert d["gba"]["iterations"] == 2 and d["gba"]["chi2_final"] < d["gba"]["chi2_initial"]


@pytest.fixture(scope="module")
def gba_single():
    """config 5, two LM iterations, unsharded: what the sharded runs below must reproduce"""
    out = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--gba-iters", "2", "--no-cpu", "--no-extra"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return _line(out.stdout)["gba"]


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_rehearsal(ranks, gba_single):
    """bench.py under torch.distributed.run with 2 and 4 ranks sharing the one GPU (gloo control plane, shared-memory
    all-reduce in RCCL's place): the N > 1 flow, the landmark shards and the rank-0 increment with several silent ranks."""
    env = dict(os.environ, CCM_BENCH_BACKEND="gloo", CCM_BENCH_COMM="shm")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                          "--master-port", str(29541 + ranks), "bench.py", "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--gba-iters", "2"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    d = _line(out.stdout)
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["frames_per_gpu"] == 256 and d["value"] > 1.0
    g = d["gba"]
    assert "error" not in g and g["n_gpus"] == ranks and g["iterations"] == 2 and g["chi2_final"] < g["chi2_initial"]
    assert g["schur_pairs_this_rank"] < 1.2 / ranks * 9190009            # each rank enumerates about its share of the pairs
    # the sharded solve at FULL size against the unsharded one: the landmark partition only changes the order of the sums
    assert g["trials"] == gba_single["trials"] and g["iterations"] == gba_single["iterations"] and g["schur_blocks"] == gba_single["schur_blocks"]
    assert abs(g["chi2_initial"] - gba_single["chi2_initial"]) <= 1e-9 * gba_single["chi2_initial"]
    assert abs(g["chi2_final"] - gba_single["chi2_final"]) <= 1e-9 * gba_single["chi2_final"], (g["chi2_final"], gba_single["chi2_final"])
