"""The CPU oracle under AddressSanitizer + UBSan (sanitizers run on the CPU build only; the GPU pool has none)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_extract_and_match_are_clean_under_asan(tmp_path):
    exe = str(tmp_path / "oracle_asan")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("orb_oracle.c", "match_oracle.c", "ba_oracle.c", "bchol_oracle.c", "bow_oracle.c")]
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
           os.path.join(ROOT, "tests", "support", "oracle_asan_main.c")] + srcs + ["-lm", "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout[-1500:] + out.stderr[-3000:]
