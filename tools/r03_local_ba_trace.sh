#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_local_ba_trace.sh -- kernels of the local BA (config 4) per call, from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lba_kt -- python3 $R/tools/bench_local_ba.py > $O/lba_kt.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/lba_kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = 61
print("kernel time per call: %.3f ms" % (tot / calls / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print("%-44s %6.1f per call  %8.1f us avg  %6.1f us per call" % (r["Name"].split("(")[0].replace("void ", "")[:44], int(r["Calls"]) / calls, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / calls / 1e3))
PY
grep "local BA" $O/lba_kt.log
