#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_gba_trace.sh TAG  -- kernel trace of config 5, optimize(20), one timed call
tag=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_gba_kt -- python3 $R/tools/bench_gba.py --iters 20 > $O/${tag}_gba_kt.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/${tag}_gba_kt/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Name"].split("(")[0].replace("void ", "")[:60], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
rows.sort(key=lambda r: -r[2])
with open("$O/${tag}_gba_kernel_stats.csv", "w") as o:
    o.write("kernel,calls,total_ns,avg_ns,percent\n")
    for r in rows: o.write("%s,%d,%.0f,%.1f,%.2f\n" % r)
PY
head -30 $O/${tag}_gba_kernel_stats.csv
