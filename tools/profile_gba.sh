#!/bin/bash
# Run on the GPU box from the repo root: tools/profile_gba.sh TAG
# kernel trace + separate --pmc passes of the config-5 global BA alone (tools/bench_gba.py).
set -e
tag=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_gba_kt -- python3 $R/tools/bench_gba.py --iters 20 > $O/${tag}_gba_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_gba_fetch -- python3 $R/tools/bench_gba.py --iters 2 > $O/${tag}_gba_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_gba_write -- python3 $R/tools/bench_gba.py --iters 2 > $O/${tag}_gba_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_gba_sq -- python3 $R/tools/bench_gba.py --iters 2 > $O/${tag}_gba_sq.log 2>&1
cd $R
python3 tools/pmc_table.py $O/${tag}_gba_fetch $O/${tag}_gba_write $O/${tag}_gba_sq > $O/${tag}_gba_pmc_table.txt
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
cat $O/${tag}_gba_kt.log | grep '^{'
