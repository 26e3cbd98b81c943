#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_single_frame_trace.sh -- kernels and copies of one single-frame ccm_orb_extract call,
# from a rocprofv3 kernel + memory-copy trace (timeline of the last call: start offsets, durations, gaps)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/sf_kt -- python3 $R/tools/bench_single_frame.py > $O/sf_kt.log 2>&1
cd $R
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("gpurun_out/sf_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
for f in glob.glob("gpurun_out/sf_kt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:30]))
ev.sort()
# the last call = events after the last gap > 60 us preceding a k_pyr_resize... take the last 40 events and cut at the last big gap
tail = ev[-60:]
cut = 0
for i in range(1, len(tail)):
    if tail[i][0] - tail[i - 1][1] > 40000: cut = i
tail = tail[cut:]
t0 = tail[0][0]; pe = t0
for s, e, n in tail:
    print("%8.1f us  dur %6.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n)); pe = max(pe, e)
print("span %.1f us" % ((tail[-1][1] - t0) / 1e3))
PY
grep "single frame" $O/sf_kt.log
