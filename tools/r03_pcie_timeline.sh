#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_pcie_timeline.sh TAG
# Timeline of the drop-in call with host buffers (ccm_orb_extract, 256 frames in 4 chunks, page-locked frames and result pools):
# rocprofv3 --kernel-trace --memory-copy-trace of tools/bench_pcie.py, then which copies ran under which kernels.
tag=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/${tag}_pcie_tl -- python3 $R/tools/bench_pcie.py --calls 2 > $O/${tag}_pcie_tl.log 2>&1
cd $R
python3 - > $O/${tag}_pcie_timeline.txt <<PY
import csv, glob
kt = glob.glob("$O/${tag}_pcie_tl/**/*kernel_trace.csv", recursive=True)
mt = glob.glob("$O/${tag}_pcie_tl/**/*memory_copy_trace.csv", recursive=True)
K = [r for f in kt for r in csv.DictReader(open(f))]
M = [r for f in mt for r in csv.DictReader(open(f))]
ev = []
for r in K:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]))
for r in M:
    d = r.get("Direction", r.get("Kind", "copy"))
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", d.replace("MEMORY_COPY_", "")))
ev.sort()
# the last batch call with page-locked pools: its four chunk uploads are the last four long host-to-device copies; it ends with the
# last long device-to-host copy that follows its last k_orient_desc<8> launch (the single-frame calls after it copy a few KB)
h2d = [i for i, e in enumerate(ev) if e[2] == "C" and "HOST_TO_DEVICE" in e[3] and e[1] - e[0] > 100000]
lo = h2d[-4]
od = [i for i, e in enumerate(ev) if e[2] == "K" and e[3].startswith("k_orient_desc<8")]
hi = od[-1]
for i in range(od[-1], len(ev)):
    if ev[i][0] - ev[od[-1]][1] > 400000: break
    if ev[i][2] == "C" and "DEVICE_TO_HOST" in ev[i][3] and ev[i][1] - ev[i][0] > 20000: hi = i
seg = ev[lo:hi + 1]
t0 = seg[0][0]
print("ccm_orb_extract, 256 frames, page-locked frames and result pools: events of the last call (us from its first upload); kernels of one chunk collapsed")
def merge(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
big = [e for e in seg if e[2] == "C" and e[1] - e[0] > 20000]
for e in big:
    print("  %-16s %8.1f .. %8.1f  (%.1f us)" % (e[3], (e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3))
kern = merge([(e[0], e[1]) for e in seg if e[2] == "K"])
# chunk = kernels between two resize chains: group kernel intervals separated by > 30 us
groups = []
for a, b in kern:
    if groups and a - groups[-1][1] < 30000: groups[-1][1] = b
    else: groups.append([a, b])
for a, b in groups:
    print("  kernels          %8.1f .. %8.1f  (%.1f us)" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
copies = merge([(e[0], e[1]) for e in big])
def overlap(A, B):
    t = 0
    for a, b in A:
        for c, d in B:
            t += max(0, min(b, d) - max(a, c))
    return t
tot = seg[-1][1] - t0
ck = sum(b - a for a, b in kern); cc = sum(b - a for a, b in copies)
print("call span %.1f us; kernels busy %.1f us; copy engines busy %.1f us; copies under kernels %.1f us (%.0f %% of the kernel time)" %
      (tot / 1e3, ck / 1e3, cc / 1e3, overlap(kern, copies) / 1e3, 100.0 * overlap(kern, copies) / max(ck, 1)))
PY
cat $O/${tag}_pcie_timeline.txt
grep '^{' $O/${tag}_pcie_tl.log | tail -1
