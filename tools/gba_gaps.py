#!/usr/bin/env python3
"""Idle time between consecutive kernels of the timed global-BA call, from a rocprofv3 kernel trace.
usage: python3 tools/gba_gaps.py <dir with *_kernel_trace.csv> [min_gap_us]"""
import csv, glob, sys, collections
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed call = the kernels after the largest idle gap following the warm-up call (setup kernels k_sp_pair_count mark a call's start)
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_sp_pair_count")]
i0 = starts[-1] if starts else 0
rows = rows[i0:]
busy = sum(e - s for s, e, _ in rows) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
print("kernels %d  span %.1f us  busy %.1f us  idle %.1f us" % (len(rows), span, busy, span - busy))
gaps = collections.defaultdict(lambda: [0, 0.0])
end = rows[0][1]
for (s, e, n), (ps, pe, pn) in zip(rows[1:], rows[:-1]):
    g = (s - end) / 1e3
    if g > thr:
        k = pn + " -> " + n
        gaps[k][0] += 1; gaps[k][1] += g
    end = max(end, e)
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%7.1f us  x%-3d  %s" % (t, c, k))
small = 0.0; end = rows[0][1]
for (s, e, n) in rows[1:]:
    g = (s - end) / 1e3
    if 0 < g <= thr: small += g
    end = max(end, e)
print("gaps <= %.0f us sum: %.1f us" % (thr, small))
