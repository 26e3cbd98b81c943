#!/usr/bin/env python3
"""Average every PMC counter per kernel launch from rocprofv3 --pmc output directories.
usage: pmc_table.py DIR [DIR ...]  -> prints kernel, counter, mean per launch, launches"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    if k.startswith("k_"):
        print("%-42s %-28s %16.1f  n=%d" % (k, c, s / n, n))
