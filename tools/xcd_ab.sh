#!/bin/bash
# A/B of the XCD-aware work order of k_fast_cells / k_orient_desc: timing and FETCH_SIZE per setting (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for x in "$@"; do
  CCM_ORB_XCD=$x python3 $R/bench.py --no-gba --no-cpu --no-extra > $O/xcd_$x.log 2>&1
  CCM_ORB_XCD=$x rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/xcd_fetch_$x -- python3 $R/bench.py --no-cpu --no-gba --no-extra --steps 3 --warmup 1 > $O/xcd_fetch_$x.log 2>&1
  python3 $R/tools/pmc_table.py $O/xcd_fetch_$x > $O/xcd_fetch_$x.txt
  python3 - <<PY
import json
l=[q for q in open("$O/xcd_$x.log") if q.startswith("{")]
j=json.loads(l[-1]); k=j["kernels"]
f={r.split()[0]: float(r.split()[2]) for r in open("$O/xcd_fetch_$x.txt") if "FETCH_SIZE" in r}
print("XCD=%s value=%.1f ms/step=%.4f fast_cells=%.4f ms (fetch %.0f MiB) orient_desc=%.4f ms (fetch %.0f MiB)" % ("$x", j["value"], j["ms_per_step"], k["k_fast_cells"]["ms_per_step"], f.get("k_fast_cells<true>",0)/1024, k["k_orient_desc"]["ms_per_step"], f.get("k_orient_desc",0)/1024))
PY
done
