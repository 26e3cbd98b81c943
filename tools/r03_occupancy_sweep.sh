#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_occupancy_sweep.sh TAG
# How close to saturation do k_fast_cells and k_orient_desc run at 8 workgroups per CU?  A larger LDS request per workgroup leaves
# k = 8, 6, 5, 4, 3, 2 workgroups per CU; a kernel that waits on latency slows down like 8 / k, one that is short of issue slots
# (or of any other shared pipe) hardly at all.
tag=$1; O=gpurun_out; mkdir -p $O
for k in 8 6 5 4 3 2; do
  lds=$(( 163840 / k - 256 )); [ $k = 8 ] && lds=0
  CCM_FC_LDS_MIN=$lds CCM_OD_LDS_PAD=$(( lds > 16384 ? lds - 16384 : 0 )) timeout -k 10 200 python3 bench.py --no-cpu --no-gba --no-extra > $O/${tag}_occ_$k.log 2>&1 || exit 1
  python3 - $k $O/${tag}_occ_$k.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("workgroups/CU %s  k_fast_cells %.4f ms  k_orient_desc %.4f ms  step %.4f ms" % (sys.argv[1], d["kernels"]["k_fast_cells"]["ms_per_step"], d["kernels"]["k_orient_desc"]["ms_per_step"], d["ms_per_step"]))
PY
done
