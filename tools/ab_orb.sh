#!/bin/bash
# Run on the GPU box from the repo root: tools/ab_orb.sh ROUNDS ab/libA.so ab/libB.so ...
# A/B timing of builds of libccm_hot.so on ONE box (box-to-box spread of k_fast_cells is ~3 %, the effects looked for are smaller):
# the front-end leg of bench.py per library, the libraries interleaved, ROUNDS times.
rounds=$1; shift
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for lib in "$@"; do
    CCM_HOT_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --no-cpu --no-gba --no-extra 2> gpurun_out/ab_err.log | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$lib round $r: step %.4f ms  value %.1f  %s %.4f ms' % (d['ms_per_step'], d['value'], r['kernel'], r['algorithmic_bytes_per_launch'] / r['achieved'] / 1e6))
" || { tail -5 gpurun_out/ab_err.log; exit 1; }
  done
done
