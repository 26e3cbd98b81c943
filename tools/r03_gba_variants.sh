#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_gba_variants.sh TAG
# config 5, optimize(20): the classic PCG iteration, the pipelined one with 4 and with 2 entries per thread and step.
tag=$1; O=gpurun_out; mkdir -p $O
CCM_PCG_PIPELINED=0 timeout -k 10 300 python3 tools/bench_gba.py --iters 20 --calls 3 > $O/${tag}_gba_classic.log 2>&1 &&
CCM_PPCG_EPS=4 timeout -k 10 300 python3 tools/bench_gba.py --iters 20 --calls 3 > $O/${tag}_gba_pipe4.log 2>&1 &&
CCM_PPCG_EPS=2 timeout -k 10 300 python3 tools/bench_gba.py --iters 20 --calls 3 > $O/${tag}_gba_pipe2.log 2>&1 &&
CCM_DEBUG=1 timeout -k 10 300 python3 tools/bench_gba.py --iters 20 --calls 1 > $O/${tag}_gba_debug.log 2>&1
grep -h '^{' $O/${tag}_gba_classic.log $O/${tag}_gba_pipe4.log $O/${tag}_gba_pipe2.log
