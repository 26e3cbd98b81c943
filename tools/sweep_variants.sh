#!/bin/bash
# Build-and-time sweep on the GPU box: tools/sweep_variants.sh <source file> <bench command> <name>=<flags> ...
# Each variant rebuilds the one source with EXTRA=<flags>, runs the bench command and writes gpurun_out/sw_<name>.log.
set -e
src=$1; shift; cmd=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
    name=${v%%=*}; flags=${v#*=}
    touch motioncheck_ccm_slam_amd/csrc/$src
    make -s -C motioncheck_ccm_slam_amd/csrc EXTRA="$flags" > gpurun_out/sw_$name.build 2>&1
    echo "== $name ($flags)" | tee gpurun_out/sw_$name.log
    timeout -k 10 300 $cmd >> gpurun_out/sw_$name.log 2>&1 && tail -2 gpurun_out/sw_$name.log
done
touch motioncheck_ccm_slam_amd/csrc/$src
make -s -C motioncheck_ccm_slam_amd/csrc > /dev/null 2>&1
