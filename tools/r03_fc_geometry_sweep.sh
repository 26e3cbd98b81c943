#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_fc_geometry_sweep.sh TAG
# k_fast_cells is latency-bound at the hardware's cap of 32 waves per CU (tools/r03_occupancy_sweep.sh): fewer waves per band and
# smaller bands put more bands on a CU.  Variants: threads per workgroup / band pitch cap / survivor-list size / list sizes.
tag=$1; O=gpurun_out; mkdir -p $O
run() {  # name, build flags, pcap, surv
  touch motioncheck_ccm_slam_amd/csrc/orb_kernels.hip
  make -s -C motioncheck_ccm_slam_amd/csrc EXTRA="$2" > $O/${tag}_fcg_$1.build 2>&1 || { echo "$1: build failed"; return; }
  CCM_FC_PCAP=$3 CCM_FC_SURV=$4 timeout -k 10 200 python3 bench.py --no-cpu --no-gba --no-extra > $O/${tag}_fcg_$1.log 2>&1 || { echo "$1: run failed"; tail -3 $O/${tag}_fcg_$1.log; return; }
  python3 - $1 $O/${tag}_fcg_$1.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("%-28s k_fast_cells %.4f ms  step %.4f ms  value %.1f" % (sys.argv[1], d["kernels"]["k_fast_cells"]["ms_per_step"], d["ms_per_step"], d["value"]))
PY
}
run base_256_144      ""                                           144 3072
run t128_p144         "-DFC_TPB=128"                               144 3072
run t128_p112         "-DFC_TPB=128 -DFC_NZ=768 -DFC_KEPT=360"     112 2304
run t128_p80          "-DFC_TPB=128 -DFC_NZ=512 -DFC_KEPT=240"     80 1536
run t128_p80_s3072    "-DFC_TPB=128 -DFC_NZ=512 -DFC_KEPT=240"     80 3072
run t64_p80           "-DFC_TPB=64 -DFC_NZ=512 -DFC_KEPT=240"      80 1536
run t64_p48           "-DFC_TPB=64 -DFC_NZ=256 -DFC_KEPT=120"      48 1024
run t192_p112         "-DFC_TPB=192 -DFC_NZ=768 -DFC_KEPT=360"     112 2304
touch motioncheck_ccm_slam_amd/csrc/orb_kernels.hip; make -s -C motioncheck_ccm_slam_amd/csrc > /dev/null 2>&1
