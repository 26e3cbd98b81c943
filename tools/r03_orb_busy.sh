#!/bin/bash
# Run on the GPU box from the repo root: tools/r03_orb_busy.sh TAG
# Direct pipe-busy counters of the extraction kernels (one --pmc pass each, no trace options): how much of a SIMD's time goes to
# vector issue, LDS, scalar, waiting -- the number the SQ_INSTS_VALU x cost bracket could not give.
tag=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/${tag}_counters_list.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_busy1 -- python3 $R/bench.py --no-cpu --no-gba --no-extra --steps 3 --warmup 1 > $O/${tag}_busy1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_busy2 -- python3 $R/bench.py --no-cpu --no-gba --no-extra --steps 3 --warmup 1 > $O/${tag}_busy2.log 2>&1 || exit 1
cd $R
python3 tools/pmc_table.py $O/${tag}_busy1 $O/${tag}_busy2 > $O/${tag}_busy_table.txt
rm -rf $O/${tag}_busy1 $O/${tag}_busy2
cat $O/${tag}_busy_table.txt
