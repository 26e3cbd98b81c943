#!/bin/bash
# Run on the GPU box from the repo root: tools/profile_round.sh TAG
# bench line + rocprofv3 kernel trace + the three separate --pmc passes, condensed into profiles/ by summarize_prof.py.
# (rocprofv3 gets the program itself after `--`; counters are collected without any trace option.)
set -e
tag=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_kt -- python3 $R/bench.py --no-cpu --no-extra > $O/${tag}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/bench.py --no-cpu --no-gba --no-extra --steps 3 --warmup 1 > $O/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/bench.py --no-cpu --no-gba --no-extra --steps 3 --warmup 1 > $O/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_sq -- python3 $R/bench.py --no-cpu --no-gba --no-extra --steps 3 --warmup 1 > $O/${tag}_sq.log 2>&1
cd $R
python3 tools/summarize_prof.py $tag $O/${tag}_kt $O/${tag}_fetch $O/${tag}_write $O/${tag}_sq | tee $O/${tag}_summary.txt
python3 tools/pmc_table.py $O/${tag}_sq > $O/${tag}_sq_table.txt
cp profiles/${tag}_kernel_stats.csv profiles/pmc_traffic.json $O/
# the bench line last: its roofline.traffic / valu_issue_frac come from the pmc_traffic.json just written
cd /tmp && python3 $R/bench.py > $O/${tag}_bench.log 2>&1; cd $R
grep '^{' $O/${tag}_bench.log | tail -1
