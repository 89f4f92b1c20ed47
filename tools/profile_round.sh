#!/bin/bash
# usage: tools/profile_round.sh <tag>    (run on the GPU box from the repo root; writes gpurun_out/<tag>_*)
# The rocprofv3 evidence of a round: kernel-trace statistics of the default bench command and of the DragonChess
# workload, and the four separate PMC passes (MI355X_MICROARCH.md) for the dominant kernel of each.
set -e
TAG=$1
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/${TAG}_prof_c2.log 2>&1
echo "c2 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_dc -- python3 $REPO/bench.py --workload dc --no-cpu-baseline > $OUT/${TAG}_prof_dc.log 2>&1
echo "dc stats done"
cd $REPO
bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_c2 --steps 4 --warmup 2 --prefill 16 --no-cpu-baseline
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_c2 k_selfplay_queue gpurun_out/${TAG}_pmc_c2/summary.json > /dev/null
python3 tools/pmc_derive.py gpurun_out/${TAG}_pmc_c2/summary.json gpurun_out/${TAG}_queue_pmc_summary.json "kernel=k_selfplay_queue<Connect4,8,x3,12>" "command=bench.py --steps 4 --warmup 2 --prefill 16 --no-cpu-baseline (timed-region dispatch)" > /dev/null
bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_dc --workload dc --steps 4 --warmup 1 --prefill 16 --no-cpu-baseline
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_dc k_dc_selfplay_fused gpurun_out/${TAG}_pmc_dc/summary.json > /dev/null
python3 tools/pmc_derive.py gpurun_out/${TAG}_pmc_dc/summary.json gpurun_out/${TAG}_dc_pmc_summary.json "kernel=k_dc_selfplay_fused" "command=bench.py --workload dc --steps 4 --warmup 1 --prefill 16 --no-cpu-baseline (timed-region dispatch)" > /dev/null
tail -1 $OUT/${TAG}_prof_c2.log | cut -c1-400
tail -1 $OUT/${TAG}_prof_dc.log | cut -c1-400
find $OUT/${TAG}_prof_c2 $OUT/${TAG}_prof_dc -name "*kernel_stats.csv" | head
