#!/bin/bash
# usage: tools/profile_round.sh <tag> <commit>    (run on the GPU box from the repo root; writes gpurun_out/<tag>_*)
# The rocprofv3 evidence of a round: kernel-trace statistics of the default bench command and of the DragonChess
# workload, and the four separate PMC passes (MI355X_MICROARCH.md: counters only, --kernel-trace --pmc) for the dominant
# kernel of each.  <commit> = `git rev-parse --short HEAD` of the tree that was sent (the box has no .git): recorded in the
# summaries, and bench.py labels the per-evaluation figures it takes from them with it.
set -e
TAG=$1
COMMIT=${2:-unknown}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -- python3 $REPO/bench.py --no-cpu-baseline --no-api > $OUT/${TAG}_prof_c2.log 2>&1
echo "c2 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_dc -- python3 $REPO/bench.py --workload dc --no-cpu-baseline > $OUT/${TAG}_prof_dc.log 2>&1
echo "dc stats done"
cd $REPO
# PMC: the timed region as ONE launch (pmc_summary.py takes the longest dispatch), short settle / prefill to keep the passes short
C2ARGS="--steps 8 --warmup 2 --prefill 16 --settle 4 --launches 1 --no-cpu-baseline --no-api"  # (the timed launch must be the longest dispatch: pmc_summary.py picks that one)
bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_c2 $C2ARGS
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_c2 k_selfplay_queue gpurun_out/${TAG}_pmc_c2/summary.json > /dev/null
python3 tools/pmc_derive.py gpurun_out/${TAG}_pmc_c2/summary.json gpurun_out/${TAG}_queue_pmc_summary.json "kernel=k_selfplay_queue<Connect4,8,x3,12>" "command=bench.py $C2ARGS (timed-region dispatch)" "commit=$COMMIT" "passes_dir=gpurun_out/${TAG}_pmc_c2" > /dev/null
DCARGS="--workload dc --steps 4 --warmup 1 --prefill 16 --launches 1 --no-cpu-baseline"
bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_dc $DCARGS
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_dc k_dc_selfplay_fused gpurun_out/${TAG}_pmc_dc/summary.json > /dev/null
python3 tools/pmc_derive.py gpurun_out/${TAG}_pmc_dc/summary.json gpurun_out/${TAG}_dc_pmc_summary.json "kernel=k_dc_selfplay_fused" "command=bench.py $DCARGS (timed-region dispatch)" "commit=$COMMIT" "passes_dir=gpurun_out/${TAG}_pmc_dc" > /dev/null
tail -1 $OUT/${TAG}_prof_c2.log | cut -c1-400
tail -1 $OUT/${TAG}_prof_dc.log | cut -c1-400
find $OUT/${TAG}_prof_c2 $OUT/${TAG}_prof_dc -name "*kernel_stats.csv" | head
