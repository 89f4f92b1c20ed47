#!/bin/bash
# usage: tools/qb.sh <label> [bench args]   -- short bench, prints games/s (env passes through; BB_LIB picks another build)
L=$1; shift
if [ -n "$BB_LIB" ]; then P=tools/bench_lib.py; else P=bench.py; fi
timeout -k 10 200 python $P --steps 16 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/qb_$L.log 2>&1 || { echo "$L FAILED"; tail -3 gpurun_out/qb_$L.log; exit 1; }
python - <<PY
import json
j=json.loads(open("gpurun_out/qb_$L.log").read().strip().splitlines()[-1])
print("$L", round(j["value"],1), "games/s; sims/s %.1fM evals/s %.1fM overflow %d" % (j["node_evals_per_sec"]/1e6, j["net_evals_per_sec_rank0"]/1e6, j["overflow"]))
PY
