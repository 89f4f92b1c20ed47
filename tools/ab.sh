#!/bin/bash
# usage: tools/ab.sh <libA.so|-> <libB.so|-> [rounds] [bench args]   -- alternates two builds on one box ("-" = the in-tree build),
# one untimed heat-up run first (the chip's clock settles under load), prints evaluations/s of every run and the medians
A=$1; B=$2; N=${3:-4}; shift 3
run() { # label lib [bench args]
  L=$1; LIB=$2; shift 2
  if [ "$LIB" = "-" ]; then P=bench.py; unset BB_LIB; else P=tools/bench_lib.py; export BB_LIB=$LIB; fi
  timeout -k 10 200 python $P --steps 32 --warmup 8 --no-cpu-baseline --no-api "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$L.json || { echo "$L FAILED"; exit 1; }
}
run heat "$A" "$@"
for i in $(seq 1 $N); do run A$i "$A" "$@" && run B$i "$B" "$@" || exit 1; done
python - <<PY
import json, statistics
for k, lib in (("A", "$A"), ("B", "$B")):
    v = [json.load(open("gpurun_out/ab_%s%d.json" % (k, i)))["net_evals_per_sec_rank0"] / 1e6 for i in range(1, $N + 1)]
    print(k, lib, "evals/s (M):", " ".join("%.1f" % x for x in v), "median %.1f" % statistics.median(v))
PY
