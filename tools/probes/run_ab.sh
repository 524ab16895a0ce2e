#!/bin/bash
# A/B of two builds of the library on one box: the default and CASTREC_LIB=<variant>, alternating, N rounds (the step differs by ~0.5 %
# between runs of one binary: a single pair does not separate a 1 % effect)
#   gpurun -- tools/probes/run_ab.sh <variant .so> [rounds]
var=$1; n=${2:-3}; mkdir -p gpurun_out/ab
for r in $(seq 1 $n); do
  for w in base var; do
    if [ $w = var ]; then export CASTREC_LIB=$PWD/$var; else unset CASTREC_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions > gpurun_out/ab/$w$r.json 2> gpurun_out/ab/$w$r.err || { echo fail; tail -3 gpurun_out/ab/$w$r.err; exit 1; }
    python - <<P
import json
d = json.loads(open("gpurun_out/ab/$w$r.json").read().strip().split("\n")[-1])
print("$w round $r:", d["value"], d["ms_per_step"], {k: v["us_per_step"] for k, v in d["kernels"].items()})
P
  done
done
