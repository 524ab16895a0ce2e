#!/bin/bash
# A/B of one library under two environments on one box, alternating, N rounds:  gpurun -- tools/probes/run_ab_env.sh "VAR=1" [rounds] [bench args]
# (the step differs by ~0.5 % between runs of one binary and by 1-2 % between boxes: compare only inside one call)
envs=$1; n=${2:-3}; shift 2; mkdir -p gpurun_out/abe
for r in $(seq 1 $n); do
  for w in base var; do
    if [ $w = var ]; then pre="env $envs"; else pre=""; fi
    $pre timeout -k 10 200 python bench.py --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions "$@" > gpurun_out/abe/$w$r.json 2> gpurun_out/abe/$w$r.err || { echo fail; tail -3 gpurun_out/abe/$w$r.err; exit 1; }
    python - <<P
import json
d = json.loads(open("gpurun_out/abe/$w$r.json").read().strip().split("\n")[-1])
print("$w round $r:", d["value"], d["ms_per_step"], {k: v["us_per_step"] for k, v in d["kernels"].items()})
P
  done
done
