#!/bin/bash
# quick headline bench on the GPU box: tools/qbench.sh <tag> [extra bench flags]  -> gpurun_out/<tag>.json + per-kernel table
tag=$1; shift
mkdir -p gpurun_out/$(dirname $tag)
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -5 gpurun_out/$tag.err; exit 1; }
python3 - gpurun_out/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("seq/s", d["value"], "ms/step", d["ms_per_step"], "launches", d["config"]["launches_per_step"])
for k, v in d["kernels"].items():
    print("  %-28s %7.1f us/step  %d launches" % (k, v["us_per_step"], v["launches"]))
PY
