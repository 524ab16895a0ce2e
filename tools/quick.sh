#!/bin/bash
# one GPU call's worth of checking after a kernel change: the model parity tests, then the headline bench line (value, ms, launch us)
#   gpurun -- tools/quick.sh <outdir under gpurun_out> [pytest -k expression]
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q ${2:+-k "$2"} > $out/tests.log 2>&1; rc=$?
tail -2 $out/tests.log | cut -c1-200
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/tests.log | head -20; exit $rc; }
timeout -k 10 200 python bench.py --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<P
import json
d = json.loads(open("$out/bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d["roofline"]["frac"])
print([(k["name"], k["us"]) for k in d.get("kernels", [])][:12] if isinstance(d.get("kernels"), list) else d.get("kernels"))
P
