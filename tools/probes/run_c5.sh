#!/bin/bash
# full-size C5 (10 M-item table, dense TF-Adam): the default build and the variants given as arguments (CASTREC_LIB), Adam time per step
mkdir -p gpurun_out/c5
for v in default "$@"; do
  if [ $v = default ]; then unset CASTREC_LIB; else export CASTREC_LIB=$PWD/$v; fi
  timeout -k 10 400 python bench.py --model sasrec --maxlen 512 --hidden_units 256 --num_heads 4 --num_blocks 2 --corpus c5 --steps 12 --warmup 3 --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions > gpurun_out/c5/run.json 2> gpurun_out/c5/run.err || { echo fail; tail -3 gpurun_out/c5/run.err; exit 1; }
  python - <<P
import json
d = json.loads(open("gpurun_out/c5/run.json").read().strip().split("\n")[-1])
print("$v:", d["value"], d["ms_per_step"], {k: v["us_per_step"] for k, v in d["kernels"].items() if "adam" in k})
P
done
