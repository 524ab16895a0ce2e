#!/bin/bash
# A/B of two builds at the C5 shape (sasrec, T 512, D 256, 2 blocks, 4 heads, batch 32, ml-1m-sized table)
var=$1; n=${2:-3}; mkdir -p gpurun_out/ab
for r in $(seq 1 $n); do
  for w in base var; do
    if [ $w = var ]; then export CASTREC_LIB=$PWD/$var; else unset CASTREC_LIB; fi
    timeout -k 10 200 python bench.py --model sasrec --maxlen 512 --hidden_units 256 --num_blocks 2 --num_heads 4 --batch_size 32 --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions > gpurun_out/ab/c5$w$r.json 2> gpurun_out/ab/c5$w$r.err || { echo fail; tail -3 gpurun_out/ab/c5$w$r.err; exit 1; }
    python - <<P
import json
d = json.loads(open("gpurun_out/ab/c5$w$r.json").read().strip().split("\n")[-1])
print("$w round $r:", d["value"], d["ms_per_step"])
P
  done
done
