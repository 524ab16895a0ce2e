#!/bin/bash
# C4 / C5 shapes through bench.py (not the headline line): value, ms/step, launches
run() { out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs "$@" --steps 100 --warmup 10 2>/dev/null | tail -n 1); echo "$* -> $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "seq/s", d["ms_per_step"], "ms", d["config"]["launches_per_step"], "launches")')"; }
run --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4 --corpus books
run --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4
run --model sasrec --maxlen 512 --hidden_units 256 --num_heads 4 --num_blocks 2 --batch_size 32
