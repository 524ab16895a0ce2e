#!/bin/bash
# The other BASELINE shapes through the same bench.py (not the headline line): one JSON value per shape.
run() { out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs "$@" --steps 200 --warmup 20 2>/dev/null | tail -n 1); echo "$* -> $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "seq/s", d["ms_per_step"], "ms", d["config"]["launches_per_step"], "launches")')"; }
run --model sasrec --maxlen 50
run --model sasrec --maxlen 200
run --model sasrec --maxlen 50 --hidden_units 64 --num_heads 2 --dropout_rate 0.5
run --model cast_4
run --model cast_9
run --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4
run --model sasrec --maxlen 512 --hidden_units 256 --num_heads 4 --num_blocks 2 --batch_size 32
run --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4 --attn_precision f32
run --model sasrec --maxlen 512 --hidden_units 256 --num_heads 4 --num_blocks 2 --batch_size 32 --attn_precision f32
