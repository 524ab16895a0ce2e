#!/bin/bash
for w in 16 32 64 128; do
 echo "wslabs $w: $(CASTREC_WSLABS=$w python bench.py --no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs --model sasrec --maxlen 512 --hidden_units 256 --num_heads 4 --num_blocks 2 --batch_size 32 --steps 100 --warmup 10 2>/dev/null | tail -n 1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
