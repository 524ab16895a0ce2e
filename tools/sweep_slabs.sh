#!/bin/bash
for ns in 160 200 224 256 320; do
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --n_slabs $ns --steps 200 --warmup 20 2>/dev/null | tail -n 1)
  echo "n_slabs $ns $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
