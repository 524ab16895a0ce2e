#!/bin/bash
# per-kernel table of the C4 shape (or "$@") through rocprofv3
cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/wprof -o w --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4 --steps 40 --warmup 5 "$@" > /root/repo/gpurun_out/wprof.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' /root/repo/gpurun_out/wprof.log | head -2
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/wprof/w_kernel_stats.csv')))
for r in rows[:16]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f}us pct {float(r['Percentage']):5.1f}")
PY
