#!/bin/bash
cd /root/repo
for v in 1 0; do
  touch context-aware-sequential-recommendation_amd/csrc/cr_wide.hip
  CASTREC_EXTRA_FLAGS="-DWD_NT8=$v" python -m castrec_amd.build > /dev/null 2>&1 || echo build failed
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/wnt$v -o w --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-gather --no-extra-precisions --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4 --steps 40 --warmup 5 > /root/repo/gpurun_out/wnt$v.log 2>&1)
  echo "NT8=$v: $(grep -o '"value": [0-9.]*' /root/repo/gpurun_out/wnt$v.log | head -1)"
  grep "k_wide\|k_adam\|wgrad" /root/repo/gpurun_out/wnt$v/w_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
done
touch context-aware-sequential-recommendation_amd/csrc/cr_wide.hip
