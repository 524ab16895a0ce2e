#!/bin/bash
# time k_wide_qkv_fwd in debug variants (WD_DBG): 0 normal, 1 prologue only, 2 no MFMA, 3 no weight traffic, 4 no result stores
cd /root/repo
for v in 0 1 2 3 4; do
  touch context-aware-sequential-recommendation_amd/csrc/cr_wide.hip
  CASTREC_EXTRA_FLAGS="-DWD_DBG=$v" python -m castrec_amd.build > /dev/null 2>&1 || echo build failed
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/wdbg$v -o w --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-gather --no-extra-precisions --model sasrec --maxlen 200 --hidden_units 128 --num_heads 4 --num_blocks 4 --steps 40 --warmup 5 > /dev/null 2>&1)
  echo "variant $v: $(grep k_wide_qkv_fwd /root/repo/gpurun_out/wdbg$v/w_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120)"
done
touch context-aware-sequential-recommendation_amd/csrc/cr_wide.hip
