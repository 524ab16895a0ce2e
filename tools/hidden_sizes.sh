#!/bin/bash
# per-launch times of the headline step at other --hidden_units (the FAMILY instantiations of cr_stack_fwd / cr_stack_block_bwd,
# cr_rlayout.hpp d_ctx<NF>) beside D = 50:  gpurun -- tools/hidden_sizes.sh [sizes...]
mkdir -p gpurun_out/hs
for D in ${@:-50 20 36 44 60 33 47}; do
  timeout -k 10 200 python bench.py --hidden_units $D --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions > gpurun_out/hs/d$D.json 2> gpurun_out/hs/d$D.err || { echo "D=$D failed"; tail -3 gpurun_out/hs/d$D.err; continue; }
  python - <<P
import json
d = json.loads(open("gpurun_out/hs/d$D.json").read().strip().split("\n")[-1])
print("D=$D:", d["value"], d["ms_per_step"], {k: round(v["us_per_step"] / v["launches"], 1) for k, v in d["kernels"].items()})
P
done
