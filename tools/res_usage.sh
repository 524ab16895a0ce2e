#!/bin/bash
# VGPRs / spills / scratch of every kernel of one source of csrc/ (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
#   tools/res_usage.sh cr_stack_bwd1.hip [extra flags]
src=$1; shift
cd "$(dirname "$0")/../context-aware-sequential-recommendation_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include -I . -fno-slp-vectorize -Wno-unused-function --cuda-device-only -c -x hip $src -o /dev/null \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
name = None
for l in sys.stdin:
    m = re.search(r"remark: +(Function Name|VGPRs|VGPR Spill|ScratchSize \[bytes/lane\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", l)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name": name = v; row = {}
    else: row[k] = v
    if k.startswith("LDS"): print("%-60s VGPRs %s spill %s scratch %s SGPRs %s" % (name, row.get("VGPRs"), row.get("VGPR Spill"), row.get("ScratchSize [bytes/lane]"), row.get("SGPRs")))
'
