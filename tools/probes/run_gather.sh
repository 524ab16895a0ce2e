#!/bin/bash
# the bench's gather block (config C5's table, fresh rows per launch): streaming row loads off / on (CASTREC_GATHER_STREAM), twice
for v in 0 1 0 1; do
  CASTREC_GATHER_STREAM=$v timeout -k 10 300 python - <<P
import sys
sys.path.insert(0, ".")
import bench
g = bench.gather_block()
print("stream=$v:", {k: (g[k]["us"], g[k].get("read_frac"), g[k].get("read_write_frac")) for k in ("embed_fwd", "read_only")})
P
done
