#!/bin/bash
# how often does the D = 128 two-rank comparison fail: N runs of the one test, failures counted
n=${1:-6}; f=0
for i in $(seq $n); do
  out=$(timeout -k 10 180 python -m pytest tests/test_dist_gpu.py -q -k "on_the_card" 2>&1 | grep -E "AssertionError: \(|passed|failed")
  echo "$out" | tr '\n' ' '; echo
  echo "$out" | grep -q failed && f=$((f+1))
done
echo "failures: $f of $n"
