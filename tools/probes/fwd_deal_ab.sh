# A/B of the forward's PAIR-mode tile deal (CASTREC_FWD_DEAL_FIXED: -1 = the formula of rounds 2-4) on top of the block backward's new deal
Q="--no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs"
run() { python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
  CASTREC_FWD_DEAL_FIXED=-1 run old_formula
  run fixed6
  CASTREC_FWD_DEAL_FIXED=2 run fixed2
done
