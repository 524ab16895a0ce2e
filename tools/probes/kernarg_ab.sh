Q="--no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs"
for r in 1 2; do
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v', d['ms_per_step'], d['value'])"
done
python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unset', d['ms_per_step'], d['value'])"
done
