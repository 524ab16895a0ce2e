# A/B of attention-pass tile dealings of the block backward (CASTREC_B1_QPK / CASTREC_B1_KPK), headline workload, interleaved rounds
Q="--no-cpu-baseline --no-gather --no-extra-precisions --no-other-configs"
run() { python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
  run base
  CASTREC_B1_KPK="0,1,2,6:12,3:11,4:9,5:7,8:10" CASTREC_B1_QPK="12,10,11,7:1,8:0,9:2,6:4,5:3" run c3
  CASTREC_B1_KPK="0,1,4,6:9,3:11,2:12,5:7,8:10" CASTREC_B1_QPK="12,10,9,7:2,8:0,11:1,6:4,5:3" run c2
  CASTREC_B1_KPK="0,2,4:12,6:10,1,3:11,5:9,7:8" CASTREC_B1_QPK="12,11,9:1,7:3,10,8:0,6:2,4:5" run c1
  CASTREC_B1_KPK="0,1,2,8:10,3:11,4:9,5:7,6:12" CASTREC_B1_QPK="12,10,11,5:3,8:0,9:2,6:4,7:1" run c3_v2
done
