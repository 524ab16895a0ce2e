# Adam launch duration (rocprofv3 kernel trace) against the number of unit workgroups (CASTREC_TG_UNITS)
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-other-configs --no-extra-precisions"
for u in 384 320 256 225 192 160 128; do
  export CASTREC_TG_UNITS=$u
  tools/prof.sh adam_grid_$u kt -- python3 bench.py $Q > gpurun_out/adam_grid_$u.log 2>&1
  echo "units $u: $(grep k_adam gpurun_out/adam_grid_$u.log | tail -1)  $(python3 tools/kt_by_position.py gpurun_out/adam_grid_$u/p_kt | tail -1)"
done
