# A/B inside the step on one box: each "NAME:ENV=VAL,ENV=VAL" of $ARMS runs bench.py with that environment; repeated $REP times.
mkdir -p gpurun_out
b() { name=$1; shift; timeout 300 env "$@" python bench.py --steps ${STEPS:-30} --warmup 5 --no_cpu_baseline --no_next_rows $CFG > gpurun_out/ab_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"frac": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*|"isolated_[a-z]*_ms": [0-9.]*' gpurun_out/ab_$name.log | head -7 | tr '\n' ' ')"; }
for rep in $(seq 1 ${REP:-2}); do
  for arm in $ARMS; do
    name=${arm%%:*}; envs=${arm#*:}; [ "$envs" = "$arm" ] && envs="PD_DUMMY=1"
    b ${name}_$rep $(echo $envs | tr ',' ' ')
  done
done
