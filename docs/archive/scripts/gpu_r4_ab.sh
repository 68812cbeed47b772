# Round 4 A/B inside the step on one box: each "NAME:ENV=VAL,ENV=VAL" of $ARMS runs bench.py with that environment; $REP times.
mkdir -p gpurun_out/r4
b() { name=$1; shift; timeout 300 env "$@" python bench.py --steps ${STEPS:-50} --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step $CFG > gpurun_out/r4/ab_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*|"isolated_[a-z]*_ms": [0-9.]*' gpurun_out/r4/ab_$name.log | head -6 | tr '\n' ' ')"; tail -3 gpurun_out/r4/ab_$name.log | grep -iE "error|Traceback"; }
for rep in $(seq 1 ${REP:-2}); do
  for arm in $ARMS; do
    name=${arm%%:*}; envs=${arm#*:}; [ "$envs" = "$arm" ] && envs="PD_DUMMY=1"
    b ${name}_$rep $(echo $envs | tr ',' ' ')
  done
done
