# Round 5: FETCH_SIZE of the headline kernels for library variants / row modes (one rocprofv3 --pmc pass each, kernel-trace only).
#   ARMS="name:PD_LIB-or-empty:PD_SWEEP_IMPL ..."
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r5/fetch_ab; rm -rf $OUT; mkdir -p $OUT
ARMS=${ARMS:-product::0 noregroup:planedepth_amd/lib/libpd_var_noregroup.so:0 fastrows::2}
for arm in $ARMS; do
  name=${arm%%:*}; rest=${arm#*:}; lib=${rest%%:*}; impl=${rest#*:}
  (cd /tmp && PD_LIB=${lib:+$REPO/$lib} PD_SWEEP_IMPL=$impl timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT -o $name -- \
     python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step --launch eager $BENCH_FLAGS > $REPO/$OUT/$name.log 2>&1); echo "$name rc=$?"
done
python - <<'PY'
import csv, glob, collections
OUT = 'gpurun_out/r5/fetch_ab'
for f in sorted(glob.glob(OUT + '/*_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if ('rowstream' in k or 'fwdstream' in k) and r['Counter_Name'] == 'FETCH_SIZE':
            acc[k.split('(')[0].replace('void pd::', '')[:40]].append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k: round(sum(v) / len(v)) for k, v in acc.items()}, '(KiB-units per launch)')
PY
