export TMPDIR=/tmp
REPO=$PWD
for v in product noho; do
  if [ $v != product ]; then export PD_LIB=$REPO/planedepth_amd/lib/libpd_var_$v.so; fi
  mkdir -p gpurun_out/atom_$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_REQ_sum --output-format csv -d $REPO/gpurun_out/atom_$v -o k -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_next_rows > $REPO/gpurun_out/atom_$v/run.log 2>&1)
  python - $v <<PY
import csv,sys,collections
v=sys.argv[1]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f"gpurun_out/atom_{v}/k_counter_collection.csv")):
    if "rowshift_bwd" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(v, {k: sum(x)/len(x) for k,x in acc.items()})
PY
done
