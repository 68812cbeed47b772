# Round 6: the host side of the eager step — autograd on the calling thread (bench.py's default now) against torch's device thread (--autograd_threads),
# and the raw-stream getter against torch.cuda.current_stream() (PD_RAW_STREAM=0), alternating on one box.
mkdir -p gpurun_out/r6
O=gpurun_out/r6/host_ab.txt; : > $O
for i in 1 2 3; do
  for arm in "threads:--autograd_threads:1" "single::1" "single_slowstream::0"; do
    name=${arm%%:*}; rest=${arm#*:}; flag=${rest%%:*}; raw=${rest##*:}
    PD_RAW_STREAM=$raw timeout 300 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_next_rows --no_ddp_step $flag > gpurun_out/r6/h_$name$i.json 2>/dev/null
    python - $name $i gpurun_out/r6/h_$name$i.json <<'PY' | tee -a $O
import json, sys
lib, i, p = sys.argv[1:4]
l = [x for x in open(p) if x.startswith("{")]
if not l: print(lib, i, "FAILED"); sys.exit(0)
d = json.loads(l[-1]); k = d["kernels"]; lp = d["launch_probe"]; w = d["windows"]
print("%-18s run %s: value %8.1f  ms/step %.4f  (%s)  windows median %8.1f [%8.1f .. %8.1f]  in-step fwd %.4f bwd %.4f  probe eager %.4f graph %.4f" % (
    lib, i, d["value"], d["ms_per_step"], d["launch"][:5], w["median"], w["min"], w["max"], k["fwd_ms"], k["bwd_ms"], lp["eager_ms_per_step"], lp["graph_ms_per_step"]))
PY
  done
done
