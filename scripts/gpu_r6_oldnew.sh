# Round 6: bench.py's default line under the round-5 build of the two headline kernels (variant `old`: default scheduler, no wave priority) and the
# product library, alternating on one box.
mkdir -p gpurun_out/r6
O=gpurun_out/r6/oldnew.txt; : > $O
for i in 1 2 3; do
  for lib in old product; do
    L=""; [ $lib = old ] && L=$PWD/planedepth_amd/lib/libpd_var_old.so
    PD_LIB=$L timeout 300 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_next_rows --no_ddp_step > gpurun_out/r6/on_$lib$i.json 2>/dev/null
    python - $lib $i gpurun_out/r6/on_$lib$i.json <<'PY' | tee -a $O
import json, sys
lib, i, p = sys.argv[1:4]
l = [x for x in open(p) if x.startswith("{")]
if not l: print(lib, i, "FAILED"); sys.exit(0)
d = json.loads(l[-1]); k = d["kernels"]; lp = d["launch_probe"]; w = d["windows"]
print("%-8s run %s: value %8.1f  ms/step %.4f  (%s)  windows median %8.1f [%8.1f .. %8.1f]  in-step fwd %.4f bwd %.4f  isolated %.4f %.4f  probe eager %.4f graph %.4f  copy %.0f GB/s" % (
    lib, i, d["value"], d["ms_per_step"], d["launch"][:5], w["median"], w["min"], w["max"], k["fwd_ms"], k["bwd_ms"], k["isolated_fwd_ms"], k["isolated_bwd_ms"],
    lp["eager_ms_per_step"], lp["graph_ms_per_step"], d["hbm_copy_measured"]["GBs"]))
PY
  done
done
