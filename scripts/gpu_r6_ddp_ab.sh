# Round 6: the DDP training step's legs (fused decoder tail + sweep) under library variants, alternating on one box.
mkdir -p gpurun_out/r6
O=gpurun_out/r6/ddp_ab.txt; : > $O
for i in 1 2; do
  for lib in ${ARMS:-old product bnp bp1}; do
    L=""; [ $lib != product ] && L=$PWD/planedepth_amd/lib/libpd_var_$lib.so
    PD_LIB=$L timeout 300 python bench.py --steps 50 --warmup 10 --no_cpu_baseline --no_next_rows > gpurun_out/r6/ddp_$lib$i.json 2>/dev/null
    python - $lib $i gpurun_out/r6/ddp_$lib$i.json <<'PY' | tee -a $O
import json, sys
lib, i, p = sys.argv[1:4]
l = [x for x in open(p) if x.startswith("{")]
if not l: print(lib, i, "FAILED"); sys.exit(0)
d = json.loads(l[-1]); s = d.get("ddp_step") or {}; t = s.get("tail_backward_as_its_own_kernel") or {}
print("%-8s run %s: value %8.1f | ddp step %.3f ms: sweep fwd %.4f bwd(+tail) %.4f tail fwd %.4f | unfused: step %.3f sweep bwd %.4f tail bwd %.4f" % (
    lib, i, d["value"], s.get("ms_per_step", 0), s.get("sweep_fwd_ms", 0), s.get("sweep_bwd_ms", 0), s.get("tail_fwd_ms", 0),
    t.get("ms_per_step", 0), t.get("sweep_bwd_ms", 0), t.get("tail_bwd_ms", 0)))
PY
  done
done
