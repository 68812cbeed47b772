# Round 4: row order of the forward / backward (ascending or descending) in the hot-path loop and inside the DDP step.
mkdir -p gpurun_out/r4
b() { name=$1; shift; timeout 600 env "$@" python bench.py --steps 50 --warmup 5 --no_cpu_baseline --no_next_rows > gpurun_out/r4/ord_$name.log 2>&1
  python - "$name" gpurun_out/r4/ord_$name.log <<'PY'
import json, sys
name, path = sys.argv[1], sys.argv[2]
line = [l for l in open(path) if l.startswith("{")]
if not line: print(name, "no JSON line"); raise SystemExit
r = json.loads(line[-1]); k = r["kernels"]; d = r.get("ddp_step", {})
print(name, "value", r["value"], "fwd", k["fwd_ms"], "bwd", k["bwd_ms"], "| ddp step", d.get("ms_per_step"), "sweep fwd/bwd", d.get("sweep_fwd_ms"), d.get("sweep_bwd_ms"),
      "tail fwd/bwd", d.get("tail_fwd_ms"), d.get("tail_bwd_ms"))
PY
}
for rep in $(seq 1 ${REP:-2}); do
  b base_$rep PD_DUMMY=1
  for v in $VARIANTS; do b ${v}_$rep PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so; done
done
