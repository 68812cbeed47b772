# Round 5: the GPU suite on the current build, then the default bench line with and without PD_BWD_PLANE_ZEROED (A/B/A).
mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r5/smoke.log 2>&1; echo "smoke rc=$?"
timeout 3000 python -m pytest tests/ -q -m gpu ${PYTEST_ARGS:-} > gpurun_out/r5/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5/pytest_gpu.log
for tag in adds1 adds0 adds1b adds0b; do
  case $tag in adds0*) export PD_PLANE_ADDS=0;; *) export PD_PLANE_ADDS=1;; esac
  timeout 600 python bench.py > gpurun_out/r5/bench_$tag.json 2> gpurun_out/r5/bench_$tag.err; echo "bench $tag rc=$?"
  python - <<PY
import json
j = json.loads(open("gpurun_out/r5/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", j["value"], j["ms_per_step"], j.get("roofline_path", {}).get("frac"), j.get("copy_bandwidth", j.get("box", "")))
PY
done
cp gpurun_out/r5/bench_adds1.json gpurun_out/r5/bench.json
