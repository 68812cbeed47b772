# Kernel timeline of the bench step: every kernel of the last steps with its duration and the gap to its predecessor.
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r3/trace
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r3/trace -o t -- python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_next_rows $CFG > $REPO/gpurun_out/r3/trace/t.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# find the steady-state steps: take the last 60 kernels before the final isolated-launch loops is hard; print a window around the
# middle occurrence of the backward kernel
idx = [i for i, r in enumerate(rows) if 'bwd_kernel' in r['Kernel_Name']]
print("backward launches:", len(idx))
mid = idx[6] if len(idx) > 8 else idx[len(idx) // 2]
lo = max(0, mid - 14)
prev_end = None
for r in rows[lo:mid + 10]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1000.0 if prev_end else 0.0
    print("%8.1f us  gap %7.1f us  %s" % ((e - s) / 1000.0, gap, r['Kernel_Name'][:90]))
    prev_end = e
PY
find gpurun_out/r3/trace -name "*.csv" -size +200k -delete
