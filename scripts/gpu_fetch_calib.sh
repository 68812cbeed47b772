# Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS path's access shapes: the
# wide_probe kernels move a known number of bytes per launch (2 tensors x B*N*H*W floats each way); compare with the
# counters.  Output: gpurun_out/r2/fetch_calib.json  (copied to profiles/r02_fetch_calibration.json)
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r2/calib
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/gpurun_out/r2/calib -o $c -- $REPO/scripts/probes/wide_probe > $REPO/gpurun_out/r2/calib/$c.log 2>&1); echo "$c rc=$?"
done
python - <<'PY'
import csv, collections, json, re
known = 2 * 8 * 49 * 192 * 640 * 4      # bytes of two [8,49,192,640] fp32 tensors
out = {"known_bytes_per_direction": known, "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/r2/calib/%s_counter_collection.csv" % c)):
        m = re.search(r"pattern<(\d+), (\d+), (\d+)>", r["Kernel_Name"])
        if not m or int(r["Grid_Size_X"] if "Grid_Size_X" in r else 0) < 0:
            continue
        acc[m.groups()].append(float(r["Counter_Value"]))
    for (P, U, mode), v in sorted(acc.items()):
        mode = int(mode)
        d = out["kernels"].setdefault("P=%s U=%s mode=%d (%s%s)" % (P, U, mode, "L" if mode & 1 else "-", "S" if mode & 2 else "-"), {})
        kib = sum(v) / len(v)
        d[c + "_KiB"] = kib
        moved = known if ((c == "FETCH_SIZE" and mode & 1) or (c == "WRITE_SIZE" and mode & 2)) else 0
        if moved:
            d[c + "_bytes_per_KiB_counted"] = moved / (kib * 1024)
json.dump(out, open("gpurun_out/r2/fetch_calib.json", "w"), indent=1, sort_keys=True)
for k, d in out["kernels"].items():
    print(k, {a: round(b, 3) for a, b in d.items()})
PY
