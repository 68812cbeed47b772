# PMC passes on the plane-uniform kernels (--mono_pose): what paces the 8 dword gathers per pixel and plane?
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r4/unipmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -oE "\b(TCP|TA|TCC|TD)_[A-Za-z0-9_]+" $OUT/counters.txt | sort -u > $OUT/mem_counters.txt; wc -l $OUT/mem_counters.txt
pmc() { tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/$OUT -o $tag -- python $REPO/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp --mono_pose --launch eager ${EXTRA} > $REPO/$OUT/$tag.log 2>&1); echo "pmc $tag rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pmc tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pmc tcp2 TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pmc ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r4/unipmc/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'uniform' not in k and 'fwdstream' not in k: continue
        out[k.split('(')[0][-45:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()): print('   %-42s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
