# A build from sources ON the GPU box (the driver normally finds the prebuilt library that travelled with the snapshot):
# remove the library, build(), smoke(), one short bench.
rm -f planedepth_amd/lib/libplanedepth_hip.so
t0=$(date +%s); python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "version|error"; echo "build wall $(( $(date +%s) - t0 )) s"
ls -la planedepth_amd/lib/libplanedepth_hip.so
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
