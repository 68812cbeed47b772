# A build from sources ON the GPU box (the driver normally finds the prebuilt library that travelled with the snapshot):
# remove the library, build(), what the built file reports about itself, smoke(), one short bench.
mkdir -p gpurun_out
{
rm -f planedepth_amd/lib/libplanedepth_hip.so
t0=$(date +%s); python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "version|error"; echo "build wall $(( $(date +%s) - t0 )) s"
ls -la planedepth_amd/lib/libplanedepth_hip.so
python -c "import __graft_entry__ as g; print('tree source_hash', g.source_hash(), '| hash compiled into the file', g.embedded_hash())"
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step 2>/dev/null | grep -oE '"value": [0-9.]*|"library": \{[^}]*\}' | head -2
} 2>&1 | tee gpurun_out/clean_build.log
