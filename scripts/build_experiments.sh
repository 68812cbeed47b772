# Build the EXPERIMENTS library: the product sources plus the kernels that lost their A/B runs (scripts/experiments/*.hip:
# four-pixels-per-lane row kernels, owned-tile backward; the one-kernel plane-uniform backward is an include of
# pd_plane_sweep_uniform.hip) with -DPD_EXPERIMENTS -> planedepth_amd/lib/libpd_experiments.so.  Their tests live in
# tests/experiments/ and run with
#   PD_TEST_EXPERIMENTS=1 PD_LIB=$PWD/planedepth_amd/lib/libpd_experiments.so python -m pytest tests/experiments -q
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -DPD_EXPERIMENTS -Iinclude -Iplanedepth_amd/csrc \
  -Iscripts/experiments $EXTRA planedepth_amd/csrc/*.hip scripts/experiments/*.hip -o planedepth_amd/lib/libpd_experiments.so 2>&1 | grep -E "error|spill"
ls -la planedepth_amd/lib/libpd_experiments.so
