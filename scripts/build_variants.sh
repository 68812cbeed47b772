# Build tuning variants of the library next to the product one: scripts/build_variants.sh name "flags" [name "flags" ...]
# -> planedepth_amd/lib/libpd_var_<name>.so (picked up through PD_LIB by scripts/gpu_variants.sh)
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Iinclude -Iplanedepth_amd/csrc $flags \
    planedepth_amd/csrc/*.hip -o planedepth_amd/lib/libpd_var_$name.so 2>&1 | grep -E "error|spill" &
done
wait
ls -la planedepth_amd/lib/
