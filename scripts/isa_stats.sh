# Static instruction statistics of one kernel: scripts/isa_stats.sh <source.hip> <mangled-name substring> [extra flags]
# -> /tmp/isa/<substring>.s plus counts (VALU / SALU / branches / bool materialisations / SGPR spill traffic).
src=$1; pat=$2; shift 2
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I$(dirname $0)/../include -I$(dirname $0)/../planedepth_amd/csrc "$@" \
  --cuda-device-only -S -o /tmp/isa/_all.s $src 2>&1 | grep -E "error" 
awk -v K="$pat" 'index($0,K) && /^_Z[^ ]*:/ && !k {k=1} k{print} k&&/\.end_amdhsa_kernel/{exit}' /tmp/isa/_all.s > /tmp/isa/$pat.s
f=/tmp/isa/$pat.s
echo "$pat lines=$(wc -l < $f) valu=$(grep -c '^\sv_' $f) salu=$(grep -c '^\ss_' $f) branches=$(grep -c s_cbranch $f) bool01=$(grep -c 'v_cndmask_b32_e64 v[0-9]*, 0, 1,' $f) lane_spills=$(grep -c 'v_readlane\|v_writelane' $f) vmem=$(grep -c '^\sbuffer_\|^\sglobal_' $f) lds=$(grep -c '^\sds_' $f)"
grep -E "^\s+\.(vgpr_count|sgpr_spill_count|vgpr_spill_count):" /tmp/isa/_all.s >/dev/null
