# Build tuning variants of the library next to the product one:
#   scripts/build_variants.sh name "flags" [name "flags" ...]  ->  planedepth_amd/lib/libpd_var_<name>.so
# (picked up by scripts/diag_kernel_ab.py, or by anything else through PD_LIB).  Objects are cached under /tmp/pd_obj: a
# variant recompiles only the sources its -D flags can reach (FILES="a.hip b.hip" narrows that by hand; default: the sweep's
# row kernels), everything else is compiled once with the product flags.
cd "$(dirname "$0")/.."
OBJ=/tmp/pd_obj; mkdir -p $OBJ/base
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Iplanedepth_amd/csrc"
FILES=${FILES:-"pd_plane_sweep_fwdstream.hip pd_plane_sweep_rowstream.hip pd_plane_sweep_rowshift.hip pd_plane_sweep.hip"}
stamp=$(cat planedepth_amd/csrc/*.h include/*.h | sha1sum | cut -c1-12)
for f in planedepth_amd/csrc/*.hip; do
  b=$(basename $f .hip); h=$(sha1sum < $f | cut -c1-12)
  [ -f $OBJ/base/$b.$h.$stamp.o ] || { rm -f $OBJ/base/$b.*.o; ( $CC -c $f -o $OBJ/base/$b.$h.$stamp.o 2>&1 | grep -E "error:" && echo "FAILED: $f" ) & }
done
wait
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    mkdir -p $OBJ/$name; objs=""
    for f in planedepth_amd/csrc/*.hip; do
      b=$(basename $f .hip)
      if echo " $FILES " | grep -q " $b.hip "; then
        $CC $flags -c $f -o $OBJ/$name/$b.o 2>&1 | grep -E "error|spill"; objs="$objs $OBJ/$name/$b.o"
      else objs="$objs $(ls $OBJ/base/$b.*.o)"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o planedepth_amd/lib/libpd_var_$name.so 2>&1 | grep -E "error"
  ) &
done
wait
ls -la planedepth_amd/lib/ | grep pd_var
