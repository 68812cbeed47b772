# Round 6: compiler scheduling strategies (-mllvm -amdgpu-sched-strategy=...) and the compile-time `g_plane wanted` row body on the headline
# kernels, one process, every arm checked bit for bit against the product library (variants: scripts/build_variants.sh, see NOTEBOOK 11.5);
# then the post-process operators on the final build (49 / 63 planes, the decoder's xz planes as per-row disparities).
mkdir -p gpurun_out/r6
ARMS="${ARMS:-product ilp memcl itilp itocc itmin track nopost wp wp6}" CHECK=1 TAG=${TAG:-sched} SHAPES="${SHAPES:-headline}" ROUNDS=${ROUNDS:-6} bash scripts/gpu_r6_ab.sh
if [ -z "$NO_PP" ]; then
O=gpurun_out/r6/postprocess_final.txt; : > $O
for a in "--planes 49" "--planes 63" "--planes 63 --xz 14" "--planes 32"; do
  echo "== diag_postprocess $a" | tee -a $O
  timeout 300 python scripts/diag_postprocess.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O
done
fi
