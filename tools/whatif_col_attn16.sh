# What-if builds of col_attn16_kernel (wrong results, timing only): which part of the 32-key tile costs what at R = C = 1024?
#   bash tools/whatif_col_attn16.sh build   (anywhere with hipcc: one library per C16_WHATIF mask in rna-msm_amd/csrc/build-whatif/,
#                                            which travels to the GPU box with the snapshot and is git-ignored)
#   bash tools/whatif_col_attn16.sh run     (on the GPU box: times the prescaled column kernel per mask, tools/attn16_ab.py)
# Masks: 32 never fall back to the TRACKED loop (needed with every other mask: garbage sums would trigger it), 1 no v_exp, 2 no row sums, 4 no LDS-DMA inside the loop, 8 no P V MFMAs, 16 no S MFMAs 64 one 64-key chunk per block instead of R / 64 (sums combine).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/rna-msm_amd/csrc/build-whatif
MASKS=${MASKS:-32 33 34 35 36 40 48 56 59 63 96 127}
if [ "$1" = build ]; then
  mkdir -p $OUT
  make -C $REPO/rna-msm_amd/csrc -j8 > /dev/null
  cd $REPO/rna-msm_amd/csrc
  OBJS=$(ls build/*.o | grep -v col_attn16)
  for m in $MASKS; do
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DC16_WHATIF=$m -c col_attn16.hip -o $OUT/col_attn16_$m.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$m.so $OBJS $OUT/col_attn16_$m.o && rm $OUT/col_attn16_$m.o ) &
  done
  wait
  ls -la $OUT
else
  cd $REPO
  for m in $MASKS; do
    echo "== C16_WHATIF=$m"
    RNAMSM_LIB_PATH=$OUT/lib_$m.so R=${R:-1024} C=${C:-1024} MODES=bf16 VARIANTS=1 ROUNDS=2 python3 tools/attn16_ab.py 2>/dev/null | tr '[' '\n' | grep "col(prescaled" | sed 's/logits.*col(prescaled q)/col(prescaled q)/'
  done
fi
