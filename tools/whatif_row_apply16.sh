# What-if builds of row_apply16x_kernel (wrong results, timing only): where do the 2.0 ms of a 1024 x 1024 bf16 launch go?
#   bash tools/whatif_row_apply16.sh build   (anywhere with hipcc: one library per R16X_WHATIF mask in rna-msm_amd/csrc/build-whatif/)
#   bash tools/whatif_row_apply16.sh run     (on the GPU box: tools/attn16_ab.py per mask; the "apply" column)
# Masks: 1 no MFMAs, 2 no LDS-DMA inside the K loop (computing on stale tiles), 4 one K tile per block, 8 no epilogue stores (sums combine).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/rna-msm_amd/csrc/build-whatif
MASKS=${MASKS:-0 1 2 3 4 8 11 15}
if [ "$1" = build ]; then
  mkdir -p $OUT
  make -C $REPO/rna-msm_amd/csrc -j8 > /dev/null
  cd $REPO/rna-msm_amd/csrc
  OBJS=$(ls build/*.o | grep -v row_attn16)
  for m in $MASKS; do
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DR16X_WHATIF=$m -c row_attn16.hip -o $OUT/row_attn16_$m.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libra_$m.so $OBJS $OUT/row_attn16_$m.o && rm $OUT/row_attn16_$m.o ) &
  done
  wait
  ls $OUT | grep libra
else
  cd $REPO
  for m in $MASKS; do
    echo "== R16X_WHATIF=$m"
    RNAMSM_LIB_PATH=$OUT/libra_$m.so R=${R:-1024} C=${C:-1024} MODES=bf16 VARIANTS=1 ROUNDS=2 python3 tools/attn16_ab.py 2>/dev/null | tr '[' '\n' | grep "apply" | sed 's/col .*//'
  done
fi
