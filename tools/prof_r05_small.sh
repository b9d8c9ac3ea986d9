# Kernel traces of the exact forward at configs[0]'s and configs[1]'s shapes (round 5: narrow row attention, mixed GEMM tiles).
# One gpurun call; leaves gpurun_out/r05_cfg0_kernel_stats.csv and r05_cfg1_kernel_stats.csv (copy to profiles/).
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-fast-mode --no-per-config"
for spec in "cfg0:512:36" "cfg1:64:128"; do
  tag=${spec%%:*}; rest=${spec#*:}; m=${rest%%:*}; l=${rest#*:}
  rm -rf $O/prof_$tag; mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- $B --steps 10 --warmup 2 --num-seqs $m --seq-len $l > $O/prof_${tag}_run.log 2>&1
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/r05_${tag}_kernel_stats.csv
  tail -c 600 $O/prof_${tag}_run.log
done
find $O -name "*.db" -delete
find $O -size +20M -delete
