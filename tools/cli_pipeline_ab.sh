# Round 6: the CLI's pooled path pipelines its token-packed groups; A/B of the split threshold (RNAMSM_PIPELINE_SPLIT_TOKENS: 0 = a pool
# that fits one group stays one group, as before; default 49152; 24576), two rounds, separate processes on one box.
set -x
O=gpurun_out
: > $O/r06_cli_pipeline_ab.log
for rnd in 1 2; do
  for st in 0 49152 24576; do
    echo "=== round $rnd RNAMSM_PIPELINE_SPLIT_TOKENS=$st" >> $O/r06_cli_pipeline_ab.log
    RNAMSM_PIPELINE_SPLIT_TOKENS=$st N=1 M=8 L=40 python3 tools/cli_throughput.py 2>/dev/null | grep -E "alignments \(" >> $O/r06_cli_pipeline_ab.log
  done
done
cat $O/r06_cli_pipeline_ab.log
