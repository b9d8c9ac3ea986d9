# Round 6: the CLI's pooled path pipelines its token-packed groups; A/B of the split threshold (RNAMSM_PIPELINE_SPLIT_TOKENS: 0 = a pool
# that fits one group stays one group; 49152; 24576), ROUNDS rounds (default 5), separate processes on one box, interleaved.
O=gpurun_out
: > $O/r06_cli_pipeline_ab.log
for rnd in $(seq 1 ${ROUNDS:-5}); do
  for st in 0 49152 24576; do
    echo "=== round $rnd RNAMSM_PIPELINE_SPLIT_TOKENS=$st" >> $O/r06_cli_pipeline_ab.log
    RNAMSM_PIPELINE_SPLIT_TOKENS=$st N=1 M=8 L=40 python3 tools/cli_throughput.py 2>/dev/null | grep -E "alignments \(" >> $O/r06_cli_pipeline_ab.log
  done
done
python3 - <<'PY'
import re, statistics, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
st = None
for line in open("gpurun_out/r06_cli_pipeline_ab.log"):
    m = re.match(r"=== round \d+ RNAMSM_PIPELINE_SPLIT_TOKENS=(\d+)", line)
    if m:
        st = int(m.group(1)); continue
    m = re.match(r"64 (\w+) alignments.*default \(token-packed groups\) [\d.]+ s = ([\d.]+) MSA/s", line)
    if m:
        res[m.group(1)][st].append(float(m.group(2)))
for kind, d in res.items():
    print(kind, " | ".join(f"split {s}: median {statistics.median(v):.1f} MSA/s (min {min(v):.1f}, max {max(v):.1f}, n={len(v)})" for s, v in sorted(d.items())))
PY
