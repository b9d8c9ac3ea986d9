# Round 6: the CLI's pooled path pipelines its token-packed groups.  A/B over repeated separate processes on one box, interleaved:
# CONFIGS (default below) = ";"-separated "ENV=val ENV=val" settings of RNAMSM_PIPELINE_SPLIT_TOKENS (a one-pool list of at least that many
# tokens is dealt into two groups; 0 = never).  (RNAMSM_FIRST_POOL_TOKENS -- the first pool of a list starting early -- existed for one A/B in
# round 6, commit d34cda6: small +2.7 % within noise, mid -1.8 %, removed; profiles/r06_cli_first_pool_ab.log.)
O=gpurun_out
LOG=$O/${LOGNAME_:-r06_cli_pipeline_ab.log}
: > $LOG
IFS=';' read -ra CFG <<< "${CONFIGS:-RNAMSM_PIPELINE_SPLIT_TOKENS=0;RNAMSM_PIPELINE_SPLIT_TOKENS=49152;RNAMSM_PIPELINE_SPLIT_TOKENS=24576}"
for rnd in $(seq 1 ${ROUNDS:-5}); do
  for c in "${CFG[@]}"; do
    echo "=== round $rnd $c" >> $LOG
    env $c N=1 M=8 L=40 python3 tools/cli_throughput.py 2>/dev/null | grep -E "alignments \(" >> $LOG
  done
done
python3 - "$LOG" <<'PY'
import re, statistics, collections, sys
res = collections.defaultdict(lambda: collections.defaultdict(list))
st = None
for line in open(sys.argv[1]):
    m = re.match(r"=== round \d+ (.*)", line)
    if m:
        st = m.group(1).replace("RNAMSM_", "").strip(); continue
    m = re.match(r"64 (\w+) alignments.*default \(token-packed groups\) [\d.]+ s = ([\d.]+) MSA/s", line)
    if m:
        res[m.group(1)][st].append(float(m.group(2)))
for kind, d in res.items():
    print(kind, " | ".join(f"[{s}] median {statistics.median(v):.1f} MSA/s ({min(v):.1f}..{max(v):.1f}, n={len(v)})" for s, v in d.items()))
PY
