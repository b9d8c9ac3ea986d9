# After `gpurun -- bash tools/prof_r05.sh` (round 5): copy what is judged from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
for t in "" _bf16 _f16x3; do
  f=$(ls -t gpurun_out/prof$t/*/*kernel_stats.csv | head -1); cp "$f" profiles/r05${t}_kernel_stats.csv
done
for n in r05_bench r05_bench_f16x3 r05_bench_bf16 r05_bench_cfg4_f32 r05_bench_cfg4_bf16 r05_bench_cfg4_f16x3 r05_bench_configs3_n1; do
  tail -1 gpurun_out/$n.json > profiles/$n.json
done
cp gpurun_out/r05_cli_throughput.log gpurun_out/r05_packed_batch_timing.log profiles/
tail -5 gpurun_out/r05_gpu_suite.log > profiles/r05_gpu_suite.log
cp gpurun_out/r05_fullsize_parity.json profiles/ 2>/dev/null || true
python tools/summarize_pmc.py r05 pmc > /dev/null
python tools/summarize_pmc.py r05_bf16 pmcb > /dev/null
python tools/summarize_pmc.py r05_f16x3 pmcf > /dev/null
python tools/summarize_pmc.py r05_cfg4_bf16 pmcc > /dev/null
python - <<'PY'
import json
for n in ("r05_bench","r05_bench_f16x3","r05_bench_bf16","r05_bench_cfg4_f32","r05_bench_cfg4_bf16","r05_bench_cfg4_f16x3","r05_bench_configs3_n1"):
    d=json.loads(open(f"profiles/{n}.json").read()); r=d["roofline"]
    print(n, round(d["value"]), "res/s", round(d["ms_per_step"],2), "ms", d["dtype"][:6], "frac", round(r["frac"],3), "traffic", r.get("traffic"))
    if n=="r05_bench":
        for k in ("fast_mode","bf16_mode"):
            m=d[k]; print("  ",k, round(m["value"]), round(m["ms_per_step"],2), "frac", round(m["roofline"]["frac"],3), {a:round(b,2) for a,b in m["kernel_ms_per_step"].items()})
        print("   small", [(c["num_seqs"],c["seq_len"],round(c["residues_per_s_batched"])) for c in d["small_msa_batches"]["cases"]])
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
grep -E "alignments|async_io" profiles/r05_cli_throughput.log
