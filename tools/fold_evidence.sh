# After `gpurun -- bash tools/prof_r03.sh`: copy what is judged from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
for t in "" _bf16 _f16x3; do
  f=$(ls -t gpurun_out/prof$t/*/*kernel_stats.csv | head -1); cp "$f" profiles/r03${t}_kernel_stats.csv
done
for n in r03_bench r03_bench_f16x3 r03_bench_bf16 r03_bench_cfg4_f32 r03_bench_cfg4_bf16 r03_bench_cfg4_f16x3 r03_bench_configs3_n1; do
  tail -1 gpurun_out/$n.json > profiles/$n.json
done
cp gpurun_out/r03_cli_throughput.log profiles/
python tools/summarize_pmc.py r03 pmc > /dev/null
python tools/summarize_pmc.py r03_bf16 pmcb > /dev/null
python tools/summarize_pmc.py r03_f16x3 pmcf > /dev/null
python tools/summarize_pmc.py r03_cfg4_bf16 pmcc > /dev/null
python - <<'PY'
import json
for n in ("r03_bench","r03_bench_f16x3","r03_bench_bf16","r03_bench_cfg4_f32","r03_bench_cfg4_bf16","r03_bench_cfg4_f16x3","r03_bench_configs3_n1"):
    d=json.loads(open(f"profiles/{n}.json").read()); r=d["roofline"]
    print(n, round(d["value"]), "res/s", round(d["ms_per_step"],2), "ms", d["dtype"][:6], "frac", round(r["frac"],3), "traffic", r.get("traffic"))
    if n=="r03_bench":
        for k in ("fast_mode","bf16_mode"):
            m=d[k]; print("  ",k, round(m["value"]), round(m["ms_per_step"],2), "frac", round(m["roofline"]["frac"],3), {a:round(b,2) for a,b in m["kernel_ms_per_step"].items()})
        print("   small", [(c["num_seqs"],c["seq_len"],round(c["residues_per_s_batched"])) for c in d["small_msa_batches"]["cases"]])
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
grep -E "alignments|async_io" profiles/r03_cli_throughput.log
