# After `gpurun -- bash tools/prof_r06.sh` (round 6): copy what is judged from gpurun_out/ (scratch) into profiles/ (tracked).
# <name>.json = the COMPLETE result (--detail-out), <name>.line.json = the compact line the driver parses.
set -e
for t in "" _bf16 _f16x3; do
  f=$(ls -t gpurun_out/prof$t/*/*kernel_stats.csv | head -1); cp "$f" profiles/r06${t}_kernel_stats.csv
done
cp gpurun_out/r06_cfg0_kernel_stats.csv gpurun_out/r06_cfg1_kernel_stats.csv profiles/
for n in r06_bench r06_bench_f16x3 r06_bench_bf16 r06_bench_cfg4_f32 r06_bench_cfg4_bf16 r06_bench_cfg4_f16x3 r06_bench_configs3_n1; do
  cp gpurun_out/$n.json profiles/$n.json
done
tail -1 gpurun_out/r06_bench.line.json > profiles/r06_bench.line.json
cp gpurun_out/r06_cli_throughput.log gpurun_out/r06_packed_batch_timing.log profiles/
tail -5 gpurun_out/r06_gpu_suite.log > profiles/r06_gpu_suite.log
cp gpurun_out/r06_fullsize_parity.json profiles/ 2>/dev/null || true
python tools/summarize_pmc.py r06 pmc > /dev/null
python tools/summarize_pmc.py r06_bf16 pmcb > /dev/null
python tools/summarize_pmc.py r06_f16x3 pmcf > /dev/null
python tools/summarize_pmc.py r06_cfg4_bf16 pmcc > /dev/null
python - <<'PY'
import json
print("compact line:", len(open("profiles/r06_bench.line.json").read().encode()), "bytes")
for n in ("r06_bench","r06_bench_f16x3","r06_bench_bf16","r06_bench_cfg4_f32","r06_bench_cfg4_bf16","r06_bench_cfg4_f16x3","r06_bench_configs3_n1"):
    d=json.loads(open(f"profiles/{n}.json").read()); r=d["roofline"]
    print(n, round(d["value"]), "res/s", round(d["ms_per_step"],2), "ms", d["dtype"][:6], "frac", round(r["frac"],3), "traffic", r.get("traffic"))
    if n=="r06_bench":
        for k in ("fast_mode","bf16_mode"):
            m=d[k]; print("  ",k, round(m["value"]), round(m["ms_per_step"],2), "frac", round(m["roofline"]["frac"],3), {a:round(b,2) for a,b in m["kernel_ms_per_step"].items()})
        print("   small", [(c["num_seqs"],c["seq_len"],round(c["residues_per_s_batched"])) for c in d["small_msa_batches"]["cases"]])
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
        print("   digest", json.dumps(d["digest"]))
PY
grep -E "alignments|async_io" profiles/r06_cli_throughput.log
