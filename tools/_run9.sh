set -x
cd /root/repo
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/test_gpu_attn16.py tests/test_gpu_fullsize.py -q -rs 2>&1 | tail -8 > gpurun_out/r4i/tests.log
cat gpurun_out/r4i/tests.log
MODES=bf16,f16x3 VARIANTS=1,4 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4i/ab_cfg3.log; cat gpurun_out/r4i/ab_cfg3.log
R=1024 C=1024 MODES=bf16,f16x3 VARIANTS=1,4 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4i/ab_cfg4.log; cat gpurun_out/r4i/ab_cfg4.log
timeout 900 python bench.py --num-seqs 1024 --seq-len 1024 --gemm-dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-per-config > gpurun_out/r4i/bench_cfg4_bf16.json 2> gpurun_out/r4i/bench_cfg4_bf16.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4i/bench_cfg4_bf16.json') if l.startswith('{')][-1])
print('cfg4 bf16: ms_per_step', d['ms_per_step'], 'value', d['value'])
r=d['roofline']
print(' gemm frac of mfma peak', r['frac'], 'own bound', r.get('frac_of_own_bound'), 'all_kernels', r['all_kernels'])
for k,v in r['per_kernel'].items(): print('  ', k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='bound_terms_ms_per_step'})
PY
