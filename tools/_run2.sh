set -x
cd /root/repo
mkdir -p gpurun_out/r4b
python tools/_dbg_col16.py 2>&1 | grep -v amdgpu.ids | grep -v "nonfinite 0  v4 err [0-9.e+-]* nonfinite 0  v5 err [0-9.e+-]* nonfinite 0  v3" > gpurun_out/r4b/dbg.log; cat gpurun_out/r4b/dbg.log | head -20
timeout 1200 python -m pytest tests/test_gpu_attn16.py -x -q 2>&1 | tail -15 > gpurun_out/r4b/test_attn16.log
cat gpurun_out/r4b/test_attn16.log
MODES=bf16 VARIANTS=1,3 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4b/ab_cfg3.log; cat gpurun_out/r4b/ab_cfg3.log
R=1024 C=1024 MODES=bf16,f16x3 VARIANTS=1,3 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4b/ab_cfg4.log; cat gpurun_out/r4b/ab_cfg4.log
R=1024 C=1024 MODES=bf16 VARIANTS=1 TAG=pa bash tools/pmc_attn16.sh > gpurun_out/r4b/pmc.log 2>&1
cat gpurun_out/pa_summary.txt
