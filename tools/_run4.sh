set -x
cd /root/repo
mkdir -p gpurun_out/r4d
timeout 1200 python -m pytest tests/test_gpu_attn16.py -x -q 2>&1 | tail -8 > gpurun_out/r4d/test_attn16.log
cat gpurun_out/r4d/test_attn16.log
MODES=bf16 Q16=1,0 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4d/ab_cfg3.log; cat gpurun_out/r4d/ab_cfg3.log
R=1024 C=1024 MODES=bf16 Q16=1,0 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4d/ab_cfg4.log; cat gpurun_out/r4d/ab_cfg4.log
R=128 C=256 MODES=bf16 Q16=1,0 ROUNDS=2 timeout 600 python tools/attn16_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4d/ab_cfg3b.log; cat gpurun_out/r4d/ab_cfg3b.log
R=1024 C=1024 MODES=bf16 VARIANTS=1 TAG=pb bash tools/pmc_attn16.sh > gpurun_out/r4d/pmc.log 2>&1
grep "row_logits" gpurun_out/pb_summary.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -8 > gpurun_out/r4d/test_kernels.log
cat gpurun_out/r4d/test_kernels.log
