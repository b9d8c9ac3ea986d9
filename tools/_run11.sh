set -x
cd /root/repo
mkdir -p gpurun_out/r4k
python tools/code_objects.py
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attn16.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q 2>&1 | tail -5
ROUNDS=2 VARIANTS=1 MODES=bf16,f16x3 python tools/attn16_ab.py
R=1024 C=1024 ROUNDS=2 VARIANTS=1 MODES=bf16,f16x3 python tools/attn16_ab.py
python tools/gemm16_planes_bench.py bf16 f16x3
timeout 300 python tools/clock_vs_data.py > gpurun_out/r4k/clock_vs_data.log 2>&1
cat gpurun_out/r4k/clock_vs_data.log | tail -30
