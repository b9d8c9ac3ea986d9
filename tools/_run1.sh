set -x
cd /root/repo
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_attn16.py -x -q -k "col or padding or forward_attn16" 2>&1 | tail -15 > gpurun_out/r4a/test_col.log
cat gpurun_out/r4a/test_col.log
VARIANTS=1,4,5,3 ROUNDS=2 timeout 600 python tools/attn16_ab.py > gpurun_out/r4a/ab_cfg3.log 2>&1; cat gpurun_out/r4a/ab_cfg3.log
R=1024 C=1024 VARIANTS=1,4,5,3 ROUNDS=2 timeout 600 python tools/attn16_ab.py > gpurun_out/r4a/ab_cfg4.log 2>&1; cat gpurun_out/r4a/ab_cfg4.log
