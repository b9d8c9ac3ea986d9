cd /root/repo
python -m pytest tests/test_gpu_kernels.py -q -k "block_order or half_width or gemm" 2>&1 | tail -3
for T in 8256 12352 18944 32896 49280; do echo "T=$T"; T=$T python tools/gemm_ab.py gemm_tile=0 2>&1 | grep -v amdgpu.ids; done
