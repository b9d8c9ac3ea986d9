set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "ragged or packed" 2>&1 | tail -15
