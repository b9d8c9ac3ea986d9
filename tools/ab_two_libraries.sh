set -e
L=rna-msm_amd/rnamsm
for rnd in 1 2 3; do
  for v in old new; do
    cp $L/ab/$v.so $L/librnamsm_hip.so
    echo "=== round $rnd $v"
    CONFIGS=base T=131072,1048576 ROUNDS=2 FORWARD=1 python3 tools/gemm16_knob_ab.py 2>/dev/null | grep -E "six GEMMs|forward|qkv|fc1" | cut -c1-110
    M=1024 L=1024 DTYPE=bf16 ROUNDS=2 N=2 python3 tools/forward_knob_ab.py attn16=1 2>/dev/null | cut -c1-60
  done
done
