set -x
cd /root/repo
mkdir -p gpurun_out/r4c
R=1024 C=1024 MODES=bf16 VARIANTS=1 TAG=pa bash tools/pmc_attn16.sh > gpurun_out/r4c/pmc.log 2>&1
cat gpurun_out/pa_summary.txt
