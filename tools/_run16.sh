cd /root/repo
for k in "" "ln_fold=3" "ln_fold=0"; do echo "KNOBS=$k"; KNOBS=$k TOKENS=2048,4096,8192,12288,16384,24576 python tools/packed_batch_timing.py 2>&1 | grep -v amdgpu.ids | cut -c1-140; done
