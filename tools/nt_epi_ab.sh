# A/B of the epilogue traffic policy (RNAMSM_NT_EPI): two builds of the library, same box, alternating processes.
# Build the second library first, into a copy next to the default one:
#   make -C rna-msm_amd/csrc clean && make -C rna-msm_amd/csrc -j8 CXXFLAGS+=-DRNAMSM_NT_EPI=0 \
#     && cp rna-msm_amd/rnamsm/librnamsm_hip.so rna-msm_amd/rnamsm/librnamsm_hip_plain.so \
#     && make -C rna-msm_amd/csrc clean && make -C rna-msm_amd/csrc -j8
for i in 1 2; do
  for lib in librnamsm_hip.so librnamsm_hip_plain.so; do
    echo "== $lib"
    RNAMSM_LIB_PATH=$PWD/rna-msm_amd/rnamsm/$lib python3 bench.py --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('f32', round(d['value']), round(d['ms_per_step'],2), 'gemm', round(d['kernel_ms_per_msa']['gemm_f32'],2), '| f16x3', round(d['fast_mode']['value']), round(d['fast_mode']['kernel_ms_per_step']['gemm_f32'],2), '| bf16', round(d['bf16_mode']['value']), round(d['bf16_mode']['kernel_ms_per_step']['gemm_f32'],2))"
  done
done
