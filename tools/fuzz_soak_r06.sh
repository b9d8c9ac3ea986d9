# Round-6 randomised evidence at the final tree (one gpurun call): forward fuzz against the fp64 truth incl. 18-40 k-token cases with
# mixed-tile GEMM plans and tile bit-identity, the packed-members-bit-identical fuzz, the kernel fuzz, the determinism soak.
set -x
O=gpurun_out
FUZZ_BIG_EVERY=4 python3 tests/analysis/fuzz_forward.py 48 811 6000 > $O/r06_fuzz_forward_big.log 2>&1; tail -1 $O/r06_fuzz_forward_big.log
python3 tests/analysis/fuzz_forward.py 100 812 24000 > $O/r06_fuzz_forward_24k.log 2>&1; tail -1 $O/r06_fuzz_forward_24k.log
FUZZ_PACKED_EVERY=1 FUZZ_MODE=f32 FUZZ_KNOBS=gemm_splitk=0,gemm_splitk_short=0,ln_fold=1 python3 tests/analysis/fuzz_forward.py 80 813 24000 > $O/r06_fuzz_packed_bits.log 2>&1; tail -1 $O/r06_fuzz_packed_bits.log
grep -c "bit-identical: True" $O/r06_fuzz_packed_bits.log; grep -c "bit-identical: False" $O/r06_fuzz_packed_bits.log
python3 tests/analysis/fuzz_kernels.py 60 814 > $O/r06_fuzz_kernels.log 2>&1; tail -1 $O/r06_fuzz_kernels.log
ITER=60 python3 tools/soak_determinism.py > $O/r06_soak.log 2>&1; tail -3 $O/r06_soak.log
