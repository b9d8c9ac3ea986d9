# Kernel trace + SQ counters of the bf16 plane GEMMs, 256x256 kernels vs the epilogue-hiding kernel (EXPERIMENTS R3.1).
# Afterwards, here: python tools/summarize_pmc.py r03_gemm16pp pmcp
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O/prof_pp $O/pmcp3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pp -- python3 $R/tools/gemm16_pp_driver.py > $O/prof_pp_run.log 2>&1
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
timeout -s KILL 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmcp3 -- python3 $R/tools/gemm16_pp_driver.py > $O/pmcp3.log 2>&1
find $O -name "*.db" -delete
