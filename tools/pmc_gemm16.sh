set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pg1
cd /tmp
ONLY=${ONLY:-qkv} REPS=4 VARIANTS=${VARIANTS:-2,5} rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pg1 -- python3 $R/tools/gemm16_pl_ab.py > $R/gpurun_out/pg1.log 2>&1
tail -2 $R/gpurun_out/pg1.log
find $R/gpurun_out -name "*.db" -delete
