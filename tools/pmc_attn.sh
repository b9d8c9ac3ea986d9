set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pa1 $R/gpurun_out/pa2
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pa1 -- python3 $R/tools/attn_ab.py > $R/gpurun_out/pa1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pa2 -- python3 $R/tools/attn_ab.py > $R/gpurun_out/pa2.log 2>&1
tail -3 $R/gpurun_out/pa2.log
find $R/gpurun_out -name "*.db" -delete
