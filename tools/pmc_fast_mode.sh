# wait / matrix-pipe profile of every kernel of the f16x3 forward (one PMC pass, no trace flags)
export TMPDIR=/tmp
R=$PWD
cd /tmp
mkdir -p $R/gpurun_out/pf; rm -rf $R/gpurun_out/pf/*
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pf -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --gemm-dtype ${MODE:-f16x3} > $R/gpurun_out/pf.log 2>&1
echo rc=$?
find $R/gpurun_out -name "*.db" -delete
