set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 $R/gpurun_out/pmc3
python3 bench.py --steps 8 --warmup 2 > $R/gpurun_out/r01_bench.json 2> $R/gpurun_out/bench.err
tail -c 3000 $R/gpurun_out/r01_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/prof_run.log 2>&1
tail -3 $R/gpurun_out/prof_run.log
find $R/gpurun_out/prof -name "*stats*" | head
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/pmc3.log 2>&1
tail -2 $R/gpurun_out/pmc3.log
# keep only small files
find $R/gpurun_out -name "*.db" -delete
find $R/gpurun_out -size +20M -delete
du -sh $R/gpurun_out
