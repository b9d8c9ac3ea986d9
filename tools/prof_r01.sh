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
# fast modes: kernel trace of the f16x3 forward, bench lines of bf16 and of BASELINE config 4 (M = L = 1024)
mkdir -p $R/gpurun_out/prof_fast
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --gemm-dtype f16x3 > $R/gpurun_out/prof_fast_run.log 2>&1
cd $R
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --gemm-dtype f16x3 > $R/gpurun_out/r01_bench_f16x3.json 2>> $R/gpurun_out/bench.err
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --gemm-dtype bf16 > $R/gpurun_out/r01_bench_bf16.json 2>> $R/gpurun_out/bench.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --num-seqs 1024 --seq-len 1024 > $R/gpurun_out/r01_bench_cfg4_f32.json 2>> $R/gpurun_out/bench.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype bf16 --num-seqs 1024 --seq-len 1024 > $R/gpurun_out/r01_bench_cfg4_bf16.json 2>> $R/gpurun_out/bench.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype f16x3 --num-seqs 1024 --seq-len 1024 > $R/gpurun_out/r01_bench_cfg4_f16x3.json 2>> $R/gpurun_out/bench.err
tail -c 600 $R/gpurun_out/r01_bench_cfg4_bf16.json
# keep only small files
find $R/gpurun_out -name "*.db" -delete
find $R/gpurun_out -size +20M -delete
du -sh $R/gpurun_out
