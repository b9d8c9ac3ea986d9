# refresh only the 16-bit-mode artefacts of profiles/ (kernel trace of the f16x3 forward + the mode bench lines)
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_fast; rm -rf $R/gpurun_out/prof_fast/*
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --gemm-dtype f16x3 > $R/gpurun_out/prof_fast_run.log 2>&1
cd $R
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --gemm-dtype f16x3 > $R/gpurun_out/r01_bench_f16x3.json 2>> $R/gpurun_out/bench.err
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --gemm-dtype bf16 > $R/gpurun_out/r01_bench_bf16.json 2>> $R/gpurun_out/bench.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype f16x3 --num-seqs 1024 --seq-len 1024 > $R/gpurun_out/r01_bench_cfg4_f16x3.json 2>> $R/gpurun_out/bench.err
find $R/gpurun_out -name "*.db" -delete
