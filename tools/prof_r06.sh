# Round-6 evidence run (one gpurun call): bench lines, rocprofv3 kernel traces and PMC passes for the exact path and the 16-bit modes.
# bench.py prints ONE compact line (<= 4 KB); the complete result of every run goes to <name>.json through --detail-out, the line to
# <name>.line.json.  Afterwards, here: bash tools/fold_evidence.sh
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O/prof $O/prof_bf16 $O/prof_f16x3
run() { n=$1; shift; python3 bench.py --detail-out $O/$n.json "$@" > $O/$n.line.json 2>> $O/bench.err; }
run r06_bench
tail -c 1400 $O/r06_bench.line.json
run r06_bench_f16x3 --no-cpu-baseline --gemm-dtype f16x3
run r06_bench_bf16 --no-cpu-baseline --gemm-dtype bf16
run r06_bench_cfg4_f32 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --num-seqs 1024 --seq-len 1024
run r06_bench_cfg4_bf16 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype bf16 --num-seqs 1024 --seq-len 1024
run r06_bench_cfg4_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype f16x3 --num-seqs 1024 --seq-len 1024
run r06_bench_configs3_n1 --workload configs3 --no-cpu-baseline
N=1 M=8 L=40 python3 tools/cli_throughput.py > $O/r06_cli_throughput.log 2>> $O/bench.err
python3 tools/packed_batch_timing.py > $O/r06_packed_batch_timing.log 2>> $O/bench.err
( time python3 -m pytest tests -m gpu -q -x ) > $O/r06_gpu_suite.log 2>&1
tail -3 $O/r06_gpu_suite.log
cd /tmp
B="python3 $R/bench.py --detail-out= --no-cpu-baseline --no-fast-mode --no-per-config"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --steps 4 --warmup 1 > $O/prof_run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- $B --steps 4 --warmup 1 --gemm-dtype bf16 > $O/prof_bf16_run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16x3 -- $B --steps 4 --warmup 1 --gemm-dtype f16x3 > $O/prof_f16x3_run.log 2>&1
for spec in "cfg0:512:36" "cfg1:64:128"; do
  tag=${spec%%:*}; rest=${spec#*:}; m=${rest%%:*}; l=${rest#*:}
  rm -rf $O/prof_$tag; mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- $B --steps 10 --warmup 2 --num-seqs $m --seq-len $l > $O/prof_${tag}_run.log 2>&1
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/r06_${tag}_kernel_stats.csv
done
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
for spec in "pmc:f32:" "pmcb:bf16:" "pmcf:f16x3:" "pmcc:bf16:--num-seqs 1024 --seq-len 1024"; do
  pre=${spec%%:*}; rest=${spec#*:}; dt=${rest%%:*}; extra=${rest#*:}
  rm -rf $O/${pre}1 $O/${pre}2 $O/${pre}3 $O/${pre}4
  mkdir -p $O/${pre}1 $O/${pre}2 $O/${pre}3 $O/${pre}4
  timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${pre}1 -- $B --steps 1 --warmup 1 --gemm-dtype $dt $extra > $O/${pre}1.log 2>&1
  timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${pre}2 -- $B --steps 1 --warmup 1 --gemm-dtype $dt $extra > $O/${pre}2.log 2>&1
  timeout -s KILL 400 rocprofv3 --pmc $SQ --output-format csv -d $O/${pre}3 -- $B --steps 1 --warmup 1 --gemm-dtype $dt $extra > $O/${pre}3.log 2>&1
  if [ "$pre" != "pmcc" ]; then
    timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/${pre}4 -- $B --steps 1 --warmup 1 --gemm-dtype $dt $extra > $O/${pre}4.log 2>&1
  fi
done
find $O -name "*.db" -delete
find $O -size +20M -delete
du -sh $O
