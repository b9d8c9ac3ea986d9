# HBM-side read traffic of the fp32 GEMM under the grouped XCD block order (one pass per group value)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for g in 1 4 8 16; do
  mkdir -p $R/gpurun_out/pgg$g; rm -rf $R/gpurun_out/pgg$g/*
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pgg$g -- python3 $R/tools/gemm_ab.py gemm_group=$g,$g > $R/gpurun_out/pgg$g.log 2>&1
  echo "group $g rc=$?"
done
find $R/gpurun_out -name "*.db" -delete
