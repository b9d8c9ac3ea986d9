# PMC passes over tools/col_attn_ab.py (both fp32 column-attention kernels, several shapes): SQ wait/issue breakdown.
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pc1 $R/gpurun_out/pc2
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pc1 -- python3 $R/tools/col_attn_ab.py > $R/gpurun_out/pc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pc2 -- python3 $R/tools/col_attn_ab.py > $R/gpurun_out/pc2.log 2>&1
tail -3 $R/gpurun_out/pc2.log
find $R/gpurun_out -name "*.db" -delete
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pc1", "pc2"):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "col_attn" not in k: continue
        name = ("dma" if "dma" in k else "reg") + ("_masked" if "<true" in k.replace(" ", "") or "ILb1" in k else "") + " grid=" + r["Grid_Size"]
        a = agg[(name, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for (name, c), (n, s) in sorted(agg.items()):
        print(f"{name:28s} {c:28s} n={n:3d} mean={s / n:.4e}")
PY
