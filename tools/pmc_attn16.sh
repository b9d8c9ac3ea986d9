# PMC passes over tools/attn16_ab.py (16-bit attention kernels; shape / modes / variants from the environment): SQ wait / issue breakdown.
set -x
export TMPDIR=/tmp
REPO=$PWD
TAG=${TAG:-pa}
mkdir -p $REPO/gpurun_out/${TAG}1 $REPO/gpurun_out/${TAG}2
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $REPO/gpurun_out/${TAG}1 -- python3 $REPO/tools/attn16_ab.py > $REPO/gpurun_out/${TAG}1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/gpurun_out/${TAG}2 -- python3 $REPO/tools/attn16_ab.py > $REPO/gpurun_out/${TAG}2.log 2>&1
tail -3 $REPO/gpurun_out/${TAG}2.log
find $REPO/gpurun_out -name "*.db" -delete
cd $REPO
python3 - <<PY
import csv, glob, collections
out = []
for d in ("${TAG}1", "${TAG}2"):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("col_attn16", "row_logits16", "row_apply16")): continue
        name = k.split("(")[0].replace("void rnamsm::", "")[:64] + " grid=" + r["Grid_Size"] + " vgpr=" + r["VGPR_Count"]
        a = agg[(name, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for (name, c), (n, s) in sorted(agg.items()):
        out.append(f"{name:100s} {c:28s} n={n:3d} mean={s / n:.4e}")
open("gpurun_out/${TAG}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
