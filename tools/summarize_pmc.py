#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (gpurun_out/pmc*/) into profiles/<tag>_pmc_*.csv and <tag>_pmc_summary.json.
HBM-side bytes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read
(MI355X_MICROARCH.md §HBM) and is doubled."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "pmc"          # gpurun_out/<pre>1..4
out = {}
for d, name in ((pre + "1", "fetch_size"), (pre + "2", "write_size"), (pre + "3", "sq"), (pre + "4", "tcc")):
    files = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not files:
        continue
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))       # the newest pass in that directory
    agg = collections.OrderedDict()
    for r in rows:
        key = (r["Kernel_Name"], r["Counter_Name"], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    with open(f"profiles/{tag}_pmc_{name}.csv", "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel_Name", "Counter_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Dispatches", "Mean_Counter_Value"])
        for k, (n, s) in agg.items():
            w.writerow(list(k) + [n, s / n])
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("rnamsm"):
            a = per[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in per.items():
        for c, (n, s) in v.items():
            out.setdefault(k, {})[c + "_per_dispatch"] = s / n
            out[k]["dispatches_" + {"1": "FETCH_SIZE", "2": "WRITE_SIZE", "3": "SQ", "4": "TCC"}[d[-1]]] = n
for k, v in out.items():
    if "FETCH_SIZE_per_dispatch" in v:
        v["hbm_read_bytes_corrected"] = 2 * 1024 * v["FETCH_SIZE_per_dispatch"]
        v["hbm_write_bytes"] = 1024 * v.get("WRITE_SIZE_per_dispatch", 0.0)
    if "TCC_HIT_sum_per_dispatch" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum_per_dispatch"] / max(1.0, v["TCC_HIT_sum_per_dispatch"] + v["TCC_MISS_sum_per_dispatch"])
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch" in v and "SQ_BUSY_CYCLES_per_dispatch" in v:
        # SQ_BUSY_CYCLES sums the 32 shader engines; MFMA busy counts cycles over 1024 SIMDs
        v["mfma_pipe_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"] / 1024.0 / (v["SQ_BUSY_CYCLES_per_dispatch"] / 32.0)
        v["wait_any_share_of_wave_cycles"] = v.get("SQ_WAIT_ANY_per_dispatch", 0.0) / max(1.0, v["SQ_WAVE_CYCLES_per_dispatch"])
        v["wait_inst_any_share_of_wave_cycles"] = v.get("SQ_WAIT_INST_ANY_per_dispatch", 0.0) / max(1.0, v["SQ_WAVE_CYCLES_per_dispatch"])
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k, {a: round(b / 1e6, 1) for a, b in v.items() if "bytes" in a},
          {a: f"{b:.3e}" for a, b in v.items() if a.startswith("SQ_")})
