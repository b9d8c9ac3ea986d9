#!/usr/bin/env python3
"""Per-dispatch MFMA-pipe utilisation and clock estimate of the 16-bit GEMM kernels from a rocprofv3 --pmc run
(gpurun_out/pg1): SQ_BUSY_CYCLES sums 32 shader engines, so clock = BUSY / 32 / duration; MFMA busy = MFMA_BUSY / 1024 SIMDs."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pg1"
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(f)):
    if "gemm16" not in r["Kernel_Name"] and "gemm_bf16" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("(")[0][-60:], int(r["Dispatch_Id"]))
    v = seen.setdefault(key, {})
    v[r["Counter_Name"]] = float(r["Counter_Value"])
    v["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (k, did), v in sorted(seen.items(), key=lambda kv: kv[0][1]):
    ghz = v["SQ_BUSY_CYCLES"] / 32 / v["dur_us"] / 1e3
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["dur_us"] * 1e3 * ghz)
    print(f"{did:4d} {k:62s} {v['dur_us']:8.1f} us  clock~{ghz:.2f} GHz  MFMA busy {100*busy:.0f}%  wait_any/wave {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.2f}")
