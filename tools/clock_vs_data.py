#!/usr/bin/env python3
"""What clock do the 16-bit GEMMs run at, and does it depend on the operand DATA?  (EXPERIMENTS.md R3.3b inferred a data-dependent
clock from timings: the same kernel on all-zero operands ran ~20 % faster.)  This reads the clock back while the kernel runs.
The QKV-shaped plain-bf16 GEMM (T x 2304 x 768, plane outputs) is queued in bursts of ~150 ms; while a burst executes the host
samples the shader clock (rocm-smi / amd-smi / sysfs, whichever this box answers) -- once per burst, ~25 bursts per operand
setting -- and the burst's own time per launch comes from HIP events.  Operand settings: zeros, constant 1.0, N(0,1).
    python tools/clock_vs_data.py > profiles/r04_clock_vs_data.log"""
import glob, json, os, re, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops

dev = torch.device("cuda:0")
BURST = int(os.environ.get("BURST", 400))
T, N, K = int(os.environ.get("T", 131072)), 2304, 768


def _run(cmd):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
    except Exception as e:      # noqa: BLE001 -- a missing tool is an answer here
        return f"<{type(e).__name__}: {e}>"


def sclk_rocm_smi():
    out = _run(["rocm-smi", "--showclocks", "--json"])
    try:
        d = json.loads(out)
        for card, kv in d.items():
            for k, v in kv.items():
                if "sclk" in k.lower():
                    m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                    if m:
                        return int(m.group(1))
    except Exception:           # noqa: BLE001
        pass
    return None


def sclk_amd_smi():
    out = _run(["amd-smi", "metric", "--clock", "--json"])
    m = re.search(r'"gfx_0"\s*:\s*\{[^}]*?"clk"\s*:\s*\{\s*"value"\s*:\s*(\d+)', out, re.S) or re.search(r'"clk"\s*:\s*\{\s*"value"\s*:\s*(\d+)', out)
    return int(m.group(1)) if m else None


def sclk_sysfs():
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(p):
                if "*" in line:
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    if m:
                        return int(m.group(1))
        except OSError:
            pass
    return None


readers = [("rocm-smi --showclocks", sclk_rocm_smi), ("amd-smi metric --clock", sclk_amd_smi), ("sysfs pp_dpm_sclk", sclk_sysfs)]
print("idle readback:", {n: f() for n, f in readers}, flush=True)
print("rocm-smi --showclocks (raw, idle):\n" + _run(["rocm-smi", "--showclocks"]), flush=True)

w = ops.split_bf16(torch.randn(N, K, device=dev) * 0.04, want_lo=False, fmt=0)
b = torch.zeros(N, device=dev)
flops = 2.0 * T * N * K
F32 = os.environ.get("MODE", "bf16") == "f32"       # MODE=f32: the exact path's fp32 GEMM (the headline kernel) instead of the bf16 one
if F32:
    wf = torch.randn(N, K, device=dev) * 0.04
    outf = torch.empty(T, N, device=dev)
print("kernel:", "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)" if F32 else "plain-bf16 plane GEMM (16x16x32 bf16 MFMA)", f"T={T} N={N} K={K}", flush=True)
for tag, make in (("zeros", lambda: torch.zeros(T, K, device=dev)), ("ones", lambda: torch.ones(T, K, device=dev)),
                  ("N(0,1)", lambda: torch.randn(T, K, device=dev)), ("zeros again", lambda: torch.zeros(T, K, device=dev))):
    a = ops.split_bf16(make(), want_lo=False, fmt=0)
    ww = w if tag != "zeros" and tag != "zeros again" else ops.split_bf16(torch.zeros(N, K, device=dev), want_lo=False, fmt=0)
    fn = lambda: ops.linear_planes(a, ww, b, out_planes=True, fmt=0)
    if F32:
        af = make()
        wz = wf if "zeros" not in tag else torch.zeros_like(wf)
        fn = lambda: ops.linear(af, wz, b, out=outf)
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    samples = {n: [] for n, _ in readers}
    per_launch = []
    for burst in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(BURST):
            fn()
        e1.record()
        time.sleep(0.02)                 # the burst is executing now (launches are asynchronous)
        for n, f in readers:
            if burst % len(readers) == readers.index((n, f)):        # one reader per burst: each costs tens of ms
                v = f()
                if v is not None:
                    samples[n].append(v)
        still_running = not e1.query()
        torch.cuda.synchronize()
        per_launch.append((e0.elapsed_time(e1) / BURST, still_running))
    ms = statistics.median(t for t, _ in per_launch)
    busy = sum(1 for _, r in per_launch if r)
    print(f"operands {tag:12s}: {ms:.4f} ms / launch = {flops / ms / 1e9:7.0f} TFLOP/s; bursts still executing after the readback: {busy}/25; "
          + "; ".join(f"{n}: median {statistics.median(v):.0f} MHz (min {min(v)}, max {max(v)}, n={len(v)})" if v else f"{n}: no answer"
                      for n, v in samples.items()), flush=True)
print("rocm-smi --showclocks (raw, after):\n" + _run(["rocm-smi", "--showclocks"]), flush=True)
