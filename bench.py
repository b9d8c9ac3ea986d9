#!/usr/bin/env python3
"""Benchmark of the hot path: MSA-residues/s for the full 10-layer forward (emb + attention maps).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload configs2|configs3]

With N > 1 and no launcher environment this process (which makes no GPU call) starts the N ranks itself, one per
GPU, and exits with their status; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
it runs as one rank.  The control plane (barriers, agreement, timing reductions) is a gloo group; RCCL (backend "nccl") carries
the gathered output arrays only, with a host-staged gloo gather as the second transport if RCCL refuses
(rnamsm.sharding.negotiate_gather_transport).  Rank 0 prints ONE compact JSON line (<= 4 KB: the bench contract's keys,
`roofline`, `cpu_baseline`, the digest) and writes the complete result to --detail-out.

Workloads (BASELINE.json `configs`):
  configs2 (default, the metric's config): one synthetic MSA of M=256 x L=512 per GPU per step -- weak scaling.
  configs3: a batch of 512 synthetic MSAs of M=128 x L=256 dealt round-robin over the ranks
            (rnamsm.sharding.shard_indices), a step = one pass over the whole batch -- strong scaling.
A step = the hot path (K0..K10 through rnamsm_forward) over the step's MSAs, inputs and weights already resident in
HBM, plus (N > 1) the gather of every emb/atp pair to rank 0 with rnamsm.sharding.RoundGatherer (point-to-point RCCL,
round k's transfers overlapping round k+1's forward) -- the same gather the CLI uses.

Three separate loops: W warm-up steps; the HEADLINE loop of exactly K steps with the per-kernel timing hooks off
(`value`, `ms_per_step`); then a ROOFLINE pass over the same work with HIP-event pairs around every launch on the
launch stream (`roofline`, `kernel_ms_per_step`) -- the headline never pays for the instrumentation.
`cpu_baseline` (N = 1) is the oracle (PyTorch-CPU restatement of the reference) on this host's cores at the best
thread count of a sweep.  `fast_mode` / `bf16_mode` are extra measurements of the same workload with every
contraction on the 16-bit matrix cores; they never replace `value`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16/f16 MFMA (no sparsity)
WORKLOADS = {"configs2": "configs2", "configs[2]": "configs2", "single": "configs2",
             "configs3": "configs3", "configs[3]": "configs3", "cfg3": "configs3", "batch512": "configs3"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 8; configs3: 1 pass over the batch)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 2; configs3: 1 short pass)")
    ap.add_argument("--workload", default="configs2", choices=sorted(WORKLOADS))
    ap.add_argument("--num-seqs", type=int, default=None, help="M (default 256; configs3: 128)")
    ap.add_argument("--seq-len", type=int, default=None, help="L, columns including <cls> (default 512; configs3: 256)")
    ap.add_argument("--num-msas", type=int, default=512, help="configs3: MSAs in the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the output gather to rank 0 (N > 1)")
    ap.add_argument("--gemm-dtype", default="f32", choices=["f32", "f16x3", "bf16"],
                    help="arithmetic of the contractions for the headline value (default: exact fp32)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra f16x3 / bf16 / outputs-only measurements")
    ap.add_argument("--no-per-config", action="store_true", help="skip the one-forward-per-BASELINE-config block")
    ap.add_argument("--outputs-only", action="store_true",
                    help="headline on the outputs-only forward (rnamsm_forward without RNAMSM_OUT_REPR: what the CLI runs; the "
                         "last layer skips the rows emb/atp do not depend on).  Default: the complete forward.")
    ap.add_argument("--digest", action="store_true",
                    help="report an order-independent bit digest of every gathered output (N=1 and N=2 must agree)")
    ap.add_argument("--backend", default=os.environ.get("RNAMSM_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="test hook: gloo stages the gather through host memory (RCCL refuses two ranks on one device)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file for the COMPLETE result (per-kernel bounds, per-config block, 16-bit modes ...); stdout carries only the "
                         "compact <= 4 KB line.  Empty string: no file")
    ap.add_argument("--one-device", action="store_true", default=os.environ.get("RNAMSM_BENCH_ONE_DEVICE") == "1",
                    help="test hook: every rank on device 0 (exercises the N > 1 flow on a one-GPU box)")
    args = ap.parse_args(argv)
    args.workload = WORKLOADS[args.workload]
    batch = args.workload == "configs3"
    args.num_seqs = args.num_seqs or (128 if batch else 256)
    args.seq_len = args.seq_len or (256 if batch else 512)
    args.steps = args.steps if args.steps is not None else (1 if batch else 8)
    args.warmup = args.warmup if args.warmup is not None else (1 if batch else 2)
    return args


# --------------------------------------------------------------------------------------------- launcher (no GPU call)
def _run_ranks(args, extra_argv, extra_env, deadline_s) -> int:
    """One attempt: start a rank per GPU as child processes, wait, return 0 / the first failing code / 124 on timeout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra_argv, env=env))
    rc = 0
    t0 = time.time()
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:                  # a rank died: its peers would wait in a collective forever
                        q.terminate()
            if live and time.time() - t0 > deadline_s:
                rc = rc or 124
                for q in live:
                    q.terminate()
                time.sleep(5)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
    return rc if rc >= 0 else 1


def launch_ranks(args) -> int:
    """Start one rank per GPU as child processes and wait for them.  This parent has not touched the GPU (importing
    torch or counting devices does not initialise HIP), so starting children is safe; nothing is ever re-exec'ed.
    If the attempt WITH the output gather fails or exceeds its deadline (a fabric that refuses or stalls point-to-point
    traffic would otherwise leave no scaling number at all), the sharded compute is measured once more without the
    gather and the result line says so."""
    deadline = float(os.environ.get("RNAMSM_BENCH_DEADLINE_S", "1500"))
    import tempfile
    with tempfile.TemporaryDirectory(prefix="rnamsm_bench_") as tmp:
        notes = os.path.join(tmp, "failures.txt")                       # ranks append what went wrong (exception texts)
        rc = _run_ranks(args, [], {"RNAMSM_BENCH_FAILURE_FILE": notes}, deadline)
        if rc != 0 and not args.no_gather and os.environ.get("RNAMSM_BENCH_FAIL_RANK") is None:
            def why_text():
                try:
                    return " | ".join(dict.fromkeys(l.strip() for l in open(notes) if l.strip()))[:600]
                except OSError:
                    return ""
            why = why_text()
            # second transport first: fresh ranks on backend gloo -- the same sharded compute on the same GPUs,
            # every output still gathered to rank 0, staged through host memory -- and only then the compute alone
            if args.backend != "gloo":
                print(f"bench.py: the {args.gpus}-rank run on backend {args.backend} ended with status {rc} ({why or 'no exception text'}); "
                      f"running it again with the gather on gloo (host-staged)", file=sys.stderr, flush=True)
                note = f"gloo fallback after: the run on backend {args.backend} ended with status {rc}" + (f": {why}" if why else "")
                rc = _run_ranks(args, ["--backend", "gloo"], {"RNAMSM_BENCH_GATHER_NOTE": note, "RNAMSM_BENCH_FAILURE_FILE": notes}, deadline)
            if rc != 0:
                why = why_text()
                print(f"bench.py: the {args.gpus}-rank run with the output gather ended with status {rc} ({why or 'no exception text'}); "
                      f"measuring the sharded compute without the gather", file=sys.stderr, flush=True)
                note = f"the run with the gather ended with status {rc}" + (f": {why}" if why else "")
                rc = _run_ranks(args, ["--no-gather", "--backend", "gloo"],
                                {"RNAMSM_BENCH_GATHER_NOTE": note, "RNAMSM_BENCH_FAILURE_FILE": notes}, deadline)
    return rc


# --------------------------------------------------------------------------------------------- helpers
def flops_per_msa(M, L, D=768, layers=10):
    """per token per layer 32 D^2 (8 proj + 2 FFN GEMMs) + 4 D (M + L) (row + column QK^T / PV)."""
    return layers * (32 * D * D + 4 * D * (M + L)) * M * L


def pmc_traffic_per_launch(kernel_prefix="rnamsm::gemm_f32_kernel", tag=""):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_summary.json: FETCH_SIZE x2 per MI355X_MICROARCH.md §HBM, + WRITE_SIZE), launch-weighted over the
    kernel's template instances.  Counters cannot be collected inside a timed run, so this is the profile's figure for
    the same workload, or None when no summary is present."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))
                   if re.fullmatch(rf"r\d+_{tag}pmc_summary\.json", os.path.basename(f)))
    if not files:
        return None, None
    summ = json.load(open(files[-1]))
    tot, n = 0.0, 0
    for name, v in summ.items():
        if name.startswith(kernel_prefix) and "hbm_read_bytes_corrected" in v:
            d = v.get("dispatches_FETCH_SIZE", 1)
            tot += d * (v["hbm_read_bytes_corrected"] + v.get("hbm_write_bytes", 0.0))
            n += d
    return (tot / n if n else None), os.path.basename(files[-1])


def pmc_traffic_by_kernel(kernel_prefix="rnamsm::gemm", tag=""):
    """The same committed PMC passes, per GEMM kernel instance (detail file only): HBM-side read / write bytes per
    launch, launches, matrix-pipe busy fraction -- next to `algorithmic_bytes_per_launch` of the family."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))
                   if re.fullmatch(rf"r\d+_{tag}pmc_summary\.json", os.path.basename(f)))
    if not files:
        return None
    out = {}
    for name, v in json.load(open(files[-1])).items():
        if name.startswith(kernel_prefix) and "hbm_read_bytes_corrected" in v:
            out[name.replace("rnamsm::", "")] = {"hbm_read_bytes": round(v["hbm_read_bytes_corrected"]), "hbm_write_bytes": round(v.get("hbm_write_bytes", 0.0)),
                                                "launches_in_the_pass": v.get("dispatches_FETCH_SIZE"),
                                                "mfma_pipe_busy_frac": v.get("mfma_pipe_busy_frac")}
    return {"source": os.path.basename(files[-1]), "replayed": True, "per_launch": out}


def host_cpu_info():
    """(model name, physical cores, logical CPUs) from /proc/cpuinfo."""
    model, cores = "unknown", set()
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


def cpu_baseline(M, L, state):
    """The oracle (port of the reference) on the host's cores, measured at the BENCHMARKED shape.
    torch's intra-op pool is swept over {os.cpu_count(), physical cores, 1/2, 1/4, 1/8 of them} on ONE layer of the
    M x L MSA of this bench (embedding and final LayerNorm included); at the fastest setting three more samples of that
    layer (their median is the cross-check `per_layer_median_s`), then ONE COMPLETE 10-layer forward, timed whole:
    `value` = M L / `full_forward_s`, no extrapolation (the sweep has paged the weights in and spun the pool up: it is the
    warm-up).  RNAMSM_BENCH_CPU_FULL=0 falls back to the per-layer figure x 10 and says so."""
    import statistics
    import torch
    from oracle import msm_oracle as O
    from rnamsm import synthetic
    model, physical, logical = host_cpu_info()
    params = O.to_torch_params(state)
    toks = torch.from_numpy(synthetic.make_tokens(M, L, 0))
    chunk = 32768 if M * L > 32768 else None                          # FFN hidden formed 32 k tokens at a time (memory only)

    def one_layer():
        t0 = time.perf_counter()
        O.forward(toks, params, layers_to_run=1, ffn_token_chunk=chunk)
        return time.perf_counter() - t0

    sweep = {}
    with torch.no_grad():
        candidates = sorted({logical, physical, physical // 2, physical // 4, physical // 8}, reverse=True)
        candidates = [n for n in candidates if 1 <= n <= logical]
        torch.set_num_threads(candidates[min(1, len(candidates) - 1)])
        one_layer()                                                   # page the weights in, spin the pool up
        for n in candidates:
            torch.set_num_threads(n)
            sweep[n] = one_layer()
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        samples = [one_layer() for _ in range(3)]
        med = statistics.median(samples)
        full_s = None
        if os.environ.get("RNAMSM_BENCH_CPU_FULL", "1") != "0":
            t0 = time.perf_counter()
            O.forward(toks, params, ffn_token_chunk=chunk)            # all ten layers, embedding, final LayerNorm, maps
            full_s = time.perf_counter() - t0
    value = M * L / full_s if full_s else M * L / (10.0 * med)
    return {"value": value, "unit": "MSA-residues/s", "cores": best, "kind": "port",
            "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
            "full_forward_s": None if full_s is None else round(full_s, 3),
            "extrapolation": None if full_s else "one of ten identical layers timed, x 10 (RNAMSM_BENCH_CPU_FULL=0)",
            "thread_sweep_at_bench_shape": {"shape": [M, L], "seconds_per_layer": {str(k): round(v, 4) for k, v in sweep.items()}},
            "per_layer_samples_s": [round(v, 4) for v in samples], "per_layer_median_s": round(med, 4),
            "per_layer_x10_cross_check_residues_per_s": M * L / (10.0 * med),
            "sample": f"oracle/msm_oracle.py on {model} ({physical} cores / {logical} threads), torch {torch.__version__} "
                      f"CPU fp32 with {best} intra-op threads (fastest of a one-layer sweep on this shape): "
                      + (f"ONE complete 10-layer forward of one M={M} L={L} MSA = {full_s:.1f} s"
                         if full_s else f"1 of 10 layers, median of 3 = {med:.2f} s, x 10")
                      + f"; cross-check: one layer (embedding + final LayerNorm included) median of 3 = {med:.2f} s"}


def torch_rocm_eager(M, L, state, dev):
    """What a user of the reference gets on THIS box today: the reference's default device is "cuda"
    (RNA_MSM_Inference.py:20), i.e. PyTorch-ROCm eager ATen kernels.  The oracle (the restatement of the reference's forward,
    oracle/msm_oracle.py; reference semantics incl. max_tokens_per_msa = 16384 -> the chunked path at this size) run on the
    device in fp32: one warm forward, one timed.  The stated same-box PyTorch baseline -- reported, never the target, never
    part of `value`."""
    import torch
    from oracle import msm_oracle as O
    from rnamsm import synthetic
    keep = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        params = O.to_torch_params(state, torch.float32, dev)
        toks = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
        secs = []
        with torch.no_grad():
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                emb, atp = O.pack_outputs(O.forward(toks, params, max_tokens=16384))
                torch.cuda.synchronize()
                secs.append(time.perf_counter() - t0)
        finite = bool(torch.isfinite(emb).all() and torch.isfinite(atp).all())
        peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9
        del params, toks, emb, atp
        torch.cuda.empty_cache()
        return {"value": M * L / secs[1], "unit": "MSA-residues/s", "kind": "port on the device (PyTorch-ROCm eager, fp32)",
                "seconds_warm": round(secs[0], 3), "seconds_timed": round(secs[1], 3), "outputs_finite": finite,
                "torch": torch.__version__, "peak_memory_gb_of_the_process": round(peak_gb, 2),
                "sample": f"oracle/msm_oracle.py on the HIP device in fp32 (torch eager ATen kernels, tf32 off), ONE complete 10-layer "
                          f"forward of one M={M} L={L} MSA after one warm forward, max_tokens_per_msa 16384 as the reference's CLI "
                          f"(RNA_MSM_Inference.py:20,30: device='cuda'); a baseline beside `value`, not the target"}
    except Exception as e:                                            # noqa: BLE001 -- a baseline must not cost the bench line
        torch.cuda.empty_cache()
        return {"value": None, "error": f"{type(e).__name__}: {str(e)[:300]}"}
    finally:
        torch.backends.cuda.matmul.allow_tf32 = keep


def make_digest(result):
    """<= 1.2 KB summary, emitted as the LAST key of the JSON line so that a record holding only the line's tail still carries
    every BASELINE config's numbers: per config [ms, k residues/s, model TFLOP/s, sum of launch bounds / sum of launch times]."""
    def r(v, n=1):
        return None if v is None else round(float(v), n)
    d = {"value_k": r(result["value"] / 1e3), "ms": r(result["ms_per_step"], 2), "n_gpus": result["n_gpus"],
         "dtype": result["dtype"].split(" ")[0], "gemm_frac_of_peak": r(result["roofline"]["frac"], 3),
         "attn_frac_of_peak": r(result["attention_mfma"]["frac"], 3)}
    pc = result.get("per_config")
    if pc:
        d["cfg_cols"] = "ms, k_res_per_s, model_TF, frac_of_bounds"
        d["cfg"] = {k.replace("configs", "c"): [r(v["ms_per_step"], 2), r(v["residues_per_s"] / 1e3), r(v["model_tflops"]), r(v["kernel_frac_of_bounds"], 3)]
                    for k, v in pc.items() if isinstance(v, dict) and "ms_per_step" in v}
    modes = {}
    for key, tag in (("bf16_mode", "bf16"), ("fast_mode", "f16x3"), ("outputs_only_mode", "outputs_only")):
        m = result.get(key)
        if m:
            modes[tag] = [r(m["ms_per_step"], 2), r(m["value"] / 1e3)]
            if "roofline" in m:
                modes[tag].append(r(m["roofline"]["all_kernels"]["frac"], 3))
    if modes:
        d["modes_at_bench_shape"] = modes
        d["modes_cols"] = "ms, k_res_per_s, frac_of_bounds"
    sb = (result.get("small_msa_batches") or {}).get("unlike_shapes")
    if sb:
        d["packed64_k_res_per_s"] = [r(sb["residues_per_s_one_by_one"] / 1e3), r(sb["residues_per_s_packed"] / 1e3)]
    for key, tag in (("cpu_baseline", "cpu_res_per_s"), ("torch_rocm_eager", "torch_rocm_eager_res_per_s")):
        b = result.get(key)
        if b:
            d[tag] = r(b.get("value"), 0)
    if result.get("compute_only_value"):
        d["compute_only_k"] = r(result["compute_only_value"] / 1e3)
    cfg = result.get("config", {})
    if cfg.get("distinct_devices") is not None:
        d["distinct_devices"] = cfg["distinct_devices"]
    d["gather"] = str(cfg.get("gather", ""))[:60]
    return d


COMPACT_LIMIT = 4096               # bytes of the LAST stdout line (the one the driver parses)


def _short(v, n):
    s = " ".join(str(v).split())                                       # one line: exception texts carry newlines
    return s if len(s) <= n else s[:n - 3] + "..."


def compact_line(result):
    """The line the driver parses: every key of the bench contract, `roofline` and `cpu_baseline` with their evidence fields, the
    digest of all BASELINE configs -- and nothing else.  At most COMPACT_LIMIT bytes whatever the run measured (free-text fields are
    clipped; should the line still be too long the optional blocks go first, the contract's keys never).  The complete result -- per-kernel
    bounds, per-config block, 16-bit modes, thread sweep -- goes to the detail file (`--detail-out`), named in `detail`."""
    def num(v, n=6):
        return v if not isinstance(v, float) else float(f"{v:.{n}g}")

    def pick(d, keys, clip=160):
        out = {}
        for k in keys:
            if d is not None and k in d:
                v = d[k]
                out[k] = _short(v, clip) if isinstance(v, str) else num(v)
        return out
    cfg = result.get("config", {})
    line = {k: num(result[k], 9) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                           "scaling", "vs_baseline", "dtype", "data") if k in result}
    line["config"] = dict(pick(cfg, ("workload",), 260), **pick(cfg, ("gather",), 200),
                          **pick(cfg, ("num_seqs", "seq_len", "msas_per_step", "world_size_initialised", "distinct_devices", "backend",
                                       "rccl_version", "devices")))
    roof = result.get("roofline") or {}
    line["roofline"] = pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "flops_per_launch",
                                   "traffic", "algorithmic_bytes_per_launch", "traffic_source", "traffic_replayed", "traffic_unit"), 120)
    if result.get("attention_mfma"):
        line["attention_mfma"] = pick(result["attention_mfma"], ("kernels", "achieved", "peak", "frac"), 60)
    for k in ("compute_only_value", "compute_only_ms_per_step", "outputs_finite", "err_word", "model_tflops"):
        if result.get(k) is not None:
            line[k] = num(result[k])
    if result.get("cpu_baseline"):
        line["cpu_baseline"] = pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu_model", "physical_cores",
                                                             "full_forward_s", "sample"), 240)
    if result.get("torch_rocm_eager"):
        line["torch_rocm_eager"] = pick(result["torch_rocm_eager"], ("value", "unit", "seconds_timed", "error"), 120)
    if result.get("output_digest"):
        line["output_digest"] = pick(result["output_digest"], ("value", "items"))
    gs = result.get("gather_stats")
    if gs:
        line["gather_stats"] = {k: ([num(x, 4) for x in v] if isinstance(v, list) else num(v, 4)) for k, v in gs.items()
                                if k in ("per_rank_bytes_received", "per_rank_host_wait_s", "per_rank_stream_wait_ms", "bytes_received")}
    if result.get("detail"):
        line["detail"] = result["detail"]
    line["digest"] = result.get("digest")
    for drop in ("gather_stats", "attention_mfma", "torch_rocm_eager", "model_tflops", "detail"):      # never needed in practice
        if len(json.dumps(line)) <= COMPACT_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) > COMPACT_LIMIT:
        line["digest"] = {k: v for k, v in (line.get("digest") or {}).items() if k in ("value_k", "ms", "n_gpus", "dtype", "gemm_frac_of_peak")}
        line["config"]["workload"] = _short(line["config"].get("workload", ""), 80)
        line["config"]["gather"] = _short(line["config"].get("gather", ""), 60)
    return line


def flush_c_stdio():
    """Flush the C library's stdout / stderr buffers of this process (what native libraries printed so far)."""
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:                                                 # noqa: BLE001 -- never the reason a line is lost
        pass


def emit(result, detail_out):
    """Write the complete result to `detail_out` (one JSON document) and print the compact line as the ONLY stdout line."""
    if detail_out:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_out)), exist_ok=True)
            with open(detail_out, "w") as f:
                json.dump(result, f)
                f.write("\n")
            result["detail"] = os.path.relpath(detail_out, ROOT) if os.path.abspath(detail_out).startswith(ROOT) else detail_out
        except OSError as e:                                          # the detail file must not cost the line
            result["detail"] = f"not written ({type(e).__name__})"
    text = json.dumps(compact_line(result))
    assert len(text) <= COMPACT_LIMIT, len(text)
    print(text, flush=True)


def rccl_version():
    """RCCL's version as torch reports it (backend "nccl" IS RCCL on ROCm); never raises: an evidence field must not cost the line."""
    try:
        import torch
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:                                            # noqa: BLE001
        return f"unavailable ({type(e).__name__})"


def device_identity(rank, local_rank, dev_index):
    """Who computed: this rank's device as the runtime names it (UUID, PCI bus id), so that a first multi-GPU line PROVES
    N distinct GPUs took part.  No GPU call beyond property queries."""
    import torch
    ident = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_count": torch.cuda.device_count()}
    try:
        p = torch.cuda.get_device_properties(dev_index)
        ident["name"] = p.name
        for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
            v = getattr(p, key, None)
            if v is not None:
                ident[key] = str(v)
    except Exception as e:                                            # noqa: BLE001 -- an evidence field must not cost the line
        ident["name"] = f"properties unavailable ({type(e).__name__})"
    return ident


def per_kernel_bounds(timings, steps):
    """{kernel family: ms, its roofline time, the fraction reached and WHICH limit binds it}: every launch is priced at
    max(executed matrix flops / MFMA peak, algorithmic bytes / 6.3 TB/s, vector-ALU issue time) (csrc/common.h KernelTimer)."""
    per = {}
    for k, v in timings.items():
        if v["launches"]:
            terms = {"mfma": v["mfma_bound_ms"], "hbm": v["hbm_bound_ms"], "valu": v.get("valu_bound_ms", 0.0)}
            per[k] = {"ms_per_step": v["ms"] / steps, "bound_ms_per_step": v["bound_ms"] / steps,
                      "frac": v["bound_ms"] / v["ms"] if v["ms"] > 0 else 0.0,
                      "bound": max(terms, key=terms.get),
                      "bound_terms_ms_per_step": {a: b / steps for a, b in terms.items()}}
    return per


def mode_roofline(timings, mode, mult, steps):
    """Roofline of a 16-bit mode.  `frac` / `peak` keep the meaning they have on the f32 line and had in earlier rounds: the GEMM
    family's executed TFLOP/s against the 2.5 PFLOP/s dense MFMA peak.  Beside it every launch is priced against its OWN
    limit -- max(executed matrix flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s, vector-ALU issue time); out_proj, the
    LayerNorms and the softmax are HBM-bound in these modes, QKV / fc1 matrix-bound -- and summed per kernel family:
    `frac_of_own_bound` (GEMMs), `per_kernel`, `all_kernels` (the whole step)."""
    g = timings["gemm_f32"]
    per = per_kernel_bounds(timings, steps)
    tot_ms = sum(v["ms"] for v in timings.values())
    tot_bound = sum(v["bound_ms"] for v in timings.values())
    raw = mult * g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    lim = mult * g["flops"] / (g["bound_ms"] * 1e-3) / 1e12 if g["bound_ms"] > 0 else 0.0
    return {"bound": "mfma", "kernel": "16-bit Linear GEMMs (gemm16_q16s_kernel / gemm16_swp_kernel; nn.Linear, K2)",
            "achieved": raw, "peak": F16_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s executed by the GEMM launches" + (" (executed = 3 x algorithmic)" if mult == 3.0 else ""),
            "frac": raw / F16_MFMA_PEAK_TFLOPS,
            "frac_of_own_bound": raw / lim if lim > 0 else 0.0,
            "own_bound_tflops": lim,
            "own_bound_what": "the same flops with every launch at its own bound, max(flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s)",
            "gemm_ms_per_step": g["ms"] / steps, "gemm_bound_ms_per_step": g["bound_ms"] / steps,
            "gemm_mfma_bound_ms": g["mfma_bound_ms"] / steps, "gemm_hbm_bound_ms": g["hbm_bound_ms"] / steps,
            "gemm_algorithmic_tflops": raw / mult,
            "peaks": {"mfma_tflops": F16_MFMA_PEAK_TFLOPS, "hbm_tbps": 6.3, "valu_lane_ops_per_s": 1024 * 32 * 2.4e9},
            "all_kernels": {"ms_per_step": tot_ms / steps, "bound_ms_per_step": tot_bound / steps,
                            "frac": tot_bound / tot_ms if tot_ms > 0 else 0.0},
            "per_kernel": per, "mode": mode}


# --------------------------------------------------------------------------------------------- one rank
def run_rank(args) -> int:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (must precede HIP init)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device: there is no CPU path"
    dev_index = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # a collective that does not complete within this ends the rank (RCCL watchdog) instead of parking it until the
        # launcher's 25-minute deadline; the launcher then measures the compute-only curve
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("RNAMSM_BENCH_PG_TIMEOUT_S", "300")))
        # the default group is ALWAYS gloo: control plane (barriers, agreement, timing reductions) and the gather's second transport;
        # RCCL ("nccl") carries only the gathered output arrays, on a group of its own created in negotiate_gather_transport --
        # so an RCCL that cannot come up costs the device-to-device gather, never the run
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)

    def note_failure(text):
        path = os.environ.get("RNAMSM_BENCH_FAILURE_FILE")
        if path:
            try:
                with open(path, "a") as f:
                    f.write(text.replace("\n", " ")[:400] + "\n")
            except OSError:
                pass

    if os.environ.get("RNAMSM_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank that dies must fail the whole run
        raise SystemExit(f"rank {rank}: RNAMSM_BENCH_FAIL_RANK")
    from rnamsm import _lib, sharding, synthetic
    from rnamsm.model import MSATransformer
    lib = _lib.load()

    M, L = args.num_seqs, args.seq_len
    batch = args.workload == "configs3"
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model = model.eval().to(dev)
    model.gemm_dtype = args.gemm_dtype
    model.check_finite = False                    # a host sync per MSA; the bench checks finiteness once, below

    # ---- the step's work list: global MSA indices of this rank, tokens already resident in HBM (seed 1234 + index)
    if batch:
        per_step = args.num_msas                                  # strong scaling: the batch is fixed, ranks share it
        warm_items = min(per_step, 2 * world)                     # a warm-up "step" is a short pass (2 MSAs per rank)
    else:
        per_step = world                                          # weak scaling: one MSA per GPU per step
        warm_items = world

    def step_items(step_no):
        """(position within the step, index of the MSA / token set) of one step's items."""
        if batch:
            return [(i, i) for i in range(per_step)]
        return [(r, step_no * world + r) for r in range(world)]

    # token sets this rank will touch (item g = step * per_step + position belongs to rank g % world)
    mine_all = sorted({msa for k in range(args.warmup + args.steps) for pos, msa in step_items(k)
                       if sharding.owner_of((k if batch else 0) * per_step + pos, world) == rank})
    toks = {i: torch.from_numpy(synthetic.make_tokens(M, L, i)).to(dev) for i in mine_all}

    ident = device_identity(rank, local_rank, dev_index)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    gather = world > 1 and not args.no_gather
    gather_failure = None
    gather_group = None                                              # None = the default group (RCCL point-to-point under "nccl")
    gather_transport = args.backend
    if gather:
        # Probe the gather once, outside the timed region (this also builds the point-to-point communicators).  A fabric that refuses
        # it does not cost the gathered outputs: the ranks agree over a gloo control group and move the gather onto that group
        # (host-staged), in the same processes; only if that fails too the line is the sharded compute alone and says so.
        inject = (f"injected gather-probe failure on rank {rank} (RNAMSM_BENCH_FAIL_GATHER_PROBE)"
                  if os.environ.get("RNAMSM_BENCH_FAIL_GATHER_PROBE") in (str(rank), "all") else None)     # test hook
        if os.environ.get("RNAMSM_BENCH_FAIL_GATHER_PROBE") == "both":                                   # ... and no fallback either
            inject = f"injected gather-probe failure on rank {rank} (RNAMSM_BENCH_FAIL_GATHER_PROBE=both)"
        group, label, mine = sharding.negotiate_gather_transport(dev, want_backend=args.backend, inject_failure=inject,
                                                                 timeout=pg_timeout)
        if mine:
            note_failure(f"rank {rank} gather probe: {mine}")
        if group is not False and label != "primary" and os.environ.get("RNAMSM_BENCH_FAIL_GATHER_PROBE") == "both":
            group, label = False, "failed: " + label
        if group is False:
            gather = False
            gather_failure = label[len("failed: "):] if label.startswith("failed: ") else label
        else:
            gather_group = group
            if label != "primary":
                gather_transport, gather_failure = "gloo", label
    digest = torch.zeros((), dtype=torch.int64, device=dev)
    delivered = [0]
    last_gather = [None]

    def on_item(index, tensors):
        nonlocal digest
        delivered[0] += 1
        if args.digest:
            for t in tensors:
                digest = digest + (index + 1) * t.contiguous().view(torch.int32).to(torch.int64).sum()

    def run_steps(first_step, nsteps, items_of=None, use_gather=True):
        """`nsteps` steps: forward of this rank's MSAs, every output handed to ONE RoundGatherer that spans all the steps
        (global item index = step * items-per-step + position), so round k's transfers to rank 0 overlap the forwards of
        round k+1 across step boundaries as well; finish() drains the last round inside the timed region."""
        per = [items_of(first_step + k) if items_of else step_items(first_step + k) for k in range(nsteps)]
        n_items = sum(len(it) for it in per)
        g = None
        if (gather and use_gather) or (args.digest and world == 1):
            g = sharding.RoundGatherer(n_items, on_item=on_item, tensors_per_item=2, dst=0, device=dev, group=gather_group)
        base = 0
        for items in per:
            for pos, msa in items:
                if sharding.owner_of(base + pos, world) != rank:
                    continue
                out = model.forward_one(toks[msa], has_padding=False, need_repr=not args.outputs_only)
                if g is not None:
                    g.submit(base + pos, (out["emb"], out["atp"]))
            base += len(items)
        if g is not None:
            g.finish()
            last_gather[0] = g
        return n_items

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64)                 # the control group is gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- warm-up (untimed), then the headline loop: exactly K steps, timing hooks off
    if args.warmup:
        run_steps(0, args.warmup, items_of=(lambda k: step_items(k)[:warm_items]) if batch else None)
    sync_all()
    digest.zero_()
    delivered[0] = 0
    lib.rnamsm_timing_enable(0)
    t0 = time.perf_counter()
    msas_timed = run_steps(0 if batch else args.warmup, args.steps)
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    headline_digest = int(digest.item()) if args.digest else None
    headline_delivered = delivered[0]
    # ---- N > 1: what the gather cost.  Per rank: bytes received, host seconds blocked on transfers, milliseconds the compute
    # stream stood waiting for the transfer stream (RoundGatherer.stats); and the same steps once more WITHOUT the gather
    # (`compute_only_value`: max-over-ranks forward time only) so a first real multi-GPU run separates compute scaling from
    # the gather's exposed time
    gather_stats = compute_only = None
    if world > 1:
        st = last_gather[0].stats() if (gather and last_gather[0] is not None) else {}
        mine = [st.get("bytes_received", 0.0), st.get("host_wait_s", 0.0), st.get("stream_wait_ms", 0.0)]
        allst = [None] * world
        dist.all_gather_object(allst, mine)
        gather_stats = {"per_rank_bytes_received": [int(a[0]) for a in allst],
                        "per_rank_host_wait_s": [round(a[1], 4) for a in allst],
                        "per_rank_stream_wait_ms": [round(a[2], 3) for a in allst],
                        "what": "RoundGatherer.stats() of the timed region: bytes into each rank (rank 0 is the sink), host seconds "
                                "blocked in transfers, and ms the compute stream waited for the transfer stream (= the gather time "
                                "NOT overlapped with the next forward)"}
        sync_all()
        tc = time.perf_counter()
        run_steps(0 if batch else args.warmup, args.steps, use_gather=False)
        sync_all()
        compute_only = max_over_ranks(time.perf_counter() - tc)

    # ---- roofline pass: the same work with HIP-event pairs around every launch (no gather: kernels only)
    roof_steps = 1 if batch else args.steps
    roof_of = (lambda k: step_items(0)[:max(world, min(per_step, 8 * world))]) if batch else None
    sync_all()
    lib.rnamsm_timing_reset()
    lib.rnamsm_timing_enable(1)
    t1 = time.perf_counter()
    run_steps(0 if batch else args.warmup, roof_steps, items_of=roof_of, use_gather=False)
    sync_all()
    roof_elapsed = time.perf_counter() - t1
    lib.rnamsm_timing_enable(0)
    timings = _lib.kernel_timings()
    roof_local = max(1, sum(1 for k in range(roof_steps) for pos, _ in (roof_of(k) if roof_of else step_items(0))
                            if sharding.owner_of(pos, world) == rank))

    # ---- extra modes (N = 1, default workload): every contraction on the 16-bit matrix cores
    fast = bf16_mode = outputs_only = small_batches = None
    first = toks[mine_all[0]]
    if args.gemm_dtype == "f32" and not args.no_fast_mode and world == 1 and not batch:
        ref = model.forward_one(first)
        ref_emb, ref_atp = ref["emb"].clone(), ref["atp"].clone()
        # yardstick: the exact path against ITSELF when alignment rows 1.. are permuted (mathematically a no-op for row
        # 0's embedding and the tied maps; only fp32 summation order changes)
        perm = torch.cat([torch.zeros(1, dtype=torch.long),
                          1 + torch.randperm(M - 1, generator=torch.Generator().manual_seed(0))]).to(dev)
        per = model.forward_one(first[perm])
        noise_emb = float(((per["emb"] - ref_emb).double().norm() / ref_emb.double().norm()).item())
        noise_atp = float((per["atp"] - ref_atp).abs().max().item())

        def measure_mode(mode, mult):
            model.gemm_dtype = mode
            out = model.forward_one(first)
            dev_emb = float(((out["emb"] - ref_emb).double().norm() / ref_emb.double().norm()).item())
            dev_atp = float((out["atp"] - ref_atp).abs().max().item())
            dev_atp_mean = float((out["atp"] - ref_atp).abs().mean().item())
            for i in mine_all[:args.warmup]:
                model.forward_one(toks[i], has_padding=False)
            sync_all()
            t2 = time.perf_counter()
            for i in mine_all[args.warmup:]:
                model.forward_one(toks[i], has_padding=False)
            sync_all()
            el2 = time.perf_counter() - t2
            lib.rnamsm_timing_reset()
            lib.rnamsm_timing_enable(1)
            for i in mine_all[args.warmup:]:
                model.forward_one(toks[i], has_padding=False)
            sync_all()
            lib.rnamsm_timing_enable(0)
            tim2 = _lib.kernel_timings()
            model.gemm_dtype = "f32"
            return {"gemm_dtype": mode, "value": args.steps * M * L / el2, "unit": "MSA-residues/s",
                    "ms_per_step": 1e3 * el2 / args.steps,
                    "deviation_from_f32_path": {"emb_rel_l2": dev_emb, "atp_max_abs": dev_atp, "atp_mean_abs": dev_atp_mean},
                    "roofline": mode_roofline(tim2, mode, mult, args.steps),
                    "attention": "16-bit kernels (row_logits16 / row_apply16 / col_attn16, same operand format)",
                    "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in tim2.items()}}

        # the CLI's forward: only emb + atp are wanted, so the last layer skips the rows they do not depend on (bit-identical
        # outputs, checked here); never the headline unless --outputs-only is given
        lean = model.forward_one(first, has_padding=False, need_repr=False)
        lean_same = bool(torch.equal(lean["emb"], ref_emb) and torch.equal(lean["atp"], ref_atp))
        for i in mine_all[:args.warmup]:
            model.forward_one(toks[i], has_padding=False, need_repr=False)
        sync_all()
        t3 = time.perf_counter()
        for i in mine_all[args.warmup:]:
            model.forward_one(toks[i], has_padding=False, need_repr=False)
        sync_all()
        el3 = time.perf_counter() - t3
        outputs_only = {"value": args.steps * M * L / el3, "unit": "MSA-residues/s", "ms_per_step": 1e3 * el3 / args.steps,
                        "emb_and_atp_bit_identical_to_the_complete_forward": lean_same,
                        "what": "rnamsm_forward without RNAMSM_OUT_REPR (the CLI's call): after the last tied row attention only "
                                "alignment row 0 feeds emb, so the last column attention's queries/out_proj, the last FFN and the "
                                "final LayerNorm run on row 0's tokens only; K/V of that attention still cover every row"}
        # small alignments leave the chip idle one at a time (a forward costs >= 5.5 ms however few tokens): B same-shape MSAs
        # through rnamsm_forward_batch share every token-parallel launch (attention kernels batched over gridDim.y)
        import numpy as _np
        small_batches = {"what": "B same-shape MSAs one by one (rnamsm_forward each) vs one rnamsm_forward_batch call: ms per MSA "
                                 "and MSA-residues/s, exact path; outputs agree to fp32 rounding (tests/test_gpu_forward.py)", "cases": []}
        for sm, sl, sb in ((8, 64, 64), (16, 128, 16), (32, 128, 8)):
            st = torch.from_numpy(_np.stack([synthetic.make_tokens(sm, sl, 900 + b) for b in range(sb)])).to(dev)
            for _ in range(2):
                model.forward_batch(st, has_padding=False)
                model.forward_one(st[0], has_padding=False)
            best = {"one": 1e9, "batch": 1e9}
            for _ in range(3):
                sync_all(); t4 = time.perf_counter()
                for b in range(sb):
                    model.forward_one(st[b], has_padding=False)
                sync_all(); best["one"] = min(best["one"], (time.perf_counter() - t4) / sb)
                sync_all(); t4 = time.perf_counter()
                model.forward_batch(st, has_padding=False)
                sync_all(); best["batch"] = min(best["batch"], (time.perf_counter() - t4) / sb)
            small_batches["cases"].append({"num_seqs": sm, "seq_len": sl, "msas_per_batch": sb,
                                           "ms_per_msa_one_by_one": 1e3 * best["one"], "ms_per_msa_batched": 1e3 * best["batch"],
                                           "residues_per_s_one_by_one": sm * sl / best["one"],
                                           "residues_per_s_batched": sm * sl / best["batch"]})
        # unlike shapes (the reference's real workload: short RNAs of unlike length and depth, RNA_MSM_Inference.py:141-148):
        # 64 alignments of 4-24 rows x 41-121 columns one by one / padded into one frame (round 3) / token-packed (round 4)
        rng_p = _np.random.default_rng(0)
        pshapes = [(int(rng_p.integers(4, 25)), int(rng_p.integers(41, 122))) for _ in range(64)]
        pm = [torch.from_numpy(synthetic.make_tokens(r, c - 1, 700 + i)).to(dev) for i, (r, c) in enumerate(pshapes)]
        preal = sum(r * c for r, c in pshapes)

        def best_of(fn, reps=3):
            fn(); sync_all()
            b = 1e9
            for _ in range(reps):
                sync_all(); t5 = time.perf_counter(); fn(); sync_all(); b = min(b, time.perf_counter() - t5)
            return b
        t_one = best_of(lambda: [model.forward_one(t, has_padding=False, need_repr=False) for t in pm], reps=2)
        t_pk = best_of(lambda: model.forward_packed(pm))
        small_batches["unlike_shapes"] = {
            "what": "64 alignments of 4-24 rows x 41-121 columns (seeded), exact path: one rnamsm_forward each vs ONE rnamsm_forward_packed "
                    "call (token-packed: no padding; outputs equal alone to fp32 rounding, tests/test_gpu_forward.py)",
            "alignments": 64, "tokens": preal, "ms_one_by_one": 1e3 * t_one, "ms_packed": 1e3 * t_pk,
            "residues_per_s_one_by_one": preal / t_one, "residues_per_s_packed": preal / t_pk, "speedup": t_one / t_pk}
        fast = measure_mode("f16x3", 3.0)
        fast["f32_path_reordering_noise"] = {"emb_rel_l2": noise_emb, "atp_max_abs": noise_atp,
                                             "what": "exact path vs itself with alignment rows 1.. permuted"}
        bf16_mode = measure_mode("bf16", 1.0)
        bf16_mode["accuracy"] = ("tests/test_gpu_fullsize.py holds this mode to 1.5x the drift of the reference's own "
                                 ".bfloat16() arithmetic against an fp64 truth at every BASELINE size "
                                 "(profiles/r06_fullsize_parity.json)")

    # ---- every BASELINE config on the driver-run line (N = 1): one MSA of each shape, 1 warm-up + 2 timed forwards with the
    # hooks off, then one instrumented forward for the per-launch roofline sums
    per_config = None
    if args.gemm_dtype == "f32" and not args.no_per_config and world == 1 and not batch:
        import numpy as _np
        per_config = {"what": "one forward per BASELINE config (complete forward, emb + atp + representation), inputs resident in "
                              "HBM: ms = mean of 2 timed forwards after 1 warm-up; model_tflops = SURVEY 8d flops / time; "
                              "kernel_frac_of_bounds = sum of every launch's own roofline time / sum of launch times"}

        def one_config(tokens, mode):
            model.gemm_dtype = mode
            cm, cl = int(tokens.shape[0]), int(tokens.shape[1])
            model.forward_one(tokens, has_padding=False)
            sync_all()
            tc0 = time.perf_counter()
            for _ in range(2):
                out_c = model.forward_one(tokens, has_padding=False)
            sync_all()
            sec = (time.perf_counter() - tc0) / 2
            ok = bool(torch.isfinite(out_c["emb"]).all()) and int(out_c["err"].item()) == 0
            lib.rnamsm_timing_reset()
            lib.rnamsm_timing_enable(1)
            model.forward_one(tokens, has_padding=False)
            sync_all()
            lib.rnamsm_timing_enable(0)
            tim = _lib.kernel_timings()
            model.gemm_dtype = "f32"
            del out_c
            tot = sum(v["ms"] for v in tim.values())
            return {"num_seqs": cm, "seq_len": cl, "dtype": mode, "ms_per_step": 1e3 * sec, "residues_per_s": cm * cl / sec,
                    "model_tflops": flops_per_msa(cm, cl) / sec / 1e12, "outputs_finite_and_err_word_clear": ok,
                    "kernel_ms": tot, "kernel_frac_of_bounds": sum(v["bound_ms"] for v in tim.values()) / tot if tot else 0.0,
                    "per_kernel": {k: {"ms": round(v["ms_per_step"], 4), "frac": round(v["frac"], 4), "bound": v["bound"]}
                                   for k, v in per_kernel_bounds(tim, 1).items()}}

        gold = os.path.join(ROOT, "tests", "golden", "tokens_2DRB_1_full.npz")
        try:
            t0c = _np.load(gold)["diversity_max_512"].astype(_np.int64)
            src0 = "the shipped 2DRB_1 alignment, 512 rows by diversity-max as the CLI selects them (tests/golden/tokens_2DRB_1_full.npz)"
        except Exception:                                              # noqa: BLE001 -- the fixture is optional here
            t0c = synthetic.make_tokens(512, 36, 0)
            src0 = "synthetic 512 x 36 (fixture not found)"
        cases = [("configs0_f32", torch.from_numpy(_np.ascontiguousarray(t0c)).to(dev), "f32", src0),
                 ("configs1_f32", torch.from_numpy(synthetic.make_tokens(64, 128, 1)).to(dev), "f32", "synthetic"),
                 ("configs3_f32", torch.from_numpy(synthetic.make_tokens(128, 256, 3)).to(dev), "f32", "synthetic, one MSA of the batch"),
                 ("configs4_bf16", torch.from_numpy(synthetic.make_tokens(1024, 1024, 4)).to(dev), "bf16", "synthetic"),
                 ("configs4_f16x3", None, "f16x3", "synthetic")]
        for name, tk, mode, src in cases:
            if tk is None:
                tk = cases[3][1]
            per_config[name] = dict(one_config(tk, mode), tokens=src)
        del cases, tk
        torch.cuda.empty_cache()

    probe_out = model.forward_one(first)
    finite = bool(torch.isfinite(probe_out["emb"]).all())
    err_word = int(probe_out["err"].item())       # 0: tokens in range (bit 0) and the folded LayerNorm's precondition held (bit 1)
    if rank == 0:
        residues = msas_timed * M * L
        g = timings["gemm_f32"]
        mult = {"f32": 1.0, "bf16": 1.0, "f16x3": 3.0}[args.gemm_dtype]
        peak = FP32_MFMA_PEAK_TFLOPS if args.gemm_dtype == "f32" else F16_MFMA_PEAK_TFLOPS
        gemm_kernel = {"f32": "gemm_f32_kernel (nn.Linear, K2)",
                       "bf16": "gemm16_q16s_kernel (QKV, fc1, fc2) + gemm16_swp_kernel<split 1, bf16> (out_proj) (nn.Linear, K2)",
                       "f16x3": "gemm16_swp_kernel<split 3, fp16> (nn.Linear, K2)"}[args.gemm_dtype]
        flop_unit = "TFLOP/s" if mult == 1.0 else "TFLOP/s (executed MFMA flops = 3 x algorithmic)"
        gemm_tflops = mult * g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        attn_ms = sum(timings[k]["ms"] for k in ("row_logits", "row_apply", "col_attn"))
        attn_fl = mult * sum(timings[k]["flops"] for k in ("row_logits", "row_apply", "col_attn"))
        kern_ms = sum(v["ms"] for v in timings.values())
        traffic, traffic_src = None, None
        if (M, L) == (256, 512):                   # the shapes the committed PMC passes were collected on
            traffic, traffic_src = {"f32": lambda: pmc_traffic_per_launch(),
                                    "bf16": lambda: pmc_traffic_per_launch("rnamsm::gemm16_", "bf16_"),
                                    "f16x3": lambda: pmc_traffic_per_launch("rnamsm::gemm16_", "f16x3_")}[args.gemm_dtype]()
        elif (M, L) == (1024, 1024) and args.gemm_dtype == "bf16":     # BASELINE configs[4]
            traffic, traffic_src = pmc_traffic_per_launch("rnamsm::gemm16_", "cfg4_bf16_")
        if world == 1:
            gather_note = "none (single GPU)"
        elif not gather:
            gather_note = ("disabled by flag" if gather_failure is None
                           else f"failed: {gather_failure} (in the probe before the timed region; this line is the sharded compute only)")
            if os.environ.get("RNAMSM_BENCH_GATHER_NOTE"):
                gather_note = f"disabled: {os.environ['RNAMSM_BENCH_GATHER_NOTE']}; sharded compute only"
        else:
            gather_note = ((f"{gather_failure} -- " if gather_failure else "")
                           + f"rnamsm.sharding.RoundGatherer: emb+atp of every MSA to rank 0 over "
                           f"{'RCCL point-to-point' if gather_transport == 'nccl' else gather_transport + ' (host-staged)'}, "
                           f"round k overlapping the forward of round k+1; {headline_delivered} MSAs delivered in the timed region")
            if os.environ.get("RNAMSM_BENCH_GATHER_NOTE"):
                gather_note = f"{os.environ['RNAMSM_BENCH_GATHER_NOTE']} -- " + gather_note
        base_cfg = "configs[3]" if batch else "configs[2]"
        if batch and (M, L) != (128, 256):
            base_cfg = "configs[3] at another MSA shape"
        elif not batch and (M, L) != (256, 512):         # --num-seqs / --seq-len given: name the BASELINE config of that shape
            base_cfg = {(64, 128): "configs[1] shape", (1024, 1024): "configs[4] shape"}.get((M, L), "custom shape")
        result = {
            "metric": f"MSA-residues/sec forward (emb+attn-map), M={M} L={L}",
            "value": residues / elapsed,
            "unit": "MSA-residues/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if batch else "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16x3 (fp16 hi/lo split, f32 accumulate)", "bf16": "bf16"}[args.gemm_dtype],
            "data": "synthetic",
            "config": {"workload": (f"BASELINE {base_cfg}: " + (f"batch of {per_step} synthetic MSAs" if batch else "synthetic MSA")
                                    + f" M={M} x L={L} (col 0 = <cls>), D=768 H=12 10 layers, "
                                    + ("dealt round-robin over the ranks, a step = one pass over the batch" if batch
                                       else "one MSA per GPU per step") + ", emb+atp outputs"
                                    + (", gathered to rank 0" if gather else "")),
                       "gather": gather_note,
                       "num_seqs": M, "seq_len": L, "msas_per_step": per_step, "sharding": f"independent MSAs over {world} rank(s)",
                       "world_size_initialised": dist.get_world_size() if world > 1 else 1,
                       "ranks_seen": [f"rank {i['rank']}: cuda:{i['device_index']} {i.get('name', '?')} uuid {i.get('uuid', '?')} "
                                      f"pci {i.get('pci_domain_id', '?')}:{i.get('pci_bus_id', '?')}:{i.get('pci_device_id', '?')} "
                                      f"({i['device_count']} visible)" for i in idents],
                       # (by UUID / PCI address; where the runtime exposes neither, by the device index the rank was given)
                       "distinct_devices": len({(i.get("uuid"), i.get("pci_domain_id"), i.get("pci_bus_id"), i.get("pci_device_id"))
                                                if (i.get("uuid") or i.get("pci_bus_id")) else ("index", i["device_index"])
                                                for i in idents}),
                       "rccl_version": rccl_version() if world > 1 and args.backend == "nccl" else None,
                       "backend": ("nccl (RCCL) for the gather, gloo control group" if args.backend == "nccl" else args.backend) if world > 1 else "none",
                       "devices": "all ranks on device 0 (test hook)" if args.one_device and world > 1 else "one per rank",
                       "warmup_note": (f"a warm-up step is a pass over the first {warm_items} MSAs of the batch" if batch else "full steps")},
            "compute_only_value": (residues / compute_only) if compute_only else None,
            "compute_only_ms_per_step": (1e3 * compute_only / args.steps) if compute_only else None,
            "gather_stats": gather_stats,
            "outputs_finite": finite, "err_word": err_word,
            "model_tflops": flops_per_msa(M, L) * msas_timed / elapsed / 1e12,
            "roofline": (dict(mode_roofline(timings, args.gemm_dtype, mult, roof_local), traffic=traffic,
                              traffic_unit="bytes/launch of the GEMM kernels (HBM-side, PMC)", traffic_source=traffic_src,
                              traffic_replayed=traffic is not None,
                              algorithmic_bytes_per_launch=g["bytes"] / max(1, g["launches"]))
                         if args.gemm_dtype != "f32" else None) or {"bound": "mfma", "kernel": gemm_kernel,
                         "achieved": gemm_tflops, "peak": peak, "unit": flop_unit,
                         "frac": gemm_tflops / peak,
                         "avg_launch_ms": g["ms"] / max(1, g["launches"]), "launches": g["launches"],
                         "flops_per_launch": g["flops"] / max(1, g["launches"]),
                         "traffic": traffic, "traffic_unit": "bytes/launch (HBM-side, PMC)",
                         "traffic_source": traffic_src,
                         # counters cannot be collected inside a timed run: the figure is the committed PMC passes' (same workload,
                         # same kernels, tools/prof_r06.sh), REPLAYED here -- not a measurement of this run
                         "traffic_replayed": traffic is not None,
                         "algorithmic_bytes_per_launch": g["bytes"] / max(1, g["launches"]),
                         "measured_in": f"separate pass after the headline loop, HIP-event pairs on the launch stream, "
                                        f"{roof_local} MSA(s) on rank 0 in {roof_elapsed:.3f} s",
                         "note": ("flops counted = 2MNK of the Linear layers only; these launches also carry the three "
                                  "LayerNorms of every layer (applied to the accumulators of the QKV / fc1 GEMMs, row statistics "
                                  "left by the out_proj / fc2 epilogues: no LayerNorm launches) -- un-fused the same kernel "
                                  "measures 0.900 and the step is 1.25 % slower (DESIGN 3.4b)")
                                 if (args.gemm_dtype == "f32" and M * L >= 4096) else None},
            "attention_mfma": {"kernels": "row_logits+row_apply+col_attn", "achieved": attn_fl / (attn_ms * 1e-3) / 1e12 if attn_ms else 0.0,
                               "peak": peak, "unit": flop_unit,
                               "frac": (attn_fl / (attn_ms * 1e-3) / 1e12 / peak) if attn_ms else 0.0},
            "kernel_ms_per_msa": {k: v["ms"] / roof_local for k, v in timings.items()},
            "per_kernel": per_kernel_bounds(timings, roof_local),
            "kernel_time_share_of_roofline_pass": kern_ms / (1e3 * roof_elapsed),
        }
        tag = {"f32": "", "bf16": "bf16_", "f16x3": "f16x3_"}[args.gemm_dtype] if (M, L) == (256, 512) else (
            "cfg4_bf16_" if (M, L) == (1024, 1024) and args.gemm_dtype == "bf16" else None)
        if tag is not None:
            result["gemm_traffic_by_kernel"] = pmc_traffic_by_kernel("rnamsm::gemm", tag)
        if args.digest:
            result["output_digest"] = {"value": headline_digest, "items": headline_delivered,
                                       "what": "sum over gathered outputs of (global item index + 1) * sum(int32 bit patterns), mod 2^64"}
        if args.outputs_only:
            result["config"]["workload"] += " -- OUTPUTS-ONLY forward (--outputs-only): the last layer computes alignment row 0 only"
        if per_config is not None:
            result["per_config"] = per_config
        if fast is not None:
            result["outputs_only_mode"] = outputs_only
            result["small_msa_batches"] = small_batches
            result["fast_mode"] = fast
            result["bf16_mode"] = bf16_mode
        if not args.no_cpu_baseline and world == 1:
            result["torch_rocm_eager"] = torch_rocm_eager(M, L, state, dev)
            result["cpu_baseline"] = cpu_baseline(M, L, state)
        result["digest"] = make_digest(result)
    # The JSON line must be the LAST line on stdout.  Native libraries write there through C stdio (RCCL's version banner: block-
    # buffered on a pipe, flushed at process exit -- seen AFTER the line on the first real RCCL refusal, round 6): every rank flushes
    # C stdio now, the ranks meet, rank 0 prints, and nothing is left in anybody's buffer to trail the line.
    flush_c_stdio()
    if world > 1:
        dist.barrier()
    if rank == 0:
        emit(result, args.detail_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        flush_c_stdio()
    return 0


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
