"""One-off: the reference's OWN fp32 arithmetic at M = L = 1024 -- the oracle in float32 on the GPU
box's host cores -- against the fp64 truth (the oracle in float64 on the device), so that tests/test_gpu_fullsize.py has
a real bar for configs[4]'s exact path instead of torch's device-fp32 error (5e-2).

Run on the GPU box, outside the test suite (the CPU forward needs minutes of all host cores):

    python tests/analysis/yardstick_m1024.py [--rows 1024 --cols 1024 --threads N]

Writes gpurun_out/yardstick_m{M}_l{L}.json; the committed copy lives in tests/golden/.  Also records a thread sweep of
one oracle layer at M=256 x L=512 (what bench.py's cpu_baseline should use on this host).
Test infrastructure: imports oracle/ as the checker only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=0, help="0 = fastest of the sweep")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--skip-hip", action="store_true")
    args = ap.parse_args()
    import torch
    import truth
    from oracle import msm_oracle as O
    from rnamsm import synthetic
    M, L = args.rows, args.cols
    out = {"shape": [M, L], "token_seed": args.seed, "weights": "rnamsm.synthetic.make_state_dict(seed=0)",
           "torch": torch.__version__, "logical_cpus": os.cpu_count()}
    cpu_params = truth.params(torch.float32, "cpu")

    # thread sweep on one layer of the bench shape (M=256 x L=512): informs bench.py's cpu_baseline too
    sweep = {}
    toks_b = torch.from_numpy(synthetic.make_tokens(256, 512, 0))
    with torch.no_grad():
        for n in (256, 128, 64, 32, 16):
            if n > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(n)
            if not sweep:
                O.forward(toks_b, cpu_params, layers_to_run=1, ffn_token_chunk=32768)     # warm-up: pool + pages
            t0 = time.perf_counter()
            O.forward(toks_b, cpu_params, layers_to_run=1, ffn_token_chunk=32768)
            sweep[n] = round(time.perf_counter() - t0, 3)
            print("sweep", n, sweep[n], flush=True)
    out["thread_sweep_s_one_layer_M256_L512"] = sweep
    best = args.threads or min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    out["threads"] = best

    toks = synthetic.make_tokens(M, L, args.seed)
    t0 = time.perf_counter()
    t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, "cuda:0")
    torch.cuda.synchronize()
    out["truth_seconds_device_fp64"] = round(time.perf_counter() - t0, 2)
    print("truth done", out["truth_seconds_device_fp64"], flush=True)
    t0 = time.perf_counter()
    c_emb, c_atp = truth.oracle_outputs(toks, torch.float32, "cpu")
    out["cpu_seconds"] = round(time.perf_counter() - t0, 2)
    out["cpu_residues_per_s"] = M * L / out["cpu_seconds"]
    out["oracle_cpu_f32"] = truth.errors(c_emb, c_atp, t_emb, t_atp)
    print("cpu done", out["cpu_seconds"], out["oracle_cpu_f32"], flush=True)
    out["oracle_dev_f32"] = truth.errors(*truth.oracle_outputs(toks, torch.float32, "cuda:0"), t_emb, t_atp)
    if not args.skip_hip:
        from rnamsm.model import MSATransformer
        m = MSATransformer(num_layers=10)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in truth.state().items()}, strict=True)
        m = m.eval().to("cuda:0")
        for mode in ("f32", "f16x3"):
            m.gemm_dtype = mode
            o = m.forward_one(torch.from_numpy(toks).to("cuda:0"))
            out[f"hip_{mode}"] = truth.errors(o["emb"], o["atp"], t_emb, t_atp)
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    path = os.path.join(dst, f"yardstick_m{M}_l{L}.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
