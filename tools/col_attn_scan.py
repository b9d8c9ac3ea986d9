import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rna-msm_amd"))
import torch
from rnamsm import ops
dev = torch.device("cuda:0"); H = 12; D = 768
def timeit(fn, n=6):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
for R, C in [(64, 2048), (128, 1024), (256, 512), (512, 256), (1024, 128), (256, 128), (256, 2048)]:
    qkv = torch.randn(R * C, 3 * D, device=dev) * 0.5
    ctx = torch.empty(R * C, D, device=dev)
    t = timeit(lambda: ops.col_attn(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], R, C, H, out=ctx))
    fl = 4.0 * C * H * R * R * 64
    nblk = C * H * ((R + 255) // 256)
    print(f"R={R:5d} C={C:5d} blocks={nblk:6d} {t:.3f} ms {fl/t/1e9:6.1f} TF  per-block {t*1e3/(nblk/256):.1f} us  ideal-mfma/blk {fl/nblk/4/4096*64/2.4e3:.1f} us")
