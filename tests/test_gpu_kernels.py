"""GPU parity tests, per kernel, through the C ABI (rnamsm.ops -> librnamsm_hip.so).

Checked against (a) the oracle on the same seeded inputs and (b) the committed golden fixtures produced by the
reference.  Tolerances: the kernels are exact-fp32 (v_mfma_f32_32x32x2_f32 == an fmaf chain) and differ from the
CPU only by summation order: rel-L2 <= 2e-5 per op, probabilities max-abs <= 2e-5; north_star's end-to-end bar is
1e-4 (tests/test_gpu_forward.py).
"""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2
from oracle import msm_oracle as O
from rnamsm import synthetic

pytestmark = pytest.mark.gpu

TOL_REL = 2e-5
TOL_PROB = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rnamsm import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(name, shape, scale=1.0):
    return torch.from_numpy((scale * synthetic.normal(name, 11, shape)).astype(np.float32))


@pytest.mark.parametrize("T,D", [(1, 128), (7, 768), (1000, 768), (4097, 128), (33, 1024)])
def test_layernorm(dev, T, D):
    from rnamsm import ops
    x, g, b = _rand("ln.x", (T, D), 3.0) + 0.5, 1 + 0.1 * _rand("ln.g", (D,)), 0.1 * _rand("ln.b", (D,))
    y = ops.layernorm(x.to(dev), g.to(dev), b.to(dev)).cpu()
    want = O.layer_norm(x.double(), g.double(), b.double())
    assert rel_l2(y, want) < 2e-6


@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (127, 128, 64), (128, 256, 768), (300, 768, 768), (1025, 2304, 768),
                                   (513, 768, 3072), (4100, 3072, 768)])
def test_gemm_plain(dev, M, N, K):
    from rnamsm import ops
    a, w, b = _rand("g.a", (M, K)), _rand("g.w", (N, K), 0.05), _rand("g.b", (N,), 0.1)
    y = ops.linear(a.to(dev), w.to(dev), b.to(dev)).cpu()
    want = a.double() @ w.double().t() + b.double()
    assert rel_l2(y, want) < 3e-6
    assert np.abs(y.numpy() - want.numpy()).max() < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("M,N,K,act,scale_cols", [(7, 128, 128, 0, 0), (300, 2304, 768, 0, 768), (1025, 3072, 768, 1, 0),
                                                  (4100, 768, 256, 1, 256), (129, 256, 1024, 0, 128)])
def test_layernorm_folded_into_the_gemm(dev, M, N, K, act, scale_cols):
    """K1 folded (include/rnamsm.h): rstd * (x Wg^T - mean c) + d  ==  Linear(LayerNorm(x)) -- against fp64, and as close
    to it as the separate LayerNorm -> GEMM launches are; with the row statistics taken from the partial sums
    (rnamsm_row_partials) and summed by the GEMM itself.  x carries a row offset of the size of its spread
    (|mean| ~ std), more than the residual stream's, so the mean * c cancellation is exercised."""
    from rnamsm import ops
    x = (_rand("lf.x", (M, K), 2.0) + _rand("lf.m", (M, 1), 2.0)).to(dev)
    w, b = _rand("lf.w", (N, K), 0.05).to(dev), _rand("lf.b", (N,), 0.1).to(dev)
    g, be = (1 + 0.1 * _rand("lf.g", (K,))).to(dev), (0.1 * _rand("lf.be", (K,))).to(dev)
    wg, c, d = ops.ln_fold_weights(w, b, g, be)
    assert torch.equal(wg, w * g)                                             # one fp32 rounding per element
    assert rel_l2(c.cpu(), (w * g).double().sum(1).cpu()) < 1e-7
    assert rel_l2(d.cpu(), (b.double() + w.double() @ be.double()).cpu()) < 1e-7
    xd = x.double()
    part = ops.row_partials(x)
    assert part.shape == (K // 32, M, 2)                                      # slab-major
    slabs = xd.view(M, K // 32, 32).transpose(0, 1)
    centred = ((slabs - slabs.mean(-1, keepdim=True)) ** 2).sum(-1)            # about the SLAB's mean (Chan-combined later)
    assert rel_l2(part[..., 0].cpu(), slabs.sum(-1).cpu()) < 1e-6 and rel_l2(part[..., 1].cpu(), centred.cpu()) < 1e-5
    ref = ops.linear(ops.layernorm(x, g, be), w, b, act=act, scale=0.125, scale_cols=scale_cols)
    want = O.layer_norm(xd, g.double(), be.double()) @ w.double().t() + b.double()
    want[:, :scale_cols] *= 0.125
    if act:
        want = O.gelu_erf(want)
    e_ref = rel_l2(ref.cpu(), want.cpu())
    st = ops.row_stats_from_partials(part, K)
    assert rel_l2(st[:, 0].cpu(), xd.mean(1).cpu()) < 1e-6
    assert rel_l2(st[:, 1].cpu(), torch.rsqrt(xd.var(1, unbiased=False) + 1e-5).cpu()) < 1e-6
    for stats in (st, None):                                                  # from the partial sums / summed in the GEMM
        y = ops.linear_lnfold(x, wg, c, d, stats, act=act, scale=0.125, scale_cols=scale_cols)
        e_fold = rel_l2(y.cpu(), want.cpu())
        assert e_fold < 3e-6 and e_fold < 2.0 * e_ref + 2e-7, (stats is None, e_fold, e_ref)
        assert float((y.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
        # bit-identical under either tile width and any block order (each element: same K order, same epilogue arithmetic)
        try:
            ops.set_param("gemm_tile", 2)
            assert torch.equal(ops.linear_lnfold(x, wg, c, d, stats, act=act, scale=0.125, scale_cols=scale_cols), y)
            ops.set_param("gemm_tile", 1)
            assert torch.equal(ops.linear_lnfold(x, wg, c, d, stats, act=act, scale=0.125, scale_cols=scale_cols), y)
        finally:
            ops.set_param("gemm_tile", 0)
        # a sub-range of output features (what the outputs-only forward does with k|v and q): same bits
        if N >= 256:
            sub = ops.linear_lnfold(x, wg[128:], c[128:], d[128:], stats, act=act, scale=0.125, scale_cols=max(0, scale_cols - 128))
            assert torch.equal(sub, y[:, 128:])


def test_folded_layernorm_statistics_are_robust_and_its_precondition_is_reported(dev):
    """(mean, rstd) come from per-slab centred sums combined as in Chan et al.: rows with a huge common offset still get
    the right variance (E[x^2] - mean^2 in fp32 would return noise for them).  What cannot be rescued is the fold's own
    subtraction of mean * c[n]: such rows set bit 1 of cond_flag so that the caller falls back to LayerNorm + Linear, and
    ordinary rows (|mean| up to ~30x the spread) leave it clear."""
    from rnamsm import ops
    M, K, N = 256, 768, 256
    base = _rand("rob.x", (M, K))
    w, b = _rand("rob.w", (N, K), 0.05).to(dev), _rand("rob.b", (N,), 0.1).to(dev)
    g, be = (1 + 0.1 * _rand("rob.g", (K,))).to(dev), (0.1 * _rand("rob.be", (K,))).to(dev)
    wg, c, d = ops.ln_fold_weights(w, b, g, be)
    for offset, flagged, tol in ((0.0, False, 3e-6), (10.0, False, 3e-5), (1000.0, True, None)):
        x = (base + offset).to(dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        part = ops.row_partials(x)
        st = ops.row_stats_from_partials(part, K, cond_flag=flag)
        # the statistics against fp64: fine even at offset 1000 (spread 1)
        xd = x.double()
        assert rel_l2(st[:, 1].cpu(), torch.rsqrt(xd.var(1, unbiased=False) + 1e-5).cpu()) < 1e-4, offset
        assert bool(flag.item() & 2) == flagged, (offset, int(flag.item()))
        y = ops.linear_lnfold(x, wg, c, d, st)
        if tol is not None:
            want = O.layer_norm(xd, g.double(), be.double()) @ w.double().t() + b.double()
            assert rel_l2(y.cpu(), want.cpu()) < tol, (offset, rel_l2(y.cpu(), want.cpu()))
        flag.zero_()
        ops.linear_lnfold(x, wg, c, d, None, cond_flag=flag)                   # the self-summing form reports it itself
        assert bool(flag.item() & 2) == flagged, (offset, int(flag.item()))
    # the forward hands its err word to the folded GEMMs: a healthy model leaves it 0 (checked on every bench / CLI run)


def test_folded_gemm_over_the_first_rows_of_a_longer_stream(dev):
    """The outputs-only forward runs its last q projection / FFN over alignment row 0 only: the first C rows of x, with
    the row sums of all T rows lying slab-major in one buffer (partials_ld = T).  Same bits as the full-length GEMM's
    first rows; a producer over the first rows writes into that layout too."""
    from rnamsm import ops
    T, Mq, D, N = 700, 140, 256, 384
    x = (_rand("ls.x", (T, D), 2.0) + 0.7).to(dev)
    w, b = _rand("ls.w", (N, D), 0.05).to(dev), _rand("ls.b", (N,), 0.1).to(dev)
    g, be = (1 + 0.1 * _rand("ls.g", (D,))).to(dev), (0.1 * _rand("ls.be", (D,))).to(dev)
    wg, c, d = ops.ln_fold_weights(w, b, g, be)
    part = ops.row_partials(x)                                             # [D/32, T, 2]
    st = ops.row_stats_from_partials(part, D)                              # [T, 2]
    full = ops.linear_lnfold(x, wg, c, d, st)
    head = ops.linear_lnfold(x[:Mq], wg, c, d, st)                         # statistics of all T rows, GEMM over the first Mq
    assert torch.equal(head, full[:Mq])
    assert torch.equal(ops.row_stats_from_partials(part, D, rows=Mq), st[:Mq])      # partials_ld = T > M = Mq
    # producer over the first Mq rows, writing into the T-row layout: only those rows' sums change
    from rnamsm import _lib
    a, wo, bo = _rand("ls.a", (Mq, D)).to(dev), _rand("ls.wo", (D, D), 0.05).to(dev), _rand("ls.bo", (D,), 0.1).to(dev)
    x2, part2 = x.clone(), part.clone()
    _lib.check(_lib.load().rnamsm_gemm_residual_stats(a.data_ptr(), D, wo.data_ptr(), bo.data_ptr(), x2.data_ptr(), D,
                                                      x2.data_ptr(), D, Mq, D, D, part2.data_ptr(), T, 0,
                                                      torch.cuda.current_stream().cuda_stream))
    assert torch.equal(x2[Mq:], x[Mq:]) and torch.equal(part2[:, Mq:], part[:, Mq:])
    assert rel_l2(part2[:, :Mq].cpu(), ops.row_partials(x2[:Mq].contiguous()).cpu()) < 1e-6
    assert rel_l2(x2[:Mq].cpu(), (x[:Mq].double() + a.double() @ wo.double().t() + bo.double()).cpu()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(12837, 768, 96), (7200, 2304, 64), (18432, 2304, 64), (8192, 768, 64), (66000, 128, 32)])
def test_mixed_tile_gemm_is_bit_identical_to_the_uniform_tilings(dev, M, N, K):
    """gemm_f32_mixed_kernel (round 5: whole rounds of 128 x 128 tiles, the tile positions of the last round as 128 x 64 halves;
    knob "gemm_tile" = 3 forces it wherever a launch has both) against the all-128 x 128 launch, for every epilogue form the
    forward uses: plain (+ GELU, + column scale), residual, residual + row sums, the folded LayerNorm (both stats forms), the
    per-row factor of ragged batches.  Shapes: a ragged last row panel; a panel count that is no multiple of 8 (the whole rounds
    must stay inside the panels every XCD owns); configs[0]'s QKV; one whole round exactly and no tail (falls back to full tiles);
    one column block."""
    from rnamsm import ops
    from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
    a, w, b = _rand("mx.a", (M, K)).to(dev), _rand("mx.w", (N, K), 0.05).to(dev), _rand("mx.b", (N,), 0.1).to(dev)
    gamma, beta = (_rand("mx.g", (K,), 0.2) + 1.0).to(dev), _rand("mx.be", (K,), 0.1).to(dev)
    wg, c, d = ops.ln_fold_weights(w, b, gamma, beta)
    res = None
    if N == K:
        res = _rand("mx.r", (M, N)).to(dev)
    rowf = torch.rand(M, device=dev)
    outs = {}
    try:
        for tile in (1, 3, 0):
            ops.set_param("gemm_tile", tile)
            o = [ops.linear(a, w, b), ops.linear(a, w, b, act=ACT_GELU_ERF, scale=0.125, scale_cols=min(N, 128)),
                 ops.linear_row_scaled(a, w, b, rowf, scale=0.125, scale_cols=min(N, 128)),
                 ops.linear_lnfold(a, wg, c, d, None, act=ACT_GELU_ERF), ops.linear_lnfold(a, wg, c, d, None, scale=0.125, scale_cols=min(N, 128))]
            sq = _rand("mx.sq", (M, N)).to(dev)                              # residual forms: A = [M, N] @ W2 [N, N] needs N == K; use a square weight
            w2 = _rand("mx.w2", (N, N), 0.03).to(dev)
            if N <= 768:
                x = sq.clone()
                out, part = ops.linear_residual_stats(sq, w2, b, x, out=x)
                st = ops.row_stats_from_partials(part, N)
                o += [out.clone(), part.clone(), ops.linear(sq, w2, b, residual=sq), ops.linear_lnfold(sq, *ops.ln_fold_weights(w2, b, torch.ones(N, device=dev), torch.zeros(N, device=dev)), st)]
            outs[tile] = o
    finally:
        ops.set_param("gemm_tile", 0)
    for tile in (3, 0):
        for i, (x, y) in enumerate(zip(outs[1], outs[tile])):
            assert torch.equal(x, y), (tile, i, float((x - y).abs().max()))
    want = a.double() @ w.double().t() + b.double()
    assert rel_l2(outs[3][0].cpu(), want.cpu()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(5, 128, 64), (300, 768, 768), (1025, 768, 3072), (200, 256, 96)])
def test_residual_gemm_leaves_the_row_sums_of_what_it_stores(dev, M, N, K):
    """rnamsm_gemm_residual_stats (out_proj / fc2 + residual add): the output is bit-identical to the plain residual GEMM's
    and row_partials [N/32, M, 2] are the (sum, sum of squares) of the stored values per 32-column slab -- under both tile
    widths, in place (Cout = residual) as the forward runs it."""
    from rnamsm import ops
    a, w, b = _rand("rs.a", (M, K)).to(dev), _rand("rs.w", (N, K), 0.05).to(dev), _rand("rs.b", (N,), 0.1).to(dev)
    res = (_rand("rs.r", (M, N), 2.0) + 1.5).to(dev)
    plain = ops.linear(a, w, b, residual=res)
    for tile in (1, 2, 0):
        try:
            ops.set_param("gemm_tile", tile)
            x = res.clone()
            out, part = ops.linear_residual_stats(a, w, b, x, out=x)
        finally:
            ops.set_param("gemm_tile", 0)
        assert out.data_ptr() == x.data_ptr() and torch.equal(out, plain)
        slabs = out.double().view(M, N // 32, 32).transpose(0, 1)
        assert part.shape == (N // 32, M, 2)
        assert rel_l2(part[..., 0].cpu(), slabs.sum(-1).cpu()) < 1e-6
        assert rel_l2(part[..., 1].cpu(), ((slabs - slabs.mean(-1, keepdim=True)) ** 2).sum(-1).cpu()) < 1e-5
        # the slab sums are those of rnamsm_row_partials on the stored tensor, up to the order of 32 additions
        again = ops.row_partials(out)
        assert rel_l2(part.cpu(), again.cpu()) < 1e-6


def test_gemm_is_exact_on_integers(dev):
    """Exact-integer data: every product and partial sum is representable, so the MFMA path must be bit-exact, and an
    asymmetric W catches a transposed or permuted tile (cdna_hip_programming.md §3)."""
    from rnamsm import ops
    M, N, K = 200, 256, 96
    a = torch.from_numpy(((np.arange(M * K).reshape(M, K) * 7 + 3) % 13 - 6).astype(np.float32))
    w = torch.from_numpy(((np.arange(N * K).reshape(N, K) * 5 + 1) % 11 - 5).astype(np.float32))
    y = ops.linear(a.to(dev), w.to(dev)).cpu()
    assert torch.equal(y, a @ w.t())


def test_gemm_epilogues(dev):
    from rnamsm import ops
    from rnamsm._lib import ACT_GELU_ERF
    M, N, K = 333, 384, 128
    a, w, b, r = _rand("e.a", (M, K)), _rand("e.w", (N, K), 0.1), _rand("e.b", (N,), 0.2), _rand("e.r", (M, N))
    ad, wd, bd, rd = (t.double() for t in (a, w, b, r))
    base = ad @ wd.t() + bd
    y = ops.linear(a.to(dev), w.to(dev), b.to(dev), scale=0.125, scale_cols=128).cpu()
    want = base.clone(); want[:, :128] *= 0.125
    assert rel_l2(y, want) < 3e-6
    y = ops.linear(a.to(dev), w.to(dev), b.to(dev), act=ACT_GELU_ERF).cpu()
    assert rel_l2(y, O.gelu_erf(base)) < 3e-6
    y = ops.linear(a.to(dev), w.to(dev), b.to(dev), residual=r.to(dev)).cpu()
    assert rel_l2(y, base + rd) < 3e-6
    # in-place residual (x = x + f(x)) as the forward driver uses it
    xr = r.to(dev).clone()
    ops.linear(a.to(dev), w.to(dev), b.to(dev), residual=xr, out=xr)
    assert rel_l2(xr.cpu(), base + rd) < 3e-6
    # strided A / out views (the q|k|v activation is addressed in place)
    big = torch.zeros(M, 3 * N, device=dev)
    ops.linear(a.to(dev), w.to(dev), b.to(dev), out=big[:, N:2 * N])
    assert rel_l2(big[:, N:2 * N].cpu(), base) < 3e-6 and float(big[:, :N].abs().max()) == 0.0


def test_gemm_rejects_bad_shapes(dev):
    from rnamsm import ops, _lib
    with pytest.raises(_lib.RnamsmError):
        ops.linear(torch.zeros(4, 32, device=dev), torch.zeros(100, 32, device=dev))
    with pytest.raises(_lib.RnamsmError):
        ops.linear(torch.zeros(4, 32), torch.zeros(128, 32))        # CPU tensors: no fallback


ATT_SHAPES = [(1, 5, 2), (7, 33, 2), (34, 66, 2), (6, 19, 12), (3, 130, 2), (65, 40, 2), (257, 9, 2), (300, 20, 1)]


def _qkv(R, C, H, tag):
    D = 64 * H
    t = _rand(f"qkv.{tag}", (R * C, 3 * D))
    return t, D


@pytest.mark.parametrize("R,C,H", ATT_SHAPES)
def test_row_attention_kernels(dev, R, C, H):
    from rnamsm import ops
    qkv, D = _qkv(R, C, H, f"row{R}_{C}")
    g = qkv.to(dev)
    scaling = ops.row_scaling(R)
    q = (qkv[:, :D].double() * scaling).view(R, C, H, 64)
    k = qkv[:, D:2 * D].double().view(R, C, H, 64)
    v = qkv[:, 2 * D:].double().view(R, C, H, 64)
    gq = (g[:, :D] * scaling).contiguous()
    partial, nsplit = ops.row_logits(gq, g[:, D:2 * D].contiguous(), R, C, H)
    logits = torch.einsum("rihd,rjhd->hij", q, k)
    assert rel_l2(partial.sum(0).cpu(), logits) < 5e-6
    probs = ops.softmax_rows(partial)
    want_p = torch.softmax(logits, -1)
    assert np.abs(probs.cpu().numpy() - want_p.numpy()).max() < TOL_PROB
    assert np.abs(probs.sum(-1).cpu().numpy() - 1).max() < 1e-5
    # strided q/k/v views into the fused activation
    partial2, _ = ops.row_logits(g[:, :D], g[:, D:2 * D], R, C, H)
    assert rel_l2(partial2.sum(0).cpu() * scaling, logits) < 5e-6
    ctx = ops.row_apply(probs, g[:, 2 * D:], R, C, H).cpu()
    want = torch.einsum("hij,rjhd->rihd", probs.cpu().double(), v).reshape(R * C, D)
    assert rel_l2(ctx, want) < 5e-6


@pytest.mark.parametrize("R,C,H", [(512, 36, 12), (7, 8, 3), (33, 32, 12), (40, 33, 4), (64, 64, 12), (13, 63, 5), (1, 40, 2), (300, 17, 12),
                                   (9, 1, 2), (70, 57, 12)])
def test_narrow_row_attention_kernels_equal_the_tile_kernels_bit_for_bit(dev, R, C, H):
    """K4 / K6 at C <= 64 (row_logits_narrow_kernel, row_apply_narrow_kernel: no LDS tile, operands straight into the MFMA registers)
    against the 128 x 128 tile kernels they replace there (knob "row_narrow" = 0): the same slabs and the same K order, so the
    partial slabs and the context are the SAME BITS -- an alignment's maps must not depend on which kernel a batch routed it to --
    and both against fp64.  Shapes: the shipped example's 512 x 36; one quadrant (C <= 32) and four; C = 64 exactly; key counts that
    are no multiple of 8 (zero-padded last group) and of 4 (maps not 16-byte aligned); one row; more rows than one block's chunk;
    also the reference's row-chunked slabs (rnamsm_row_logits_chunked)."""
    from rnamsm import ops
    qkv, D = _qkv(R, C, H, f"narrow{R}_{C}")
    g = qkv.to(dev)
    q = qkv[:, :D].double().view(R, C, H, 64)
    k = qkv[:, D:2 * D].double().view(R, C, H, 64)
    v = qkv[:, 2 * D:].double().view(R, C, H, 64)
    outs = {}
    try:
        for narrow in (1, 0):
            ops.set_param("row_narrow", narrow)
            partial, nsplit = ops.row_logits(g[:, :D], g[:, D:2 * D], R, C, H)
            probs = ops.softmax_rows(partial, logit_scale=ops.depth_scaling(R))
            ctx = ops.row_apply(probs, g[:, 2 * D:], R, C, H)
            chunked = ops.row_logits(g[:, :D], g[:, D:2 * D], R, C, H, rows_per_chunk=max(1, R // 3))[0] if R >= 3 else None
            outs[narrow] = (partial.clone(), ctx.clone(), None if chunked is None else chunked.clone())
    finally:
        ops.set_param("row_narrow", 1)
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    if outs[1][2] is not None:
        assert torch.equal(outs[1][2], outs[0][2])
    logits = torch.einsum("rihd,rjhd->hij", q, k)
    assert rel_l2(outs[1][0].sum(0).cpu(), logits) < 5e-6
    p = torch.softmax(outs[1][0].sum(0).double().cpu() * ops.depth_scaling(R), -1)
    want = torch.einsum("hij,rjhd->rihd", p, v).reshape(R * C, D)
    assert rel_l2(outs[1][1].cpu(), want) < 5e-6
    # padding inside the fused activation's neighbourhood must not leak in: a non-finite key row right after the alignment's last token
    # of a row (the zero-padded last key group reads it) leaves the context untouched
    if C % 8 and R > 1:
        g2 = g.clone()
        probs = ops.softmax_rows(outs[1][0], logit_scale=ops.depth_scaling(R))
        base = ops.row_apply(probs, g2[:, 2 * D:], R, C, H).clone()
        g2[C:2 * C, 2 * D:] = float("inf")                                # row 1's V: row 0's padded key slots read into it
        got = ops.row_apply(probs, g2[:, 2 * D:], R, C, H)
        assert torch.equal(got[:C], base[:C])


@pytest.mark.parametrize("R,C,H", ATT_SHAPES)
def test_col_attention_kernel(dev, R, C, H):
    from rnamsm import ops
    qkv, D = _qkv(R, C, H, f"col{R}_{C}")
    g = qkv.to(dev)
    q = (qkv[:, :D].double() * 0.125).view(R, C, H, 64)
    k = qkv[:, D:2 * D].double().view(R, C, H, 64)
    v = qkv[:, 2 * D:].double().view(R, C, H, 64)
    gq = g.clone(); gq[:, :D] *= 0.125
    ctx = ops.col_attn(gq[:, :D], gq[:, D:2 * D], gq[:, 2 * D:], R, C, H).cpu()
    p = torch.softmax(torch.einsum("ichd,jchd->hcij", q, k), -1)
    want = torch.einsum("hcij,jchd->ichd", p, v).reshape(R * C, D)
    assert rel_l2(ctx, want) < 5e-6
    assert np.abs(ctx.numpy() - want.numpy()).max() < 2e-5 * max(1.0, float(want.abs().max()))
    # the probabilities on request (rnamsm_col_attn_probs: the reference's second return value, modules.py:917), without and
    # with a padding mask (padded keys get -10000 before the softmax, modules.py:911-915), fp32 and plane operands
    if R * R * C * H <= 4_000_000:
        got = ops.col_attn_probs(gq[:, :D], gq[:, D:2 * D], R, C, H).cpu()
        assert got.shape == (H, C, R, R) and np.abs(got.numpy() - p.numpy()).max() < 2e-6
        pad = torch.from_numpy(synthetic.normal(f"colpad{R}_{C}", 3, (R, C)) > 0.8)
        pad[0] = False                                               # keep one real key per column
        w = torch.einsum("ichd,jchd->hcij", q, k).masked_fill(pad.t()[None, :, None, :], -10000)
        got = ops.col_attn_probs(gq[:, :D], gq[:, D:2 * D], R, C, H, pad_mask=pad.to(torch.uint8).view(-1).to(dev)).cpu()
        assert np.abs(got.numpy() - torch.softmax(w, -1).numpy()).max() < 2e-6
        hi, lo = ops.split_bf16(g[:, :2 * D].contiguous(), fmt=1)       # unscaled q | k as fp16 hi/lo planes
        got = ops.col_attn_probs16((hi[:, :D], lo[:, :D]), (hi[:, D:], lo[:, D:]), R, C, H, fmt=1, scale=0.125).cpu()
        assert np.abs(got.numpy() - p.numpy()).max() < 5e-6


@pytest.mark.parametrize("R,C,H", [(1, 7, 3), (2, 64, 12), (8, 64, 12), (15, 33, 2), (16, 130, 4), (13, 5, 1), (17, 33, 2), (64, 128, 12)])
def test_col_attention_for_shallow_alignments_one_wave_per_problem(dev, R, C, H):
    """col_attn_small_kernel (R <= 16: one wave per (column, head), v_mfma_f32_16x16x4_f32, no LDS) against fp64 and against the
    128-query-block kernels it replaces (knob "col_small" = 0), without and with a padding mask (incl. a column whose keys are
    all padded: uniform weights, as the -10000 fill gives), R = 1 (ctx = v), and restricted to the first query rows
    (rnamsm_col_attn_fused_queries: bit-identical to the full launch; the two shapes above 16 rows hold that for the block kernels).
    (A one-wave-per-problem kernel for R = 17..64 on 32x32 tiles was built in round 6, correct, and not faster: EXPERIMENTS R6.6.)"""
    from rnamsm import ops
    qkv, D = _qkv(R, C, H, f"cs{R}_{C}")
    g = qkv.to(dev)
    g[:, :D] *= 0.125
    q = (qkv[:, :D].double() * 0.125).view(R, C, H, 64)
    k = qkv[:, D:2 * D].double().view(R, C, H, 64)
    v = qkv[:, 2 * D:].double().view(R, C, H, 64)
    pad = torch.from_numpy(synthetic.normal(f"cspad{R}_{C}", 3, (R, C)) > 0.6)
    pad[:, C // 2] = True                                          # a fully padded column
    for mask in (None, pad):
        w = torch.einsum("ichd,jchd->hcij", q, k)
        if mask is not None:
            w = w.masked_fill(mask.t()[None, :, None, :], -10000)
        want = torch.einsum("hcij,jchd->ichd", torch.softmax(w, -1), v).reshape(R * C, D)
        gm = None if mask is None or R == 1 else mask.to(torch.uint8).view(-1).to(dev)
        if mask is not None and R == 1:
            continue
        try:
            ops.set_param("col_small", 1)
            small = ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, pad_mask=gm).cpu()
            ops.set_param("col_small", 0)
            big = ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, pad_mask=gm).cpu()
        finally:
            ops.set_param("col_small", 1)
        assert rel_l2(small, want) < 5e-6, (R, C, H, mask is not None)
        assert np.abs(small.numpy() - want.numpy()).max() < 2e-5 * max(1.0, float(want.abs().max()))
        assert rel_l2(small, big) < 2e-6
        if R > 2:
            qr = max(1, R // 3)
            part = torch.full((R * C, D), float("nan"), device=dev)
            ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, pad_mask=gm, out=part, q_rows=qr)
            assert torch.equal(part[:qr * C].cpu(), small[:qr * C]) and bool(torch.isnan(part[qr * C:]).all())


@pytest.mark.parametrize("R,C,H", [(1, 7, 3), (8, 64, 12), (16, 130, 4), (17, 33, 2), (100, 30, 3), (129, 5, 1), (300, 4, 2)])
def test_col_attention_on_prescaled_q_without_a_running_maximum(dev, R, C, H):
    """rnamsm_col_attn_fused_prescaled (round 4; what the exact-path forward calls without padding): q carries dh^-1/2 * log2(e), the
    first pass exponentiates the raw scores (no running maximum, no rescale) and a block whose row sums leave [2^-64, 2^100]
    redoes its column with the online softmax.  Against fp64 on the prescaled operands at the bar of the natural-domain kernel,
    against that kernel on q / log2(e) (rounding apart), with the TRACKED loop forced (knob "col_fast" = 0), restricted to the
    first query rows (bit-identical to the full launch), and with scores far outside exp2's range in both directions: a query
    whose best key sits ~200 log2 units up (overflow without a reference) and one whose every score is below -130 (all
    exponentials underflow: a zero row sum) -- finite and right through the fallback."""
    from rnamsm import ops
    LOG2E = 1.4426950408889634
    qkv, D = _qkv(R, C, H, f"cp{R}_{C}")
    g = qkv.to(dev)
    g[:, :D] *= 0.125 * LOG2E

    def truth(t):
        q = t[:, :D].double().view(R, C, H, 64) / LOG2E
        k = t[:, D:2 * D].double().view(R, C, H, 64)
        v = t[:, 2 * D:].double().view(R, C, H, 64)
        return torch.einsum("hcij,jchd->ichd", torch.softmax(torch.einsum("ichd,jchd->hcij", q, k), -1), v).reshape(R * C, D)
    want = truth(g.cpu())
    fast = ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, prescaled=True)
    assert rel_l2(fast.cpu(), want) < 5e-6
    assert np.abs(fast.cpu().numpy() - want.numpy()).max() < 2e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(fast, ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, prescaled=True))
    try:
        ops.set_param("col_fast", 0)
        tracked = ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, prescaled=True)
    finally:
        ops.set_param("col_fast", 1)
    assert rel_l2(tracked.cpu(), want) < 5e-6 and rel_l2(fast.cpu(), tracked.cpu()) < 2e-6
    nat = g.clone()
    nat[:, :D] /= LOG2E
    assert rel_l2(fast.cpu(), ops.col_attn(nat[:, :D], nat[:, D:2 * D], nat[:, 2 * D:], R, C, H).cpu()) < 2e-6
    qr = max(1, R // 3)
    part = torch.full((R * C, D), float("nan"), device=dev)
    ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, out=part, prescaled=True, q_rows=qr)
    assert torch.equal(part[:qr * C], fast[:qr * C]) and bool(torch.isnan(part[qr * C:]).all())
    if R >= 8:
        wild = g.clone()
        qv, kv = wild[:, :D].view(R, C, H, 64), wild[:, D:2 * D].view(R, C, H, 64)
        qv[5, C // 2, H - 1] = 26.0 * kv[7, C // 2, H - 1] / kv[7, C // 2, H - 1].norm()       # q . k_7 = 26 |k_7| ~ 208 log2 units
        u = torch.zeros(64, device=dev)
        u[0] = 1.0
        kv[:, 0, 0] += 8.0 * u                                                                     # the keys of (column 0, head 0) share a component ...
        qv[3, 0, 0] = -30.0 * u                                                                    # ... against which this query scores ~ -240 on every key
        sc = torch.einsum("ichd,jchd->hcij", qv.double(), kv.double())
        assert float(sc.max()) > 128 and float(sc[0, 0, 3].max()) < -130                  # beyond exp2 in fp32 on both sides
        got = ops.col_attn(wild[:, :D], wild[:, D:2 * D], wild[:, 2 * D:], R, C, H, prescaled=True)
        assert bool(torch.isfinite(got).all())
        assert rel_l2(got.cpu(), truth(wild.cpu())) < 5e-6


@pytest.mark.parametrize("R,C,H", [(512, 36, 12), (300, 100, 4), (129, 7, 2)])
def test_col_attention_fallback_recomputes_only_the_waves_that_asked(dev, R, C, H):
    """Round 5: when the FAST loop of the prescaled fp32 column kernel leaves exp2's range for some query, the block makes its second
    (TRACKED) pass together -- it shares the key ring -- but only the WAVES that asked for it recompute; the other waves of the block
    keep their FAST result.  So a query row's arithmetic depends on its own wave's 32 rows only: every row outside the wild
    queries' waves comes out with the same bits as without the wild queries, and everything is finite and right against fp64."""
    from rnamsm import ops
    LOG2E = 1.4426950408889634
    qkv, D = _qkv(R, C, H, f"cm{R}_{C}")
    g = qkv.to(dev)
    g[:, :D] *= 0.125 * LOG2E
    wild = g.clone()
    qv, kv = wild[:, :D].view(R, C, H, 64), wild[:, D:2 * D].view(R, C, H, 64)
    qv[5, C // 2, H - 1] = 26.0 * kv[7, C // 2, H - 1] / kv[7, C // 2, H - 1].norm()          # overflow without a reference: wave 0 of block 0
    qv[R - 3, 1, 0] = 26.0 * kv[2, 1, 0] / kv[2, 1, 0].norm()                                    # ... and a wave of the LAST block of another column
    base = ops.col_attn(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], R, C, H, prescaled=True).clone()
    got = ops.col_attn(wild[:, :D], wild[:, D:2 * D], wild[:, 2 * D:], R, C, H, prescaled=True).clone()
    assert bool(torch.isfinite(got).all())
    same = base.view(R, C, H, 64) == got.view(R, C, H, 64)
    same[0:32, C // 2, H - 1] = True
    same[(R - 3) // 32 * 32:(R - 3) // 32 * 32 + 32, 1, 0] = True
    assert bool(same.all())
    cols = sorted({0, 1, C // 2, C - 1})

    def truth(t):
        q = t[:, :D].double().view(R, C, H, 64)[:, cols] / LOG2E
        k = t[:, D:2 * D].double().view(R, C, H, 64)[:, cols]
        v = t[:, 2 * D:].double().view(R, C, H, 64)[:, cols]
        return torch.einsum("hcij,jchd->ichd", torch.softmax(torch.einsum("ichd,jchd->hcij", q, k), -1), v)
    for t, o in ((g, base), (wild, got)):
        assert rel_l2(o.view(R, C, H, 64)[:, cols].cpu(), truth(t).cpu()) < 5e-6


def test_col_attention_online_softmax_rescale_is_exercised(dev):
    """Forces the running-max rescale branch: one late key dominates every query (spike placed in the last
    64-key chunk), and a second case puts the dominant key first so later tiles never rescale."""
    from rnamsm import ops
    R, C, H = 200, 3, 1
    for spike_row in (190, 0, 65):
        qkv = _rand(f"spike{spike_row}", (R * C, 192))
        qkv[:, :64] = qkv[:, :64].abs() * 0.2
        k = qkv[:, 64:128].view(R, C, 64)
        k[spike_row] = 6.0                                   # q.k ~ +60 for that key: exp underflows elsewhere
        g = qkv.to(dev)
        ctx = ops.col_attn(g[:, :64], g[:, 64:128], g[:, 128:], R, C, H).cpu()
        q = qkv[:, :64].double().view(R, C, 1, 64); kk = qkv[:, 64:128].double().view(R, C, 1, 64)
        v = qkv[:, 128:].double().view(R, C, 1, 64)
        p = torch.softmax(torch.einsum("ichd,jchd->hcij", q, kk), -1)
        want = torch.einsum("hcij,jchd->ichd", p, v).reshape(R * C, 64)
        assert rel_l2(ctx, want) < 5e-6


@pytest.mark.parametrize("R,C", [(3, 9), (16, 33), (5, 200), (40, 130), (3, 600), (9, 8)])
def test_embed_ln_matches_oracle_with_and_without_pad(dev, R, C):
    from rnamsm import ops
    state = synthetic.make_state_dict(seed=3, embed_dim=128, num_layers=1, num_heads=2)
    params = O.to_torch_params(state)
    toks = torch.from_numpy(synthetic.make_tokens(R, C, 5))
    args = [params[k].to(dev) for k in ("embed_tokens.weight", "embed_positions.weight")]
    rowpos = params["msa_position_embedding"].view(-1).to(dev)
    lnw, lnb = params["emb_layer_norm_before.weight"].to(dev), params["emb_layer_norm_before.bias"].to(dev)
    x = ops.embed_ln(toks.to(dev), args[0], args[1], rowpos, lnw, lnb, pad_idx=1).cpu()
    assert rel_l2(x, O.embed(toks, params).reshape(R * C, -1)) < 2e-6
    # positions must skip <pad> exactly like cumsum(mask)*mask + pad (modules.py:288-290), integer-exact
    toks[1, 2] = 1; toks[R - 1, C - 1] = 1; toks[2, 0] = 1
    x = ops.embed_ln(toks.to(dev), args[0], args[1], rowpos, lnw, lnb, pad_idx=1).cpu()
    assert rel_l2(x, O.embed(toks, params).reshape(R * C, -1)) < 2e-6
    # (round 5: a wave walks runs of 8 consecutive tokens and carries the position count) pads scattered over run starts, run
    # interiors and row starts
    mask = torch.from_numpy(synthetic.normal(f"embpad{R}_{C}", 3, (R, C)) > 0.6)
    toks2 = toks.clone(); toks2[mask] = 1
    x = ops.embed_ln(toks2.to(dev), args[0], args[1], rowpos, lnw, lnb, pad_idx=1).cpu()
    assert rel_l2(x, O.embed(toks2, params).reshape(R * C, -1)) < 2e-6
    bad = toks.clone(); bad[0, 1] = 12
    with pytest.raises(IndexError):
        ops.embed_ln(bad.to(dev), args[0], args[1], rowpos, lnw, lnb, pad_idx=1)


@pytest.mark.parametrize("C", [21, 150, 2, 66])
def test_pack_outputs_is_a_pure_copy(dev, C):
    from rnamsm import ops
    D, NL, H, R = 128, 3, 2, 4
    x = _rand("pk.x", (R * C, D)); p = _rand("pk.p", (NL, H, C, C))
    emb, atp = ops.pack_outputs(x.to(dev), p.to(dev), C)
    assert torch.equal(emb.cpu(), x.view(R, C, D)[0, 1:])
    assert torch.equal(atp.cpu(), p[..., 1:, 1:].reshape(NL * H, C - 1, C - 1))


OP_CASES = ["d128_r7_c33", "d128_r7_c33_chunk", "d128_r1_c5", "d128_r34_c66", "d768_r6_c19"]


@pytest.mark.parametrize("name", OP_CASES)
def test_modules_match_reference_fixtures(dev, name):
    """The reference-interface modules (rnamsm.modules) against outputs of the reference's own modules."""
    from rnamsm import modules as M
    g = golden(f"op_{name}.npz")
    D, H, R, C, max_tokens = (int(v) for v in g["meta"])
    mt = 2 ** 30 if max_tokens < 0 else max_tokens
    state = synthetic.make_state_dict(seed=7, embed_dim=D, num_layers=1, num_heads=H)
    x = torch.from_numpy(synthetic.normal(f"x:{name}", 7, (R, C, 1, D)).astype(np.float32)).to(dev)

    def load(mod, prefix):
        sd = {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in state.items() if k.startswith(prefix + ".")}
        mod.load_state_dict(sd, strict=True)
        return mod.eval().to(dev)

    row = load(M.RowSelfAttention(D, H, max_tokens_per_msa=mt), "layers.0.row_self_attention.layer")
    y, p = row(x)
    assert y.shape == (R, C, 1, D) and p.shape == (H, 1, C, C)
    assert rel_l2(y.cpu()[:, :, 0], g["row_out"]) < TOL_REL
    assert np.abs(p.cpu().numpy()[:, 0] - g["row_probs"]).max() < TOL_PROB
    col = load(M.ColumnSelfAttention(D, H, max_tokens_per_msa=mt), "layers.0.column_self_attention.layer")
    y, cp = col(x)
    assert rel_l2(y.cpu()[:, :, 0], g["col_out"]) < TOL_REL
    # the second return value of the reference (modules.py:917-945): [H, C, B=1, R, R], materialised by rnamsm_col_attn_probs
    assert cp.shape == (H, C, 1, R, R)
    if "col_probs" in g:
        assert np.abs(cp.cpu().numpy()[:, :, 0] - g["col_probs"]).max() < TOL_PROB
    col.return_probs = False                                     # what MSATransformer's layers run: no second launch
    y2, none = col(x)
    assert none is None and torch.equal(y2, y)
    col.return_probs = True
    for mode in ("f16x3", "bf16"):                               # the plane form of the same kernel
        col.gemm_dtype = mode
        _, cp16 = col(x)
        if "col_probs" in g:
            assert np.abs(cp16.cpu().numpy()[:, :, 0] - g["col_probs"]).max() < (TOL_PROB if mode == "f16x3" else 5e-2)
    col.gemm_dtype = "f32"
    if "ffn_out" in g:
        ff = load(M.FeedForwardNetwork(D, 4 * D, max_tokens_per_msa=mt), "layers.0.feed_forward_layer.layer")
        assert rel_l2(ff(x).cpu()[:, :, 0], g["ffn_out"]) < TOL_REL
        blk = load(M.NormalizedResidualBlock(M.FeedForwardNetwork(D, 4 * D), D), "layers.0.feed_forward_layer")
        assert rel_l2(blk(x).cpu()[:, :, 0], g["ffn_block_out"]) < TOL_REL
    layer = load(M.AxialTransformerLayer(D, 4 * D, H, max_tokens_per_msa=mt), "layers.0")
    y, cp, rp = layer(x, need_head_weights=True)
    assert rel_l2(y.cpu()[:, :, 0], g["layer_out"]) < TOL_REL
    assert np.abs(rp.cpu().numpy()[:, 0] - g["layer_row_probs"]).max() < TOL_PROB
    with pytest.raises(NotImplementedError):
        row(x, self_attn_mask=torch.zeros(1))


@pytest.mark.parametrize("name", ["t37_b3_e128", "t70_b2_e768"])
def test_generic_mha_matches_reference_fixture(dev, name):
    """the 1-D MHA entry point ([T,B,E] self-attention) against msm/multihead_attention.py's output."""
    from rnamsm import modules as M
    g = golden(f"mha_{name}.npz")
    T, B, E, H = (int(v) for v in g["meta"])
    state = synthetic.make_state_dict(seed=11, embed_dim=E, num_layers=1, num_heads=H)
    prefix = "layers.0.row_self_attention.layer"
    mha = M.MultiheadAttention(E, H, self_attention=True)
    mha.load_state_dict({k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in state.items() if k.startswith(prefix + ".")},
                        strict=True)
    mha = mha.eval().to(dev)
    x = torch.from_numpy(synthetic.normal(f"mha:{name}", 11, (T, B, E)).astype(np.float32)).to(dev)
    # the reference's defaults (need_weights=True): head-averaged probabilities next to the output
    y, w = mha(x, x, x)
    assert y.shape == (T, B, E) and w.shape == (B, T, T)
    assert rel_l2(y.cpu(), g["out"]) < TOL_REL
    assert np.abs(w.cpu().numpy() - g["avg_weights"]).max() < TOL_PROB
    # need_weights=False: the fused kernel, no probabilities
    y2, w2 = mha(x, x, x, need_weights=False)
    assert w2 is None and rel_l2(y2.cpu(), g["out"]) < TOL_REL
    # key_padding_mask + weights, as msm/modules.py:123-131 calls it (fixture from the reference with both set)
    gm = golden(f"mha_masks_{name}.npz")
    kpm = torch.from_numpy(gm["key_padding_mask"]).to(dev)
    assert rel_l2(y.cpu(), gm["out_default"]) < TOL_REL and np.abs(w.cpu().numpy() - gm["avg_weights_default"]).max() < TOL_PROB
    y3, w3 = mha(x, x, x, key_padding_mask=kpm, need_weights=True)
    assert rel_l2(y3.cpu(), gm["out_masked"]) < TOL_REL
    assert np.abs(w3.cpu().numpy() - gm["avg_weights_masked"]).max() < TOL_PROB
    y4, w4 = mha(x, x, x, key_padding_mask=kpm, need_head_weights=True)
    assert w4.shape == (H, B, T, T) and torch.equal(y4, y3)
    assert np.abs(w4.cpu().numpy() - gm["head_weights_masked"]).max() < TOL_PROB
    assert float(w4[:, 0, :, T - 5:].max()) == 0.0 and float(w4[:, B - 1, :, 3].max()) == 0.0
    y5, w5 = mha(x, x, x, key_padding_mask=kpm, need_weights=False)              # fused kernel with the key mask
    assert w5 is None and rel_l2(y5.cpu(), gm["out_masked"]) < TOL_REL
    # the same entry point in the fp32-grade 16-bit mode (f16x3: operands as fp16 hi/lo planes)
    mha.gemm_dtype = "f16x3"
    y6, w6 = mha(x, x, x, key_padding_mask=kpm, need_head_weights=True)
    assert rel_l2(y6.cpu(), gm["out_masked"]) < 2e-5 and np.abs(w6.cpu().numpy() - gm["head_weights_masked"]).max() < 2e-5
    y7, w7 = mha(x, x, x, key_padding_mask=kpm, need_weights=False)
    assert w7 is None and rel_l2(y7.cpu(), gm["out_masked"]) < 2e-5
    mha.gemm_dtype = "f32"
    y8, w8 = mha(x, x, x, attn_mask=torch.zeros(T, T, device=dev))               # a zero attn_mask changes nothing (weights route)
    assert rel_l2(y8.cpu(), g["out"]) < TOL_REL and np.abs(w8.cpu().numpy() - g["avg_weights"]).max() < TOL_PROB
    with pytest.raises(ValueError):
        mha(x, x, x, attn_mask=torch.zeros(T, T, dtype=torch.bool, device=dev))  # the reference ADDS the mask: float only
    with pytest.raises(ValueError):
        mha(x, x, x, key_padding_mask=torch.zeros(T, B, dtype=torch.bool, device=dev))


def test_generic_mha_with_attn_mask_matches_reference_fixture(dev):
    """msm/multihead_attention.py:353-357 (round 5): a float [T, T] attn_mask is added to the scores of every batch element and
    head -- here a causal mask (-inf above the diagonal) plus finite biases -- alone and with a key_padding_mask; fixture from
    the reference itself (tests/golden/make_golden_r5.py).  Queries left without any admissible key are NaN rows, as there."""
    from rnamsm import modules as M
    g = golden("mha_attn_mask.npz")
    T, B, E, H = (int(v) for v in g["meta"])
    state = synthetic.make_state_dict(seed=int(g["seed"]), embed_dim=E, num_layers=1, num_heads=H)
    prefix = "layers.0.row_self_attention.layer"
    mha = M.MultiheadAttention(E, H, self_attention=True)
    mha.load_state_dict({k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in state.items() if k.startswith(prefix + ".")}, strict=True)
    mha = mha.eval().to(dev)
    x = torch.from_numpy(synthetic.normal("mha:attn_mask", 13, (T, B, E)).astype(np.float32)).to(dev)
    am = torch.from_numpy(g["attn_mask"]).to(dev)
    kpm = torch.from_numpy(g["key_padding_mask"]).to(dev)
    for mode, tol, ptol in (("f32", TOL_REL, TOL_PROB), ("f16x3", 2e-5, 2e-5)):
        mha.gemm_dtype = mode
        y, w = mha(x, x, x, attn_mask=am)
        assert rel_l2(y.cpu(), g["out"]) < tol and np.abs(w.cpu().numpy() - g["avg_weights"]).max() < ptol, mode
        assert float(w[:, 0, 1:].abs().max()) == 0.0                           # query 0 attends to key 0 only
        yh, wh = mha(x, x, x, attn_mask=am, need_head_weights=True)
        assert torch.equal(yh, y) and np.abs(wh.cpu().numpy() - g["head_weights"]).max() < ptol, mode
        yn, wn = mha(x, x, x, attn_mask=am, need_weights=False)                    # no weights returned; same route, same bits
        assert wn is None and torch.equal(yn, y) and rel_l2(yn.cpu(), g["out_noweights"]) < max(tol, 1e-5), mode
        yk, wk = mha(x, x, x, attn_mask=am, key_padding_mask=kpm)
        want, want_w = g["out_kpm"], g["avg_weights_kpm"]
        live = np.isfinite(want)
        assert np.array_equal(np.isnan(yk.cpu().numpy()), ~live) and np.array_equal(np.isnan(wk.cpu().numpy()), np.isnan(want_w)), mode
        assert rel_l2(np.where(live, yk.cpu().numpy(), 0.0), np.where(live, want, 0.0)) < tol, mode
        assert np.nanmax(np.abs(wk.cpu().numpy() - want_w)) < ptol, mode
    mha.gemm_dtype = "f32"


def test_generic_mha_options_match_reference_fixture(dev):
    """VERDICT r05 "missing" 3: everything msm/multihead_attention.py:154-434 does outside plain self-attention -- cross-attention
    in both length orders, kdim / vdim, add_bias_kv + add_zero_attn, bias=False, incremental decoding through the caller's
    incremental_state (self-attention: the buffer grows; encoder-decoder: static_kv), before_softmax on the reference's own route,
    and a FINITE large-negative attn_mask together with key padding -- against outputs of the reference itself
    (tests/golden/mha_general.npz, make_golden_r6.py; weights and inputs redrawn by tests/mha_cases.py)."""
    import mha_cases as MC
    from rnamsm import modules as M
    g = golden("mha_general.npz")
    E, H = MC.E, MC.H
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def build(tag, wkw=None, **kw):
        m = M.MultiheadAttention(E, H, **kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in MC.weights(tag, **(wkw or {})).items()}, strict=True)
        return m.eval().to(dev)

    def close(a, b, tol=TOL_REL):
        a, b = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)), (a.shape, b.shape)
        return rel_l2(np.nan_to_num(a), np.nan_to_num(b)) < tol

    def wclose(a, b):
        return float(np.nanmax(np.abs(a.detach().cpu().numpy() - b))) < TOL_PROB

    #  finite large-negative attn_mask + key padding: plain self-attention route, key padding folded in as -inf
    T, B = (int(v) for v in g["finite.meta"][:2])
    m = build("finite", self_attention=True)
    x, am, kpm = tt(MC.rnd("finite.x", (T, B, E))), tt(g["finite.attn_mask"]), tt(g["finite.kpm"])
    for mode, tol in (("f32", TOL_REL), ("f16x3", 2e-5)):
        m.gemm_dtype = mode
        y, w = m(x, x, x, attn_mask=am, key_padding_mask=kpm)
        assert close(y, g["finite.out"], tol) and float(np.abs(w.cpu().numpy() - g["finite.avg_weights"]).max()) < max(TOL_PROB, tol), mode
        assert float(w[0, :, :3].abs().max()) == 0.0 and float(w[1, :, 5:8].abs().max()) == 0.0       # padded keys: exactly 0
        yh, wh = m(x, x, x, attn_mask=am, key_padding_mask=kpm, need_head_weights=True)
        assert float(np.abs(wh.cpu().numpy() - g["finite.head_weights"]).max()) < max(TOL_PROB, tol), mode
    # cross-attention, queries shorter and longer than the keys
    m = build("cross", encoder_decoder_attention=True)
    for tag, Tq, S in (("short_q", 9, 23), ("long_q", 27, 13)):
        q, kv, kpm = tt(MC.rnd(f"cross.{tag}.q", (Tq, 3, E))), tt(MC.rnd(f"cross.{tag}.kv", (S, 3, E))), tt(g[f"cross.{tag}.kpm"])
        y, w = m(q, kv, kv)
        assert close(y, g[f"cross.{tag}.out"]) and wclose(w, g[f"cross.{tag}.avg_weights"]) and tuple(w.shape) == (3, Tq, S)
        y, w = m(q, kv, kv, key_padding_mask=kpm)
        assert close(y, g[f"cross.{tag}.out_kpm"]) and wclose(w, g[f"cross.{tag}.avg_weights_kpm"])
        y, w = m(q, kv, kv, key_padding_mask=kpm, need_head_weights=True)
        assert close(y, g[f"cross.{tag}.out_kpm"]) and wclose(w, g[f"cross.{tag}.head_weights_kpm"])
        y, w = m(q, kv, kv, key_padding_mask=kpm, need_weights=False)
        assert w is None and close(y, g[f"cross.{tag}.out_kpm"])
    # kdim / vdim with separate key and value
    m = build("kdim", wkw=dict(kdim=96, vdim=160), kdim=96, vdim=160)
    q, k, v = tt(MC.rnd("kdim.q", (11, 2, E))), tt(MC.rnd("kdim.k", (17, 2, 96))), tt(MC.rnd("kdim.v", (17, 2, 160)))
    y, w = m(q, k, v)
    assert close(y, g["kdim.out"]) and wclose(w, g["kdim.avg_weights"])
    y, w = m(q, k, v, need_head_weights=True)
    assert close(y, g["kdim.out"]) and wclose(w, g["kdim.head_weights"])
    # bias_kv + zero_attn: two more keys than queries
    m = build("biaskv", wkw=dict(bias_kv=True), self_attention=True, add_bias_kv=True, add_zero_attn=True)
    x, am, kpm = tt(MC.rnd("biaskv.x", (14, 3, E))), tt(MC.rnd("biaskv.am", (14, 14), 0.5)), tt(g["biaskv.kpm"])
    y, w = m(x, x, x)
    assert close(y, g["biaskv.out"]) and wclose(w, g["biaskv.avg_weights"]) and tuple(w.shape) == (3, 14, 16)
    y, w = m(x, x, x, attn_mask=am, key_padding_mask=kpm)
    assert close(y, g["biaskv.out_masked"]) and wclose(w, g["biaskv.avg_weights_masked"])
    y, w = m(x, x, x, attn_mask=am, key_padding_mask=kpm, need_head_weights=True)
    assert close(y, g["biaskv.out_masked"]) and wclose(w, g["biaskv.head_weights_masked"])
    # bias=False
    m = build("nobias", wkw=dict(bias=False), self_attention=True, bias=False)
    x = tt(MC.rnd("nobias.x", (10, 2, E)))
    y, w = m(x, x, x, need_head_weights=True)
    assert close(y, g["nobias.out"]) and wclose(w, g["nobias.head_weights"])
    # incremental self-attention: one position per call, the buffer in the caller's dict grows
    m = build("incr", self_attention=True)
    x = tt(MC.rnd("incr.x", (6, 2, E)))
    state, ys = {}, []
    for s_ in range(6):
        y, w = m(x[s_:s_ + 1], x[s_:s_ + 1], x[s_:s_ + 1], incremental_state=state)
        ys.append(y)
        assert tuple(w.shape) == (2, 1, s_ + 1)
    assert close(torch.cat(ys), g["incr.out_steps"]) and wclose(w, g["incr.last_weights"])
    buf = m._get_input_buffer(state)
    assert len(state) == 1 and close(buf["prev_key"], g["incr.final_prev_key"]) and close(buf["prev_value"], g["incr.final_prev_value"])
    causal = torch.triu(torch.full((6, 6), float("-inf"), device=dev), diagonal=1)
    yfull, _ = m(x, x, x, attn_mask=causal)                         # the same six outputs in one causal call (plain route)
    assert rel_l2(yfull.cpu(), g["incr.out_steps"]) < TOL_REL
    state, ys = {}, []
    for s_ in range(4):                                              # a key padding mask entering on the third step
        kp = torch.tensor([[False], [True]], device=dev) if s_ == 2 else None
        ys.append(m(x[s_:s_ + 1], x[s_:s_ + 1], x[s_:s_ + 1], incremental_state=state, key_padding_mask=kp)[0])
    assert close(torch.cat(ys), g["incr.out_steps_kpm"])
    assert np.array_equal(m._get_input_buffer(state)["prev_key_padding_mask"].cpu().numpy(), g["incr.final_kpm"])
    m.reorder_incremental_state(state, torch.tensor([1, 0], device=dev))
    assert np.array_equal(m._get_input_buffer(state)["prev_key_padding_mask"].cpu().numpy(), g["incr.final_kpm"][::-1])
    # incremental encoder-decoder attention: keys / values projected once (static_kv), then key = value = None
    m = build("incr_ed", encoder_decoder_attention=True)
    enc, q, ekpm = tt(MC.rnd("incr_ed.enc", (12, 2, E))), tt(MC.rnd("incr_ed.q", (3, 2, E))), tt(g["incr_ed.kpm"])
    state = {}
    ys = [m(q[0:1], enc, enc, key_padding_mask=ekpm, incremental_state=state, static_kv=True)[0]]
    for s_ in (1, 2):
        y, w = m(q[s_:s_ + 1], None, None, key_padding_mask=ekpm, incremental_state=state, static_kv=True)
        ys.append(y)
    assert close(torch.cat(ys), g["incr_ed.out_steps"]) and wclose(w, g["incr_ed.last_weights"])
    # before_softmax on the reference's own route; ignored -- as there -- on its functional route
    m = build("pre", self_attention=True)
    x, am, kpm = tt(MC.rnd("pre.x", (9, 2, E))), tt(MC.rnd("pre.am", (9, 9), 0.3)), tt(g["pre.kpm"])
    sc, vv = m(x, x, x, attn_mask=am, key_padding_mask=kpm, before_softmax=True, need_head_weights=True)
    fin = np.isfinite(g["pre.scores"])
    assert np.array_equal(np.isneginf(sc.cpu().numpy()), ~fin)
    assert float(np.abs(np.where(fin, sc.cpu().numpy(), 0) - np.where(fin, g["pre.scores"], 0)).max()) < 5e-5 and close(vv, g["pre.values"])
    y0, w0 = m(x, x, x, attn_mask=am, key_padding_mask=kpm)
    y1, w1 = m(x, x, x, attn_mask=am, key_padding_mask=kpm, before_softmax=True)
    assert torch.equal(y0, y1) and torch.equal(w0, w1) and tuple(y1.shape) == (9, 2, E)


def test_residual_block_around_a_foreign_layer_and_all_masked_mha_match_reference_fixtures(dev):
    """VERDICT r03 "missing" 3, against outputs of the reference itself (tests/golden/make_golden_r4.py):
    (a) NormalizedResidualBlock around a layer that is NOT one of this package's modules (modules.py:369-401 wraps anything):
        LayerNorm by rnamsm_layernorm, the layer called with the caller's extra keyword, a tuple result split into
        (x, *rest), the residual added by rnamsm_add -- plain and tuple-returning layer;
    (b) MultiheadAttention with a batch element whose keys are ALL masked: NaN output and weights for that element as in
        msm/multihead_attention.py:360-371 (masked_fill(-inf) then softmax), the other elements at the usual bar; every
        route (averaged weights, per-head weights, the fused no-weights kernel, the f16x3 mode)."""
    from rnamsm import modules as M
    g = golden("residual_block_foreign.npz")
    D = g["x"].shape[-1]

    class Foreign(torch.nn.Module):
        def __init__(self, as_tuple):
            super().__init__()
            self.lin = torch.nn.Linear(D, D)
            self.as_tuple = as_tuple

        def forward(self, x, gain=1.0):
            y = torch.tanh(self.lin(x)) * gain
            return (y, x.mean(dim=-1)) if self.as_tuple else y

    x = torch.from_numpy(g["x"]).to(dev)
    for as_tuple in (False, True):
        blk = M.NormalizedResidualBlock(Foreign(as_tuple), D)
        with torch.no_grad():
            blk.layer.lin.weight.copy_(torch.from_numpy(g["lin_w"])); blk.layer.lin.bias.copy_(torch.from_numpy(g["lin_b"]))
            blk.layer_norm.weight.copy_(torch.from_numpy(g["ln_g"])); blk.layer_norm.bias.copy_(torch.from_numpy(g["ln_b"]))
        blk = blk.eval().to(dev)
        with torch.no_grad():
            res = blk(x, gain=0.5)
        if as_tuple:
            assert isinstance(res, tuple) and len(res) == 2
            assert rel_l2(res[0].cpu(), g["tuple_out"]) < TOL_REL and rel_l2(res[1].cpu(), g["tuple_extra"]) < 1e-4
        else:
            assert torch.is_tensor(res) and res.shape == x.shape and rel_l2(res.cpu(), g["plain_out"]) < TOL_REL
    gm = golden("mha_all_masked.npz")
    T, B, E, H = (int(v) for v in gm["meta"])
    state = synthetic.make_state_dict(seed=11, embed_dim=E, num_layers=1, num_heads=H)
    prefix = "layers.0.row_self_attention.layer"
    mha = M.MultiheadAttention(E, H, self_attention=True)
    mha.load_state_dict({k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in state.items() if k.startswith(prefix + ".")}, strict=True)
    mha = mha.eval().to(dev)
    xm = torch.from_numpy(synthetic.normal("mha:all_masked", 11, (T, B, E)).astype(np.float32)).to(dev)
    kpm = torch.from_numpy(gm["key_padding_mask"]).to(dev)
    live = [0, 2]

    def check(y, w, want_w, tol_rel, tol_p):
        assert bool(torch.isnan(y[:, 1]).all()) and bool(np.isnan(gm["out"][:, 1]).all())
        assert rel_l2(y[:, live].cpu(), gm["out"][:, live]) < tol_rel
        if w is not None:
            wl, wd = (w[:, live], w[:, 1]) if w.dim() == 4 else (w[live], w[1])
            gl = want_w[:, live] if want_w.ndim == 4 else want_w[live]
            assert bool(torch.isnan(wd).all()) and np.abs(wl.cpu().numpy() - gl).max() < tol_p
    for mode, tr, tp in (("f32", TOL_REL, TOL_PROB), ("f16x3", 2e-5, 2e-5)):
        mha.gemm_dtype = mode
        check(*mha(xm, xm, xm, key_padding_mask=kpm, need_weights=True), gm["avg_weights"], tr, tp)
        check(*mha(xm, xm, xm, key_padding_mask=kpm, need_head_weights=True), gm["head_weights"], tr, tp)
        check(*mha(xm, xm, xm, key_padding_mask=kpm, need_weights=False), None, tr, tp)
    mha.gemm_dtype = "f32"


def test_greedy_select_on_device_equals_host_and_reference(dev, full_2drb1_a2m):
    """device greedy max/min-Hamming sub-sampling picks exactly the reference's rows (fixture from
    MSA.greedy_select) and exactly the host implementation's rows on larger random alignments with many ties."""
    import os
    from conftest import GOLDEN
    from rnamsm import msa
    from rnamsm.alphabet import RNAAlphabet
    g = golden("tokens_2DRB_1_first64.npz")
    path = os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")
    a = RNAAlphabet()
    assert np.array_equal(msa.load_msa_tokens(path, a, 16, "diversity-max", device=dev), g["diversity_max_16"])
    assert np.array_equal(msa.load_msa_tokens(path, a, 16, "diversity-min", device=dev), g["diversity_min_16"])
    # the shipped alignment at the CLI default (BASELINE configs[0]): 1176 rows -> 512.  Candidates tie in their mismatch
    # totals here and the winner depends on numpy's pairwise summation order (a running sum differs from step 46 on)
    full = full_2drb1_a2m
    assert np.array_equal(msa.load_msa_tokens(full, a, 512, "diversity-max", device=dev),
                          golden("tokens_2DRB_1_full.npz")["diversity_max_512"])
    rng = np.random.RandomState(3)
    for N, L, K in ((50, 7, 50), (300, 33, 40), (5000, 120, 64), (1500, 513, 17), (1200, 35, 300), (700, 20, 700 - 1)):
        toks = np.concatenate([np.zeros((N, 1), np.int64), rng.choice([4, 5, 6, 7, 10], size=(N, L), p=[.3, .3, .2, .1, .1])], 1)
        toks[rng.randint(0, N, N // 3)] = toks[0]            # duplicated rows: exact ties, first index must win
        for mode in ("max", "min"):
            want = msa.greedy_select(toks, K, mode)
            got = msa.greedy_select_device(toks, K, mode, dev)
            assert np.array_equal(got, want), (N, L, K, mode)
            # both launch schemes (one launch per step with a wave per row; three with a thread per row) on every case,
            # whatever the default picks by depth
            from rnamsm import ops
            for scheme in (0, 2):
                try:
                    ops.set_param("greedy_fused", scheme)
                    assert np.array_equal(msa.greedy_select_device(toks, K, mode, dev), want), (N, L, K, mode, scheme)
                finally:
                    ops.set_param("greedy_fused", 1)
    # a history deeper than one leaf of numpy's pairwise recursion in every piece: 1023 steps (the model's limit)
    N, L = 1100, 24
    toks = np.concatenate([np.zeros((N, 1), np.int64), rng.choice([4, 5, 6, 7, 10], size=(N, L))], 1)
    want = msa.greedy_select(toks, 1024, "max")
    for scheme in (0, 2):
        try:
            ops.set_param("greedy_fused", scheme)
            assert np.array_equal(msa.greedy_select_device(toks, 1024, "max", dev), want), scheme
        finally:
            ops.set_param("greedy_fused", 1)


def test_msa_weights_on_device_equal_host_and_reference(dev, full_2drb1_a2m):
    """§8 f3, `sample-pretrained` (utils/align.py:150-163, 250-253): rnamsm_msa_weights gives the reference's float64
    sequence weights bit for bit (fixture: the shipped 1176-row alignment), so the weighted draw picks the same rows; and
    equals the host implementation on shapes that exercise every lane-group width (L = 3 .. 300)."""
    import os
    from conftest import GOLDEN
    from rnamsm import msa
    from rnamsm.alphabet import RNAAlphabet
    g = golden("msa_weights_2DRB_1.npz")
    path = full_2drb1_a2m
    a = RNAAlphabet()
    toks = msa.load_msa_tokens(path, a, None)
    assert np.array_equal(msa.msa_weights(toks, float(g["seqid_cutoff"]), device=dev), g["weights"])
    got = msa.load_msa_tokens(path, a, 512, "sample-pretrained", device=dev, rng=np.random.RandomState(42))
    assert np.array_equal(got, g["tokens_seed42_n512"])
    rng = np.random.RandomState(5)
    for N, L, cut in ((40, 3, 0.4), (257, 16, 0.2), (600, 33, 0.3), (300, 300, 0.25), (1000, 64, 0.2)):
        t = np.concatenate([np.zeros((N, 1), np.int64), rng.choice([4, 5, 6, 7, 10], size=(N, L), p=[.4, .3, .1, .1, .1])], 1)
        t[rng.randint(0, N, N // 4)] = t[1]                           # clusters of identical rows
        assert np.array_equal(msa.msa_weights(t, cut, device=dev), msa.msa_weights(t, cut)), (N, L)


@pytest.mark.parametrize("M,N,K,cols", [(117, 384, 128, 128), (1000, 2304, 768, 768), (300, 128, 64, 0)])
def test_gemm_with_a_per_row_factor_on_the_scaled_columns(dev, M, N, K, cols):
    """rnamsm_gemm_row_scaled (the general form of zero_rows): ((x W^T + b) * scale) * f[m] on the first
    `cols` columns, untouched elsewhere; with f in {0, 1} it IS the zero_rows kernel bit for bit, and it equals the two
    launches it replaces in the ragged batch driver (GEMM with the scale, then the row factor) bit for bit."""
    from rnamsm import ops
    x, w, b = _rand(f"rs.x{M}", (M, K)), _rand(f"rs.w{N}", (N, K), 0.1), _rand(f"rs.b{N}", (N,), 0.1)
    f = torch.from_numpy(np.abs(synthetic.normal(f"rs.f{M}", 3, (M,))).astype(np.float32))
    f[::5] = 0.0                                                   # <pad> tokens
    g = ops.linear_row_scaled(x.to(dev), w.to(dev), b.to(dev), f.to(dev), scale=0.125, scale_cols=cols)
    want = x.double() @ w.double().t() + b.double()
    want[:, :cols] *= 0.125 * f.double()[:, None]
    assert rel_l2(g.cpu(), want) < 3e-6
    two = ops.linear(x.to(dev), w.to(dev), b.to(dev), scale=0.125, scale_cols=cols)
    two[:, :cols] *= f.to(dev)[:, None]
    assert torch.equal(g, two)
    if cols:
        ind = (f != 0).to(torch.float32)
        a = ops.linear_row_scaled(x.to(dev), w.to(dev), b.to(dev), ind.to(dev), scale=0.125, scale_cols=cols)
        z = ops.linear(x.to(dev), w.to(dev), b.to(dev), scale=0.125, scale_cols=cols, zero_rows=(f == 0).to(torch.uint8).to(dev))
        assert torch.equal(a, z)


def test_padding_mask_kernel_semantics(dev):
    """SURVEY §8 f2 at kernel level: zero_rows on the q columns of the QKV GEMM, -10000 key fill in the row softmax,
    -10000 fill of padded keys in fused column attention (including a fully padded column -> uniform weights)."""
    from rnamsm import ops
    R, C, H = 9, 13, 2
    D = 64 * H
    rng = np.random.RandomState(0)
    pad = torch.from_numpy(rng.rand(R, C) < 0.2)
    pad[:, 4] = True                                            # one column padded in every row
    pad[0, 7] = True
    mask = pad.to(torch.uint8).contiguous().view(-1).to(dev)
    x, w, b = _rand("pm.x", (R * C, D)), _rand("pm.w", (3 * D, D), 0.1), _rand("pm.b", (3 * D,), 0.1)
    qkv = ops.linear(x.to(dev), w.to(dev), b.to(dev), scale=0.25, scale_cols=D, zero_rows=mask)
    want = x.double() @ w.double().t() + b.double()
    want[:, :D] *= 0.25
    want[pad.view(-1), :D] = 0
    assert rel_l2(qkv.cpu(), want) < 3e-6 and float(qkv[mask.bool(), :D].abs().max()) == 0.0
    q, k, v = (want[:, i * D:(i + 1) * D].view(R, C, H, 64) for i in range(3))
    partial, _ = ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H)
    probs = ops.softmax_rows(partial, key_mask=mask[:C]).cpu()
    logits = torch.einsum("rihd,rjhd->hij", q, k).masked_fill(pad[0][None, None, :], -10000)
    assert np.abs(probs.numpy() - torch.softmax(logits, -1).numpy()).max() < TOL_PROB
    ctx = ops.col_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], R, C, H, pad_mask=mask).cpu()
    wc = torch.einsum("ichd,jchd->hcij", q, k).masked_fill(pad.t()[None, :, None, :], -10000)
    want_ctx = torch.einsum("hcij,jchd->ichd", torch.softmax(wc, -1), v).reshape(R * C, D)
    assert rel_l2(ctx, want_ctx) < 5e-6


@pytest.mark.parametrize("R,C,max_tokens", [(9, 13, 13 * 2), (9, 13, 1), (70, 33, 33 * 16), (6, 140, 140 * 4)])
def test_chunked_path_padding_mask_kernels(dev, R, C, max_tokens):
    """SURVEY §8 f2 "chunk quirk" at kernel level (modules.py:717-750): rnamsm_row_logits_chunked emits one slab per
    reference row chunk and rnamsm_softmax_rows_chunked fills every slab with -10000 where the chunk's own first row is
    padded before adding the slabs in chunk order.  Checked against that arithmetic in fp64 (logits without an
    all-padded chunk start: there -10000 lands in the sum and fp32 quantises it, see the model-level test)."""
    from rnamsm import ops
    H = 2
    D = 64 * H
    rng = np.random.RandomState(R + C)
    pad = torch.from_numpy(rng.rand(R, C) < 0.15)
    pad[0, C // 2] = True
    nchunks, rpc = ops.row_chunks(R, C, max_tokens)
    assert nchunks == -(-R // max(1, max_tokens // C)) and rpc == max(1, max_tokens // C)
    assert ops.row_chunks(R, C, R * C) == (0, 0)                      # at the budget: the reference's direct path
    mask = pad.to(torch.uint8).contiguous().view(-1).to(dev)
    x, w, b = _rand("cp.x", (R * C, D)), _rand("cp.w", (3 * D, D), 0.1), _rand("cp.b", (3 * D,), 0.1)
    qkv = ops.linear(x.to(dev), w.to(dev), b.to(dev), scale=0.25, scale_cols=D, zero_rows=mask)
    want = x.double() @ w.double().t() + b.double()
    want[:, :D] *= 0.25
    want[pad.view(-1), :D] = 0
    q, k = (want[:, i * D:(i + 1) * D].view(R, C, H, 64) for i in range(2))
    partial, n = ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H, rows_per_chunk=rpc)
    assert n == nchunks and partial.shape == (nchunks, H, C, C)
    logits = 0
    for c in range(nchunks):
        s = slice(c * rpc, (c + 1) * rpc)
        wc = torch.einsum("rihd,rjhd->hij", q[s], k[s])
        assert rel_l2(partial[c].cpu(), wc) < 3e-6                     # slab c is chunk c's logits
        logits = logits + wc.masked_fill(pad[c * rpc][None, None, :], -10000)
    probs = ops.softmax_rows(partial, chunk_pad_mask=mask, rows_per_chunk=rpc).cpu()
    assert np.abs(probs.numpy() - torch.softmax(logits, -1).numpy()).max() < 3e-4   # fp32 quantisation of n * -10000 sums
    masked_anywhere = torch.stack([pad[c * rpc] for c in range(nchunks)]).any(0)
    assert float(probs[:, :, masked_anywhere].max()) == 0.0            # a pad on ANY chunk-starting row kills the key
    direct = ops.softmax_rows(ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H)[0], key_mask=mask[:C]).cpu()
    if bool((masked_anywhere & ~pad[0]).any()):
        assert float((probs - direct).abs().max()) > 1e-3              # ... which the direct semantics do not do


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (300, 384, 128), (1025, 768, 768), (513, 768, 3072)])
def test_gemm_16bit_matrix_core_modes(dev, M, N, K):
    """rnamsm_gemm_bf16 (include/rnamsm.h): stated error of each operand mode vs fp64 -- bf16 (2^-9 operands),
    f16x3 (hi/lo fp16, ~2^-22: the same as the exact-fp32 kernel) -- with every epilogue."""
    from rnamsm import ops
    from rnamsm._lib import ACT_GELU_ERF
    a, w, b, r = _rand("h.a", (M, K)), _rand("h.w", (N, K), 0.05), _rand("h.b", (N,), 0.1), _rand("h.r", (M, N))
    base = a.double() @ w.double().t() + b.double()
    planes = {0: ops.split_bf16(w.to(dev), fmt=0), 1: ops.split_bf16(w.to(dev), fmt=1)}
    for split, fmt, tol in ((1, 0, 6e-3), (3, 1, 3e-6)):
        hi, lo = planes[fmt]
        y = ops.linear_bf16(a.to(dev), hi, lo if split == 3 else None, b.to(dev), split=split, fmt=fmt).cpu()
        assert rel_l2(y, base) < tol, (split, fmt)
        y = ops.linear_bf16(a.to(dev), hi, lo if split == 3 else None, b.to(dev), act=ACT_GELU_ERF, residual=r.to(dev),
                            scale=0.5, scale_cols=128, split=split, fmt=fmt).cpu()
        want = base.clone(); want[:, :128] *= 0.5
        assert rel_l2(y, O.gelu_erf(want) + r.double()) < tol, (split, fmt)
    # exact-integer operands are exactly representable in both 16-bit formats: bit-exact and transposition-proof
    ai = torch.from_numpy(((np.arange(M * K).reshape(M, K) * 7 + 3) % 13 - 6).astype(np.float32))
    wi = torch.from_numpy(((np.arange(N * K).reshape(N, K) * 5 + 1) % 11 - 5).astype(np.float32))
    for split, fmt in ((1, 0), (3, 1)):
        hi, lo = ops.split_bf16(wi.to(dev), fmt=fmt)
        assert torch.equal(ops.linear_bf16(ai.to(dev), hi, lo if split == 3 else None, split=split, fmt=fmt).cpu(), ai @ wi.t())
    # the removed hi/lo-bf16 mode answers UNSUPPORTED (never another arithmetic silently)
    hi, lo = ops.split_bf16(wi.to(dev), fmt=0)
    with pytest.raises(NotImplementedError):
        ops.linear_bf16(ai.to(dev), hi, lo, split=3, fmt=0)


@pytest.mark.parametrize("M,N,K", [(2048, 256, 64), (2300, 768, 768), (4100, 512, 3072)])
def test_plane_input_gemm_kernels_agree(dev, M, N, K):
    """The plane-input 16-bit GEMM has four staging variants (rnamsm_set_param "gemm16_dma": 0 register-staged, 1 LDS-DMA
    128x128, 2 LDS-DMA 256x256, 3 LDS-DMA 256x256 with software-pipelined fragments = default, 4 = 3 with 32-deep K tiles
    for plain bf16 too) and a block-order knob ("gemm_group").  Same
    operands, same MFMA order per output element within a k16 step -> every variant must match the fp64 result of the
    plane values at fp32-accumulation accuracy, for fp32 and plane outputs, and be exact on integers."""
    from rnamsm import ops, _lib
    from rnamsm._lib import ACT_GELU_ERF
    lib = _lib.load()
    a, w, b, r = _rand("p.a", (M, K)), _rand("p.w", (N, K), 0.05), _rand("p.b", (N,), 0.1), _rand("p.r", (M, N))
    ai = torch.from_numpy(((np.arange(M * K).reshape(M, K) * 7 + 3) % 13 - 6).astype(np.float32))
    wi = torch.from_numpy(((np.arange(N * K).reshape(N, K) * 5 + 1) % 11 - 5).astype(np.float32))
    default = lib.rnamsm_get_param(b"gemm16_dma")
    try:
        for variant, group in ((0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 5), (4, 0)):
            _lib.check(lib.rnamsm_set_param(b"gemm16_dma", variant))
            _lib.check(lib.rnamsm_set_param(b"gemm_group", group))
            for split, fmt, tol in ((1, 0, 2e-6), (3, 1, 3e-6)):
                ht = torch.float16 if fmt == 1 else torch.bfloat16
                ap = ops.split_bf16(a.to(dev), want_lo=split == 3, fmt=fmt)
                wp = ops.split_bf16(w.to(dev), want_lo=split == 3, fmt=fmt)
                eff = lambda pl: sum(p.view(ht).double().cpu() for p in pl if p is not None)
                base = eff(ap) @ eff(wp).t() + b.double()
                y = ops.linear_planes(ap, wp, b.to(dev), residual=r.to(dev), fmt=fmt).cpu()
                assert rel_l2(y, base + r.double()) < tol, (variant, split, fmt)
                oh, ol = ops.linear_planes(ap, wp, b.to(dev), act=ACT_GELU_ERF, out_planes=True, fmt=fmt)
                got = eff((oh, ol))
                assert rel_l2(got, O.gelu_erf(base)) < (6e-3 if split == 1 else 4e-5 if fmt == 0 else 3e-6), (variant, split, fmt)
                oh, ol = ops.linear_planes(ap, wp, b.to(dev), scale=0.25, scale_cols=128, out_planes=True, fmt=fmt)   # QKV form
                want = base.clone(); want[:, :128] *= 0.25
                assert rel_l2(eff((oh, ol)), want) < (6e-3 if split == 1 else 4e-5 if fmt == 0 else 3e-6), (variant, split, fmt)
                aip = ops.split_bf16(ai.to(dev), want_lo=split == 3, fmt=fmt)
                wip = ops.split_bf16(wi.to(dev), want_lo=split == 3, fmt=fmt)
                assert torch.equal(ops.linear_planes(aip, wip, fmt=fmt).cpu(), ai @ wi.t()), (variant, split, fmt)
    finally:
        _lib.check(lib.rnamsm_set_param(b"gemm16_dma", default))
        _lib.check(lib.rnamsm_set_param(b"gemm_group", 0))


@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
def test_layernorm_folded_into_the_16bit_gemms(dev, split, fmt):
    """K1 folded in the 16-bit modes: a producer (out_proj shape: x += ctx W^T + b, new x also as planes + slab sums) feeding a
    consumer (QKV shape with q scaling; fc1 shape with GELU) that reads the RAW x planes and applies (mean, rstd) to its
    accumulators -- against fp64 LayerNorm -> Linear on the values the planes hold, and next to the unfused plane path
    (rnamsm_layernorm_split + rnamsm_gemm_bf16)."""
    from rnamsm import ops
    from rnamsm._lib import ACT_GELU_ERF
    M, D, F = 2304, 768, 1024
    ht = torch.float16 if fmt == 1 else torch.bfloat16
    eff = lambda pl: sum(p.view(ht).double() for p in pl if p is not None)
    lo = split == 3
    x0 = (_rand("f16.x", (M, D), 1.5) + 0.3).to(dev)
    ctx = ops.split_bf16(_rand("f16.ctx", (M, D)).to(dev), want_lo=lo, fmt=fmt)
    wo = ops.split_bf16(_rand("f16.wo", (D, D), 0.05).to(dev), want_lo=lo, fmt=fmt)
    bo = _rand("f16.bo", (D,), 0.1).to(dev)
    # ---- producer
    x = x0.clone()
    xpl, part = ops.linear_planes_residual_stats(ctx, wo, bo, x, fmt=fmt)
    want_x = x0.double() + eff(ctx) @ eff(wo).t() + bo.double()
    assert rel_l2(x.cpu(), want_x.cpu()) < (2e-6 if split == 1 else 4e-5 if fmt == 0 else 3e-6)
    assert torch.equal(x, ops.linear_planes(ctx, wo, bo, residual=x0, fmt=fmt))          # the plain kernel's bits
    assert rel_l2(eff(xpl).cpu(), x.double().cpu()) < (4e-3 if split == 1 else 2e-5 if fmt == 0 else 2e-7)      # planes of the new x
    slabs = x.double().view(M, D // 32, 32).transpose(0, 1)
    assert rel_l2(part[..., 0].cpu(), slabs.sum(-1).cpu()) < 1e-6
    assert rel_l2(part[..., 1].cpu(), ((slabs - slabs.mean(-1, keepdim=True)) ** 2).sum(-1).cpu()) < 1e-5
    st = ops.row_stats_from_partials(part, D)
    # ---- consumers
    g, be = (1 + 0.1 * _rand("f16.g", (D,))).to(dev), (0.1 * _rand("f16.be", (D,))).to(dev)
    xv = eff(xpl)                                                             # what the consumer multiplies
    mean, rstd = x.double().mean(1, keepdim=True), torch.rsqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-5)
    for N, act, scale_cols in ((3 * D, 0, D), (F, ACT_GELU_ERF, 0)):
        w, b = _rand(f"f16.w{N}", (N, D), 0.05).to(dev), _rand(f"f16.b{N}", (N,), 0.1).to(dev)
        wg32, _, dvec = ops.ln_fold_weights(w, b, g, be)
        wg = ops.split_bf16(wg32, want_lo=lo, fmt=fmt)
        cvec = eff(wg).sum(1).float()                                        # of the plane values the GEMM multiplies
        oh, ol = ops.linear_planes_lnfold(xpl, wg, cvec, dvec, st, act=act, scale=0.125, scale_cols=scale_cols, fmt=fmt)
        want = rstd * (xv @ eff(wg).t() - mean * eff(wg).sum(1)) + dvec.double()
        want[:, :scale_cols] *= 0.125
        if act:
            want = O.gelu_erf(want)
        tol = 6e-3 if split == 1 else 4e-5 if fmt == 0 else 3e-6           # the output planes' own rounding
        assert rel_l2(eff((oh, ol)).cpu(), want.cpu()) < tol, (N, rel_l2(eff((oh, ol)).cpu(), want.cpu()))
        # and the mode's accuracy against exact LayerNorm -> Linear in fp64, next to the unfused plane path
        exact = O.layer_norm(x.double(), g.double(), be.double()) @ w.double().t() + b.double()
        exact[:, :scale_cols] *= 0.125
        if act:
            exact = O.gelu_erf(exact)
        xn = ops.layernorm_split(x, g, be, split=split, fmt=fmt)
        wpl = ops.split_bf16(w, want_lo=lo, fmt=fmt)
        uh, ul = ops.linear_planes(xn, wpl, b, act=act, scale=0.125, scale_cols=scale_cols, out_planes=True, fmt=fmt)
        e_fold, e_plain = rel_l2(eff((oh, ol)).cpu(), exact.cpu()), rel_l2(eff((uh, ul)).cpu(), exact.cpu())
        assert e_fold < 1.5 * e_plain + 1e-6, (N, e_fold, e_plain)


@pytest.mark.parametrize("M,N,K,act", [(16384, 2304, 768, 0), (9000, 3072, 768, 1), (2048, 1280, 64, 0)])
def test_16x16x32_gemm_with_the_next_tile_requested_before_the_stores(dev, M, N, K, act):
    """Plain-bf16 QKV / fc1 shapes run the 16x16x32-MFMA kernel whose persistent blocks request the next tile's first operands
    BEFORE the current tile's 16 stores and then wait with vmcnt(16) (in-order retirement: everything but those stores).  A
    miscount would let a wave read LDS ahead of its DMA -- rarely, and only with many tiles per block -- so: many tiles per
    block (up to 9 at 256 blocks), a ragged last row panel (M = 9000: that tile drains instead), reruns bit-identical, and
    the values against fp64 on what the planes hold."""
    from rnamsm import ops
    ht = torch.bfloat16
    a = ops.split_bf16(_rand("hq.a", (M, K)).to(dev), want_lo=False)
    w = ops.split_bf16(_rand("hq.w", (N, K), 0.05).to(dev), want_lo=False)
    b = _rand("hq.b", (N,), 0.1).to(dev)
    first = ops.linear_planes(a, w, b, act=act, scale=0.125, scale_cols=768 if not act else 0, out_planes=True)[0].clone()
    for _ in range(25):
        again = ops.linear_planes(a, w, b, act=act, scale=0.125, scale_cols=768 if not act else 0, out_planes=True)[0]
        assert torch.equal(again, first)
    rows = torch.cat([torch.arange(0, 512), torch.arange(M - 300, M)]).to(dev)      # first tiles and the ragged tail
    want = a[0].view(ht)[rows].double() @ w[0].view(ht).double().t() + b.double()
    if act:
        want = O.gelu_erf(want)
    else:
        want[:, :768] *= 0.125
    assert rel_l2(first.view(ht)[rows].double().cpu(), want.cpu()) < 6e-3


@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
@pytest.mark.parametrize("M,N,K,kind", [(16384, 2304, 768, "planes"), (9000, 3072, 768, "gelu"), (8200, 768, 3072, "residual"),
                                        (4100, 768, 768, "residual"), (2048, 1280, 128, "planes")])
def test_the_256x256_gemms_are_bit_identical_across_reruns(dev, M, N, K, kind, split, fmt):
    """The 256x256 16-bit GEMMs overlap LDS-DMA writes with fragment reads behind counted waits (the upper wave group issues its
    requests one micro-step after the lower one; the plain-bf16 16x16x32 kernel stages by operand behind two barriers per tile --
    gemm16_q16s_kernel, also taken by fc2, K >= 2048).  A wave that reads LDS ahead of a DMA shows as a rare mismatch: many tiles
    per block, ragged last row panels, 2 .. 48 K tiles, twelve reruns bit-identical; plain-bf16 fc2 also against float64.
    (Until round 6 this test walked the issue schedules of the removed knob "gemm16_dephase": all bit-identical.)"""
    from rnamsm import ops
    from rnamsm._lib import ACT_GELU_ERF
    lo = split == 3
    a = ops.split_bf16(_rand("dp.a", (M, K)).to(dev), want_lo=lo, fmt=fmt)
    w = ops.split_bf16(_rand("dp.w", (N, K), 0.05).to(dev), want_lo=lo, fmt=fmt)
    b = _rand("dp.b", (N,), 0.1).to(dev)
    r = _rand("dp.r", (M, N)).to(dev) if kind == "residual" else None

    def run():
        if kind == "residual":
            return [ops.linear_planes(a, w, b, residual=r, fmt=fmt)]
        out = ops.linear_planes(a, w, b, act=ACT_GELU_ERF if kind == "gelu" else 0, out_planes=True, fmt=fmt)
        return [t for t in out if t is not None]

    want = [t.clone() for t in run()]
    for _ in range(12):
        got = [t.clone() for t in run()]
        assert all(torch.equal(g, w_) for g, w_ in zip(got, want)), kind
    if split == 1 and K >= 2048 and N <= 1024:
        ht = torch.bfloat16
        base = a[0].view(ht).double() @ w[0].view(ht).double().t() + b.double() + r.double()
        assert rel_l2(want[0].double().cpu(), base.cpu()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(300, 2304, 96), (4100, 1280, 64), (1025, 768, 768), (129, 3072, 32), (9000, 2304, 64),
                                   (2100, 3072, 32), (82, 2304, 768)])
def test_gemm_block_order_never_changes_results(dev, M, N, K):
    """rnamsm_set_param("gemm_group"): the XCD-aware block order (whole panels, groups of G panels, by-shape default)
    only permutes which block computes which tile -- every setting must give bit-identical output, including ragged M
    (padding blocks of the grid must exit without touching memory).  Round 4: no padding GROUPS any more -- an XCD's last
    panels form a smaller group (9000 rows: 71 panels = 8 + 1 per XCD) -- and by default a GEMM of at most 512 tiles is dealt
    flat, tile = block id (82 x 2304: a lone small alignment's QKV, 18 tiles: 117 -> 28 us)."""
    from rnamsm import ops, _lib
    lib = _lib.load()
    a, w, b = _rand("go.a", (M, K)), _rand("go.w", (N, K), 0.05), _rand("go.b", (N,), 0.1)
    ga, gw, gb = a.to(dev), w.to(dev), b.to(dev)
    try:
        outs = []
        for g, tile in ((0, 0), (1, 1), (3, 1), (8, 1), (64, 1), (1, 2), (5, 2), (0, 2)):     # tile 1 = 128x128, 2 = 128x64
            _lib.check(lib.rnamsm_set_param(b"gemm_group", g))
            _lib.check(lib.rnamsm_set_param(b"gemm_tile", tile))
            guard = torch.full((M + 1, N), 7.0, device=dev)                  # one sentinel row behind the output
            ops.linear(ga, gw, gb, out=guard[:M])
            assert bool((guard[M] == 7.0).all()), (g, tile)
            outs.append(guard[:M].clone())
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        assert rel_l2(outs[0].cpu(), a.double() @ w.double().t() + b.double()) < 3e-6
        with pytest.raises(_lib.RnamsmError):
            _lib.check(lib.rnamsm_set_param(b"gemm_group", 65))
    finally:
        _lib.check(lib.rnamsm_set_param(b"gemm_group", 0))
        _lib.check(lib.rnamsm_set_param(b"gemm_tile", 0))


def test_gemm_half_width_tile_epilogues(dev):
    """The 128x64 block tile (rnamsm_set_param("gemm_tile", 2); chosen by shape on small problems) with every epilogue:
    bias, column scale, erf-GELU, residual, ragged M -- bit-identical to the 128x128 tile."""
    from rnamsm import ops, _lib
    from rnamsm._lib import ACT_GELU_ERF
    lib = _lib.load()
    M, N, K = 333, 384, 160
    a, w, b, r = _rand("hw.a", (M, K)), _rand("hw.w", (N, K), 0.05), _rand("hw.b", (N,), 0.1), _rand("hw.r", (M, N))
    try:
        got = {}
        for tile in (1, 2):
            _lib.check(lib.rnamsm_set_param(b"gemm_tile", tile))
            got[tile] = (ops.linear(a.to(dev), w.to(dev), b.to(dev), act=ACT_GELU_ERF, residual=r.to(dev), scale=0.5, scale_cols=96).cpu(),
                         ops.linear(a.to(dev), w.to(dev), None, scale=0.25, scale_cols=N).cpu())
        assert torch.equal(got[1][0], got[2][0]) and torch.equal(got[1][1], got[2][1])
        want = a.double() @ w.double().t() + b.double(); want[:, :96] *= 0.5
        assert rel_l2(got[2][0], O.gelu_erf(want) + r.double()) < 3e-6
    finally:
        _lib.check(lib.rnamsm_set_param(b"gemm_tile", 0))


def test_full_grid_exact_kernels_on_integers(dev):
    """BASELINE configs[2] sizes (T = 131072 tokens: 18432-block GEMM grids, every CU at its resident-block limit): on
    small-integer operands the exact-fp32 kernels must reproduce the integer contraction bit for bit over the WHOLE
    output -- reference = torch's own GPU fp32 matmul / einsum, also exact on such data (plumbing as a checker)."""
    from rnamsm import ops
    R, C, H = 256, 512, 12
    D = 64 * H
    T = R * C

    def ints(shape, mul, mod, off):
        i = torch.arange(int(np.prod(shape)), device=dev, dtype=torch.int64)
        return (((i * mul + (i // 191) * 3) % mod) - off).to(torch.float32).view(*shape)

    a = ints((T, D), 7, 13, 6)
    for N, K in ((3 * D, D), (D, 4 * D)):
        w = ints((N, K), 5, 11, 5)
        x = a if K == D else ints((T, K), 3, 7, 3)
        assert torch.equal(ops.linear(x, w), x @ w.t()), (N, K)
        del w
    qkv = ints((T, 3 * D), 7, 5, 2)
    q, k, v = (qkv[:, i * D:(i + 1) * D].view(R, C, H, 64) for i in range(3))
    part, _ = ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H)
    assert torch.equal(part.sum(0), torch.einsum("rihd,rjhd->hij", q, k))
    p = ints((H, C, C), 5, 7, 3)
    ctx = ops.row_apply(p, qkv[:, 2 * D:], R, C, H)
    assert torch.equal(ctx.view(R, C, H, 64), torch.einsum("hij,rjhd->rihd", p, v))


def test_integration_md_ctypes_stub_runs_and_matches_the_mirror_module(dev):
    """INTEGRATION.md (Level 2) shows the ctypes stub a maintainer of the reference would paste into
    RowSelfAttention.forward.  Execute exactly that text against the library: it must agree with the mirror module bit
    for bit (same kernels, same order), so the document cannot drift from the C ABI."""
    import os
    import re
    from conftest import ROOT
    from rnamsm import modules
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\nimport ctypes, numpy, torch\n(.*?)```", src, re.S).group(0)[len("```python\n"):-3]
    code = code.replace('ctypes.CDLL("rna-msm_amd/rnamsm/librnamsm_hip.so")',
                        f'ctypes.CDLL("{os.path.join(ROOT, "rna-msm_amd", "rnamsm", "librnamsm_hip.so")}")')
    ns = {}
    exec(code, ns)                                            # the repo's own documentation, not external input
    torch.manual_seed(0)
    m = modules.RowSelfAttention(768, 12).to(dev).eval()
    x = torch.randn(6, 19, 1, 768, device=dev)
    want, want_p = m(x)
    got, got_p = ns["row_self_attention_forward"](m, x)
    assert torch.equal(got, want) and torch.equal(got_p, want_p)


def test_generic_mha_edge_shapes_against_oracle(dev):
    """Edge shapes of the MHA entry point in every route (fused, weights, masked; fp32 and f16x3): a single position (T = 1:
    the softmax over one key is 1, weights are all ones), a single batch element, T crossing the 128-row tile edge."""
    from rnamsm import modules as M
    for T, B, E, H in ((1, 3, 128, 2), (2, 1, 128, 2), (130, 2, 128, 2)):
        state = synthetic.make_state_dict(seed=5, embed_dim=E, num_layers=1, num_heads=H)
        prefix = "layers.0.row_self_attention.layer"
        mha = M.MultiheadAttention(E, H, self_attention=True)
        mha.load_state_dict({k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in state.items() if k.startswith(prefix + ".")},
                            strict=True)
        mha = mha.eval().to(dev)
        x = torch.from_numpy(synthetic.normal(f"mhae:{T}", 5, (T, B, E)).astype(np.float32))
        kpm = torch.zeros(B, T, dtype=torch.bool)
        if T > 2:
            kpm[0, T - 3:] = True
        st = O.to_torch_params(state, torch.float64)
        want, wts = O.multihead_self_attention(x.double(), st, prefix, H, key_padding_mask=kpm if T > 2 else None,
                                               return_weights=True)
        for mode, tol in (("f32", 5e-6), ("f16x3", 2e-5)):
            mha.gemm_dtype = mode
            kw = dict(key_padding_mask=kpm.to(dev)) if T > 2 else {}
            y, w = mha(x.to(dev), **kw)                                   # default: head-averaged weights
            assert rel_l2(y.cpu(), want) < tol and np.abs(w.cpu().numpy() - wts.mean(0).numpy()).max() < 5 * tol, (T, mode)
            y2, w2 = mha(x.to(dev), need_weights=False, **kw)             # fused kernel
            assert w2 is None and rel_l2(y2.cpu(), want) < tol, (T, mode)
            y3, w3 = mha(x.to(dev), need_head_weights=True, **kw)
            assert w3.shape == (H, B, T, T) and np.abs(w3.cpu().numpy() - wts.numpy()).max() < 5 * tol, (T, mode)
            if T == 1:
                assert float((w3 - 1).abs().max()) == 0.0


def test_small_m_gemm_tiles_run_side_by_side(dev):
    """Performance guard for EXPERIMENTS R4.8: a GEMM with one row panel and 18 column tiles (a lone small alignment's QKV) must
    take about as long as one with 6 (its tiles are independent and 238 CUs are idle) -- with padded XCD groups the hardware's
    in-turn CU assignment had stacked the 18 tiles on four CUs: 117 us against 26 us, a ratio of 4.5.  Bound 2.0 (measured
    1.1), on the per-launch time inside a queue of 200 launches; timing ratios of one process on one device are stable."""
    import time
    from rnamsm import ops
    M, K = 82, 768
    a = torch.randn(M, K, device=dev)

    def per_launch(N):
        w, b, out = torch.randn(N, K, device=dev) * 0.04, torch.zeros(N, device=dev), torch.empty(M, N, device=dev)
        for _ in range(20):
            ops.linear(a, w, b, out=out)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200):
                ops.linear(a, w, b, out=out)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
        return best
    narrow, wide, widest = per_launch(768), per_launch(2304), per_launch(3072)
    assert wide < 2.0 * narrow and widest < 2.0 * narrow, (narrow * 1e6, wide * 1e6, widest * 1e6)
