"""GPU parity tests of the 16-bit attention kernels (K4', K5', K6', K7' of include/rnamsm.h) through the C ABI.

Two references per case, both fp64 on the host:
  * "eff": the contraction evaluated on the values the planes actually hold (hi, or hi + lo).  What the kernel adds to
    that is fp32 accumulation and, for the x3 modes, the dropped lo*lo term (2^-16 relative for bf16 pairs, 2^-22 for
    fp16 pairs): a TIGHT bound that catches any wrong lane / swizzle / transposed-read mapping in every mode;
  * the contraction on the original fp32 operands, at the mode's stated accuracy (bf16 2^-9 operands,
    f16x3 ~2^-22 = fp32-grade).
Integer operands are exactly representable in both 16-bit formats, so row_logits16 / row_apply16 must be bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from rnamsm import synthetic

pytestmark = pytest.mark.gpu

# C >= 256 (with R >= 4) takes the 256x256-tile row kernels: ragged both ways, R not a multiple of 4, C = 256 exactly
ATT_SHAPES = [(1, 5, 2), (7, 33, 2), (34, 66, 2), (6, 19, 12), (3, 130, 2), (65, 40, 2), (257, 9, 2), (300, 20, 1), (16, 200, 3),
              (8, 300, 2), (5, 520, 1), (9, 257, 2), (4, 256, 1), (3, 300, 1),
              (64, 128, 12), (96, 70, 12)]       # thousands of blocks, several resident per CU: staging races show up here
# (split, fmt, tolerance vs eff operands, tolerance vs fp32 operands)
MODES = [(1, 0, 2e-6, 8e-3), (3, 1, 3e-6, 3e-6)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rnamsm import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(name, shape, scale=1.0):
    return torch.from_numpy((scale * synthetic.normal(name, 23, shape)).astype(np.float32))


def _planes(x_dev, split, fmt):
    """fp32 device tensor -> ((hi, lo|None) int16 planes, fp64 host tensor of the values they hold)."""
    from rnamsm import ops
    hi, lo = ops.split_bf16(x_dev, want_lo=(split == 3), fmt=fmt)
    ht = torch.float16 if fmt == 1 else torch.bfloat16
    eff = hi.view(ht).double()
    if lo is not None:
        eff = eff + lo.view(ht).double()
    return (hi, lo), eff.cpu()


def _views(pl, a, b):
    return (pl[0][:, a:b], None if pl[1] is None else pl[1][:, a:b])


@pytest.mark.parametrize("split,fmt,tol_eff,tol_f32", MODES)
@pytest.mark.parametrize("R,C,H", ATT_SHAPES)
def test_row_attention_16bit(dev, R, C, H, split, fmt, tol_eff, tol_f32):
    from rnamsm import ops
    D = 64 * H
    qkv = _rand(f"row16.{R}.{C}", (R * C, 3 * D))
    scale = ops.row_scaling(R)                  # applied by the kernel to the fp32 logits; q stays unscaled in the planes
    pl, eff = _planes(qkv.to(dev), split, fmt)

    def parts(t):
        t = t.double()
        return t[:, :D].view(R, C, H, 64), t[:, D:2 * D].view(R, C, H, 64), t[:, 2 * D:].view(R, C, H, 64)

    partial, nsplit = ops.row_logits16(_views(pl, 0, D), _views(pl, D, 2 * D), R, C, H, fmt=fmt, scale=scale)
    got = partial.sum(0).cpu()
    qe, ke, ve = parts(eff)
    q, k, v = parts(qkv)
    assert rel_l2(got, scale * torch.einsum("rihd,rjhd->hij", qe, ke)) < tol_eff
    assert rel_l2(got, scale * torch.einsum("rihd,rjhd->hij", q, k)) < tol_f32

    PS = 4096.0                                 # plane scale used by rnamsm_forward (keeps fp16 lo planes normal)
    probs, pp = ops.softmax_rows_planes(partial, split=split, fmt=fmt, plane_scale=PS)
    want_p = torch.softmax(partial.sum(0).double().cpu(), -1)
    assert np.abs(probs.cpu().numpy() - want_p.numpy()).max() < 2e-6
    ht = torch.float16 if fmt == 1 else torch.bfloat16
    ldp = (C + 63) // 64 * 64
    assert pp[0].shape == (H * C, ldp)
    assert torch.equal(pp[0][:, :C].reshape(H, C, C), (probs * PS).to(ht).view(torch.int16))    # hi = round(P * 2^12)
    assert int(pp[0][:, C:].abs().max() if ldp > C else 0) == 0                                  # zero tail
    p_eff = pp[0].view(ht).double()
    if split == 3:
        assert int(pp[1][:, C:].abs().max() if ldp > C else 0) == 0
        p_eff = p_eff + pp[1].view(ht).double()
        assert np.abs(p_eff[:, :C].cpu().numpy().reshape(H, C, C) / PS - probs.double().cpu().numpy()).max() < (2e-5 if fmt == 0 else 3e-7)
    p_eff = p_eff[:, :C].reshape(H, C, C).cpu() / PS

    ctx = ops.row_apply16(pp, _views(pl, 2 * D, 3 * D), R, C, H, fmt=fmt, out_scale=1.0 / PS).cpu()
    assert rel_l2(ctx, torch.einsum("hij,rjhd->rihd", p_eff, ve).reshape(R * C, D)) < tol_eff
    assert rel_l2(ctx, torch.einsum("hij,rjhd->rihd", want_p, v).reshape(R * C, D)) < tol_f32


@pytest.mark.parametrize("R,C", [(5, 150), (6, 300)])
@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
def test_row_kernels_16bit_are_exact_on_integers(dev, split, fmt, R, C):
    """Small integers are exact in bf16 and fp16 (lo planes = 0) and every partial sum is exact in fp32: the result must
    equal the integer contraction bit for bit.  The data are asymmetric in every index, so a transposed tile, a swapped
    k order inside a fragment or a wrong swizzle cannot cancel."""
    from rnamsm import ops
    H = 2
    D = 64 * H
    n = R * C * 3 * D
    qkv = torch.from_numpy((((np.arange(n, dtype=np.int64) * 7 + (np.arange(n, dtype=np.int64) // 191) * 3) % 13) - 6)
                           .astype(np.float32)).view(R * C, 3 * D)
    pl, eff = _planes(qkv.to(dev), split, fmt)
    assert torch.equal(eff.float(), qkv)
    partial, _ = ops.row_logits16(_views(pl, 0, D), _views(pl, D, 2 * D), R, C, H, fmt=fmt)
    q = qkv[:, :D].double().view(R, C, H, 64); k = qkv[:, D:2 * D].double().view(R, C, H, 64)
    v = qkv[:, 2 * D:].double().view(R, C, H, 64)
    assert torch.equal(partial.sum(0).cpu().double(), torch.einsum("rihd,rjhd->hij", q, k))
    # integer "probabilities" handed to row_apply16 as planes [H*C, ldp]
    ldp = (C + 63) // 64 * 64
    m = H * C * ldp
    pint = torch.from_numpy((((np.arange(m, dtype=np.int64) * 5 + (np.arange(m, dtype=np.int64) // 67)) % 7) - 3)
                            .astype(np.float32)).view(H * C, ldp)
    pint[:, C:] = 0
    pp, _ = _planes(pint.to(dev), split, fmt)
    ctx = ops.row_apply16(pp, _views(pl, 2 * D, 3 * D), R, C, H, fmt=fmt).cpu()
    want = torch.einsum("hij,rjhd->rihd", pint[:, :C].double().view(H, C, C), v).reshape(R * C, D)
    assert torch.equal(ctx.double(), want)


@pytest.mark.parametrize("R,C,H", [(8, 300, 2), (5, 520, 1), (9, 257, 2), (4, 256, 1), (3, 300, 1), (64, 512, 12), (1, 256, 1)])
def test_plain_bf16_row_kernels_with_64_deep_tiles(dev, R, C, H):
    """Plain bf16 at C >= 256: rnamsm_row_apply16 stages 64 keys per tile (whole cache lines per P row), the logits run on
    row_logits16q_kernel from C >= 384 -- the shapes around those switches against the usual references and bounds, plus the integer
    case.  (The 32-key apply tiles and the 64-deep 32x32x16 logits variant of the removed knob "row16_bk64" went in round 6.)"""
    test_row_attention_16bit(dev, R, C, H, 1, 0, 2e-6, 8e-3)
    if H == 2:
        test_row_kernels_16bit_are_exact_on_integers(dev, 1, 0, R, C)


def _col_ref(q, k, v):
    p = torch.softmax(torch.einsum("ichd,jchd->hcij", q, k), -1)
    return torch.einsum("hcij,jchd->ichd", p, v), p


@pytest.mark.parametrize("split,fmt,tol_eff,tol_f32", MODES)
@pytest.mark.parametrize("R,C,H", ATT_SHAPES)
def test_col_attention_16bit(dev, R, C, H, split, fmt, tol_eff, tol_f32):
    """P is rounded to the operand format inside the kernel (bf16: 2^-9 per element, pairs: 2^-17 / 2^-22), so the
    plain-bf16 bound vs the eff reference is the P rounding, not the q/k/v rounding."""
    from rnamsm import ops
    D = 64 * H
    qkv = _rand(f"col16.{R}.{C}", (R * C, 3 * D))
    pl, eff = _planes(qkv.to(dev), split, fmt)
    ctx = ops.col_attn16(_views(pl, 0, D), _views(pl, D, 2 * D), _views(pl, 2 * D, 3 * D), R, C, H, fmt=fmt, scale=0.125).cpu()

    def ref(t):
        t = t.double()
        return _col_ref(0.125 * t[:, :D].view(R, C, H, 64), t[:, D:2 * D].view(R, C, H, 64), t[:, 2 * D:].view(R, C, H, 64))[0]

    assert rel_l2(ctx, ref(eff).reshape(R * C, D)) < (3e-3 if split == 1 else tol_eff)
    assert rel_l2(ctx, ref(qkv).reshape(R * C, D)) < tol_f32


@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
def test_col_attention_16bit_selects_the_right_value_rows(dev, split, fmt):
    """One-hot attention: query i = +-e_(i % 64) and key perm[i] carries +-40 on that axis, so ctx[i] must be v[perm[i]]
    to rounding; with integer v[j][d] = (3j + 5d) % 251 (exact in both formats) a wrong key order in the transposed V
    read or in the P fragment shows up as a wrong integer."""
    from rnamsm import ops
    R, C, H = 96, 2, 1
    q = torch.zeros(R, C, 64); k = torch.zeros(R, C, 64); v = torch.zeros(R, C, 64)
    perm = (np.arange(R) * 37 + 11) % R                        # query i attends key perm[i]
    for i in range(R):
        q[i, :, i % 64] = 1.0 if i < 64 else -1.0              # 96 distinct directions: +-e_(i%64), sign by i // 64 ...
    for i in range(R):
        j = int(perm[i])
        k[j, :, i % 64] += 40.0 if i < 64 else -40.0
    # a key can be the target of one query only (perm is a bijection); cross terms: query i (dir e_m, sign s) sees key
    # perm[i'] with i' % 64 == m, i' != i: score -40 -> never the maximum.
    v[:] = ((3 * torch.arange(R)[:, None, None] + 5 * torch.arange(64)[None, None, :]) % 251).float()
    qkv = torch.cat([q, k, v], -1).view(R * C, 192)
    pl, _ = _planes(qkv.to(dev), split, fmt)
    ctx = ops.col_attn16(_views(pl, 0, 64), _views(pl, 64, 128), _views(pl, 128, 192), R, C, H, fmt=fmt).cpu().view(R, C, 64)
    want = v[torch.from_numpy(perm.astype(np.int64))]
    assert np.abs(ctx.numpy() - want.numpy()).max() < 1e-3


@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
def test_col_attention_16bit_online_softmax_rescale(dev, split, fmt):
    from rnamsm import ops
    R, C, H = 200, 3, 1
    for spike_row in (190, 0, 65):
        qkv = _rand(f"spike16.{spike_row}", (R * C, 192))
        qkv[:, :64] = qkv[:, :64].abs() * 0.2
        qkv[:, 64:128].view(R, C, 64)[spike_row] = 6.0
        pl, eff = _planes(qkv.to(dev), split, fmt)
        ctx = ops.col_attn16(_views(pl, 0, 64), _views(pl, 64, 128), _views(pl, 128, 192), R, C, H, fmt=fmt).cpu()
        e = eff.double()
        want = _col_ref(e[:, :64].view(R, C, 1, 64), e[:, 64:128].view(R, C, 1, 64), e[:, 128:].view(R, C, 1, 64))[0]
        assert rel_l2(ctx, want.reshape(R * C, 64)) < (3e-3 if split == 1 else 3e-6)


@pytest.mark.parametrize("split,fmt", [(1, 0)])
@pytest.mark.parametrize("R", [40, 200, 300])
def test_col_attention_16bit_fast_loop_falls_back_when_a_score_overflows_its_reference(dev, split, fmt, R):
    """The bf16 column kernel fixes every query's softmax reference from its first 32 keys (no running maximum) and redoes a
    column with the tracked loop when a row sum reaches 2^96.  Here one key far down the column scores ~390 log2 units above
    everything in the first tile for the queries of ONE wave-sized group: the fallback must produce the exact softmax (the spike
    key's value row), the untouched columns / heads must equal the tracked kernel's output, and forcing the tracked loop
    ("attn16" = 5) must agree everywhere."""
    from rnamsm import ops, _lib
    lib = _lib.load()
    C, H = 3, 2
    D = 64 * H
    qkv = _rand(f"ovf16.{R}", (R * C, 3 * D), 0.5)
    spike = R - 3                                           # beyond the first tile for every R here
    qv = qkv.view(R, C, 3 * D)
    qv[:, 1, 0:64] = 0.0
    qv[5:9, 1, 0] = 48.0                                    # queries 5..8 of column 1, head 0: q = 48 e_0
    qv[:, 1, D:D + 64][:, 0] = 0.0                          # every key of that column / head: k_0 = 0 ...
    qv[spike, 1, D] = 45.0                                  # ... but the spike key: score 2160 * 0.125 = 270 nats
    pl, eff = _planes(qkv.to(dev), split, fmt)
    args = (_views(pl, 0, D), _views(pl, D, 2 * D), _views(pl, 2 * D, 3 * D), R, C, H)
    got = ops.col_attn16(*args, fmt=fmt, scale=0.125).cpu()
    try:
        _lib.check(lib.rnamsm_set_param(b"attn16", 5))
        tracked = ops.col_attn16(*args, fmt=fmt, scale=0.125).cpu()
    finally:
        _lib.check(lib.rnamsm_set_param(b"attn16", 1))
    e = eff.double()
    want = _col_ref(0.125 * e[:, :D].view(R, C, H, 64), e[:, D:2 * D].view(R, C, H, 64), e[:, 2 * D:].view(R, C, H, 64))[0].reshape(R * C, D)
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) < (3e-3 if split == 1 else 4e-5)
    assert rel_l2(tracked, want) < (3e-3 if split == 1 else 4e-5)
    gv, wv = got.view(R, C, D), want.view(R, C, D)
    v_spike = e[:, 2 * D:].view(R, C, D)[spike, 1, :64]
    for i in range(5, 9):                                   # the spiked queries see the spike key only
        assert np.abs(gv[i, 1, :64].numpy() - v_spike.numpy()).max() < 2e-2 * max(1.0, float(v_spike.abs().max()))
        assert np.abs(gv[i, 1, :64].numpy() - wv[i, 1, :64].numpy()).max() < 1e-2


@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
@pytest.mark.parametrize("R,C,H", [(7, 33, 2), (130, 5, 2), (300, 4, 1), (256, 3, 2)])
def test_col_attention_16bit_plane_outputs_equal_the_rounded_fp32_output(dev, split, fmt, R, C, H):
    """The forward reads the context as 16-bit planes written through the kernel's LDS-transposed epilogue: hi must be the
    fp32 output rounded to the format and hi + lo must reproduce it to the pair's precision."""
    from rnamsm import ops
    D = 64 * H
    qkv = _rand(f"colpl.{R}.{C}", (R * C, 3 * D))
    pl, _ = _planes(qkv.to(dev), split, fmt)
    args = (_views(pl, 0, D), _views(pl, D, 2 * D), _views(pl, 2 * D, 3 * D), R, C, H)
    f32 = ops.col_attn16(*args, fmt=fmt, scale=0.125)
    hi, lo = ops.col_attn16(*args, fmt=fmt, scale=0.125, out_planes=True)
    ht = torch.float16 if fmt == 1 else torch.bfloat16
    # (the fp32 and the plane instance are two compilations of the kernel: their fp32 values may differ in the last bit, so
    # the bars carry one fp32 ulp / one 16-bit rounding flip at a tie)
    u16 = 2.0 ** (-8 if fmt == 0 else -11)                   # unit roundoff of the 16-bit format (8 / 11 significand bits)
    d_hi = (hi.view(ht).double() - f32.double()).abs()
    assert bool((d_hi <= u16 * f32.double().abs() * (1 + 1e-3) + 1e-7).all())                  # hi = the output rounded to the format
    if split == 3:
        assert lo is not None
        back = hi.view(ht).double() + lo.view(ht).double()
        err = (back - f32.double()).abs()
        # hi + lo reproduces the fp32 output to the pair's precision (2^-16 bf16, 2^-21 fp16 relative; fp16 lo below 2^-14 is
        # subnormal: absolute step 6e-8).  Before round 4 one element in ~30 000 was off by a whole 16-bit ulp (fp contraction
        # across the split, half16.h: pinned).
        bar = (2.0 ** -16 if fmt == 0 else 2.0 ** -21) * f32.double().abs() + 2e-7
        assert bool((err <= bar).all()), float((err / bar).max())
    else:
        assert lo is None


@pytest.mark.parametrize("split", [1])
@pytest.mark.parametrize("R,C,H", [(7, 33, 2), (40, 9, 2), (130, 5, 2), (300, 4, 1), (256, 3, 2), (64, 128, 12)])
def test_col_attention_16bit_with_prescaled_q(dev, split, R, C, H):
    """rnamsm_col_attn16_prescaled (what rnamsm_forward runs in the bf16 modes): the q planes hold q * dh^-0.5 * log2(e), rounded
    once; the kernel exponentiates the scores as they come (no reference, no multiply).  Against the fp64 softmax of the values
    the planes hold, at the bounds of the unscaled entry point; and forcing the online-softmax loop ("attn16" = 5) agrees."""
    from rnamsm import ops, _lib
    lib = _lib.load()
    D = 64 * H
    qkv = _rand(f"colpre.{R}.{C}", (R * C, 3 * D))
    c2 = 0.125 * 1.4426950408889634
    pre = qkv.clone()
    pre[:, :D] *= c2
    pl, eff = _planes(pre.to(dev), split, 0)
    args = (_views(pl, 0, D), _views(pl, D, 2 * D), _views(pl, 2 * D, 3 * D), R, C, H)
    got = ops.col_attn16(*args, fmt=0, prescaled=True).cpu()
    try:
        _lib.check(lib.rnamsm_set_param(b"attn16", 5))
        tracked = ops.col_attn16(*args, fmt=0, prescaled=True).cpu()
    finally:
        _lib.check(lib.rnamsm_set_param(b"attn16", 1))
    e = eff.double()
    ln2 = 0.6931471805599453                                 # the planes' q is in log2 units: softmax_e(q' k ln 2)
    want = _col_ref(ln2 * e[:, :D].view(R, C, H, 64), e[:, D:2 * D].view(R, C, H, 64), e[:, 2 * D:].view(R, C, H, 64))[0].reshape(R * C, D)
    tol = 3e-3 if split == 1 else 4e-5
    assert rel_l2(got, want) < tol and rel_l2(tracked, want) < tol
    # and against the fp32 operands at the mode's accuracy
    t = qkv.double()
    want32 = _col_ref(0.125 * t[:, :D].view(R, C, H, 64), t[:, D:2 * D].view(R, C, H, 64), t[:, 2 * D:].view(R, C, H, 64))[0].reshape(R * C, D)
    assert rel_l2(got, want32) < (8e-3 if split == 1 else 4e-5)


def test_16bit_attention_rejects_inconsistent_planes(dev):
    from rnamsm import ops, _lib
    R, C, H = 4, 8, 1
    x = _rand("rej", (R * C, 192)).to(dev)
    hi, lo = ops.split_bf16(x, fmt=0)
    with pytest.raises(_lib.RnamsmError):          # lo given for q but not for k
        ops.row_logits16((hi[:, :64], lo[:, :64]), (hi[:, 64:128], None), R, C, H)
    with pytest.raises(_lib.RnamsmError):          # fp16 needs hi/lo pairs
        ops.col_attn16((hi[:, :64], None), (hi[:, 64:128], None), (hi[:, 128:], None), R, C, H, fmt=1)


def test_forward_attn16_switch(dev):
    """rnamsm_set_param("attn16"): the f16x3 forward with 16-bit attention must stay within the fp32-grade distance of
    the same forward with exact-fp32 attention (both inside the 1e-4 bar of tests/test_gpu_forward.py)."""
    from rnamsm import _lib
    from rnamsm.model import MSATransformer
    lib = _lib.load()
    state = synthetic.make_state_dict(seed=0)
    m = MSATransformer(num_layers=10).cuda().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    toks = torch.from_numpy(synthetic.make_tokens(24, 70, 3)).cuda()
    m.gemm_dtype = "f16x3"
    try:
        assert lib.rnamsm_get_param(b"attn16") == 1
        a = m.forward_one(toks)
        _lib.check(lib.rnamsm_set_param(b"attn16", 0))
        b = m.forward_one(toks)
    finally:
        _lib.check(lib.rnamsm_set_param(b"attn16", 1))
    assert not torch.equal(a["emb"], b["emb"])                 # the switch really changes the kernels
    assert rel_l2(a["emb"].cpu().numpy(), b["emb"].cpu().numpy()) < 2e-5
    assert np.abs(a["atp"].cpu().numpy() - b["atp"].cpu().numpy()).max() < 2e-5


def _int_tensor(shape, mul, mod, off, dev):
    n = int(np.prod(shape))
    i = torch.arange(n, device=dev, dtype=torch.int64)
    return (((i * mul + (i // 191) * 3) % mod) - off).to(torch.float32).view(*shape)


@pytest.mark.parametrize("split,fmt", [(1, 0), (3, 1)])
def test_full_grid_16bit_kernels_equal_the_exact_kernels_on_integers(dev, split, fmt):
    """BASELINE configs[2] grid sizes (tens of thousands of blocks, every CU holding its maximum of resident blocks): on
    small-integer operands the 16-bit kernels and the exact-fp32 kernels are both exact, so their outputs must be EQUAL
    -- GEMM (256x256 software-pipelined tiles, QKV- and fc2-shaped), row logits (256x256 tiles in the hi/lo modes) and
    row apply (256x256 tiles).  A staging race or a wrong tile edge anywhere in the grid breaks the equality."""
    from rnamsm import ops
    R, C, H = 256, 512, 12
    D = 64 * H
    T = R * C
    a = _int_tensor((T, D), 7, 13, 6, dev)
    for N, K in ((3 * D, D), (D, 4 * D)):
        w = _int_tensor((N, K), 5, 11, 5, dev)
        x = a if K == D else _int_tensor((T, K), 3, 7, 3, dev)
        want = ops.linear(x, w)
        got = ops.linear_planes(ops.split_bf16(x, want_lo=split == 3, fmt=fmt), ops.split_bf16(w, want_lo=split == 3, fmt=fmt), fmt=fmt)
        assert torch.equal(got, want), (N, K)
        del want, got, w
    qkv = _int_tensor((T, 3 * D), 7, 5, 2, dev)                       # |q.k| sums stay far below 2^24
    pl = ops.split_bf16(qkv, want_lo=split == 3, fmt=fmt)
    v = lambda lo_, hi_: (pl[0][:, lo_:hi_], None if pl[1] is None else pl[1][:, lo_:hi_])
    p16, _ = ops.row_logits16(v(0, D), v(D, 2 * D), R, C, H, fmt=fmt)
    p32, _ = ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H)
    assert torch.equal(p16.sum(0), p32.sum(0))
    ldp = (C + 63) // 64 * 64
    pint = _int_tensor((H * C, ldp), 5, 7, 3, dev)
    pint[:, C:] = 0
    pp = ops.split_bf16(pint, want_lo=split == 3, fmt=fmt)
    c16 = ops.row_apply16(pp, v(2 * D, 3 * D), R, C, H, fmt=fmt)
    c32 = ops.row_apply(pint[:, :C].reshape(H, C, C).contiguous(), qkv[:, 2 * D:], R, C, H)
    assert torch.equal(c16, c32)


def test_full_grid_col_attention_16bit_against_the_exact_kernel(dev):
    """Same idea for column attention (its softmax rules out bit equality): at the BASELINE configs[2] grid the f16x3
    kernel must sit at fp32-grade distance from the exact-fp32 kernel on operands both represent exactly -- any block
    that read a K/V chunk before it had landed would be off by orders of magnitude more."""
    from rnamsm import ops
    R, C, H = 256, 512, 12
    D = 64 * H
    qkv = (_int_tensor((R * C, 3 * D), 7, 9, 4, dev) / 4.0).contiguous()         # multiples of 1/4 in [-1, 1]: exact in fp16
    pl = ops.split_bf16(qkv, fmt=1)
    v = lambda lo_, hi_: (pl[0][:, lo_:hi_], pl[1][:, lo_:hi_])
    c16 = ops.col_attn16(v(0, D), v(D, 2 * D), v(2 * D, 3 * D), R, C, H, fmt=1, scale=0.125)
    q32 = qkv.clone(); q32[:, :D] *= 0.125
    c32 = ops.col_attn(q32[:, :D], q32[:, D:2 * D], q32[:, 2 * D:], R, C, H)
    diff = (c16 - c32).abs()
    assert float(diff.max()) < 2e-5 and float(diff.double().norm() / c32.double().norm()) < 2e-6


@pytest.mark.parametrize("split,fmt,tol", [(1, 0, 3e-3), (3, 1, 3e-6)])
def test_padding_masks_in_the_16bit_attention_kernels(dev, split, fmt, tol):
    """§8 f2 on the 16-bit kernels: q planes zeroed at padded tokens (rnamsm_zero_plane_rows), -10000 on keys whose
    first-row token is padded (row softmax), -10000 on padded keys of a column (col_attn16) -- against the reference's
    direct-path arithmetic (modules.py:767-785, 911-915) evaluated in fp64 on the values the planes hold."""
    from rnamsm import ops
    R, C, H = 37, 45, 2
    D = 64 * H
    qkv = _rand("pad16", (R * C, 3 * D))
    pad = torch.zeros(R, C, dtype=torch.bool)
    pad[:, 40:] = True            # padded columns (in every row, so also in row 0)
    pad[30:, :] = True            # padded alignment rows
    pad[3, 7] = True              # an isolated pad
    mask = pad.to(torch.uint8).to(dev).contiguous()
    pl, eff = _planes(qkv.to(dev), split, fmt)
    ht = torch.float16 if fmt == 1 else torch.bfloat16
    # q *= 1 - padding_mask
    ops.zero_plane_rows(_views(pl, 0, D), mask.view(-1), D)
    got_q = pl[0][:, :D].view(ht).double().cpu() + (pl[1][:, :D].view(ht).double().cpu() if pl[1] is not None else 0)
    want_q = eff[:, :D] * (~pad).view(-1, 1)
    assert torch.equal(got_q, want_q)
    assert torch.equal(pl[0][:, D:].cpu(), _planes(qkv.to(dev), split, fmt)[0][0][:, D:].cpu())      # k, v untouched
    q = want_q.view(R, C, H, 64); k = eff[:, D:2 * D].view(R, C, H, 64); v = eff[:, 2 * D:].view(R, C, H, 64)
    # tied row attention with the first-row key mask
    scale = ops.row_scaling(R)
    partial, _ = ops.row_logits16(_views(pl, 0, D), _views(pl, D, 2 * D), R, C, H, fmt=fmt, scale=scale)
    probs, pp = ops.softmax_rows_planes(partial, split=split, fmt=fmt, key_mask=mask[0].contiguous(), plane_scale=4096.0)
    logits = scale * torch.einsum("rihd,rjhd->hij", q, k)
    want_p = torch.softmax(logits.masked_fill(pad[0][None, None, :], -10000.0), -1)
    assert np.abs(probs.cpu().numpy() - want_p.numpy()).max() < (2e-3 if split == 1 else 2e-6)
    assert float(probs[:, :, 40:].max()) == 0.0
    # column attention with the per-column key mask (q NOT zeroed there: use the un-zeroed q planes)
    pl2, eff2 = _planes(qkv.to(dev), split, fmt)
    ctx = ops.col_attn16(_views(pl2, 0, D), _views(pl2, D, 2 * D), _views(pl2, 2 * D, 3 * D), R, C, H, fmt=fmt, scale=0.125,
                         pad_mask=mask).cpu()
    q2 = 0.125 * eff2[:, :D].view(R, C, H, 64)
    wc = torch.einsum("ichd,jchd->hcij", q2, k).masked_fill(pad.t()[None, :, None, :], -10000.0)
    want_ctx = torch.einsum("hcij,jchd->ichd", torch.softmax(wc, -1), v).reshape(R * C, D)
    assert rel_l2(ctx, want_ctx) < tol
