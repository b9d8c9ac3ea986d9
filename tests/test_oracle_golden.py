"""Pins the oracle (oracle/msm_oracle.py, oracle/tokenizer_oracle.py) against fixtures produced by
the reference itself (tests/golden/make_golden.py).  CPU only.

Tolerances: the oracle is an independent fp32 restatement, so it differs from the reference by
fp32 summation order only.  The reference's own fp32-vs-fp64 noise floor on these inputs is
emb rel-L2 ~1e-6, atp max-abs ~6e-6 (printed by make_golden.py); the oracle must sit within
1e-5 rel-L2 / 2e-5 max-abs of the reference and, run in fp64, within the same distance of the
reference's fp64 run.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden, rel_l2
from oracle import msm_oracle as O
from oracle import tokenizer_oracle as TO
from rnamsm import synthetic

OP_CASES = ["d128_r7_c33", "d128_r7_c33_chunk", "d128_r1_c5", "d128_r34_c66", "d768_r6_c19"]
FWD_CASES = ["m8_c17", "m8_c17_chunk", "m16_c33", "m5_c41"]


def test_tokens_2drb1_bit_exact():
    text = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read()
    want = golden("tokens_2DRB_1_first64.npz")["tokens"]
    got = TO.encode_msa(text)
    assert got.dtype == np.int64 and got.shape == want.shape
    assert np.array_equal(got, want)
    assert (got[:, 0] == 0).all()


def test_tokens_every_character_class_bit_exact():
    text = open(os.path.join(GOLDEN, "synthetic_chars.a2m_msa2")).read()
    want = golden("tokens_synthetic_chars.npz")["tokens"]
    assert np.array_equal(TO.encode_msa(text), want)


def test_token_errors_match_reference():
    meta = json.load(open(os.path.join(GOLDEN, "tokens_meta.json")))
    assert meta["vocab"] == TO.TOK_TO_IDX
    assert meta["errors"]["ragged"].startswith("AssertionError")
    assert meta["errors"]["invalid_char"] == "ValueError:Invalid tokens in input"
    with pytest.raises(AssertionError):
        TO.encode_msa(">a\nACGU\n>b\nACG\n")
    with pytest.raises(ValueError, match="Invalid tokens in input"):
        TO.encode_msa(">a\nACGE\n>b\nACGU\n")


@pytest.mark.parametrize("name", OP_CASES)
def test_ops_match_reference(name):
    g = golden(f"op_{name}.npz")
    D, H, R, C, max_tokens = (int(v) for v in g["meta"])
    mt = None if max_tokens < 0 else max_tokens
    state = O.to_torch_params(synthetic.make_state_dict(seed=7, embed_dim=D, num_layers=1, num_heads=H))
    x = torch.from_numpy(synthetic.normal(f"x:{name}", 7, (R, C, 1, D)).astype(np.float32))[:, :, 0]
    out, probs = O.row_attention(x, state, "layers.0.row_self_attention.layer", H, mt)
    assert rel_l2(out, g["row_out"]) < 1e-5
    assert np.abs(probs.numpy() - g["row_probs"]).max() < 2e-5
    out, probs = O.col_attention(x, state, "layers.0.column_self_attention.layer", H, return_probs=True)
    assert rel_l2(out, g["col_out"]) < 1e-5
    if "col_probs" in g:
        assert np.abs(probs.numpy() - g["col_probs"]).max() < 2e-5
    if "ffn_out" in g:
        assert rel_l2(O.ffn(x, state, "layers.0.feed_forward_layer.layer"), g["ffn_out"]) < 1e-5
        pre = "layers.0.feed_forward_layer"
        blk = x + O.ffn(O.layer_norm(x, state[f"{pre}.layer_norm.weight"], state[f"{pre}.layer_norm.bias"]),
                        state, f"{pre}.layer")
        assert rel_l2(blk, g["ffn_block_out"]) < 1e-5
        pre = "layers.0.row_self_attention"
        ln = O.layer_norm(x, state[f"{pre}.layer_norm.weight"], state[f"{pre}.layer_norm.bias"])
        assert rel_l2(ln, g["ln_out"]) < 1e-5
    y, rp = O.axial_layer(x, state, 0, H, mt)
    assert rel_l2(y, g["layer_out"]) < 1e-5
    assert np.abs(rp.numpy() - g["layer_row_probs"]).max() < 2e-5


@pytest.fixture(scope="module")
def full_state():
    return synthetic.make_state_dict(seed=0)


@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_matches_reference(name, full_state):
    g = golden(f"forward_{name}.npz")
    M, C, max_tokens = (int(v) for v in g["meta"])
    toks = synthetic.make_tokens(M, C, {"m8_c17": 0, "m8_c17_chunk": 0, "m16_c33": 1, "m5_c41": 2}[name])
    assert np.array_equal(toks, g["tokens"])          # the token generator itself is part of the fixture
    params = O.to_torch_params(full_state)
    res = O.forward(torch.from_numpy(toks), params, max_tokens=max_tokens)
    emb, atp = O.pack_outputs(res)
    assert emb.shape == g["emb"].shape and atp.shape == g["atp"].shape
    assert rel_l2(emb, g["emb"]) < 1e-5
    assert np.abs(atp.numpy() - g["atp"]).max() < 2e-5
    assert np.abs(res["row_attentions"][0].numpy() - g["attn_full_layer0"]).max() < 2e-5
    # f1 contact head and f4 LM head of the same forward (model.py:402, 412-414)
    c = O.contact_head(res["row_attentions"], params["contact_head.regression.weight"], params["contact_head.regression.bias"])
    assert c.shape == g["contacts"].shape and np.abs(c.numpy() - g["contacts"]).max() < 2e-6
    logits = O.lm_head(res["representation"], params)
    assert logits.shape == g["logits"].shape and rel_l2(logits, g["logits"]) < 1e-5


@pytest.mark.parametrize("name", ["m8_c17", "m16_c33"])
def test_forward_fp64_matches_reference_fp64(name, full_state):
    g = golden(f"forward_{name}.npz")
    g64 = golden(f"forward_{name}_fp64.npz")
    params = O.to_torch_params(full_state, torch.float64)
    res = O.forward(torch.from_numpy(g["tokens"]), params)
    emb, atp = O.pack_outputs(res)
    assert rel_l2(emb, g64["emb"]) < 1e-6              # the fixture is a float32-weights fp64 run
    assert np.abs(atp.numpy() - g64["atp"]).max() < 1e-6


def test_shipped_output_contract():
    info = json.load(open(os.path.join(GOLDEN, "shapes_2DRB_1.json")))
    assert info["emb"] == {"shape": [35, 768], "dtype": "float32", "fortran_order": False, "row_sum_max": None}
    assert info["atp"]["shape"] == [120, 35, 35] and info["atp"]["dtype"] == "float32"
    assert info["atp"]["row_sum_max"] <= 1.0 + 1e-5   # <cls> column stripped => rows sum to < 1


def test_rejects_more_than_1024_rows_and_padding(full_state):
    params = {k: torch.zeros(1) for k in ()}
    with pytest.raises(RuntimeError, match="maximum MSA"):
        O.embed(torch.zeros(1025, 4, dtype=torch.int64), {})


def test_padded_ragged_batch_matches_reference(full_state):
    """padding-mask semantics (zeroed embeddings and q, -10000 key fills) on a B=2 ragged batch."""
    g = golden("forward_padded_b2.npz")
    params = O.to_torch_params(full_state)
    for b in range(2):
        res = O.forward(torch.from_numpy(g["tokens"][b]), params, force_mask=True)
        assert rel_l2(res["representation"], g["rep10"][b]) < 1e-5
        assert np.abs(res["row_attentions"].numpy() - g["row_attentions"][b]).max() < 2e-5
        x0 = O.embed(torch.from_numpy(g["tokens"][b]), params)
        assert rel_l2(x0, g["rep0"][b]) < 1e-5


@pytest.mark.parametrize("name", ["t37_b3_e128", "t70_b2_e768"])
def test_generic_mha_matches_reference(name):
    g = golden(f"mha_{name}.npz")
    T, B, E, H = (int(v) for v in g["meta"])
    st = O.to_torch_params(synthetic.make_state_dict(seed=11, embed_dim=E, num_layers=1, num_heads=H))
    x = torch.from_numpy(synthetic.normal(f"mha:{name}", 11, (T, B, E)).astype(np.float32))
    y = O.multihead_self_attention(x, st, "layers.0.row_self_attention.layer", H)
    assert rel_l2(y, g["out"]) < 1e-5


# ---------------------------------------------------------------------------- round-2 fixtures (make_golden_r2.py)
def test_padded_batch_on_the_chunked_path_matches_reference(full_state):
    """SURVEY §8 f2 "chunk quirk" (modules.py:717-750): with R*C > max_tokens_per_msa every row chunk is filled with
    -10000 from ITS first row and the filled slabs are summed.  The fixture holds the reference's chunked AND direct
    outputs of the same padded B=2 batch; they differ by 0.95 (a pad on a chunk-starting row masks that key) and by
    3.6e-4 (an all-padded chunk start shifts every logit by -10000), so this pins both semantics of the oracle."""
    g = golden("forward_padded_b2_chunked.npz")
    params = O.to_torch_params(full_state)
    mt = int(g["max_tokens"])
    for b in range(2):
        toks = torch.from_numpy(g["tokens"][b])
        res = O.forward(toks, params, max_tokens=mt, force_mask=True)
        assert rel_l2(res["representation"], g["rep10_chunked"][b]) < 3e-5
        d = np.abs(res["row_attentions"].numpy() - g["row_attentions_chunked"][b])
        if b == 0:
            assert d.max() < 2e-5
        else:
            # element 1 has an all-padded chunk start: its logits carry -10000 and are quantised to fp32's 9.8e-4 step
            # there, so any fp32 re-ordering upstream can flip a rounding: bounded by one step, and ~13x closer on
            # average than the direct semantics are (8.6e-6)
            assert d.max() < 1e-3 and d.mean() < 2e-6
            assert d[0].max() == 0.0 or d[0].max() < 2e-6          # layer 0: same inputs, same quantisation
        res = O.forward(toks, params, force_mask=True)
        assert rel_l2(res["representation"], g["rep10_direct"][b]) < 1e-5
        assert np.abs(res["row_attentions"].numpy() - g["row_attentions_direct"][b]).max() < 2e-5
    d = np.abs(g["row_attentions_chunked"] - g["row_attentions_direct"])
    assert d[0].max() > 0.5 and 1e-5 < d[1].max() < 1e-2          # where the two paths diverge, and by how much


def test_oracle_bf16_mode_reproduces_the_reference_bf16_drift(full_state):
    """BASELINE configs[4]'s yardstick: the reference model in .bfloat16() (fixture) against the fp64 run.  The oracle in
    bf16 issues the same ATen calls (bf16 matmuls with fp32 accumulation, fp32-inside LayerNorm / softmax / GELU), so its
    drift must be the reference's drift -- not bit-identical (einsum contraction order), but the same size."""
    g, g64 = golden("forward_m16_c33_bf16.npz"), golden("forward_m16_c33_fp64.npz")
    ref_emb, ref_atp = rel_l2(g["emb"], g64["emb"]), float(np.abs(g["atp"] - g64["atp"]).max())
    assert 5e-3 < ref_emb < 5e-2 and 1e-2 < ref_atp < 3e-1        # SURVEY §6 measured 2.2e-2 / 9e-2
    params = O.to_torch_params(full_state, torch.bfloat16)
    emb, atp = O.pack_outputs(O.forward(torch.from_numpy(g["tokens"]), params))
    assert emb.dtype == torch.bfloat16
    o_emb = rel_l2(emb.float().numpy(), g64["emb"])
    o_atp = float(np.abs(atp.float().numpy() - g64["atp"]).max())
    assert 0.5 * ref_emb < o_emb < 2.0 * ref_emb, (o_emb, ref_emb)
    assert 0.4 * ref_atp < o_atp < 2.5 * ref_atp, (o_atp, ref_atp)
    assert rel_l2(emb.float().numpy(), g["emb"]) < 3.0 * ref_emb


@pytest.mark.parametrize("name", ["t37_b3_e128", "t70_b2_e768"])
def test_generic_mha_weights_and_key_padding_mask_match_reference(name):
    """msm/multihead_attention.py:154-397 as msm/modules.py:123-131 calls it: need_weights=True (the default) returns the
    head-averaged probabilities [B,T,T]; need_head_weights the per-head ones [H,B,T,T]; key_padding_mask [B,T]."""
    g = golden(f"mha_masks_{name}.npz")
    T, B, E, H = (int(v) for v in g["meta"])
    st = O.to_torch_params(synthetic.make_state_dict(seed=11, embed_dim=E, num_layers=1, num_heads=H))
    x = torch.from_numpy(synthetic.normal(f"mha:{name}", 11, (T, B, E)).astype(np.float32))
    pre = "layers.0.row_self_attention.layer"
    y, w = O.multihead_self_attention(x, st, pre, H, return_weights=True)
    assert rel_l2(y, g["out_default"]) < 1e-5 and np.abs(w.mean(0).numpy() - g["avg_weights_default"]).max() < 2e-6
    kpm = torch.from_numpy(g["key_padding_mask"])
    y, w = O.multihead_self_attention(x, st, pre, H, key_padding_mask=kpm, return_weights=True)
    assert rel_l2(y, g["out_masked"]) < 1e-5
    assert w.shape == g["head_weights_masked"].shape and np.abs(w.numpy() - g["head_weights_masked"]).max() < 2e-6
    assert np.abs(w.mean(0).numpy() - g["avg_weights_masked"]).max() < 2e-6
    assert float(w[:, 0, :, T - 5:].max()) == 0.0 and float(w[:, B - 1, :, 3].max()) == 0.0


def test_generic_mha_with_attn_mask_matches_reference():
    """msm/multihead_attention.py:353-357: the float attn_mask is added to the scores before the key padding fill and the softmax
    (fixture: a causal mask plus finite biases, alone and with a key_padding_mask; NaN rows where no key is admissible)."""
    g = golden("mha_attn_mask.npz")
    T, B, E, H = (int(v) for v in g["meta"])
    st = O.to_torch_params(synthetic.make_state_dict(seed=int(g["seed"]), embed_dim=E, num_layers=1, num_heads=H))
    x = torch.from_numpy(synthetic.normal("mha:attn_mask", 13, (T, B, E)).astype(np.float32))
    pre = "layers.0.row_self_attention.layer"
    am = torch.from_numpy(g["attn_mask"])
    y, w = O.multihead_self_attention(x, st, pre, H, return_weights=True, attn_mask=am)
    assert rel_l2(y, g["out"]) < 1e-5 and np.abs(w.numpy() - g["head_weights"]).max() < 2e-6
    assert np.abs(w.mean(0).numpy() - g["avg_weights"]).max() < 2e-6
    yk, wk = O.multihead_self_attention(x, st, pre, H, key_padding_mask=torch.from_numpy(g["key_padding_mask"]), return_weights=True,
                                        attn_mask=am)
    live = np.isfinite(g["out_kpm"])
    assert np.array_equal(np.isnan(yk.numpy()), ~live)
    assert rel_l2(np.where(live, yk.numpy(), 0.0), np.where(live, g["out_kpm"], 0.0)) < 1e-5
    assert np.nanmax(np.abs(wk.mean(0).numpy() - g["avg_weights_kpm"])) < 2e-6


def test_full_2drb1_alignment_tokens_and_default_subsampling_bit_exact(full_2drb1_a2m):
    """BASELINE configs[0] at full depth: the shipped 1176-row alignment (rebuilt from the reference reader's token matrix)
    through the reader, then `diversity-max` to the CLI default of 512 rows (utils/align.py:128-148) -- tokens bit-exact
    with the reference's."""
    g = golden("tokens_2DRB_1_full.npz")
    toks = TO.encode_msa(open(full_2drb1_a2m).read())
    assert toks.shape == tuple(g["depth_seqlen"]) == (1176, 36)
    assert np.array_equal(toks, g["all_tokens"].astype(np.int64)) and np.array_equal(toks[:512], g["first_512"])
    from rnamsm.msa import greedy_select
    sel = greedy_select(toks, 512, "max")
    assert np.array_equal(toks[sel], g["diversity_max_512"])


def test_a_padded_frame_differs_from_the_alignment_alone_only_through_the_padded_depth(full_state):
    """What rnamsm_forward_batch's `true_rows` rests on, shown on the oracle (the reference's semantics, pinned above on its
    padded fixture): an alignment framed with <pad> COLUMNS comes out as alone -- masked keys get probability exactly 0, padded
    queries are zeroed, padded values reach no real token -- while <pad> ROWS change its outputs, because align_scaling divides
    the tied logits by the square root of the PADDED depth (modules.py:713-715).  With that one factor put right (the HIP
    ragged batch) the framed alignment equals the alignment alone (tests/test_gpu_forward.py)."""
    params = O.to_torch_params(full_state)
    toks = torch.from_numpy(synthetic.make_tokens(3, 9, 11))
    alone = O.forward(toks, params)
    wide = torch.full((3, 13), 1, dtype=torch.int64)
    wide[:, :9] = toks
    cols = O.forward(wide, params)
    assert rel_l2(cols["representation"][:, :9], alone["representation"]) < 1e-5
    assert np.abs(cols["row_attentions"][..., :9, :9].numpy() - alone["row_attentions"].numpy()).max() < 1e-5
    assert float(cols["row_attentions"][..., :9, 9:].abs().max()) == 0.0            # padded keys: probability exactly 0
    deep = torch.full((6, 9), 1, dtype=torch.int64)
    deep[:3] = toks
    rows = O.forward(deep, params)
    assert np.abs(rows["row_attentions"].numpy() - alone["row_attentions"].numpy()).max() > 1e-2   # sqrt(6) instead of sqrt(3)


def test_oracle_lm_head_on_masked_copies_matches_the_reference_likelihood_pattern():
    """tests/golden/likelihood_m8_c17.npz (make_golden_r3.py): the reference's `model(batch)["logits"]` on masked copies of an
    alignment, picked at (copy i, row 0, position i) as utils/likelihood.py:60-82 does -- the oracle's forward + lm_head."""
    g = golden("likelihood_m8_c17.npz")
    state = synthetic.make_state_dict(seed=0)
    params = O.to_torch_params(state)
    batch, idx = torch.from_numpy(g["masked_tokens"]), g["indices"]
    got = np.stack([O.lm_head(O.forward(batch[b], params)["representation"], params)[0, int(idx[b])].numpy() for b in range(len(idx))])
    assert rel_l2(got, g["picked_logits"]) < 1e-5


@pytest.mark.parametrize("name", ["d128_r7_c33", "d768_r6_c19"])
def test_oracle_layer_three_tuple_matches_reference(name):
    """tests/golden/layer_probs_*.npz (make_golden_r3.py): AxialTransformerLayer.forward(need_head_weights=True) of the
    reference -- output, column probabilities, row probabilities -- without and with a padding mask, against the oracle's
    row_attention / col_attention(return_probs=True) / ffn chained as modules.py:242-267 chains them."""
    g = golden(f"layer_probs_{name}.npz")
    D, H, R, C = (int(v) for v in g["meta"])
    params = O.to_torch_params(synthetic.make_state_dict(seed=7, embed_dim=D, num_layers=1, num_heads=H))
    x0 = torch.from_numpy(synthetic.normal(f"x:{name}", 7, (R, C, 1, D)).astype(np.float32))[:, :, 0]
    for pad, sfx in ((None, ""), (torch.from_numpy(g["pad"])[0], "_masked")):
        ln = lambda x, pre: O.layer_norm(x, params[f"{pre}.layer_norm.weight"], params[f"{pre}.layer_norm.bias"])
        pre = "layers.0.row_self_attention"
        y, rp = O.row_attention(ln(x0, pre), params, f"{pre}.layer", H, None, pad)
        x1 = x0 + y
        pre = "layers.0.column_self_attention"
        y, cp = O.col_attention(ln(x1, pre), params, f"{pre}.layer", H, return_probs=True, pad=pad)
        x2 = x1 + y
        pre = "layers.0.feed_forward_layer"
        x3 = x2 + O.ffn(ln(x2, pre), params, f"{pre}.layer")
        assert rel_l2(x3.numpy(), g["out" + sfx]) < 1e-5
        assert np.abs(cp.numpy() - g["col_probs" + sfx]).max() < 2e-5
        assert np.abs(rp.numpy() - g["row_probs" + sfx]).max() < 2e-5


MSM_VARIANT_CASES = ["d128_b2_r5_c9", "d768_b1_r4_c7"]


def msm_variant_state(g):
    """The weights behind a msm_variant_*.npz fixture: synthetic.make_state_dict(seed, D, L, H) with the per-channel
    msa_position_embedding of the msm/ shell variant (rows past the stored ones are never read: zeros)."""
    D, H, L, B, R, C, seed = (int(v) for v in g["meta"])
    state = synthetic.make_state_dict(seed=seed, embed_dim=D, num_layers=L, num_heads=H)
    rows = np.zeros((1, 1024, 1, D), dtype=np.float32)
    rows[:, :R] = g["msa_position_embedding_rows"]
    state["msa_position_embedding"] = rows
    return state, (D, H, L, B, R, C)


@pytest.mark.parametrize("name", MSM_VARIANT_CASES)
def test_oracle_matches_the_msm_variant_of_the_model_shell(name):
    """msm/model.py:206-423 (per-channel msa_position_embedding, col_attentions next to row_attentions), VERDICT r03 item 6."""
    g = golden(f"msm_variant_{name}.npz")
    state, (D, H, L, B, R, C) = msm_variant_state(g)
    params = O.to_torch_params(state)
    for b in range(B):
        out = O.forward(torch.from_numpy(g["tokens"][b]), params, num_layers=L, num_heads=H, return_col_attentions=True)
        assert rel_l2(out["representation"].numpy(), g["repr_last"][b]) < 1e-5
        assert np.abs(out["row_attentions"].numpy() - g["row_attentions"][b]).max() < 2e-5
        assert out["col_attentions"].shape == g["col_attentions"][b].shape == (L, H, C, R, R)
        assert np.abs(out["col_attentions"].numpy() - g["col_attentions"][b]).max() < 2e-5
        assert rel_l2(O.lm_head(out["representation"], params).numpy(), g["logits"][b]) < 1e-5


def test_general_mha_restatement_matches_the_reference_on_every_option():
    """oracle.multihead_attention (cross-attention, kdim / vdim, bias_kv, zero_attn, bias=False, incremental state, static_kv,
    before_softmax, a finite large-negative attn_mask with key padding) against outputs of the reference itself
    (tests/golden/mha_general.npz, tests/golden/make_golden_r6.py; msm/multihead_attention.py:154-434)."""
    import mha_cases as MC
    g = golden("mha_general.npz")
    E, H = MC.E, MC.H
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    W = lambda tag, **kw: {k: tt(v) for k, v in MC.weights(tag, **kw).items()}

    def close(a, b, tol=1e-5):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
        return rel_l2(np.nan_to_num(a), np.nan_to_num(b)) < tol

    # finite large-negative mask + key padding
    T, B = (int(v) for v in g["finite.meta"][:2])
    x = tt(MC.rnd("finite.x", (T, B, E)))
    y, w = O.multihead_attention(x, x, x, W("finite"), H, key_padding_mask=tt(g["finite.kpm"]), attn_mask=tt(g["finite.attn_mask"]))
    assert close(y, g["finite.out"]) and np.abs(w.numpy() - g["finite.head_weights"]).max() < 2e-6
    assert np.abs(w.mean(0).numpy() - g["finite.avg_weights"]).max() < 2e-6
    # cross-attention
    for tag, Tq, S in (("short_q", 9, 23), ("long_q", 27, 13)):
        q, kv = tt(MC.rnd(f"cross.{tag}.q", (Tq, 3, E))), tt(MC.rnd(f"cross.{tag}.kv", (S, 3, E)))
        y, w = O.multihead_attention(q, kv, kv, W("cross"), H)
        assert close(y, g[f"cross.{tag}.out"]) and np.abs(w.mean(0).numpy() - g[f"cross.{tag}.avg_weights"]).max() < 2e-6
        y, w = O.multihead_attention(q, kv, kv, W("cross"), H, key_padding_mask=tt(g[f"cross.{tag}.kpm"]))
        assert close(y, g[f"cross.{tag}.out_kpm"]) and np.abs(w.numpy() - g[f"cross.{tag}.head_weights_kpm"]).max() < 2e-6
    # kdim / vdim
    q, k, v = tt(MC.rnd("kdim.q", (11, 2, E))), tt(MC.rnd("kdim.k", (17, 2, 96))), tt(MC.rnd("kdim.v", (17, 2, 160)))
    y, w = O.multihead_attention(q, k, v, W("kdim", kdim=96, vdim=160), H)
    assert close(y, g["kdim.out"]) and np.abs(w.numpy() - g["kdim.head_weights"]).max() < 2e-6
    # bias_kv + zero_attn
    x = tt(MC.rnd("biaskv.x", (14, 3, E)))
    wb = W("biaskv", bias_kv=True)
    y, w = O.multihead_attention(x, x, x, wb, H, add_zero_attn=True)
    assert close(y, g["biaskv.out"]) and np.abs(w.mean(0).numpy() - g["biaskv.avg_weights"]).max() < 2e-6
    y, w = O.multihead_attention(x, x, x, wb, H, add_zero_attn=True, key_padding_mask=tt(g["biaskv.kpm"]),
                                 attn_mask=tt(MC.rnd("biaskv.am", (14, 14), 0.5)))
    assert close(y, g["biaskv.out_masked"]) and np.abs(w.numpy() - g["biaskv.head_weights_masked"]).max() < 2e-6
    # bias=False
    x = tt(MC.rnd("nobias.x", (10, 2, E)))
    y, w = O.multihead_attention(x, x, x, W("nobias", bias=False), H)
    assert close(y, g["nobias.out"]) and np.abs(w.numpy() - g["nobias.head_weights"]).max() < 2e-6
    # incremental self-attention
    x = tt(MC.rnd("incr.x", (6, 2, E)))
    wi, saved, ys = W("incr"), {}, []
    for s in range(6):
        y, w = O.multihead_attention(x[s:s + 1], x[s:s + 1], x[s:s + 1], wi, H, saved=saved)
        ys.append(y)
    assert close(torch.cat(ys), g["incr.out_steps"]) and np.abs(w.mean(0).numpy() - g["incr.last_weights"]).max() < 2e-6
    assert close(saved["prev_key"], g["incr.final_prev_key"]) and close(saved["prev_value"], g["incr.final_prev_value"])
    saved, ys = {}, []
    for s in range(4):
        kp = torch.tensor([[False], [True]]) if s == 2 else None
        ys.append(O.multihead_attention(x[s:s + 1], x[s:s + 1], x[s:s + 1], wi, H, saved=saved, key_padding_mask=kp)[0])
    assert close(torch.cat(ys), g["incr.out_steps_kpm"]) and np.array_equal(saved["prev_key_padding_mask"].numpy(), g["incr.final_kpm"])
    # incremental encoder-decoder attention, static_kv
    enc, q = tt(MC.rnd("incr_ed.enc", (12, 2, E))), tt(MC.rnd("incr_ed.q", (3, 2, E)))
    we, saved, ekpm = W("incr_ed"), {}, tt(g["incr_ed.kpm"])
    ys = [O.multihead_attention(q[0:1], enc, enc, we, H, key_padding_mask=ekpm, saved=saved, static_kv=True)[0]]
    for s in (1, 2):
        y, w = O.multihead_attention(q[s:s + 1], None, None, we, H, key_padding_mask=ekpm, saved=saved, static_kv=True)
        ys.append(y)
    assert close(torch.cat(ys), g["incr_ed.out_steps"]) and np.abs(w.mean(0).numpy() - g["incr_ed.last_weights"]).max() < 2e-6
    # before_softmax
    x = tt(MC.rnd("pre.x", (9, 2, E)))
    sc, vv = O.multihead_attention(x, x, x, W("pre"), H, key_padding_mask=tt(g["pre.kpm"]), attn_mask=tt(MC.rnd("pre.am", (9, 9), 0.3)),
                                   before_softmax=True)
    fin = np.isfinite(g["pre.scores"])
    assert np.array_equal(np.isneginf(sc.numpy()), ~fin) and np.abs(np.where(fin, sc.numpy() - g["pre.scores"], 0)).max() < 2e-5
    assert close(vv, g["pre.values"])
