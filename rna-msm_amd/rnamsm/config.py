"""Structured configuration with hydra-style `group.key=value` overrides.

The reference's CLI is a hydra app over dataclasses (RNA_MSM_Inference.py:20-87); hydra / omegaconf are not
available here, so this is a small dotted-override parser over equivalent dataclasses with the same group
names, keys and defaults.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Tuple


@dataclass
class DataConfig:                       # RNA_MSM_Inference.py:20-32
    device: str = "cuda"
    root_path: str = "."
    MSA_path: str = "results"
    MSA_list: str = "rna_id.txt"
    model_path: str = "pretrained/RNA_MSM_pretrained.ckpt"
    num_workers: int = 3
    architecture: str = "rna language"
    max_seqlen: int = 1024
    max_tokens: int = 16384
    max_seqs_per_msa: int = 512
    sample_method: str = "hhfilter"
    # extra (not in the reference): the small alignments (<= 16384 tokens each; 8192 in bf16) of the id list share launch sets
    # (MSATransformer.forward_ragged).  On by default since round 3: a lone forward of a few hundred tokens costs 2.5 ms on a mostly
    # idle chip; data.batch_small_msas=false restores the strictly one-by-one loop of the reference (RNA_MSM_Inference.py:141-148)
    batch_small_msas: bool = True
    # (round 3's framed ragged batches in a 16-bit mode, on request only; the default route below needs no opt-in)
    batch_small_msas_16bit: bool = False
    # round 4: the groups are TOKEN-PACKED -- back to back on the token axis, nothing padded (rnamsm_forward_packed) -- instead of
    # padded into a frame; false = the framed ragged batch of round 3 (A/B, tools/cli_throughput.py).  Round 5: in exact mode an
    # alignment's files are BYTE FOR BYTE those of the one-by-one loop, whatever else is in the list (one arithmetic per alignment:
    # tests/test_gpu_cli.py); in a 16-bit mode (model.gemm_dtype = bf16 | f16x3) the packed batch runs in that mode too (Linear
    # layers on the 16-bit matrix cores, attention on the exact kernels), a lone small alignment as a packed batch of one, so an
    # alignment's files depend on its company at fp32-accumulation rounding at most (which GEMM tile the batch's token count selects)
    pack_small_msas: bool = True


@dataclass
class MSATransformerModelConfig:        # RNA_MSM_Inference.py:35-43
    embed_dim: int = 768
    num_attention_heads: int = 12
    num_layers: int = 10
    embed_positions_msa: bool = True
    dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.1
    # extra (not in the reference): arithmetic of the Linear GEMMs -- f32 (exact, default) | f16x3 | bf16
    gemm_dtype: str = "f32"


@dataclass
class OptimizerConfig:                  # RNA_MSM_Inference.py:46-54 (unused by inference, kept for key parity)
    name: str = "adam"
    learning_rate: float = 3e-4
    weight_decay: float = 3e-4
    lr_scheduler: str = "warmup_cosine"
    warmup_steps: int = 16000
    adam_betas: Tuple[float, float] = (0.9, 0.999)
    max_steps: int = 500000


@dataclass
class Config:
    data: DataConfig = field(default_factory=DataConfig)
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)
    model: MSATransformerModelConfig = field(default_factory=MSATransformerModelConfig)


def _coerce(text: str, current):
    if isinstance(current, bool):
        if text.lower() in ("true", "1", "yes"):
            return True
        if text.lower() in ("false", "0", "no"):
            return False
        raise ValueError(f"expected a boolean, got {text!r}")
    if isinstance(current, int):
        return int(text)
    if isinstance(current, float):
        return float(text)
    if isinstance(current, tuple):
        parts = [p for p in text.strip("()[] ").split(",") if p.strip()]
        return tuple(float(p) for p in parts)
    return text


def parse_overrides(argv: List[str], cfg: Config = None) -> Config:
    """`data.MSA_path=results model.num_layers=10 ...`; unknown groups/keys and malformed tokens raise."""
    cfg = cfg or Config()
    for tok in argv:
        if "=" not in tok:
            raise ValueError(f"override {tok!r} is not of the form group.key=value")
        dotted, value = tok.split("=", 1)
        dotted = dotted.lstrip("+")
        parts = dotted.split(".")
        if len(parts) != 2:
            raise ValueError(f"override key {dotted!r} must be group.key")
        group, key = parts
        if not hasattr(cfg, group):
            raise KeyError(f"unknown config group {group!r} (have: data, model, optimizer)")
        node = getattr(cfg, group)
        names = {f.name for f in dataclasses.fields(node)}
        if key not in names:
            raise KeyError(f"unknown key {key!r} in group {group!r}")
        setattr(node, key, _coerce(value, getattr(node, key)))
    return cfg
