#!/usr/bin/env python3
"""Drop-in for the reference's RNA_MSM_Inference.py on MI355X.

    python RNA_MSM_Inference.py data.root_path=$PWD data.MSA_path=results \
        data.model_path=pretrained/RNA_MSM_pretrained.ckpt data.MSA_list=rna_id.txt

Same `group.key=value` overrides and defaults as the reference (RNA_MSM_Inference.py:20-87), same outputs
(`<id>_emb.npy`, `<id>_atp.npy` next to the alignments).  Multi-GPU: launch one process per GPU with
`python -m torch.distributed.run --nproc-per-node N RNA_MSM_Inference.py ...`; ids are sharded over ranks.
Extra, non-reference override: `data.sample_method=first|diversity-max|diversity-min` (the default `hhfilter`
needs the external binary and is only accepted when the alignment already has <= max_seqs_per_msa rows).
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))


def main(argv=None):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from rnamsm.config import Config, DataConfig, parse_overrides
    from rnamsm.inference import extract_feat

    # reference defaults are relative to the script's directory (RNA_MSM_Inference.py:16,23,26)
    defaults = Config(data=DataConfig(root_path=ROOT, model_path=os.path.join(ROOT, "pretrained", "RNA_MSM_pretrained.ckpt")))
    cfg = parse_overrides(list(sys.argv[1:] if argv is None else argv), defaults)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    gather = os.environ.get("RNAMSM_GATHER_TO_RANK0", "0") == "1"
    if world > 1:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # default group gloo (rank / world bookkeeping, agreement); RCCL carries only the gathered output arrays, on a group of
        # its own that extract_feat creates when RNAMSM_GATHER_TO_RANK0=1 -- with a host-staged gloo gather as the second transport
        dist.init_process_group("gloo")
    try:
        extract_feat(cfg, gather_to_rank0=gather)
    finally:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
