"""Import harness for the upstream reference (authoring container ONLY).

Used solely by tests/golden/make_golden.py to produce the committed fixtures.
/root/reference does not exist on the GPU box; nothing under tests/ imports
this module at test time.

The reference needs four packages that are absent offline (pytorch_lightning,
tape, numba, Bio); they are replaced by the smallest stubs that let
model.py / dataset.py / utils.tokenization import.  hhfilter is never called
(it would write into the read-only tree).
"""
import os
import sys
import types

REF = os.environ.get("RNAMSM_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    import torch.nn as nn

    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if "pytorch_lightning" not in sys.modules:
        _stub("pytorch_lightning", LightningModule=nn.Module,
              seed_everything=lambda *a, **k: None)
    if "tape" not in sys.modules:
        tape = _stub("tape")
        tape.tokenizers = _stub("tape.tokenizers", TAPETokenizer=object)
    if "numba" not in sys.modules:
        _stub("numba", njit=lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f)))
    if "Bio" not in sys.modules:
        def _parse(handle, fmt):
            assert fmt == "fasta"
            close = False
            if not hasattr(handle, "read"):
                handle = open(handle)
                close = True
            try:
                name, chunks = None, []
                for line in handle:
                    line = line.rstrip("\n").rstrip("\r")
                    if line.startswith(">"):
                        if name is not None:
                            yield types.SimpleNamespace(description=name, seq="".join(chunks).replace(" ", "").replace("\r", ""))
                        name, chunks = line[1:].rstrip(), []
                    elif name is not None:
                        chunks.append(line.rstrip())        # Bio.SeqIO.FastaIO.SimpleFastaParser: rstrip, join, drop ' ' and '\r'
                if name is not None:
                    yield types.SimpleNamespace(description=name, seq="".join(chunks).replace(" ", "").replace("\r", ""))
            finally:
                if close:
                    handle.close()
        bio = _stub("Bio")
        bio.SeqIO = _stub("Bio.SeqIO", parse=_parse)
        bio.Seq = _stub("Bio.Seq", Seq=str)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def reference_modules():
    """Returns (model_module, modules_module, msm, Vocab, MSA) from the reference."""
    install()
    import model as ref_model        # /root/reference/model.py (live CLI model)
    import modules as ref_modules    # /root/reference/modules.py
    import msm as ref_msm
    from utils.tokenization import Vocab
    from utils.align import MSA
    return ref_model, ref_modules, ref_msm, Vocab, MSA
