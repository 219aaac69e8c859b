"""Harness that imports the REFERENCE Python models (read-only, /root/reference) on a GPU-less
host so golden vectors can be generated.  Only used by tests/golden/make_golden.py in the build
container; nothing here (and nothing it imports) travels to or runs on the GPU box.

The reference compress()/decompress() call torch.cuda.Event/Stream/stream/synchronize
unconditionally (video_model.py:315-337, image_model.py:163-180), which raise without a GPU;
SURVEY.md section 8c prescribes harness-side no-op stand-ins set on the torch.cuda module
(the reference files themselves are not edited).
"""
import contextlib
import os
import sys

REF_ROOT = os.environ.get("DCVC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_RANS_DIR = os.path.join(REPO, "oracle", "_ref")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


def load(coder="reference"):
    """Returns (DMC, DMCI, ref_layers_module, ref_cuda_inference_module, MLCodec_extensions_cpp, stream_helper).

    coder="reference": the reference's own rANS module compiled into oracle/_ref.
    coder="shim": opendcvc_amd.mlcodec_shim registered as MLCodec_extensions_cpp - the REFERENCE's models then
    entropy-code through the host coder of libdcvc_amd.so (drop-in seam 3, tests/test_mlcodec_shim.py)."""
    import torch

    os.environ.setdefault("SUPPRESS_CUSTOM_KERNEL_WARNING", "1")
    sys.dont_write_bytecode = True
    if coder == "shim":
        if REPO not in sys.path:
            sys.path.insert(0, REPO)
        from opendcvc_amd import mlcodec_shim
        mlcodec_shim.install()
    else:
        sys.modules.pop("MLCodec_extensions_cpp", None) if getattr(
            sys.modules.get("MLCodec_extensions_cpp"), "__name__", "") == "opendcvc_amd.mlcodec_shim" else None
    for p in ((REF_ROOT,) if coder == "shim" else (REF_RANS_DIR, REF_ROOT)):
        if p not in sys.path:
            sys.path.insert(0, p)

    class _NoEvent:
        def record(self, *a, **k):
            pass

        def wait(self, *a, **k):
            pass

        def synchronize(self):
            pass

    class _NoStream:
        def __init__(self, *a, **k):
            pass

    torch.cuda.Event = _NoEvent
    torch.cuda.Stream = _NoStream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None

    import MLCodec_extensions_cpp as ref_rans
    from src.models.video_model import DMC
    from src.models.image_model import DMCI
    import src.layers.layers as ref_layers
    import src.layers.cuda_inference as ref_ops
    import src.utils.stream_helper as ref_stream
    return DMC, DMCI, ref_layers, ref_ops, ref_rans, ref_stream
