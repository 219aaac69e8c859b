import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """fp16 error statistics of the layer tests (max / mean |err| over the tensor RMS, per case), kept next to
    the other GPU-run artefacts so that the bounds in test_gpu_layers.py can be checked against measurements."""
    mod = sys.modules.get("test_gpu_layers")
    stats = getattr(mod, "F16_STATS", None)
    if stats:
        import json
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump({k: dict(max_over_rms=v[0], mean_over_rms=v[1]) for k, v in stats.items()},
                  open(os.path.join(out, "f16_err.json"), "w"), indent=1)
