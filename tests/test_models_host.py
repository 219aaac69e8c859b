"""Host-side behaviour of the model classes that needs no GPU: the fp32 master copy of bit_estimator_z (the z CDF
tables define the stream), the per-instance one-thread scope, the harness's GPU-id / CPU-slice rules."""
import threading
import warnings

import numpy as np
import pytest
import torch

from opendcvc_amd import harness, weights
from opendcvc_amd._lib import DcvcError
from opendcvc_amd.models import DMC


def _sd(seed=5):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.make_state_dict("dmc", seed).items()}


def _z_tables(m):
    """the factorized CDF tables update() hands to the coder (group 1)"""
    from opendcvc_amd import entropy
    pre = "bit_estimator_z."
    params = {k[len(pre):]: v.detach().float().cpu() for k, v in m.state_dict().items() if k.startswith(pre)}
    params.update(m._z_master or {})
    return entropy.factorized_cdf_tables(params, m.qp_total, m.z_channel)


def test_z_master_survives_half_and_own_half_state_dict():
    m = DMC()
    m.load_state_dict(_sd())
    m.update(0.12)
    want = _z_tables(m)
    m.half()
    m.update(0.12)                                   # after .half(): same tables (fp32 master)
    assert all(np.array_equal(a, b) for a, b in zip(_z_tables(m), want))
    m.load_state_dict(m.state_dict())                # its own fp16 state dict coming back: master kept
    assert m._z_master is not None
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m.update(0.2)                                # re-thresholding such a model still works, silently
    assert all(np.array_equal(a, b) for a, b in zip(_z_tables(m), want))


def test_foreign_half_checkpoint_warns_instead_of_raising(monkeypatch):
    other = DMC()
    other.load_state_dict(_sd(6))
    other.half()
    m = DMC()
    m.load_state_dict(_sd(5))
    m.half()
    m.load_state_dict(other.state_dict())            # another model's fp16 values: the old master does not apply
    assert m._z_master is None
    with pytest.warns(RuntimeWarning, match="fp16 copies of bit_estimator_z"):
        m.update(0.12)
    monkeypatch.setenv("DCVC_STRICT_Z_TABLES", "1")
    with pytest.raises(DcvcError):
        m.update(0.12)


def test_failed_load_keeps_the_master():
    m = DMC()
    m.load_state_dict(_sd())
    master = {k: v.clone() for k, v in m._z_master.items()}
    bad = _sd(9)
    bad.pop("q_encoder")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)                       # strict load fails ...
    assert set(m._z_master) == set(master) and all(torch.equal(m._z_master[k], master[k]) for k in master)


def test_model_scope_is_per_instance_and_reentrant():
    a, b = DMC(), DMC()
    out = []

    def other(m):
        try:
            with m._frame():
                out.append("ok")
        except DcvcError:
            out.append("refused")

    with a._frame():
        with a._frame():
            for m in (a, b):
                t = threading.Thread(target=other, args=(m,))
                t.start()
                t.join()
    assert out == ["refused", "ok"]


def test_worker_gpu_ids_follow_the_parents_visible_devices():
    assert harness.visible_gpu_ids({"HIP_VISIBLE_DEVICES": "4,5"}) == ["4", "5"]
    assert harness.visible_gpu_ids({"ROCR_VISIBLE_DEVICES": "2", "CUDA_VISIBLE_DEVICES": "7"}) == ["2"]
    assert harness.visible_gpu_ids({}) is None


def test_worker_cpu_slices():
    # 16 workers on a 16-CPU grant inside a 256-thread mask: three CPUs each (codec thread + two rANS workers), disjoint
    parts = [harness.worker_cpus(i, 16, list(range(256)), 16) for i in range(16)]
    assert all(len(p) == harness.MIN_WORKER_CPUS for p in parts) and len({c for p in parts for c in p}) == 48
    # fewer CPUs than that: neighbours share, nobody is left with one CPU
    parts = [harness.worker_cpus(i, 16, list(range(16)), 16) for i in range(16)]
    assert all(len(p) == harness.MIN_WORKER_CPUS for p in parts)
    assert [harness.worker_cpus(i, 2, list(range(8)), None) for i in range(2)] == [[0, 1, 2, 3], [4, 5, 6, 7]]
