"""Drop-in seam 3: opendcvc_amd.mlcodec_shim presents the API of the reference's pybind module
MLCodec_extensions_cpp (src/cpp/py_rans/py_rans.cpp:366-393) over the C ABI of libdcvc_amd.so.
  * known-answer tests against the streams the reference's own module produced (tests/golden/rans_kat.npz) - always run;
  * the boundary proven against the reference's real caller: the REFERENCE's DMCI / DMC (torch CPU fallback ops) with the
    shim registered as MLCodec_extensions_cpp must reproduce the golden sequences byte for byte (build container only:
    skipped where /root/reference is absent, e.g. on the GPU box)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from opendcvc_amd import _lib

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libdcvc_amd.so not built")


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "rans_kat.npz"))


def _packed(k):
    return ((k["sym"].astype(np.int32) << 8) + k["idx"]).astype(np.int16)


@pytest.mark.parametrize("two", [False, True])
def test_encoder_reproduces_reference_streams(kat, two):
    from opendcvc_amd import mlcodec_shim as M
    enc = M.RansEncoder()
    assert enc.add_cdf(kat["cdf"], kat["sizes"], kat["offsets"]) == 0
    enc.set_use_two_encoders(two)
    assert enc.get_use_two_encoders() is two
    p = _packed(kat)
    enc.reset()
    enc.encode_z(kat["z"], 0, 0, 6)
    enc.encode_y(p, 0)
    enc.encode_y(p[:777], 0)
    enc.encode_y(p[:0], 0)                                   # zero-length task
    enc.flush()
    out = enc.get_encoded_stream()
    assert out.dtype == np.uint8 and np.array_equal(out, kat[f"stream_two{int(two)}"])
    small = ((np.clip(kat["sym"], -2, 2).astype(np.int32) << 8) + kat["idx"]).astype(np.int16)
    for m in (64, 201):
        enc.reset()
        enc.encode_y(small[:m], 0)
        enc.flush()
        assert np.array_equal(enc.get_encoded_stream(), kat[f"stream_two{int(two)}_y{m}"])


@pytest.mark.parametrize("two", [False, True])
def test_decoder_reads_reference_streams(kat, two):
    from opendcvc_amd import mlcodec_shim as M
    dec = M.RansDecoder()
    assert dec.add_cdf(kat["cdf"], kat["sizes"], kat["offsets"]) == 0
    dec.set_use_two_decoders(two)
    assert dec.get_use_two_decoders() is two
    dec.set_stream(kat[f"stream_two{int(two)}"])
    dec.decode_z(kat["z"].size, 0, 0, 6)
    z = dec.get_decoded_tensor()
    assert z.dtype == np.int8 and np.array_equal(z, kat["z"])
    dec.decode_y(kat["idx"], 0)
    assert np.array_equal(dec.get_decoded_tensor(), kat["sym"].astype(np.int8))
    assert np.array_equal(dec.decode_and_get_y(kat["idx"][:777], 0), kat["sym"][:777].astype(np.int8))


def test_pmf_and_cdf_buffer(kat):
    from opendcvc_amd import mlcodec_shim as M
    for i in range(len(kat["pmf_len"])):
        n = int(kat["pmf_len"][i])
        got = M.pmf_to_quantized_cdf([float(v) for v in kat["pmf_in"][i][:n]], 16)
        assert isinstance(got, list) and got == [int(v) for v in kat["pmf_out"][i][:n + 1]]
    enc, dec = M.RansEncoder(), M.RansDecoder()
    for c in (enc, dec):
        assert c.add_cdf(kat["cdf"], kat["sizes"], kat["offsets"]) == 0
        assert c.add_cdf(kat["cdf"][:8], kat["sizes"][:8], kat["offsets"][:8]) == 1
        c.empty_cdf_buffer()
        assert c.add_cdf(kat["cdf"][:8], kat["sizes"][:8], kat["offsets"][:8]) == 0
    with pytest.raises(_lib.DcvcError):
        enc.add_cdf(kat["cdf"], kat["sizes"][:5], kat["offsets"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/models"), reason="reference sources not present (build container only)")
@pytest.mark.parametrize("name,args", [("seq_64", (64, 64, 7, 32, False, 4)), ("seq_64_two", (64, 64, 3, 21, True, 0)),
                                       ("seq_80x48", (48, 80, 3, 40, False, 0))])
def test_reference_models_run_on_the_shim(golden_dir, name, args):
    """The reference's own DMCI / DMC and entropy_models.py, entropy coding through mlcodec_shim: streams, sizes and
    reconstructions equal the records the reference produced with ITS coder (tests/golden/make_golden.py)."""
    sys.path.insert(0, golden_dir)
    import torch
    import ref_harness
    DMC, DMCI, *_ = ref_harness.load(coder="shim")
    import MLCodec_extensions_cpp as bound
    assert bound.__name__ == "opendcvc_amd.mlcodec_shim"
    import make_golden
    torch.set_grad_enabled(False)
    i_net, p_net = make_golden.load_models(DMC, DMCI)
    assert type(i_net.entropy_coder.encoder).__module__ == "opendcvc_amd.mlcodec_shim"
    h, w, n, qp, two, reset = args
    rec, _ = make_golden.run_sequence(i_net, p_net, h, w, n, qp, two, reset, False)
    if name == "seq_64":
        want = json.loads(str(np.load(os.path.join(golden_dir, "seq_64.npz"))["meta"]))
    else:
        want = json.load(open(os.path.join(golden_dir, "sequences.json")))[name]
    for fi, (g, f) in enumerate(zip(rec["frames"], want["frames"])):
        assert (g["bytes"], g["sha256"]) == (f["bytes"], f["sha256"]), f"frame {fi}: stream differs from the reference coder's"
        assert g["x_hat_u8_sha256"] == f["x_hat_u8_sha256"] and g["psnr"] == pytest.approx(f["psnr"], abs=1e-9)
