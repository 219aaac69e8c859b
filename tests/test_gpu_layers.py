"""HIP conv kernels (through the C ABI) against the CPU oracle on the same seeded inputs.
fp32 mode: bit-exact (the kernels use the oracle's fma order on the fp32-input MFMA);
fp16 mode: compared with the oracle's fp16-storage emulation (Net.dcb_f16 / Net.conv_f16: same packed
weights, rounds to fp16 exactly where the kernels store, fp32 accumulate).  What is left between the two
is the summation order inside the MFMA and the hardware exp2 / rcp, i.e. occasional one-ulp fp16 rounding
flips that travel through the following layers: the bound below is a few fp16 ulps of the tensor's RMS."""
import numpy as np
import pytest
import torch

import dcvc_oracle as O

pytestmark = pytest.mark.gpu

# fp16 mode vs the fp16-storage oracle: max |err| <= F16_MAX_RMS * rms and mean |err| <= F16_MEAN_RMS * rms
# (one fp16 ulp is 2^-11 = 4.9e-4 relative; measured on MI355X: see tools/f16_err.py / profiles/r02_f16_err.json)
F16_MAX_RMS = 6e-3
F16_MEAN_RMS = 2e-4
F16_STATS = {}   # what -> (max/rms, mean/rms), collected for tools/f16_err.py


def _rng(seed):
    return np.random.default_rng(seed)


def make_dcb_weights(rng, prefix, cin, c, adaptor):
    sd = {}

    def conv(name, co, ci, k=1, gain=1.0):
        sd[f"{prefix}.{name}.weight"] = (rng.standard_normal((co, ci, k, k)) * gain / np.sqrt(ci * k * k)).astype(np.float32)
        sd[f"{prefix}.{name}.bias"] = (rng.standard_normal(co) * 0.1).astype(np.float32)

    if adaptor:
        conv("adaptor", c, cin)
    conv("dc.0", c, c)
    sd[f"{prefix}.dc.2.weight"] = (rng.standard_normal((c, 1, 3, 3)) / 3).astype(np.float32)
    sd[f"{prefix}.dc.2.bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    conv("dc.3", c, c, gain=0.5)
    conv("ffn.0", 4 * c, c)
    conv("ffn.2", c, 2 * c, gain=0.5)
    return sd


def to_dev(x_hwc, cp, dtype):
    H, W, C = x_hwc.shape
    t = torch.zeros((H, W, cp), dtype=dtype, device="cuda")
    t[:, :, :C] = torch.from_numpy(x_hwc).to(dtype)
    return t


def compare(got, ref, dtype, what):
    if dtype == torch.float32:
        assert np.array_equal(got, ref), f"{what}: fp32 path not bit-exact, max|d|={np.abs(got - ref).max()}"
    else:
        rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-6
        d = np.abs(got.astype(np.float64) - ref)
        F16_STATS[what] = (float(d.max()) / rms, float(d.mean()) / rms)
        assert d.max() <= F16_MAX_RMS * rms, f"{what}: fp16 max err {d.max()} vs rms {rms} ({d.max() / rms:.2e})"
        assert d.mean() <= F16_MEAN_RMS * rms, f"{what}: fp16 mean err {d.mean()} vs rms {rms} ({d.mean() / rms:.2e})"


DCB_CASES = [
    # cin, c, adaptor, shortcut, quant, H, W, split (concat of two sources)
    (256, 256, False, False, False, 16, 24, None),
    (256, 256, False, False, True, 13, 21, None),
    (512, 256, True, False, False, 9, 17, 256),
    (192, 256, True, False, False, 8, 8, None),
    (128, 128, False, True, False, 5, 7, None),
    (368, 368, False, False, True, 10, 11, None),
    (192, 368, True, False, True, 7, 9, None),
    (368, 192, True, False, False, 6, 6, None),
    (256, 320, True, False, False, 8, 10, None),
    (320, 320, False, False, True, 9, 9, None),
    (512, 384, True, False, False, 7, 8, 128),
    (384, 384, False, False, False, 8, 9, None),
    (512, 512, True, False, False, 6, 10, 256),
    (512, 512, False, False, False, 5, 9, None),
    (256, 128, True, False, False, 4, 4, None),
    # degenerate maps through the 32-pixel ring tails (one ragged tile; a strip of tiles one row high)
    (256, 256, False, True, True, 1, 1, None),
    (384, 384, False, False, False, 3, 50, None),
    (448, 256, True, False, False, 2, 9, 192),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", DCB_CASES)
def test_depth_conv_block(case, dtype):
    from opendcvc_amd import nn
    cin, c, adaptor, shortcut, quant, H, W, split = case
    rng = _rng(100 + DCB_CASES.index(case))
    sd = make_dcb_weights(rng, "m", cin, c, adaptor)
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    q = rng.uniform(0.5, 1.5, c).astype(np.float32) if quant else None
    if dtype == torch.float16:   # the oracle sees the same fp16-rounded input
        x = x.astype(np.float16).astype(np.float32)
    net = O.Net(sd)
    ref = (net.dcb_f16 if dtype == torch.float16 else net.dcb)(x, "m", shortcut=shortcut, q=q)
    blk = nn.DepthConvBlock(sd, "m", dtype, shortcut=shortcut)
    qd = torch.from_numpy(q).cuda() if quant else None
    if split:
        x0 = to_dev(x[:, :, :split], split, dtype)
        x1 = to_dev(x[:, :, split:], cin - split, dtype)
        out = blk(x0, x1, quant=qd)
    else:
        out = blk(to_dev(x, blk.cin_p, dtype), quant=qd)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert not np.any(got[:, :, c:]), "pad channels must stay zero"
    compare(got[:, :, :c], ref, dtype, f"dcb {case}")


@pytest.mark.parametrize("c", [128, 256, 320, 368, 512])
def test_depth_conv_block_large_map_fp16(c):
    """Maps of 12 000+ pixels take the large-map kernels (widths 256 / 384: 128-pixel tiles, one workgroup per CU,
    dcb_tail128_kernel; 64-pixel tiles otherwise: two workgroups per CU at 128, eight-wave workgroups above, the ragged
    one at 320): same check as above, edge tiles included."""
    from opendcvc_amd import nn
    H, W = 101, 123
    rng = _rng(300 + c)
    sd = make_dcb_weights(rng, "m", c, c, False)
    x = rng.standard_normal((H, W, c)).astype(np.float16).astype(np.float32)
    q = rng.uniform(0.5, 1.5, c).astype(np.float32)
    ref = O.Net(sd).dcb_f16(x, "m", shortcut=True, q=q)
    blk = nn.DepthConvBlock(sd, "m", torch.float16, shortcut=True)
    out = blk(to_dev(x, blk.cin_p, torch.float16), quant=torch.from_numpy(q).cuda())
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert not np.any(got[:, :, c:]), "pad channels must stay zero"
    compare(got[:, :, :c], ref, torch.float16, f"dcb large map c={c}")


ADAPTOR_LARGE = [(192, 256, 256), (256, 256, 256), (64, 0, 256), (256, 0, 320), (192, 0, 368), (320, 256, 384), (128, 192, 320)]


def _adaptor_block_large(cin0, cin1, c, H=101, W=123):
    from opendcvc_amd import nn
    rng = _rng(500 + cin0 + 3 * cin1 + 7 * c)
    cin = cin0 + cin1
    sd = make_dcb_weights(rng, "m", cin, c, True)
    x = rng.standard_normal((H, W, cin)).astype(np.float16).astype(np.float32)
    blk = nn.DepthConvBlock(sd, "m", torch.float16)
    x0 = to_dev(x[:, :, :cin0], cin0, torch.float16)
    out = blk(x0, to_dev(x[:, :, cin0:], cin1, torch.float16)) if cin1 else blk(x0)
    torch.cuda.synchronize()
    return sd, x, out.float().cpu().numpy()


@pytest.mark.parametrize("cin0,cin1,c", ADAPTOR_LARGE)
def test_adaptor_block_large_map_fp16(cin0, cin1, c):
    """Adaptor blocks (one or two sources) on a large map: the 128-pixel head (dcb_head128_kernel: sources through LDS in
    128-channel chunks, 64-channel remainders, ragged width 320, edge tiles) in front of the 128-pixel tail."""
    sd, x, got = _adaptor_block_large(cin0, cin1, c)
    ref = O.Net(sd).dcb_f16(x, "m", shortcut=False, q=None)
    assert not np.any(got[:, :, c:]), "pad channels must stay zero"
    compare(got[:, :, :c], ref, torch.float16, f"adaptor block large map {cin0}+{cin1}->{c}")


def test_head128_equals_head64_bitwise(tmp_path):
    """The 128-pixel head accumulates in the same k order as dcb_head_kernel: identical block outputs with DCVC_H128=0
    (a separate process: the switch is read once)."""
    import os, subprocess, sys
    outs = {}
    for v in ("1", "0"):
        path = tmp_path / f"o{v}.npy"
        here = os.path.dirname(os.path.abspath(__file__))
        code = ("import sys, numpy as np; sys.path[:0] = [%r, %r, %r]; import test_gpu_layers as t; "
                "np.save(%r, np.concatenate([t._adaptor_block_large(192, 256, 256)[2].ravel(), "
                "t._adaptor_block_large(256, 0, 320)[2].ravel()]))"
                % (here, os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), str(path)))
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, DCVC_H128=v))
        outs[v] = np.load(path)
    assert np.array_equal(outs["1"], outs["0"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("c,hw", [(256, (21, 19)), (128, (9, 33)), (368, (12, 11)), (320, (101, 123)), (256, (101, 123)), (368, (99, 125))])
def test_chained_blocks_equal_separate_calls(c, hw, dtype):
    """dcb_chain (the next block's first conv computed in the previous block's epilogue) is bit-identical to
    calling the blocks one by one, in both modes; a block with a quant step or a different width ends a chain.
    (fp16: holds because every fp32 -> fp16 store goes through Traits<half_t>::from_f, which keeps the compiler from
    rounding some products once (v_fma_mixlo_f16) and others twice depending on the kernel: gemm_core.hpp.)"""
    from opendcvc_amd import nn
    from opendcvc_amd._lib import DcvcError
    if dtype == torch.float32 and hw[0] > 50:
        pytest.skip("large map only needed for the fp16 64-pixel kernels")
    H, W = hw
    rng = _rng(400 + c)
    blocks = [nn.DepthConvBlock(make_dcb_weights(rng, "m", 2 * c if i == 0 else c, c, i == 0), "m", dtype) for i in range(4)]
    x0 = to_dev(rng.standard_normal((H, W, 2 * c)).astype(np.float32), blocks[0].cin_p, dtype)
    q = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).cuda()
    want = blocks[0](x0)
    for i, b in enumerate(blocks[1:]):
        want = b(want, quant=q if i == 2 else None)
    got = nn.dcb_chain(blocks, x0, quant=q)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # a quant step in the middle: the follower must not be fused (the library refuses it if asked directly)
    with pytest.raises(DcvcError):
        blocks[1](want, quant=q, next_block=blocks[2])
    other = nn.DepthConvBlock(make_dcb_weights(rng, "m", 64, 64, False), "m", dtype)
    assert not other.can_follow(blocks[1], None) and blocks[2].can_follow(blocks[1], None)
    assert not blocks[2].can_follow(blocks[1], q)


def _small_map_outputs():
    """fp16 blocks on maps below 12 000 pixels, widths 256 / 368 (-> 384) / 512 / 128: a 4-block chain behind an adaptor block
    (fused heads, quant at the end), a shortcut + quant block, a 2-block chain ending in a fused 1x1 conv with quant.  (Width 128:
    the blocks without adaptor and without a head computed by their predecessor take dcb_tail128_kernel<128, G32, HEADIN> - the
    head inside - unless DCVC_T32=0 / DCVC_T32_128=0.)"""
    from opendcvc_amd import _lib, nn
    outs = []
    for c, (H, W) in ((256, (21, 19)), (368, (12, 27)), (512, (14, 20)), (256, (68, 120)), (128, (17, 30)), (128, (35, 61))):
        rng = _rng(4000 + c + H)
        blocks = [nn.DepthConvBlock(make_dcb_weights(rng, "m", 2 * c if i == 0 else c, c, i == 0), "m", torch.float16) for i in range(4)]
        x0 = to_dev(rng.standard_normal((H, W, 2 * c)).astype(np.float32), blocks[0].cin_p, torch.float16)
        q = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).cuda()
        outs.append(nn.dcb_chain(blocks, x0, quant=q))
        sc = nn.DepthConvBlock(make_dcb_weights(rng, "m", c, c, False), "m", torch.float16, shortcut=True)
        x1 = to_dev(rng.standard_normal((H, W, c)).astype(np.float32), sc.cin_p, torch.float16)
        outs.append(sc(x1, quant=q))
        csd = {"o.weight": (rng.standard_normal((c, c, 1, 1)) / np.sqrt(c)).astype(np.float32),
               "o.bias": (rng.standard_normal(c) * 0.1).astype(np.float32)}
        conv = nn.Conv2d(csd, "o", torch.float16, epilogue=_lib.EPI_BIAS_QUANT)
        outs.append(nn.dcb_chain(blocks[1:3], x1, then_conv=conv, conv_quant=q))
    torch.cuda.synchronize()
    return np.concatenate([o.float().cpu().numpy().ravel() for o in outs])


def test_tail32_equals_tail_kernel_bitwise(tmp_path):
    """The 32-pixel form of dcb_tail128_kernel (small maps, widths 256 / 384 / 512, and 128 with its head computed inside) stores
    and rounds where dcb_tail_kernel does and accumulates in the same k order: identical outputs with DCVC_T32=0 (a separate
    process: the switch is read once)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for v in ("1", "0"):
        path = tmp_path / f"o{v}.npy"
        code = ("import sys, numpy as np; sys.path[:0] = [%r, %r, %r]; import test_gpu_layers as t; np.save(%r, t._small_map_outputs())"
                % (here, os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), str(path)))
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, DCVC_T32=v))
        outs[v] = np.load(path)
    assert np.isfinite(outs["1"]).all() and np.array_equal(outs["1"], outs["0"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("quant", [False, True])
@pytest.mark.parametrize("c,hw", [(256, (21, 19)), (128, (9, 33)), (320, (40, 37)), (320, (101, 123)), (256, (101, 123)), (384, (99, 125))])
def test_chain_then_conv_equals_separate_calls(c, hw, quant, dtype):
    """dcb_chain(..., then_conv=conv) (the 1x1 conv after a run of blocks computed in the last block's tail, the run's
    own result never written: dcvc_dcb_forward_then_conv) = the run followed by the conv kernel, bit for bit, with and
    without the conv's quant vector; a conv that does not qualify (3x3, other width) takes the separate launch."""
    from opendcvc_amd import _lib, nn
    if dtype == torch.float32 and hw[0] > 50:
        pytest.skip("large map only needed for the fp16 64-pixel kernels")
    H, W = hw
    rng = _rng(900 + c + H)
    blocks = [nn.DepthConvBlock(make_dcb_weights(rng, "m", c, c, False), "m", dtype) for _ in range(2)]
    csd = {"o.weight": (rng.standard_normal((c, c, 1, 1)) / np.sqrt(c)).astype(np.float32),
           "o.bias": (rng.standard_normal(c) * 0.1).astype(np.float32)}
    conv = nn.Conv2d(csd, "o", dtype, epilogue=_lib.EPI_BIAS_QUANT if quant else _lib.EPI_BIAS)
    assert conv.fusable_after(blocks[-1])
    x0 = to_dev(rng.standard_normal((H, W, c)).astype(np.float32), blocks[0].cin_p, dtype)
    q = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).cuda() if quant else None
    want = conv(blocks[1](blocks[0](x0)), quant=q)
    got = nn.dcb_chain(blocks, x0, then_conv=conv, conv_quant=q)
    buf = torch.full((H, W, 2 * conv.cout_p), 3.0, dtype=dtype, device="cuda")
    nn.dcb_chain(blocks, x0, then_conv=conv, conv_quant=q, out=buf[:, :, conv.cout_p:])
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(buf[:, :, conv.cout_p:], want) and bool((buf[:, :, :conv.cout_p] == 3.0).all())
    c3 = {"o.weight": (rng.standard_normal((c, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32), "o.bias": csd["o.bias"]}
    conv3 = nn.Conv2d(c3, "o", dtype, pad=1)
    assert not conv3.fusable_after(blocks[-1])
    assert torch.equal(nn.dcb_chain(blocks, x0, then_conv=conv3), conv3(blocks[1](blocks[0](x0))))


def test_dcb_writes_into_concat_slice():
    from opendcvc_amd import nn
    rng = _rng(5)
    sd = make_dcb_weights(rng, "m", 256, 256, False)
    x = rng.standard_normal((6, 7, 256)).astype(np.float32)
    ref = O.Net(sd).dcb(x, "m")
    blk = nn.DepthConvBlock(sd, "m", torch.float32)
    buf = torch.full((6, 7, 512), 7.0, device="cuda")
    blk(to_dev(x, 256, torch.float32), out=buf[:, :, 256:])
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.array_equal(got[:, :, 256:], ref) and np.all(got[:, :, :256] == 7.0)


CONV_CASES = [
    # cin, cout, k, stride, pad, epilogue, H, W
    (256, 256, 1, 1, 0, "quant", 9, 13),
    (256, 128, 3, 2, 1, "bias", 12, 18),
    (368, 256, 3, 2, 1, "bias", 8, 10),
    (128, 1024, 3, 1, 1, "shuffle", 6, 9),
    (128, 512, 1, 1, 0, "shuffle", 5, 6),
    (256, 1472, 1, 1, 0, "shuffle", 4, 5),
    (128, 128, 2, 2, 0, "bias", 8, 12),
    (256, 256, 2, 2, 0, "bias", 10, 6),
    (512, 514, 1, 1, 0, "bias", 5, 7),
    (514, 256, 1, 1, 0, "bias", 5, 7),
    (320, 192, 1, 1, 0, "bias", 7, 7),
    (384, 256, 1, 1, 0, "wsilu", 6, 5),
    # 3x3 s1 p1 with 256-channel output slices: conv3x3_t128_kernel in fp16 (several tiles, ragged edges, K = 64 ... 256)
    (128, 1024, 3, 1, 1, "shuffle", 19, 35),
    (192, 512, 3, 1, 1, "bias", 11, 17),
    (64, 256, 3, 1, 1, "bias", 9, 20),
    (256, 256, 3, 1, 1, "bias", 17, 16),
]
EPI = {"bias": 0, "quant": 1, "shuffle": 2, "wsilu": 3}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv(case, dtype):
    from opendcvc_amd import nn
    cin, cout, k, stride, pad, epi, H, W = case
    rng = _rng(200 + CONV_CASES.index(case))
    sd = {"m.weight": (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32),
          "m.bias": (rng.standard_normal(cout) * 0.1).astype(np.float32)}
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    if dtype == torch.float16:
        x = x.astype(np.float16).astype(np.float32)
    q = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    if dtype == torch.float16:
        ref = O.Net(sd).conv_f16(x, "m", stride, pad, epilogue=epi if epi in ("quant", "wsilu") else "bias", q=q)
        if epi == "shuffle":
            ref = O.pixel_shuffle(ref, 2)
    else:
        ref = O.Net(sd).conv(x, "m", stride, pad)
        if epi == "quant":
            ref = ref * q
        elif epi == "shuffle":
            ref = O.pixel_shuffle(ref, 2)
        elif epi == "wsilu":
            ref = O.wsilu(ref)
    conv = nn.Conv2d(sd, "m", dtype, stride, pad, EPI[epi])
    out = conv(to_dev(x, conv.cin_p, dtype), quant=torch.from_numpy(q).cuda() if epi == "quant" else None)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    c_log = cout // 4 if epi == "shuffle" else cout
    assert got.shape[:2] == ref.shape[:2]
    assert not np.any(got[:, :, c_log:]), "pad channels must stay zero"
    compare(got[:, :, :c_log], ref, dtype, f"conv {case}")


def _conv3x3_out(cin, cout, epi, H, W, k=3, stride=1, pad=1, scale=False):
    from opendcvc_amd import nn
    rng = _rng(700 + cin + cout + k + stride)
    sd = {"m.weight": (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32),
          "m.bias": (rng.standard_normal(cout) * 0.1).astype(np.float32)}
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    conv = nn.Conv2d(sd, "m", torch.float16, stride, pad, EPI[epi])
    q = torch.from_numpy(rng.uniform(0.5, 1.5, cin).astype(np.float32)).cuda() if scale else None
    out = conv(to_dev(x, conv.cin_p, torch.float16), in_scale=q)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def test_conv3x3_t128_equals_conv_kernel_bitwise(tmp_path):
    """conv3x3_t128_kernel and conv_s2_t32_kernel (stride-2 convs, with and without the input scale, ragged edges) accumulate
    (tap, k) in conv_kernel's order: identical outputs with DCVC_C128=0 (a separate process: the switch is read once)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for v in ("1", "0"):
        path = tmp_path / f"o{v}.npy"
        code = ("import sys, numpy as np; sys.path[:0] = [%r, %r, %r]; import test_gpu_layers as t; "
                "np.save(%r, np.concatenate([t._conv3x3_out(128, 1024, 'shuffle', 68, 120).ravel(), "
                "t._conv3x3_out(192, 256, 'bias', 21, 37).ravel(), "
                "t._conv3x3_out(256, 256, 'bias', 136, 240, 2, 2, 0, True).ravel(), t._conv3x3_out(256, 128, 'bias', 37, 51, 3, 2, 1).ravel(), "
                "t._conv3x3_out(128, 128, 'bias', 13, 18, 2, 2, 0).ravel(), t._conv3x3_out(368, 256, 'bias', 9, 7, 3, 2, 1, True).ravel()]))"
                % (here, os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), str(path)))
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, DCVC_C128=v))
        outs[v] = np.load(path)
    assert np.array_equal(outs["1"], outs["0"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("geom", [(256, 256, 2, 2, 0, 14, 22), (256, 128, 3, 1, 1, 9, 11), (320, 192, 1, 1, 0, 7, 9), (256, 128, 3, 2, 1, 12, 18)])
def test_conv_with_input_scale_equals_scale_then_conv(geom, dtype):
    """Conv2d(x, in_scale=q) (the per-channel factor applied while the input tile is staged: temporal_prior_encoder reading
    ctx_t = x1 * q_feature) = dcvc_scale_channels followed by the plain conv, bit for bit - all staging forms (one tap at
    a time, halo tile, 1x1)."""
    import ctypes
    from opendcvc_amd import _lib, nn
    cin, cout, k, stride, pad, H, W = geom
    rng = _rng(700 + cin + k + stride)
    sd = {"m.weight": (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32),
          "m.bias": (rng.standard_normal(cout) * 0.1).astype(np.float32)}
    conv = nn.Conv2d(sd, "m", dtype, stride, pad)
    x = to_dev(rng.standard_normal((H, W, cin)).astype(np.float32), conv.cin_p, dtype)
    q = torch.from_numpy(rng.uniform(0.5, 1.5, cin).astype(np.float32)).cuda()
    xs = torch.empty_like(x)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().dcvc_scale_channels(nn.dtype_code(dtype), nn._p(x), conv.cin_p, nn._p(q), H * W, conv.cin_p,
                                              nn._p(xs), conv.cin_p, st), "scale_channels")
    want = conv(xs)
    got = conv(x, in_scale=q)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert not torch.equal(got, conv(x))          # (the factor is really applied)


def test_conv_concat_sources():
    from opendcvc_amd import nn
    rng = _rng(9)
    sd = {"m.weight": (rng.standard_normal((128, 512, 1, 1)) / 22).astype(np.float32),
          "m.bias": rng.standard_normal(128).astype(np.float32)}
    x = rng.standard_normal((5, 6, 512)).astype(np.float32)
    ref = O.Net(sd).conv(x, "m")
    conv = nn.Conv2d(sd, "m", torch.float32)
    out = conv(to_dev(x[:, :, :128], 128, torch.float32), to_dev(x[:, :, 128:], 384, torch.float32))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_bad_arguments_fail_loudly():
    from opendcvc_amd import nn
    from opendcvc_amd._lib import DcvcError
    rng = _rng(1)
    sd = make_dcb_weights(rng, "m", 256, 256, False)
    blk = nn.DepthConvBlock(sd, "m", torch.float32)
    with pytest.raises(DcvcError):
        blk(torch.zeros((4, 4, 128), device="cuda"))
