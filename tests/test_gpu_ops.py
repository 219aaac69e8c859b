"""Operator-module seam (opendcvc_amd.ops == the reference's inference_extensions_cuda API) on the GPU
against the reference's golden vectors (tests/golden/ops_small.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

import dcvc_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-5, atol=2e-5)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_small.npz"))


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_process_with_mask_bit_exact(g):
    from opendcvc_amd import ops
    outs = ops.process_with_mask_cuda(cu(g["pwm.y"]), cu(g["pwm.scales"]), cu(g["pwm.means"]), cu(g["pwm.mask"]), 0.12)
    for got, name in zip(outs, ("y_res", "y_q", "y_hat", "s_hat")):
        assert np.array_equal(got.cpu().numpy(), g["pwm." + name]), name


def test_build_index_enc_dec(g):
    from opendcvc_amd import ops
    sc, sym = cu(g["idx.scales"]), cu(g["idx.symbols"])
    args = (0.11, 16.0, O.LOG_SCALE_MIN, O.LOG_STEP_RECIP, 0.12)
    idx = torch.empty(sc.shape, dtype=torch.uint8, device="cuda")
    cond = torch.empty(sc.shape, dtype=torch.bool, device="cuda")
    ops.build_index_dec_cuda(idx, cond, sc, *args)
    idx, cond = idx.cpu().numpy(), cond.cpu().numpy()
    assert np.array_equal(cond, g["idx.dec_cond"])
    s = np.clip(g["idx.scales"], np.float32(0.11), np.float32(16.0))
    assert np.array_equal(idx, O.scale_to_index(s, 0.11, 16.0, O.LOG_SCALE_MIN, O.LOG_STEP_RECIP))   # == oracle
    assert np.mean(idx != g["idx.dec_idx"]) <= 1e-3                                                    # ~ reference
    out = torch.empty(sc.shape, dtype=torch.int16, device="cuda")
    ops.build_index_enc_cuda(out, cond_t := torch.empty(sc.shape, dtype=torch.bool, device="cuda"), sym, sc, *args)
    packed = out.cpu().numpy()[cond_t.cpu().numpy()]
    assert np.mean(packed != g["idx.enc_packed"]) <= 1e-3


def test_small_ops_bit_exact(g):
    from opendcvc_amd import ops
    z = cu(g["z.in"])
    z8 = ops.round_and_to_int8_cuda(z)
    assert np.array_equal(z.cpu().numpy(), g["z.hat"]) and np.array_equal(z8.cpu().numpy(), g["z.int8"])
    y = cu(g["pwm.y"])
    q = ops.clamp_reciprocal_with_quant_cuda(cu(g["crq.q"]), y, 0.5)
    assert np.array_equal(q.cpu().numpy(), g["crq.q_out"]) and np.array_equal(y.cpu().numpy(), g["crq.y_out"])
    assert np.array_equal(ops.replicate_pad_cuda(cu(g["pad.x"]), 3, 9).cpu().numpy(), g["pad.y"])
    x = cu(g["ps8.x"])
    out = torch.empty((1, 3, 16, 24), device="cuda")
    ops.bias_pixel_shuffle_8_cuda(out, x, cu(g["ps8.b"]), 192, 6, 3, True)
    assert np.array_equal(out.cpu().numpy(), g["ps8.y"])


def test_mask_collapse_restore_and_scaling_ops():
    from opendcvc_amd import ops
    rng = np.random.default_rng(3)
    C, H, W = 8, 5, 7
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    m = (rng.random((1, C, H, W)) > 0.5).astype(np.float32)
    out = torch.empty((1, C // 2, H, W), device="cuda")
    ops.combine_for_reading_2x_cuda(out, cu(x), cu(m))
    xm = x * m
    assert np.array_equal(out.cpu().numpy(), xm[:, :C // 2] + xm[:, C // 2:])
    for groups, fn in ((2, ops.restore_y_2x_cuda), (4, ops.restore_y_4x_cuda)):
        y = rng.integers(-5, 6, (1, C // groups, H, W)).astype(np.float32)
        out = torch.empty((1, C, H, W), device="cuda")
        fn(out, cu(y), cu(x), cu(m))
        assert np.array_equal(out.cpu().numpy(), (np.concatenate([y] * groups, 1) + x) * m)
    a, b = cu(x), cu(m + 1)
    qq = rng.uniform(0.5, 2, (1, C, H, W)).astype(np.float32)
    ops.add_and_multiply_cuda(a, b, cu(qq))
    assert np.array_equal(a.cpu().numpy(), (x + (m + 1)) * qq)
    bias = rng.standard_normal(C).astype(np.float32)
    qs = rng.uniform(0.5, 2, (1, C, 1, 1)).astype(np.float32)
    t = cu(x)
    ops.bias_quant_cuda(t, cu(bias), cu(qs))
    assert np.array_equal(t.cpu().numpy(), (x + bias[None, :, None, None]) * qs)


def test_bias_wsilu_depthwise_matches_oracle():
    from opendcvc_amd import ops
    rng = np.random.default_rng(4)
    C, H, W = 16, 6, 9
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, 1, 3, 3)) / 3).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    got = ops.bias_wsilu_depthwise_conv2d_cuda(cu(x), cu(w), cu(b)).cpu().numpy()
    act = O.wsilu(O.nchw_to_hwc(x) + b)
    ref = np.empty((H, W, C), np.float32)
    import ctypes
    O.lib().orc_dw3x3(O._ptr(act), H, W, C, O._ptr(w), None, O._ptr(ref))
    assert np.array_equal(got, O.hwc_to_nchw(ref))


def _sub(g, prefix):
    pre = prefix + ".w."
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("name,short", [("dcb_plain", False), ("dcb_adapt", False), ("dcb_short", True),
                                        ("dcb_quant", False), ("dcb_force", False), ("dcb_adapt_short_q", True)])
def test_depth_conv_proxy_vs_reference(g, name, short):
    from opendcvc_amd import ops
    w = {k: cu(v) for k, v in _sub(g, name).items()}
    p = ops.DepthConvProxy()
    base = [w["dc.0.weight"], w["dc.0.bias"], w["dc.2.weight"], w["dc.2.bias"], w["dc.3.weight"], w["dc.3.bias"],
            w["ffn.0.weight"], w["ffn.0.bias"], w["ffn.2.weight"], w["ffn.2.bias"]]
    if "adaptor.weight" in w:
        p.set_param_with_adaptor(*base, w["adaptor.weight"], w["adaptor.bias"], short)
    else:
        p.set_param(*base, short)
    x = cu(g[name + ".x"])
    if (name + ".q") in g.files:
        y = p.forward_with_quant_step(x, cu(g[name + ".q"]))
    else:
        y = p.forward(x)
    np.testing.assert_allclose(y.cpu().numpy(), g[name + ".y"], **TOL)
    cat = p.forward_with_cat(x, x, True) if (name + ".q") not in g.files else None
    if cat is not None:
        assert cat.shape[1] == x.shape[1] + g[name + ".y"].shape[1]


@pytest.mark.parametrize("name,pad", [("subpel1", 0), ("subpel3", 1)])
def test_subpel_proxy_vs_reference(g, name, pad):
    from opendcvc_amd import ops
    w = _sub(g, name)
    p = ops.SubpelConv2xProxy()
    p.set_param(cu(w["conv.0.weight"]), cu(w["conv.0.bias"]), pad)
    np.testing.assert_allclose(p.forward(cu(g[name + ".x"])).cpu().numpy(), g[name + ".y"], **TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("hw", [(16, 16), (64, 264), (136, 1928), (8, 4096)])
def test_unshuffle8_shuffle8_layout_kernels(dtype, hw):
    """dcvc_unshuffle8 == F.pixel_unshuffle(x, 8) in HWC; dcvc_shuffle8_clamp == clamp(F.pixel_shuffle(x + bias, 8))
    (LDS-tiled kernels: full and ragged 32-pixel segments, odd row counts, both storage types)."""
    import ctypes
    from opendcvc_amd import _lib, nn as L
    H, W = hw
    lib = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(H * 7 + W)
    x = (torch.rand((1, 3, H, W), generator=gen) * 2 - 0.5).to(dtype).cuda()
    out = torch.zeros((H // 8, W // 8, 192), dtype=dtype, device="cuda")
    _lib.check(lib.dcvc_unshuffle8(L.dtype_code(dtype), L._p(x), 3, H, W, L._p(out), 192, st), "unshuffle8")
    want = torch.nn.functional.pixel_unshuffle(x, 8)[0].permute(1, 2, 0).contiguous()
    assert torch.equal(out, want)
    bias = torch.rand(192, generator=gen).cuda() - 0.5
    back = torch.empty((1, 3, H, W), dtype=dtype, device="cuda")
    _lib.check(lib.dcvc_shuffle8_clamp(L.dtype_code(dtype), L._p(out), 192, L._p(bias), 3, H // 8, W // 8, 1, L._p(back), st),
               "shuffle8_clamp")
    ref = (out.float() + bias).to(dtype).float().clamp(0, 1).to(dtype)
    ref = torch.nn.functional.pixel_shuffle(ref.permute(2, 0, 1)[None], 8)
    assert torch.equal(back, ref)


@pytest.mark.parametrize("parts,n", [(2, 5000), (4, 777), (2, 522240), (1, 1)])
def test_compact_symbols_writes_kept_entries_in_order_to_pinned_memory(parts, n):
    """dcvc_compact_symbols (the encoder's symbol hand-off without a copy command; replaces the reference's boolean-mask
    compaction + .cpu(), cuda_inference.py:159): kept entries of every part in order, counts, nothing else touched."""
    import ctypes
    from opendcvc_amd import _lib, entropy, nn
    rng = np.random.default_rng(parts * 1000 + n)
    a = rng.integers(-32768, 32767, (parts, n), dtype=np.int16)
    skip = rng.random((parts, n)) < 0.7
    a = np.where(skip, (a & ~0xFF) | 0xFF, np.where((a & 0xFF) == 0xFF, a & ~1, a)).astype(np.int16)
    dev = torch.from_numpy(a).cuda()
    out, cnt = entropy.PinnedBuffer(parts * n * 2), entropy.PinnedBuffer(4 * parts)
    out.u8[:] = 0xAB
    ws = torch.zeros(256 * parts, dtype=torch.int32, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().dcvc_compact_symbols(nn._p(dev), n, parts, ctypes.c_void_p(out.ptr), ctypes.c_void_p(cnt.ptr),
                                               nn._p(ws), st), "compact")
    torch.cuda.synchronize()
    got, kept = out.view(np.int16, parts * n).reshape(parts, n), cnt.view(np.int32, parts)
    for p in range(parts):
        want = a[p][~skip[p]]
        assert kept[p] == want.size
        assert np.array_equal(got[p, :want.size], want)
        assert np.all(got[p, want.size:].view(np.uint8) == 0xAB)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("groups,H,W,C,thres", [(2, 68, 120, 256, 0.12), (2, 17, 30, 256, 0.12), (4, 9, 13, 64, 0.12), (4, 68, 120, 256, 0.3),
                                                (2, 136, 240, 256, 0.12), (2, 20, 24, 128, -1.0), (2, 12, 16, 64, 100.0)])
def test_decoder_hand_off_compacted_on_the_device(dtype, groups, H, W, C, thres):
    """dcvc_prior_dec_index_compact / dcvc_prior_dec_restore_compact (the decoder's hand-off with the kept entries compacted on
    the device; same stream order as the reference's boolean-mask gather, entropy_models.py:330-341) against the whole-array
    entry points: the pinned buffer holds exactly the non-sentinel indexes of dcvc_prior_dec_index in CHW order and their
    count, nothing behind them is touched, and the restore from compacted symbols equals dcvc_prior_dec_restore from the
    scattered array bit for bit.  Cases: HW a multiple of 16 and not, every step of both group counts, several scan rounds per
    block (136x240), everything kept (thres < 0), nothing kept."""
    import ctypes
    from opendcvc_amd import _lib, entropy, nn
    lib = _lib.lib()
    rng = np.random.default_rng(H * W + C + groups)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dt = nn.dtype_code(dtype)
    n = (C // groups) * H * W
    cap = (n + 15) // 16 * 16
    scales = torch.from_numpy(np.exp(rng.normal(-2.4, 1.0, (H, W, C))).astype(np.float32)).to(dtype).cuda()
    means = torch.from_numpy(rng.normal(0, 2, (H, W, C)).astype(np.float32)).to(dtype).cuda()
    prev = torch.from_numpy(rng.normal(0, 2, (H, W, C)).astype(np.float32)).to(dtype).cuda()
    ws_bytes = int(lib.dcvc_prior_dec_compact_ws_bytes(H, W, C, groups))
    assert ws_bytes > cap
    for step in range(groups):
        full = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        _lib.check(lib.dcvc_prior_dec_index(dt, groups, step, nn._p(scales), scales.stride(1), H, W, C, thres, nn._p(full), st), "index")
        idx = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
        hidx, hcnt, hsym = entropy.PinnedBuffer(cap), entropy.PinnedBuffer(16), entropy.PinnedBuffer(cap)
        hidx.u8[:] = 0xAB
        _lib.check(lib.dcvc_prior_dec_index_compact(dt, groups, step, nn._p(scales), scales.stride(1), H, W, C, thres, nn._p(idx),
                                                    nn._p(ws), ctypes.c_void_p(hidx.ptr), ctypes.c_void_p(hcnt.ptr), st), "index_compact")
        torch.cuda.synchronize()
        f = full.cpu().numpy()
        assert np.array_equal(idx.cpu().numpy(), f)
        keep = f != 0xFF
        count = int(hcnt.view(np.int32, 1)[0])
        assert count == int(keep.sum())
        if thres < 0:
            assert count == n
        assert np.array_equal(hidx.u8[:count], f[keep]) and np.all(hidx.u8[count:] == 0xAB)
        sym_full = np.where(keep, rng.integers(-128, 128, n), 0).astype(np.int8)
        hsym.u8[:] = 0x55
        hsym.view(np.int8, cap)[:count] = sym_full[keep]
        want = torch.empty((H, W, C), dtype=dtype, device="cuda")
        got = torch.empty((H, W, C), dtype=dtype, device="cuda")
        sym_dev = torch.from_numpy(sym_full).cuda()
        yin = prev if step else None
        _lib.check(lib.dcvc_prior_dec_restore(dt, groups, step, nn._p(sym_dev), nn._p(means), means.stride(1), H, W, C,
                                              nn._p(yin) if step else None, prev.stride(1), nn._p(want), want.stride(1), st), "restore")
        _lib.check(lib.dcvc_prior_dec_restore_compact(dt, groups, step, ctypes.c_void_p(hsym.ptr), nn._p(idx), nn._p(ws), nn._p(means),
                                                      means.stride(1), H, W, C, nn._p(yin) if step else None, prev.stride(1),
                                                      nn._p(got), got.stride(1), st), "restore_compact")
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        prev = want
