"""One rate point (or a qp sweep) over one planar 8-bit YUV 4:2:0 sequence on the HIP path, logged in the
reference's JSON schema - SURVEY.md section 8(f)-3.

Restates (no code shared) the behaviour of the reference harness:
  test_video.py:130-353   run_one_point_with_stream: encode every frame into the NAL container, write the
                          .bin, decode it back, distortion per decoded frame, timing, JSON log
  test_video.py:94-127    get_distortion (YUV 4:2:0: per-plane PSNR, combined (6 Y + U + V) / 8)
  test_video.py:448-463   the qp points of a sweep
  src/utils/metrics.py:81-96   calc_psnr
  src/utils/common.py:63-177   generate_log_json
  test_video.py:381-442,472-532   the job fan-out: a JSON dataset manifest, one job per (sequence, rate point), a pool
                          of spawned worker processes (-w), worker n on GPU n % gpu_num, one merged JSON log
  src/utils/common.py:49-60      dump_json (floats with six digits)
  test_video.py:66-127, src/utils/video_reader.py:10-47, transforms.py:27-53, metrics.py:9-79   PNG (RGB) sources: BT.709
                          rgb <-> ycbcr around the codec, RGB PSNR; --calc_ssim: MS-SSIM per plane (YUV: (6 Y + U + V) / 8)
The codec calls are the drop-in DMCI / DMC of opendcvc_amd.models.

    python -m opendcvc_amd.harness --test-config cfg.json -w 16 --gpus 8 --output-path out.json     # configs[4]
"""
import importlib
import io
import json
import math
import os
import time

import numpy as np

from .bitstream import StreamReader, StreamWriter
from .pipeline import (SequenceDecoder, SequenceEncoder, load_yuv420_frame, store_yuv420_frame,
                       use_two_entropy_coders)


# ---------------------------------------------------------------------------------- metrics / log
def psnr_from_mse(mse, data_range=255.0):
    """metrics.py:81-96: -999.9 for nan/inf, 999.9 below 1e-10, capped at 99.9"""
    if math.isnan(mse) or math.isinf(mse):
        return -999.9
    psnr = 10.0 * math.log10(data_range * data_range / mse) if mse > 1e-10 else 999.9
    return min(psnr, 99.9)


def calc_psnr(a, b, data_range=255.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return psnr_from_mse(float(np.mean(np.square(a - b))), data_range)


def yuv420_distortion(x_hat, y, u, v):
    """test_video.py:94-111 on the device: x_hat [1,3,H',W'] (model dtype, cropped here), y/u/v uint8 planes.
    The planes of the reconstruction are the clamped, NOT rounded, values (chroma = 2x2 mean), in the
    reconstruction's own dtype like the reference's tensors; squared errors are summed in float64."""
    import torch
    H, W = y.shape
    x = x_hat[:, :, :H, :W]
    y_rec = torch.clamp(x[:, :1] * 255, 0, 255)
    uv_rec = torch.clamp(torch.nn.functional.avg_pool2d(x[:, 1:], 2) * 255, 0, 255)
    out = []
    for rec, src in ((y_rec[0, 0], y), (uv_rec[0, 0], u), (uv_rec[0, 1], v)):
        out.append(psnr_from_mse(float(torch.mean(torch.square(rec.double() - src.double())))))
    return [(6 * out[0] + out[1] + out[2]) / 8] + out


# MS-SSIM as the reference computes it for --calc_ssim (metrics.py:9-79: the multi-scale SSIM of Wang et al. with an 11x11 Gaussian
# window, sigma 1.5, 'valid' windows, 2x2 box + decimation between the scales, four scales instead of five below 176 pixels -
# "according to HM" - and none below 88).  Host numpy / scipy, like the reference's: a metric of the harness, not the hot path.
_MSSSIM_WEIGHTS = {5: (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), 4: (0.0517, 0.3295, 0.3462, 0.2726)}


def _gauss_window(size=11, sigma=1.5):
    ax = np.arange(size, dtype=np.float64) - size // 2
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma * sigma))
    return g / g.sum()


def _ssim_and_cs(a, b, window, data_range):
    from scipy import signal
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    local = lambda img: signal.fftconvolve(window, img, mode="valid")
    mu_a, mu_b = local(a), local(b)
    var_a, var_b, cov = local(a * a) - mu_a * mu_a, local(b * b) - mu_b * mu_b, local(a * b) - mu_a * mu_b
    cs = (2.0 * cov + c2) / (var_a + var_b + c2)
    return ((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (var_a + var_b + c2)), cs


def calc_msssim(a, b, data_range=255):
    """a, b: 2-D arrays (one plane each)"""
    from scipy import ndimage
    h, w = a.shape
    if h < 88 or w < 88:
        raise ValueError("MS-SSIM needs planes of at least 88 x 88 (the reference asserts)")
    levels = 5 if (h >= 176 and w >= 176) else 4
    weights = np.asarray(_MSSSIM_WEIGHTS[levels])
    window, box = _gauss_window(), np.full((2, 2), 0.25)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ssim_mean, cs_mean = [], []
    for _ in range(levels):
        ssim_map, cs_map = _ssim_and_cs(a, b, window, data_range)
        ssim_mean.append(ssim_map.mean())
        cs_mean.append(cs_map.mean())
        a = ndimage.convolve(a, box, mode="reflect")[::2, ::2]
        b = ndimage.convolve(b, box, mode="reflect")[::2, ::2]
    cs_mean, ssim_mean = np.asarray(cs_mean), np.asarray(ssim_mean)
    with np.errstate(invalid="ignore"):       # (unrelated pictures: a negative contrast term to a fractional power is NaN, as in the reference)
        return float(np.prod(cs_mean[:levels - 1] ** weights[:levels - 1]) * ssim_mean[levels - 1] ** weights[levels - 1])


def calc_msssim_rgb(a, b, data_range=255):
    """a, b: [3, H, W]; the mean over the three planes"""
    return sum(calc_msssim(a[i], b[i], data_range) for i in range(3)) / 3


def rgb_distortion(x_hat, rgb, calc_ssim=False):
    """test_video.py:116-126 on the device: x_hat [1,3,H',W'] YCbCr (model dtype), rgb uint8 [3,H,W] (device) ->
    ([psnr], [msssim]): clamp(ycbcr2rgb(x_hat) * 255, 0, 255) in the reconstruction's own dtype (one fused kernel), squared
    errors in float64; MS-SSIM on the host."""
    import torch
    rec = reconstruct_rgb(x_hat, rgb.shape[1], rgb.shape[2])
    psnr = psnr_from_mse(float(torch.mean(torch.square(rec.double() - rgb.double()))))
    ms = calc_msssim_rgb(rgb.cpu().numpy(), rec.float().cpu().numpy().astype(np.float64)) if calc_ssim else 0.0
    return [psnr], [ms]


def reconstruct_rgb(x_hat, height, width):
    """decoded [1,3,H',W'] YCbCr -> clamp(ycbcr2rgb * 255, 0, 255) of the height x width picture, [3,H,W] in x_hat's dtype
    (transforms.py:41-53, test_video.py:118-119)"""
    import ctypes
    import torch
    from . import _lib
    from . import nn as L
    x = x_hat.contiguous()
    _, _, Hp, Wp = x.shape
    out = torch.empty((3, height, width), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().dcvc_frame_to_rgb(L.dtype_code(x.dtype), L._p(x), Hp, Wp, height, width, L._p(out),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dcvc_frame_to_rgb")
    return out


def load_rgb_frame(rgb, dtype, pad_to=16):
    """uint8 device tensor [3,H,W] (RGB) -> padded YCbCr model input [1,3,H',W'] (one fused kernel; reference:
    np_image_to_tensor + rgb2ycbcr + the cast + replicate_pad, test_video.py:59-63,84-90,179)"""
    import ctypes
    import torch
    from . import _lib
    from . import nn as L
    _, H, W = rgb.shape
    pr, pb = (-W) % pad_to, (-H) % pad_to
    out = torch.empty((1, 3, H + pb, W + pr), dtype=dtype, device=rgb.device)
    _lib.check(_lib.lib().dcvc_rgb_to_frame(L.dtype_code(dtype), L._p(rgb.contiguous()), H, W, pb, pr, L._p(out),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dcvc_rgb_to_frame")
    return out


def yuv420_msssim(x_hat, y, u, v):
    """--calc_ssim on a YUV 4:2:0 source (test_video.py:106-112): MS-SSIM of the planes yuv420_distortion compares, combined
    (6 Y + U + V) / 8; host computation on the clamped, not rounded, planes"""
    import torch
    H, W = y.shape
    x = x_hat[:, :, :H, :W]
    y_rec = torch.clamp(x[:, :1] * 255, 0, 255)[0, 0]
    uv_rec = torch.clamp(torch.nn.functional.avg_pool2d(x[:, 1:], 2) * 255, 0, 255)[0]
    vals = [calc_msssim(src.cpu().numpy(), rec.float().cpu().numpy().astype(np.float64))
            for rec, src in ((y_rec, y), (uv_rec[0], u), (uv_rec[1], v))]
    return [(6 * vals[0] + vals[1] + vals[2]) / 8] + vals


def summarize(frame_pixel_num, test_time, frame_types, bits, psnrs, ssims, verbose=False,
              avg_encoding_time=None, avg_decoding_time=None):
    """The reference's per-sequence log (common.py:63-177): averages over I frames (type 0), P frames and
    all frames of bpp / PSNR / MS-SSIM (plus the Y, U, V components when 4-tuples are given), optional
    per-frame lists, timings.  Same keys, same insertion order."""
    n = len(frame_types)
    P = np.asarray(psnrs, np.float64).reshape(n, -1)
    S = np.asarray(ssims, np.float64).reshape(n, -1)
    B = np.asarray(bits, np.float64)
    T = np.asarray(frame_types)
    yuv = P.shape[1] > 1
    if yuv and not (P.shape[1] == 4 and S.shape[1] == 4):
        raise ValueError("per-frame metrics must be 1 value or (all, y, u, v)")
    comp = (("", 0), ("_y", 1), ("_u", 2), ("_v", 3)) if yuv else (("", 0),)
    is_i = T == 0
    log = {"frame_pixel_num": frame_pixel_num, "i_frame_num": int(is_i.sum()), "p_frame_num": int((~is_i).sum())}

    def block(tag, sel, count):
        log[f"ave_{tag}_frame_bpp"] = float(B[sel].sum() / count / frame_pixel_num) if count else 0
        for kind, M in (("psnr", P), ("msssim", S)):
            log[f"ave_{tag}_frame_{kind}"] = float(M[sel, 0].sum() / count) if count else 0
        for kind, M in (("psnr", P), ("msssim", S)):
            for suffix, k in comp[1:]:
                log[f"ave_{tag}_frame_{kind}{suffix}"] = float(M[sel, k].sum() / count) if count else 0

    if log["i_frame_num"] == 0:
        raise ZeroDivisionError("a sequence log needs at least one I frame")
    block("i", is_i, log["i_frame_num"])
    if verbose:
        log["frame_bpp"] = list(B / frame_pixel_num)
        log["frame_psnr"] = [float(v) for v in P[:, 0]]
        log["frame_msssim"] = [float(v) for v in S[:, 0]]
        log["frame_type"] = list(frame_types)
        for kind, M in (("psnr", P), ("msssim", S)):
            for suffix, k in comp[1:]:
                log[f"frame_{kind}{suffix}"] = [float(v) for v in M[:, k]]
    log["test_time"] = test_time
    block("p", ~is_i, log["p_frame_num"])
    log["ave_all_frame_bpp"] = float(B.sum() / (n * frame_pixel_num))
    log["ave_all_frame_psnr"] = float(P[:, 0].sum() / n)
    log["ave_all_frame_msssim"] = float(S[:, 0].sum() / n)
    if avg_encoding_time is not None and avg_decoding_time is not None:
        log["avg_frame_encoding_time"] = avg_encoding_time
        log["avg_frame_decoding_time"] = avg_decoding_time
    for kind, M in (("psnr", P), ("msssim", S)):
        for suffix, k in comp[1:]:
            log[f"ave_all_frame_{kind}{suffix}"] = float(M[:, k].sum() / n)
    return log


def sweep_qps(rate_num, qp_num=64):
    """test_video.py:452-455: rate_num points spread over the qp range, rounded half up"""
    if not 2 <= rate_num <= qp_num:
        raise ValueError("rate_num must be in [2, qp_num]")
    return [int(i + 0.5) for i in np.linspace(0, qp_num - 1, num=rate_num)]


# ---------------------------------------------------------------------------------- sequence I/O
class YUV420FileReader:
    """planar 8-bit 4:2:0 (video_reader.py:50-90): Y (H x W), U, V (H/2 x W/2) per frame"""

    def __init__(self, path, width, height):
        self.f = open(path, "rb")
        self.w, self.h = width, height

    def read(self):
        ys, cs = self.w * self.h, (self.w // 2) * (self.h // 2)
        buf = self.f.read(ys + 2 * cs)
        if len(buf) < ys + 2 * cs:
            raise EOFError("YUV file ended")
        a = np.frombuffer(buf, np.uint8)
        return (a[:ys].reshape(self.h, self.w), a[ys:ys + cs].reshape(self.h // 2, self.w // 2),
                a[ys + cs:].reshape(self.h // 2, self.w // 2))

    def close(self):
        self.f.close()


class PNGSequenceReader:
    """a directory of im1.png, im2.png, ... or im00001.png, ... (video_reader.py:10-47): uint8 [3, H, W] RGB per frame"""

    def __init__(self, path, width, height, start_num=1):
        names = set(os.listdir(path))
        if "im1.png" in names:
            self.digits = 1
        elif "im00001.png" in names:
            self.digits = 5
        else:
            raise ValueError(f"{path}: unknown image naming convention (expected im1.png ... or im00001.png ...)")
        self.path, self.w, self.h, self.index = path, width, height, start_num

    def read(self):
        from PIL import Image
        name = os.path.join(self.path, "im%s.png" % str(self.index).zfill(self.digits))
        if not os.path.exists(name):
            raise EOFError("PNG sequence ended")
        rgb = np.asarray(Image.open(name).convert("RGB"), np.uint8).transpose(2, 0, 1)
        if rgb.shape != (3, self.h, self.w):
            raise ValueError(f"{name}: {rgb.shape[2]}x{rgb.shape[1]}, expected {self.w}x{self.h}")
        self.index += 1
        return (rgb,)

    def close(self):
        pass


def _to_device(planes, device):
    import torch
    return [torch.from_numpy(np.array(p, copy=True)).to(device) for p in planes]     # (the planes may be read-only views of the file buffer)


# ---------------------------------------------------------------------------------- one rate point
def run_one_point(i_net, p_net, src_path, width, height, frame_num, qp_i, qp_p=None, intra_period=-1,
                  reset_interval=32, bin_path=None, rec_path=None, verbose=0, verbose_json=False, device="cuda:0",
                  src_type="yuv420", calc_ssim=False):
    """Encodes `frame_num` frames of a YUV 4:2:0 file (src_type "yuv420") or of a directory of PNGs ("png": RGB, converted
    to YCbCr around the codec) into the reference's container (optionally written to bin_path), decodes the container again,
    and returns the reference-schema log.  i_net / p_net: DMCI / DMC (weights loaded, .update() called, on `device`,
    optionally .half()).  calc_ssim: MS-SSIM per frame (host computation, slow) instead of zeros.  rec_path: the decoded
    sequence as a planar YUV file / as PNGs in that directory."""
    if src_type not in ("yuv420", "png"):
        raise ValueError(f"src_type {src_type!r}: the reference harness reads 'yuv420' or 'png'")
    png = src_type == "png"
    make_reader = (lambda: PNGSequenceReader(src_path, width, height)) if png else (lambda: YUV420FileReader(src_path, width, height))
    to_input = (lambda planes, dt: load_rgb_frame(planes[0], dt)) if png else (lambda planes, dt: load_yuv420_frame(*planes, dt))
    import torch
    dev = torch.device(device)
    dtype = next(p_net.parameters()).dtype
    two = use_two_entropy_coders(height, width)
    for m in (i_net, p_net):
        m.set_use_two_entropy_coders(two)
    t_start = time.time()
    reader = make_reader()
    enc = SequenceEncoder(i_net, p_net, qp_i, qp_p, intra_period, reset_interval)
    out = io.BytesIO()
    writer = StreamWriter(out)
    frame_types, bits, enc_time, dec_time, psnrs, ssims = [], [], [], [], [], []
    for _ in range(frame_num):
        planes = _to_device(reader.read(), dev)
        torch.cuda.synchronize(dev)
        t0 = time.time()
        pkt = enc.encode(to_input(planes, dtype))
        bits.append(8 * writer.write_frame(height, width, two, pkt))
        torch.cuda.synchronize(dev)
        enc_time.append(time.time() - t0)
        frame_types.append(0 if pkt.is_i else 1)
    reader.close()
    stream = out.getvalue()
    if bin_path:
        with open(bin_path, "wb") as f:
            f.write(stream)

    reader = make_reader()
    stream_reader = StreamReader(io.BytesIO(stream))
    rec = None
    if rec_path and png:
        os.makedirs(rec_path, exist_ok=True)
    elif rec_path:
        rec = open(rec_path, "wb")
    from .pipeline import FramePacket
    dec = SequenceDecoder(i_net, p_net, height, width, two)
    for fi in range(frame_num):
        planes = _to_device(reader.read(), dev)
        torch.cuda.synchronize(dev)
        t0 = time.time()
        sps, is_i, qp, payload = stream_reader.read_frame()
        dec.h, dec.w, dec.two = sps["height"], sps["width"], bool(sps["ec_part"])
        x_hat = dec.decode(FramePacket(is_i, qp, sps["use_ada_i"], payload))
        torch.cuda.synchronize(dev)
        dec_time.append(time.time() - t0)
        if png:
            p_, s_ = rgb_distortion(x_hat, planes[0], calc_ssim)
            psnrs.append(p_)
            ssims.append(s_)
            if rec_path:        # clamp * 255 rounded to uint8 (test_video.py:314-318), names like the source's
                from PIL import Image
                rgb8 = reconstruct_rgb(x_hat, height, width).float().round().to(torch.uint8).cpu().numpy()
                Image.fromarray(rgb8.transpose(1, 2, 0)).save(os.path.join(rec_path, "im%s.png" % str(fi + 1).zfill(reader.digits)))
            continue
        y, u, v = planes
        psnrs.append(yuv420_distortion(x_hat, y, u, v))
        ssims.append(yuv420_msssim(x_hat, y, u, v) if calc_ssim else [0.0, 0.0, 0.0, 0.0])
        if rec is not None:     # clamp * 255, Y rounded, chroma truncated (test_video.py:307-311)
            for plane in store_yuv420_frame(x_hat, height, width):
                rec.write(plane.cpu().numpy().tobytes())
    reader.close()
    if rec is not None:
        rec.close()
    test_time = time.time() - t_start
    avg_e = avg_d = None
    if verbose >= 1 and frame_num > 10:     # the first 10 frames are warm-up (test_video.py:328-333)
        avg_e = sum(enc_time[10:]) / len(enc_time[10:])
        avg_d = sum(dec_time[10:]) / len(dec_time[10:])
    return summarize(height * width, test_time, frame_types, bits, psnrs, ssims, verbose=verbose_json,
                     avg_encoding_time=avg_e, avg_decoding_time=avg_d)


def run_sweep(make_nets, src_path, width, height, frame_num, rate_num=4, qp_i=None, qp_p=None, bin_prefix=None, **kw):
    """RD points of one sequence (test_video.py:448-463 + the per-point loop :472-510): {qp_i: log}.
    make_nets() -> (i_net, p_net) ready to use; one pair codes every point, like a reference worker.
    bin_prefix: write each point's container to <prefix>_q<qp_i>.bin (the reference's naming, test_video.py:367)."""
    qi = list(qp_i) if qp_i is not None else sweep_qps(rate_num)
    qp = list(qp_p) if qp_p is not None else qi
    i_net, p_net = make_nets()
    out = {}
    for q, qq in zip(qi, qp):
        out[q] = run_one_point(i_net, p_net, src_path, width, height, frame_num, q, qq,
                               bin_path=f"{bin_prefix}_q{q}.bin" if bin_prefix else None, **kw)
        out[q].update(qp_i=q, qp_p=qq)          # the keys test_video.worker adds to a point's result (:373-376)
    return out


# ---------------------------------------------------------------------------------- job fan-out (test_video.py main)
def dump_json(obj, fp, float_digits=6, indent=2):
    """The reference's log writer (common.py:49-60): json.dump with every float printed with `float_digits` digits."""
    enc = json.JSONEncoder(indent=indent)
    it = json.encoder._make_iterencode({}, enc.default, json.encoder.encode_basestring_ascii, " " * indent,
                                       lambda o: format(o, ".%df" % float_digits), enc.key_separator, enc.item_separator,
                                       enc.sort_keys, enc.skipkeys, False)
    for chunk in it(obj, 0):
        fp.write(chunk)


def jobs_from_config(config, opts):
    """One job per (dataset, sequence, rate point), in the reference's submission order (test_video.py:472-510).
    config: the parsed dataset manifest (dataset_config_example_yuv420.json); opts: see run_config()."""
    qi = list(opts["qp_i"]) if opts.get("qp_i") else sweep_qps(opts.get("rate_num", 4))
    qp = list(opts["qp_p"]) if opts.get("qp_p") else qi
    assert len(qi) == len(qp)
    root = opts.get("force_root_path") or config["root_path"]
    jobs = []
    for ds_name, ds in config["test_classes"].items():
        if ds["test"] == 0:
            continue
        if ds["src_type"] not in ("yuv420", "png"):
            raise ValueError(f"{ds_name}: src_type {ds['src_type']!r} (the reference harness reads 'yuv420' or 'png')")
        for seq, info in ds["sequences"].items():
            for rate_idx, (q, qq) in enumerate(zip(qi, qp)):
                ip = info["intra_period"]
                if opts.get("force_intra"):                      # every frame an I frame (test_video.py:490-491)
                    ip = 1
                if opts.get("force_intra_period", 0) > 0:
                    ip = opts["force_intra_period"]
                fn = opts["force_frame_num"] if opts.get("force_frame_num", 0) > 0 else info["frames"]
                jobs.append(dict(ds_name=ds_name, seq=seq, rate_idx=rate_idx, qp_i=q, qp_p=qq,
                                 src_path=os.path.join(root, ds["base_path"], seq), src_width=info["width"],
                                 src_height=info["height"], frame_num=fn, intra_period=ip,
                                 reset_interval=opts.get("reset_interval", 32), src_type=ds["src_type"]))
    return jobs


def merge_results(config, results):
    """{dataset: {sequence: {"000": result, "001": ...}}} (test_video.py:517-528)"""
    log = {}
    for ds_name, ds in config["test_classes"].items():
        if ds["test"] == 0:
            continue
        log[ds_name] = {seq: {} for seq in ds["sequences"]}
    for res in results:
        log[res["ds_name"]][res["seq"]][f"{res['rate_idx']:03d}"] = res
    return log


_WORKER = {}


def worker_gpu(process_name, gpu_num):
    """test_video.py:384-388: worker process n (the number at the end of its multiprocessing name) codes on GPU
    n % gpu_num; -1 without GPUs."""
    idx = int(process_name[process_name.rfind("-") + 1:])
    return idx % gpu_num if gpu_num > 0 else -1


MIN_WORKER_CPUS = 3      # codec thread + the two rANS worker threads of a 1080p stream


def worker_cpus(idx, workers, allowed, quota=None):
    """CPU set of pool worker `idx` of `workers`: its contiguous slice of the allowed CPUs; with a cgroup quota below the
    mask only as many CPUs of the slice as its share of the quota is worth (spreading wider buys no CPU time) - but never
    fewer than MIN_WORKER_CPUS (neighbouring workers then share some)."""
    n = len(allowed)
    lo, hi = idx * n // workers, (idx + 1) * n // workers
    if quota is not None:
        hi = min(hi, lo + max(MIN_WORKER_CPUS, quota // workers))
    if hi - lo < MIN_WORKER_CPUS:
        lo = max(0, min(lo, n - MIN_WORKER_CPUS))
        hi = min(n, lo + MIN_WORKER_CPUS)
    return allowed[lo:hi] or allowed


def visible_gpu_ids(env=None):
    """The physical device ids the parent process was given (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES, a scheduler's assignment), in order; None if none is set.  Workers index THIS list
    (the reference's --cuda_idx, test_video.py:389-393) - worker n must not land on physical GPU n when the job was
    given GPUs 4,5."""
    env = os.environ if env is None else env
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v:
            ids = [t.strip() for t in v.split(",") if t.strip()]
            if ids:
                return ids
    return None


def count_gpus():
    """number of AMD GPUs the workers may use, without a HIP call in the parent: the visible-devices list if there is
    one, else the render nodes in sysfs"""
    ids = visible_gpu_ids()
    if ids is not None:
        return len(ids)
    from . import dist
    return len(dist.gpu_local_cpus())


def _init_worker(opts, gpu_num):
    """Runs once in every spawned worker (test_video.py:381-414): picks the GPU, pins the process, loads both models."""
    import multiprocessing
    gpu = worker_gpu(multiprocessing.current_process().name, gpu_num)
    if gpu >= 0:
        ids = opts.get("gpu_ids")
        os.environ["HIP_VISIBLE_DEVICES"] = str(ids[gpu] if ids else gpu)     # before the first GPU call of the process
    workers_here = max(1, opts.get("workers", 1))
    try:
        # Every worker stays on its slice of the allowed CPUs, never on fewer than MIN_WORKER_CPUS of them: a worker runs the
        # codec thread plus two rANS worker threads, and a cgroup cpu.max quota limits CPU TIME, not parallelism - cutting a
        # slice down to quota // workers (one CPU at -w 16 on a 16-CPU grant) would put the host coder behind the kernel
        # launches it is meant to overlap.  Slices of neighbouring workers overlap when there are fewer CPUs than that.
        from . import dist
        idx = (int(multiprocessing.current_process().name.rsplit("-", 1)[1]) - 1) % workers_here
        os.sched_setaffinity(0, worker_cpus(idx, workers_here, sorted(os.sched_getaffinity(0)), dist.cgroup_cpu_quota()))
    except (OSError, ValueError, ImportError):
        pass
    mod, fn = opts.get("codec", "opendcvc_amd.harness:default_nets").split(":")
    _WORKER["gpu"] = gpu
    _WORKER["nets"] = getattr(importlib.import_module(mod), fn)(opts)
    mod, fn = opts.get("runner", "opendcvc_amd.harness:run_job").split(":")
    _WORKER["run"] = getattr(importlib.import_module(mod), fn)
    _WORKER["opts"] = opts


def default_nets(opts):
    """(DMCI, DMC) ready to code: checkpoints if given (reference keys), else the synthetic weights; on cuda:0 of the
    worker (its HIP_VISIBLE_DEVICES names one GPU); fp16 like the reference harness unless opts['fp32']."""
    import torch
    from . import weights
    from .models import DMC, DMCI
    torch.set_num_threads(1)          # src/utils/common.py:23
    nets = []
    for cls, name, path in ((DMCI, "dmci", opts.get("model_i")), (DMC, "dmc", opts.get("model_p"))):
        m = cls()
        if path:
            ck = torch.load(path, map_location="cpu", weights_only=True)
            ck = ck.get("state_dict", ck)
            ck = ck.get("net", ck)
            m.load_state_dict({k[7:] if k.startswith("module.") else k: v for k, v in ck.items()})
        else:
            m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
                               weights.make_state_dict(name, 1234).items()})
        m.to("cuda:0").eval()
        m.update(opts.get("force_zero_thres", 0.12))
        if not opts.get("fp32"):
            m.half()
        nets.append(m)
    return nets


def run_job(nets, job, opts):
    """one (sequence, rate point) on this worker's GPU -> the reference-schema log of the point (test_video.py:354-379 worker +
    :130-137, 251-255, 345-346): with a stream folder the point's container <stream_path>/<dataset>/<sequence>_q<qp>.bin and its log
    <...>.json are written; check_existing returns the stored log of a point whose container and log exist with the right
    frame count instead of coding it again; save_decoded_frame writes the reconstruction beside them (<...>.yuv, or PNGs in
    the dataset's stream folder for PNG sources)."""
    bin_path = json_path = rec_path = None
    if opts.get("stream_path"):
        folder = os.path.join(opts["stream_path"], job["ds_name"])
        os.makedirs(folder, exist_ok=True)
        bin_path = os.path.join(folder, f"{job['seq']}_q{job['qp_i']}.bin")         # test_video.py:364-367
        json_path = bin_path[:-4] + ".json"
        if opts.get("save_decoded_frame"):
            rec_path = folder if job.get("src_type") == "png" else bin_path[:-4] + ".yuv"
        if opts.get("check_existing") and os.path.exists(json_path) and os.path.exists(bin_path):
            with open(json_path) as f:
                log = json.load(f)
            if log.get("i_frame_num", 0) + log.get("p_frame_num", 0) == job["frame_num"]:
                return log
            print(f"incorrect log for {json_path}, try to rerun.")
    log = run_one_point(nets[0], nets[1], job["src_path"], job["src_width"], job["src_height"], job["frame_num"],
                        job["qp_i"], job["qp_p"], intra_period=job["intra_period"], reset_interval=job["reset_interval"],
                        bin_path=bin_path, rec_path=rec_path, verbose=opts.get("verbose", 0),
                        verbose_json=opts.get("verbose_json", False), device="cuda:0", src_type=job.get("src_type", "yuv420"),
                        calc_ssim=bool(opts.get("calc_ssim")))
    if json_path:
        with open(json_path, "w") as f:
            json.dump(log, f, indent=2)
    return log


def _worker(job):
    res = _WORKER["run"](_WORKER["nets"], job, _WORKER["opts"])
    res["ds_name"], res["seq"], res["rate_idx"] = job["ds_name"], job["seq"], job["rate_idx"]      # test_video.py:371-376
    res["qp_i"], res["qp_p"] = job["qp_i"], job["qp_p"]
    if _WORKER["opts"].get("record_gpu"):
        res["gpu"] = _WORKER["gpu"]
    return res


def run_config(config, opts, workers=1, gpus=1):
    """The reference's main loop (test_video.py:417-532): every (sequence, rate point) of the manifest as a job on a pool
    of `workers` spawned processes, worker n on GPU n % gpus (so -w may exceed the GPU count: two streams per GPU
    give ~16 % more frames/s on an MI355X, profiles/r02_multistream.txt); returns the merged log."""
    import concurrent.futures
    import multiprocessing
    jobs = jobs_from_config(config, opts)
    opts = dict(opts, workers=workers)
    ctx = multiprocessing.get_context("spawn")
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers, mp_context=ctx, initializer=_init_worker,
                                                initargs=(opts, gpus)) as pool:
        results = [f.result() for f in [pool.submit(_worker, j) for j in jobs]]
    return merge_results(config, results)


def _str2bool(v):
    """the reference's str2bool (test_video.py:24-28): --flag 1 / true / yes / y / t"""
    if isinstance(v, bool):
        return v
    if str(v).lower() in ("yes", "y", "true", "t", "1"):
        return True
    if str(v).lower() in ("no", "n", "false", "f", "0"):
        return False
    raise ValueError("boolean value expected, got %r" % (v,))


def build_parser():
    """The command line.  Every option of the reference's test_video.py (parse_args, test_video.py:30-56) is accepted under
    its own spelling and value convention too (`--test_config`, `--model_path_i`, `--write_stream 1`, `--cuda_idx 0 1`, ...):
    the command of the reference's README runs unchanged as `python -m opendcvc_amd.harness ...`."""
    import argparse
    ap = argparse.ArgumentParser(description="DCVC-RT rate points of one YUV 4:2:0 sequence on the MI355X path")
    flag = dict(nargs="?", const=True, default=False, type=_str2bool)       # `--flag` or the reference's `--flag True`
    ap.add_argument("--test-config", "--test_config", help="JSON dataset manifest (reference: dataset_config_example_yuv420.json): every "
                    "sequence x rate point becomes a job on the worker pool (reference: test_video.py --test_config)")
    ap.add_argument("-w", "--worker", type=int, default=1, help="worker processes (may exceed --gpus)")
    ap.add_argument("--gpus", type=int, default=None, help="GPUs to spread the workers over (default: all visible)")
    ap.add_argument("--gpu-ids", type=lambda v: [t for t in v.split(",") if t], default=None,
                    help="physical device ids for the workers, comma separated (reference: --cuda_idx); default: the "
                         "parent's HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES list, else 0..gpus-1")
    ap.add_argument("--cuda_idx", type=int, nargs="+", default=None, help="the reference's spelling of --gpu-ids: 0 1 2 ...")
    ap.add_argument("--cuda", **dict(flag, default=True), help="accepted for the reference's command line; there is no CPU path: "
                    "--cuda 0 is an error")
    ap.add_argument("--write_stream", "--write-stream", **flag, help="the reference's switch for writing the containers (to "
                    "--stream_path, default out_bin); here giving --stream-path is enough")
    ap.add_argument("--force-root-path", "--force_root_path")
    ap.add_argument("--force-frame-num", "--force_frame_num", type=int, default=-1)
    ap.add_argument("--force-intra-period", "--force_intra_period", type=int, default=-1)
    ap.add_argument("--stream-path", "--stream_path", help="write every point's container to <stream-path>/<dataset>/<sequence>_q<qp>.bin")
    ap.add_argument("--output-path", "--output_path", help="merged JSON log of the manifest run")
    ap.add_argument("--calc-ssim", "--calc_ssim", **flag, help="MS-SSIM per frame (reference --calc_ssim; host computation, slow)")
    ap.add_argument("--force-intra", "--force_intra", **flag, help="every frame an I frame (reference --force_intra)")
    ap.add_argument("--check-existing", "--check_existing", **flag,
                    help="with --stream-path: do not code a point again whose .bin and .json exist (reference --check_existing)")
    ap.add_argument("--save-decoded-frame", "--save_decoded_frame", **flag,
                    help="with --stream-path: write the reconstruction beside the .bin (reference --save_decoded_frame)")
    ap.add_argument("--src-type", choices=("yuv420", "png"), default="yuv420",
                    help="--src is a planar 8-bit YUV 4:2:0 file, or a directory of im1.png ... / im00001.png ... (RGB)")
    ap.add_argument("--src")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--frames", type=int)
    ap.add_argument("--rate-num", "--rate_num", type=int, default=4)
    ap.add_argument("--qp-i", "--qp_i", type=int, nargs="*")
    ap.add_argument("--qp-p", "--qp_p", type=int, nargs="*")
    ap.add_argument("--intra-period", type=int, default=-1)
    ap.add_argument("--reset-interval", "--reset_interval", type=int, default=32)
    ap.add_argument("--model-i", "--model_path_i", help="DMCI checkpoint (.pth.tar); synthetic weights if omitted")
    ap.add_argument("--model-p", "--model_path_p", help="DMC checkpoint")
    ap.add_argument("--force-zero-thres", "--force_zero_thres", type=float, default=0.12)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--bin-prefix", help="write <prefix>_q<qp>.bin")
    ap.add_argument("--out", help="JSON output path (default: stdout)")
    ap.add_argument("--verbose", type=int, default=1)
    ap.add_argument("--verbose-json", "--verbose_json", **flag, help="per-frame lists in the log (reference --verbose_json)")
    return ap


def manifest_options(args, ap):
    """(opts, gpus) of a manifest run from the parsed command line (both spellings)"""
    if not args.cuda:
        ap.error("--cuda 0: this framework has no CPU path (the reference's torch fallback is what oracle/ restates for the tests)")
    gpu_ids = args.gpu_ids or ([str(i) for i in args.cuda_idx] if args.cuda_idx else None) or visible_gpu_ids()
    gpus = args.gpus if args.gpus is not None else (len(gpu_ids) if gpu_ids else count_gpus())
    if gpu_ids and gpus > len(gpu_ids):
        ap.error("--gpus %d but only %d device ids are given / visible (%s)" % (gpus, len(gpu_ids), ",".join(gpu_ids)))
    stream_path = args.stream_path or ("out_bin" if args.write_stream else None)      # (the reference's default folder)
    opts = dict(gpu_ids=gpu_ids, rate_num=args.rate_num, qp_i=args.qp_i, qp_p=args.qp_p, force_root_path=args.force_root_path,
                force_frame_num=args.force_frame_num, force_intra_period=args.force_intra_period,
                reset_interval=args.reset_interval, model_i=args.model_i, model_p=args.model_p,
                force_zero_thres=args.force_zero_thres, fp32=args.fp32, stream_path=stream_path,
                verbose=args.verbose, verbose_json=args.verbose_json, calc_ssim=args.calc_ssim,
                force_intra=args.force_intra, check_existing=args.check_existing, save_decoded_frame=args.save_decoded_frame)
    return opts, gpus


def main(argv=None):
    import torch
    from . import weights
    from .models import DMC, DMCI
    ap = build_parser()
    args = ap.parse_args(argv)
    if args.test_config:
        with open(args.test_config) as f:
            config = json.load(f)
        opts, gpus = manifest_options(args, ap)
        t0 = time.time()
        log = run_config(config, opts, workers=args.worker, gpus=gpus)
        out_path = args.output_path or args.out
        if out_path:
            os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
            with open(out_path, "w") as f:
                dump_json(log, f, float_digits=6, indent=2)
        else:
            import sys
            dump_json(log, sys.stdout, float_digits=6, indent=2)
        print(f"\nTest finished: {sum(len(v) for d in log.values() for v in d.values())} points, "
              f"{(time.time() - t0) / 60:.1f} min")
        return
    if not (args.src and args.width and args.height and args.frames):
        ap.error("either --test-config or --src/--width/--height/--frames")

    def make_nets():
        nets = []
        for cls, name, path in ((DMCI, "dmci", args.model_i), (DMC, "dmc", args.model_p)):
            m = cls()
            if path:
                ck = torch.load(path, map_location="cpu", weights_only=True)
                ck = ck.get("state_dict", ck)
                ck = ck.get("net", ck)
                m.load_state_dict({k[7:] if k.startswith("module.") else k: v for k, v in ck.items()})
            else:
                m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
                                   weights.make_state_dict(name, 1234).items()})
            m.to("cuda").eval()
            m.update(args.force_zero_thres)
            if not args.fp32:
                m.half()
            nets.append(m)
        return nets

    res = run_sweep(make_nets, args.src, args.width, args.height, args.frames, args.rate_num,
                    args.qp_i or None, args.qp_p or None, bin_prefix=args.bin_prefix,
                    intra_period=args.intra_period, reset_interval=args.reset_interval, verbose=args.verbose,
                    verbose_json=args.verbose_json, src_type=args.src_type, calc_ssim=args.calc_ssim)
    text = json.dumps({str(k): v for k, v in res.items()}, indent=2)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
