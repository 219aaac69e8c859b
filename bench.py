#!/usr/bin/env python3
"""bench.py - DCVC-RT 1080p YUV420 encode+decode throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame of a 32-frame-GOP 1080p sequence (configs[1]) ENCODED and DECODED by the HIP
path (fp16 storage / fp32 accumulate, like the reference's published numbers), including the host
rANS coding, with the padded input frames already resident in HBM.  The timed region runs the
encoder and the decoder as a two-stage pipeline (two host threads, two HIP streams on the same GPU:
frame n decodes while frame n+1 encodes); after it, the same frames are run one direction at a time
for the per-direction fps (`enc_fps_per_gpu`, `dec_fps_per_gpu`, `sequential_*`), which is how the
reference times them and what `vs_baseline` compares.  With N > 1 every rank
codes its own independent stream (weak scaling, no data-path collective; the weights are
broadcast once from rank 0 over RCCL).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     dominant kernel (dcb_tail_kernel<f16>, C = 256 at 136x240): algorithmic FLOP per
               launch / HIP-event time on its stream, against the 2.5 PFLOP/s dense f16 MFMA peak.
  cpu_baseline the CPU oracle (port of the reference's torch fallback path) timed on the host
               cores on a bounded sample (one 1080p P frame, encode + decode).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from opendcvc_amd import _lib, weights  # noqa: E402
from opendcvc_amd import dist as dist_utils  # noqa: E402
from opendcvc_amd import nn as L  # noqa: E402
from opendcvc_amd.models import DMC, DMCI  # noqa: E402
from opendcvc_amd.pipeline import EncodeDecodePipeline, SequenceDecoder, SequenceEncoder, use_two_entropy_coders  # noqa: E402

HEIGHT, WIDTH = 1080, 1920
GOP = 32
QP = 32
THRES = 0.12
MFMA_F16_PEAK_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
# reference README.md:35 (A100, fp16): 125.2 enc fps / 112.8 dec fps -> one frame through both
BASELINE_ENC_FPS, BASELINE_DEC_FPS = 125.2, 112.8


def load_models(dtype, device, world, rank):
    sds = {}
    for name in ("dmci", "dmc"):
        sd = weights.make_state_dict(name, 1234) if rank == 0 else None
        sds[name] = dist_utils.broadcast_state_dict(name, sd, device, rank, world)   # one RCCL broadcast per model

    def make(cls, name):
        m = cls()
        m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sds[name].items()})
        m.to(device).eval()
        m.update(THRES)
        if dtype == torch.float16:
            m.half()
        return m
    return (make(DMCI, "dmci"), make(DMC, "dmc")), (make(DMCI, "dmci"), make(DMC, "dmc"))


def make_frames(seed, dtype, device):
    pr, pb = DMCI.get_padding_size(HEIGHT, WIDTH, 16)
    frames = []
    for fi in range(GOP):
        x = weights.synthetic_frame_yuv444(HEIGHT, WIDTH, fi, seed)
        x = np.pad(x, ((0, 0), (0, 0), (0, pb), (0, pr)), mode="edge")
        frames.append(torch.from_numpy(x).to(device=device, dtype=dtype))
    return frames


def roofline_leg(p_net, device, dtype):
    """HIP-event timing of the dominant kernel on its own stream."""
    blk = p_net._layers["fe2"][0]          # DepthConvBlock C=256 at H/8 x W/8
    H, W, C = (HEIGHT + (-HEIGHT) % 16) // 8, (WIDTH + (-WIDTH) % 16) // 8, 256     # feature map of the padded frame
    x = (torch.randn((H, W, C), device=device) * 0.5).to(dtype)
    out = torch.empty_like(x)
    lib = _lib.lib()
    scratch = L.Scratch.get(lib.dcvc_dcb_scratch_bytes(blk.h, H, W), device)
    head, tail = ctypes.c_float(), ctypes.c_float()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for iters in (20, 300):     # sustained enough for the clocks to settle (30 launches read ~8 % slower)
        _lib.check(lib.dcvc_dcb_profile(blk.h, L._p(x), C, C, H, W, L._p(out), C, L._p(scratch), st, iters,
                                        ctypes.byref(head), ctypes.byref(tail)), "dcvc_dcb_profile")
    P = H * W
    flop = 2.0 * P * (7 * C * C + 9 * C)               # W2 + W3(4x) + W4(2x) + depthwise, per launch
    achieved = flop / (tail.value * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_dcb_tail.json")
    if os.path.exists(pmc) and (H, W) == (136, 240):     # the PMC passes were taken at this shape
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"kernel": "dcb_tail_kernel<f16,MT=4,NTW=4> (C=256, %dx%d)" % (H, W), "bound": "mfma",
            "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic,
            "flop_per_launch": flop, "kernel_ms": round(tail.value, 4),
            "head_kernel_ms": round(head.value, 4)}


def cpu_baseline_leg():
    """The oracle (CPU port of the reference path) on one 1080p P frame, encode + decode."""
    ncores = min(len(os.sched_getaffinity(0)), 32)
    os.environ["OMP_NUM_THREADS"] = str(ncores)      # read by libgomp when the oracle library loads
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import dcvc_oracle as O
    pr, pb = DMCI.get_padding_size(HEIGHT, WIDTH, 16)
    pad = lambda a: np.pad(a, ((0, 0), (0, 0), (0, pb), (0, pr)), mode="edge")
    x0 = pad(weights.synthetic_frame_yuv444(HEIGHT, WIDTH, 0, 0))
    x1 = pad(weights.synthetic_frame_yuv444(HEIGHT, WIDTH, 1, 0))
    p = O.OracleDMC(weights.make_state_dict("dmc", 1234))
    p.update(THRES)
    p.set_use_two_entropy_coders(True)
    p.clear_dpb()
    p.add_ref_frame(None, x0)
    t0 = time.perf_counter()
    enc = p.compress(x1, QP)
    t1 = time.perf_counter()
    p.clear_dpb()
    p.add_ref_frame(None, x0)
    p.decompress(enc["bit_stream"], dict(height=HEIGHT, width=WIDTH, ec_part=1, use_ada_i=0), QP)
    t2 = time.perf_counter()
    return {"value": round(1.0 / (t2 - t0), 4), "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": "one 1080p (1088x1920 padded) P frame, encode %.2f s + decode %.2f s, fp32 C/OpenMP oracle"
                      % (t1 - t0, t2 - t1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)     # eight whole GOPs (I frames at the GOP's rate, 1 in 32), under a second
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frame", default="1920x1080",
                    help="WxH of the synthetic sequence (default: the BASELINE.json configs[1] size; 3840x2160 = configs[3])")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (what profiles/r01_roofline_kernel_stats.csv was collected on)")
    args = ap.parse_args()

    global HEIGHT, WIDTH
    WIDTH, HEIGHT = (int(v) for v in args.frame.lower().split("x"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _lib.require_gpu()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    dtype = torch.float16
    torch.set_grad_enabled(False)

    (ie, pe), (idec, pdec) = load_models(dtype, device, world, rank)
    two = use_two_entropy_coders(HEIGHT, WIDTH)
    for m in (ie, pe, idec, pdec):
        m.set_use_two_entropy_coders(two)
    if args.roofline_only:
        pe._ensure_layers()
        print(json.dumps({"roofline": roofline_leg(pe, device, dtype)}), flush=True)
        return
    frames = make_frames(rank, dtype, device)
    enc = SequenceEncoder(ie, pe, QP, intra_period=GOP, reset_interval=GOP)
    # decoder output deferred by one frame: the reconstruction network of frame n runs in the host-decoding gaps
    # of frame n+1 (opendcvc_amd/pipeline.py); every timed frame is still completed inside the timed region (flush)
    dec = SequenceDecoder(idec, pdec, HEIGHT, WIDTH, two, defer_output=True)

    state = {"i": 0, "t_enc": 0.0, "t_dec": 0.0, "bytes": 0, "n_i": 0, "j": 0}

    def step_sequential():
        """encode, wait, decode, wait - used for the per-direction fps (how the reference times them)"""
        x = frames[state["i"] % GOP]
        state["i"] += 1
        t0 = time.perf_counter()
        with torch.cuda.stream(s_enc):
            pkt = enc.encode(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(s_dec):
            dec.decode(pkt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        state["t_enc"] += t1 - t0
        state["t_dec"] += t2 - t1

    pipe = EncodeDecodePipeline(enc, dec, device)
    s_enc, s_dec = pipe.enc_stream, pipe.dec_stream

    def run_pipelined(nsteps, timed):
        """nsteps frames through encoder AND decoder (opendcvc_amd/pipeline.py: two host threads, two HIP
        streams; frame n decodes while frame n+1 encodes).  Every frame is fully encoded and fully
        decoded inside the call."""
        first = state["i"]
        state["i"] += nsteps

        def on_packet(pkt):
            if timed:
                state["bytes"] += len(pkt.bit_stream)
                state["n_i"] += int(pkt.is_i)

        def on_frame(_):
            state["j"] += 1

        pipe.run((frames[k % GOP] for k in range(first, first + nsteps)), on_packet, on_frame)

    def barrier():
        dist_utils.barrier(world)
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    run_pipelined(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_pipelined(args.steps, True)
    barrier()
    elapsed = dist_utils.max_over_ranks(time.perf_counter() - t0, device, world)
    assert state["j"] == args.warmup + args.steps

    # outside the timed region: the same frames one direction at a time
    n_seq = min(args.steps, GOP)
    t_seq0 = time.perf_counter()
    for _ in range(n_seq):
        step_sequential()
    with torch.cuda.stream(s_dec):
        dec.flush()
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t_seq0

    if rank == 0:
        K, N = args.steps, world
        value = N * K / elapsed
        seq_value = n_seq / t_seq                   # this rank, encode then decode one after the other
        if (WIDTH, HEIGHT) == (3840, 2160):       # README complexity table, A100 fp16 at 4K: 35.5 / 29.5 fps
            base = 1.0 / (1.0 / 35.5 + 1.0 / 29.5)
        else:
            base = 1.0 / (1.0 / BASELINE_ENC_FPS + 1.0 / BASELINE_DEC_FPS)
        out = {
            "metric": "%s YUV420 encode+decode FPS (frames/s through encode AND decode, whole job)"
                      % ("1080p" if (WIDTH, HEIGHT) == (1920, 1080) else args.frame),
            "value": round(value, 3), "unit": "frames/s", "n_gpus": N, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(seq_value / base, 4), "dtype": "f16", "data": "synthetic",
            "config": {"workload": "DCVC-RT inter-coding, %s YUV420 32-frame GOP, single q (qp 32), one stream per MI355X "
                                   "(BASELINE.json configs[%d])" % (("1080p", 1) if (WIDTH, HEIGHT) == (1920, 1080) else (args.frame, 3)),
                       "frame": "%dx%d padded to %dx%d" % (WIDTH, HEIGHT, WIDTH + (-WIDTH) % 16, HEIGHT + (-HEIGHT) % 16), "intra_period": GOP, "i_frames_timed": state["n_i"],
                       "entropy_coders": 2 if two else 1, "force_zero_thres": THRES,
                       "weights": "synthetic seed 1234 (opendcvc_amd/weights.py)",
                       "pipeline": "encoder and decoder on two host threads / two HIP streams of the same GPU: frame n "
                                   "decodes while frame n+1 encodes; every timed frame is encoded and decoded; the "
                                   "decoder emits P pictures one call late (their reconstruction network fills the next "
                                   "frame's host-decoding gaps)",
                       "baseline_note": "vs_baseline = sequential_fps_per_gpu / (1/(1/125.2+1/112.8)) fps (reference README, "
                                        "A100 fp16, encode and decode timed one after the other as in the reference)"},
            "sequential_fps_per_gpu": round(seq_value, 3), "sequential_ms_per_step": round(1e3 * t_seq / n_seq, 3),
            "enc_fps_per_gpu": round(n_seq / state["t_enc"], 2), "dec_fps_per_gpu": round(n_seq / state["t_dec"], 2),
            "bpp": round(state["bytes"] * 8.0 / (K * HEIGHT * WIDTH), 5),
        }
        # whole-frame fractions (conv-hook GFLOP and fused-unit algorithmic bytes of a steady P frame at
        # 1088x1920, scaled by the padded pixel count) from the per-direction times of the sequential pass
        scale = ((WIDTH + (-WIDTH) % 16) * (HEIGHT + (-HEIGHT) % 16)) / (1920.0 * 1088.0)
        t_enc, t_dec = state["t_enc"] / n_seq, state["t_dec"] / n_seq
        out["frame_roofline"] = {
            "enc_tflops": round(590.4e9 * scale / t_enc / 1e12, 1), "dec_tflops": round(691.6e9 * scale / t_dec / 1e12, 1),
            "enc_frac_mfma": round(590.4e9 * scale / t_enc / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            "dec_frac_mfma": round(691.6e9 * scale / t_dec / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            "enc_frac_hbm": round(0.76e9 * scale / t_enc / 8.0e12, 4), "dec_frac_hbm": round(0.78e9 * scale / t_dec / 8.0e12, 4),
            "note": "P-frame work only; the timed pass includes the GOP's I frame and the host entropy coding"}
        out["roofline"] = roofline_leg(pe, device, dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg()
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
