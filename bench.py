#!/usr/bin/env python3
"""bench.py - DCVC-RT 1080p YUV420 encode+decode throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame of a 32-frame-GOP 1080p sequence (configs[1]) ENCODED and DECODED by the HIP
path (fp16 storage / fp32 accumulate, like the reference's published numbers), including the host
rANS coding, with the padded input frames already resident in HBM.  The timed region runs the
encoder and the decoder as a two-stage pipeline (two host threads, two HIP streams on the same GPU:
frame n decodes while frame n+1 encodes).  Whatever --steps is, the timed window carries I frames at
(at least) the GOP's rate: the window is placed in the GOP so that an I frame falls inside it
(steps < 32: one I frame in the middle, i.e. MORE than the 1-in-32 share; whole GOPs otherwise) - the
frames needed to get there are extra untimed warm-up (`config.alignment_frames`).  After the timed
region one whole GOP is run one direction at a time for the per-direction fps (`enc_fps_per_gpu`,
`dec_fps_per_gpu`, `sequential_*`), which is how the reference times them and what `vs_baseline`
compares, and for bpp / PSNR (per plane and (6Y+U+V)/8, against the 8-bit source planes, the
reference's get_distortion).  With N > 1 every rank
codes its own independent stream (weak scaling, no data-path collective; the weights are
broadcast once from rank 0 over RCCL).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     dominant kernel (dcb_tail128_kernel<256>: the DepthConvBlock tail, C = 256 at 136x240): algorithmic FLOP per
               launch / HIP-event time on its stream, against the 2.5 PFLOP/s dense f16 MFMA peak.
  cpu_baseline the CPU oracle (port of the reference's torch fallback path) on the host cores, real 1080p P frames: all
               cores 1 warm-up + 5 timed (~25 s) = `value`; one thread (what the reference harness pins,
               src/utils/common.py:23) one timed frame (~30 s).
  exact_mode   the fp32 mode that is bit-exact with the CPU oracle: P-frame encode / decode fps and the fraction of the
               157.3 TFLOP/s fp32 MFMA peak.
  gop_weighted_value   32 / (t_I + 31 t_P) from the timed window: does not depend on --steps.
  reference_gop        bpp / PSNR of the same GOP coded by the reference in .half() on a CPU (fixture tests/golden/bench_gop_f16.json)
               and this line's deviation from it.
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
from opendcvc_amd.pipeline import (EncodeDecodePipeline, SequenceDecoder, SequenceEncoder, load_yuv420_frame,  # noqa: E402
                                   use_two_entropy_coders)

HEIGHT, WIDTH = 1080, 1920
GOP = 32
QP = 32
THRES = 0.12
MFMA_F16_PEAK_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
# reference README.md:35 (A100, fp16): 125.2 enc fps / 112.8 dec fps -> one frame through both
BASELINE_ENC_FPS, BASELINE_DEC_FPS = 125.2, 112.8


WEIGHTS_VIA = {"how": "generated in process"}


def load_models(dtype, device, world, rank, coll_device=None):
    """Rank 0 generates the weights, every other rank receives them in ONE RCCL broadcast per model (the only
    collective of the path, BASELINE.json north_star).  Should the broadcast itself fail on a node (a fabric / RCCL
    problem, not a codec one) the ranks fall back to generating the same deterministic weights themselves and the line
    says so (`rccl_ranks` 0, `config.weights_via`) - the throughput measurement does not depend on how the weights arrived."""
    sds = {}
    for name in ("dmci", "dmc"):
        sd = weights.make_state_dict(name, 1234) if rank == 0 else None
        if world > 1 and WEIGHTS_VIA["how"] != "generated per rank":
            try:
                sd = dist_utils.broadcast_state_dict(name, sd, coll_device or device, rank, world)   # one RCCL broadcast per model
                WEIGHTS_VIA["how"] = ("rccl" if torch.distributed.get_backend() == "nccl" else torch.distributed.get_backend()) + " broadcast from rank 0"
            except Exception as e:                                                   # noqa: BLE001
                print("bench.py rank %d: weight broadcast failed (%s: %s); generating the weights locally" %
                      (rank, type(e).__name__, e), file=sys.stderr, flush=True)
                WEIGHTS_VIA["how"] = "generated per rank"
        sds[name] = sd if sd is not None else weights.make_state_dict(name, 1234)

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
    """One GOP of synthetic planar 8-bit YUV 4:2:0 frames (SURVEY 8d recipe), uploaded once: the uint8 planes (kept
    for the PSNR) and the padded model inputs made from them by the fused loader kernel (get_src_frame +
    replicate_pad of the reference, test_video.py:74-91,179)."""
    planes, frames = [], []
    for fi in range(GOP):
        yuv = [torch.from_numpy(a).to(device) for a in weights.synthetic_frame_yuv420(HEIGHT, WIDTH, fi, seed)]
        planes.append(yuv)
        frames.append(load_yuv420_frame(yuv[0], yuv[1], yuv[2], dtype))
    return planes, frames


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
    burst = ctypes.c_float()
    for iters in (20, 300):     # sustained enough for the clocks to settle (30 launches read ~8 % slower)
        _lib.check(lib.dcvc_dcb_profile(blk.h, L._p(x), C, C, H, W, L._p(out), C, L._p(scratch), st, iters,
                                        ctypes.byref(head), ctypes.byref(tail)), "dcvc_dcb_profile")
        # the kernel alone, `iters` launches between ONE pair of events: no event (= dispatch gap) between the launches
        _lib.check(lib.dcvc_dcb_profile_tail(blk.h, L._p(x), C, C, H, W, L._p(out), C, L._p(scratch), st, iters,
                                             ctypes.byref(burst)), "dcvc_dcb_profile_tail")
    P = H * W
    flop = 2.0 * P * (7 * C * C + 9 * C)               # W2 + W3(4x) + W4(2x) + depthwise, per launch
    achieved = flop / (burst.value * 1e-3) / 1e12
    traffic = None
    for name in ("r04_pmc_dcb_tail.json", "r03_pmc_dcb_tail.json"):      # the latest PMC passes of this kernel (tools/final_measure.sh)
        pmc = os.path.join(REPO, "profiles", name)
        if traffic is None and os.path.exists(pmc) and (H, W) == (136, 240):     # the PMC passes were taken at this shape
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
    return {"kernel": "dcb_tail128_kernel<256> (DepthConvBlock tail, f16, C=256, %dx%d, 128-pixel tiles)" % (H, W), "bound": "mfma",
            "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic,
            "flop_per_launch": flop, "kernel_ms": round(burst.value, 4),
            "kernel_ms_event_per_launch": round(tail.value, 4), "head_kernel_ms": round(head.value, 4),
            "timing": "kernel_ms = 300 back-to-back launches of the kernel between one pair of HIP events on its stream "
                      "(compare: rocprofv3 --kernel-trace --stats average, profiles/); kernel_ms_event_per_launch = with an "
                      "event recorded between every two kernels (includes the dispatch gap an event costs)"}


def window_start(steps):
    """Frame index (mod GOP) at which the timed window starts: whole GOPs start on their I frame, a shorter window
    gets the I frame in its middle."""
    return (GOP - (steps % GOP) // 2) % GOP


def measure(run, steps, warmup, world, device, sync):
    """The timed region of the driver contract: `warmup` (+ the frames needed to place the window in the GOP) untimed
    steps, barrier + device sync, EXACTLY `steps` steps, barrier + device sync, MAX over ranks.
    run(n, timed) codes n frames; sync() waits for the device.  Returns (elapsed seconds = MAX over ranks, alignment
    frames, this rank's own elapsed seconds before the closing barrier)."""
    align = (window_start(steps) - warmup) % GOP
    sync()
    run(warmup + align, False)
    dist_utils.barrier(world)
    sync()
    t0 = time.perf_counter()
    run(steps, True)
    sync()
    local = time.perf_counter() - t0          # this rank's own K steps (per-rank fps on the line)
    dist_utils.barrier(world)
    sync()
    return dist_utils.max_over_ranks(time.perf_counter() - t0, device, world), align, local


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup CPU quota if there is one (a GPU box
    shows all 256 hardware threads of the host but grants a 16-CPU share) and to 32 (the oracle's OpenMP loops are
    not tuned beyond that)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline_leg():
    """The oracle (CPU port of the reference path, fp32 C/OpenMP) on the host cores, on REAL 1080p P frames
    (padded 1088x1920), encode + decode each.  All-cores leg = `value`: 1 warm-up + 5 timed frames (~4 s per frame on 16
    cores).  One-thread leg (what the reference harness pins, src/utils/common.py:23): ONE timed frame of the same
    workload, no warm-up (~30 s) - a measurement of the stated workload, not a scaled crop."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import dcvc_oracle as O
    O.lib()
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    ncores = usable_cores()
    H, W = HEIGHT + (-HEIGHT) % 16, WIDTH + (-WIDTH) % 16

    def leg(nt, warm, timed):
        if gomp is not None:
            gomp.omp_set_num_threads(nt)
        frames = [weights.synthetic_frame_yuv444(H, W, fi, 0) for fi in range(warm + timed + 1)]
        sps = dict(height=H, width=W, ec_part=int(use_two_entropy_coders(H, W)), use_ada_i=0)
        enc_net = O.OracleDMC(weights.make_state_dict("dmc", 1234))
        dec_net = O.OracleDMC(weights.make_state_dict("dmc", 1234))
        for m in (enc_net, dec_net):
            m.update(THRES)
            m.set_use_two_entropy_coders(bool(sps["ec_part"]))
            m.clear_dpb()
            m.add_ref_frame(None, frames[0])
        times = []
        for fi in range(1, warm + timed + 1):
            t0 = time.perf_counter()
            e = enc_net.compress(frames[fi], QP)
            dec_net.decompress(e["bit_stream"], sps, QP)
            times.append(time.perf_counter() - t0)
        return timed / sum(times[warm:])

    full = leg(ncores, 1, 5)
    one = leg(1, 0, 1) if (gomp is not None and ncores > 1) else None
    return {"value": round(full, 4), "unit": "frames/s", "cores": ncores, "kind": "port",
            "one_thread_value": None if one is None else round(one, 5), "cpu_model": cpu_model(),
            "sample": "all cores: 1 warm-up + 5 timed P frames of the SAME workload (%dx%d padded to %dx%d), encode + decode, "
                      "fp32 C/OpenMP oracle (port of the reference's torch fallback path); one_thread_value: ONE timed P frame "
                      "of the same workload on one thread (the reference harness pins one thread per worker, "
                      "src/utils/common.py:23), no warm-up" % (WIDTH, HEIGHT, W, H)}


def exact_mode_leg(device, world, rank, frames16):
    """The fp32 'exact' mode (bit-exact with the CPU oracle: the repo's cross-device bit-exact mode)
    on the same frames: 1 I + 6 P frames, encoder loop then decoder loop, every frame synchronised; fps over the P frames
    and the fraction of the 157.3 TFLOP/s fp32 MFMA peak their conv work stands for."""
    (ie, pe), (idec, pdec) = load_models(torch.float32, device, world, rank)
    two = use_two_entropy_coders(HEIGHT, WIDTH)
    for m in (ie, pe, idec, pdec):
        m.set_use_two_entropy_coders(two)
    n = 7
    frames = [f.float() for f in frames16[:n]]
    enc = SequenceEncoder(ie, pe, QP, intra_period=GOP, reset_interval=GOP)
    dec = SequenceDecoder(idec, pdec, HEIGHT, WIDTH, two)
    pkts, t_enc, t_dec = [], [], []
    for rep in range(2):                      # the first pass captures the graphs
        enc = SequenceEncoder(ie, pe, QP, intra_period=GOP, reset_interval=GOP)
        dec = SequenceDecoder(idec, pdec, HEIGHT, WIDTH, two)
        pkts, t_enc, t_dec = [], [], []
        for k in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pkts.append(enc.encode(frames[k]))
            torch.cuda.synchronize()
            t_enc.append(time.perf_counter() - t0)
        for pkt in pkts:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode(pkt)
            torch.cuda.synchronize()
            t_dec.append(time.perf_counter() - t0)
    scale = ((WIDTH + (-WIDTH) % 16) * (HEIGHT + (-HEIGHT) % 16)) / (1920.0 * 1088.0)
    te, td = sum(t_enc[2:]) / (n - 2), sum(t_dec[2:]) / (n - 2)       # steady P frames (the one after the I frame differs)
    return {"enc_fps": round(1.0 / te, 2), "dec_fps": round(1.0 / td, 2),
            "enc_frac_of_157_TFLOPs": round(590.4e9 * scale / te / 157.3e12, 4),
            "dec_frac_of_157_TFLOPs": round(691.6e9 * scale / td / 157.3e12, 4),
            "i_frame_enc_ms": round(1e3 * t_enc[0], 2), "i_frame_dec_ms": round(1e3 * t_dec[0], 2),
            "note": "fp32 storage, v_mfma_f32_16x16x4_f32, explicit fma order: tensors, symbols and streams bit-exact with the "
                    "CPU oracle (tests/test_gpu_codec.py); 5 steady P frames, sequential, each frame synchronised"}


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process per GPU, the way
    the reference harness starts its own workers (test_video.py:381-395,439-442: a spawned pool, worker n on GPU n) -
    here as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, a CHILD process
    (never an exec: this parent has imported torch but made no GPU call, and it stays that way).  The ranks' stdout /
    stderr are ours; returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                 # (the launcher would set and announce it otherwise)
    return subprocess.call(cmd, env=env)


def stub_main(args, world, rank, local):
    """DCVC_BENCH_STUB=1 - a control-flow rehearsal for the CPU tests, NOT a measurement and never a fallback: the same
    launcher, rank checks, CPU pinning, weight broadcast, measure() and rank-0 line as the real run, on the gloo backend
    with a stub in place of the codec (rank r sleeps (r + 1) ms per frame).  The line says so in `data` and `metric`."""
    import torch.distributed as dist
    cpus = dist_utils.pin_rank_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else None
    device = torch.device("cpu")
    if world > 1:
        dist.init_process_group("gloo")
    sd = weights.make_state_dict("dmc", 1234) if rank == 0 else None
    got = dist_utils.broadcast_state_dict("dmc", sd, device, rank, world)
    digest = float(sum(float(np.asarray(v, np.float64).sum()) for v in got.values()))
    state = {"n": 0}

    def run(n, timed):
        state["n"] += n
        time.sleep(0.001 * (rank + 1) * n)

    elapsed, align, mine = measure(run, args.steps, args.warmup, world, device, lambda: None)
    fps = dist_utils.gather_over_ranks(args.steps / mine, device, world)
    digests = dist_utils.gather_over_ranks(digest, device, world)
    if rank == 0:
        print(json.dumps({
            "metric": "STUB control-flow rehearsal (gloo, sleeping stub codec) - not a measurement",
            "value": round(world * args.steps / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
            "rank_fps": {"min": round(min(fps), 3), "max": round(max(fps), 3)},
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "weights_identical_on_all_ranks": len(set(digests)) == 1,
            "config": {"workload": "stub", "alignment_frames": align, "frames_run": state["n"],
                       "cpus_per_rank": len(cpus) if cpus else len(os.sched_getaffinity(0)),
                       "cpus_granted": dist_utils.cpus_granted(int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)     # eight whole GOPs (I frames at the GOP's rate, 1 in 32), under a second
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-mode", action="store_true", help="skip the fp32 exact-mode leg")
    ap.add_argument("--frame", default="1920x1080",
                    help="WxH of the synthetic sequence (default: the BASELINE.json configs[1] size; 3840x2160 = configs[3])")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (what profiles/r01_roofline_kernel_stats.csv was collected on)")
    args = ap.parse_args()

    global HEIGHT, WIDTH
    WIDTH, HEIGHT = (int(v) for v in args.frame.lower().split("x"))
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the N ranks (child processes) and hand their exit code on
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): one rank per GPU, --gpus N must equal "
                         "WORLD_SIZE" % (args.gpus, world))
    if os.environ.get("DCVC_BENCH_STUB") == "1":
        return stub_main(args, world, rank, local)
    # one process per GPU: keep this rank's threads (2 pipeline threads + the rANS workers) on its share of the cores,
    # those of its GPU's NUMA node when sysfs tells - before anything touches the GPU
    cpus = dist_utils.pin_rank_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else None
    n_dev = _lib.require_gpu()
    # DCVC_BENCH_REHEARSE=1 (developer rehearsal of the N > 1 path on a ONE-GPU box, labelled on the line, never a
    # measurement of configs[4]): the ranks share the GPUs there are (rank r on GPU r % count) and talk over gloo - RCCL
    # refuses two ranks on one device.  Everything else is the real run: the real codecs, pinning, measure(), the line.
    rehearse = world > 1 and os.environ.get("DCVC_BENCH_REHEARSE") == "1"
    if world > 1 and local >= n_dev and not rehearse:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible): one rank per GPU" % (local, n_dev))
    gpu = local % n_dev
    torch.cuda.set_device(gpu)
    device = torch.device("cuda", gpu)
    coll_device = torch.device("cpu") if rehearse else device      # where the collectives' tensors live
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    dtype = torch.float16
    torch.set_grad_enabled(False)
    # no intra-op thread pool (the reference harness does the same, src/utils/common.py:23): its idle workers compete with
    # the two pipeline threads and the rANS workers for the granted cores - three runs each on one box, --steps 20:
    # 377 / 331 / 351 frames/s with the pool, 377 / 380 / 380 without (profiles/r04_host_threads_ab.txt)
    torch.set_num_threads(1)

    (ie, pe), (idec, pdec) = load_models(dtype, device, world, rank, coll_device)
    two = use_two_entropy_coders(HEIGHT, WIDTH)
    for m in (ie, pe, idec, pdec):
        m.set_use_two_entropy_coders(two)
    if args.roofline_only:
        pe._ensure_layers()
        print(json.dumps({"roofline": roofline_leg(pe, device, dtype)}), flush=True)
        return
    planes, frames = make_frames(rank, dtype, device)
    enc = SequenceEncoder(ie, pe, QP, intra_period=GOP, reset_interval=GOP)
    # decoder output deferred by one frame: the reconstruction network of frame n runs in the host-decoding gaps
    # of frame n+1 (opendcvc_amd/pipeline.py); every timed frame is still completed inside the timed region (flush)
    dec = SequenceDecoder(idec, pdec, HEIGHT, WIDTH, two, defer_output=True)

    state = {"i": 0, "t_enc": 0.0, "t_dec": 0.0, "bytes": 0, "n_i": 0, "j": 0, "seq_bytes": 0, "psnr": [], "done": [], "kinds": []}
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
                state["kinds"].append(bool(pkt.is_i))

        def on_frame(_):
            state["j"] += 1
            if timed:
                state["done"].append(time.perf_counter())      # (host time of the frame's hand-over by the decoder stage)

        pipe.run((frames[k % GOP] for k in range(first, first + nsteps)), on_packet, on_frame)

    # Placement of the timed window in the GOP (the encoder codes an I frame whenever its frame counter is a multiple
    # of 32): whole GOPs start on an I frame; a shorter window gets one I frame in its middle - never a P-only window.
    K = args.steps
    elapsed, align, mine = measure(run_pipelined, K, args.warmup, world, coll_device, torch.cuda.synchronize)
    rank_fps = dist_utils.gather_over_ranks(K / mine, coll_device, world)         # every rank's own frames/s (rank 0 reports)
    assert state["i"] % GOP == (window_start(K) + K) % GOP
    assert state["j"] == args.warmup + align + K
    assert state["n_i"] >= max(1, K // GOP), "the timed window must contain I frames at the GOP's rate"

    # outside the timed region: ONE WHOLE GOP (I frame first) one direction at a time, like the reference times them;
    # bpp and PSNR of that GOP
    from opendcvc_amd.harness import yuv420_distortion
    run_pipelined((-state["i"]) % GOP, False)
    assert state["i"] % GOP == 0
    n_seq = GOP
    # encoder pass, then decoder pass (the reference codes a whole sequence before it decodes it, test_video.py:172-298).
    # Both run with their deferred forms: a P frame's entropy coding / reconstruction network is finished underneath
    # the next frame's kernels, so packets and pictures come out one call late; every frame is complete at the end.
    enc.defer = True
    pkts, pending = [], []
    t_seq0 = time.perf_counter()
    for k in range(n_seq):
        state["i"] += 1
        ta = time.perf_counter()
        with torch.cuda.stream(s_enc):
            pkts += enc.encode(frames[k])
        torch.cuda.synchronize()
        state["t_enc"] += time.perf_counter() - ta
    ta = time.perf_counter()
    pkts += enc.flush()
    state["t_enc"] += time.perf_counter() - ta
    enc.defer = False
    for pkt in pkts:
        tb = time.perf_counter()
        with torch.cuda.stream(s_dec):
            pending += dec.decode(pkt)
        torch.cuda.synchronize()
        state["t_dec"] += time.perf_counter() - tb
        state["seq_bytes"] += len(pkt.bit_stream)
    tb = time.perf_counter()
    with torch.cuda.stream(s_dec):
        pending += dec.flush()
    torch.cuda.synchronize()
    state["t_dec"] += time.perf_counter() - tb
    t_seq = time.perf_counter() - t_seq0
    assert len(pkts) == n_seq
    assert len(pending) == n_seq
    for k, x_hat in enumerate(pending):             # (not timed) the reference's per-plane PSNR on the 8-bit planes
        state["psnr"].append(yuv420_distortion(x_hat, *planes[k]))
    psnr = np.mean(np.asarray(state["psnr"], np.float64), axis=0)

    # A figure that does not depend on where a short window sits in the GOP: the time of a P frame = the median interval
    # between two completed frames of the timed window, the time of an I frame = what is left of the window per I frame;
    # one GOP = 1 I + 31 P.  (With whole GOPs in the window - the default - this is `value`.)
    gaps = np.diff(np.asarray(state["done"], np.float64)) if len(state["done"]) > 2 else np.asarray([elapsed / K])
    t_p = float(np.median(gaps))
    t_i = max(t_p, (elapsed - (K - state["n_i"]) * t_p) / max(1, state["n_i"]))
    gop_weighted = GOP / (t_i + (GOP - 1) * t_p)

    if rank == 0:
        N = world
        value = N * K / elapsed
        seq_value = n_seq / t_seq                   # this rank, encode then decode one after the other
        if (WIDTH, HEIGHT) == (3840, 2160):       # README complexity table, A100 fp16 at 4K: 35.5 / 29.5 fps
            base = 1.0 / (1.0 / 35.5 + 1.0 / 29.5)
        else:
            base = 1.0 / (1.0 / BASELINE_ENC_FPS + 1.0 / BASELINE_DEC_FPS)
        out = {
            "metric": "%s%s YUV420 encode+decode FPS (frames/s through encode AND decode, whole job)"
                      % ("REHEARSAL (%d ranks sharing %d GPU(s), gloo) - not a measurement of the multi-GPU configuration: "
                         % (world, n_dev) if rehearse else "", "1080p" if (WIDTH, HEIGHT) == (1920, 1080) else args.frame),
            "value": round(value, 3), "unit": "frames/s", "n_gpus": N, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(seq_value / base, 4), "dtype": "f16", "data": "synthetic",
            "rank_fps": {"min": round(min(rank_fps), 3), "max": round(max(rank_fps), 3),
                         "note": "each rank's own K timed frames / its own time (before the closing barrier)"},
            "rccl_ranks": (torch.distributed.get_world_size() if WEIGHTS_VIA["how"].startswith("rccl") else 0) if world > 1 else 1,
            "gop_weighted_value": round(N * gop_weighted, 3),
            "gop_weighted_note": "frames/s of one 32-frame GOP = 32 / (t_I + 31 t_P), t_P = median interval between completed "
                                 "frames of the timed window (%.3f ms), t_I = the rest of the window per I frame (%.3f ms): "
                                 "independent of --steps" % (1e3 * t_p, 1e3 * t_i),
            "config": {"workload": "DCVC-RT inter-coding, %s YUV420 32-frame GOP, single q (qp 32), one stream per MI355X (%s)"
                                   % (("1080p", "BASELINE.json configs[1]") if (WIDTH, HEIGHT) == (1920, 1080)
                                      else (args.frame, "BASELINE.json configs[3]") if (WIDTH, HEIGHT) == (3840, 2160)
                                      else (args.frame, "not a BASELINE.json configuration")),
                       "frame": "%dx%d padded to %dx%d" % (WIDTH, HEIGHT, WIDTH + (-WIDTH) % 16, HEIGHT + (-HEIGHT) % 16), "intra_period": GOP, "i_frames_timed": state["n_i"],
                       "i_frame_share_timed": round(state["n_i"] / float(K), 4), "i_frame_share_gop": round(1.0 / GOP, 4),
                       "alignment_frames": align, "cpus_per_rank": len(cpus) if cpus else len(os.sched_getaffinity(0)),
                       "cpus_granted": dist_utils.cpus_granted(int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))),
                       "entropy_coders": 2 if two else 1, "force_zero_thres": THRES,
                       "weights": "synthetic seed 1234 (opendcvc_amd/weights.py)", "weights_via": WEIGHTS_VIA["how"],
                       "pipeline": "encoder and decoder on two host threads / two HIP streams of the same GPU: frame n "
                                   "decodes while frame n+1 encodes; every timed frame is encoded and decoded; the "
                                   "decoder emits P pictures one call late (their reconstruction network fills the next "
                                   "frame's host-decoding gaps)",
                       "baseline_note": "vs_baseline = sequential_fps_per_gpu / (1/(1/125.2+1/112.8)) fps (reference README, "
                                        "A100 fp16, encode and decode timed one after the other as in the reference); the "
                                        "sequential pass is one whole GOP, encoder loop then decoder loop, each frame "
                                        "synchronised, P-frame streams / pictures completed one call late"},
            "sequential_fps_per_gpu": round(seq_value, 3), "sequential_ms_per_step": round(1e3 * t_seq / n_seq, 3),
            "enc_fps_per_gpu": round(n_seq / state["t_enc"], 2), "dec_fps_per_gpu": round(n_seq / state["t_dec"], 2),
            "bpp": round(state["bytes"] * 8.0 / (K * HEIGHT * WIDTH), 5),
            "gop_bpp": round(state["seq_bytes"] * 8.0 / (n_seq * HEIGHT * WIDTH), 5),
            "psnr": {"weighted_6y_u_v": round(float(psnr[0]), 4), "y": round(float(psnr[1]), 4), "u": round(float(psnr[2]), 4),
                     "v": round(float(psnr[3]), 4),
                     "note": "mean over one GOP, decoded vs the 8-bit source planes (reference get_distortion, "
                             "test_video.py:94-111); synthetic untrained weights: single-digit dB by construction - the "
                             "number pins parity with the reference (tests/golden), not picture quality"},
        }
        # the same GOP coded by the REFERENCE in its fp16 arithmetic on a CPU (tests/golden/make_golden_bench_gop.py): how far the
        # line's bpp / PSNR are from it (data file only; nothing of the reference runs here)
        ref_path = os.path.join(REPO, "tests", "golden", "bench_gop_f16.json")
        if (WIDTH, HEIGHT) == (1920, 1080) and os.path.exists(ref_path):
            try:
                ref = json.load(open(ref_path))
                out["reference_gop"] = {
                    "bpp": round(ref["gop_bpp"], 5), "psnr_weighted_6y_u_v": round(ref["psnr_mean"][0], 4),
                    "bpp_rel_dev": round(out["gop_bpp"] / ref["gop_bpp"] - 1.0, 5),
                    "psnr_dev_db": round(out["psnr"]["weighted_6y_u_v"] - ref["psnr_mean"][0], 5),
                    "note": "the reference's DMCI / DMC in .half() on a CPU on the same 32 frames, qp and weights "
                            "(tests/golden/bench_gop_f16.json): gop_bpp and psnr of this line against it"}
            except Exception:      # noqa: BLE001  (a missing / unreadable fixture must not cost the measurement)
                pass
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
        if world == 1 and not args.no_exact_mode:
            out["exact_mode"] = exact_mode_leg(device, world, rank, frames)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg()
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
