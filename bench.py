#!/usr/bin/env python3
"""Headline benchmark: training images/sec, FCRN ResNet-50 (UpProj) 640x480 bf16 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = forward + SILog + backward + gradient all-reduce (N > 1) + Adam on one batch of
synthetic 32 x 3 x 480 x 640 images per GPU (weak scaling), inputs resident in HBM.
Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel
(the MFMA implicit-GEMM convolution, timed per launch with HIP events on its own stream
during the timed steps) and the CPU-oracle baseline timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, BATCH = 480, 640, 32
PEAK_BF16_TFLOPS = 2500.0                 # dense MFMA bf16 (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0                        # HBM3E (MI355X_MICROARCH.md)
ACT_NAME = os.environ.get("MDE_ACT_DTYPE", "bf16").lower()          # the library build: bf16 (default) or fp16 storage
ACT_NAME = "fp16" if ACT_NAME in ("fp16", "float16", "half", "f16") else "bf16"
LOSS_SCALE = float(os.environ.get("MDE_LOSS_SCALE", "4096" if ACT_NAME == "fp16" else "1"))
ALGO_GFLOP_PER_IMAGE = 410.9              # useful conv MACs x 2 x 3 (fwd+dgrad+wgrad)
ALGO_CONV_BYTES_PER_CALL = 153.8e6        # 22.0 GB per step (in + out + weights of the 143 forward / input-gradient calls, bf16) / 143:
#                                           the convolutions alone.  Since round 3 fifty-odd of those launches also carry a BatchNorm
#                                           site's backward sums and an identity shortcut's gradient: what THEY are asked to read is
#                                           counted per launch (ops.conv_gemm's nbytes) and reported as algorithmic_bytes


def synthetic(n, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rgb = torch.rand(n, 3, H, W, generator=g, device=device)
    depth = 0.05 + 0.95 * torch.rand(n, 1, H, W, generator=g, device=device)
    hole = torch.rand(n, 1, H, W, generator=g, device=device) < 0.10
    return rgb, depth.masked_fill(hole, 0.0)


TRAFFIC_FILE = os.path.join("profiles", "r04_hbm_traffic.json")
TRAFFIC_SOURCES = ("conv_gemm.hip", "conv_wgrad.hip", "mde_common.h")


def kernel_source_hash():
    """sha256 over the sources of the two GEMM kernels: the traffic figures in profiles/ are only quoted for the
    kernels they were collected on (tools/hbm_traffic.py stores this hash next to them)."""
    import hashlib
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, "mono_depth_estimation_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel):
    """(HBM bytes per launch of `kernel`, provenance note) from the rocprofv3 PMC passes committed under profiles/
    (counters cannot be read from inside the process; collection + the gfx950 FETCH_SIZE x2 correction are
    documented in that file).  (None, why) when the file is absent or was collected on other kernel sources."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            j = json.load(f)
        if j.get("_source_hash") != kernel_source_hash():
            return None, "%s was collected on kernel sources %s, the build is %s: stale, not quoted" % (
                TRAFFIC_FILE, j.get("_source_hash"), kernel_source_hash())
        return round(j[kernel]["hbm_bytes_per_launch"]), "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, sources %s)" % (
            TRAFFIC_FILE, j["_source_hash"])
    except (OSError, KeyError, ValueError) as e:
        return None, "no traffic file (%s)" % type(e).__name__


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 CPUs but grants a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(threads):
    """The CPU oracle (a port of the reference's fp32 torch.nn path, pinned to the reference by
    tests/golden) on the host cores, as SURVEY.md section 8d specifies: FCRN-50 480x640, batch 4,
    1 warm-up + 3 timed steps of forward + SILog + backward + Adam, median images/sec."""
    from oracle import fcrn as ofcrn
    from oracle import losses as OL
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = ofcrn.FCRNOracle(50, (H, W), out_channels=1)
    net.conv3.weight.data.mul_(0.05)
    opt = torch.optim.Adam([{"params": net.get_1x_lr_params(), "lr": 1e-4},
                            {"params": net.get_10x_lr_params(), "lr": 1e-3}], lr=1e-4)
    n = 4
    rgb, tgt = synthetic(n, 1234, "cpu")
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = OL.silog(net(rgb), tgt)
        loss.backward()
        opt.step()
        if i:
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(n / med, 4), "unit": "images/sec", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": "oracle FCRN-50 fp32 480x640, batch %d, 1 warm-up + %d timed train steps (fwd+SILog+bwd+Adam), median"
                      % (n, len(times))}


# BASELINE.json configurations 3 / 4 / 5 at their per-GPU sizes.  Forward MACs per image are SURVEY.md 8a / 8d's dense-as-written
# figures (x 2 x 3 = training FLOP); the reference's own module configuration supplies loss and optimiser.
OTHER_CONFIGS = {
    "bts": {"batch": 16, "hw": (480, 640), "gmac": 121.48,
            "workload": "BTS DenseNet-161 + LPG / atrous decoder (reference network/Bts.py, bts_size 512, max_depth 1), %dx3x480x640 per "
                        "GPU -> 5 maps, train step = fwd + SILog(0.85) on the final depth + bwd + AdamW(eps 1e-3, wd 1e-2 / 0)"},
    "midas": {"batch": 32, "hw": (384, 384), "gmac": 103.47,
              "workload": "MiDaS ResNeXt-101 32x8d (reference network/MiDaS.py, features 256), %dx3x384x384 per GPU -> 7-channel sigmoid, "
                          "train step = fwd + MidasLoss(0.5, ssimse) on channel 0 + bwd + Adam(0.1 lr encoder, lr decoder)"},
    "vnl": {"batch": 16, "hw": (480, 640), "gmac": 348.42,
            "workload": "VNL ResNeXt-50 32x4d stride 16, 150 bins (reference network/VNL.py), %dx3x480x640 per GPU -> logits + softmax, "
                        "train step = fwd + ModelLoss (WCEL + 6 x virtual-normal loss) + bwd + SGD(momentum 0.9, wd 5e-4)"},
}


def build_other(name, n, dev):
    """-> (net, step(): one training step through the drop-in nn.Module path).  Mirrors tools/config_bench.py."""
    import numpy as np
    from mono_depth_estimation_amd import criteria
    h, w = OTHER_CONFIGS[name]["hw"]
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + int(os.environ.get("RANK", "0")))
    x = torch.rand(n, 3, h, w, generator=g, device=dev)
    gt = 0.05 + 0.95 * torch.rand(n, 1, h, w, generator=g, device=dev)
    gt = gt.masked_fill(torch.rand(n, 1, h, w, generator=g, device=dev) < 0.10, 0.0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if name == "bts":
        from mono_depth_estimation_amd.network import Bts
        net = Bts.BtsModel(max_depth=1.0, bts_size=512, encoder_version="densenet161_bts", out_channels=1).to(dev).train()
        crit = criteria.silog_loss(0.85)
        fwd_loss = lambda: crit(net(x)[4], gt)
        opt = lambda: net._store.adam_step(1e-4, 1e-4, eps=1e-3, weight_decay=(1e-2, 0.0), decoupled=True, grad_scale=1.0 / (world * LOSS_SCALE))
    elif name == "midas":
        from mono_depth_estimation_amd.network import MiDaS
        net = MiDaS.MidasNet(features=256).to(dev).train()
        crit = criteria.MidasLoss(alpha=0.5, loss="ssimse")
        fwd_loss = lambda: crit(net(x)[:, :1], gt)
        opt = lambda: net._store.adam_step(1e-5, 1e-4, grad_scale=1.0 / (world * LOSS_SCALE))
    else:
        from types import SimpleNamespace
        from mono_depth_estimation_amd.network import VNL
        C, dmin, dmax = 150, 0.01, 1.1
        interval = (np.log10(dmax) - np.log10(dmin)) / C
        p = SimpleNamespace(depth_min=dmin, encoder="resnext50_32x4d_body_stride16", pretrained=0, freeze_backbone=False, init_type="xavier",
                            enc_dim_in=[64, 256, 512, 1024, 2048], enc_dim_out=[512, 256, 256, 256], dec_dim_in=[512, 256, 256, 256, 256, 256],
                            dec_dim_out=[256, 256, 256, 256, 256], dec_out_c=C, focal_x=519.0, focal_y=519.0, crop_size=(h, w), diff_loss_weight=6,
                            depth_min_log=np.log10(dmin), depth_bin_interval=interval,
                            wce_loss_weight=[[np.exp(-0.2 * (i - j) ** 2) for i in range(C)] for j in np.arange(C)],
                            depth_bin_border=np.array([np.log10(dmin) + interval * (i + 0.5) for i in range(C)]))
        net = VNL.MetricDepthModel(p).to(dev).train()
        crit = criteria.ModelLoss(p)
        bins = criteria.depth_to_bins(gt, dmin, dmax, C)

        def fwd_loss():
            logit, prob = net(x)
            return crit(criteria.bins_to_depth(prob, p.depth_bin_border), logit, bins, gt)
        opt = lambda: net._store.sgd_step(1e-4, 1e-5, momentum=0.9, weight_decay=5e-4, grad_scale=1.0 / (world * LOSS_SCALE))
    return net, fwd_loss, opt


def run_other_config(args, dev, rank, world, use_dist, t_start):
    """The same contract (warm-up, K timed steps between barriers, max over ranks, one JSON line with a roofline leg) for
    BASELINE.json configurations 3 / 4 / 5.  The gradient exchange (N > 1) is overlapped with backward as in the FCRN path: the
    module's backward feeds the reducer (TapeModule.set_grad_reducer -> TapeEngine.backward(on_progress)), a bucket's all-reduce
    goes out on the exchange stream as soon as the tape has passed the bucket's first element."""
    import torch.distributed as dist
    from mono_depth_estimation_amd import dp, ops
    cfg = OTHER_CONFIGS[args.config]
    # the contract is ONE line on stdout: whatever a module prints while it is built (MidasNet reports the weight file it
    # loads, as the reference's does: MiDaS.py) goes to stderr
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        net, fwd_loss, opt = build_other(args.config, args.batch, dev)
    state = {}

    def step():
        net.zero_grad(set_to_none=True)
        loss = fwd_loss()
        (loss * LOSS_SCALE if LOSS_SCALE != 1.0 else loss).backward()      # (fp16 build: static loss scale, divided out by the optimiser step)
        if use_dist:
            state["red"].finish()
        opt()
        state["loss"] = loss

    if use_dist:
        store = net._store
        dist.broadcast(store.P, 0)
        dist.broadcast(store.B, 0)
        # (no extra_streams: the plans -- and their weight-gradient streams -- do not exist before the first forward; the engine
        #  joins that stream into the main one before each of the handful of bucket callbacks instead)
        state["red"] = dp.FlatGradReducer(store.G, store.layer_boundaries(), target_bytes=int(os.environ.get("MDE_DP_BUCKET_MB", "64")) << 20,
                                          wire_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None)
        net.set_grad_reducer(state["red"])

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    fence()
    timer = None if args.no_launch_timing else ops.LaunchTimer()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timer is not None and i == args.steps - 1:
            ops.TIMER = timer            # per-launch HIP events in ONE step (they serialise neighbouring kernels)
        step()
    fence()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    drift = None
    if use_dist:       # replicas must hold bit-identical weights after the timed steps (see the FCRN path below)
        P = net._store.P
        cs = torch.stack([P.double().sum(), P.double().square().sum()])
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        drift = float((hi - lo).abs().max())
    log("timed region done: %.1f ms/step" % (1e3 * dt / args.steps))
    if rank == 0:
        ips = args.batch * world * args.steps / dt
        gflop = cfg["gmac"] * 6.0
        out = {
            "metric": "training images/sec, %s, %s" % (args.config.upper(), ACT_NAME), "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": ACT_NAME, "data": "synthetic",
            "config": {"workload": cfg["workload"] % args.batch + (" + flat-gradient all-reduce overlapped with backward (RCCL, %s buckets)" % args.grad_dtype
                                                                   if use_dist else ""),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": "dp%d" % world,
                       "final_loss": round(float(state["loss"]), 5), "algorithmic_gflop_per_image": round(gflop, 1),
                       **({"replica_drift": drift, "exchange_buckets": len(state["red"].buckets),
                           "buckets_issued_before_backward_ended": state["red"].early} if use_dist else {})},
            "step_mfma_frac": round(ips * gflop / 1e3 / (world * PEAK_BF16_TFLOPS), 4),
        }
        if timer is not None:
            if args.per_shape:
                print_per_shape(timer, 1)
            summ = timer.summary()
            n, fl, sec = summ.get("conv_gemm_nt", (0, 0.0, 1.0))
            ach = fl / sec / 1e12 if n else 0.0
            out["roofline"] = {"bound": "mfma", "kernel": "conv_gemm_nt (implicit-GEMM conv fwd / input gradient, bf16 MFMA)",
                               "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                               "traffic": None, "launches": n, "avg_launch_us": round(1e6 * sec / max(n, 1), 2),
                               "avg_launch_gflop": round(fl / max(n, 1) / 1e9, 3),
                               "share_of_step_time": round(sec / (dt / args.steps), 4), "timed_launch_steps": 1}
            if "conv_wgrad_tn" in summ:
                n2, fl2, sec2 = summ["conv_wgrad_tn"]
                out["roofline"]["wgrad_kernel"] = {"achieved": round(fl2 / sec2 / 1e12, 2), "launches": n2,
                                                   "avg_launch_us": round(1e6 * sec2 / n2, 2),
                                                   "share_of_step_time": round(sec2 / (dt / args.steps), 4)}
            add_binding_roofline(out["roofline"], timer)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def add_binding_roofline(roof, timer):
    """`frac` above prices every launch of the kernel against the MFMA peak.  Many of them cannot get there: a 1x1 convolution
    over 64 or 128 channels moves its tensors once and is done (64 -> 256 channels: 51 FLOP per byte; the chip's ridge is
    2500 / 8 = 312).  `vs_binding_roofline` prices each launch against whichever of the two rooflines binds IT -- max(FLOP /
    2.5 PFLOP/s, algorithmic bytes / 8 TB/s) -- and reports the sum of those floors over the measured time, with the count of
    launches whose floor is the HBM one."""
    bind = timer.binding(PEAK_BF16_TFLOPS * 1e12, PEAK_HBM_TBS * 1e12)
    for kind, key in (("conv_gemm_nt", None), ("conv_wgrad_tn", "wgrad_kernel")):
        if kind not in bind:
            continue
        lo, nh, t = bind[kind]
        d = roof if key is None else roof.get(key)
        if d is not None:
            d["vs_binding_roofline"] = {"frac": round(lo / t, 4), "hbm_bound_launches": nh,
                                        "peaks": "%.0f TFLOP/s bf16 dense, %.0f TB/s HBM" % (PEAK_BF16_TFLOPS, PEAK_HBM_TBS)}


def print_per_shape(timer, timed_launch_steps):
    """One line per (kernel, shape, algorithmic bytes): launches per step, time per launch, TFLOP/s, TB/s of the algorithmic
    bytes (operands, results and whatever the fused epilogue reads, once each), and the time above a PRACTICAL floor --
    max(flops / 1100 TFLOP/s, bytes / 5.5 TB/s), the rates the best launches of this step reach -- per step."""
    tab = {}
    for (kind, flops, e0, e1, tag), nbytes in zip(timer.records, timer.bytes):
        n, t = tab.get((kind, tag, flops, nbytes), (0, 0.0))
        tab[(kind, tag, flops, nbytes)] = (n + 1, t + e0.elapsed_time(e1) * 1e-3)
    rows = []
    for (kind, tag, flops, nbytes), (n, t) in tab.items():
        floor = max(flops / 1.1e15, nbytes / 5.5e12)
        rows.append((n * max(0.0, t / n - floor), kind, tag, flops, nbytes, n, t, floor))
    for excess, kind, tag, flops, nbytes, n, t, floor in sorted(rows, key=lambda r: -r[0]):
        print("%-14s %-46s x%-3d %8.1f us  %7.1f TF/s  %5.2f TB/s  floor %6.1f us (%s)  %5.2f ms/step  above floor %5.2f ms/step" % (
            kind, tag, n // timed_launch_steps, 1e6 * t / n, flops / (t / n) / 1e12, nbytes / (t / n) / 1e12, 1e6 * floor,
            "hbm" if nbytes / 5.5e12 > flops / 1.1e15 else "mfma", 1e3 * t / timed_launch_steps, 1e3 * excess / timed_launch_steps), file=sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak scaling) / global batch (strong scaling); "
                                                            "default: the configuration's own (fcrn 32, bts 16, midas 32, vnl 16)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch images on EVERY GPU; strong: --batch images split over the GPUs (SURVEY 8e)")
    ap.add_argument("--grad-dtype", choices=("bf16", "fp32"), default=os.environ.get("MDE_DP_GRAD_DTYPE", "fp32"),
                    help="wire format of the gradient all-reduce buckets (N > 1).  fp32 is what the reference's DDP sums in and the "
                         "default; bf16 halves the bytes on xGMI, is opt-in and is named in config.grad_wire_dtype")
    ap.add_argument("--config", choices=("fcrn", "bts", "midas", "vnl"), default="fcrn",
                    help="BASELINE.json configuration: fcrn = the headline (configuration 2, the default); bts / midas / vnl = "
                         "configurations 3 / 4 / 5 at their per-GPU sizes (16x480x640, 32x384x384, 16x480x640), module path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--per-shape", action="store_true", help="print a per-shape table of the GEMM launches to stderr")
    args = ap.parse_args()

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.batch is None:
        args.batch = OTHER_CONFIGS[args.config]["batch"] if args.config != "fcrn" else BATCH
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if args.scaling == "strong":
        if args.batch % world:
            sys.exit("bench.py --scaling strong: global batch %d is not divisible by %d GPUs" % (args.batch, world))
        args.batch //= world                                  # from here on: images per GPU
    # MDE_BENCH_DEVICE / MDE_DIST_BACKEND: rehearsal of the N > 1 path on a one-GPU box (every rank on device 0, gloo instead of
    # RCCL, which refuses two ranks on one device): tests/test_bench_multirank_gpu.py.  The driver's runs set neither.
    dev_index = int(os.environ.get("MDE_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with 1 rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("MDE_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mono_depth_estimation_amd import dp, ops
    from mono_depth_estimation_amd.network import FCRN

    torch.manual_seed(0)                                  # identical initial weights on every rank
    if args.config != "fcrn":
        return run_other_config(args, dev, rank, world, use_dist, t_start)
    net = FCRN.ResNet(layers=50, decoder="upproj", output_size=(H, W), out_channels=1, pretrained=False)
    net.conv3.weight.data.mul_(0.05)                      # keep the sigmoid unsaturated at init (see tests/golden)
    net = net.to(dev).train()
    x, tgt = synthetic(args.batch, 1234 + rank, dev)
    eng = net._engine(x)
    store = net._store
    if use_dist:
        dist.broadcast(store.P, 0)
        dist.broadcast(store.B, 0)
    reducer = dp.FlatGradReducer(store.G, eng.grad_boundaries(), target_bytes=int(os.environ.get("MDE_DP_BUCKET_MB", "64")) << 20,
                                 extra_streams=[eng.side], wire_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None)
    ws, loss = ops.silog_ws(dev), torch.empty(1, device=dev)
    dy = torch.empty(args.batch, 1, H, W, device=dev)
    lr = 1e-4
    # the fp16 storage build (MDE_ACT_DTYPE=fp16) trains with a static loss scale, as the reference's precision=16 run does with
    # torch's GradScaler: the loss gradient goes into the backward pass multiplied by it, the optimiser step divides it out
    S = LOSS_SCALE
    gscale = torch.full((1,), S, device=dev) if S != 1.0 else None

    def step():
        y = eng.forward(x, True)
        ops.silog_fwd(y, tgt, 0.85, ws, loss)
        ops.silog_bwd(y, tgt, 0.85, ws, gscale, dy)
        store.G.zero_()
        eng.backward(dy, reducer.ready, consumer_waits_side=True)
        reducer.finish()
        store.adam_step(lr, 10 * lr, grad_scale=1.0 / (world * S))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    log("plan built, starting warm-up")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    fence()
    # Per-launch HIP events (the roofline leg) are recorded during ONE step of the timed region, the
    # last one: an event pair around every GEMM launch serialises neighbouring kernels and costs
    # ~1.3 ms per instrumented step (measured: 34.4 vs 33.0 ms/step with every step instrumented).
    # In that step the weight-gradient GEMMs also stay on the main stream (engine.EngineCore.wgrad), so
    # every duration describes ONE kernel; in the other steps they overlap the BatchNorm chain from a
    # second stream (the rocprofv3 summary that must agree with these events is taken with
    # MDE_WGRAD_STREAM=0: profiles/).
    timer = None if args.no_launch_timing else ops.LaunchTimer()
    timed_launch_steps = 0 if timer is None else 1
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - timed_launch_steps:
            ops.TIMER = timer
        step()
    fence()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    final_loss = float(loss)
    # replicas must hold bit-identical weights after the timed steps (every rank applies the same reduced gradient): the
    # spread of a checksum of the flat parameter buffer over the ranks, 0.0 when the exchange works
    drift = None
    if use_dist:
        cs = torch.stack([store.P.double().sum(), store.P.double().square().sum()])
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        drift = float((hi - lo).abs().max())
    log("timed region done: %.1f ms/step" % (1e3 * dt / args.steps))

    if rank == 0:
        ips = args.batch * world * args.steps / dt
        out = {
            "metric": "training images/sec, FCRN 640x480 %s" % ACT_NAME, "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": ACT_NAME, "data": "synthetic",
            "config": {"workload": "FCRN ResNet-50 + UpProj (reference network/FCRN.py), %dx3x480x640 per GPU -> 1x480x640 "
                                   "depth, train step = fwd + SILog(0.85) + bwd + Adam(lr, 10*lr)%s" % (
                                       args.batch, " + flat-gradient all-reduce (RCCL, %s buckets)" % args.grad_dtype if world > 1 else ""),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": "dp%d" % world,
                       "final_loss": round(final_loss, 5), **({"replica_drift": drift} if drift is not None else {})},
            "step_mfma_frac": round(ips * ALGO_GFLOP_PER_IMAGE / 1e3 / (world * PEAK_BF16_TFLOPS), 4),
        }
        if timer is not None and args.per_shape:
            print_per_shape(timer, timed_launch_steps)
        if timer is not None:
            summ = timer.summary()
            n, fl, sec = summ.get("conv_gemm_nt", (0, 0.0, 1.0))
            ach = fl / sec / 1e12 if n else 0.0
            traffic, traffic_src = measured_traffic("conv_gemm_nt")
            # what an average call is asked to move: its input, output and weights once each (bf16) + the tensors its fused
            # epilogue reads (ops.conv_gemm's nbytes, summed over the timed launches)
            conv_bytes = [b for (k, *_), b in zip(timer.records, timer.bytes) if k == "conv_gemm_nt"]
            algo_bytes = sum(conv_bytes) / max(len(conv_bytes), 1) if conv_bytes else ALGO_CONV_BYTES_PER_CALL
            out["roofline"] = {
                "bound": "mfma", "kernel": "conv_gemm_nt (implicit-GEMM conv fwd/dgrad/up-projection, bf16 MFMA)",
                "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                # algorithmic HBM bytes of an average call (its input + output + weights, bf16, once each: DESIGN.md section 3)
                # and the measured traffic over it -- the waste ratio the reviews track
                "algorithmic_bytes": round(algo_bytes),
                "algorithmic_bytes_convolutions_alone": ALGO_CONV_BYTES_PER_CALL,
                "traffic_over_algorithmic": round(traffic / algo_bytes, 3) if traffic else None,
                "launches": n, "avg_launch_us": round(1e6 * sec / max(n, 1), 2),
                "avg_launch_gflop": round(fl / max(n, 1) / 1e9, 3),
                "share_of_step_time": round(sec / timed_launch_steps / (dt / args.steps), 4),
                "timed_launch_steps": timed_launch_steps,
                "timed_launch_mode": "one stream (each duration is one kernel alone); the other steps run the weight-gradient GEMMs "
                                     "on a second stream" if eng.side is not None else "one stream",
            }
            if "conv_wgrad_tn" in summ:
                n2, fl2, sec2 = summ["conv_wgrad_tn"]
                out["roofline"]["wgrad_kernel"] = {"achieved": round(fl2 / sec2 / 1e12, 2), "launches": n2,
                                                   "avg_launch_us": round(1e6 * sec2 / n2, 2),
                                                   "share_of_step_time": round(sec2 / timed_launch_steps / (dt / args.steps), 4)}
            add_binding_roofline(out["roofline"], timer)
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle baseline on %d host cores" % host_cores())
            out["cpu_baseline"] = cpu_baseline(host_cores())
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
