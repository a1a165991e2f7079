#!/usr/bin/env python
"""Headline benchmark: clips/s of forward + backward (+ gradient all-reduce + solver step) of
R50-I3D-NL + LFB-NL on synthetic 32 x 224^2 clips (BASELINE.json `metric`), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is the whole-job clips/s with the inputs already resident in
HBM (weak scaling: 8 clips per GPU, i.e. global batch 64 at 8 GPUs as the north star asks).
`roofline` is the live HIP-event measurement of the dominant kernel family (the implicit-GEMM
NT kernels: conv fprop + dgrad + the batched attention GEMMs): algorithmic FLOPs / summed launch
time against the dense bf16 MFMA peak.  The events bracket every launch of the LAST TIMED STEP on the
stream it is launched on, in the normal two-stream schedule (wgrads overlap the dgrad chain), i.e.
they are the durations a rocprofv3 kernel trace of the same command shows (profiles/).
`cpu_baseline` is the fp32 CPU oracle (a port: the reference has no runnable CPU path) timed on this
box's host cores on ONE clip; `fp32_path` is the same GPU step on the exact-fp32 parity path.
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "split": 2500.0, "mix": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0                           # HBM3E spec peak (same guide; ~6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)    # ~3 s timed region at ~20 ms / step
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ava_r50_lfb_nl",
                    help="ava_r50_lfb_nl (metric config) | charades_r50_baseline | charades_r50_lfb_nl | ava_r101_lfb_nl_3l")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32", "split", "mix"],
                    help="fp16 (default: the 16-bit path -- same speed as bf16, gradients 3x closer to the oracle), bf16, the "
                         "parity-grade paths split / mix, or fp32 (exact-fp32 MFMA)")
    ap.add_argument("--clips-per-gpu", type=int, default=8)
    ap.add_argument("--rois-per-clip", type=int, default=0,
                    help="0 = SURVEY 8d C4 draw U{1..5} per clip (seeded per rank); N > 0 = exactly N per clip")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the extra fp32 parity-path measurement")
    ap.add_argument("--fp32-steps", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--crop", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--set", nargs="*", default=[], metavar="KEY VAL",
                    help="extra cfg overrides, e.g. --set MODEL.FREEZE_BACKBONE False (SURVEY 8d C3 'unfrozen')")
    ap.add_argument("--single-stream", action="store_true",
                    help="development: parameter-gradient kernels on the main stream too (uncontended per-launch times for --detail)")
    ap.add_argument("--solver", default="after", choices=["after", "eager", "tail"],
                    help="after: one solver pass after backward | eager: per gradient bucket during backward | tail: the "
                         "finished buckets beside the last wgrad of backward (Engine.EAGER_SOLVER)")
    ap.add_argument("--no-forward-branches", action="store_true",
                    help="development: every forward step on the main stream (Engine.FORWARD_BRANCHES = False)")
    ap.add_argument("--graph", default="off", choices=["off", "step", "forward"],
                    help="development: replay the single-GPU step (or its forward pass) as one captured HIP graph "
                         "(Engine.STEP_GRAPH; measured slower than the two-stream enqueue, see engine.py)")
    ap.add_argument("--detail", default="", help="write the per-launch GEMM table of the profiled step to this file")
    ap.add_argument("--engine", nargs="*", default=[], metavar="ATTR=VALUE",
                    help="development: Engine class switches for an A/B inside one call, e.g. --engine PLANES=False STEM_PLANES=False")
    ap.add_argument("--launch", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (with VLFB_DIST_FORCE=1 the "
                         "one-rank job then runs the RCCL leg: communicator, bucketed all-reduce, stream hand-over)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched job (0 = pick a free one)")
    ap.add_argument("--split-steps", type=int, default=20, help="timed steps of the extra split-bf16 parity-path measurement")
    ap.add_argument("--no-split-line", action="store_true", help="skip the extra split-bf16 parity-path measurement")
    ap.add_argument("--dp-steps", type=int, default=10,
                    help="data-parallel jobs: timed steps of the same step WITHOUT the gradient exchange (exposed all-reduce time)")
    ap.add_argument("--mix-steps", type=int, default=20, help="timed steps of the extra 'mix' (split forward + fp16 backward) measurement")
    ap.add_argument("--no-mix-line", action="store_true", help="skip the extra 'mix' path measurement")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec this script as N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1 -- the job create_data_parallel_model describes
    (lib/models/model_builder_video.py:142-157), one process per GPU instead of one process for all.  Rank 0 prints the
    JSON line; the return code is the job's."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["VLFB_BENCH_CHILD"] = "1"
    return subprocess.call(cmd, env=env)


def kernel_source_hash():
    """sha256 over the HIP sources of libvlfb_hip.so (what a committed PMC traffic file is valid for)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "video-long-term-feature-banks_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def parity_record(workload, dtype):
    """measured gradient / output error of a path at the benchmarked clip size: the committed summary of
    tests/test_model_gpu.py::test_full_size_clip_matches_oracle (profiles/parity_fullsize_<workload>.json), quoted only while
    it was measured on the current kernel sources"""
    path = os.path.join(ROOT, "profiles", "parity_fullsize_%s.json" % workload)
    if not os.path.exists(path):
        return {"source": "none: no committed full-size parity summary for this workload"}
    rec = json.load(open(path))
    if rec.get("csrc_sha256") != kernel_source_hash():
        return {"source": "none: %s was measured on other kernel sources (csrc hash differs)" % os.path.relpath(path, ROOT)}
    out = dict(rec["paths"].get(dtype, {}))
    out["source"] = "%s (%s, %s; tests/test_model_gpu.py::test_full_size_clip_matches_oracle on these kernel sources)" % (
        os.path.relpath(path, ROOT), rec["size"], rec["metric"])
    return out


def data_parallel_report(eng, args, lr, step_s, device):
    """What a reader needs to judge the N > 1 line (create_data_parallel_model, model_builder_video.py:126-157): who took part
    (ranks of the communicator, backend), that every rank holds the same weights after the timed steps (bit-equality of the
    flat parameter buffer), how much of the gradient exchange the step does NOT hide (same step with and without the
    all-reduce, same run, max over ranks) and what the exchange costs alone (the payload in the step's buckets)."""
    import torch
    import torch.distributed as td
    from vlfb import dist
    from vlfb.engine import Engine
    world = td.get_world_size()
    rep = {"backend": td.get_backend(), "ranks": world, "buckets": len(eng.comm.buckets), "bucket_mb": args.bucket_mb,
           "payload_mb": round(eng.flat_grad.numel() * 4 / 1e6, 1), "handoff": Engine.BUCKET_HANDOFF}

    def vmax(x):
        t = torch.tensor([x], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    # (a) identical weights on every rank after the timed steps (same sums, same solver arithmetic)
    torch.cuda.synchronize()
    words = eng.flat_param.view(torch.int32).to(torch.int64)
    chk = torch.stack([words.sum(), (words * (torch.arange(words.numel(), device=device) % 8191 + 1)).sum()])
    both = torch.stack([chk, -chk])
    td.all_reduce(both, op=td.ReduceOp.MAX)
    rep["weights_bit_identical_across_ranks"] = bool((both[0] == chk).all().item() and (both[1] == -chk).all().item())
    # (b) the payload alone, in the step's buckets (the figure scratch/rccl_allreduce_bench.py reports)
    buf = torch.zeros_like(eng.flat_grad)
    for it in range(3 + 10):
        if it == 3:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
        works = [td.all_reduce(buf[s:e], op=td.ReduceOp.SUM, async_op=True) for s, e, _ in eng.comm.buckets]
        for w in works:
            w.wait()
    torch.cuda.synchronize()
    alone = vmax((time.perf_counter() - t0) / 10)
    rep["allreduce_alone_ms"] = round(alone * 1e3, 3)
    rep["allreduce_alone_busbw_GBps"] = round(2.0 * (world - 1) / world * buf.numel() * 4 / alone / 1e9, 1) if world > 1 else None
    # (c) the same step without the exchange (LAST: the ranks' weights diverge from here on)
    comm, eng.comm = eng.comm, None
    try:
        for _ in range(2):
            eng.train_step(lr)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.dp_steps):
            eng.train_step(lr)
        torch.cuda.synchronize()
        no_comm = vmax((time.perf_counter() - t0) / max(args.dp_steps, 1))
    finally:
        eng.comm = comm
    rep["ms_per_step_without_allreduce"] = round(no_comm * 1e3, 3)
    rep["exposed_allreduce_ms"] = round((step_s - no_comm) * 1e3, 3)
    rep["hidden_fraction"] = round(max(0.0, min(1.0, 1.0 - (step_s - no_comm) / alone)), 3) if alone > 0 else None
    return rep


def cpu_baseline(workload, frames, crop, rois_per_clip):
    """the fp32 torch-CPU oracle forward+backward of ONE full clip on the host cores (bounded: at most
    3 runs / ~30 s).  32 threads: torch's CPU conv3d collapses when all 256 hardware threads of the
    GPU box are used (measured: 0.2 s at 16 threads vs 137 s at 256 for an 8x64^2 clip)."""
    import torch
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from oracle import model as om
    load_preset(workload, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", frames,
                           "TRAIN.CROP_SIZE", crop])
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    inputs = om.synth_inputs(cfg, 1, "train", seed=2, rois_per_clip=[rois_per_clip] if cfg.DATASET == "ava" else None,
                             crop=crop, frames=frames)
    params = om.synth_params(cfg, seed=2)
    times = []
    budget = time.time() + 30.0
    t0 = time.time()
    om.run(cfg, params, inputs, "train", torch.float32, True, lambda name: 1)      # warm-up
    warm = time.time() - t0
    while len(times) < 3 and (not times or time.time() + times[-1] < budget):
        t0 = time.time()
        om.run(cfg, params, inputs, "train", torch.float32, True, lambda name: 1)
        times.append(time.time() - t0)
    best = sorted(times)[len(times) // 2]
    return {"value": 1.0 / best, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "fp32 torch-CPU oracle fwd+bwd of one %dx%dx%d clip: 1 warm-up (%.2f s) + %d timed run(s), median %.2f s, "
                      "%d threads of %d" % (frames, crop, crop, warm, len(times), best, cores, os.cpu_count() or 1)}


def main():
    args = parse()
    if "RANK" not in os.environ and not os.environ.get("VLFB_BENCH_CHILD") and (args.gpus > 1 or args.launch):
        sys.exit(self_launch(args))
    import torch
    from vlfb import dist
    dist.init_from_env()
    world, rank = dist.world_size(), dist.rank()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the job has %d rank(s) (WORLD_SIZE); start it as `python bench.py --gpus %d` "
                         "(self-launching) or through torch.distributed.run with --nproc-per-node %d"
                         % (args.gpus, world, args.gpus, args.gpus))
    # (VLFB_BENCH_ONE_DEVICE=1, with VLFB_DIST_BACKEND=gloo: every rank on cuda:0 -- how a one-GPU box drives the N > 1
    #  code of this file, tests/test_bench_launch_gpu.py; never a measurement)
    one_device = os.environ.get("VLFB_BENCH_ONE_DEVICE", "0") == "1"
    device = "cuda:0" if one_device else "cuda:%d" % dist.local_rank()
    torch.cuda.set_device(device)

    from vlfb import hip, synth
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    import utils.lr_policy as lr_policy

    clips = args.clips_per_gpu
    load_preset(args.workload, ["NUM_GPUS", world, "TRAIN.BATCH_SIZE", clips * world,
                                "TRAIN.VIDEO_LENGTH", args.frames, "TRAIN.CROP_SIZE", args.crop] + list(args.set))
    model = ModelBuilder(train=True, split="train", name="bench")
    model.build_model(suffix="_train")
    for kv in args.engine:
        k, v = kv.split("=", 1)
        assert hasattr(Engine, k), "Engine has no switch %r" % k
        if "," in v:
            v = tuple(int(x) for x in v.split(","))
        setattr(Engine, k, {"True": True, "False": False}.get(v, v if not str(v).lstrip("-").isdigit() else int(v)))
    Engine.FORWARD_BRANCHES = not args.no_forward_branches
    Engine.STEP_GRAPH = {"off": False, "step": True, "forward": "forward"}[args.graph]
    Engine.EAGER_SOLVER = {"after": False, "eager": True, "tail": "tail"}[args.solver]
    eng = Engine(model, args.dtype, device=device, base_seed=cfg.RNG_SEED, side_stream=not args.single_stream)
    rois = args.rois_per_clip if args.rois_per_clip > 0 else synth.rois_per_clip_draw(clips, seed=cfg.RNG_SEED + rank)
    batch = synth.inputs(cfg, clips, rois, seed=cfg.RNG_SEED + rank, crop=args.crop, frames=args.frames)
    n_rois = int(batch["proposals_train"].shape[0]) if "proposals_train" in batch else 0
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
    eng.feed_params(synth.params(model, seed=cfg.RNG_SEED))
    for k, v in batch.items():
        if k in model.input_blob_names:
            eng.feed(k, v)
    eng.enable_data_parallel(args.bucket_mb)
    lr = float(lr_policy.get_lr_at_iter(0))

    for _ in range(args.warmup):
        eng.train_step(lr)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            # HIP-event brackets around every GEMM launch of the last timed step, each on the stream the
            # launch goes to, in the normal two-stream schedule: the durations are the ones the step pays
            # (and the ones a rocprofv3 kernel trace of this command reports).  Events cannot sit inside a
            # replayed graph, so this one step is enqueued launch by launch -- inside the timed region.
            hip.PROFILE = []
        eng.train_step(lr)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    prof, hip.PROFILE = hip.PROFILE, None
    if world > 1:
        import torch.distributed as td
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    # host-side cost of enqueuing one step (untimed extra step): if this approaches ms_per_step the
    # loop is launch-bound and wants HIP-graph capture
    th0 = time.perf_counter()
    eng.train_step(lr)
    host_ms = (time.perf_counter() - th0) * 1e3
    torch.cuda.synchronize()
    loss = float(eng.fetch("loss").reshape(-1)[0])

    # ---- live roofline of the GEMM kernel families (this rank) -----------------------------------
    fam = {"nt": [0.0, 0.0, 0, 0.0, 0.0], "tn": [0.0, 0.0, 0, 0.0, 0.0]}
    rows = []
    for mode, flops, e0, e1, tag, nbytes in prof or []:
        f = fam["tn" if mode == hip.WGRAD else "nt"]
        sec = e0.elapsed_time(e1) * 1e-3
        f[0] += flops
        f[1] += sec
        f[2] += 1
        f[3] += nbytes
        # per-launch attainable time: whichever of the MFMA and the HBM roof binds THIS launch
        f[4] += max(flops / (PEAK_TFLOPS[args.dtype] * 1e12), nbytes / (HBM_PEAK_GBPS * 1e9))
        rows.append((sec, flops, tag))
    # HBM bytes per launch: PMC counters cannot be read from inside this process, so they come from the
    # committed rocprofv3 --pmc passes of THIS command line (profiles/*_hbm_traffic.json, made by
    # scratch/pmc_traffic.py); any other workload / dtype / batch reports null and says why
    # A traffic file is only valid for the kernels it was measured on: it carries the hash of csrc/ at measurement time
    # (scratch/pmc_traffic.py) and anything else -- another workload, an edited kernel, an unstamped file -- reports null.
    traffic, traffic_source = {}, "none: no committed PMC pass for this workload/dtype/batch"
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if args.workload == "ava_r50_lfb_nl" and clips == 8 and os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("dtype", "bf16") != args.dtype:
            traffic_source = "none: profiles/hbm_traffic.json was measured with --dtype %s" % t.get("dtype", "bf16")
        elif t.get("csrc_sha256") == kernel_source_hash():
            traffic = {"nt": t["gemm_nt"]["bytes_per_launch"], "tn": t["gemm_tn"]["bytes_per_launch"]}
            traffic_source = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on " \
                             "this kernel source, read side x2 per MI355X_MICROARCH.md; not measured in this run)"
        else:
            traffic_source = "none: profiles/hbm_traffic.json was measured on other kernel sources (csrc hash differs)"
    if args.detail and rank == 0:
        with open(args.detail, "w") as fh:
            agg = collections.OrderedDict()
            for sec, flops, tag in rows:
                a = agg.setdefault(tag, [0.0, 0.0, 0])
                a[0] += sec; a[1] += flops; a[2] += 1
            fh.write("%10s %8s %6s %9s  %s\n" % ("total_us", "TFLOP/s", "calls", "GFLOP", "launch"))
            for tag, (sec, flops, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                fh.write("%10.1f %8.1f %6d %9.2f  %s\n" % (sec * 1e6, flops / sec / 1e12, n, flops / 1e9, tag))
    peak = PEAK_TFLOPS[args.dtype]

    def roof(key, kernel):
        fl, sec, n, nb, att = fam[key]
        ach = fl / sec / 1e12 if sec > 0 else 0.0
        return {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic.get(key), "traffic_source": traffic_source,
                "source": "HIP events on the launch stream around every launch of the last timed step "
                          "(two-stream steady state); sum of algorithmic FLOP / sum of launch durations",
                "algorithmic_bytes_per_launch": round(nb / max(n, 1)),
                "algorithmic_GBps": round(nb / sec / 1e9, 1) if sec > 0 else 0.0, "launches_per_step": n,
                "avg_launch_us": round(sec / max(n, 1) * 1e6, 2), "gflop_per_step": round(fl / 1e9, 1),
                "ms_per_step": round(sec * 1e3, 3),
                # sum over launches of max(flops/MFMA peak, algorithmic bytes/HBM peak): what the same
                # launch list would take with every launch on its own roof (thin-K layers are HBM-bound)
                "attainable_ms_per_step": round(att * 1e3, 3),
                "frac_of_attainable": round(att / sec, 4) if sec > 0 else 0.0}

    clips_total = clips * world * args.steps
    value = clips_total / elapsed
    out = collections.OrderedDict([
        ("metric", "clips/sec fwd+bwd R50-I3D-NL+LFB-NL, 32x224^2 synthetic"
         if (args.workload, args.frames, args.crop) == ("ava_r50_lfb_nl", 32, 224)
         else "clips/sec fwd+bwd %s, %dx%d^2 synthetic" % (args.workload, args.frames, args.crop)),
        ("value", round(value, 3)), ("unit", "clips/s"), ("n_gpus", world), ("steps", args.steps),
        ("warmup", args.warmup), ("ms_per_step", round(elapsed / args.steps * 1e3, 3)),
        ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None), ("dtype", args.dtype),
        ("data", "synthetic"),
        ("config", {"workload": "%s fwd+bwd+allreduce+sgd, %d clips/GPU (global batch %d), %s, %dx%dx%d clips"
                                % (args.workload, clips, clips * world,
                                   ("%d RoIs/clip" % args.rois_per_clip) if args.rois_per_clip > 0 else
                                   ("RoIs/clip ~ U{1..5} (%d on rank 0)" % n_rois), args.frames, args.crop, args.crop)
                                + ((" [" + " ".join(args.set) + "]") if args.set else ""),
                    "parallelism": "dp%d" % world, "final_loss": loss}),
        ("roofline", roof("nt", "gemm_nt_kernel (implicit-GEMM conv fprop+dgrad, attention NT GEMMs)")),
        ("roofline_wgrad", roof("tn", "gemm_tn_kernel (implicit-GEMM conv wgrad, attention TN GEMMs)")),
        # algorithmic FLOP of every contraction the step launched (this rank's plan: convs, attention products, head) per
        # wall second against the MFMA peak of all GPUs
        ("model_flops_utilisation", round((fam["nt"][0] + fam["tn"][0]) * args.steps / elapsed / 1e12 / peak, 4)),
        ("gflop_per_step_per_gpu", round((fam["nt"][0] + fam["tn"][0]) / 1e9, 1)),
        ("host_enqueue_ms_per_step", round(host_ms, 2)),
        ("host_enqueue_path", "recorded call list (Engine.STEP_TRACE)" if eng._trace is not None else "step objects"),
    ])
    if eng.comm is not None:       # the gradient exchange of this job (None on a one-process run without a process group)
        out["allreduce"] = data_parallel_report(eng, args, lr, elapsed / args.steps, device)
    # The parity-grade paths on the same workload, so that the numbers next to the parity claims exist:
    #   split_path: fp32 storage, every contraction as split-bf16 products on the bf16 matrix cores (three MFMAs per product,
    #               Engine.SPLIT_MATH; csrc/vlfb_gemm_split.hip) -- outputs AND every parameter gradient within 1e-3 of
    #               the fp64 oracle at the benchmarked size (profiles/r03_parity_fullsize_*.txt)
    #   fp32_path:  the same with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 matrix rate)
    step_flops = fam["nt"][0] + fam["tn"][0]

    def extra_path(dtype, steps, what, peak_tf):
        eng2 = Engine(model, dtype, device=device, base_seed=cfg.RNG_SEED)
        eng2.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
        eng2.feed_params(synth.params(model, seed=cfg.RNG_SEED))
        for k, v in batch.items():
            if k in model.input_blob_names:
                eng2.feed(k, v)
        eng2.train_step(lr)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(steps):
            eng2.train_step(lr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t2) / max(steps, 1)
        return {"value": round(clips / dt, 3), "unit": "clips/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "dtype": what,
                "model_flops_utilisation": round(step_flops / dt / 1e12 / peak_tf, 4), "peak_tflops": peak_tf,
                "parity": parity_record(args.workload, dtype)}

    out["parity"] = parity_record(args.workload, args.dtype)
    if world == 1 and args.dtype not in ("fp32", "split", "mix"):
        del eng
        torch.cuda.empty_cache()
        for key, dtype, steps, skip, what, peak_tf in (
                ("mix_path", "mix", args.mix_steps, args.no_mix_line,
                 "fp32 storage + split-bf16 forward products (3 MFMAs per product), fp16 backward (fp16 gradient storage, two-term "
                 "fp16 weights in DGRAD, fp32 / split products around the non-local softmax)", 2500.0 * 3.0 / (3 + 2 + 1)),
                ("split_path", "split", args.split_steps, args.no_split_line,
                 "fp32 storage + split-bf16 products on v_mfma_f32_16x16x32_bf16 (%d per product forward, %d backward)" % Engine.SPLIT_MATH,
                 2500.0 * 3.0 / (Engine.SPLIT_MATH[0] + 2 * Engine.SPLIT_MATH[1])),   # 1/3 of a step's FLOP are forward
                ("fp32_path", "fp32", args.fp32_steps, args.no_fp32_line,
                 "fp32 storage + v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak)", PEAK_TFLOPS["fp32"])):
            if skip:
                continue
            try:
                out[key] = extra_path(dtype, steps, what, peak_tf)
                torch.cuda.empty_cache()
            except Exception as e:   # a report, never a gate
                out[key] = {"value": None, "error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.frames, args.crop, max(args.rois_per_clip, 3))
            except Exception as e:  # the baseline is a report, never a gate
                out["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    dist.barrier()
    if dist.initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
