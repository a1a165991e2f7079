#!/usr/bin/env python
"""Headline benchmark: clips/s of forward + backward (+ gradient all-reduce + solver step) of
R50-I3D-NL + LFB-NL on synthetic 32 x 224^2 clips (BASELINE.json `metric`), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is the whole-job clips/s with the inputs already resident in
HBM (weak scaling: 8 clips per GPU, i.e. global batch 64 at 8 GPUs as the north star asks).

The default dtype is `mix` (split-bf16 forward on fp32 storage + fp16 backward): the
fastest path whose outputs AND parameter gradients are inside the north star's 1e-3 at the benchmarked
size (`parity`, quoted from the committed full-size comparison while the kernel sources are the ones it
was measured on).  The 16-bit paths are faster and miss that gate by an order of magnitude: they are
sub-records (`fp16_path`; `--dtype bf16` for the other one), as are `split_path` / `fp32_path`.

`roofline` is a live HIP-event measurement of the dominant kernel family of the benchmarked dtype --
for `mix` the split-bf16 NT kernels of the forward pass -- on ONE extra step behind the timed region
(same two-stream schedule, every GEMM launch bracketed on the stream it is launched on: the durations
a rocprofv3 kernel trace of this command shows, profiles/): `achieved` = algorithmic FLOP / launch
time, `peak` = 2.5 PFLOP/s dense 16-bit MFMA / MFMA instructions per algorithmic product of that
family (3 for a split-bf16 product, 2 for an fp16 DGRAD with two-term weights, 1 otherwise; the
exact-fp32 kernels: 157.3 TFLOP/s), so `frac` is the fraction of the MFMA roof the family's
instructions reach.  `roofline_families` has every family of the step, `roofline_wgrad` the dominant
TN (weight-gradient) family.
`cpu_baseline` is the fp32 CPU oracle (a port: the reference has no runnable CPU path) timed on this
box's host cores on ONE clip, and the same clip's outputs on the GPU checked against it.
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

MFMA16_PEAK, MFMA32_PEAK = 2500.0, 157.3         # MI355X dense MFMA peaks, TFLOP/s: 16-bit operands / exact fp32 (MI355X_MICROARCH.md)
FAMILY_KERNELS = collections.OrderedDict([
    ("nt_pair", "gemm_nt_kernel<f16, PAIR> / gemm_nt8_kernel<f16, PAIR> / stem_fprop_pair_kernel (two-plane fp16 implicit-GEMM conv FPROP: three fp16 MFMAs per product on pre-split operands; the `mix` forward)"),
    ("nt_split", "gemm_nt_sp_kernel / gemm_nt_pl_kernel / gemm_skinny_nt_sp_kernel (split-bf16 implicit-GEMM conv FPROP / DGRAD + NT attention products on fp32 storage)"),
    ("nt_16", "gemm_nt_kernel / gemm_nt8_kernel / gemm_nts_kernel / stem_fprop_kernel / conv_rows64_kernel / gemm_skinny_nt_kernel "
              "(16-bit implicit-GEMM conv FPROP + DGRAD, NT attention products)"),
    ("nt_f32", "gemm_nt_kernel<float> (exact-fp32 MFMA conv FPROP + DGRAD, NT attention products)"),
    ("tn_16", "gemm_tn_tr_kernel / gemm_tn8_kernel / stem_wgrad_kernel / wgrad_rows_kernel (16-bit implicit-GEMM conv WGRAD, TN attention products)"),
    ("tn_split", "gemm_tn_sp_kernel / gemm_tn_tr_kernel<SP> (split-bf16 conv WGRAD + TN attention products on fp32 storage)"),
    ("tn_f32", "gemm_tn_kernel<float> (exact-fp32 MFMA conv WGRAD)"),
])
HBM_PEAK_GBPS = 8000.0                           # HBM3E spec peak (same guide; ~6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)    # ~3 s timed region at ~20 ms / step
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ava_r50_lfb_nl",
                    help="ava_r50_lfb_nl (metric config) | charades_r50_baseline | charades_r50_lfb_nl | ava_r101_lfb_nl_3l")
    ap.add_argument("--dtype", default="mix", choices=["mix", "fp16", "bf16", "split", "fp32"],
                    help="mix (default: split-bf16 forward + fp16 backward -- the fastest path inside the 1e-3 gradient gate), the "
                         "16-bit throughput paths fp16 / bf16 (outside it), split (all split-bf16, 25x inside it) or fp32 (exact-fp32 MFMA)")
    ap.add_argument("--clips-per-gpu", type=int, default=8)
    ap.add_argument("--rois-per-clip", type=int, default=0,
                    help="0 = SURVEY 8d C4 draw U{1..5} per clip (seeded per rank); N > 0 = exactly N per clip")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the extra fp32 parity-path measurement")
    ap.add_argument("--fp32-steps", type=int, default=10)
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
    ap.add_argument("--mix-steps", type=int, default=20, help="timed steps of the extra 'mix' measurement (when --dtype is another path)")
    ap.add_argument("--no-mix-line", action="store_true", help="skip the extra 'mix' path measurement")
    ap.add_argument("--fp16-steps", type=int, default=20, help="timed steps of the extra fp16 throughput-path measurement")
    ap.add_argument("--no-fp16-line", action="store_true", help="skip the extra fp16 path measurement")
    ap.add_argument("--bf16-steps", type=int, default=20, help="timed steps of the extra bf16 throughput-path measurement (BASELINE.json configs[1])")
    ap.add_argument("--no-bf16-line", action="store_true", help="skip the extra bf16 path measurement")
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


def parity_source_hash():
    """sha256 over everything that decides the arithmetic of a step: the HIP sources AND the engine (which dtype every blob /
    gradient slot has, which launches run: lib/vlfb/engine.py, hip.py) -- what a committed parity summary is valid for"""
    import hashlib
    h = hashlib.sha256(kernel_source_hash().encode())
    for f in ("engine.py", "hip.py"):
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib", "vlfb", f), "rb").read())
    return h.hexdigest()


def parity_record(workload, dtype):
    """measured gradient / output error of a path at the benchmarked clip size: the committed summary of
    tests/test_model_gpu.py::test_full_size_clip_matches_oracle (profiles/parity_fullsize_<workload>.json), quoted only while
    it was measured on the current kernel sources"""
    path = os.path.join(ROOT, "profiles", "parity_fullsize_%s.json" % workload)
    if not os.path.exists(path):
        return {"source": "none: no committed full-size parity summary for this workload"}
    rec = json.load(open(path))
    if rec.get("source_sha256") != parity_source_hash():
        return {"source": "none: %s was measured on other kernel / engine sources (hash of csrc + lib/vlfb/engine.py + hip.py differs)"
                          % os.path.relpath(path, ROOT)}
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
    comm, eng.comm = eng.comm, None                         # (same host path as the timed step: the recorded call list, re-recorded without the collectives)
    trace = eng.STEP_TRACE
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
        eng.STEP_TRACE = trace
    rep["ms_per_step_without_allreduce"] = round(no_comm * 1e3, 3)
    rep["exposed_allreduce_ms"] = round((step_s - no_comm) * 1e3, 3)
    rep["hidden_fraction"] = round(max(0.0, min(1.0, 1.0 - (step_s - no_comm) / alone)), 3) if alone > 0 else None
    return rep


def cpu_baseline(workload, frames, crop, rois_per_clip, dtype=None, device=None):
    """the fp32 torch-CPU oracle forward+backward of ONE full clip on the host cores (bounded: at most
    3 runs / ~30 s).  32 threads: torch's CPU conv3d collapses when all 256 hardware threads of the
    GPU box are used (measured: 0.2 s at 16 threads vs 137 s at 256 for an 8x64^2 clip).
    The oracle is the checker here as well: the same clip goes through a one-clip engine of the benchmarked dtype on the GPU
    and its outputs are compared with the timed run's (`gpu_outputs_vs_this_run`; never the thing measured)."""
    import collections as _c
    import numpy as np
    import torch
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from vlfb import rng as vrng
    from oracle import model as om
    load_preset(workload, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", frames,
                           "TRAIN.CROP_SIZE", crop])
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    inputs = om.synth_inputs(cfg, 1, "train", seed=2, rois_per_clip=[rois_per_clip] if cfg.DATASET == "ava" else None,
                             crop=crop, frames=frames)
    params = om.synth_params(cfg, seed=2)
    seed_fn = lambda name: vrng.dropout_seed(cfg.RNG_SEED, name, 0)       # the masks the engine draws at iteration 0
    times = []
    budget = time.time() + 30.0
    t0 = time.time()
    blobs, _ = om.run(cfg, params, inputs, "train", torch.float32, True, seed_fn)      # warm-up
    warm = time.time() - t0
    while len(times) < 3 and (not times or time.time() + times[-1] < budget):
        t0 = time.time()
        blobs, _ = om.run(cfg, params, inputs, "train", torch.float32, True, seed_fn)
        times.append(time.time() - t0)
    best = sorted(times)[len(times) // 2]
    rec = {"value": 1.0 / best, "unit": "clips/s", "cores": cores, "kind": "port",
           "sample": "fp32 torch-CPU oracle fwd+bwd of one %dx%dx%d clip: 1 warm-up (%.2f s) + %d timed run(s), median %.2f s, "
                     "%d threads of %d" % (frames, crop, crop, warm, len(times), best, cores, os.cpu_count() or 1)}
    if dtype is not None:
        try:
            from models.model_builder_video import ModelBuilder
            from vlfb.engine import Engine
            model = ModelBuilder(train=True, split="train", name="bench_check")
            model.build_model(suffix="_train")
            eng = Engine(model, dtype, device=device, base_seed=cfg.RNG_SEED)
            eng.plan(_c.OrderedDict((k + "_train", v.shape) for k, v in inputs.items() if (k + "_train") in model.input_blob_names))
            eng.feed_params(params)
            for k, v in inputs.items():
                if (k + "_train") in model.input_blob_names:
                    eng.feed(k + "_train", v)
            eng.forward()
            torch.cuda.synchronize()
            rel = lambda a, r: float(np.linalg.norm((np.asarray(a, np.float64) - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-300))
            chk = {}
            for name in ("res5_2_branch2c_bn", "pool5", "prob"):
                if name in blobs:
                    got = eng.fetch(name)
                    chk[name] = rel(got, blobs[name].detach().double().numpy().reshape(got.shape))
            ref_loss = float(blobs["loss"].detach())
            chk["loss"] = abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) / abs(ref_loss)
            rec["gpu_outputs_vs_this_run"] = {"dtype": dtype, "relative_l2": {k: float("%.3e" % v) for k, v in chk.items()},
                                              "within_1e-3": bool(max(chk.values()) < 1e-3)}
        except Exception as e:     # a report, never a gate
            rec["gpu_outputs_vs_this_run"] = {"error": repr(e)}
    return rec


DTYPE_NOTES = {
    "mix": "forward: trunk activations and weights as TWO fp16 planes (hi + lo, ~22 bits), a product = hi.hi + hi.lo + lo.hi on "
           "v_mfma_f32_16x16x32_f16 (3 MFMAs per product, fp32 accumulate, nothing converted in the k-loop); non-local / FBO internals "
           "fp32 with split-bf16 products; backward on v_mfma_f32_16x16x32_f16 with fp16 gradient storage (the hi plane is the fp16 "
           "operand) -- two-term fp16 weights in DGRAD (2 MFMAs per product; one gradient tile in LDS per pair of weight tiles where the "
           "128-row kernel runs it), two-term gradients on the residual stream / sums of DGRADs, fp32 gradients + split products around "
           "the non-local softmax and in the head",
    "fp16": "fp16 storage and v_mfma_f32_16x16x32_f16 operands end to end, fp32 accumulate, static power-of-two loss scale",
    "bf16": "bf16 storage and v_mfma_f32_16x16x32_bf16 operands end to end, fp32 accumulate",
    "split": "fp32 storage, every contraction as split-bf16 products (3 x v_mfma_f32_16x16x32_bf16 per product) in both directions",
    "fp32": "fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s peak)",
}


def profile_step(eng, lr, hip, torch):
    """one train step with HIP events around every GEMM launch (hip.PROFILE) -> ({family: sums}, [(sec, flops, tag)])"""
    hip.PROFILE = []
    try:
        eng.train_step(lr)
        torch.cuda.synchronize()
    finally:
        prof, hip.PROFILE = hip.PROFILE, None
    fam, rows = collections.OrderedDict(), []
    for mode, flops, e0, e1, tag, nbytes, family, mpp in prof or []:
        sec = e0.elapsed_time(e1) * 1e-3
        f = fam.setdefault(family, {"flops": 0.0, "mfma_flops": 0.0, "sec": 0.0, "n": 0, "bytes": 0.0, "att": 0.0})
        peak = MFMA32_PEAK if family.endswith("_f32") else MFMA16_PEAK
        f["flops"] += flops
        f["mfma_flops"] += flops * mpp
        f["sec"] += sec
        f["n"] += 1
        f["bytes"] += nbytes
        # per-launch attainable time: whichever of the MFMA and the HBM roof binds THIS launch
        f["att"] += max(flops * mpp / (peak * 1e12), nbytes / (HBM_PEAK_GBPS * 1e9))
        rows.append((sec, flops, tag))
    return fam, rows


def dominant(fam, side):
    """the family of this side (nt / tn) that the step spends most launch time in"""
    keys = [k for k in fam if k.startswith(side)]
    return max(keys, key=lambda k: fam[k]["sec"]) if keys else None


def roof_record(key, fam, traffic, traffic_source, brief=False):
    if key is None:
        return None
    f = fam[key]
    fl, sec, n = f["flops"], f["sec"], f["n"]
    base = MFMA32_PEAK if key.endswith("_f32") else MFMA16_PEAK
    mpp = f["mfma_flops"] / fl if fl > 0 else 1.0        # MFMA instructions per algorithmic product (FLOP-weighted)
    ach = fl / sec / 1e12 if sec > 0 else 0.0
    peak = base / mpp
    rec = collections.OrderedDict([
        ("kernel", FAMILY_KERNELS[key]), ("bound", "mfma"), ("achieved", round(ach, 2)), ("peak", round(peak, 1)),
        ("unit", "TFLOP/s"), ("frac", round(ach / peak, 4)),
        ("mfma_per_product", round(mpp, 3)), ("mfma_rate_tflops", round(ach * mpp, 1)), ("mfma_peak_tflops", base),
        ("launches_per_step", n), ("avg_launch_us", round(sec / max(n, 1) * 1e6, 2)),
        ("gflop_per_step", round(fl / 1e9, 1)), ("ms_per_step", round(sec * 1e3, 3)),
        ("traffic", traffic.get(key)),
        # measured HBM bytes (PMC) / algorithmic bytes per launch: > 1 = re-reads (operand panels per column tile, slabs)
        ("traffic_ratio", round(traffic[key] / (f["bytes"] / max(n, 1)), 3) if traffic.get(key) and f["bytes"] > 0 else None)])
    if brief:
        return rec
    rec.update([
        ("traffic_source", traffic_source),
        ("source", "HIP events on the launch stream around every launch of ONE step behind the timed region (two-stream steady "
                   "state); achieved = sum of algorithmic FLOP / sum of launch durations; peak = %.1f TFLOP/s dense MFMA / "
                   "%.3g MFMA instructions per algorithmic product" % (base, mpp)),
        ("algorithmic_bytes_per_launch", round(f["bytes"] / max(n, 1))),
        ("algorithmic_GBps", round(f["bytes"] / sec / 1e9, 1) if sec > 0 else 0.0),
        # sum over launches of max(MFMA FLOP / MFMA peak, algorithmic bytes / HBM peak): what the same launch list
        # would take with every launch on its own roof (thin-K layers are HBM-bound)
        ("attainable_ms_per_step", round(f["att"] * 1e3, 3)),
        ("frac_of_attainable", round(f["att"] / sec, 4) if sec > 0 else 0.0)])
    return rec


def traffic_record(args, clips):
    """HBM bytes per launch and family: PMC counters cannot be read from inside this process, so they come from the committed
    rocprofv3 --pmc passes of THIS command line (profiles/hbm_traffic.json, made by tools/pmc_traffic.py).  A traffic file
    is only valid for the kernels it was measured on: it carries the hash of csrc/ at measurement time, and anything else --
    another workload / dtype / batch, an edited kernel, an unstamped file -- reports null and says why."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not (args.workload == "ava_r50_lfb_nl" and clips == 8 and os.path.exists(tpath)):
        return {}, "none: no committed PMC pass for this workload/dtype/batch"
    t = json.load(open(tpath))
    if t.get("dtype") != args.dtype:
        return {}, "none: profiles/hbm_traffic.json was measured with --dtype %s" % t.get("dtype")
    if t.get("csrc_sha256") != kernel_source_hash():
        return {}, "none: profiles/hbm_traffic.json was measured on other kernel sources (csrc hash differs)"
    return ({k: v["bytes_per_launch"] for k, v in t.get("families", {}).items()},
            "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this kernel source, "
            "read side x2 per MI355X_MICROARCH.md; not measured in this run)")


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
    for i in range(args.steps):                  # the timed region: K steps, nothing else
        eng.train_step(lr)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
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
    host_path = "recorded call list (Engine.STEP_TRACE)" if eng._trace is not None else "step objects"
    # ---- live roofline of the GEMM kernel families (this rank): ONE more step, behind the timed region, with HIP-event
    # brackets around every GEMM launch on the stream it goes to, in the normal two-stream schedule (events cannot sit
    # inside a recorded / replayed step, so this step is enqueued launch by launch)
    fam, rows = profile_step(eng, lr, hip, torch)
    traffic, traffic_source = traffic_record(args, clips)
    if args.detail and rank == 0:
        with open(args.detail, "w") as fh:
            agg = collections.OrderedDict()
            for sec, flops, tag in rows:
                a = agg.setdefault(tag, [0.0, 0.0, 0])
                a[0] += sec; a[1] += flops; a[2] += 1
            fh.write("%10s %8s %6s %9s  %s\n" % ("total_us", "TFLOP/s", "calls", "GFLOP", "launch"))
            for tag, (sec, flops, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                fh.write("%10.1f %8.1f %6d %9.2f  %s\n" % (sec * 1e6, flops / sec / 1e12, n, flops / 1e9, tag))
    step_flops = sum(f["flops"] for f in fam.values())
    step_mfma_flops = sum(f["mfma_flops"] for f in fam.values())
    nt_key, tn_key = dominant(fam, "nt"), dominant(fam, "tn")

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
                    "parallelism": "dp%d" % world, "final_loss": loss,
                    "arithmetic": DTYPE_NOTES[args.dtype]}),
        ("roofline", roof_record(nt_key, fam, traffic, traffic_source)),
        ("roofline_wgrad", roof_record(tn_key, fam, traffic, traffic_source)),
        ("roofline_families", collections.OrderedDict((k, roof_record(k, fam, traffic, traffic_source, brief=True))
                                                      for k in FAMILY_KERNELS if k in fam)),
        # algorithmic FLOP of every contraction the step launched (this rank's plan: convs, attention products, head) per
        # wall second against the 16-bit MFMA peak of all GPUs; and the same with every product counted as the MFMA
        # instructions it is executed with (3 per split-bf16 product, 2 per two-term DGRAD product)
        ("model_flops_utilisation", round(step_flops * args.steps / elapsed / 1e12 / MFMA16_PEAK, 4)),
        ("mfma_flops_utilisation", round(step_mfma_flops * args.steps / elapsed / 1e12 / MFMA16_PEAK, 4)),
        ("gflop_per_step_per_gpu", round(step_flops / 1e9, 1)),
        ("host_enqueue_ms_per_step", round(host_ms, 2)),
        ("host_enqueue_path", host_path),
    ])
    if eng.comm is not None:       # the gradient exchange of this job (None on a one-process run without a process group)
        out["allreduce"] = data_parallel_report(eng, args, lr, elapsed / args.steps, device)
    out["parity"] = parity_record(args.workload, args.dtype)

    # The other paths on the same workload, so that every number quoted next to a parity claim exists in THIS line:
    #   mix_path   (when --dtype is not mix): the default path, see the module docstring
    #   fp16_path:  16-bit storage and MFMA operands end to end -- the throughput path; gradients 15x outside the gate
    #   split_path: fp32 storage, every contraction as split-bf16 products (three MFMAs per product) -- 25x inside it
    #   fp32_path:  the same with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit matrix rate)
    def extra_path(dtype, steps):
        eng2 = Engine(model, dtype, device=device, base_seed=cfg.RNG_SEED)
        eng2.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
        eng2.feed_params(synth.params(model, seed=cfg.RNG_SEED))
        for k, v in batch.items():
            if k in model.input_blob_names:
                eng2.feed(k, v)
        eng2.train_step(lr)
        eng2.train_step(lr)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(steps):
            eng2.train_step(lr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t2) / max(steps, 1)
        fam2, _ = profile_step(eng2, lr, hip, torch)
        fl2 = sum(f["flops"] for f in fam2.values())
        rec = {"value": round(clips / dt, 3), "unit": "clips/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
               "dtype": dtype, "arithmetic": DTYPE_NOTES[dtype],
               "model_flops_utilisation": round(fl2 / dt / 1e12 / MFMA16_PEAK, 4),
               "roofline": roof_record(dominant(fam2, "nt"), fam2, {}, "none: PMC passes are made for the default dtype", brief=True),
               "roofline_wgrad": roof_record(dominant(fam2, "tn"), fam2, {}, "none: PMC passes are made for the default dtype", brief=True),
               "parity": parity_record(args.workload, dtype)}
        del eng2
        return rec

    if world == 1:
        del eng
        torch.cuda.empty_cache()
        for key, dtype, steps, skip in (("mix_path", "mix", args.mix_steps, args.no_mix_line),
                                        ("fp16_path", "fp16", args.fp16_steps, args.no_fp16_line),
                                        ("bf16_path", "bf16", args.bf16_steps, args.no_bf16_line),
                                        ("split_path", "split", args.split_steps, args.no_split_line),
                                        ("fp32_path", "fp32", args.fp32_steps, args.no_fp32_line)):
            if skip or dtype == args.dtype:
                continue
            try:
                out[key] = extra_path(dtype, steps)
                torch.cuda.empty_cache()
            except Exception as e:   # a report, never a gate
                out[key] = {"value": None, "error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.frames, args.crop, max(args.rois_per_clip, 3), args.dtype, device)
            except Exception as e:  # the baseline is a report, never a gate
                out["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    dist.barrier()
    if dist.initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
