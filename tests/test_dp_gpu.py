"""Clip-level data parallelism end to end on the GPU engine: two processes (each one replica with
1 clip) must produce, after the bucketed all-reduce, exactly the gradients of ONE process holding
both clips (loss is pre-scaled 1/NUM_GPUS, gradients are summed).

Only one GPU is visible on the test box and RCCL refuses two ranks on one device, so the two
replicas share cuda:0 and talk over gloo; the code path (broadcast of weights, GradComm buckets
issued during backward, side-stream join, solver wait) is the one `bench.py --gpus N` uses over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OV = ["TRAIN.VIDEO_LENGTH", 16, "TRAIN.CROP_SIZE", 64]


def _setup_paths():
    for p in (os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_replica(world, rank, clips_per_rank, all_inputs, params, seed_iter=0):
    import collections
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    load_preset("charades_r50_baseline", ["NUM_GPUS", world, "TRAIN.BATCH_SIZE", clips_per_rank * world,
                                          "TRAIN.DROPOUT_RATE", 0.0] + OV)
    model = ModelBuilder(train=True, split="train", name="dp")
    model.build_model(suffix="_train")
    eng = Engine(model, "fp32", device="cuda:0", base_seed=2)
    sl = slice(rank * clips_per_rank, (rank + 1) * clips_per_rank)
    mine = {k + "_train": v[sl] for k, v in all_inputs.items()}
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in mine.items()))
    eng.feed_params(params)
    for k, v in mine.items():
        eng.feed(k, v)
    eng.enable_data_parallel(bucket_mb=4)
    eng.forward()
    eng.backward()
    if eng.comm is not None:
        eng.comm.wait()
    torch.cuda.synchronize()
    return eng


def _worker(rank, world, port, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), VLFB_FORCE_DEVICE="0", VLFB_DIST_BACKEND="gloo")
    from vlfb import dist
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from oracle import model as om
    dist.init_from_env()
    load_preset("charades_r50_baseline", ["NUM_GPUS", world, "TRAIN.BATCH_SIZE", world] + OV)
    inputs = om.synth_inputs(cfg, world, "train", seed=2, crop=64, frames=16)
    params = om.synth_params(cfg, seed=2)
    if rank != 0:   # wrong weights on purpose: enable_data_parallel must broadcast rank 0's
        params = {k: v * 0.5 for k, v in params.items()}
    eng = _run_replica(world, rank, 1, inputs, params)
    assert eng.comm is not None and len(eng.comm.buckets) > 5
    out = {n: eng.fetch_grad(n) for n in ("pred_w", "res5_2_branch2c_w", "res3_1_branch2b_w", "conv1_w")}
    out["_w"] = eng.fetch_param("res4_0_branch2a_w")
    q.put((rank, out))
    dist.barrier()
    torch.distributed.destroy_process_group()


def test_two_replicas_match_one_process_with_both_clips():
    import torch.multiprocessing as mp
    _setup_paths()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process, both clips, NUM_GPUS = 1
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from oracle import model as om
    load_preset("charades_r50_baseline", ["NUM_GPUS", 2, "TRAIN.BATCH_SIZE", 2] + OV)
    inputs = om.synth_inputs(cfg, 2, "train", seed=2, crop=64, frames=16)
    params = om.synth_params(cfg, seed=2)
    eng = _run_replica(1, 0, 2, inputs, params)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert np.array_equal(results[0]["_w"], results[1]["_w"]), "weights were not broadcast"
    for n in ("pred_w", "res5_2_branch2c_w", "res3_1_branch2b_w", "conv1_w"):
        assert np.array_equal(results[0][n], results[1][n]), "ranks disagree after all-reduce: " + n
        assert rel(results[0][n], eng.fetch_grad(n)) < 2e-4, (n, rel(results[0][n], eng.fetch_grad(n)))


def _nccl_worker(port, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      VLFB_DIST_FORCE="1", VLFB_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from vlfb import dist
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from oracle import model as om
    dist.init_from_env()
    assert torch.distributed.get_backend() == "nccl"
    load_preset("charades_r50_baseline", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1] + OV)
    inputs = om.synth_inputs(cfg, 1, "train", seed=2, crop=64, frames=16)
    params = om.synth_params(cfg, seed=2)
    eng = _run_replica(1, 0, 1, inputs, params)
    assert eng.comm is not None and len(eng.comm.buckets) > 5
    q.put({n: eng.fetch_grad(n) for n in ("pred_w", "res3_1_branch2b_w", "conv1_w")})
    dist.barrier()
    torch.distributed.destroy_process_group()


def test_rccl_bucketed_allreduce_one_rank():
    """the RCCL (backend "nccl") leg of GradComm on the one visible GPU: communicator set-up, weight
    broadcast, async bucket all-reduces issued during backward, solver-side wait.  A one-rank sum is
    the identity, so the gradients must equal those of a run without a process group."""
    import torch.multiprocessing as mp
    _setup_paths()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(port, q))
    p.start()
    got = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from oracle import model as om
    load_preset("charades_r50_baseline", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1] + OV)
    inputs = om.synth_inputs(cfg, 1, "train", seed=2, crop=64, frames=16)
    params = om.synth_params(cfg, seed=2)
    eng = _run_replica(1, 0, 1, inputs, params)
    assert eng.comm is None
    for n, g in got.items():
        assert np.array_equal(g, eng.fetch_grad(n)), n
