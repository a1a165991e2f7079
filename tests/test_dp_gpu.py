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


def _collect(q, procs, n, timeout=300):
    out = []
    for _ in range(n):
        item = q.get(timeout=timeout)
        if isinstance(item, tuple) and isinstance(item[1], dict) and "_error" in item[1]:
            for p in procs:
                p.kill()
            pytest.fail("worker failed:\n" + item[1]["_error"])
        out.append(item)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:
        import traceback
        q.put((-1, {"_error": traceback.format_exc()}))
        raise


def _worker_body(rank, world, port, q):
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
    results = dict(_collect(q, procs, 2))
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


def _nccl_worker(port, q, handoff):
    try:
        _nccl_worker_body(port, q, handoff)
    except BaseException:
        import traceback
        q.put((-1, {"_error": traceback.format_exc()}))
        raise


def _nccl_worker_body(port, q, handoff):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      VLFB_DIST_FORCE="1", VLFB_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from vlfb import dist
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from oracle import model as om
    from vlfb.engine import Engine
    Engine.BUCKET_HANDOFF = handoff
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


@pytest.mark.parametrize("handoff", ["streams", "join"])
def test_rccl_bucketed_allreduce_one_rank(handoff):
    """the RCCL (backend "nccl") leg of GradComm on the one visible GPU: communicator set-up, weight
    broadcast, async bucket all-reduces issued during backward, solver-side wait -- with the bucket hand-off on the
    third stream (two events, no compute stream waits) and with the conservative main-stream join
    (Engine.BUCKET_HANDOFF).  A one-rank sum is the identity, so the gradients must equal those of a run without a
    process group."""
    import torch.multiprocessing as mp
    _setup_paths()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(port, q, handoff))
    p.start()
    got = _collect(q, [p], 1)[0]
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


def _replay_worker(port, q, handoff):
    try:
        _setup_paths()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                          VLFB_DIST_FORCE="1", VLFB_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import time
        from vlfb import dist
        from vlfb.presets import load_preset
        from core.config import config as cfg
        from oracle import model as om
        from vlfb.engine import Engine
        Engine.BUCKET_HANDOFF = handoff
        dist.init_from_env()
        load_preset("charades_r50_baseline", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1] + OV)
        inputs = om.synth_inputs(cfg, 1, "train", seed=2, crop=64, frames=16)
        params = om.synth_params(cfg, seed=2)
        out = {}
        for trace in (True, False):
            eng = _run_replica(1, 0, 1, inputs, params)       # (plans, feeds, enables data parallel; one forward / backward)
            assert eng.comm is not None and len(eng.comm.buckets) > 5
            eng.STEP_TRACE = trace
            host = []
            for it in range(5):
                t0 = time.perf_counter()
                eng.train_step(0.01)
                host.append(time.perf_counter() - t0)
                torch.cuda.synchronize()
            names = [n for _, _, n in (eng._trace or [])]
            out[trace] = {"w": {n: eng.fetch_param(n) for n in ("pred_w", "res3_1_branch2b_w", "conv1_w")}, "loss": float(eng.fetch("loss").reshape(-1)[0]),
                          "recorded": len(names), "reductions": names.count("all-reduce buckets"), "waits": names.count("comm wait"),
                          "host_ms": [round(h * 1e3, 2) for h in host]}
            del eng
        q.put(out)
        dist.barrier()
        torch.distributed.destroy_process_group()
    except BaseException:
        import traceback
        q.put((-1, {"_error": traceback.format_exc()}))
        raise


@pytest.mark.parametrize("handoff", ["streams", "join"])
def test_data_parallel_step_replays_from_the_recorded_call_list(handoff):
    """Engine.STEP_TRACE on a data-parallel step (model_builder_video.py:126-157): the bucket all-reduces and the communicator's
    bookkeeping are recorded host actions (Engine.traced), a replayed step re-issues them between the same launches -- five
    steps over RCCL (one rank) end with the weights of five steps walked through the step objects, bit for bit, and the
    replayed steps cost the host less than the recorded one."""
    import torch.multiprocessing as mp
    _setup_paths()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_replay_worker, args=(port, q, handoff))
    p.start()
    got = _collect(q, [p], 1)[0]
    a, b = got[True], got[False]
    assert a["recorded"] > 100 and b["recorded"] == 0
    assert a["reductions"] >= 1 and a["waits"] == 1, a
    assert a["loss"] == b["loss"]
    for n in a["w"]:
        assert np.array_equal(a["w"][n], b["w"][n]), n
    print("\n[dp step, %s] host ms per step: recorded list %s | step objects %s" % (handoff, a["host_ms"], b["host_ms"]))
    assert min(a["host_ms"][2:]) < min(b["host_ms"][2:])


# ---- AVA: a different number of RoIs on every rank ---------------------------------------------------------
AVA_OV = ["TRAIN.VIDEO_LENGTH", 16, "TRAIN.CROP_SIZE", 64]
ROIS = {0: [1], 1: [4]}
CHECK = ("pred_w", "lfb_nl0_theta_w", "res5_2_branch2c_w", "nonlocal_conv4_1_out_w", "conv1_w")


def _ava_rank_inputs(cfg, rank):
    from oracle import model as om
    return om.synth_inputs(cfg, 1, "train", seed=20 + rank, rois_per_clip=ROIS[rank], crop=64, frames=16)


def _ava_worker(rank, world, port, q, dtype="fp32"):
    try:
        _ava_worker_body(rank, world, port, q, dtype)
    except BaseException:                 # the parent must hear about it at once, not after its queue time-out
        import traceback
        q.put((rank, {"_error": traceback.format_exc()}))
        raise


def _ava_worker_body(rank, world, port, q, dtype):
    import collections
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), VLFB_FORCE_DEVICE="0", VLFB_DIST_BACKEND="gloo")
    from vlfb import dist
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from oracle import model as om
    dist.init_from_env()
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", world, "TRAIN.BATCH_SIZE", world] + AVA_OV)
    inputs = _ava_rank_inputs(cfg, rank)
    params = om.synth_params(cfg, seed=2)
    model = ModelBuilder(train=True, split="train", name="dp")
    model.build_model(suffix="_train")
    eng = Engine(model, dtype, device="cuda:0", base_seed=2)
    assert eng.replica == rank
    eng.plan(collections.OrderedDict((k + "_train", v.shape) for k, v in inputs.items()))
    eng.feed_params(params)
    for k, v in inputs.items():
        eng.feed(k + "_train", v)
    eng.enable_data_parallel(bucket_mb=4)
    eng.forward()
    eng.backward()
    eng.comm.wait()
    torch.cuda.synchronize()
    drop = [s for s in eng.steps if type(s).__name__ == "DropoutStep" and s.out.name == "pool5_dropout"][0]
    out = {n: eng.fetch_grad(n) for n in CHECK}
    out["_mask_row0"] = drop.mask.view(-1)[:drop.ch * drop.inner].cpu().numpy().copy()
    out["_loss"] = float(eng.fetch("loss").reshape(-1)[0])
    out["_loss_scale"] = float(eng.loss_scale)
    q.put((rank, out))
    dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "fp16", "mix"])
def test_ranks_with_different_roi_counts_sum_their_per_gpu_normalised_losses(dtype):
    """Each GPU normalises its loss by ITS OWN number of valid targets and scales by 1/NUM_GPUS
    (resnet_video.py:333-338), so with 1 RoI on rank 0 and 4 on rank 1 the all-reduced gradient is the SUM
    of the two per-rank oracle gradients -- not the gradient of one 5-RoI batch.  Also: the replicas draw
    different dropout masks (one RNG per GPU in the reference).
    fp16 / mix: the summed gradients carry the fp16 loss scale, so every rank has to choose the SAME scale although it
    holds a different number of RoI rows (it used to be sized from the rows: 2^13 on the 1-RoI rank, 2^15 on the 4-RoI one,
    and the ranks' weights drifted apart)."""
    import torch.multiprocessing as mp
    _setup_paths()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ava_worker, args=(r, 2, port, q, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(_collect(q, procs, 2))
    assert results[0]["_loss_scale"] == results[1]["_loss_scale"]
    assert (results[0]["_loss_scale"] == 1.0) == (dtype == "fp32")
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from vlfb import rng as vrng
    from oracle import model as om
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 2, "TRAIN.BATCH_SIZE", 2] + AVA_OV)
    params = om.synth_params(cfg, seed=2)
    total, losses = None, []
    for rank in range(2):
        blobs, grads = om.run(cfg, params, _ava_rank_inputs(cfg, rank), "train", torch.float64, True,
                              lambda name, r=rank: vrng.dropout_seed(2, name, 0, r), num_gpus=2)
        losses.append(float(blobs["loss"].detach()))
        total = {n: g.clone() for n, g in grads.items()} if total is None else {n: total[n] + grads[n] for n in total}
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for rank in range(2):
        assert abs(results[rank]["_loss"] - losses[rank]) < {"fp32": 1e-4, "mix": 1e-4, "fp16": 2e-3}[dtype] * abs(losses[rank]), \
            (rank, results[rank]["_loss"], losses[rank])
    tol = {"fp32": 5e-3, "mix": 1e-2, "fp16": 1e-1}[dtype]     # (the bars of test_model_gpu for each path at this size)
    for n in CHECK:
        assert np.array_equal(results[0][n], results[1][n]), "ranks disagree after all-reduce: " + n
        assert rel(results[0][n], total[n].numpy()) < tol, (n, rel(results[0][n], total[n].numpy()))
    assert not np.array_equal(results[0]["_mask_row0"], results[1]["_mask_row0"]), "replicas share a dropout mask"


def test_data_parallel_refuses_a_job_size_that_differs_from_num_gpus():
    """enable_data_parallel: the loss scale and the per-GPU batch come from cfg.NUM_GPUS"""
    import collections
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      VLFB_DIST_FORCE="1", VLFB_DIST_BACKEND="gloo")
    from vlfb import dist, hip
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from oracle import model as om
    try:
        dist.init_from_env()
        load_preset("charades_r50_baseline", ["NUM_GPUS", 2, "TRAIN.BATCH_SIZE", 2] + OV)   # 2 GPUs configured, 1 rank
        model = ModelBuilder(train=True, split="train", name="dp")
        model.build_model(suffix="_train")
        eng = Engine(model, "fp32", device="cuda:0", base_seed=2)
        inputs = om.synth_inputs(cfg, 1, "train", seed=2, crop=64, frames=16)
        eng.plan(collections.OrderedDict((k + "_train", v.shape) for k, v in inputs.items()))
        with pytest.raises(hip.VlfbError):
            eng.enable_data_parallel()
    finally:
        for k in ("VLFB_DIST_FORCE", "VLFB_DIST_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
