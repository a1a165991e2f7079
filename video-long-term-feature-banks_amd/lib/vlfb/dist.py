"""Process-group helpers: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  Replaces the reference's single-process
data_parallel_model.Parallelize_GPU(use_nccl=...) (lib/models/model_builder_video.py:142-157)."""
import os

import torch
import torch.distributed as td


def initialized():
    return td.is_available() and td.is_initialized()


def world_size():
    return td.get_world_size() if initialized() else 1


def rank():
    return td.get_rank() if initialized() else 0


def local_rank():
    """index of this process's GPU.  VLFB_FORCE_DEVICE pins every rank to one device (used by the
    single-GPU multi-process test, which then has to run over gloo)."""
    if "VLFB_FORCE_DEVICE" in os.environ:
        return int(os.environ["VLFB_FORCE_DEVICE"])
    return int(os.environ.get("LOCAL_RANK", "0"))


def forced():
    return os.environ.get("VLFB_DIST_FORCE", "0") == "1"


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).
    A one-rank job skips the process group unless VLFB_DIST_FORCE=1 (lets a single-GPU box drive the
    RCCL code path: communicator set-up, bucketed all-reduce, stream hand-over)."""
    if initialized():
        return
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not forced():
        return
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("VLFB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
    td.init_process_group(backend=backend)


def barrier():
    if initialized():
        td.barrier()


def all_ok(flag):
    """logical AND of `flag` over all ranks (a barrier that also carries a success bit); plain `flag` without a group"""
    if not initialized():
        return bool(flag)
    dev = "cuda:%d" % local_rank() if td.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], device=dev, dtype=torch.int32)
    td.all_reduce(t, op=td.ReduceOp.MIN)
    return bool(int(t.item()))
