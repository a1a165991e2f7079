"""bench.py starts its own ranks: `python bench.py --gpus N` without a launcher re-executes itself under
torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous) and rank 0 prints the one JSON line.  On the one-GPU
box the path is driven with --launch and VLFB_DIST_FORCE=1, which makes the single rank a real RCCL job: communicator,
weight broadcast, bucketed all-reduce during backward (model_builder_video.py:142-157)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VLFB_BENCH_CHILD", "VLFB_DIST_FORCE"):
        env.pop(k, None)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--clips-per-gpu", "1", "--frames", "16",
           "--crop", "64", "--no-cpu-baseline", "--no-fp32-line", "--no-split-line", "--no-fp16-line"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    return r


def test_self_launched_one_rank_rccl_job_prints_one_json_line():
    r = _run(["--gpus", "1", "--launch"], {"VLFB_DIST_FORCE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    # the line is the gate-meeting path's: dtype mix, its dominant family (the two-plane fp16 NT kernels of the forward pass)
    # priced against the MFMA peak / 3 instructions per product, every family of the step listed
    assert out["dtype"] == "mix" and out["roofline"]["kernel"].startswith("gemm_nt_kernel<f16, PAIR>"), out["roofline"]
    assert abs(out["roofline"]["peak"] - 2500.0 / 3) < 0.1 and abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-3
    assert {"nt_pair", "nt_split", "nt_16", "tn_16", "tn_split"} <= set(out["roofline_families"]), sorted(out["roofline_families"])
    assert 1.0 < out["roofline_families"]["nt_16"]["mfma_per_product"] <= 2.0       # two-term DGRADs among the fp16 NT launches
    ar = out["allreduce"]
    assert ar["backend"] == "nccl" and ar["buckets"] >= 1 and ar["ranks"] == 1, ar
    # what makes an N > 1 line checkable: identical weights on every rank, the step without the exchange, the exchange alone
    assert ar["weights_bit_identical_across_ranks"] is True, ar
    assert ar["ms_per_step_without_allreduce"] > 0 and ar["allreduce_alone_ms"] > 0 and "exposed_allreduce_ms" in ar, ar


def test_job_size_mismatch_is_an_error_not_a_hang():
    """a rank count that differs from --gpus must fail before any GPU work (it used to be a bare assert)"""
    r = _run(["--gpus", "2"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


def test_two_rank_job_runs_the_n_gt_1_code_of_bench():
    """`python bench.py --gpus 2` end to end on the one-GPU box: both ranks on cuda:0 (VLFB_BENCH_ONE_DEVICE) over gloo --
    RCCL refuses two ranks on one device.  Not a measurement; it executes every world > 1 statement of bench.py
    (self-launch with two processes, per-rank seeds and RoI draws, broadcast of rank 0's weights, the bucketed sum
    all-reduce during backward, max-over-ranks timing, the data-parallel report) and checks what the line claims:
    2 ranks took part, twice the clips, and the ranks hold bit-identical weights after training on DIFFERENT batches"""
    r = _run(["--gpus", "2", "--mix-steps", "0"], {"VLFB_DIST_BACKEND": "gloo", "VLFB_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) >= 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "dp2"
    assert "global batch 2" in out["config"]["workload"]
    assert abs(out["value"] - 2 * 1 * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-2 * out["value"]
    ar = out["allreduce"]
    assert ar["backend"] == "gloo" and ar["ranks"] == 2 and ar["buckets"] >= 1
    assert ar["weights_bit_identical_across_ranks"] is True, ar
    assert ar["allreduce_alone_ms"] > 0 and ar["ms_per_step_without_allreduce"] > 0
