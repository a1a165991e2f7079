#!/usr/bin/env python
"""RCCL sum-all-reduce microbenchmark for the gradient payloads of the hot path:
the fp32 trainable-gradient buffer of ava_r50_lfb_nl (155.7 MB) in ~32 MB buckets, of charades_r50_baseline
(139.4 MB) and of ava_r101_lfb_nl_3l (254.5 MB), plus single buckets of 8 / 32 / 64 / 128 MB.  Reports the
time per pass, algorithmic bandwidth (bytes / s) and ring bus bandwidth (2 (n-1)/n x that).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 \\
           scratch/rccl_allreduce_bench.py [--iters 20]

Not run this round: no multi-GPU node is reachable from the builder's box (the driver's SCALE run is the
first time two ranks meet).  With one rank it only checks that the script and the RCCL communicator work."""
import argparse
import json
import os
import time

import torch
import torch.distributed as td


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bucket-mb", type=int, default=32)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    td.init_process_group("nccl")
    n, rank = td.get_world_size(), td.get_rank()
    payloads = [("ava_r50_lfb_nl grads, %d MB buckets" % args.bucket_mb, 155.7e6, args.bucket_mb << 20),
                ("charades_r50_baseline grads, %d MB buckets" % args.bucket_mb, 139.4e6, args.bucket_mb << 20),
                ("ava_r101_lfb_nl_3l grads, %d MB buckets" % args.bucket_mb, 254.5e6, args.bucket_mb << 20)]
    payloads += [("single bucket %d MB" % mb, mb * 2 ** 20, mb << 20) for mb in (8, 32, 64, 128)]
    out = []
    for name, nbytes, bucket in payloads:
        numel = int(nbytes) // 4
        buf = torch.ones(numel, device="cuda", dtype=torch.float32)
        per = max(bucket // 4, 1)
        chunks = [buf[i:i + per] for i in range(0, numel, per)]

        def one_pass():
            works = [td.all_reduce(c, op=td.ReduceOp.SUM, async_op=True) for c in chunks]
            for w in works:
                w.wait()
        for _ in range(3):
            one_pass()
        torch.cuda.synchronize()
        td.barrier()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            one_pass()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
        alg = numel * 4 / dt
        out.append({"payload": name, "bytes": numel * 4, "buckets": len(chunks), "ms": round(dt * 1e3, 3),
                    "algbw_GBps": round(alg / 1e9, 1), "busbw_GBps": round(alg * 2 * (n - 1) / n / 1e9, 1)})
    if rank == 0:
        print(json.dumps({"world_size": n, "results": out}))
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
