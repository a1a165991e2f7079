"""Gradient all-reduce for clip-level data parallelism.

The reference lets Caffe2 emit one NCCLAllreduce per gradient blob after the whole backward
(lib/models/model_builder_video.py:142-157, use_nccl).  Here the trainable gradients live in ONE
flat fp32 buffer laid out in backward-completion order, cut into ~32 MB buckets; as soon as the
last gradient of a bucket has been produced its sum-all-reduce is issued asynchronously
(ProcessGroupNCCL = RCCL runs it on its own HIP stream behind an event, so it overlaps the rest
of backward over xGMI), and the solver waits for all buckets.  The loss is already scaled by
1/NUM_GPUS on every rank (resnet_video.py:333-338), so the reduction is a plain SUM.

Backend-agnostic on purpose: the CPU tests drive the same class over `gloo`.
"""
import torch.distributed as td


def make_buckets(segments, bucket_bytes, esize=4):
    """cut [(offset, count, ready_step)] (flat order; ready_step = backward step after which the segment is final, roughly
    non-decreasing) into [(start, end, ready_step)] buckets of at least `bucket_bytes` (the last one may be smaller).  A bucket's
    ready step is the MAX over its segments and buckets are issued in order, so a segment that becomes final a few steps
    after its flat neighbours (the strided shortcuts the engine re-orders) only delays its own bucket."""
    buckets = []
    start, end, ready = None, 0, -1
    for off, cnt, step in segments:
        if start is None:
            start = off
        end = max(end, off + cnt)
        ready = max(ready, step)
        if (end - start) * esize >= bucket_bytes:
            buckets.append((start, end, ready))
            start = None
    if start is not None:
        buckets.append((start, end, ready))
    return buckets


class GradComm(object):
    def __init__(self, flat_grad, segments, bucket_bytes=32 << 20, group=None, buckets=None):
        """segments: [(offset, count, ready_step)] in flat order; ready_step = index of the backward
        step after which that segment's gradient is final (see make_buckets).  `buckets` overrides the cut
        (the engine's per-bucket solver uses the same buckets)."""
        self.flat = flat_grad
        self.group = group
        self.buckets = list(buckets) if buckets is not None else make_buckets(segments, bucket_bytes, flat_grad.element_size())
        self._next = 0
        self._works = []

    def begin(self):
        self._next = 0
        self._works = []

    def due(self, step_index):
        """True when at least one bucket becomes ready after backward step `step_index`"""
        return self._next < len(self.buckets) and self.buckets[self._next][2] <= step_index

    def after_step(self, step_index):
        """call after backward step `step_index` has been enqueued; returns the reductions issued by this call"""
        issued = []
        while self._next < len(self.buckets) and self.buckets[self._next][2] <= step_index:
            s, e, _ = self.buckets[self._next]
            issued.append(td.all_reduce(self.flat[s:e], op=td.ReduceOp.SUM, group=self.group, async_op=True))
            self._next += 1
        self._works.extend(issued)
        return issued

    def wait(self):
        """flush buckets not yet issued, then make the current stream wait for every reduction"""
        self.after_step(1 << 60)
        for w in self._works:
            w.wait()
        self._works = []
