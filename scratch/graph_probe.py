"""Development probe: GPU time per tiny dependent kernel, stream launches vs a captured HIP graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"))
import torch
from vlfb import hip
hip.lib()
x = torch.zeros(4096, device="cuda", dtype=torch.float32)
N = 300
def chain():
    for _ in range(N):
        hip.call("vlfb_zero_f32", hip.ptr(x), 4096)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); chain(); e1.record(); torch.cuda.synchronize()
    print("stream launches: %.2f us per kernel (host-bound or GPU-bound)" % (e0.elapsed_time(e1) * 1e3 / N))
    # pre-enqueue behind a long kernel so the host is not the limit
    big = torch.empty(256 << 20, device="cuda", dtype=torch.float32)
    e0.record()
    for _ in range(4): big.zero_()
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    chain(); e1.record(); torch.cuda.synchronize()
    print("stream launches queued behind work: %.2f us per kernel" % (e2.elapsed_time(e1) * 1e3 / N))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("graph replay: %.2f us per kernel" % (e0.elapsed_time(e1) * 1e3 / N))
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("graph replay host time: %.1f us for %d kernels" % ((t1 - t0) * 1e6, N))
