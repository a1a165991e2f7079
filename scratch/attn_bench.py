"""fused attention-score kernels vs the composed path, model shapes (development timing)"""
import sys; sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
from vlfb import hip
hip.lib()
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, L1, L2, Ci in ((8, 3136, 784, 256), (32, 3136, 784, 256), (8, 4096, 1024, 256)):
    th = torch.randn(B, L1, Ci, device=dev).bfloat16(); ph = torch.randn(B, L2, Ci, device=dev).bfloat16()
    g = torch.randn(B, L2, Ci, device=dev).bfloat16(); dY = torch.randn(B, L1, Ci, device=dev).bfloat16()
    P = torch.empty(B, L1, L2, device=dev, dtype=torch.bfloat16); dS = torch.empty_like(P)
    S = torch.empty(B, L1, L2, device=dev, dtype=torch.float32)
    sc = Ci ** -0.5
    d = hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1, Ws=L1, batch=B,
                      Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2)
    def unf_f():
        hip.conv_run(d, th, ph, None, S); hip.call("vlfb_softmax_fwd", hip.ptr(S), hip.ptr(P), hip.BF16, B * L1, L2, sc)
    def fus_f():
        hip.call("vlfb_attn_scores_fwd", hip.ptr(th), hip.ptr(ph), hip.ptr(P), hip.BF16, B, L1, L2, Ci, sc)
    def unf_b():
        hip.conv_run(d, dY, g, None, S); hip.call("vlfb_softmax_bwd", hip.ptr(S), hip.ptr(P), hip.ptr(dS), hip.BF16, B * L1, L2, sc)
    def fus_b():
        hip.call("vlfb_attn_scores_bwd", hip.ptr(dY), hip.ptr(g), hip.ptr(P), hip.ptr(dS), hip.BF16, B, L1, L2, Ci, sc)
    unf_f()
    print("B%d L1 %d L2 %d Ci %d: fwd composed %.1f us fused %.1f us | bwd composed %.1f us fused %.1f us" % (
        B, L1, L2, Ci, timeit(unf_f), timeit(fus_f), timeit(unf_b), timeit(fus_b)))
