"""r6 probe (TIMING ONLY -- the results of the experimental library are garbage): what would the two-term fp16 DGRAD gain if
the two terms of a tap (Wh, Wl) shared ONE activation tile in LDS?  scratch/r6/libvlfb_exp.so is the library with the A-tile
DMA of every odd k-tile skipped when VLFB_SKIPA=1 (an upper bound for the DMA side of an "A-reuse" kernel: the LDS reads and
the MFMAs stay).  Run twice: VLFB_SKIPA=0 / 1."""
import os
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
from vlfb import hip
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvlfb_exp.so')
hip.lib()
dev = torch.device('cuda:0')


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(name, N, Cin, Cout, T, H, W, k, s, p, dl):
    """DGRAD of conv (Cin -> Cout) as the mix path launches it: fp16, doubled outermost tap (kt = 2, dt = 0)"""
    To, Ho, Wo = [(x + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for x, kk, ss, pp, dd in zip((T, H, W), k, s, p, dl)]
    taps = k[0] * k[1] * k[2]
    dy = torch.randn(N, To, Ho, Wo, Cout, device=dev).half()
    wd = (torch.randn(Cin, 2 * taps, Cout, device=dev) * 0.05).half()
    dx = torch.empty(N, T, H, W, Cin, device=dev, dtype=torch.float16)
    assert k[0] == 1
    G = dict(kt=2, kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=0, ph=p[1], pw=p[2], dt=0, dh=dl[1], dw=dl[2])
    d = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo, Cs=Cout, Cn=Cin, **G)
    fl = 2.0 * N * To * Ho * Wo * Cout * taps * Cin * 2
    t = timeit(lambda: hip.conv_run(d, dy, wd, None, dx))
    print('%-26s %-26s %8.1f us  %6.1f TF (both terms)' % (name, hip.conv_plan(d), t, fl / t / 1e6))
    if 'nt8' in hip.conv_plan(d) or '256' in hip.conv_plan(d):
        d.algo = hip.ALGO_TILE128
        t = timeit(lambda: hip.conv_run(d, dy, wd, None, dx))
        print('%-26s %-26s %8.1f us  %6.1f TF (both terms)' % ('', hip.conv_plan(d), t, fl / t / 1e6))


print('VLFB_SKIPA =', os.environ.get('VLFB_SKIPA', '0'))
N = 8
run('res2 1x1 64->256 dgrad', N, 64, 256, 32, 56, 56, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res2 3x3 64->64 dgrad', N, 64, 64, 32, 56, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))
run('res3 1x1 128->512 dgrad', N, 128, 512, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res3 1x1 512->128 dgrad', N, 512, 128, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res3 3x3 128->128 dgrad', N, 128, 128, 16, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))
run('res4 1x1 256->1024 dgrad', N, 256, 1024, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res4 1x1 1024->256 dgrad', N, 1024, 256, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res4 3x3 256->256 dgrad', N, 256, 256, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))
run('res5 1x1 512->2048 dgrad', N, 512, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res5 1x1 2048->512 dgrad', N, 2048, 512, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res5 3x3 d2 512->512 dgrad', N, 512, 512, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2))
