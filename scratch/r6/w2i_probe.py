"""r6 probe: the two-term fp16 DGRAD launches of the mix step, doubled-tap form (VLFB_MIX_W2: kt' = 2 kt, dt = 0) against the
interleaved form (VLFB_MATH_F16W2: one gradient tile per pair of weight tiles), each alone on the device."""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
import os
from vlfb import hip
if os.environ.get('VLFB_LIB'):
    hip.LIB_PATH = os.path.abspath(os.environ['VLFB_LIB'])      # (an experimental build of the library)
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


def run(name, N, Cin, Cout, T, H, W, k, p, dl, epi=False):
    To, Ho, Wo = [(x + 2 * pp - dd * (kk - 1) - 1) + 1 for x, kk, pp, dd in zip((T, H, W), k, p, dl)]
    taps = k[0] * k[1] * k[2]
    dy = torch.randn(N, To, Ho, Wo, Cout, device=dev).half()
    wd = (torch.randn(Cin, 2 * taps, Cout, device=dev) * 0.05).half()
    dx = torch.empty(N, T, H, W, Cin, device=dev, dtype=torch.float16)
    R = torch.randn(N, T, H, W, Cin, device=dev).half() if epi else None
    Mk = torch.randn(N, T, H, W, Cin, device=dev).half() if epi else None
    g1 = dict(kt=k[0], kh=k[1], kw=k[2], pt=p[0], ph=p[1], pw=p[2], dt=dl[0], dh=dl[1], dw=dl[2])
    rows = dict(N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo, Cs=Cout, Cn=Cin)
    if k[0] == 1:
        g2, rows2 = dict(g1, kt=2, dt=0), rows
    else:
        g2 = dict(kt=2, kh=k[0], kw=1, pt=0, ph=p[0], pw=0, dt=0, dh=dl[0], dw=1)
        rows2 = dict(N=N, Tr=1, Hr=T, Wr=H * W, Ts=1, Hs=To, Ws=Ho * Wo, Cs=Cout, Cn=Cin)
    d2 = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, **rows2, **g2)
    di = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16W2, **rows, **g1)
    fl = 2.0 * N * To * Ho * Wo * Cout * taps * Cin * 2
    t2 = timeit(lambda: hip.conv_run(d2, dy, wd, None, dx, R=R, mask=Mk))
    ti = timeit(lambda: hip.conv_run(di, dy, wd, None, dx, R=R, mask=Mk))
    print('%-34s %-26s %8.1f us %6.1f TF | %-24s %8.1f us %6.1f TF | %.2fx' % (
        name + (' +R+mask' if epi else ''), hip.conv_plan(d2), t2, fl / t2 / 1e6, hip.conv_plan(di), ti, fl / ti / 1e6, t2 / ti))


N = 8
one, z = (1, 1, 1), (0, 0, 0)
for epi in (False, True):
    run('res2 1x1 64<-256', N, 64, 256, 32, 56, 56, one, z, one, epi)
    run('res2 3x1x1 256<-64', N, 256, 64, 32, 56, 56, (3, 1, 1), (1, 0, 0), one, epi)
    run('res2 3x3 64<-64', N, 64, 64, 32, 56, 56, (1, 3, 3), (0, 1, 1), one, epi)
    run('res3 1x1 128<-512', N, 128, 512, 16, 28, 28, one, z, one, epi)
    run('res3 3x1x1 512<-128', N, 512, 128, 16, 28, 28, (3, 1, 1), (1, 0, 0), one, epi)
    run('res3 1x1 512<-128', N, 512, 128, 16, 28, 28, one, z, one, epi)
    run('res3 3x3 128<-128', N, 128, 128, 16, 28, 28, (1, 3, 3), (0, 1, 1), one, epi)
    run('res4 1x1 256<-1024', N, 256, 1024, 16, 14, 14, one, z, one, epi)
    run('res4 3x1x1 1024<-256', N, 1024, 256, 16, 14, 14, (3, 1, 1), (1, 0, 0), one, epi)
    run('res4 1x1 1024<-256', N, 1024, 256, 16, 14, 14, one, z, one, epi)
    run('res4 3x3 256<-256', N, 256, 256, 16, 14, 14, (1, 3, 3), (0, 1, 1), one, epi)
    run('res5 1x1 512<-2048', N, 512, 2048, 16, 14, 14, one, z, one, epi)
    run('res5 3x1x1 2048<-512', N, 2048, 512, 16, 14, 14, (3, 1, 1), (1, 0, 0), one, epi)
    run('res5 1x1 2048<-512', N, 2048, 512, 16, 14, 14, one, z, one, epi)
    run('res5 1x1 1024<-512 (res5_0 2a)', N, 1024, 512, 16, 14, 14, one, z, one, epi)
    run('nl 1x1 512<-256 (theta), 4x28x28', N, 512, 256, 4, 28, 28, one, z, one, epi)
