"""micro-benchmark of representative implicit-GEMM launches (bf16): prints TFLOP/s per shape"""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
from vlfb import hip
hip.lib()
dev = torch.device('cuda:0')
bf = torch.bfloat16
code = hip.BF16
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = sys.argv[3] if len(sys.argv) > 3 else ''


def geom(k, s, p, d):
    return dict(kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2], dt=d[0], dh=d[1], dw=d[2])


def run(name, N, Cin, Cout, T, H, W, k, s, p, d):
    if only and only not in name:
        return
    To, Ho, Wo = [(x + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for x, kk, ss, pp, dd in zip((T, H, W), k, s, p, d)]
    taps = k[0] * k[1] * k[2]
    x = torch.randn(N, T, H, W, Cin, device=dev).to(bf)
    w = (torch.randn(Cout, taps, Cin, device=dev) * 0.05).to(bf)
    wd = (torch.randn(Cin, taps, Cout, device=dev) * 0.05).to(bf)
    y = torch.empty(N, To, Ho, Wo, Cout, device=dev, dtype=bf)
    g = torch.randn(N, To, Ho, Wo, Cout, device=dev).to(bf)
    dx = torch.empty_like(x)
    dw = torch.empty(Cout, taps, Cin, device=dev, dtype=torch.float32)
    res = torch.randn(N, To, Ho, Wo, Cout, device=dev).to(bf)
    bias = torch.randn(Cout, device=dev)
    G = geom(k, s, p, d)
    df = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=code, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1, bias_mode=hip.BIAS_COL, **G)
    dd_ = hip.conv_desc(mode=hip.DGRAD, dtype=code, out_dtype=code, N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo, Cs=Cout, Cn=Cin, **G)
    dwd = hip.conv_desc(mode=hip.WGRAD, dtype=code, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, **G)
    ws = torch.empty(max(hip.conv_workspace_bytes(dwd), 16) // 4, device=dev, dtype=torch.float32)
    fl = 2.0 * N * To * Ho * Wo * Cout * taps * Cin
    fns = {'fprop': lambda: hip.conv_run(df, x, w, None, y, bias=bias, R=res),
           'dgrad': lambda: hip.conv_run(dd_, g, wd, None, dx, R=x, mask=x),
           'wgrad': lambda: hip.conv_run(dwd, x, None, g, dw, workspace=ws)}
    for mode, fn in fns.items():
        if which not in ('all', mode):
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print('%-14s %-6s %9.1f us %8.1f TFLOP/s' % (name, mode, us, fl / us / 1e6))


run('res5_2b', 8, 512, 512, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2))
run('res5_2c', 8, 512, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res4_2a_t3', 8, 1024, 256, 16, 14, 14, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1))
run('res3_2b', 8, 128, 128, 16, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))
run('res2_2b', 8, 64, 64, 32, 56, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1))
run('res2_2c', 8, 64, 256, 32, 56, 56, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))


def run_stem(splits_list=(0, 16, 24, 32, 40, 48, 64)):
    """conv1 wgrad (packed stem: Cin 3->4, kw 7->8, W-padded input) with forced split counts"""
    N, T, H, W, Cout = 8, 32, 224, 224, 64
    To, Ho, Wo = 32, 112, 112
    x = torch.randn(N, T, H, W + 8, 4, device=dev).to(bf)
    g = torch.randn(N, To, Ho, Wo, Cout, device=dev).to(bf)
    dw = torch.empty(Cout, 5, 7, 8, 4, device=dev, dtype=torch.float32)
    G = dict(kt=5, kh=7, kw=7, st=1, sh=2, sw=2, pt=2, ph=3, pw=3 - 4, dt=1, dh=1, dw=1)
    fl = 2.0 * N * To * Ho * Wo * Cout * 5 * 7 * 7 * 3
    for sp in splits_list:
        d = hip.conv_desc(mode=hip.WGRAD, dtype=code, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W + 8,
                          Cs=4, Cn=Cout, pack_w=8, splits=sp, **G)
        ws = torch.empty(max(hip.conv_workspace_bytes(d), 16) // 4, device=dev, dtype=torch.float32)
        fn = lambda: hip.conv_run(d, x, None, g, dw, workspace=ws)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print('stem wgrad splits=%-3d %9.1f us %8.1f TFLOP/s (algorithmic)' % (sp, us, fl / us / 1e6))


if which == 'stem':
    run_stem()
