"""r6 probe: forward launches of the `mix` step: split-bf16 on fp32 storage (round 5) vs two fp16 planes (F16X3) vs plain fp16"""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
import os
from vlfb import hip
if os.environ.get('VLFB_LIB'):
    hip.LIB_PATH = os.path.abspath(os.environ['VLFB_LIB'])      # (an experimental build of the library)
hip.lib()
dev = torch.device('cuda:0')


def geom(k, s, p, dl):
    return dict(kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2], dt=dl[0], dh=dl[1], dw=dl[2])


def timeit(fn, reps=20):
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


def run(name, N, Cin, Cout, T, H, W, k, s, p, dl, res=True, algo=0):
    To, Ho, Wo = [(x + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for x, kk, ss, pp, dd in zip((T, H, W), k, s, p, dl)]
    taps = k[0] * k[1] * k[2]
    x = torch.randn(N, T, H, W, Cin, device=dev)
    wf = (torch.randn(Cout, taps, Cin, device=dev) * 0.05)
    wp = torch.empty(3, Cout, taps, Cin, device=dev, dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", wf.data_ptr(), None, wp.data_ptr(), None, hip.SPLIT, Cout, taps, Cin)
    wh = torch.empty(2, Cout, taps, Cin, device=dev, dtype=torch.float16)
    hip.call("vlfb_weight_prep", wf.data_ptr(), None, wh.data_ptr(), None, hip.MIXH, Cout, taps, Cin)
    n_out = N * To * Ho * Wo * Cout
    y = torch.empty(n_out, device=dev)
    yh = torch.empty(n_out, device=dev, dtype=torch.float16)
    r = torch.randn(n_out, device=dev) if res else None
    bias = torch.randn(Cout, device=dev)
    G = geom(k, s, p, dl)
    base = dict(mode=hip.FPROP, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1, bias_mode=hip.BIAS_COL, **G)
    d_sp = hip.conv_desc(dtype=hip.F32, out_dtype=hip.F32, o_planes=1, math=hip.MATH_BF16X3, b_pstride=wf.numel(), **base)
    fl = 2.0 * n_out * taps * Cin
    t_sp = timeit(lambda: hip.conv_run(d_sp, x, wp, None, y, bias=bias, R=r, O_planes=yh))
    xp = torch.empty(2 * x.numel(), device=dev, dtype=torch.float16)
    hip.call("vlfb_pair_split", x.data_ptr(), xp.data_ptr(), x.numel())
    rp = None
    if res:
        rp = torch.empty(2 * n_out, device=dev, dtype=torch.float16)
        hip.call("vlfb_pair_split", r.data_ptr(), rp.data_ptr(), n_out)
    yp = torch.empty(2 * n_out, device=dev, dtype=torch.float16)
    d_h2 = hip.conv_desc(dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16X3, a_pstride=x.numel(), b_pstride=wf.numel(), alpha=1.0 / 1024, algo=algo, **base)
    d_h2 = hip.conv_desc(dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16X3, a_pstride=x.numel(), b_pstride=wf.numel(), alpha=1.0 / 1024, algo=hip.ALGO_TILE128, **base)
    t_h2 = timeit(lambda: hip.conv_run(d_h2, xp, wh, None, yp, bias=bias, R=rp, R_lo=rp[n_out:] if res else None, O_lo=yp[n_out:]))
    t_h8, same = float('nan'), None
    try:
        d_h8 = hip.conv_desc(dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16X3, a_pstride=x.numel(), b_pstride=wf.numel(), alpha=1.0 / 1024, algo=hip.ALGO_PIPE256, **base)
        yq = torch.empty(2 * n_out, device=dev, dtype=torch.float16)
        t_h8 = timeit(lambda: hip.conv_run(d_h8, xp, wh, None, yq, bias=bias, R=rp, R_lo=rp[n_out:] if res else None, O_lo=yq[n_out:]))
        torch.cuda.synchronize()
        same = torch.equal(yp, yq)
        plan8 = hip.conv_plan(d_h8)
    except hip.VlfbError:
        plan8 = '-'
    xh, w16 = x.half(), wf.half()
    rh = r.half() if res else None
    d_16 = hip.conv_desc(dtype=hip.F16, out_dtype=hip.F16, **base)
    t_16 = timeit(lambda: hip.conv_run(d_16, xh, w16, None, yh, bias=bias, R=rh))
    print('%-24s split %8.1f us %6.1f TF | pair %8.1f us %6.1f TF | pair 8-phase %8.1f us %6.1f TF (%s, ==%s) | fp16 %7.1f us (%s)' % (
        name, t_sp, fl / t_sp / 1e6, t_h2, fl / t_h2 / 1e6, t_h8, fl / t_h8 / 1e6, plan8, same, t_16, hip.conv_plan(d_16)))
    return t_sp, t_h2, min(t_h2, t_h8) if t_h8 == t_h8 else t_h2


C = 8
tot = [0.0, 0.0, 0.0]
# (calls per step, shape): the forward conv launches of ava_r50_lfb_nl at 8 clips
shapes = [
    (4, ('res2 2c/br1 64->256', C, 64, 256, 32, 56, 56, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))),
    (2, ('res2 2a 256->64 k311', C, 256, 64, 32, 56, 56, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (1, ('res2 2a 64->64 k311', C, 64, 64, 32, 56, 56, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (3, ('res2 2b 64->64 k133', C, 64, 64, 32, 56, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), False)),
    (1, ('res3_0 2a 256->128 k311', C, 256, 128, 16, 56, 56, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (1, ('res3_0 2b s2', C, 128, 128, 16, 56, 56, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1), False)),
    (1, ('res3_0 br1 s2', C, 256, 512, 16, 56, 56, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1), False)),
    (4, ('res3 2c 128->512', C, 128, 512, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))),
    (2, ('res3 2a 512->128', C, 512, 128, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
    (1, ('res3 2a 512->128 k311', C, 512, 128, 16, 28, 28, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (3, ('res3 2b 128 k133', C, 128, 128, 16, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), False)),
    (2, ('nl3 theta 512->256', C, 512, 256, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
    (1, ('res4_0 2a 512->256 k311', C, 512, 256, 16, 28, 28, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (6, ('res4 2c 256->1024', C, 256, 1024, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))),
    (3, ('res4 2a 1024->256', C, 1024, 256, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
    (2, ('res4 2a 1024->256 k311', C, 1024, 256, 16, 14, 14, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (5, ('res4 2b 256 k133', C, 256, 256, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), False)),
    (3, ('nl4 theta 1024->512', C, 1024, 512, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
    (1, ('res5 2a 1024->512', C, 1024, 512, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
    (3, ('res5 2c 512->2048', C, 512, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))),
    (1, ('res5 2a 2048->512', C, 2048, 512, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
    (1, ('res5 2a 2048->512 k311', C, 2048, 512, 16, 14, 14, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), False)),
    (3, ('res5 2b 512 k133 d2', C, 512, 512, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2), False)),
    (1, ('res5 br1 1024->2048', C, 1024, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), False)),
]
for calls, a in shapes:
    t = run(*a)
    for i in range(3):
        tot[i] += calls * t[i]
print('forward conv time per step over these launches: split %.2f ms | pair (128-row) %.2f ms | pair (best of both) %.2f ms' % tuple(v / 1e3 for v in tot))
