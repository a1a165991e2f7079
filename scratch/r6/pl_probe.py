"""r6 probe: (1) does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs?  (2) forward launches of the `mix` step as the
split kernel runs them today (fp32 A split in the loop, fp32 output + fp16 copy) vs the plane kernel (A as two 16-bit
planes, no VALU in the loop) -- the kernel structure a two-plane fp16 activation format would run on."""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
from vlfb import hip
hip.lib()
dev = torch.device('cuda:0')

# ---- (1) subnormal operands of the fp16 MFMA -------------------------------------------------
M, K, N = 128, 64, 128
a = torch.full((M, K), 2.0 ** -20, device=dev, dtype=torch.float16)        # subnormal in fp16 (min normal 2^-14)
w = torch.full((N, K), 1.0, device=dev, dtype=torch.float16)
o = torch.empty(M, N, device=dev, dtype=torch.float32)
d = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F32, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=K, Cn=N, algo=hip.ALGO_TILE128)
hip.conv_run(d, a, w, None, o)
torch.cuda.synchronize()
print('fp16 MFMA, A = 2^-20 (subnormal) x W = 1, K = 64: out = %g (exact %g)' % (o[0, 0].item(), 64 * 2.0 ** -20))
a2 = torch.full((M, K), 2.0 ** -24, device=dev, dtype=torch.float16)
hip.conv_run(d, a2, w, None, o)
torch.cuda.synchronize()
print('fp16 MFMA, A = 2^-24 (smallest subnormal): out = %g (exact %g)' % (o[0, 0].item(), 64 * 2.0 ** -24))
w2 = torch.full((N, K), 2.0 ** -20, device=dev, dtype=torch.float16)
a3 = torch.full((M, K), 1024.0, device=dev, dtype=torch.float16)
hip.conv_run(d, a3, w2, None, o)
torch.cuda.synchronize()
print('fp16 MFMA, W = 2^-20 subnormal x A = 1024: out = %g (exact %g)' % (o[0, 0].item(), 64 * 1024 * 2.0 ** -20))


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


def run(name, N, Cin, Cout, T, H, W, k, s, p, dl, res=True):
    To, Ho, Wo = [(x + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for x, kk, ss, pp, dd in zip((T, H, W), k, s, p, dl)]
    taps = k[0] * k[1] * k[2]
    x = torch.randn(N, T, H, W, Cin, device=dev)
    xp = torch.empty(2, N, T, H, W, Cin, device=dev, dtype=torch.bfloat16)
    hip.call("vlfb_split_planes", x.data_ptr(), xp.data_ptr(), 2, 1, x.numel() // 8, 8, 0)
    wf = (torch.randn(Cout, taps, Cin, device=dev) * 0.05)
    wp = torch.empty(3, Cout, taps, Cin, device=dev, dtype=torch.bfloat16)
    hip.call("vlfb_split_planes", wf.data_ptr(), wp.data_ptr(), 3, 1, wf.numel() // 8, 8, 0)
    y = torch.empty(N, To, Ho, Wo, Cout, device=dev)
    yh = torch.empty(N, To, Ho, Wo, Cout, device=dev, dtype=torch.float16)
    r = torch.randn(N, To, Ho, Wo, Cout, device=dev) if res else None
    bias = torch.randn(Cout, device=dev)
    G = geom(k, s, p, dl)
    base = dict(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1,
                bias_mode=hip.BIAS_COL, math=hip.MATH_BF16X3, b_pstride=wf.numel(), **G)
    d_sp = hip.conv_desc(o_planes=1, **base)
    d_pl = hip.conv_desc(a_planes=2, a_pstride=x.numel(), **base)
    fl = 2.0 * N * To * Ho * Wo * Cout * taps * Cin
    t_sp = timeit(lambda: hip.conv_run(d_sp, x, wp, None, y, bias=bias, R=r, O_planes=yh))
    try:
        t_pl = timeit(lambda: hip.conv_run(d_pl, xp, wp, None, y, bias=bias, R=r))
    except hip.VlfbError as e:
        t_pl = float('nan')
        print('   (planes: %s)' % e)
    # the 16-bit kernel of the same launch (what an all-fp16 forward costs)
    xh, wh = x.half(), wf.half()
    rh = r.half() if res else None
    d_16 = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, relu=1,
                         bias_mode=hip.BIAS_COL, **G)
    t_16 = timeit(lambda: hip.conv_run(d_16, xh, wh, None, yh, bias=bias, R=rh))
    print('%-22s %-28s split %8.1f us %6.1f TF | planes %8.1f us %6.1f TF | fp16 %7.1f us (%s)' % (
        name, hip.conv_plan(d_pl) if t_pl == t_pl else '-', t_sp, fl / t_sp / 1e6, t_pl, fl / t_pl / 1e6, t_16, hip.conv_plan(d_16)))


C = 8
run('res2 2c 64->256', C, 64, 256, 32, 56, 56, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res2 2a 256->64 k311', C, 256, 64, 32, 56, 56, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), res=False)
run('res2 2b 64->64 k133', C, 64, 64, 32, 56, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), res=False)
run('res3 2c 128->512', C, 128, 512, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res3 2a 512->128', C, 512, 128, 16, 28, 28, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), res=False)
run('res3 2b 128 k133', C, 128, 128, 16, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), res=False)
run('res4 2c 256->1024', C, 256, 1024, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res4 2a 1024->256', C, 1024, 256, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), res=False)
run('res4 2a 1024->256 k311', C, 1024, 256, 16, 14, 14, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), res=False)
run('res4 2b 256 k133', C, 256, 256, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), res=False)
run('nl theta 1024->512', C, 1024, 512, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), res=False)
run('res5 2c 512->2048', C, 512, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
run('res5 2a 2048->512', C, 2048, 512, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), res=False)
run('res5 2a 2048->512 k311', C, 2048, 512, 16, 14, 14, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), res=False)
run('res5 2b 512 k133 d2', C, 512, 512, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2), res=False)
run('res5 br1 1024->2048', C, 1024, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), res=False)
