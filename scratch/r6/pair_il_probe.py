"""r6 probe: two fp16 planes [2][M][C] vs the same planes interleaved in 32-k groups [M][C/32][hi 32 | lo 32] (whole 128-byte
lines per k-tile row) -- plain-row launches, same kernel, same math"""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
from vlfb import hip
hip.lib()
dev = torch.device('cuda:0')


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


def il(planes, rows, C):
    """[2][rows][C] -> [rows][C/32][2][32]"""
    return planes.view(2, rows, C // 32, 32).permute(1, 2, 0, 3).contiguous()


def run(name, M, Cin, Cout):
    x = torch.randn(M, Cin, device=dev)
    wf = torch.randn(Cout, 1, Cin, device=dev) * 0.05
    wh = torch.empty(2, Cout, 1, Cin, device=dev, dtype=torch.float16)
    hip.call("vlfb_weight_prep", wf.data_ptr(), None, wh.data_ptr(), None, hip.MIXH, Cout, 1, Cin)
    xp = torch.empty(2 * x.numel(), device=dev, dtype=torch.float16)
    hip.call("vlfb_pair_split", x.data_ptr(), xp.data_ptr(), x.numel())
    n_out = M * Cout
    yp = torch.empty(2 * n_out, device=dev, dtype=torch.float16)
    yq = torch.empty(2 * n_out, device=dev, dtype=torch.float16)
    bias = torch.randn(Cout, device=dev)
    base = dict(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16X3, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=Cin, Cn=Cout,
                relu=1, bias_mode=hip.BIAS_COL, alpha=1.0 / 1024)
    d_pl = hip.conv_desc(a_pstride=x.numel(), b_pstride=wf.numel(), **base)
    d_il = hip.conv_desc(a_pstride=32, b_pstride=32, lda=2 * Cin, ldb=2 * Cin, **base)
    xi, wi = il(xp, M, Cin), il(wh.view(-1), Cout, Cin)
    t_pl = timeit(lambda: hip.conv_run(d_pl, xp, wh, None, yp, bias=bias, O_lo=yp[n_out:]))
    t_il = timeit(lambda: hip.conv_run(d_il, xi, wi, None, yq, bias=bias, O_lo=yq[n_out:]))
    torch.cuda.synchronize()
    same = torch.equal(yp, yq)
    xh, w16 = x.half(), wf.half()
    yh = torch.empty(n_out, device=dev, dtype=torch.float16)
    d_16 = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=Cin, Cn=Cout, relu=1, bias_mode=hip.BIAS_COL,
                         algo=hip.ALGO_TILE128)
    t_16 = timeit(lambda: hip.conv_run(d_16, xh, w16, None, yh, bias=bias))
    fl = 2.0 * M * Cout * Cin
    print('%-22s planes %7.1f us %6.1f TF | interleaved %7.1f us %6.1f TF (bit-identical: %s) | fp16 128x128 %6.1f us %6.1f TF' % (
        name, t_pl, fl / t_pl / 1e6, t_il, fl / t_il / 1e6, same, t_16, fl / t_16 / 1e6))


run('res5 2c 512->2048', 25088, 512, 2048)
run('res5 br1 1024->2048', 25088, 1024, 2048)
run('res5 2a 2048->512', 25088, 2048, 512)
run('res4 2c 256->1024', 25088, 256, 1024)
run('res4 2a 1024->256', 25088, 1024, 256)
run('res3 2c 128->512', 100352, 128, 512)
run('res3 2a 512->128', 100352, 512, 128)
run('res2 2c 64->256', 802816, 64, 256)
run('big 8192^2 x 4096', 8192, 4096, 8192)
