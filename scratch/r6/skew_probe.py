"""r6 probe: thin-K two-plane launches with a residual (the 2c layers) -- does a start skew of the second workgroup of a CU overlap
its k-loop with the first one's epilogue?  MEASURED: no -- 123.6 / 80.0 / 233.7 us without, 136-181 / 78-107 / 235-285 us with a skew of
1-8 x 8 k cycles (res3 / res4 / res5 2c): the serial look of these launches (time = MFMA time + epilogue HBM time) is tile
quantisation -- 1568 tiles on 512 slots = 3.06 rounds -- not lockstep.  The VLFB_SKEW switch this script drove was removed again
(git history: the commit before "revert the skew experiment").  Kept as the record of the negative result."""
import os, sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import torch
from vlfb import hip
hip.lib()
dev = torch.device('cuda:0')
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def run(name, M, Cin, Cout):
    x = torch.randn(M, Cin, device=dev); wf = torch.randn(Cout, 1, Cin, device=dev) * 0.05
    wh = torch.empty(2, Cout, 1, Cin, device=dev, dtype=torch.float16)
    hip.call("vlfb_weight_prep", wf.data_ptr(), None, wh.data_ptr(), None, hip.MIXH, Cout, 1, Cin)
    xp = torch.empty(2 * x.numel(), device=dev, dtype=torch.float16); hip.call("vlfb_pair_split", x.data_ptr(), xp.data_ptr(), x.numel())
    n_out = M * Cout
    r = torch.randn(n_out, device=dev); rp = torch.empty(2 * n_out, device=dev, dtype=torch.float16); hip.call("vlfb_pair_split", r.data_ptr(), rp.data_ptr(), n_out)
    yp = torch.empty(2 * n_out, device=dev, dtype=torch.float16); bias = torch.randn(Cout, device=dev)
    d = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16X3, N=1, Tr=1, Hr=1, Wr=M, Ts=1, Hs=1, Ws=M, Cs=Cin, Cn=Cout,
                      relu=1, bias_mode=hip.BIAS_COL, alpha=1.0 / 1024, a_pstride=x.numel(), b_pstride=wf.numel(), algo=hip.ALGO_TILE128)
    t = timeit(lambda: hip.conv_run(d, xp, wh, None, yp, bias=bias, R=rp, R_lo=rp[n_out:], O_lo=yp[n_out:]))
    print('skew %s  %-20s %7.1f us  %6.1f TF  (%s)' % (os.environ.get('VLFB_SKEW', '0'), name, t, 2.0 * M * Cout * Cin / t / 1e6, hip.conv_plan(d)))
run('res3 2c 128->512', 100352, 128, 512)
run('res4 2c 256->1024', 25088, 256, 1024)
run('res5 2c 512->2048', 25088, 512, 2048)
run('res2 2c 64->256', 802816, 64, 256)
