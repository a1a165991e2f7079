"""Development probe (not product, not a test): one layer shape of the 8-clip ava_r50_lfb_nl step through vlfb_conv_run with
split-bf16 math, timed with HIP events; for rocprofv3 --pmc passes (scratch/r3/pmc_sp.sh).
usage: sp_probe.py <case> [reps] [algo]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"))
import torch
from vlfb import hip

CASES = {
    # name: (mode, N, Cin, Cout, T, H, W, k, s, p, d)
    "res5_3x3": (8, 512, 512, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    "res4_3x3": (8, 256, 256, 16, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    "res5_2c": (8, 512, 2048, 16, 14, 14, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    "res4_2a_t": (8, 1024, 256, 16, 14, 14, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
    "res3_3x3": (8, 128, 128, 16, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    "res2_2c": (8, 64, 256, 32, 56, 56, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
}


def main():
    case = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    algo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    N, Cin, Cout, T, H, W, k, s, p, d = CASES[case]
    dev = torch.device("cuda:0")
    g = dict(kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2], dt=d[0], dh=d[1], dw=d[2])
    taps = k[0] * k[1] * k[2]
    x = torch.randn(N, T, H, W, Cin, device=dev)
    dy = torch.randn(N, T, H, W, Cout, device=dev)
    w = torch.randn(Cout, taps, Cin, device=dev) * 0.05
    wf = torch.empty(3, Cout, taps, Cin, device=dev, dtype=torch.bfloat16)
    wd = torch.empty(2, Cin, taps, Cout, device=dev, dtype=torch.bfloat16)
    hip.call("vlfb_weight_prep", hip.ptr(w), None, hip.ptr(wf), hip.ptr(wd), hip.SPLIT, Cout, taps, Cin)
    o = torch.empty(N, T, H, W, Cout, device=dev)
    dx = torch.empty(N, T, H, W, Cin, device=dev)
    dw = torch.empty(Cout, taps, Cin, device=dev)
    plane = Cout * taps * Cin
    common = dict(dtype=hip.F32, out_dtype=hip.F32, N=N, algo=algo, **g)
    descs = {
        "fprop": (hip.conv_desc(mode=hip.FPROP, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, math=6, b_pstride=plane, **common), (x, wf, None, o)),
        "dgrad": (hip.conv_desc(mode=hip.DGRAD, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=Cout, Cn=Cin, math=3, b_pstride=plane, **common), (dy, wd, None, dx)),
        "wgrad": (hip.conv_desc(mode=hip.WGRAD, Tr=T, Hr=H, Wr=W, Ts=T, Hs=H, Ws=W, Cs=Cin, Cn=Cout, math=3, **common), (x, None, dy, dw)),
    }
    ws = torch.empty(max(hip.conv_workspace_bytes(descs["wgrad"][0]), 16) // 4, device=dev)
    flops = 2.0 * N * T * H * W * Cin * Cout * taps
    for name, (desc, args) in descs.items():
        for _ in range(2):
            hip.conv_run(desc, *args, workspace=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.conv_run(desc, *args, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%-10s %-6s %9.1f us  %7.1f TFLOP/s (algorithmic)" % (case, name, us, flops / us / 1e6))


if __name__ == "__main__":
    main()
