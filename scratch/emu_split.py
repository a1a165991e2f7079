"""CPU experiment (test infrastructure): how does the parameter-gradient error scale with the precision of the FORWARD
operands?  Rounds stored activations + weight operands to `bits` significant bits inside the fp64 oracle (8 = bf16,
16 = hi+lo bf16 pair, 24 = fp32) and prints the gradient error table.  Usage: python scratch/emu_split.py [preset] bits..."""
import sys
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib'); sys.path.insert(0, '.')
import numpy as np, torch
import scratch.emu_bf16 as E

def make_round(bits):
    def r(t):
        m, e = torch.frexp(t)
        s = float(2 ** bits)
        return torch.ldexp(torch.round(m * s) / s, e)
    return r

if __name__ == "__main__":
    preset = sys.argv[1]
    torch.set_num_threads(8)
    ref_b, ref_g = E.run(preset, {})
    for bits in [int(x) for x in sys.argv[2:]]:
        E.r16 = make_round(bits)
        for vname in ("fwd_only", "bwd_only"):
            b, g = E.run(preset, E.VARIANTS[vname])
            errs = sorted(((E.rel(g[n], ref_g[n]), n) for n in ref_g if float(ref_g[n].norm()) > 1e-12), reverse=True)
            e = np.array([x for x, _ in errs])
            print("bits %2d %-9s prob %.2e | grads median %.2e p90 %.2e max %.2e (%s) conv1_w %.2e" % (
                bits, vname, E.rel(b["prob"], ref_b["prob"]), np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], errs[0][1],
                E.rel(g["conv1_w"], ref_g["conv1_w"])), flush=True)
