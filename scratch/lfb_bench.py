"""time the device feature-bank kernels at the AVA training shape (24 RoIs, window 60 x 5, dim 2048)"""
import sys, time
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
import numpy as np, torch
from vlfb.lfb_bank import DeviceBank
bank = DeviceBank(64, 900, 16, 2048, "bf16", step_base=902)
rng = np.random.default_rng(0)
for it in range(40):
    r = 512
    f = torch.randn(r, 2048, device="cuda").to(torch.bfloat16)
    bank.append(f, rng.integers(0, 64, r), rng.integers(902, 1802, r))
print("bank GB", bank.bank.numel() * 2 / 1e9, "features", int(bank.counts().sum()), "dropped", int(bank.dropped.item()))
R = 24
out = torch.empty(R, 300, 2048, device="cuda", dtype=torch.bfloat16)
v = rng.integers(0, 64, R); s = rng.integers(902, 1802, R)
for _ in range(3):
    bank.sample_window(v, s, np.arange(R), 60, 5, 1, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    bank.sample_window(v, s, np.arange(R), 60, 5, 1, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("sample_window: %.1f us per call incl. host (%d MB written)" % (dt * 1e6, out.numel() * 2 / 1e6))
f = torch.randn(64, 2048, device="cuda").to(torch.bfloat16)
t0 = time.perf_counter()
for _ in range(20):
    bank.append(f, rng.integers(0, 64, 64), rng.integers(902, 1802, 64))
torch.cuda.synchronize(); print("append 64 rows: %.1f us per call incl. host" % ((time.perf_counter() - t0) / 20 * 1e6))
