"""Where does the "mix" backward lose its digits?  Full-size clip, identical decisions: the engine's activation gradients
(hi + lo term where a slot keeps two) against the oracle's along the backward chain, then conv1_w recomputed on the CPU from
mixed operands (engine gradient x exact clip, oracle gradient x fp16 clip) to separate its WGRAD arithmetic from upstream.
Usage: python scratch/r5/diag_mix.py [preset]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
from test_model_gpu import build, rel, SMALL
from vlfb.engine import Engine
from oracle import model as om

preset = sys.argv[1] if len(sys.argv) > 1 else "ava_r50_lfb_nl"
FULL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224]
size = SMALL if os.environ.get("MIX_SMALL") else FULL
cfg, model, eng, inputs, params, seed_fn = build(preset, "mix", size)
eng.forward(); eng.backward(); torch.cuda.synchronize()
dec = eng.discrete_decisions()
torch.set_num_threads(min(32, os.cpu_count()))
P = collections.OrderedDict(); spec = om.param_spec(cfg)
for k, v in params.items():
    t = torch.from_numpy(v).double()
    if spec[k]["trainable"]:
        t.requires_grad_(True)
    P[k] = t
I = {k: (torch.from_numpy(v) if v.dtype != np.float32 or k == "proposals" else torch.from_numpy(v).double()) for k, v in inputs.items()}
B = om.forward(cfg, P, I, "train", False, torch.float64, seed_fn, decisions=dec)
for k, t in B.items():
    if isinstance(t, torch.Tensor) and t.requires_grad and t.dtype == torch.float64:
        t.retain_grad()
B["loss"].backward()


def eng_grad(name):
    b = eng.env[name].root
    s = b.slot
    t = s.cur.detach().double().cpu()
    if s.cur_lo is not None:
        t = t + s.cur_lo.detach().double().cpu()
    t = t / eng.loss_scale / b.grad_scale
    bb = eng.env[name]
    order = [ax for ax in range(len(bb.shape)) if ax != bb.caxis] + [bb.caxis]
    stor = t[:bb.numel].view([bb.shape[ax] for ax in order])
    inv = [order.index(ax) for ax in range(len(bb.shape))]
    return stor.permute(inv).contiguous().numpy()


print("== activation gradients, backward order (rel L2 vs oracle on identical decisions; '2t' = two-term slot) ==")
for name in reversed(list(B.keys())):
    t = B[name]
    if not (isinstance(t, torch.Tensor) and t.grad is not None) or name not in eng.env:
        continue
    b = eng.env[name]
    if b.root.slot is None or b.root.slot.cur is None or getattr(b.root, "dead", False):
        continue
    try:
        got = eng_grad(name)
    except Exception as e:
        print(name, "fetch failed", e); continue
    ref = t.grad.numpy().reshape(got.shape)
    if b.root.relu:
        ref = ref * dec["relu"][name].reshape(got.shape) if name in dec["relu"] else ref * (t.detach().numpy().reshape(got.shape) > 0)
    print("%-36s %.3e  %s%s (norm %.3e)" % (name, rel(got, ref), "2t " if b.root.slot.cur_lo is not None else "   ",
                                           "f32" if b.root.slot.cur.dtype == torch.float32 else "f16", np.linalg.norm(ref)), flush=True)
print("== parameter gradients, backward order ==")
for n in eng.train_order:
    if P[n].grad is None:
        continue
    print("%-36s %.3e" % (n, rel(eng.fetch_grad(n), P[n].grad.numpy())))
# conv1_w from mixed operands
x = I["data"]
g_ref = B["conv1"].grad if "conv1" in B and B["conv1"].grad is not None else None
name1 = "res_conv1_bn"
g_eng = torch.from_numpy(eng_grad(name1))           # gradient at the stem's output (post affine + ReLU), masked
s = P["res_conv1_bn_s"].detach().view(1, -1, 1, 1, 1)
def wgrad(xx, gy):                                  # d/dW of conv1 for an output-of-affine gradient gy
    w = P["conv1_w"].detach().clone().requires_grad_(True)
    y = F.conv3d(xx, w, None, (1, 2, 2), (2, 3, 3))
    y.backward(gy * s)
    return w.grad.numpy()
ref_w = P["conv1_w"].grad.numpy()
g_or = B[name1].grad * torch.from_numpy(dec["relu"][name1].reshape(B[name1].shape)).double()
print("conv1_w: engine %.3e | exact wgrad of the ENGINE's output gradient %.3e | exact gradient x fp16(clip) %.3e | engine gradient x fp16(clip) %.3e"
      % (rel(eng.fetch_grad("conv1_w"), ref_w), rel(wgrad(x, g_eng), ref_w), rel(wgrad(x.half().double(), g_or), ref_w),
         rel(wgrad(x.half().double(), g_eng), ref_w)))
