import sys, collections; sys.path.insert(0,'video-long-term-feature-banks_amd/lib'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from test_model_gpu import build, rel
from oracle import model as om
preset = sys.argv[1] if len(sys.argv)>1 else 'charades_r50_baseline'
dtype = sys.argv[2] if len(sys.argv)>2 else 'fp32'
cfg, model, eng, inputs, params, seed_fn = build(preset, dtype)
eng.forward(); eng.backward(); torch.cuda.synchronize()
# oracle with retained activation grads
P = collections.OrderedDict(); spec = om.param_spec(cfg)
for k,v in params.items():
    t = torch.from_numpy(v).double()
    if spec[k]['trainable']: t.requires_grad_(True)
    P[k]=t
I = {k:(torch.from_numpy(v) if v.dtype!=np.float32 or k=='proposals' else torch.from_numpy(v).double()) for k,v in inputs.items()}
B = om.forward(cfg, P, I, 'train', False, torch.float64, seed_fn)
for k,t in B.items():
    if isinstance(t, torch.Tensor) and t.requires_grad and t.dtype==torch.float64: t.retain_grad()
B['loss'].backward()
print('== activation grads (backward order) ==')
for name in reversed(list(B.keys())):
    t = B[name]
    if not (isinstance(t, torch.Tensor) and t.grad is not None): continue
    if name not in eng.env: continue
    b = eng.env[name]
    if b.root.slot is None or b.root.slot.cur is None: continue
    try:
        got = eng.fetch(name+'_grad')
    except Exception as e:
        print(name, 'fetch failed', e); continue
    ref = t.grad.numpy().reshape(got.shape)
    # oracle grad is wrt post-relu value; engine's finished grad is masked by (y>0)
    if b.root.relu: ref = ref * (t.detach().numpy().reshape(got.shape) > 0)
    print('%-34s %.3e   (norm %.3e)' % (name, rel(got, ref), np.linalg.norm(ref)))
print('== param grads (backward order) ==')
for n in eng.train_order:
    ref = P[n].grad.numpy(); got = eng.fetch_grad(n)
    print('%-34s %.3e   (norm %.3e)' % (n, rel(got, ref), np.linalg.norm(ref)))
