"""Same-box A/B per DGRAD launch of the benchmarked `mix` plan: the planner's choice against algo = PIPE256 (the 256-row
phase-pipelined kernel) and algo = TILE128 -- with two-term weights the k-loop of every DGRAD is twice as long, which is the
regime the pipelined kernel was built for.  Usage: python scratch/r5/dgrad_algo_ab.py [dtype]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT)
import torch
from vlfb import hip, synth
from vlfb.presets import load_preset
from core.config import config as cfg
from models.model_builder_video import ModelBuilder
from vlfb.engine import Engine, ConvStep

dtype = sys.argv[1] if len(sys.argv) > 1 else "mix"
load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 8, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224])
model = ModelBuilder(train=True, split="train", name="ab")
model.build_model(suffix="_train")
eng = Engine(model, dtype, device="cuda:0", base_seed=2)
rois = synth.rois_per_clip_draw(8, seed=2)
batch = synth.inputs(cfg, 8, rois, seed=2, crop=224, frames=32)
eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
eng.feed_params(synth.params(model, seed=2))
for k, v in batch.items():
    if k in model.input_blob_names:
        eng.feed(k, v)
eng.forward(); eng.backward(); torch.cuda.synchronize()
ALGOS = [("auto", 0), ("tile128", 1), ("pipe256", 2)]
seen = {}
tot = collections.defaultdict(float)
print("%-62s %10s %10s %10s   plan(auto)" % ("launch", "auto us", "tile128", "pipe256"))
for st in eng.steps:
    if not isinstance(st, ConvStep) or st.d_d is None or getattr(st, "bwd_split", False) or st.group != 1:
        continue
    d0 = st.d_d
    tag = hip.conv_tag(d0) + (" f32out" if d0.out_dtype == hip.F32 else "")
    if tag in seen:
        for k, v in seen[tag].items():
            tot[k] += v
        continue
    g = torch.randn(st.out.numel, device="cuda:0").to(torch.float16 if d0.dtype == hip.F16 else torch.bfloat16)
    out = torch.empty(st.x.numel, device="cuda:0", dtype=torch.float32 if d0.out_dtype == hip.F32 else g.dtype)
    res = {}
    for name, algo in ALGOS:
        d = hip.ConvDesc.from_buffer_copy(bytes(d0)); d.algo = algo
        try:
            hip.conv_workspace_bytes(d)
            for _ in range(3):
                hip.conv_run(d, g, st.w_d, None, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.conv_run(d, g, st.w_d, None, out)
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / 20
        except hip.VlfbError:
            res[name] = float("nan")
    seen[tag] = res
    for k, v in res.items():
        tot[k] += v
    print("%-62s %10.1f %10.1f %10.1f   %s" % (tag, res["auto"], res["tile128"], res["pipe256"], hip.conv_plan(d0)), flush=True)
best = sum(min(v for v in r.values() if v == v) for r in seen.values())
print("sum over the step's DGRAD launches (us): auto %.0f  tile128 %.0f  pipe256 %.0f  (nan = not plannable)" % (tot["auto"], tot["tile128"], tot["pipe256"]))
