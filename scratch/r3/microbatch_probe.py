"""Development probe: aggregate clips/s of TWO independent 4-clip engines running concurrently on separate HIP streams
versus ONE 8-clip engine (does filling the tails / pairing phases of two micro-batches pay?).
usage: microbatch_probe.py [dtype] [steps]"""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")); sys.path.insert(0, ROOT)
import torch
from vlfb import synth
from vlfb.presets import load_preset
from core.config import config as cfg
from models.model_builder_video import ModelBuilder
from vlfb.engine import Engine
import utils.lr_policy as lr_policy

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def make(clips, seed):
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", clips, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224])
    model = ModelBuilder(train=True, split="train", name="mb%d" % seed)
    model.build_model(suffix="_train")
    eng = Engine(model, dtype, device="cuda:0", base_seed=cfg.RNG_SEED)
    rois = synth.rois_per_clip_draw(clips, seed=seed)
    batch = synth.inputs(cfg, clips, rois, seed=seed, crop=224, frames=32)
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
    eng.feed_params(synth.params(model, seed=cfg.RNG_SEED))
    for k, v in batch.items():
        if k in model.input_blob_names:
            eng.feed(k, v)
    return eng


lr = 0.01
one = make(8, 2)
for _ in range(3):
    one.train_step(lr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one.train_step(lr)
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / steps
del one
torch.cuda.empty_cache()
engs = [make(4, 2), make(4, 3)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for _ in range(3):
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.train_step(lr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.forward()
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e._backward_and_solve()
torch.cuda.synchronize()
t_two = (time.perf_counter() - t0) / steps
# the same two engines back to back on ONE stream (cost of the smaller launches alone)
t0 = time.perf_counter()
for _ in range(steps):
    for e in engs:
        e.train_step(lr)
torch.cuda.synchronize()
t_seq = (time.perf_counter() - t0) / steps
print("%s: one 8-clip engine %.2f ms (%.1f clips/s) | two 4-clip engines on two streams %.2f ms (%.1f clips/s) | the same two back to back %.2f ms (%.1f clips/s)"
      % (dtype, t_one * 1e3, 8 / t_one, t_two * 1e3, 8 / t_two, t_seq * 1e3, 8 / t_seq))
