import sys, time, os
sys.path.insert(0,'video-long-term-feature-banks_amd/lib'); sys.path.insert(0,'.')
import torch
from vlfb.presets import load_preset
from core.config import config as cfg
from oracle import model as om
print('cores', os.cpu_count())
for threads in (16, 64, 256):
    for (fr, cr) in ((8, 64), (8, 112)):
        torch.set_num_threads(threads)
        load_preset('ava_r50_lfb_nl', ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", fr, "TRAIN.CROP_SIZE", cr])
        inputs = om.synth_inputs(cfg, 1, "train", seed=2, rois_per_clip=[3], crop=cr, frames=fr)
        params = om.synth_params(cfg, seed=2)
        t0=time.time(); om.run(cfg, params, inputs, "train", torch.float32, False, lambda n: 1); t1=time.time()
        om.run(cfg, params, inputs, "train", torch.float32, True, lambda n: 1); t2=time.time()
        print('threads %d clip %dx%d: fwd %.2fs fwd+bwd %.2fs' % (threads, fr, cr, t1-t0, t2-t1), flush=True)
