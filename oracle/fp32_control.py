"""TEST INFRASTRUCTURE (not product code).  The reference-width witness of the parity protocol:

the reference runs fp32 arithmetic; its parameter gradients differ from the fp64 oracle's by much more than fp32 rounding
whenever ONE ReLU / max-pool / RoI-bin decision falls the other way (a pre-activation within fp32 rounding of zero).  This
script evaluates the ORACLE ITSELF in fp32 (torch CPU: plain fp32 convolutions, the arithmetic width of the reference's
Caffe2 CPU path) against the fp64 oracle on one clip at the benchmarked size (32 x 224^2), every trainable parameter:

  raw                   fp32 oracle vs fp64 oracle, each on its own decisions
  identical decisions   fp32 oracle vs the fp64 oracle RE-EVALUATED on the fp32 run's decisions (oracle.model.run(decisions=...))
  flips                 ReLU units / max-pool selections / RoI bins that the two widths decide differently

No GPU, no engine: an independent control for the claim that the raw table measures ties, not arithmetic.
usage: python oracle/fp32_control.py <preset> <out.txt> [threads]      (ava_r50_lfb_nl | charades_r50_baseline)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (d if d > 0 else 1.0))


def main(preset, out_path, threads):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from vlfb import rng as vrng
    from oracle import model as om
    torch.set_num_threads(threads)
    load_preset(preset, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224])
    # (the inputs and parameters of tests/test_model_gpu.py::test_full_size_clip_matches_oracle)
    inputs = om.synth_inputs(cfg, 1, "train", seed=cfg.RNG_SEED, rois_per_clip=[2] if cfg.DATASET == "ava" else None,
                             crop=cfg.TRAIN.CROP_SIZE, frames=cfg.TRAIN.VIDEO_LENGTH)
    params = om.synth_params(cfg, seed=cfg.RNG_SEED)
    seed_fn = lambda name: vrng.dropout_seed(cfg.RNG_SEED, name, 0)
    t0 = time.time()
    d64 = {"_record": True}
    b64, g64 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=d64)
    t1 = time.time()
    d32 = {"_record": True}
    b32, g32 = om.run(cfg, params, inputs, "train", torch.float32, True, seed_fn, decisions=d32)
    t2 = time.time()
    dec = {"relu": d32["relu"], "pool": d32["pool"], "roi_bin": d32.get("roi_bin")}
    _, g64d = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
    assert not dec["_missing"], sorted(dec["_missing"])
    t3 = time.time()
    relu_flip = {n: int((d32["relu"][n] != d64["relu"][n]).sum()) for n in d64["relu"]}
    pool_flip = {n: int((d32["pool"][n] != d64["pool"][n]).sum()) for n in d64["pool"]}
    roi_flip = int((d32["roi_bin"] != d64["roi_bin"]).sum()) if d64.get("roi_bin") is not None else 0
    n_relu = sum(int(np.prod(m.shape)) for m in d64["relu"].values())
    gmax = max(float(g.norm()) for g in g64.values())
    names = [n for n in g64 if float(g64[n].norm()) > 1e-9 * gmax]
    raw = {n: rel(g32[n].numpy(), g64[n].numpy()) for n in names}
    same = {n: rel(g32[n].numpy(), g64d[n].numpy()) for n in names}
    acts = [(n, rel(b32[n].detach().numpy(), b64[n].detach().numpy())) for n in
            ("res_conv1_bn", "pool1", "res2_2_branch2c_bn", "res3_3_branch2c_bn", "res4_5_branch2c_bn", "res5_2_branch2c_bn", "pool5", "pred", "prob")
            if n in b64 and n in b32]

    def stats(d):
        e = np.sort(list(d.values()))
        return float(np.median(e)), float(e[int(0.9 * (len(e) - 1))]), float(e[-1])
    L = []
    L.append("fp32 torch-CPU oracle vs fp64 torch-CPU oracle, %s, 1 clip 32x224x224 (oracle/fp32_control.py; %d threads; fp64 %.0f s, fp32 %.0f s, fp64 on"
             " the fp32 decisions %.0f s)" % (preset, threads, t1 - t0, t2 - t1, t3 - t2))
    L.append("activations / outputs (relative L2):  " + ", ".join("%s=%.2e" % x for x in acts))
    L.append("decisions taken differently by the two widths: %d of %d ReLU units (%s), %d max-pool selections (%s), %d RoI arg-max bins"
             % (sum(relu_flip.values()), n_relu, ", ".join("%s: %d" % (n, c) for n, c in relu_flip.items() if c) or "none",
                sum(pool_flip.values()), ", ".join("%s: %d" % (n, c) for n, c in pool_flip.items() if c) or "none", roi_flip))
    L.append("parameter gradients (%d tensors), relative L2:" % len(names))
    L.append("   raw (each width on its own decisions):            median %.3e  p90 %.3e  max %.3e" % stats(raw))
    L.append("   fp64 re-evaluated on the fp32 run's decisions:    median %.3e  p90 %.3e  max %.3e" % stats(same))
    for n in sorted(names, key=lambda k: -raw[k]):
        L.append("  %-44s %.3e   (same decisions: %.3e)" % (n, raw[n], same[n]))
    txt = "\n".join(L) + "\n"
    with open(out_path, "w") as fh:
        fh.write(txt)
    print("\n".join(L[:6]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1))
