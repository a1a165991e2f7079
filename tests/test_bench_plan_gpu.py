"""Parity ON THE BENCHMARKED PLAN: the engine bench.py times -- ava_r50_lfb_nl, 8 clips of 32 x 224^2 per GPU, the bench's RoI
draw (per-GPU batch of lib/utils/misc.py:68-72 at the north-star point) -- against engines that run ONE of its clips at a
time, i.e. the size at which test_model_gpu.py::test_full_size_clip_matches_oracle holds every path to the fp64 oracle.

The planner picks tiles, split-K slab counts, 256-row / streaming / whole-row kernel families from the row count M, so the
8-clip launches (M = 25 088 ... 3 211 264) are not the 1-clip launches (M = 3 136 ... 401 408).  This test closes the gap:
  * forward: every checked blob of clip i inside the 8-clip engine is BIT-IDENTICAL to the 1-clip engine's (each output
    row of every kernel family is the same sum in the same order whatever the tile / family);
  * backward: every parameter gradient of the 8-clip step equals the sum of the per-clip gradients (each clip's loss
    carries its share R_i / R of the per-GPU normaliser, resnet_video.py:333-338) -- up to the fp32 summation order of the
    split-K slabs and, on the 16-bit paths, the last-bit rounding of a differently grouped fp32 sum;
  * the plan under test is the plan under the stopwatch: the kernel-family table of this engine (Engine.plan_table, from
    the library's planner) equals the table of the engine bench.py builds, and it contains every family the bench line's
    roofline talks about (256-row pipelined, streaming, direct-convolution, whole-row, split-K with slabs).
Dropout is off in BOTH engines of a comparison (its mask is a function of the RoI row index, which restarts at 0 in a
1-clip engine); dropout steps are elementwise and take no part in the plan table.
"""
import collections
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CLIPS = 8
NO_DROPOUT = ["TRAIN.DROPOUT_RATE", 0.0, "FBO_NL.INPUT_DROPOUT_ON", False, "FBO_NL.LFB_DROPOUT_ON", False]
BLOBS = ["res2_2_branch2c_bn", "nonlocal_conv3_1_sum", "res3_3_branch2c_bn", "nonlocal_conv4_1_sum", "res4_5_branch2c_bn",
         "res5_2_branch2c_bn"]
ROW_BLOBS = ["box_pooled", "pool5", "pred", "prob"]


def _model(n_clips, overrides=()):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", n_clips, "TRAIN.VIDEO_LENGTH", 32,
                                   "TRAIN.CROP_SIZE", 224] + list(overrides))
    model = ModelBuilder(train=True, split="train", name="plan%d" % n_clips)
    model.build_model(suffix="_train")
    return cfg, model


def _engine(model, dtype, batch, params, **kw):
    from vlfb.engine import Engine
    eng = Engine(model, dtype, base_seed=2, **kw)
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
    if not kw.get("dry_run"):
        eng.feed_params(params)
        for k, v in batch.items():
            if k in model.input_blob_names:
                eng.feed(k, v)
    return eng


def _bench_batch(cfg):
    from vlfb import synth
    rois = synth.rois_per_clip_draw(CLIPS, seed=cfg.RNG_SEED)          # rank 0 of bench.py
    return rois, synth.inputs(cfg, CLIPS, rois, seed=cfg.RNG_SEED, crop=224, frames=32)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (d if d > 0 else 1.0))


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "mix", "split"])
def test_eight_clip_step_equals_the_one_clip_steps(dtype):
    from vlfb import synth
    from vlfb.engine import LossStep
    cfg, model8 = _model(CLIPS, NO_DROPOUT)
    rois, batch = _bench_batch(cfg)
    params = synth.params(model8, seed=cfg.RNG_SEED)
    eng8 = _engine(model8, dtype, batch, params)
    table8 = eng8.plan_table()
    eng8.forward()
    eng8.backward()
    torch.cuda.synchronize()
    launched8 = eng8.plan_table(launched=True)        # ("split": with the operand-plane variants the pass really used)
    fwd8 = {n: eng8.fetch(n) for n in BLOBS + ROW_BLOBS}
    grad8 = {n: eng8.fetch_grad(n) for n in eng8.trainable}
    loss8 = float(eng8.fetch("loss").reshape(-1)[0])
    loss_scale8 = eng8.loss_scale
    R = sum(rois)
    del eng8
    torch.cuda.empty_cache()

    # ---- the plan under test is the plan bench.py times (its engine: same model WITH dropout, same batch) ----------
    cfg, model_b = _model(CLIPS)
    bench_table = _engine(model_b, dtype, batch, None, dry_run=True).plan_table()
    assert [r[2:] for r in table8] == [r[2:] for r in bench_table], "the tested plan is not the benchmarked plan"
    fams = collections.Counter(r[3].split()[0] for r in launched8)
    slabs = [int(r[3].split("splits=")[1]) for r in launched8 if "splits=" in r[3]]
    print("\n[%s] kernel families of the 8-clip plan: %s; split-K launches: %d (up to %d slabs)"
          % (dtype, dict(fams), sum(1 for s_ in slabs if s_ > 1), max(slabs)))
    if dtype in ("fp16", "bf16"):
        for fam in ("nt", "nt8", "nt_stream", "conv_rows64", "stem_fprop", "nt_skinny", "tn_tr", "tn8", "wgrad_rows",
                    "wgrad_rows_fat", "stem_wgrad"):
            assert fams[fam] > 0, "family %s missing from the benchmarked plan: %r" % (fam, dict(fams))
    elif dtype == "mix":
        for fam in ("nt_split", "nt", "nt8", "tn_tr", "tn8", "wgrad_rows", "stem_wgrad", "tn_split"):
            assert fams[fam] > 0, "family %s missing from the benchmarked plan: %r" % (fam, dict(fams))
    else:
        for fam in ("nt_split", "nt_planes", "tn_split", "tn_tr_planes"):
            assert fams[fam] > 0, "family %s missing from the benchmarked plan: %r" % (fam, dict(fams))
    assert max(slabs) > 8, "no split-K wgrad with many slabs in the plan"
    out = os.environ.get("VLFB_PARITY_DIR")
    if out:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "bench_plan_%s.txt" % dtype), "w") as fh:
            fh.write("# kernel family / tile / split count of every implicit-GEMM launch of the benchmarked step "
                     "(ava_r50_lfb_nl, 8 clips 32x224x224, %d RoIs, dtype %s): Engine.plan_table()\n" % (R, dtype))
            for row in launched8:
                fh.write("%-44s %-7s %-64s %s\n" % row)

    # ---- one clip at a time -----------------------------------------------------------------------------------------
    row0 = np.cumsum([0] + rois)
    engines = {}
    gsum = {n: np.zeros_like(g, dtype=np.float64) for n, g in grad8.items()}
    loss_sum = 0.0
    worst_fwd = 0.0
    for i in range(CLIPS):
        r = rois[i]
        sl = slice(int(row0[i]), int(row0[i + 1]))
        one = collections.OrderedDict()
        one["data_train"] = batch["data_train"][i:i + 1]
        prop = batch["proposals_train"][sl].copy()
        prop[:, 0] = 0
        one["proposals_train"] = prop
        one["labels_train"] = batch["labels_train"][sl]
        one["lfb_train"] = batch["lfb_train"][sl]
        if r not in engines:
            cfg, model1 = _model(1, NO_DROPOUT)
            engines[r] = _engine(model1, dtype, one, params, loss_scale=loss_scale8 if dtype in ("fp16", "mix") else None)
        eng = engines[r]
        for k, v in one.items():
            eng.feed(k, v)
        for st in eng.steps:                        # this clip's share of the per-GPU loss normaliser
            if isinstance(st, LossStep):
                st.scale = float(r) / float(R)
        eng.forward()
        eng.backward()
        torch.cuda.synchronize()
        for n in BLOBS:
            got = eng.fetch(n)
            want = fwd8[n]
            per = want.shape[0] // CLIPS                  # (grouped non-local sums hold N * G rows)
            d = np.abs(got.astype(np.float64) - want[i * per:(i + 1) * per]).max()
            worst_fwd = max(worst_fwd, float(d))
            assert d == 0.0, "%s of clip %d differs between the 8-clip and the 1-clip plan by %g" % (n, i, d)
        for n in ROW_BLOBS:
            got = eng.fetch(n)
            d = np.abs(got.astype(np.float64) - fwd8[n][sl]).max()
            assert d == 0.0, "%s rows of clip %d differ by %g" % (n, i, d)
        loss_sum += float(eng.fetch("loss").reshape(-1)[0])
        for n in gsum:
            gsum[n] += eng.fetch_grad(n).astype(np.float64)
    assert abs(loss_sum - loss8) < 2e-6 * abs(loss8), (loss_sum, loss8)
    gmax = max(np.linalg.norm(g) for g in gsum.values())
    # a bias on phi shifts every logit of a softmax row equally: its gradient is mathematically zero and what the engines
    # hold is rounding noise (absolute check, as in test_model_gpu.py)
    zero = [n for n in gsum if n.endswith("_phi_b")]
    for n in zero:
        assert np.linalg.norm(grad8[n]) < 1e-3 * gmax and np.linalg.norm(gsum[n]) < 1e-3 * gmax, n
    errs = sorted(((rel(grad8[n], gsum[n]), n) for n in gsum if n not in zero and np.linalg.norm(gsum[n]) > 1e-9 * gmax),
                  reverse=True)
    e = np.array([x for x, _ in errs])
    print("[%s] 8-clip gradients vs the sum of the 1-clip gradients: median %.2e p90 %.2e max %.2e (%s)"
          % (dtype, np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], errs[0][1]))
    # fp32-storage paths: only the fp32 summation order of the split-K slabs differs.  16-bit gradient storage: a 1-clip
    # run rounds scale * (p - t) / normaliser with the scale multiplied in a different order -- last-bit differences of the
    # loss gradient that the fp16 / bf16 chain then carries (measured values in the message above)
    tol = {"split": 5e-5, "mix": 2e-3, "fp16": 2e-3, "bf16": 1e-2}[dtype]
    assert e[0] < tol, errs[:5]
