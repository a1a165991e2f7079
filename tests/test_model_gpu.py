"""End-to-end parity of the HIP engine against the CPU oracle (fp64) on the BASELINE.json
configurations at sizes the oracle finishes in seconds: forward blobs, loss and EVERY parameter
gradient.

Tolerances (relative L2 per tensor; measured values are printed with -s):
  fp32 path (exact-fp32 MFMA): forward blobs, prob and loss 1e-3 -- the north-star bar ("within
      1e-3 relative fp32"; measured 5e-7).  Parameter gradients: median 1e-3, max 5e-3.  The max is
      looser because ReLU / max-pool decisions are discrete: ONE unit of res5 (262k active units at
      this test size) whose pre-activation ties at zero differently in fp32 and fp64 moves the whole
      backward signal by 1/sqrt(262k) = 2e-3 (measured: the error appears at the first masked
      backward op and stays flat; torch-CPU fp32 vs fp64 shows the same effect when it hits a tie).
  bf16 path (bf16 storage / MFMA operands, fp32 accumulation, fp32 affinity+softmax+loss):
      activations 2e-2 (measured 6e-3), prob/loss 1e-3 (measured 8e-4 / 3e-5).  Parameter gradients
      are NOT within 1e-3 and cannot be with bf16 forward storage: oracle/bf16_budget.py re-runs the
      fp64 oracle with a bf16 rounding at the engine's storage points and attributes the error --
      backward roundings (incl. the residual-stream gradient accumulators) give p90 3e-3, the forward
      roundings (activations + weight operands: flipped ReLU / max-pool decisions, perturbed saved
      activations) give p90 4e-2..6e-2 and 0.17..0.18 on the worst tensor.  The engine is held to that
      budget per tensor (test_bf16_gradient_error_stays_within_the_forward_rounding_budget) and to the
      measured envelope here: 90th percentile 0.12, max 0.30 (conv1_w, end of the ~50-layer chain).
"""
import collections
import math

import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16, "TRAIN.CROP_SIZE", 64]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (d if d > 0 else 1.0)


def build(preset, dtype, overrides=SMALL, train=True, **engine_kw):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from vlfb import rng as vrng
    from oracle import model as om
    load_preset(preset, overrides)
    n_clips = cfg.TRAIN.BATCH_SIZE // cfg.NUM_GPUS
    split = "train" if train else "test"
    model = ModelBuilder(train=train, split=split, name=split)
    model.build_model(suffix="_" + split)
    inputs = om.synth_inputs(cfg, n_clips, "train", seed=cfg.RNG_SEED, rois_per_clip=[2, 3][:n_clips] if cfg.DATASET == "ava" else None,
                             crop=cfg.TRAIN.CROP_SIZE, frames=cfg.TRAIN.VIDEO_LENGTH)
    params = om.synth_params(cfg, seed=cfg.RNG_SEED)
    eng = Engine(model, dtype, base_seed=cfg.RNG_SEED, **engine_kw)
    sfx = "_" + split
    eng.plan(collections.OrderedDict((k + sfx, v.shape) for k, v in inputs.items()
                                     if (k + sfx) in model.input_blob_names))
    eng.feed_params(params)
    for k, v in inputs.items():
        if (k + sfx) in model.input_blob_names:
            eng.feed(k + sfx, v)
    seed_fn = lambda name: vrng.dropout_seed(cfg.RNG_SEED, name, 0)
    return cfg, model, eng, inputs, params, seed_fn


CHECK_BLOBS = ["res_conv1_bn", "pool1", "res2_2_branch2c_bn", "pool2", "nonlocal_conv3_1_sum",
               "res3_3_branch2c_bn", "nonlocal_conv4_1_sum", "res4_5_branch2c_bn", "res5_2_branch2c_bn",
               "pool5", "pred", "prob"]


@pytest.mark.parametrize("dtype", ["fp32", "split", "mix", "bf16"])
@pytest.mark.parametrize("preset", ["charades_r50_baseline", "ava_r50_lfb_nl", "charades_r50_lfb_nl"])
def test_forward_backward_matches_oracle(preset, dtype):
    from oracle import model as om
    cfg, model, eng, inputs, params, seed_fn = build(preset, dtype)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    tol_act = 1e-3 if dtype in ("fp32", "split", "mix") else 2e-2
    report = []
    for name in CHECK_BLOBS:
        if name not in blobs:
            continue
        got = eng.fetch(name)
        ref = blobs[name].detach().numpy().reshape(got.shape)
        report.append((name, rel(got, ref)))
    loss = float(eng.fetch("loss").reshape(-1)[0])
    report.append(("loss", abs(loss - float(blobs["loss"].detach())) / abs(float(blobs["loss"].detach()))))
    worst_act = max(r for _, r in report)
    print("\n[%s %s] activations:" % (preset, dtype), ", ".join("%s=%.2e" % x for x in report))
    assert set(grads) == set(eng.trainable), "trainable set differs from the oracle's"
    gmax = max(float(g.norm()) for g in grads.values())
    greport = []
    for n in eng.trainable:
        ref = grads[n].numpy()
        got = eng.fetch_grad(n)
        if np.linalg.norm(ref) < 1e-9 * gmax:
            # mathematically zero (a bias on phi shifts every logit of a row equally): absolute check
            assert np.linalg.norm(got) < 1e-4 * gmax, (n, np.linalg.norm(got))
            continue
        greport.append((n, rel(got, ref)))
    worst = sorted(greport, key=lambda x: -x[1])[:6]
    errs = np.sort([e for _, e in greport])
    med, p90 = float(np.median(errs)), float(errs[int(0.9 * (len(errs) - 1))])
    print("[%s %s] gradients: median %.2e p90 %.2e worst:" % (preset, dtype, med, p90),
          ", ".join("%s=%.2e" % x for x in worst))
    assert worst_act < tol_act, report
    out_err = dict(report)
    assert out_err["prob"] < 1e-3 and out_err["loss"] < 1e-3, report
    if dtype in ("fp32", "split", "mix"):
        # raw comparison: one ReLU / max-pool tie decided differently in fp32 and fp64 shifts every gradient upstream of
        # it (module docstring) -- regression gate only
        assert med < (2e-3 if dtype == "mix" else 1e-3) and worst[0][1] < (1e-2 if dtype == "mix" else 5e-3), (med, worst)
        # the parity claim: on the SAME branches (oracle re-evaluated with this engine's discrete decisions) every
        # parameter gradient is inside the north-star bar
        dec = eng.discrete_decisions()
        _, g2 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
        assert not dec["_missing"], sorted(dec["_missing"])
        cond = sorted(((rel(eng.fetch_grad(n), g2[n].numpy()), n) for n, _ in greport), reverse=True)
        print("[%s %s] gradients on identical ReLU / max-pool decisions: median %.2e worst %s"
              % (preset, dtype, float(np.median([x for x, _ in cond])), ["%s=%.2e" % (n, x) for x, n in cond[:4]]))
        if dtype == "mix":
            # "mix" = the split forward (same decisions, same saved activations) + an fp16 backward with every storage point
            # that was found to matter kept wider: two-term weights in DGRAD, two-term gradients on the
            # residual stream and wherever several conv DGRADs are summed, fp32 gradients + split products around the
            # non-local softmax, the whole head in fp32.  Measured at this size (round 5): median 0.8e-4 .. 1.3e-4, max
            # 8.6e-4 (the theta weights of ONE non-local block, whichever the rounding pattern hits) -- EVERY gradient inside
            # the north-star bar here too (round 4: 1.02e-3 with a 1.5e-3 gate); the benchmarked size has more margin
            # (test_full_size_clip_matches_oracle: 7.1e-4 / 4.5e-4).
            ce = np.sort([x for x, _ in cond])
            assert float(np.median(ce)) < 2.5e-4 and float(ce[int(0.9 * (len(ce) - 1))]) < 6e-4 and cond[0][0] < 1e-3, cond[:5]
        else:
            assert cond[0][0] < 1e-3, cond[:5]
    else:
        assert p90 < 0.12 and worst[0][1] < 0.30, (p90, worst)


@pytest.mark.parametrize("preset", ["charades_r50_baseline", "ava_r50_lfb_nl"])
def test_bf16_gradient_error_stays_within_the_forward_rounding_budget(preset):
    """every parameter gradient of the bf16 engine is within a small factor of what the fp64 oracle gives
    when bf16 roundings are inserted at the engine's storage points (oracle/bf16_budget.py), and the
    backward-only roundings explain less than a fifth of it (=> fp32 gradient accumulators would not help)"""
    from oracle import model as om, bf16_budget as bb
    cfg, model, eng, inputs, params, seed_fn = build(preset, "bf16")
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    budget = bb.gradient_budget(cfg, params, inputs, grads, None, seed_fn)
    bwd_only = bb.gradient_budget(cfg, params, inputs, grads, dict(bwd=True, bwd_res=True), seed_fn)
    gmax = max(float(g.norm()) for g in grads.values())
    names = [n for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
    err = {n: rel(eng.fetch_grad(n), grads[n].numpy()) for n in names}
    p90 = lambda d: float(np.sort([d[n] for n in names])[int(0.9 * (len(names) - 1))])
    ratio = sorted(((err[n] / (budget[n] + 3e-3), n) for n in names), reverse=True)
    print("\n[%s] engine p90 %.3e | budget p90 %.3e | backward-only budget p90 %.3e | worst engine/budget %s"
          % (preset, p90(err), p90(budget), p90(bwd_only), ["%s=%.2f" % (n, r) for r, n in ratio[:4]]))
    assert p90(bwd_only) < 0.2 * p90(budget) and p90(bwd_only) < 1e-2
    assert p90(err) < 3.0 * p90(budget), (p90(err), p90(budget))
    for r, n in ratio:
        assert r < 4.0, "gradient of %s: engine error %.3e, emulated budget %.3e" % (n, err[n], budget[n])


@pytest.mark.parametrize("preset", ["charades_r50_baseline", "ava_r50_lfb_nl"])
def test_fp16_path_matches_oracle(preset):
    """fp16 storage + v_mfma_f32_16x16x32_f16 (BASELINE.json configs[4] names fp16 MFMA) with the static,
    shape-derived loss scale (a power of two, divided out by the solver): three more mantissa bits than bf16,
    so the forward-rounding budget of oracle/bf16_budget.py shrinks ~8x.  Measured: activations 8e-4, prob 1e-4,
    parameter gradients p90 1e-2..2e-2 / max 7e-2 (conv1_w)."""
    from oracle import model as om
    cfg, model, eng, inputs, params, seed_fn = build(preset, "fp16")
    assert eng.loss_scale >= 1024.0 and math.log2(eng.loss_scale) == int(math.log2(eng.loss_scale))
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    acts = []
    for name in CHECK_BLOBS:
        if name in blobs:
            got = eng.fetch(name)
            acts.append((name, rel(got, blobs[name].detach().numpy().reshape(got.shape))))
    ref_loss = float(blobs["loss"].detach())
    loss_err = abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) / abs(ref_loss)
    gmax = max(float(g.norm()) for g in grads.values())
    gerr = sorted(((rel(eng.fetch_grad(n), grads[n].numpy()), n) for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax),
                  reverse=True)
    e = np.sort([x for x, _ in gerr])
    p90 = float(e[int(0.9 * (len(e) - 1))])
    print("\n[%s fp16] activations max %.2e, prob %.2e, loss %.2e; gradients median %.2e p90 %.2e worst %s"
          % (preset, max(v for _, v in acts), dict(acts)["prob"], loss_err, float(np.median(e)), p90,
             ["%s=%.2e" % (n, x) for x, n in gerr[:4]]))
    assert max(v for _, v in acts) < 3e-3 and dict(acts)["prob"] < 1e-3 and loss_err < 1e-3, acts
    assert p90 < 3e-2 and gerr[0][0] < 0.1, (p90, gerr[:3])
    # the solver divides the loss scale out: one step moves the weights like the fp32 engine's step does
    w0 = eng.fetch_param("pred_w").copy()
    g0 = eng.fetch_grad("pred_w").astype(np.float64)
    eng.sgd_step(0.01)
    torch.cuda.synchronize()
    wd, mu = float(cfg.SOLVER.WEIGHT_DECAY), float(cfg.SOLVER.MOMENTUM)
    want = w0 - (1 + mu) * 0.01 * (g0 + wd * w0)
    assert rel(eng.fetch_param("pred_w"), want) < 1e-5


def test_fp16_scaled_activation_gradients_are_fetched_unscaled():
    """fp16 keeps the theta / phi gradients of a non-local block times a power of two (Blob.grad_scale) so that they
    stay in the fp16 normal range; Engine.fetch('<blob>_grad') has to divide that out again (and the loss scale)"""
    got = {}
    for dtype in ("fp32", "fp16"):
        cfg, model, eng, inputs, params, seed_fn = build("charades_r50_baseline", dtype)
        eng.forward()
        eng.backward()
        torch.cuda.synchronize()
        if dtype == "fp16":
            assert eng.env["nonlocal_conv4_1_theta"].root.grad_scale > 1.0 and eng.loss_scale > 1.0
        got[dtype] = {n: eng.fetch(n + "_grad") for n in ("nonlocal_conv4_1_theta", "nonlocal_conv4_1_phi", "nonlocal_conv4_1_g")}
        del eng
    for n in got["fp32"]:
        assert rel(got["fp16"][n], got["fp32"][n]) < 5e-2, (n, rel(got["fp16"][n], got["fp32"][n]))


def test_c5_r101_64_frame_clip_fp16_full_size():
    """BASELINE.json configs[4] / SURVEY 8d C5 at real size: ava_r101_lfb_nl_3l (R101-I3D-NL, 23 res4 blocks,
    3-layer FBO-NL), ONE 64-frame 224^2 clip (pool stride 32, 8 non-local groups in res3, res4 affinities
    6272 x 1568), fp16 MFMA path, every output and parameter gradient against the fp64 oracle"""
    import os
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ov = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 64, "TEST.VIDEO_LENGTH", 64, "TRAIN.CROP_SIZE", 224]
    cfg, model, eng, inputs, params, seed_fn = build("ava_r101_lfb_nl_3l", "fp16", ov)
    att = [s for s in eng.steps if type(s).__name__ == "AttentionStep" and not s.single]
    assert att[0].theta.shape[0] == 8 and (att[-1].L1, att[-1].L2) == (6272, 1568)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    for name in ("res5_2_branch2c_bn", "pool5", "prob"):
        got = eng.fetch(name)
        assert rel(got, blobs[name].detach().numpy().reshape(got.shape)) < (1e-3 if name == "prob" else 3e-3), name
    ref_loss = float(blobs["loss"].detach())
    assert abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) < 1e-3 * abs(ref_loss)
    gmax = max(float(g.norm()) for g in grads.values())
    ge = sorted(((rel(eng.fetch_grad(n), grads[n].numpy()), n, float(grads[n].norm())) for n in eng.trainable
                 if float(grads[n].norm()) > 1e-9 * gmax), reverse=True)
    e = np.sort([x for x, _, _ in ge])
    print("\n[C5 r101 64f fp16] %d gradients: median %.2e p90 %.2e max %.2e; worst %s" % (
        len(e), np.median(e), e[int(0.9 * (len(e) - 1))], e[-1], ["%s=%.2e (|g| %.1e)" % (n, x, g) for x, n, g in ge[:8]]))
    # measured: median 2.3e-2, p90 3.7e-2; max 0.24 on the theta / phi weights of ONE res4 non-local block (13) whose
    # synthetic attention is peaky -- the same three tensors are the worst ones of the bf16 path too (unchanged by the
    # fp16 range scalings, i.e. forward-rounding sensitivity, not underflow)
    assert e[int(0.9 * (len(e) - 1))] < 5e-2 and e[-1] < 0.35


FULL = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 32, "TRAIN.CROP_SIZE", 224]


@pytest.mark.parametrize("preset", ["charades_r50_baseline", "ava_r50_lfb_nl"])
def test_full_size_clip_matches_oracle(preset):
    """ONE clip at the benchmarked size (32 x 224^2: M = 401 408 / 100 352 / 12 544 / 3 136 rows per stage,
    K up to 6144, 3136 x 784 non-local affinities, the 256-row pipelined kernels, split wgrads with
    hundreds of slabs) against the fp64 oracle: forward blobs, loss and EVERY parameter gradient, on the
    fp32 parity path and on the bf16 path that bench.py times.  The per-tensor table goes to
    $VLFB_PARITY_DIR (committed under profiles/)."""
    import os
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = None
    lines = []
    summary = {}
    for dtype in ("fp32", "split", "mix", "fp16", "bf16"):
        cfg, model, eng, inputs, params, seed_fn = build(preset, dtype, FULL)
        eng.forward()
        eng.backward()
        torch.cuda.synchronize()
        if ref is None:
            ref = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
        blobs, grads = ref
        acts = []
        for name in CHECK_BLOBS:
            if name in blobs:
                got = eng.fetch(name)
                acts.append((name, rel(got, blobs[name].detach().numpy().reshape(got.shape))))
        ref_loss = float(blobs["loss"].detach())
        acts.append(("loss", abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) / abs(ref_loss)))
        gmax = max(float(g.norm()) for g in grads.values())
        names = [n for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
        got_g = {n: eng.fetch_grad(n) for n in names}
        gerr = [(n, rel(got_g[n], grads[n].numpy())) for n in names]
        e = np.sort([x for _, x in gerr])
        med, p90, mx = float(np.median(e)), float(e[int(0.9 * (len(e) - 1))]), float(e[-1])
        lines.append("== %s %s, 1 clip 32x224x224: activations/outputs (relative L2 vs fp64 oracle)" % (preset, dtype))
        lines += ["  %-28s %.3e" % x for x in acts]
        lines.append("   parameter gradients: median %.3e  p90 %.3e  max %.3e (%d tensors)" % (med, p90, mx, len(gerr)))
        summary[dtype] = {"activations_max": max(v for k, v in acts if k not in ("loss",)), "prob": dict(acts)["prob"],
                          "loss": dict(acts)["loss"], "grad_raw": {"median": med, "p90": p90, "max": mx}}
        cond = None
        if dtype not in ("bf16", "fp16"):
            # The same comparison on IDENTICAL BRANCHES: the fp64 oracle re-evaluated with this engine's ReLU sign patterns
            # and max-pool selections (a pre-activation within fp32 rounding of zero is decided either way; ONE such unit of
            # res5 under AVA's sparse loss gradient moves every upstream gradient by ~1e-3)
            dec = eng.discrete_decisions()
            nflip = sum(int(((blobs[n].detach().numpy() > 0) != m).sum()) for n, m in dec["relu"].items() if n in blobs)
            _, g2 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
            assert not dec["_missing"], "decision sites the engine did not cover: %r" % sorted(dec["_missing"])
            cond = [(n, rel(got_g[n], g2[n].numpy())) for n in names]
            ce = np.sort([x for _, x in cond])
            lines.append("   same parameter gradients against the oracle evaluated on THIS engine's ReLU / max-pool decisions "
                         "(%d ReLU units of the checked blobs decided differently): median %.3e  p90 %.3e  max %.3e"
                         % (nflip, float(np.median(ce)), float(ce[int(0.9 * (len(ce) - 1))]), float(ce[-1])))
            cd = dict(cond)
            summary[dtype]["grad_identical_decisions"] = {"median": float(np.median(ce)), "p90": float(ce[int(0.9 * (len(ce) - 1))]),
                                                          "max": float(ce[-1]), "worst": max(cond, key=lambda x: x[1])[0],
                                                          "relu_units_decided_differently": nflip}
            lines += ["  %-44s %.3e   (same decisions: %.3e)" % (n, x, cd[n]) for n, x in sorted(gerr, key=lambda x: -x[1])]
        else:
            lines += ["  %-44s %.3e" % x for x in sorted(gerr, key=lambda x: -x[1])]
        print("\n".join(lines[-(len(gerr) + len(acts) + 3):][:len(acts) + 9]))
        a = dict(acts)
        if dtype == "mix":
            assert max(a.values()) < 1e-3, acts
            # split forward + fp16 backward: EVERY gradient inside the north-star bar at the benchmarked size, with margin --
            # measured (round 5) median 1.06e-4 (AVA) / 5.4e-5 (Charades), p90 1.8e-4 / 1.9e-4, max 7.1e-4 / 4.5e-4 (conv1_w
            # both: the end of the chain, and a sum of random-sign terms on the noise clip, so its relative error IS the
            # element-wise error of the trunk gradient); round 4: 4.0e-4 / 6.2e-4 / 9.7e-4.  Gate = measured max x 1.15.
            # Raw median 1.6e-3 (AVA: ties, as on the split path).
            assert float(np.median(ce)) < 2e-4 and float(ce[int(0.9 * (len(ce) - 1))]) < 3.5e-4 and float(ce[-1]) < 8.2e-4, \
                sorted(cond, key=lambda x: -x[1])[:5]
            assert med < 3e-3 and mx < 2e-2, (med, mx)
        elif dtype in ("fp32", "split"):
            assert max(a.values()) < 1e-3, acts
            # arithmetic parity (identical branches): EVERY gradient inside the north-star bar
            assert float(ce[-1]) < 1e-3, sorted(cond, key=lambda x: -x[1])[:5]
            # raw comparison: tie-limited (measured median 2.6e-5 / 3.2e-5 on the fp32 path, 1.6e-3 / 1.4e-4 on the split path
            # whose forward decides 307 instead of 20 units differently; max 1.0e-2, conv1_w) -- a regression gate at twice the
            # measured values, not the parity claim
            assert med < 3e-3 and mx < 2e-2, (med, mx)
        else:
            assert max(v for k, v in a.items() if k not in ("prob", "loss")) < 2e-2, acts
            assert a["prob"] < 1e-3 and a["loss"] < 1e-3, acts
            # measured: charades p90 5e-2 / max 0.24, ava p90 6.8e-2 / max 0.32 (conv1_w, the end of the chain;
            # 401 408-row tensors amplify the forward-rounding budget of oracle/bf16_budget.py a little further)
            assert p90 < 0.10 and mx < 0.35, (p90, mx)
        del eng
        torch.cuda.empty_cache()
    out = os.environ.get("VLFB_PARITY_DIR")
    if out:
        import hashlib, glob, json
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_fullsize_%s.txt" % preset), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        # machine-readable summary, stamped with the hash of the kernel sources it was measured on (bench.py quotes it
        # next to each path's clips/s only while that hash is the current one)
        # (bench.parity_source_hash: the HIP sources + lib/vlfb/engine.py + hip.py -- everything that decides the arithmetic)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import bench
        with open(os.path.join(out, "parity_fullsize_%s.json" % preset), "w") as fh:
            json.dump({"preset": preset, "size": "1 clip 32x224x224", "metric": "relative L2 vs the fp64 oracle, per tensor",
                       "source_sha256": bench.parity_source_hash(), "paths": summary}, fh, indent=1)


@pytest.mark.parametrize("preset", ["epic_verb_r50_lfb_nl", "epic_noun_r50_lfb_nl", "epic_verb_r50_baseline"])
def test_epic_single_label_head_train_and_test(preset):
    """EPIC-Kitchens models: Softmax / SoftmaxWithLoss head (resnet_video.py:339-350), integer class labels, no
    res5 dilation, 40 / 120-feature banks; train-mode forward + every parameter gradient and the test-mode
    probabilities against the fp64 oracle on the fp32 path"""
    from oracle import model as om
    cfg, model, eng, inputs, params, seed_fn = build(preset, "fp32")
    assert not cfg.MODEL.MULTI_LABEL and inputs["labels"].shape == (2,)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    got = eng.fetch("prob")
    assert abs(got.sum(axis=1) - 1).max() < 1e-5
    assert rel(got, blobs["prob"].detach().numpy()) < 1e-3
    ref_loss = float(blobs["loss"].detach())
    assert abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) < 1e-3 * abs(ref_loss)
    assert set(grads) == set(eng.trainable)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = [rel(eng.fetch_grad(n), grads[n].numpy()) for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
    # max: one ReLU / max-pool tie decided differently in fp32 and fp64 shifts every gradient upstream of it
    # (module docstring); with a single-label loss on 2 clips fewer units carry the signal (measured 7.8e-3 on
    # conv1_w with a median of 3e-6)
    assert np.median(errs) < 1e-3 and max(errs) < 2e-2, (np.median(errs), max(errs))
    # test mode: Softmax only
    cfg2, model2, eng2, inputs2, params2, _ = build(preset, "fp32", SMALL + ["TEST.BATCH_SIZE", 2, "TEST.VIDEO_LENGTH", 16,
                                                                             "TEST.CROP_SIZE", 64], train=False)
    eng2.forward()
    torch.cuda.synchronize()
    tb, _ = om.run(cfg2, {k: v for k, v in params2.items() if k in eng2.param_views}, inputs2, "test", torch.float64, False, None)
    assert rel(eng2.fetch("prob"), tb["prob"].numpy()) < 1e-3


def _survey_literal_params(cfg, seed=2):
    """SURVEY.md 8(d), literally: MSRA normal for `*_w` convs of the backbone, N(0, 0.01) for non-local / FC / FBO
    weights (incl. the normally zero-initialised output convs), affine s ~ U(0.5, 1.5), b ~ N(0, 0.1) -- WITHOUT the
    smaller residual-exit gains and theta/phi boosts oracle.synth_params applies to keep 16 un-normalised blocks O(1)"""
    from oracle import model as om
    gen = np.random.default_rng(seed)
    out = collections.OrderedDict()
    for name, sp in om.param_spec(cfg).items():
        shape, kind = sp["shape"], sp["kind"]
        if kind == "msra":
            v = gen.standard_normal(shape) * math.sqrt(2.0 / (shape[0] * int(np.prod(shape[2:]))))
        elif kind in ("gauss", "nl_out", "fbo_out"):
            v = gen.standard_normal(shape) * 0.01
        elif kind == "zero_bias":
            v = gen.standard_normal(shape) * 0.01
        elif kind == "affine_s":
            v = gen.uniform(0.5, 1.5, shape)
        else:
            v = gen.standard_normal(shape) * 0.1
        out[name] = v.astype(np.float32)
    return out


def test_survey_literal_weight_recipe_fp32():
    """the parity bar does not lean on the tamed synthetic weights: with the survey's literal recipe (activations grow
    through the un-normalised residual stack) the fp32 path still matches the fp64 oracle on outputs and gradients"""
    from oracle import model as om
    cfg, model, eng, inputs, _, seed_fn = build("charades_r50_baseline", "fp32")
    params = _survey_literal_params(cfg)
    eng.feed_params(params)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    for name in ("res5_2_branch2c_bn", "pool5", "pred", "prob"):
        got = eng.fetch(name)
        assert rel(got, blobs[name].detach().numpy().reshape(got.shape)) < 1e-3, name
    ref_loss = float(blobs["loss"].detach())
    assert abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) < 1e-3 * abs(ref_loss)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = [rel(eng.fetch_grad(n), grads[n].numpy()) for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
    assert np.median(errs) < 1e-3 and max(errs) < 2e-2, (np.median(errs), max(errs))


VARIANTS = {
    "charades_r50_lfb_avg": ("charades_r50_lfb_avg", SMALL),
    "ava_r50_lfb_max": ("ava_r50_lfb_max", SMALL),
    "ava_r101_lfb_nl_3l": ("ava_r101_lfb_nl_3l", SMALL),
    # the LFB model with the backbone trained too (the YAML freezes it)
    "charades_r50_lfb_nl_unfrozen": ("charades_r50_lfb_nl", SMALL + ["MODEL.FREEZE_BACKBONE", False]),
    # 64-frame clips (pool stride 32, 8 groups of 4 frames in the res3 non-local blocks)
    "ava_r101_lfb_nl_3l_64f": ("ava_r101_lfb_nl_3l", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 64,
                                                      "TRAIN.CROP_SIZE", 64]),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_other_heads_and_depths_match_oracle_fp32(variant):
    """FBO-avg / FBO-max heads (lfb_helper.py:106-127), the R101 / 3-layer FBO-NL variant, the unfrozen
    LFB model and 64-frame clips"""
    from oracle import model as om
    preset, overrides = VARIANTS[variant]
    cfg, model, eng, inputs, params, seed_fn = build(preset, "fp32", overrides)
    if variant.endswith("64f"):
        groups = [s for s in eng.steps if type(s).__name__ == "AttentionStep"][0]
        assert groups.theta.shape[0] == 8          # (T/2)/4 groups per clip
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    for name in ("pool5", "prob"):
        got = eng.fetch(name)
        assert rel(got, blobs[name].detach().numpy().reshape(got.shape)) < 1e-3, name
    ref_loss = float(blobs["loss"].detach())
    assert abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) < 1e-3 * abs(ref_loss)
    assert set(grads) == set(eng.trainable)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = [rel(eng.fetch_grad(n), grads[n].numpy()) for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
    assert np.median(errs) < 1e-3 and max(errs) < 5e-3, (np.median(errs), max(errs))


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_other_heads_and_depths_match_oracle_mix(variant):
    """the same graphs on the default `mix` dtype, on identical ReLU / max-pool / RoI-bin decisions: every head variant goes
    through Engine._plan_head_f32 / _plan_fbo_f32 (an FBO-avg / -max head has no fp32 FBO branch; three layers; the
    non-pre-activation Charades head; R101; 64-frame clips with 8 non-local groups) -- outputs within 1e-3, every parameter
    gradient inside the bar the small test size allows (measured max 9.7e-4 -- the theta weights of one non-local block, R101 included; 1.2e-3 gate, as the 2-clip 16 x 64^2 size
    has ~30x fewer positions to average over than the benchmarked one, where the gate is 8.2e-4)"""
    from oracle import model as om
    preset, overrides = VARIANTS[variant]
    cfg, model, eng, inputs, params, seed_fn = build(preset, "mix", overrides)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    dec = eng.discrete_decisions()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
    assert not dec["_missing"], sorted(dec["_missing"])
    for name in ("pool5", "prob"):
        got = eng.fetch(name)
        assert rel(got, blobs[name].detach().numpy().reshape(got.shape)) < 1e-3, name
    assert set(grads) == set(eng.trainable)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = sorted(((rel(eng.fetch_grad(n), grads[n].numpy()), n) for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax),
                  reverse=True)
    e = np.array([x for x, _ in errs])
    print("\n[%s mix] identical decisions: median %.2e p90 %.2e worst %s" % (
        variant, np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], ["%s=%.2e" % (n, x) for x, n in errs[:3]]))
    assert np.median(e) < 3e-4 and errs[0][0] < 1.2e-3, errs[:4]


def _shipped_not_yet_covered():
    from vlfb.presets import PRESETS
    covered = {v[0] for v in VARIANTS.values() if v[1] is SMALL} | {"ava_r50_lfb_nl", "charades_r50_baseline"}
    return sorted(set(PRESETS) - covered)


# one of each kind the variants above do not reach: AVA baseline, R101 with an avg head, the undilated R101 Charades model with
# FBO-NL, a max head on Charades, the EPIC noun model (bank window 120) and an EPIC verb avg head
ORACLE_SUBSET = ["ava_r50_baseline", "ava_r101_lfb_avg", "charades_r101_lfb_nl", "charades_r50_lfb_max", "epic_noun_r50_lfb_nl",
                 "epic_verb_r50_lfb_avg"]
SMALL8 = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64]


@pytest.mark.parametrize("preset", ORACLE_SUBSET)
def test_more_shipped_configs_match_the_oracle_fp32(preset):
    """forward + backward on the exact-fp32 path against the fp64 oracle -- outputs, loss and every parameter gradient.  The
    gradients are compared RAW, i.e. including the ReLU / max-pool ties that fp64 and fp32 decide differently:
    at 8 frames of 64 x 64 one flipped unit of res5 moves every upstream gradient, and the sparser the loss gradient the
    more (measured: median 1.4e-3 / max 6.6e-3 on the AVA baseline, 1.9e-3 / 6.8e-3 on the R101 avg head, < 1e-3 medians on
    the four clip-level models).  The bars are therefore the raw bars; what the arithmetic itself achieves is measured on
    identical decisions at full size (test_full_size_clip_matches_oracle).  The fp64 oracle backward of an R101 takes ~25 s
    on the box's host cores, hence a subset at 8 frames; every other config runs in the test below."""
    from oracle import model as om
    assert set(ORACLE_SUBSET) <= set(_shipped_not_yet_covered())
    cfg, model, eng, inputs, params, seed_fn = build(preset, "fp32", SMALL8)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    for name in ("pool5", "prob"):
        got = eng.fetch(name)
        assert rel(got, blobs[name].detach().numpy().reshape(got.shape)) < 1e-3, name
    ref_loss = float(blobs["loss"].detach())
    assert abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) < 1e-3 * abs(ref_loss)
    assert set(grads) == set(eng.trainable)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = [rel(eng.fetch_grad(n), grads[n].numpy()) for n in eng.trainable if float(grads[n].norm()) > 1e-9 * gmax]
    assert np.median(errs) < 5e-3 and max(errs) < 2e-2, (np.median(errs), max(errs))


@pytest.mark.parametrize("dtype", ["mix", "fp16"])
def test_every_shipped_config_trains_on_the_default_path(dtype):
    """all 26 presets (= the reference's configs/*.yaml, tests/test_ref_graph.py) take two training steps on the path the
    benchmark runs by default (`mix`: every head variant -- RoI / basic, FBO-NL 2 / 3 layers pre-act or not, FBO avg / max,
    EPIC softmax -- through the fp32 head planning) and on the fp16 throughput path: finite loss, a finite non-zero gradient for every trainable parameter, the loss of the second step
    differs from the first (the solver moved the weights).  No oracle here (see above for cost): the arithmetic of each
    step kind is held to the oracle by the tests above; this one is about every shipped graph reaching the kernels."""
    from vlfb.presets import PRESETS
    presets = sorted(PRESETS)
    if dtype != "mix":        # the throughput sub-path: one preset per dataset / head kind / depth (the suite's wall time)
        presets = ["ava_r50_baseline", "ava_r50_lfb_nl", "ava_r50_lfb_max", "ava_r101_lfb_nl_3l", "charades_r50_baseline",
                   "charades_r50_lfb_avg", "charades_r101_lfb_nl", "epic_verb_r50_lfb_nl", "epic_noun_r50_baseline"]
        assert set(presets) <= set(PRESETS)
    for preset in presets:
        cfg, model, eng, inputs, params, seed_fn = build(preset, dtype, SMALL8)
        eng.forward()
        eng.backward()
        torch.cuda.synchronize()
        loss0 = float(eng.fetch("loss").reshape(-1)[0])
        assert np.isfinite(loss0) and loss0 > 0, (preset, loss0)
        assert len(eng.trainable) > 0
        for n in eng.trainable:
            g = eng.fetch_grad(n)
            assert np.isfinite(g).all(), (preset, n)
        norms = [float(np.linalg.norm(eng.fetch_grad(n))) for n in eng.trainable]
        assert max(norms) > 0, preset
        eng.train_step(1e-3)
        eng.forward()
        torch.cuda.synchronize()
        loss1 = float(eng.fetch("loss").reshape(-1)[0])
        assert np.isfinite(loss1) and loss1 != loss0, (preset, loss0, loss1)
        del eng
        torch.cuda.empty_cache()


def test_a_batch_padded_with_ignored_rows_is_the_same_batch():
    """How a ragged AVA batch runs on ONE plan: the engine is planned for a fixed number of RoI rows, and a step with fewer
    RoIs pads `proposals` with any box of an existing clip, `labels` with -1 (ignored by SigmoidCrossEntropyLoss: no loss
    term, no gradient, not counted by the normaliser -- Caffe2's semantics, restated in csrc/vlfb_head.hip and oracle/model.py)
    and `lfb` with zeros.  Loss, the probabilities of the real rows and every parameter gradient are those of the unpadded
    batch (in the fp64 oracle exactly; here up to the fp32 summation order of the head GEMMs, whose row count changed)."""
    from vlfb.engine import Engine
    cfg, model, eng, inputs, params, seed_fn = build("ava_r50_lfb_nl", "fp32", SMALL8)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    loss0 = float(eng.fetch("loss").reshape(-1)[0])
    prob0 = eng.fetch("prob")
    grads0 = {n: eng.fetch_grad(n) for n in eng.trainable}
    del eng
    torch.cuda.empty_cache()
    extra, R = 3, inputs["proposals"].shape[0]
    padded = dict(inputs)
    padded["proposals"] = np.concatenate([inputs["proposals"], np.zeros((extra, 5), inputs["proposals"].dtype)])
    padded["labels"] = np.concatenate([inputs["labels"], -np.ones((extra, inputs["labels"].shape[1]), inputs["labels"].dtype)])
    padded["lfb"] = np.concatenate([inputs["lfb"], np.zeros((extra,) + inputs["lfb"].shape[1:], inputs["lfb"].dtype)])
    eng = Engine(model, "fp32", base_seed=cfg.RNG_SEED)
    eng.plan(collections.OrderedDict((k + "_train", v.shape) for k, v in padded.items() if (k + "_train") in model.input_blob_names))
    eng.feed_params(params)
    for k, v in padded.items():
        if (k + "_train") in model.input_blob_names:
            eng.feed(k + "_train", v)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    loss1 = float(eng.fetch("loss").reshape(-1)[0])
    assert abs(loss1 - loss0) < 1e-6 * abs(loss0), (loss0, loss1)
    prob1 = eng.fetch("prob")
    assert prob1.shape[0] == R + extra and rel(prob1[:R], prob0) < 1e-6
    gmax = max(float(np.linalg.norm(g)) for g in grads0.values())
    for n, g0 in grads0.items():
        g1 = eng.fetch_grad(n)
        assert np.isfinite(g1).all(), n
        assert np.linalg.norm(g1.astype(np.float64) - g0) <= 1e-5 * np.linalg.norm(g0) + 1e-7 * gmax, \
            (n, float(np.linalg.norm(g1.astype(np.float64) - g0)), float(np.linalg.norm(g0)))
    # Engine.feed pads by itself when a RoI input has fewer rows than the plan: feeding the UNPADDED arrays is the same step
    grads1 = {n: eng.fetch_grad(n) for n in grads0}
    for k in ("proposals", "labels", "lfb"):
        eng.feed(k + "_train", inputs[k])
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    assert float(eng.fetch("loss").reshape(-1)[0]) == loss1
    for n in grads0:
        assert np.array_equal(eng.fetch_grad(n), grads1[n]), n
    with pytest.raises(AssertionError, match="planned"):
        eng.feed("proposals_train", np.zeros((R + extra + 1, 5), np.float32))       # more rows than the plan holds


def test_roi_head_integer_decisions_are_bit_exact():
    """RoIAlign batch index / sampling grid / bilinear corners inside the full AVA model"""
    from oracle import model as om
    from oracle.roi_align import roi_align_loop
    from vlfb.engine import RoiAlignMaxStep
    cfg, model, eng, inputs, params, seed_fn = build("ava_r50_lfb_nl", "fp32", debug_roi=True)
    eng.forward()
    torch.cuda.synchronize()
    step = [s for s in eng.steps if isinstance(s, RoiAlignMaxStep)][0]
    feat = eng.fetch("blob_pooled_4d")
    _, dbg = roi_align_loop(feat.astype(np.float32), inputs["proposals"], 7, 1.0 / 16)
    got = step.dbg.cpu().numpy().reshape(dbg.shape)
    assert np.array_equal(got, dbg)


def test_sgd_step_matches_reference_update_rule():
    """WeightedSum + MomentumSGDUpdate(nesterov) (model_builder_video.py:348-389) over two steps"""
    from oracle import model as om
    cfg, model, eng, inputs, params, seed_fn = build("charades_r50_baseline", "fp32")
    lr, wd, mu = 0.02, float(cfg.SOLVER.WEIGHT_DECAY), float(cfg.SOLVER.MOMENTUM)
    names = ["pred_w", "res5_2_branch2c_w", "conv1_w", "nonlocal_conv4_1_theta_b"]
    p0 = {n: eng.fetch_param(n).astype(np.float64) for n in names}
    eng.forward(); eng.backward()
    g0 = {n: eng.fetch_grad(n).astype(np.float64) for n in names}
    eng.sgd_step(lr)
    torch.cuda.synchronize()
    for n in names:
        g = g0[n] + wd * p0[n]
        m = lr * g
        want = p0[n] - ((1 + mu) * m)
        assert rel(eng.fetch_param(n), want) < 1e-6, n
        assert rel(eng.fetch_momentum(n), m) < 1e-5, n
    # the operand copies were refreshed: a second forward sees the new weights
    loss0 = float(eng.fetch("loss").reshape(-1)[0])
    eng.forward()
    torch.cuda.synchronize()
    loss1 = float(eng.fetch("loss").reshape(-1)[0])
    assert loss1 != loss0 and np.isfinite(loss1)


def test_test_mode_forward_and_lfb_inference():
    """test split (no dropout, Sigmoid only) and lfb_infer_only (stops at the head pool)"""
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from oracle import model as om
    ov = ["NUM_GPUS", 1, "TEST.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16, "TEST.VIDEO_LENGTH", 16,
          "TEST.CROP_SIZE", 64, "TRAIN.CROP_SIZE", 64]
    load_preset("ava_r50_lfb_nl", ov)
    inputs = om.synth_inputs(cfg, 2, "test", seed=3, rois_per_clip=[1, 2], crop=64, frames=16)
    params = om.synth_params(cfg, seed=3)
    for infer_only in (False, True):
        model = ModelBuilder(train=False, split="test", name="test")
        model.build_model(suffix="_test", lfb_infer_only=infer_only)
        eng = Engine(model, "fp32")
        names = [n for n in model.input_blob_names]
        eng.plan(collections.OrderedDict((n, inputs[n[:-5]].shape) for n in names))
        eng.feed_params({k: v for k, v in params.items() if k in eng.param_views})
        for n in names:
            eng.feed(n, inputs[n[:-5]])
        eng.forward()
        torch.cuda.synchronize()
        blobs, _ = om.run(cfg, {k: v for k, v in params.items() if k in eng.param_views}, inputs, "test",
                          torch.float64, False, None, lfb_infer_only=infer_only)
        key = "box_pooled" if infer_only else "prob"
        got = eng.fetch(key)
        assert rel(got, blobs[key].numpy().reshape(got.shape)) < 1e-3


def test_test_crop_256_shapes_nonlocal_4096x1024():
    """TEST.CROP_SIZE 256: spatial x8/7, so the grouped res3 non-local block
    attends 4096 x 1024 positions and the head pool is 16x16; forward parity in test mode"""
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from oracle import model as om
    load_preset("charades_r50_baseline", ["NUM_GPUS", 1, "TEST.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 8,
                                          "TEST.VIDEO_LENGTH", 8, "TEST.CROP_SIZE", 256])
    inputs = om.synth_inputs(cfg, 1, "test", seed=4, crop=256, frames=8)
    params = om.synth_params(cfg, seed=4)
    model = ModelBuilder(train=False, split="test", name="test")
    model.build_model(suffix="_test")
    eng = Engine(model, "fp32")
    names = list(model.input_blob_names)
    eng.plan(collections.OrderedDict((n, inputs[n[:-5]].shape) for n in names))
    att = [s for s in eng.steps if type(s).__name__ == "AttentionStep"]
    # 8 frames -> 4 at res3 -> one group of 4 frames: 4*32*32 queries x 4*16*16 keys
    assert (att[0].theta.shape[2], att[0].phi.shape[2]) == (4096, 1024)
    eng.feed_params({k: v for k, v in params.items() if k in eng.param_views})
    for n in names:
        eng.feed(n, inputs[n[:-5]])
    eng.forward()
    torch.cuda.synchronize()
    blobs, _ = om.run(cfg, {k: v for k, v in params.items() if k in eng.param_views}, inputs, "test",
                      torch.float64, False, None)
    for key in ("prob", "pool5"):
        got = eng.fetch(key)
        assert rel(got, blobs[key].numpy().reshape(got.shape)) < 1e-3, key


@pytest.mark.parametrize("dtype", ["fp16", "mix"])
def test_a_training_step_is_reproducible_bit_for_bit(dtype):
    """SURVEY.md 8b 'determinism': no reduction of the step depends on the order in which workgroups finish -- split-K slabs,
    bias column sums and the RoIAlign backward (the reference scatters with atomics; overlapping RoIs of a clip hit the same
    pixels) are all folded in a fixed order.  Two engines built from the same seed take three steps each: every parameter
    and every momentum buffer must be identical, bit for bit."""
    states = []
    for run in range(2):
        cfg, model, eng, inputs, params, seed_fn = build("ava_r50_lfb_nl", dtype)
        for it in range(3):
            eng.train_step(0.02)
        torch.cuda.synchronize()
        states.append((eng.flat_param.clone(), eng.flat_mom.clone(), eng.flat_grad.clone()))
        del eng
    for a, b, what in zip(states[0], states[1], ("parameters", "momentum", "gradients")):
        assert torch.equal(a, b), "%s differ between two identical runs (%d of %d words)" % (
            what, int((a != b).sum()), a.numel())


@pytest.mark.parametrize("dtype", ["fp32", "split", "mix", "fp16"])
def test_dot_product_nonlocal_variant(dtype):
    """NONLOCAL.USE_SOFTMAX False (reference nonlocal_helper.py:107-119; no shipped yaml sets it): the affinity divided by the
    number of keys instead of the softmax.  Forward blobs, loss and every parameter gradient against the oracle.  The bench's
    weight recipe (trained-BatchNorm-like gains) is used: without the softmax's normalisation the blocks square the
    activation scale, and the oracle recipe's unit gains overflow fp64 after five blocks."""
    from oracle import model as om
    from vlfb import synth
    cfg, model, eng, inputs, params, seed_fn = build("charades_r50_baseline", dtype, SMALL + ["NONLOCAL.USE_SOFTMAX", False])
    params = synth.params(model, seed=cfg.RNG_SEED)
    eng.feed_params(params)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    tol = {"fp32": 1e-5, "split": 1e-4, "mix": 1e-4, "fp16": 5e-3}[dtype]
    for name in ("nonlocal_conv3_1_sum", "nonlocal_conv4_1_affinity_sc", "res4_5_branch2c_bn", "res5_2_branch2c_bn", "prob"):
        got = eng.fetch(name)
        e = rel(got, blobs[name].detach().numpy().reshape(got.shape))
        assert e < tol, (name, e)
    ref_loss = float(blobs["loss"].detach())
    assert abs(float(eng.fetch("loss").reshape(-1)[0]) - ref_loss) < 1e-3 * abs(ref_loss)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = sorted(((rel(eng.fetch_grad(n), grads[n].numpy()), n) for n in eng.trainable
                   if float(grads[n].norm()) > 1e-9 * gmax), reverse=True)
    e = np.array([x for x, _ in errs])
    print("\n[dot-product NL %s] gradients median %.2e p90 %.2e max %.2e (%s)" % (
        dtype, np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], errs[0][1]))
    # raw comparison (ties: module docstring); measured median / max: fp32 1.5e-4 / 1.3e-3, split 3.3e-4 / 5.0e-3,
    # mix 4.1e-4 / 5.1e-3, fp16 5.2e-3 / 6.1e-2 (conv1_w every time)
    gtol = {"fp32": (1e-3, 5e-3), "split": (1e-3, 1e-2), "mix": (2e-3, 1e-2), "fp16": (3e-2, 0.3)}[dtype]
    assert np.median(e) < gtol[0] and e[0] < gtol[1], errs[:5]


@pytest.mark.parametrize("dtype", ["fp32", "split", "mix", "fp16", "bf16"])
def test_grouped_convolution_model(dtype):
    """RESNETS.NUM_GROUPS 2 x WIDTH_PER_GROUP 32 (grouped 1x3x3 convs in every bottleneck, resnet_helper.py:56-63; no shipped
    yaml): forward blobs, loss and every parameter gradient against the oracle (torch's grouped conv3d in fp64)"""
    from oracle import model as om
    ov = SMALL + ["RESNETS.NUM_GROUPS", 2, "RESNETS.WIDTH_PER_GROUP", 32]
    cfg, model, eng, inputs, params, seed_fn = build("charades_r50_baseline", dtype, ov)
    assert params["res2_0_branch2b_w"].shape == (64, 32, 1, 3, 3)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    blobs, grads = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn)
    exact = dtype in ("fp32", "split", "mix")
    for name in ("res2_2_branch2c_bn", "res3_3_branch2c_bn", "res4_5_branch2c_bn", "res5_2_branch2c_bn", "prob"):
        got = eng.fetch(name)
        e = rel(got, blobs[name].detach().numpy().reshape(got.shape))
        assert e < (1e-4 if exact else 2e-2), (name, e)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = sorted(((rel(eng.fetch_grad(n), grads[n].numpy()), n) for n in eng.trainable
                   if float(grads[n].norm()) > 1e-9 * gmax), reverse=True)
    e = np.array([x for x, _ in errs])
    print("\n[grouped conv %s] gradients median %.2e p90 %.2e max %.2e (%s)" % (
        dtype, np.median(e), np.sort(e)[int(0.9 * (len(e) - 1))], e[0], errs[0][1]))
    if exact:
        dec = eng.discrete_decisions()
        _, g2 = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec)
        cond = sorted(((rel(eng.fetch_grad(n), g2[n].numpy()), n) for _, n in errs), reverse=True)
        print("[grouped conv %s] on identical decisions: max %.2e (%s)" % (dtype, cond[0][0], cond[0][1]))
        assert cond[0][0] < (1.5e-3 if dtype == "mix" else 1e-3), cond[:5]
        assert np.median(e) < 2e-3 and e[0] < 2e-2, errs[:5]
    else:
        assert np.sort(e)[int(0.9 * (len(e) - 1))] < 0.12 and e[0] < 0.35, errs[:5]
    eng.sgd_step(0.01)
    torch.cuda.synchronize()
