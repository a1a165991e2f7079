"""Device-resident long-term feature bank (vlfb_lfb_* through the C ABI) against oracle/lfb.py:
bit-exact -- these are byte movers plus integer decisions."""
import numpy as np
import pytest
import torch

from oracle import lfb as ol

pytestmark = pytest.mark.gpu


def _ava_iterations(rng, n_iter=6, n_videos=4, dim=64, sec_lo=902, sec_hi=960):
    feats, meta = [], []
    for _ in range(n_iter):
        r = int(rng.integers(1, 40))
        f = rng.standard_normal((r, dim, 1, 1, 1)).astype(np.float32)
        m = np.zeros((r, 4), np.float32)
        m[:, 0] = rng.integers(0, n_videos, r)
        m[:, 1] = rng.integers(sec_lo, sec_lo + 12, r) if rng.random() < 0.5 else rng.integers(sec_lo, sec_hi, r)
        feats.append([f])
        meta.append([m])
    return feats, meta


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_append_matches_construct_ava_lfb(dtype):
    from vlfb.lfb_bank import DeviceBank
    rng = np.random.default_rng(0)
    feats, meta = _ava_iterations(rng)
    want = ol.construct_ava_lfb(feats, meta)
    cap = max(len(l) for v in want.values() for l in v.values())
    bank = DeviceBank(4, 60, cap, 64, dtype, step_base=902)
    for f, m in zip(feats, meta):
        src = torch.as_tensor(f[0]).cuda()
        bank.append_ava(src if dtype == "fp32" else src.to(torch.bfloat16), m[0])
    bank.check_no_drops()
    got = bank.to_reference()
    q = (lambda a: a) if dtype == "fp32" else (lambda a: torch.as_tensor(a).to(torch.bfloat16).float().numpy())
    for v in want:
        assert sorted(got[v]) == sorted(want[v])
        for s in want[v]:
            assert len(got[v][s]) == len(want[v][s])
            for a, b in zip(got[v][s], want[v][s]):
                assert np.array_equal(a, q(b.astype(np.float32)))          # same feature in the same list position
    assert int(bank.counts().sum()) == sum(f[0].shape[0] for f in feats)


def test_append_overflow_padding_and_bad_keys_are_counted_not_written():
    from vlfb.lfb_bank import DeviceBank
    from vlfb import hip
    bank = DeviceBank(2, 5, 2, 16, "fp32")
    f = torch.arange(6 * 16, dtype=torch.float32).reshape(6, 16).cuda()
    bank.append(f, [0, 0, 0, -1, 1, 1], [3, 3, 3, 0, 9, 4])      # third (0,3) overflows, -1 is padding, step 9 is out of range
    c = bank.counts()
    assert c[0, 3] == 2 and c[1, 4] == 1 and c.sum() == 3
    assert int(bank.dropped.item()) == 2
    with pytest.raises(hip.VlfbError):
        bank.check_no_drops()
    ref = bank.to_reference()
    assert np.array_equal(ref[0][3][1], f[1].cpu().numpy()) and np.array_equal(ref[1][4][0], f[5].cpu().numpy())


@pytest.mark.parametrize("dtype,out_dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16), ("bf16", torch.float32)])
def test_sample_window_matches_oracle(dtype, out_dtype):
    from vlfb.lfb_bank import DeviceBank
    rng = np.random.default_rng(3)
    feats, meta = _ava_iterations(rng, n_iter=12, dim=128)
    lfb = ol.construct_ava_lfb(feats, meta)
    bank = DeviceBank.from_reference(lfb, dtype=dtype)
    W, K = 16, 5
    vids = [0, 1, 1, 3, 2, 0]
    secs = [905, 930, 930, 959, 902, 1100]          # last: window entirely outside the bank
    sids = [10, 11, 11, 12, 13, 14]
    got = bank.sample_window(vids, secs, sids, W, K, seed=77, out_dtype=out_dtype).float().cpu().numpy()
    rt = (lambda a: a) if dtype == "fp32" else (lambda a: torch.as_tensor(a.astype(np.float32)).to(torch.bfloat16).float().numpy())
    for r, (v, s, sid) in enumerate(zip(vids, secs, sids)):
        want = ol.sample_lfb_ava(lfb.get(v, {}), s, W, K, 128, 77, sid, sorted(lfb).index(v))
        assert np.array_equal(got[r], rt(want)), r
    assert np.array_equal(got[1], got[2])               # RoIs of one clip share the sample
    assert np.all(got[5] == 0)


def test_sample_frames_matches_oracle_and_reference_round_trip():
    from vlfb.lfb_bank import DeviceBank
    rng = np.random.default_rng(4)
    frames = ol.charades_lfb_frames([500, 90, 30, 260], clips_per_second=2)
    feats = rng.standard_normal((len(frames) + 5, 32, 1, 1, 1)).astype(np.float32)
    lfb = ol.construct_frame_level_lfb([[feats[:40]], [feats[40:]]], frames)
    bank = DeviceBank(4, 500 // 12, 1, 32, "fp32")
    bank.append_frames(torch.as_tensor(feats[:40]).cuda(), frames[:40], 12)
    bank.append_frames(torch.as_tensor(feats[40:]).cuda(), frames[40:], 12)       # 5 padding rows at the end
    bank.check_no_drops()
    back = bank.to_reference(frame_level=True, sample_freq=12)
    for v in lfb:
        assert sorted(back[v]) == sorted(lfb[v])
        for f in lfb[v]:
            assert np.array_equal(back[v][f], lfb[v][f])
    vids = [0, 0, 0, 1, 2, 3, 3]
    centers = [250, 5, 499, 45, 15, 130, 259]
    got = bank.sample_frames(vids, centers, window=20, clips_per_second=2).cpu().numpy()
    for r, (v, c) in enumerate(zip(vids, centers)):
        assert np.array_equal(got[r], ol.sample_lfb_charades(lfb[v], c, 20, 2, 32).astype(np.float32)), r
    # a second bank built from the pickled form behaves the same
    bank2 = DeviceBank.from_reference(lfb, dtype="fp32", frame_level=True, sample_freq=12)
    assert np.array_equal(bank2.sample_frames(vids, centers, 20, 2).cpu().numpy(), got)


def test_bank_feeds_the_model_input_without_leaving_the_device():
    """infer `box_pooled` with the LFB-inference graph, append it on the device, sample into the
    training graph's `lfb` input tensor: same forward result as feeding the oracle's sample by hand"""
    import collections
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from vlfb import synth
    ov = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TEST.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TEST.VIDEO_LENGTH", 8,
          "TRAIN.CROP_SIZE", 64, "TEST.CROP_SIZE", 64]
    # 1) baseline model in lfb_infer_only mode produces box_pooled for every box
    load_preset("ava_r50_baseline", ov)
    m = ModelBuilder(train=False, split="test", name="infer")
    m.build_model(suffix="_infer_test", lfb_infer_only=True)
    eng = Engine(m, "bf16", device="cuda:0", base_seed=2)
    batch = synth.inputs(cfg, 2, 3, seed=9, crop=64, frames=8, suffix="_infer_test")
    shapes = collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in m.input_blob_names)
    eng.plan(shapes)
    eng.feed_params(synth.params(m, seed=2))
    for k in shapes:
        eng.feed(k, batch[k])
    eng.forward()
    feats_dev, code = eng.blob_tensor("box_pooled")
    R = batch["proposals_infer_test"].shape[0]
    from vlfb.lfb_bank import DeviceBank
    bank = DeviceBank(2, 30, 8, 2048, "bf16", step_base=902)
    vids = batch["proposals_infer_test"][:, 0].astype(np.int64)
    secs = 902 + 5 + (np.arange(R) % 3)
    bank.append(feats_dev.view(R, 2048), vids, secs)
    bank.check_no_drops()
    host = eng.fetch("box_pooled").reshape(R, 2048)
    ref = bank.to_reference()
    assert np.array_equal(np.stack(ref[int(vids[0])][int(secs[0])])[0], host[0])

    # 2) the LFB model trains on windows sampled from that bank, written into its input tensor
    load_preset("ava_r50_lfb_nl", ov + ["LFB.WINDOW_SIZE", 6])
    m2 = ModelBuilder(train=True, split="train", name="train")
    m2.build_model(suffix="_train")
    eng2 = Engine(m2, "bf16", device="cuda:0", base_seed=2)
    b2 = synth.inputs(cfg, 2, 3, seed=9, crop=64, frames=8)
    shapes2 = collections.OrderedDict((k, v.shape) for k, v in b2.items() if k in m2.input_blob_names)
    eng2.plan(shapes2)
    eng2.feed_params(synth.params(m2, seed=2))
    for k in shapes2:
        eng2.feed(k, b2[k])
    K = cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP
    clip = b2["proposals_train"][:, 0].astype(np.int64)
    lfb_dev, _ = eng2.blob_tensor("lfb_train")
    bank.sample_window(clip, np.full(len(clip), 908), clip + 100, 6, K, seed=5, out=lfb_dev)
    eng2.forward()
    loss_dev = float(eng2.fetch("loss").reshape(-1)[0])
    want = np.stack([ol.sample_lfb_ava(ref[int(c)], 908, 6, K, 2048, 5, int(c) + 100, int(c)) for c in clip])
    assert want.any()
    eng2.feed("lfb_train", want.astype(np.float32))
    eng2.forward()
    assert float(eng2.fetch("loss").reshape(-1)[0]) == loss_dev


def test_epic_verb_and_noun_banks_match_the_reference_samplers():
    """EPIC-Kitchens banks on the device (epic.py:310-374): verb = one clip feature per second, the first
    WINDOW_SIZE of them inside +-(WINDOW_SIZE * 30) // 2 frames; noun = up to 10 detector features per
    sampled frame, frames in order, truncated to WINDOW_SIZE rows.  Bit-exact against oracle/lfb.py on
    fp32 banks, incl. empty frames, clip centres near both ends of a video and windows that overflow."""
    import torch
    from vlfb.lfb_bank import DeviceBank
    from oracle import lfb as ol
    rng = np.random.default_rng(11)
    D = 64
    # ---- verb ----
    n_frames = [5400, 900, 31]
    verb = {v: {f: rng.standard_normal(D).astype(np.float32) for f in range(0, n, 30) if rng.uniform() > 0.15}
            for v, n in enumerate(n_frames)}
    bank = DeviceBank.from_epic(verb, noun=False, dtype="fp32")
    for W in (40, 7):
        vids, centres = [], []
        for v, n in enumerate(n_frames):
            for c in list(rng.integers(0, n, 6)) + [0, n - 1, 15, 29, 30]:
                vids.append(v); centres.append(int(c))
        got = bank.sample_epic_verb(vids, centres, W).cpu().numpy()
        for i, (v, c) in enumerate(zip(vids, centres)):
            want = ol.sample_verb_lfb_epic(c, verb[v], W, D)
            assert np.array_equal(got[i], want.astype(np.float32)), (W, v, c)
    # ---- noun ----
    noun = {}
    for v, n in enumerate(n_frames):
        noun[v] = {}
        for f in range(0, n, 30):
            k = int(rng.integers(0, 15))
            noun[v][f] = rng.standard_normal((k, D)).astype(np.float32) if k else []
    nbank = DeviceBank.from_epic(noun, noun=True, dtype="fp32")
    for W in (120, 25):
        vids, centres = [], []
        for v, n in enumerate(n_frames):
            for c in list(rng.integers(0, n, 6)) + [0, n - 1, 45, 170]:
                vids.append(v); centres.append(int(c))
        got = nbank.sample_epic_noun(vids, centres, W, max_per_frame=10).cpu().numpy()
        for i, (v, c) in enumerate(zip(vids, centres)):
            want = ol.sample_noun_lfb_epic(c, noun[v], W, D, 10, 1)
            assert np.array_equal(got[i], want.astype(np.float32)), (W, v, c)


def test_one_bank_per_clip_in_inference_is_bit_identical_to_one_copy_per_roi():
    """SURVEY.md 8f-1, "project the bank once per clip": the reference's data layer hands every RoI a copy of its clip's bank
    (lib/datasets/ava_data_input.py:191-192) and the graph projects each copy (lfb_helper.py:320-338).  An inference plan
    whose `lfb` blob has one row per CLIP runs lfb_1x1 and the phi / g convs of every FBO layer on n_clips x K rows and lets a
    RoI read its clip's bank through the batch-index column of `proposals` (vlfb_fbo_attn_fwd_shared): same outputs, bit for
    bit, with R / n_clips times fewer projected rows.  A training plan (dropout on the bank) refuses the shape."""
    import collections
    import pytest
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine, ConvStep
    from oracle import model as om
    ov = ["NUM_GPUS", 1, "TEST.BATCH_SIZE", 2, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16, "TEST.VIDEO_LENGTH", 16, "TEST.CROP_SIZE", 64,
          "TRAIN.CROP_SIZE", 64]
    load_preset("ava_r50_lfb_nl", ov)
    rois = [2, 3]
    inputs = om.synth_inputs(cfg, 2, "test", seed=5, rois_per_clip=rois, crop=64, frames=16)
    params = om.synth_params(cfg, seed=5)
    first = np.cumsum([0] + rois[:-1])
    per_clip = inputs["lfb"][first]                                    # the bank of each clip, once
    owner = inputs["proposals"][:, 0].astype(np.int64)
    assert np.array_equal(inputs["lfb"], per_clip[owner])              # (what the data layer duplicated)
    R, K = inputs["lfb"].shape[:2]
    for dtype in ("mix", "fp32"):
        out = {}
        for mode in ("per_roi", "per_clip"):
            model = ModelBuilder(train=False, split="test", name="test")
            model.build_model(suffix="_test")
            eng = Engine(model, dtype)
            feed = dict(inputs)
            if mode == "per_clip":
                feed["lfb"] = per_clip
            names = list(model.input_blob_names)
            eng.plan(collections.OrderedDict((n, feed[n[:-5]].shape) for n in names))
            eng.feed_params({k: v for k, v in params.items() if k in eng.param_views})
            for n in names:
                eng.feed(n, feed[n[:-5]])
            eng.forward()
            torch.cuda.synchronize()
            bank_convs = [s for s in eng.steps if isinstance(s, ConvStep) and s.out.name.startswith("lfb") and s.x.shape[2] == K]
            assert len(bank_convs) == 1 + 2 * cfg.FBO_NL.NUM_LAYERS          # lfb_1x1 + phi, g per layer
            out[mode] = (eng.fetch("pool5"), eng.fetch("prob"), sum(s.x.shape[0] * K for s in bank_convs))
            del eng
        assert np.array_equal(out["per_roi"][0], out["per_clip"][0]) and np.array_equal(out["per_roi"][1], out["per_clip"][1])
        assert out["per_roi"][2] == R * K * 5 and out["per_clip"][2] == 2 * K * 5
    # the oracle on the duplicated banks (the reference's formulation) agrees with the per-clip plan
    blobs, _ = om.run(cfg, {k: v for k, v in params.items() if k in om.param_spec(cfg)}, inputs, "test", torch.float64, False, None)
    got = out["per_clip"][1]
    assert np.linalg.norm(got - blobs["prob"].numpy().reshape(got.shape)) < 1e-4 * np.linalg.norm(got)
    # training: the bank of every RoI gets its own dropout mask (lfb_helper.py:333-336) -- one row per clip is refused
    load_preset("ava_r50_lfb_nl", ov)
    model = ModelBuilder(train=True, split="train", name="train")
    model.build_model(suffix="_train")
    tin = om.synth_inputs(cfg, 2, "train", seed=5, rois_per_clip=rois, crop=64, frames=16)
    shapes = collections.OrderedDict((k + "_train", v.shape) for k, v in tin.items())
    shapes["lfb_train"] = (2,) + tuple(tin["lfb"].shape[1:])
    with pytest.raises(ValueError, match="one bank per clip"):
        Engine(model, "mix").plan(shapes)
