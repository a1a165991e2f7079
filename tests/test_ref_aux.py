"""Host logic and oracle restatements against outputs of the REFERENCE's own functions.

tests/golden/ref_aux.npz was produced by oracle/make_ref_aux_golden.py, which imports the reference's modules from
/root/reference in the build container and calls them on seeded synthetic inputs (what it had to substitute -- a dict for
the Caffe2 workspace, the restated INTER_LINEAR for cv2.resize, Python-2 leftovers -- is listed in its docstring).  Nothing
here needs the reference tree: the fixture travels.

  * learning-rate schedule            utils.lr_policy (product)            == lib/utils/lr_policy.py, bit for bit (float32)
  * per-GPU batch / crop, unscoping   utils.misc (product)                 == lib/utils/misc.py
  * bank construction and sampling    oracle.lfb (checker of csrc/vlfb_lfb.hip) == tools/lfb_loader.py, datasets/{ava,charades,epic}.py
  * clip + box preprocessing          oracle.preprocess (checker of csrc/vlfb_data.hip) == data_input_helper.py / image_processor.py
  * checkpoint import / export        utils.checkpoints (product)          == lib/utils/checkpoints.py
"""
import json
import os
import pickle

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_aux.npz")
Z = np.load(GOLDEN)
META = json.loads(bytes(Z["meta"]).decode())


def _load(name, overrides):
    """the same configuration the generator loaded (YAML of that name + overrides), from this repo's presets / config"""
    from core import config as C
    from vlfb.presets import PRESETS, load_preset
    if name in PRESETS:
        return load_preset(name, overrides)
    # no preset of that name: the reference's effective configuration from the graph fixture (tests/test_ref_graph.py)
    tree = json.loads(json.dumps(_ref_cfgs()[name]))
    tree["LFB"].pop("NUM_LFB_FEAT")
    tree["SOLVER"]["STEPS"] = None
    C.reset_cfg()
    C.merge_dicts(tree, C.config)
    if overrides:
        C.cfg_from_list([str(x) for x in overrides])
    C.assert_and_infer_cfg()
    return C.config


_REF_CFGS = {}


def _ref_cfgs():
    if not _REF_CFGS:
        import gzip
        with gzip.open(os.path.join(os.path.dirname(GOLDEN), "ref_graphs.json.gz"), "rb") as fh:
            for g in json.loads(fh.read().decode())["graphs"]:
                if not g["overrides"]:
                    _REF_CFGS[g["config"]] = g["cfg"]
    return _REF_CFGS


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


# ---- learning rate ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(len(META["lr"])), ids=lambda k: "%s%s" % (META["lr"][k]["config"], "+" * bool(META["lr"][k]["overrides"])))
def test_lr_schedule_bit_for_bit(k):
    import utils.lr_policy as lr_policy
    case = META["lr"][k]
    _load(case["config"], case["overrides"])
    for it, want, raises in zip(Z["lr_%d_iters" % k], Z["lr_%d_values" % k], Z["lr_%d_raises" % k]):
        if raises:
            with pytest.raises(IndexError):
                lr_policy.get_lr_at_iter(int(it))
            continue
        got = lr_policy.get_lr_at_iter(int(it))
        assert isinstance(got, np.float32)
        assert bits(got) == bits(want), (int(it), float(got), float(want))


def test_lr_change_ratio():
    from models.model_builder_video import _get_lr_change_ratio
    for (a, b), want in zip(Z["lr_change_pairs"], Z["lr_change_ratio"]):
        assert _get_lr_change_ratio(a, b) == want


# ---- misc ----------------------------------------------------------------------------------------------------------------
def test_batch_crop_and_unscope():
    import utils.misc as misc
    for case in META["misc"]["sizes"]:
        _load(case["config"], case["overrides"])
        for s in ("train", "val", "test"):
            assert misc.get_batch_size(s) == case["batch"][s] and misc.get_crop_size(s) == case["crop"][s]
    for name, want in META["misc"]["unscope"]:
        assert misc.unscope_name(name) == want


# ---- long-term feature bank ----------------------------------------------------------------------------------------------------
def _lfb_case(kind):
    return [c for c in META["lfb"] if c["kind"] == kind][0]


def test_ava_bank_construction_and_windowed_draw(monkeypatch):
    from oracle import lfb as ol
    case = _lfb_case("ava")
    D = case["dim"]
    rows = Z["lfb_ava_batch_rows"]
    feats, metas, at = [], [], 0
    for it in range(rows.shape[0]):
        fi, mi = [], []
        for g in range(rows.shape[1]):
            n = int(rows[it, g])
            fi.append(Z["lfb_ava_feats"][at:at + n].reshape(n, D, 1, 1, 1))
            mi.append(Z["lfb_ava_meta"][at:at + n])
            at += n
        feats.append(fi)
        metas.append(mi)
    bank = ol.construct_ava_lfb(feats, metas)
    keys, got = [], []
    for v in sorted(bank):
        for s in sorted(bank[v]):
            for f in bank[v][s]:
                keys.append((v, s))
                got.append(f)
    assert np.array_equal(np.array(keys), Z["lfb_ava_bank_keys"])
    assert np.array_equal(np.array(got, dtype=np.float32), Z["lfb_ava_bank_rows"])
    # the draw: the reference consumes np.random.choice(range(n), k, replace=False) once per occupied second, in window
    # order (ava.py:315-319).  With the oracle's counter-based choice swapped for exactly that call, under the same seed,
    # every other line of sample_lfb must agree: window bounds, row layout j * K + k, zero rows, dtype
    monkeypatch.setattr(ol, "choice_without_replacement",
                        lambda n, k, seed, sample_id, video, step: list(np.random.choice(range(n), k, replace=False)))
    for i, d in enumerate(case["draws"]):
        np.random.seed(d["np_seed"])
        out = ol.sample_lfb_ava(bank[d["video"]], d["sec"], case["window"], case["max_per_step"], D, 0, 0, d["video"])
        want = Z["lfb_ava_sample_%d" % i]
        assert out.dtype == want.dtype and np.array_equal(out, want), i
    monkeypatch.undo()
    # ... and the oracle's own draw has the reference's structure: same rows occupied, each from the right second, distinct
    for i, d in enumerate(case["draws"]):
        out = ol.sample_lfb_ava(bank[d["video"]], d["sec"], case["window"], case["max_per_step"], D, 77, i, d["video"])
        want = Z["lfb_ava_sample_%d" % i]
        assert np.array_equal(np.any(out != 0, 1), np.any(want != 0, 1))
        K, lower = case["max_per_step"], d["sec"] - case["window"] // 2
        for j in range(case["window"]):
            have = [tuple(r) for r in out[j * K:(j + 1) * K] if np.any(r != 0)]
            pool = [tuple(np.float64(f)) for f in bank[d["video"]].get(lower + j, [])]
            assert len(set(have)) == len(have) and all(h in pool for h in have)


def test_charades_bank_and_window():
    from oracle import lfb as ol
    case = _lfb_case("charades")
    D, per = case["dim"], case["per_gpu"]
    frames = ol.charades_lfb_frames(case["num_frames"], case["clips_per_second"])
    assert np.array_equal(np.array(frames), Z["lfb_ch_frames"])
    flat = Z["lfb_ch_feats"]
    feats = [[flat[(2 * b + g) * per:(2 * b + g + 1) * per].reshape(per, D, 1, 1, 1) for g in range(2)]
             for b in range(len(flat) // (2 * per))]
    bank = ol.construct_frame_level_lfb(feats, frames)
    keys = [(v, f) for v in sorted(bank) for f in sorted(bank[v])]
    assert np.array_equal(np.array(keys), Z["lfb_ch_bank_keys"])
    assert np.array_equal(np.array([bank[v][f] for v, f in keys], dtype=np.float32), Z["lfb_ch_bank_rows"])
    for (v, c), want in zip(Z["lfb_ch_queries"], Z["lfb_ch_samples"]):
        out = ol.sample_lfb_charades(bank[int(v)], int(c), case["window"], case["clips_per_second"], D)
        assert out.dtype == want.dtype and np.array_equal(out, want), (v, c)


def test_epic_verb_and_noun_windows():
    from oracle import lfb as ol
    case = _lfb_case("epic_verb")
    bank = {int(f): r for f, r in zip(Z["lfb_ev_bank_keys"], Z["lfb_ev_bank_rows"])}
    for c, want in zip(Z["lfb_ev_queries"], Z["lfb_ev_samples"]):
        out = ol.sample_verb_lfb_epic(int(c), bank, case["window"], case["dim"])
        assert np.array_equal(out.astype(np.float32), want.astype(np.float32)), c      # (the reference returns float32 here)
    case = _lfb_case("epic_noun")
    bank, at = {}, 0
    for f, n in Z["lfb_en_counts"]:
        bank[int(f)] = Z["lfb_en_rows"][at:at + n] if n else []
        at += int(n)
    for c, want in zip(Z["lfb_en_queries"], Z["lfb_en_samples"]):
        out = ol.sample_noun_lfb_epic(int(c), bank, case["window"], case["dim"], case["max_per_frame"],
                                      case["frames_per_second"])
        assert np.array_equal(out, want), c


# ---- preprocessing -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(len(META["prep"])))
def test_clip_and_box_preprocessing(k):
    """random jitter size, crop offsets, flip decision (in the reference's draw order), test-time scale / forced flip /
    three-way shift crop, box clipping-scaling-cropping-flipping, /255, mean / std, BGR -> RGB: the reference's code with
    its cv2.resize replaced by the oracle's restatement, against the oracle end to end"""
    from oracle import preprocess as op
    case = META["prep"][k]
    cfg = _load("ava_r50_lfb_nl", ["TRAIN.JITTER_SCALES", str(case["jitter"]), "TEST.SCALE", case["test_scale"],
                                   "DATASET", case["dataset"], "AVA.FORCE_TEST_FLIP", case["force_flip"],
                                   "MODEL.USE_BGR", case["use_bgr"]])
    frames = list(Z["prep_frames_%dx%d" % tuple(case["frames"])])
    boxes = Z["prep_%d_boxes_in" % k].copy() if case["boxes"] else None
    np.random.seed(case["np_seed"])
    clip, out_boxes = op.images_and_boxes_preprocessing(frames, case["split"], case["crop"], case["shift"], cfg, boxes=boxes,
                                                        rng=np.random)
    want = Z["prep_%d_clip" % k]
    assert clip.shape == want.shape and str(clip.dtype) == case["clip_dtype"]
    assert np.array_equal(clip, want), float(np.abs(clip.astype(np.float64) - want).max())
    if case["boxes"]:
        assert np.array_equal(np.asarray(out_boxes, dtype=np.float64), Z["prep_%d_boxes_out" % k])
    else:
        assert out_boxes is None


def test_normalisation_constants():
    """data_input_helper.py:40-41: float32 copies of cfg.DATA_MEAN / cfg.DATA_STD (BGR order)"""
    cfg = _load("ava_r50_lfb_nl", [])
    mean, std = META["prep_mean_std"]
    assert [float(np.float32(v)) for v in cfg.DATA_MEAN] == mean and [float(np.float32(v)) for v in cfg.DATA_STD] == std


# ---- checkpoints ---------------------------------------------------------------------------------------------------------------
def _group(prefix):
    return {n[len(prefix):]: Z[n] for n in Z.files if n.startswith(prefix)}


def _scalar_or_array(v):
    return v.item() if v.shape == () and v.dtype.kind in "iu" else v


def test_classification_checkpoint_conversion(tmp_path):
    """BN statistics folded into the affine pair, `epoch / model_iter / lr` and every `*_momentum` dropped
    (checkpoints.py:88-146), in the reference's float32 arithmetic"""
    from utils import checkpoints as ck
    src = {n: _scalar_or_array(v) for n, v in _group("ckpt_cls_in/").items()}
    path = str(tmp_path / "cls.pkl")
    with open(path, "wb") as fh:
        pickle.dump({"blobs": src}, fh, protocol=2)
    got = ck.load_and_convert_caffe2_cls_model(path)["blobs"]
    want = _group("ckpt_cls_out/")
    assert sorted(got) == sorted(want)
    for n in want:
        g = np.asarray(got[n])
        assert g.dtype == want[n].dtype and np.array_equal(g, want[n]), n


class _Fill(object):
    def __init__(self, shape):
        self.shape = tuple(shape)


class _Engine(object):
    allocated = True

    def __init__(self, init, train):
        self.train = train
        self.p = {n: v.copy() for n, v in init.items() if not n.endswith("_momentum")}
        self.m = {n[:-len("_momentum")]: v.copy() for n, v in init.items() if n.endswith("_momentum")}
        self.lr = None

    def feed_params(self, d):
        for n, v in d.items():
            assert v.dtype == np.float32 and v.shape == self.p[n].shape, n
            self.p[n] = v.copy()

    def feed_momentum(self, d):
        for n, v in d.items():
            assert v.dtype == np.float32 and v.shape == self.m[n].shape, n
            self.m[n] = v.copy()

    def fetch_param(self, n):
        return self.p[n]

    def fetch_momentum(self, n):
        return self.m[n]

    def set_lr(self, lr):
        self.lr = np.float32(lr)


class _Model(object):
    def __init__(self, ck, train):
        self.train = train
        self.params = list(ck["params"])
        self.computed = list(ck["computed"])
        self.frozen = set(ck["frozen"])
        self.param_init_net = type("P", (), {})()
        self.param_init_net.fills = {n: _Fill(s) for n, s in ck["shapes"].items()}

    def TrainableParams(self, scope=""):
        return [p for p in self.params if p not in self.frozen]

    def GetParams(self, namescope=None):
        return list(self.params)

    def GetComputedParams(self, namescope=None):
        return list(self.computed)

    def GetAllParams(self, namescope=None):
        return self.params + self.computed


@pytest.mark.parametrize("r", range(len(META["ckpt"]["runs"])))
def test_initialisation_from_a_weights_file(r, tmp_path):
    """classifier rule (same element count -> reshaped, else left alone), 2-D -> 3-D inflation (repeat over kT, / kT) of
    weights AND of their momentum, blobs missing from the file keep their values, float64 -> float32, momentum only
    for the trainable parameters of a training net and only when asked, lr from the file or 1.0 with
    TRAIN.RESET_START_ITER, wrapped and bare pickles (checkpoints.py:271-383); then the file a training net writes (:421-459)"""
    from utils import checkpoints as ck
    from vlfb import dist
    info = META["ckpt"]
    run = info["runs"][r]
    _load("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 8, "TEST.BATCH_SIZE", 8, "TRAIN.RESET_START_ITER",
                             run["reset_start_iter"]])
    blobs = {n: _scalar_or_array(v) for n, v in _group("ckpt_file/").items()}
    if not run["file_has_lr"]:
        del blobs["lr"]
    path = str(tmp_path / "w.pkl")
    with open(path, "wb") as fh:
        pickle.dump({"blobs": blobs} if run["wrapped"] else blobs, fh, protocol=2)
    train = run["net"] == "train"
    model = _Model(info, train)
    model.engine = _Engine(_group("ckpt_init/"), train)
    model_iter, prev_lr = ck.initialize_master_gpu_model_params(model, path, load_momentum=run["load_momentum"])
    assert (model_iter, prev_lr) == (run["model_iter"], run["prev_lr"])
    want = _group("ckpt_run%d/gpu_0/" % r)
    assert bits(model.engine.lr) == bits(want.pop("lr"))
    for n, w in want.items():
        got = model.engine.m[n[:-len("_momentum")]] if n.endswith("_momentum") else model.engine.p[n]
        assert got.dtype == w.dtype and np.array_equal(got, w), n
    assert set(want) == set(model.engine.p) | {n + "_momentum" for n in model.engine.m}
    if "saved_keys" in run:
        out = str(tmp_path / "saved.pkl")
        ck.save_model_params(model, out, 99)
        with open(out, "rb") as fh:
            saved = pickle.load(fh)
        assert list(saved) == ["blobs"] and sorted(saved["blobs"]) == run["saved_keys"]
        ref = _group("ckpt_saved/")
        for n in run["saved_keys"]:
            assert np.array_equal(np.asarray(saved["blobs"][n]), ref[n]), n
            assert np.asarray(saved["blobs"][n]).dtype == ref[n].dtype, n


# ---- AVA multi-crop merge -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(len(META["multicrop"])))
def test_multi_crop_merge(k):
    """lib/utils/metrics.py:623-711 run on synthetic score files: which of the three shifted crops see a box (after the
    flip of its x range), the mean of their sigmoids, then the sum over scales and flips (written with '%f')"""
    from oracle import multicrop as omc
    case = META["multicrop"][k]
    boxes, logits = Z["mc_%d_boxes" % k], Z["mc_%d_logits" % k]
    total = 0.0
    for si, scale in enumerate(case["scales"]):
        for fi, flip in enumerate([False, True]):
            got = omc.merge_three_shifts(list(logits[si, fi]), boxes, flip, scale, case["height"], case["width"])
            want = Z["mc_%d_combined" % k][si, fi]
            # the reference parses the logit back from the csv (written here with repr(): exact) and writes str(float)
            assert np.array_equal(got, want), (scale, flip, float(np.abs(got - want).max()))
            total = total + want
    final = Z["mc_%d_final" % k]
    assert np.array_equal(np.array([[float("%f" % v) for v in row] for row in total]), final)


# ---- solver structure -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(len(META["solver"])), ids=lambda k: META["solver"][k]["config"])
def test_parameter_update_rules(k):
    """model_builder_video.py:348-389 run on this repo's parameter catalogue: one WeightedSum (gradient += decay * param,
    decay class by '_bn' in the name) and one MomentumSGDUpdate (SOLVER.MOMENTUM / NESTEROV, zero-filled `<param>_momentum`)
    per TRAINABLE parameter and nothing for the others -- against the engine's decay ranges, trainable set and the
    constants its fused solver launch reads"""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_lowering import plan
    case = META["solver"][k]
    small = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64]
    cfg, m, eng = plan(case["config"], small + case["overrides"])
    assert m.GetParams() == case["params"] and m.TrainableParams() == case["trainable"]
    sol = case["solver"]
    assert (cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_BN, cfg.SOLVER.MOMENTUM, cfg.SOLVER.NESTEROV) == \
        (sol["WEIGHT_DECAY"], sol["WEIGHT_DECAY_BN"], sol["MOMENTUM"], sol["NESTEROV"])
    fills = {c[1][1]: c for c in case["calls"] if c[0] == "param_init_net.ConstantFill"}
    assert fills["lr"][2] == {"shape": [1], "value": case["current_lr"]}
    decay = {"weight_decay": fills["weight_decay"][2]["value"], "weight_decay_bn": fills["weight_decay_bn"][2]["value"]}
    assert decay == {"weight_decay": sol["WEIGHT_DECAY"], "weight_decay_bn": sol["WEIGHT_DECAY_BN"]}
    assert fills["ONE"][2]["value"] == 1.0
    want_wd, updated = {}, []
    for name, args, kw in case["calls"]:
        if name == "WeightedSum":
            (grad, one, param, wd), out = args
            assert one == "ONE" and grad == param + "_grad" and out == grad
            want_wd[param] = decay[wd]
        elif name == "net.MomentumSGDUpdate":
            ins, outs = args
            grad, mom, lr, param = ins
            assert outs == [grad, mom, param] and lr == "lr" and mom == param + "_momentum"
            assert kw == {"momentum": sol["MOMENTUM"], "nesterov": sol["NESTEROV"]}
            assert fills[mom][1][0] == [param] and fills[mom][2] == {"value": 0.0}
            updated.append(param)
    assert updated == case["trainable"] and sorted(want_wd) == sorted(updated)
    # the engine: the same parameters are the solver's, each under the same decay
    assert sorted(eng.train_layout) == sorted(updated)
    got_wd = {}
    for n, (off, cnt, shape) in eng.train_layout.items():
        r = [w for lo, hi, w in eng.wd_ranges if lo <= off and off + cnt <= hi]
        assert len(r) == 1, n
        got_wd[n] = r[0]
    assert got_wd == want_wd
    if "NONLOCAL.USE_BN" in case["overrides"]:
        assert len(set(want_wd.values())) == 2          # (the batch-norm case exercises both classes)


@pytest.mark.parametrize("k", range(len(META["prep"])))
def test_product_clip_geometry_reproduces_the_reference_clip(k):
    """the PRODUCT's host side of the device preprocessing (datasets.data_input_helper.plan_clip: the same np.random calls in
    the same order -> resized size, window origin, walk direction, transformed boxes) against the reference's output:
    boxes bit for bit, and the window it tells the kernel to read -- taken here from frames resized on the host by the
    restated INTER_LINEAR -- is the reference's clip after the kernel's /255, mean / std and channel flip"""
    from datasets import data_input_helper as dh
    from oracle import preprocess as op
    case = META["prep"][k]
    cfg = _load("ava_r50_lfb_nl", ["TRAIN.JITTER_SCALES", str(case["jitter"]), "TEST.SCALE", case["test_scale"],
                                   "DATASET", case["dataset"], "AVA.FORCE_TEST_FLIP", case["force_flip"],
                                   "MODEL.USE_BGR", case["use_bgr"]])
    h, w = case["frames"]
    frames = list(Z["prep_frames_%dx%d" % (h, w)])
    boxes = Z["prep_%d_boxes_in" % k].copy() if case["boxes"] else None
    np.random.seed(case["np_seed"])
    plan, out_boxes = dh.plan_clip(h, w, case["split"], case["crop"], case["shift"], boxes, np.random)
    if case["boxes"]:
        assert np.array_equal(np.asarray(out_boxes, dtype=np.float64), Z["prep_%d_boxes_out" % k])
    else:
        assert out_boxes is None
    crop = case["crop"]
    rs = [op.resize_u8(f, plan["resized_w"], plan["resized_h"]) if (plan["resized_h"], plan["resized_w"]) != (h, w) else f
          for f in frames]
    cols = plan["x0"] + (-1 if plan["flip"] else 1) * np.arange(crop)
    win = np.stack([r[plan["y0"]:plan["y0"] + crop][:, cols] for r in rs]).astype(np.float32)      # (T, crop, crop, 3) BGR
    mean = np.array(cfg.DATA_MEAN, dtype=np.float32)
    std = np.array(cfg.DATA_STD, dtype=np.float32)
    want = Z["prep_%d_clip" % k]                                  # (3, T, crop, crop), RGB unless MODEL.USE_BGR
    got = ((win / np.float32(255.0)) - mean) / std                # (the kernel's arithmetic, csrc/vlfb_data.hip)
    got = got.transpose(3, 0, 1, 2)
    if not case["use_bgr"]:
        got = got[::-1]
    assert np.array_equal(np.ascontiguousarray(got), want)


# ---- the product's bank-step arithmetic (host side of csrc/vlfb_lfb.hip) ----------------------------------------------------------
def _compact(rows_by_step, lo, hi, window, dim, max_per_step=1):
    """what the device gather does with a step range: occupied steps in order, up to max_per_step rows each, the first
    `window` rows, zero padded (vlfb.lfb_bank.DeviceBank._sample_packed)"""
    out = np.zeros((window, dim))
    k = 0
    for t in range(int(lo), int(hi) + 1):
        for r in rows_by_step.get(t, [])[:max_per_step]:
            if k < window:
                out[k] = r
                k += 1
    return out


def test_product_window_steps_select_what_the_reference_selects():
    """vlfb.lfb_bank.{frame,epic_verb,epic_noun}_window_steps: the bank-step range handed to the gather kernel covers exactly
    the frames the reference's samplers visit (rounding of the window start, inclusive ends, floor / ceil onto bank
    steps, truncation toward zero in the noun window) -- reference outputs on synthetic banks as the judge"""
    from vlfb import lfb_bank as lb
    # Charades: bank step t holds frame sample_freq * (t + 1) - 1
    case = _lfb_case("charades")
    sf = lb.FPS // case["clips_per_second"]
    banks = {}
    for (v, f), row in zip(Z["lfb_ch_bank_keys"], Z["lfb_ch_bank_rows"]):
        assert (f + 1) % sf == 0
        banks.setdefault(int(v), {})[(int(f) + 1) // sf - 1] = [row]
    q = Z["lfb_ch_queries"]
    lo, hi = lb.frame_window_steps(q[:, 1], case["window"], case["clips_per_second"])
    for i, (v, c) in enumerate(q):
        got = _compact(banks[int(v)], lo[i], hi[i], case["window"], case["dim"])
        assert np.array_equal(got, Z["lfb_ch_samples"][i]), (v, c)
    # EPIC verb: bank step t holds frame 30 t
    case = _lfb_case("epic_verb")
    bank = {int(f) // lb.EPIC_FPS: [r] for f, r in zip(Z["lfb_eva_bank_keys"], Z["lfb_eva_bank_rows"])}
    q = Z["lfb_eva_queries"]
    lo, hi = lb.epic_verb_window_steps(q, case["window"])
    for i, c in enumerate(q):
        got = _compact(bank, lo[i], hi[i], case["window"], case["dim"])
        assert np.array_equal(got.astype(np.float32), Z["lfb_eva_samples"][i].astype(np.float32)), c
    # EPIC noun: up to max_per_frame detections per bank step, steps one detector frame apart
    case = _lfb_case("epic_noun")
    bank, at = {}, 0
    for f, n in Z["lfb_en_counts"]:
        bank[int(f) // (lb.EPIC_FPS // case["frames_per_second"])] = list(Z["lfb_en_rows"][at:at + n])
        at += int(n)
    q = Z["lfb_en_queries"]
    lo, hi = lb.epic_noun_window_steps(q, case["window"], case["max_per_frame"], case["frames_per_second"])
    for i, c in enumerate(q):
        got = _compact(bank, lo[i], hi[i], case["window"], case["dim"], case["max_per_frame"])
        assert np.array_equal(got, Z["lfb_en_samples"][i]), c


@pytest.mark.parametrize("k", range(len(META["multicrop"])))
def test_product_multi_crop_merge(k):
    """vlfb.multicrop.{shift_validity, merge_shifts, merge_scales_and_flips} (the host arithmetic of the product's AVA
    multi-crop tester) against the reference's merge of synthetic score files, exactly"""
    from vlfb import multicrop as mc
    case = META["multicrop"][k]
    boxes, logits = Z["mc_%d_boxes" % k], Z["mc_%d_logits" % k]
    per_pass = []
    for si, scale in enumerate(case["scales"]):
        for fi, flip in enumerate([False, True]):
            valid = mc.shift_validity(boxes, flip, scale, case["height"], case["width"])
            got = mc.merge_shifts(logits[si, fi], valid)
            assert np.array_equal(got, Z["mc_%d_combined" % k][si, fi]), (scale, flip)
            per_pass.append(got)
    total = mc.merge_scales_and_flips(per_pass)
    assert np.array_equal(np.array([[float("%f" % v) for v in row] for row in total]), Z["mc_%d_final" % k])


# ---- configuration semantics ---------------------------------------------------------------------------------------------------
def _get(cfg, dotted):
    node = cfg
    for k in dotted.split("."):
        node = node[k]
    return node


@pytest.mark.parametrize("case", META["config"]["cfg_from_list"], ids=lambda c: " ".join(c["args"])[:40])
def test_cfg_from_list_like_the_reference(case):
    """lib/core/config.py:431-451 on KEY VAL lists: literals evaluated, the type of the default enforced (an int for a float
    key is refused, so is a float for an int), None-default keys accept anything, unknown keys / odd lists are assertions"""
    from core import config as C
    C.reset_cfg()
    want = case["result"]
    if "raises" in want:
        with pytest.raises(BaseException) as e:
            C.cfg_from_list(list(case["args"]))
        assert type(e.value).__name__ == want["raises"]
    else:
        C.cfg_from_list(list(case["args"]))
        for k, v in want["values"].items():
            got = _get(C.config, k)
            assert got == v and type(got) is type(v), (k, got, v)
    C.reset_cfg()


@pytest.mark.parametrize("case", META["config"]["merge_dicts"], ids=lambda c: json.dumps(c["tree"])[:40])
def test_merge_dicts_like_the_reference(case):
    """lib/core/config.py:394-421 on YAML-shaped trees: values set, and the error CLASS where it refuses (KeyError for an
    unknown top-level key, ValueError for a type change -- also for a list on the None-default SOLVER.STEPS --, a plain
    Exception for anything that goes wrong below the top level)"""
    from core import config as C
    C.reset_cfg()
    want = case["result"]
    if "raises" in want:
        with pytest.raises(BaseException) as e:
            C.merge_dicts(case["tree"], C.config)
        assert type(e.value).__name__ == want["raises"]
    else:
        C.merge_dicts(case["tree"], C.config)
        for k, v in want["values"].items():
            got = _get(C.config, k)
            assert got == v and type(got) is type(v), (k, got, v)
    C.reset_cfg()


# ---- start of training: which file, with or without momentum, from which iteration ----------------------------------------------
@pytest.mark.parametrize("k", range(len(META["ckpt"]["policy"])),
                         ids=lambda k: "".join("%s%d" % (n[0], META["ckpt"]["policy"][k][n]) for n in
                                               ("resume", "params_file", "have_checkpoints", "convert", "reset_start_iter")))
def test_start_of_training_policy(k, tmp_path):
    """checkpoints.py:180-236 load_model_from_params_file (+ :51-80 checkpoint discovery, :149-177 convert_model) run over
    CHECKPOINT.RESUME x TRAIN.PARAMS_FILE x checkpoints on disk x CHECKPOINT.CONVERT_MODEL x TRAIN.RESET_START_ITER: the file
    that gets loaded (the newest c2_model_iter*.pkl by NUMBER, the pre-trained file, the converted file), whether momentum
    comes with it, the iteration training starts from, model.current_lr and the loaded values"""
    from utils import checkpoints as ck
    info, case = META["ckpt"], META["ckpt"]["policy"][k]
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "checkpoints"))
    pre = {n: _scalar_or_array(v) for n, v in _group("ckpt_pre/").items()}
    pf = os.path.join(work, "pretrained.pkl")
    with open(pf, "wb") as fh:
        pickle.dump({"blobs": pre}, fh, protocol=2)
    if case["have_checkpoints"]:
        base = {n: _scalar_or_array(v) for n, v in _group("ckpt_file/").items()}
        for name, (it, lr) in info["policy_checkpoints"].items():
            b = dict(base)
            b.update({"model_iter": it, "lr": np.float32(lr)})
            with open(os.path.join(work, "checkpoints", name), "wb") as fh:
                pickle.dump({"blobs": b}, fh, protocol=2)
        open(os.path.join(work, "checkpoints", "notes.txt"), "w").close()
        open(os.path.join(work, "checkpoints", "c2_model_iter99999.txt"), "w").close()
    _load("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 8, "TEST.BATCH_SIZE", 8, "CHECKPOINT.DIR", work,
                             "CHECKPOINT.RESUME", case["resume"], "CHECKPOINT.CONVERT_MODEL", case["convert"],
                             "TRAIN.RESET_START_ITER", case["reset_start_iter"],
                             "TRAIN.PARAMS_FILE", pf if case["params_file"] else ""])
    model = _Model(info, True)
    model.engine = _Engine(_group("ckpt_init/"), True)
    model.current_lr = -1.0
    loads = []
    real = ck.initialize_params_from_file

    def logged(model, weights_file, load_momentum=True):
        loads.append([os.path.basename(weights_file), bool(load_momentum)])
        return real(model=model, weights_file=weights_file, load_momentum=load_momentum)
    ck.initialize_params_from_file = logged
    try:
        assert ck.find_checkpoint() == case["have_checkpoints"]
        latest = ck.get_checkpoint_resume_file()
        assert (os.path.basename(latest) if latest else None) == case["latest"]
        start = ck.load_model_from_params_file(model)
    finally:
        ck.initialize_params_from_file = real
    assert start == case["start_iter"] and loads == case["loads"]
    assert float(model.current_lr) == case["current_lr"]
    if case["lr_blob"] is not None:
        assert float(model.engine.lr) == case["lr_blob"]
    for n, want in _group("ckpt_policy%d/" % k).items():
        got = model.engine.m[n[:-len("_momentum")]] if n.endswith("_momentum") else model.engine.p[n]
        assert np.array_equal(got, want), n


# ---- learning-rate updates and momentum correction over a run --------------------------------------------------------------------
@pytest.mark.parametrize("k", range(len(META["lr_updates"])))
def test_lr_updates_and_momentum_correction_follow_the_reference(k):
    """model_builder_video.py:252-290 of the reference driven over the iterations of a run (warm-up, the two decays, the
    end): this repo's ModelBuilder.SetCurrentLr / UpdateWorkspaceLr must hand the engine a new learning rate at exactly
    the iterations the reference rewrites its lr blobs, with the same float32 value, and rescale the update history at the
    same iterations by the same factor (SOLVER.SCALE_MOMENTUM / SCALE_MOMENTUM_THRESHOLD)"""
    from models.model_builder_video import ModelBuilder
    case = META["lr_updates"][k]
    _load(case["config"], ["NUM_GPUS", 2, "TRAIN.BATCH_SIZE", 16, "TEST.BATCH_SIZE", 16] + case["overrides"])
    events = []

    class Eng(object):
        def set_lr(self, lr):
            events.append(["set_lr", float(np.float32(lr))])

        def scale_momentum(self, f):
            events.append(["correct_momentum", float(f)])
    m = ModelBuilder(train=True, split="train", name="t")
    m.engine = Eng()
    trace = case["trace"]
    m.current_lr = 0
    m.SetCurrentLr(trace[0]["iter"])
    assert float(m.current_lr) == trace[0]["set_current"]
    for t in trace[1:]:
        del events[:]
        m.UpdateWorkspaceLr(t["iter"])
        assert float(m.current_lr) == t["current_lr"], t
        fed = sorted(set(t["fed"].values()))
        assert [e[1] for e in events if e[0] == "set_lr"] == fed, (t, events)       # one lr per GPU there, one engine here
        assert [e for e in events if e[0] == "correct_momentum"] == t["events"], (t, events)
