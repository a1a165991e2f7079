"""The caffe2.python.workspace-shaped facade: Feed/Fetch with scoped blob names, CreateNet, RunNet
(fwd + bwd + solver), parameters in the reference's layouts under their Caffe2 names."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_train_net_style_loop_through_the_facade():
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb import workspace, synth
    workspace.ResetWorkspace()
    workspace.set_compute_dtype("fp32")
    load_preset("charades_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 16,
                                        "TRAIN.CROP_SIZE", 64])
    model = ModelBuilder(train=True, split="train", name="train", use_cudnn=True, cudnn_exhaustive_search=True)
    model.build_model(suffix="_train")
    workspace.register(model)
    batch = synth.inputs(cfg, 2, seed=5, crop=64, frames=16)
    for k, v in batch.items():
        workspace.FeedBlob("gpu_0/" + k, v)
    eng = workspace.CreateNet(model.net)
    assert model.scope == "gpu_0/" and set(model.TrainableParams()) == set(eng.trainable)
    w0 = workspace.FetchBlob("gpu_0/pred_w").copy()
    frozen0 = workspace.FetchBlob("gpu_0/res4_2_branch2a_w").copy()
    assert w0.shape == (157, 2560) and frozen0.shape == (256, 1024, 3, 1, 1)
    model.UpdateWorkspaceLr(0)
    losses = []
    for it in range(3):
        workspace.RunNet(model.net)
        losses.append(float(workspace.FetchBlob("gpu_0/loss")))
    assert all(np.isfinite(losses))
    # NaN guard: every step's loss from the device ring with one sync, then nothing until the next step
    import utils.misc as misc
    assert np.allclose(misc.check_nan_losses(), losses, rtol=0, atol=0) and misc.check_nan_losses(model) == []
    assert not np.allclose(workspace.FetchBlob("gpu_0/pred_w"), w0)                      # head is trained
    assert np.array_equal(workspace.FetchBlob("gpu_0/res4_2_branch2a_w"), frozen0)        # FREEZE_BACKBONE
    assert workspace.FetchBlob("gpu_0/pred_w_momentum").shape == (157, 2560)
    assert workspace.FetchBlob("gpu_0/prob").shape == (2, 157)
    assert workspace.FetchBlob("gpu_0/pool5").shape == (2, 2560, 1, 1, 1)
    # feeding a parameter in the reference layout round-trips
    new = np.random.default_rng(0).standard_normal((157, 2560)).astype(np.float32)
    workspace.FeedBlob("gpu_0/pred_w", new)
    assert np.array_equal(workspace.FetchBlob("gpu_0/pred_w"), new)
    workspace.FeedBlob("gpu_0/pred_b", np.full(157, np.nan, np.float32))
    workspace.RunNet(model.net)
    with pytest.raises(FloatingPointError):
        misc.check_nan_losses(model)
    workspace.ResetWorkspace()


def test_checkpoint_round_trip_through_the_device_engine(tmp_path):
    """save after two solver steps, resume into a fresh engine: parameters, momentum and the next
    step's loss are bit-identical; a 2-D (image model) weight file inflates into the 3-D kernels"""
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb import workspace, synth
    from utils import checkpoints as ck

    def fresh():
        workspace.ResetWorkspace()
        workspace.set_compute_dtype("fp32")
        load_preset("charades_r50_baseline", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 1, "TRAIN.VIDEO_LENGTH", 8,
                                              "TRAIN.CROP_SIZE", 64, "TRAIN.DROPOUT_RATE", 0.0])
        cfg.CHECKPOINT.DIR = str(tmp_path)
        cfg.TRAIN.PARAMS_FILE = ""
        model = ModelBuilder(train=True, split="train", name="train")
        model.build_model(suffix="_train")
        workspace.register(model)
        for k, v in synth.inputs(cfg, 1, seed=3, crop=64, frames=8).items():
            workspace.FeedBlob("gpu_0/" + k, v)
        return model, workspace.CreateNet(model.net)

    model, eng = fresh()
    model.UpdateWorkspaceLr(0)
    for _ in range(2):
        workspace.RunNet(model.net)
    path = ck.create_and_get_checkpoint_directory() + "/c2_model_iter2.pkl"
    ck.save_model_params(model, path, model_iter=1)
    names = ["conv1_w", "res3_1_branch2b_w", "nonlocal_conv4_1_theta_b", "pred_w", "res_conv1_bn_s"]
    want_p = {n: workspace.FetchBlob("gpu_0/" + n).copy() for n in names}
    want_m = {n: workspace.FetchBlob("gpu_0/%s_momentum" % n).copy() for n in names[:4]}
    workspace.RunNet(model.net)
    want_loss = float(workspace.FetchBlob("gpu_0/loss"))

    model2, eng2 = fresh()
    assert ck.load_model_from_params_file(model2) == 2
    for n in names:
        assert np.array_equal(workspace.FetchBlob("gpu_0/" + n), want_p[n]), n
    for n in names[:4]:
        assert np.array_equal(workspace.FetchBlob("gpu_0/%s_momentum" % n), want_m[n]), n
    assert abs(float(workspace.FetchBlob("gpu_0/lr")) - float(model.current_lr)) < 1e-9
    workspace.RunNet(model2.net)
    assert float(workspace.FetchBlob("gpu_0/loss")) == want_loss

    # image-model file: 2-D conv weights -> inflated over kT, classifier of another size skipped
    blobs = ck.read_blobs(path)
    tname = next(n for n in model.params if n.endswith("branch2a_w") and blobs[n].shape[2] == 3)
    w3 = blobs[tname]                                     # (Ci, Cin, 3, 1, 1) in the I3D graph
    blobs[tname] = w3.sum(axis=2)
    blobs["pred_w"] = np.zeros((400, 2048), np.float32)
    img = str(tmp_path / "image_model.pkl")
    ck.write_blobs(img, blobs)
    model3, eng3 = fresh()
    pred0 = workspace.FetchBlob("gpu_0/pred_w").copy()
    ck.initialize_params_from_file(model3, img, load_momentum=False)
    got = workspace.FetchBlob("gpu_0/" + tname)
    np.testing.assert_allclose(got, np.repeat(w3.sum(axis=2, keepdims=True), 3, axis=2) / 3.0, rtol=1e-6)
    assert np.array_equal(workspace.FetchBlob("gpu_0/pred_w"), pred0)
    workspace.ResetWorkspace()


def test_train_and_test_nets_of_one_scope_share_their_parameter_blobs():
    """tools/train_net.py keeps train_model and test_model in ONE workspace and evaluates the weights being
    trained (train_net.py:60-77, 113-131).  Here: the test net's parameters alias the train net's, its MFMA
    operand copies follow the solver steps, FetchBlob('gpu_0/prob') answers from the net that ran last, and
    a parameter fed once is seen by both nets."""
    import torch
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb import workspace, synth
    from vlfb.engine import Engine
    workspace.ResetWorkspace()
    workspace.set_compute_dtype("fp32")
    ov = ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TEST.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TEST.VIDEO_LENGTH", 8,
          "TRAIN.CROP_SIZE", 64, "TEST.CROP_SIZE", 64, "TRAIN.DROPOUT_RATE", 0.0]
    load_preset("charades_r50_baseline", ov)
    train = ModelBuilder(train=True, split="train", name="train")
    train.build_model(suffix="_train")
    test = ModelBuilder(train=False, split="val", name="val")
    test.build_model(suffix="_test")
    workspace.register(train)
    workspace.register(test)
    tb = synth.inputs(cfg, 2, seed=5, crop=64, frames=8, suffix="_train")
    vb = synth.inputs(cfg, 2, seed=6, crop=64, frames=8, suffix="_test")
    for k, v in list(tb.items()) + list(vb.items()):
        workspace.FeedBlob("gpu_0/" + k, v)
    e_train = workspace.CreateNet(train.net)
    e_test = workspace.CreateNet(test.net)
    assert e_test.param_views["pred_w"].data_ptr() == e_train.param_views["pred_w"].data_ptr()
    assert e_test.param_views["conv1_w"].data_ptr() == e_train.param_views["conv1_w"].data_ptr()
    workspace.RunNet(test.net)
    p0 = workspace.FetchBlob("gpu_0/prob").copy()              # evaluation before training
    train.UpdateWorkspaceLr(0)
    for _ in range(3):
        workspace.RunNet(train.net)
    p_train = workspace.FetchBlob("gpu_0/prob").copy()         # the TRAIN net ran last: its prob
    workspace.RunNet(test.net)
    p1 = workspace.FetchBlob("gpu_0/prob").copy()              # now the test net's, with the trained weights
    assert not np.allclose(p1, p0) and not np.array_equal(p1, p_train)
    # reference: a stand-alone test engine fed with the trained parameters gives the same probabilities
    ref = Engine(test, "fp32")
    import collections
    ref.plan(collections.OrderedDict((n, vb[n].shape) for n in test.input_blob_names))
    ref.feed_params({n: e_train.fetch_param(n) for n in ref.param_views})
    for n in test.input_blob_names:
        ref.feed(n, vb[n])
    ref.forward()
    torch.cuda.synchronize()
    assert np.array_equal(ref.fetch("prob"), p1)
    test.engine = e_test
    # one FeedBlob reaches both nets
    new_b = np.linspace(-1, 1, 157).astype(np.float32)
    workspace.FeedBlob("gpu_0/pred_b", new_b)
    assert np.array_equal(e_train.fetch_param("pred_b"), new_b) and np.array_equal(e_test.fetch_param("pred_b"), new_b)
    workspace.RunNet(test.net)
    assert not np.array_equal(workspace.FetchBlob("gpu_0/prob"), p1)
    workspace.ResetWorkspace()
