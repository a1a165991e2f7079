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
    assert not np.allclose(workspace.FetchBlob("gpu_0/pred_w"), w0)                      # head is trained
    assert np.array_equal(workspace.FetchBlob("gpu_0/res4_2_branch2a_w"), frozen0)        # FREEZE_BACKBONE
    assert workspace.FetchBlob("gpu_0/pred_w_momentum").shape == (157, 2560)
    assert workspace.FetchBlob("gpu_0/prob").shape == (2, 157)
    assert workspace.FetchBlob("gpu_0/pool5").shape == (2, 2560, 1, 1, 1)
    # feeding a parameter in the reference layout round-trips
    new = np.random.default_rng(0).standard_normal((157, 2560)).astype(np.float32)
    workspace.FeedBlob("gpu_0/pred_w", new)
    assert np.array_equal(workspace.FetchBlob("gpu_0/pred_w"), new)
    workspace.ResetWorkspace()
