"""train_step() solves every gradient bucket during backward, as soon as the bucket is final (Engine._bucket_ready:
solver + operand refresh on a third HIP stream behind the bucket's all-reduce), instead of one solver pass after
backward.  Same kernels on the same values, so after several steps the parameters, the momentum and the MFMA
operand copies must be BIT-IDENTICAL to the reference order (backward, then sgd_step)."""
import collections

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(eager, dtype):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from vlfb import synth
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64])
    model = ModelBuilder(train=True, split="train", name="eager_%s" % eager)
    model.build_model(suffix="_train")
    eng = Engine(model, dtype, device="cuda:0", base_seed=3)
    eng.EAGER_SOLVER = eager
    batch = synth.inputs(cfg, 2, 2, seed=3, crop=64, frames=8)
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
    eng._plan_solver_buckets(bucket_mb=4)           # several buckets even at this size
    eng.feed_params(synth.params(model, seed=3))
    for k, v in batch.items():
        if k in model.input_blob_names:
            eng.feed(k, v)
    return eng


@pytest.mark.parametrize("mode", [True, "tail"], ids=["eager", "tail"])
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_bucketwise_solver_is_bit_identical_to_the_solver_pass_after_backward(dtype, mode):
    """mode True: every bucket as soon as it is final; "tail": the finished buckets together, beside the last wgrad"""
    from vlfb.engine import ConvStep
    a, b = _engine(mode, dtype), _engine(False, dtype)
    assert len(a.sol_buckets) > 3
    covered = sorted((x["start"], x["end"]) for x in a.sol_buckets)
    assert covered[0][0] == 0 and all(covered[i][1] <= covered[i + 1][0] for i in range(len(covered) - 1))
    assert sum(len(x["names"]) for x in a.sol_buckets) == len(a.train_order)
    for it in range(3):
        lr = 0.01 * (it + 1)
        a.train_step(lr)
        b.forward(); b.backward(); b.sgd_step(lr)
    torch.cuda.synchronize()
    assert a.iteration == b.iteration == 3
    assert torch.equal(a.flat_param, b.flat_param) and float(a.flat_param.abs().sum()) > 0
    assert torch.equal(a.flat_mom, b.flat_mom)
    convs_a = [s for s in a.steps if isinstance(s, ConvStep)]
    convs_b = [s for s in b.steps if isinstance(s, ConvStep)]
    for sa, sb in zip(convs_a, convs_b):
        assert torch.equal(sa.w_f.view(torch.int16 if dtype == "bf16" else torch.int32),
                           sb.w_f.view(torch.int16 if dtype == "bf16" else torch.int32)), sa.wname
        if sa.w_d is not None:
            assert torch.equal(sa.w_d.view(torch.int16 if dtype == "bf16" else torch.int32),
                               sb.w_d.view(torch.int16 if dtype == "bf16" else torch.int32)), sa.wname
        if sa.eff_bias is not None:
            assert torch.equal(sa.eff_bias, sb.eff_bias)
    la, lb = a.recent_losses(), b.recent_losses()
    assert la == lb and len(la) == 3


def test_forward_branches_on_the_second_stream_change_no_bit():
    """Engine.forward runs the projection shortcuts and the pool -> phi / g branches of the non-local blocks on the
    parameter-gradient stream (events on the edges that cross streams).  Same kernels, same data: activations, loss
    and the parameters after two steps are bit-identical to the one-stream order."""
    a, b = _engine(False, "bf16"), _engine(False, "bf16")
    b.FORWARD_BRANCHES = False
    assert len(a._fwd_side) >= 3 * 5 + 2 and all(a._fwd_wait[i] for i in a._fwd_side if a.steps[i].name().startswith("maxpool"))
    for it in range(2):
        a.train_step(0.02)
        b.train_step(0.02)
    torch.cuda.synchronize()
    assert torch.equal(a.flat_param, b.flat_param) and torch.equal(a.flat_mom, b.flat_mom)
    assert a.recent_losses() == b.recent_losses()
    for name in ("nonlocal_conv3_1_sum", "res5_2_branch2c_bn", "res2_0_branch2c_bn"):
        ta, tb = a.env[name].root.tensor, b.env[name].root.tensor
        assert torch.equal(ta.view(torch.int16), tb.view(torch.int16)), name
