"""With Engine.STEP_GRAPH set, train_step() replays the whole step as one captured HIP graph from its second call on: the
same kernels on the same buffers, with the learning rate and the dropout seeds read from device memory instead of
the launch packet.  Parameters, momentum, losses and dropout masks must be BIT-IDENTICAL to the stream path, also
when the learning rate changes every step, when graph and stream steps alternate, and when the captured step forks
onto the third (solver) stream."""
import collections

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(graph, dtype="bf16", eager_solver=False, name="g"):
    from vlfb.presets import load_preset
    from core.config import config as cfg
    from models.model_builder_video import ModelBuilder
    from vlfb.engine import Engine
    from vlfb import synth
    load_preset("ava_r50_lfb_nl", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "TRAIN.VIDEO_LENGTH", 8, "TRAIN.CROP_SIZE", 64])
    model = ModelBuilder(train=True, split="train", name="graph_%s_%s" % (name, graph))
    model.build_model(suffix="_train")
    eng = Engine(model, dtype, device="cuda:0", base_seed=5)
    eng.STEP_GRAPH = graph
    eng.STEP_TRACE = False          # the reference side of every comparison walks the step objects
    eng.EAGER_SOLVER = eager_solver
    batch = synth.inputs(cfg, 2, 2, seed=5, crop=64, frames=8)
    eng.plan(collections.OrderedDict((k, v.shape) for k, v in batch.items() if k in model.input_blob_names))
    eng._plan_solver_buckets(bucket_mb=4)
    eng.feed_params(synth.params(model, seed=5))
    for k, v in batch.items():
        if k in model.input_blob_names:
            eng.feed(k, v)
    return eng


def _same_state(a, b):
    from vlfb.engine import DropoutStep
    torch.cuda.synchronize()
    assert a.iteration == b.iteration
    assert torch.equal(a.flat_param, b.flat_param) and float(a.flat_param.abs().sum()) > 0
    assert torch.equal(a.flat_mom, b.flat_mom)
    da = [s for s in a.steps if isinstance(s, DropoutStep)]
    db = [s for s in b.steps if isinstance(s, DropoutStep)]
    assert len(da) >= 2                          # the FBO dropouts and the classifier dropout
    for sa, sb in zip(da, db):
        assert torch.equal(sa.mask, sb.mask) and 0 < int(sa.mask.sum()) < sa.mask.numel()


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_captured_step_is_bit_identical_to_the_stream_step(dtype):
    a, b = _engine(True, dtype), _engine(False, dtype)
    masks = []
    for it in range(5):
        lr = 0.01 * (it + 1)                     # a new learning rate and new dropout seeds every replay
        a.train_step(lr)
        b.train_step(lr)
        masks.append([s.mask.clone() for s in a._drop_steps])
    assert a._graph is not None and b._graph is None
    _same_state(a, b)
    assert not torch.equal(masks[-1][0], masks[-2][0])          # the seeds did change between replays
    la, lb = a.recent_losses(), b.recent_losses()
    assert la == lb and len(la) == 5
    # graph and stream steps mix: the pieces called one by one, then a replay again
    a.forward(); a.backward(); a.sgd_step(0.03)
    b.train_step(0.03)
    a.train_step(0.02)
    b.train_step(0.02)
    _same_state(a, b)
    assert a.recent_losses() == b.recent_losses()


def test_captured_step_with_the_bucketwise_solver_on_the_third_stream():
    a, b = _engine(True, "bf16", eager_solver=True, name="e"), _engine(False, "bf16", eager_solver=False, name="e")
    for it in range(4):
        a.train_step(0.02)
        b.train_step(0.02)
    assert a._graph is not None
    _same_state(a, b)
    assert a.recent_losses() == b.recent_losses()


def test_forward_only_capture():
    a, b = _engine("forward", name="f"), _engine(False, name="f")
    for it in range(4):
        a.train_step(0.01 * (it + 1))
        b.train_step(0.01 * (it + 1))
    assert a._graph is not None
    _same_state(a, b)
    assert a.recent_losses() == b.recent_losses()


def test_parameters_fed_between_replays_are_picked_up():
    """feed_params() between steps changes the fp32 parameters behind the captured step's back: the operand copies are
    rebuilt before the next replay (Engine._graph_step checks the parameter version like forward() does)"""
    from vlfb import synth
    a, b = _engine(True), _engine(False)
    for eng in (a, b):
        eng.train_step(0.01)
        eng.train_step(0.01)
        eng.feed_params(synth.params(eng.model, seed=11))
        eng.train_step(0.01)
    assert a._graph is not None
    _same_state(a, b)


# ---- Engine.STEP_TRACE: the same idea one level up -- the recorded list of library calls and stream edges is re-issued
# on the ordinary streams (no HIP graph), so only the Python between the calls disappears
def _trace_engine(dtype="bf16", eager_solver=False, name="t"):
    eng = _engine(False, dtype, eager_solver, name="trace_" + name)
    eng.STEP_TRACE = True
    return eng


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32", "split", "mix"])
def test_replayed_trace_is_bit_identical_to_the_stream_step(dtype):
    a, b = _trace_engine(dtype), _engine(False, dtype)
    masks = []
    for it in range(5):
        lr = 0.01 * (it + 1)
        a.train_step(lr)
        b.train_step(lr)
        masks.append([s.mask.clone() for s in a._drop_steps])
    assert a._trace is not None and len(a._trace) > 100 and b._trace is None
    _same_state(a, b)
    assert not torch.equal(masks[-1][0], masks[-2][0])
    la, lb = a.recent_losses(), b.recent_losses()
    assert la == lb and len(la) == 5
    # the pieces one by one, then a replay again; parameters fed behind the trace's back
    from vlfb import synth
    a.forward(); a.backward(); a.sgd_step(0.03)
    b.train_step(0.03)
    for eng in (a, b):
        eng.feed_params(synth.params(eng.model, seed=11))
        eng.train_step(0.02)
        eng.train_step(0.02)
    _same_state(a, b)
    assert a.recent_losses() == b.recent_losses()


def test_trace_is_recorded_again_on_another_stream_and_with_the_third_stream_solver():
    a, b = _trace_engine("bf16", eager_solver=True, name="e"), _engine(False, "bf16", eager_solver=False, name="e")
    for it in range(3):
        a.train_step(0.02)
        b.train_step(0.02)
    first = a._trace
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for it in range(3):
            a.train_step(0.02)
    torch.cuda.current_stream().wait_stream(s)
    for it in range(3):
        b.train_step(0.02)
    assert a._trace is not first
    _same_state(a, b)
    assert a.recent_losses() == b.recent_losses()


@pytest.mark.parametrize("dtype", ["bf16", "split"])
def test_parameter_gradient_placement_does_not_change_the_step(dtype):
    """Engine.WGRAD_LAG moves the parameter-gradient launches of a backward step behind later dgrads (or, at infinity, behind
    the whole chain): the same kernels on the same operands, so parameters, momentum and losses stay bit-identical -- on
    the step-object path and on the recorded one"""
    ref = _engine(False, dtype, name="lag0")
    variants = []
    for lag, trace in ((3, False), (1 << 20, False), (7, True)):
        e = _engine(False, dtype, name="lag%d%d" % (lag, trace))
        e.WGRAD_LAG, e.STEP_TRACE = lag, trace
        variants.append(e)
    for it in range(4):
        for e in [ref] + variants:
            e.train_step(0.01 * (it + 1))
    want = ref.recent_losses()
    for e in variants:
        _same_state(e, ref)
        assert e.recent_losses() == want
    assert variants[2]._trace is not None
