"""`ModelBuilder`: the graph-builder surface tools/train_net.py, tools/test_net.py and
tools/lfb_loader.py drive in the reference (lib/models/model_builder_video.py:66-389), kept
call-compatible so lib/models/* style builders run unchanged -- but instead of a Caffe2 NetDef
executed by cuDNN/cuBLAS, the recorded ops are lowered by vlfb.engine to fused HIP kernels for
gfx950 (libvlfb_hip.so).

Differences that are deliberate (MI355X-first):
  * one process per GPU (torch.distributed / RCCL) instead of one process driving 8 GPUs through
    Parallelize_GPU; the per-GPU namescope is still `gpu_{local_rank}/` so blob names seen through
    FetchBlob match (reference :142-157).
  * Conv -> AffineNd -> ReLU -> Sum chains are fused at lowering time; the frozen affine scale is
    folded into the MFMA weight operand.
  * SpatialBN graphs (USE_AFFINE False / NONLOCAL.USE_BN): per-GPU statistics, vlfb_bn_{fwd,bwd}.
"""
import logging

import numpy as np

from core.config import config as cfg
from models import resnet_video
from vlfb.net import Net, ParamInitNet
import utils.lr_policy as lr_policy
import utils.misc as misc

logger = logging.getLogger(__name__)

model_creator_map = {"resnet_video": resnet_video}


def _triple(v, default):
    if v is None:
        return list(default)
    v = list(v)
    return v


class ModelBuilder(object):
    """CNNModelHelper-like op emitter + parameter registry."""

    def __init__(self, **kwargs):
        self.order = "NCHW"
        self.train = kwargs.get("train", False)
        self.split = kwargs.get("split", "train")
        self.force_fw_only = kwargs.get("force_fw_only", False)
        self.name = kwargs.get("name", "vlfb_model")
        self.use_cudnn = kwargs.get("use_cudnn", True)  # accepted, meaningless here
        self.net = Net(self.name)
        self.param_init_net = ParamInitNet()
        self.params = []
        self.weights = []
        self.biases = []
        self.computed_params = []
        self.param_to_grad = {}
        self.affine_params = set()
        self.do_not_update_params = []
        self.data_loader = None
        self.input_db = None
        self.engine = None
        self.scope = ""
        self.input_blob_names = []
        self.loss_blob = None
        self.current_lr = 0
        self.SetCurrentLr(0)

    # ---- parameter bookkeeping ---------------------------------------------------------------
    def _new_param(self, name, shape, init, is_weight):
        fill, kw = init
        self.param_init_net.__getattr__(fill)([], name, shape=list(shape), **kw)
        self.params.append(name)
        (self.weights if is_weight else self.biases).append(name)
        return name

    def GetParams(self, namescope=None):
        return list(self.params)

    def GetAllParams(self, namescope=None):
        return list(self.params) + list(self.computed_params)

    def GetComputedParams(self, namescope=None):
        return list(self.computed_params)

    def TrainableParams(self, scope=""):
        return [p for p in self.params
                if p in self.param_to_grad and p not in self.do_not_update_params]

    # ---- operator emitters (names and argument meaning follow CNNModelHelper) ---------------------
    def ConvNd(self, blob_in, blob_out, dim_in, dim_out, kernel, weight_init=None, bias_init=None,
               strides=None, pads=None, dilations=None, group=1, no_bias=False, **kwargs):
        if group < 1 or dim_in % group or dim_out % group:
            raise ValueError("ConvNd %s: group = %d does not divide %d -> %d channels" % (blob_out, group, dim_in, dim_out))
        kernel = list(kernel)
        nd = len(kernel)
        # Caffe2's grouped Conv: weight (dim_out, dim_in / group, k...), output channel block g reads input channel block g
        w = self._new_param(blob_out + "_w", [dim_out, dim_in // group] + kernel,
                            weight_init or ("XavierFill", {}), True)
        inputs = [blob_in, w]
        if not no_bias:
            inputs.append(self._new_param(blob_out + "_b", [dim_out],
                                          bias_init or ("ConstantFill", {"value": 0.0}), False))
        return self.net.add("Conv", inputs, [blob_out], kernels=kernel,
                            strides=_triple(strides, [1] * nd), pads=_triple(pads, [0] * 2 * nd),
                            dilations=_triple(dilations, [1] * nd), group=group)

    def AffineNd(self, blob_in, blob_out, dim_in, share_with=None, inplace=False):
        """per-channel scale/bias with NO gradient to scale/bias (the reference's native op,
        caffe2_customized_ops/video/affine_nd_op.{cc,cu}); params `<name>_s` = 1, `<name>_b` = 0."""
        blob_out = blob_out or self.net.NextName()
        prefix = blob_out if share_with is None else share_with
        if share_with is None:
            s = self._new_param(prefix + "_s", [dim_in], ("ConstantFill", {"value": 1.0}), True)
            b = self._new_param(prefix + "_b", [dim_in], ("ConstantFill", {"value": 0.0}), False)
            self.affine_params.update((s, b))
        else:
            s, b = prefix + "_s", prefix + "_b"
        return self.net.add("AffineNd", [blob_in, s, b], [blob_in if inplace else blob_out])

    def Conv3dAffine(self, blob_in, prefix, dim_in, dim_out, kernels, strides, pads, group=1,
                     suffix="_bn", inplace_affine=False, dilations=None, **kwargs):
        """bias-free MSRA conv followed by AffineNd; extra kwargs (bn_init) are ignored exactly as
        in the reference (:200-221)."""
        conv = self.ConvNd(blob_in, prefix, dim_in, dim_out, kernels, strides=strides, pads=pads,
                           group=group, weight_init=("MSRAFill", {}),
                           bias_init=("ConstantFill", {"value": 0.0}), no_bias=1,
                           dilations=dilations if dilations is not None else [1, 1, 1])
        return self.AffineNd(conv, prefix + suffix, dim_out, inplace=inplace_affine)

    def Conv3dBN(self, blob_in, prefix, dim_in, dim_out, kernels, strides, pads, group=1, bn_init=None, **kwargs):
        """bias-free MSRA conv followed by SpatialBN (model_builder_video.py:176-197); bn_init != 1 re-fills the scale
        (the zero-initialised last BN of a bottleneck, resnet_helper.py:70).
        As in the reference, a `dilations=` argument is NOT forwarded to the convolution (it lands in **kwargs, :179):
        a batch-norm graph with cfg.DILATIONS = 2 gets the pads of the dilated 3x3 (resnet_helper.py:57) on an
        undilated kernel, res5_0_branch2b grows by 2 pixels per side and the residual Sum no longer fits -- in Caffe2
        at run time, here when the engine plans the graph.  Batch-norm graphs therefore need
        MODEL.DILATIONS_AFTER_CONV5 False, here as there (tests/test_ref_graph.py holds this to the reference's class)."""
        conv = self.ConvNd(blob_in, prefix, dim_in, dim_out, kernels, strides=strides, pads=pads,
                           group=group, weight_init=("MSRAFill", {}),
                           bias_init=("ConstantFill", {"value": 0.0}), no_bias=1)
        out = self.SpatialBN(conv, prefix + "_bn", dim_out, epsilon=cfg.MODEL.BN_EPSILON,
                             momentum=cfg.MODEL.BN_MOMENTUM, is_test=self.split in ["test", "val"])
        if bn_init is not None and bn_init != 1.0:
            self.param_init_net.ConstantFill([prefix + "_bn_s"], prefix + "_bn_s", value=bn_init)
        return out

    def SpatialBN(self, blob_in, blob_out, dim_in, epsilon=1e-5, momentum=0.9, is_test=False, **kwargs):
        """CNNModelHelper.SpatialBN: scale `<name>_s` = 1 and bias `<name>_b` = 0 are trained, the running statistics
        `<name>_rm` = 0 / `<name>_riv` = 1 are computed parameters (saved with the model, never handed to the solver)"""
        s = self._new_param(blob_out + "_s", [dim_in], ("ConstantFill", {"value": 1.0}), True)
        b = self._new_param(blob_out + "_b", [dim_in], ("ConstantFill", {"value": 0.0}), False)
        rm, riv = blob_out + "_rm", blob_out + "_riv"
        self.param_init_net.ConstantFill([], rm, shape=[dim_in], value=0.0)
        self.param_init_net.ConstantFill([], riv, shape=[dim_in], value=1.0)
        self.computed_params += [rm, riv]
        return self.net.add("SpatialBN", [blob_in, s, b, rm, riv], [blob_out], epsilon=float(epsilon),
                            momentum=float(momentum), is_test=int(bool(is_test)))

    def Relu(self, blob_in, blob_out, **kwargs):
        return self.net.add("Relu", [blob_in], [blob_out])

    def Relu_(self, blob_in):
        out = blob_in if cfg.MODEL.ALLOW_INPLACE_RELU else blob_in + "_relu"
        return self.Relu(blob_in, out)

    def MaxPool(self, blob_in, blob_out, kernels=None, strides=None, pads=None, **kwargs):
        return self.net.add("MaxPool", [blob_in], [blob_out], kernels=list(kernels),
                            strides=list(strides), pads=list(pads))

    def AveragePool(self, blob_in, blob_out, kernels=None, strides=None, pads=None, **kwargs):
        return self.net.add("AveragePool", [blob_in], [blob_out], kernels=list(kernels),
                            strides=list(strides), pads=list(pads))

    def Reshape(self, blob_in, blobs_out, shape=None, **kwargs):
        ins = blob_in if isinstance(blob_in, (list, tuple)) else [blob_in]
        return self.net.add("Reshape", ins, blobs_out, shape=None if shape is None else list(shape))

    def Transpose(self, blob_in, blob_out, axes=None, **kwargs):
        return self.net.add("Transpose", [blob_in], [blob_out], axes=list(axes))

    def Squeeze(self, blob_in, blob_out, dims=None, **kwargs):
        return self.net.add("Squeeze", [blob_in], [blob_out], dims=list(dims))

    def Softmax(self, blob_in, blob_out, axis=1, **kwargs):
        return self.net.add("Softmax", [blob_in], [blob_out], axis=axis)

    def Scale(self, blob_in, blob_out, scale=1.0, **kwargs):
        return self.net.add("Scale", [blob_in], [blob_out], scale=float(scale))

    def LayerNorm(self, blob_in, blobs_out, axis=1, epsilon=1e-5, **kwargs):
        return self.net.add("LayerNorm", [blob_in], blobs_out, axis=axis, epsilon=epsilon)

    def Dropout(self, blob_in, blob_out, ratio=0.5, is_test=False, **kwargs):
        return self.net.add("Dropout", [blob_in], [blob_out], ratio=float(ratio), is_test=bool(is_test))

    def FC(self, blob_in, blob_out, dim_in, dim_out, weight_init=None, bias_init=None, **kwargs):
        w = self._new_param(blob_out + "_w", [dim_out, dim_in], weight_init or ("XavierFill", {}), True)
        b = self._new_param(blob_out + "_b", [dim_out], bias_init or ("ConstantFill", {"value": 0.0}), False)
        return self.net.add("FC", [blob_in, w, b], [blob_out])

    def Sigmoid(self, blob_in, blob_out, **kwargs):
        return self.net.add("Sigmoid", [blob_in], [blob_out])

    def SigmoidCrossEntropyLoss(self, blobs_in, blobs_out, scale=1.0, normalize=1, **kwargs):
        return self.net.add("SigmoidCrossEntropyLoss", blobs_in, blobs_out, scale=float(scale),
                            normalize=normalize)

    def SoftmaxWithLoss(self, blobs_in, blobs_out, scale=1.0, **kwargs):
        return self.net.add("SoftmaxWithLoss", blobs_in, blobs_out, scale=float(scale))

    def StopGradient(self, blob_in, blob_out, **kwargs):
        return self.net.add("StopGradient", [blob_in], [blob_out])

    def RoIAlign(self, blobs_in, blobs_out, pooled_w=7, pooled_h=7, spatial_scale=1.0 / 16,
                 sampling_ratio=0, **kwargs):
        return self.net.add("RoIAlign", blobs_in, blobs_out, pooled_w=pooled_w, pooled_h=pooled_h,
                            spatial_scale=float(spatial_scale), sampling_ratio=sampling_ratio)

    def DequeueBlobs(self, queue_name, blob_names):
        # inputs are fed straight into device tensors (vlfb.workspace.FeedBlob); there is no
        # Caffe2 BlobsQueue here
        self.input_blob_names = [str(b) for b in blob_names]
        return self.input_blob_names

    def WeightedSum(self, blobs_in, blob_out):
        return self.net.add("WeightedSum", blobs_in, [blob_out])

    # ---- learning rate -----------------------------------------------------------------------
    def SetCurrentLr(self, cur_iter):
        self.current_lr = lr_policy.get_lr_at_iter(cur_iter)

    def UpdateWorkspaceLr(self, cur_iter):
        """(reference :258-284) new LR into the solver; optional momentum correction on jumps"""
        new_lr = lr_policy.get_lr_at_iter(cur_iter)
        if new_lr != self.current_lr:
            ratio = _get_lr_change_ratio(self.current_lr, new_lr)
            if ratio > 1.1:
                logger.info("Setting learning rate to {:.6f} at iteration {}".format(new_lr, cur_iter))
            self._SetNewLr(self.current_lr, new_lr)

    def _SetNewLr(self, cur_lr, new_lr):
        assert cur_lr > 0
        ratio = _get_lr_change_ratio(cur_lr, new_lr)
        if cfg.SOLVER.SCALE_MOMENTUM and cur_lr > 1e-7 and ratio > cfg.SOLVER.SCALE_MOMENTUM_THRESHOLD:
            self._CorrectMomentum(new_lr / cur_lr)
        self.current_lr = new_lr
        if self.engine is not None:
            self.engine.set_lr(float(new_lr))

    def _CorrectMomentum(self, correction):
        if correction < 0.9 or correction > 1.1:
            logger.info("Scaling update history by {:.6f} (new/old lr)".format(correction))
        if self.engine is not None:
            self.engine.scale_momentum(float(correction))

    # ---- model construction -------------------------------------------------------------------
    def build_model(self, suffix, lfb=None, lfb_infer_only=False, shift=1, node_id=0):
        """Record the per-GPU replica graph.  The reference also creates its dataset / DataLoader
        here (:97-117); real-data I/O is out of scope -- blobs are fed with
        vlfb.workspace.FeedBlob (synthetic clips in bench/tests)."""
        self.suffix = suffix
        self.lfb_infer_only = lfb_infer_only
        self.crop_size = misc.get_crop_size(self.split)
        self.create_data_parallel_model(
            model=self, db_loader=None, split=self.split, node_id=node_id, train=self.train,
            force_fw_only=self.force_fw_only, suffix=suffix, lfb_infer_only=lfb_infer_only)

    def create_data_parallel_model(self, model, db_loader, split, node_id, train=True,
                                   force_fw_only=False, suffix="", lfb_infer_only=False):
        from vlfb import dist
        self.scope = "gpu_{}/".format(dist.local_rank() + cfg.ROOT_GPU_ID)
        forward_pass = create_model(model=self, split=split, suffix=suffix, lfb_infer_only=lfb_infer_only)
        add_inputs(model=self, data_loader=db_loader, suffix=suffix)(self)
        losses = forward_pass(self, 1.0 / cfg.NUM_GPUS)
        self.loss_blob = losses[0] if losses and losses[0] is not None else None
        self.with_update = bool(train and not force_fw_only)
        self._derive_param_to_grad()

    def _derive_param_to_grad(self):
        """which parameters receive a gradient: reverse reachability from the loss, cut at
        StopGradient; AffineNd has no gradient for scale/bias (affine_nd_op.cc:45-53)."""
        self.param_to_grad = {}
        if self.loss_blob is None or not self.with_update:
            return
        from vlfb.net import ssa_form
        ssa = ssa_form(self.net.ops)
        last = {}
        for _, _, outs in ssa:
            for b, v in outs:
                last[b] = v
        live = {(self.loss_blob, last.get(self.loss_blob, 0))}
        for op, ins, outs in reversed(ssa):
            if op.type == "StopGradient" or not any(o in live for o in outs):
                continue
            if op.type == "AffineNd":
                live.add(ins[0])
            else:
                live.update(ins)
        for p in self.params:
            if (p, 0) in live and p not in self.affine_params:
                self.param_to_grad[p] = p + "_grad"

    def start_data_loader(self):
        logger.info("no data loader: inputs are fed through vlfb.workspace.FeedBlob")

    def shutdown_data_loader(self):
        pass


def create_model(model, split, suffix, lfb_infer_only):
    model_name = cfg.MODEL.MODEL_NAME
    assert model_name in model_creator_map, "Unknown model_type {}".format(model_name)

    def model_creator(model, loss_scale):
        _, _, loss = model_creator_map[model_name].create_model(
            model=model, data="data{}".format(suffix), labels="labels{}".format(suffix),
            split=split, suffix=suffix, lfb_infer_only=lfb_infer_only)
        return [loss]
    return model_creator


def add_inputs(model, data_loader, suffix):
    def input_fn(model):
        names = ["data{}".format(suffix), "labels{}".format(suffix)]
        if cfg.DATASET == "ava":
            names += ["proposals{}".format(suffix)]
        if cfg.LFB.ENABLED and not getattr(model, "lfb_infer_only", False):
            names += ["lfb{}".format(suffix)]
        model.DequeueBlobs("vlfb_feed", names)
        model.StopGradient("data{}".format(suffix), "data{}".format(suffix))
    return input_fn


def add_parameter_update_ops(model):
    """The reference emits WeightedSum + MomentumSGDUpdate per parameter (:348-389); here the
    solver is one fused kernel over the flat parameter bucket (vlfb.engine.Engine.sgd_step)."""
    def param_update_ops(model):
        model.with_update = True
    return param_update_ops


def _get_lr_change_ratio(cur_lr, new_lr):
    eps = 1e-10
    return np.max((new_lr / np.max((cur_lr, eps)), cur_lr / np.max((new_lr, eps))))
