"""Residual stages of the 3-D ResNet (call-compatible with the reference's
lib/models/resnet_helper.py:35-194).  The emitted op sequence per block is what the lowering
pass fuses into three MFMA convolutions with affine/ReLU/residual epilogues."""
import logging

import numpy as np

from core.config import config as cfg
import models.nonlocal_helper as nonlocal_helper

logger = logging.getLogger(__name__)


def _conv_op(model):
    return model.Conv3dAffine if cfg.MODEL.USE_AFFINE else model.Conv3dBN


def bottleneck_transformation_3d(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner,
                                 group=1, use_temp_conv=1, temp_stride=1):
    """(k x 1 x 1) -> (1 x 3 x 3) -> (1 x 1 x 1); the temporal kernel sits in the first conv."""
    conv = _conv_op(model)
    dil = cfg.DILATIONS
    kt = 1 + 2 * use_temp_conv
    layers = (
        ("_branch2a", dim_in, dim_inner, [kt, 1, 1], [temp_stride, 1, 1], [use_temp_conv, 0, 0], None, True, {}),
        ("_branch2b", dim_inner, dim_inner, [1, 3, 3], [1, stride, stride], [0, dil, dil], [1, dil, dil], True,
         {"group": group}),
        ("_branch2c", dim_inner, dim_out, [1, 1, 1], [1, 1, 1], [0, 0, 0], None, False,
         {"bn_init": cfg.MODEL.BN_INIT_GAMMA}),
    )
    blob = blob_in
    for suffix, cin, cout, kernel, strides, pad, dilations, relu, extra in layers:
        kw = dict(extra)
        if dilations is not None:
            kw["dilations"] = dilations
        blob = conv(blob, prefix + suffix, cin, cout, kernel, strides=strides, pads=pad * 2,
                    inplace_affine=False, **kw)
        if suffix == "_branch2b":
            logger.info("%s using dilation %d" % (prefix, dil))
        if relu:
            blob = model.Relu_(blob)
    return blob


def _add_shortcut_3d(model, blob_in, prefix, dim_in, dim_out, stride, temp_stride=1):
    """type-B shortcut: identity unless the shape changes"""
    if dim_in == dim_out and temp_stride == 1 and stride == 1:
        return blob_in
    return _conv_op(model)(blob_in, prefix, dim_in, dim_out, [1, 1, 1],
                           strides=[temp_stride, stride, stride], pads=[0, 0, 0] * 2, group=1,
                           inplace_affine=False)


def _generic_residual_block_3d(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner,
                               group=1, use_temp_conv=0, temp_stride=1, trans_func=None):
    """relu(F(x) + shortcut(x))"""
    if trans_func is None:
        trans_func = globals()[cfg.RESNETS.TRANS_FUNC]
    branch = trans_func(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner, group=group,
                        use_temp_conv=use_temp_conv, temp_stride=temp_stride)
    shortcut = _add_shortcut_3d(model, blob_in, prefix + "_branch1", dim_in, dim_out, stride,
                                temp_stride=temp_stride)
    total = model.net.Sum([branch, shortcut],
                          branch if cfg.MODEL.ALLOW_INPLACE_SUM else prefix + "_sum")
    return model.Relu_(total)


def _stage(model, blob_in, dim_in, dim_out, stride, num_blocks, prefix, dim_inner, group,
           use_temp_convs, temp_strides, after_block):
    use_temp_convs = list(np.zeros(num_blocks).astype(int) if use_temp_convs is None else use_temp_convs)
    temp_strides = list(np.ones(num_blocks).astype(int) if temp_strides is None else temp_strides)
    while len(use_temp_convs) < num_blocks:
        use_temp_convs.append(0)
        temp_strides.append(1)
    for idx in range(num_blocks):
        blob_in = _generic_residual_block_3d(
            model, blob_in, dim_in, dim_out, 2 if (idx == 0 and stride == 2) else 1,
            "{}_{}".format(prefix, idx), dim_inner, group, use_temp_convs[idx], temp_strides[idx])
        dim_in = dim_out
        blob_in = after_block(idx, blob_in, dim_in)
    return blob_in, dim_in


def res_stage_nonlocal(model, block_fn, blob_in, dim_in, dim_out, stride, num_blocks, prefix,
                       dim_inner=None, group=None, use_temp_convs=None, temp_strides=None,
                       batch_size=None, nonlocal_name=None, nonlocal_mod=1000):
    """a stage with a (full space-time) non-local block after every `nonlocal_mod`-th block"""
    def after(idx, blob, dim):
        if idx % nonlocal_mod == nonlocal_mod - 1:
            return nonlocal_helper.add_nonlocal(model, blob, dim, dim, batch_size,
                                                nonlocal_name + "_{}".format(idx), int(dim / 2))
        return blob
    return _stage(model, blob_in, dim_in, dim_out, stride, num_blocks, prefix, dim_inner, group,
                  use_temp_convs, temp_strides, after)


def res_stage_nonlocal_group(model, block_fn, blob_in, dim_in, dim_out, stride, num_blocks, prefix,
                             dim_inner=None, group=None, use_temp_convs=None, temp_strides=None,
                             batch_size=None, pool_stride=None, spatial_dim=None, group_size=None,
                             nonlocal_name=None, nonlocal_mod=1000):
    """as above, but the non-local blocks attend within temporal groups of `group_size` frames"""
    def after(idx, blob, dim):
        if idx % nonlocal_mod == nonlocal_mod - 1:
            return nonlocal_helper.add_nonlocal_group(
                model, blob, dim, dim, batch_size, pool_stride, spatial_dim, spatial_dim, group_size,
                nonlocal_name + "_{}".format(idx), int(dim / 2))
        return blob
    return _stage(model, blob_in, dim_in, dim_out, stride, num_blocks, prefix, dim_inner, group,
                  use_temp_convs, temp_strides, after)
