"""Space-time non-local block (https://arxiv.org/abs/1711.07971), call-compatible with the
reference's lib/models/nonlocal_helper.py:29-213.  The BatchMatMul/Scale/Softmax/BatchMatMul run
recorded here is lowered to batched MFMA GEMMs + a wavefront-reduction softmax; the
Transpose/Reshape/Transpose wrappers of the grouped variant are pure views in the
channels-last layout and cost nothing."""
from core.config import config as cfg


def _pointwise(model, blob_in, name, dim_in, dim_out, weight_init):
    return model.ConvNd(blob_in, name, dim_in, dim_out, [1, 1, 1], strides=[1, 1, 1],
                        pads=[0, 0, 0] * 2, weight_init=weight_init,
                        bias_init=("ConstantFill", {"value": 0.}), no_bias=cfg.NONLOCAL.NO_BIAS)


def _flatten3(model, blob, batch_size, dim_inner):
    """(B, C, T, H, W) -> (B, C, T*H*W); second output keeps the 5-d shape"""
    out = blob if cfg.MODEL.ALLOW_INPLACE_RESHAPE else blob + "_re"
    return model.Reshape(blob, [out, blob + "_shape5d"], shape=(batch_size, dim_inner, -1))


def spacetime_nonlocal(model, blob_in, dim_in, dim_out, batch_size, prefix, dim_inner, is_test,
                       max_pool_stride=2):
    gauss = ("GaussianFill", {"std": cfg.NONLOCAL.CONV_INIT_STD})
    theta = _pointwise(model, blob_in, prefix + "_theta", dim_in, dim_inner, gauss)
    if cfg.NONLOCAL.USE_MAXPOOL is True:
        keys_in = model.MaxPool(blob_in, prefix + "_pool",
                                kernels=[1, max_pool_stride, max_pool_stride],
                                strides=[1, max_pool_stride, max_pool_stride], pads=[0, 0, 0] * 2)
    else:
        keys_in = blob_in
    phi = _pointwise(model, keys_in, prefix + "_phi", dim_in, dim_inner, gauss)
    g = _pointwise(model, keys_in, prefix + "_g", dim_in, dim_inner, gauss)

    theta, theta_shape_5d = _flatten3(model, theta, batch_size, dim_inner)
    phi, _ = _flatten3(model, phi, batch_size, dim_inner)
    g, _ = _flatten3(model, g, batch_size, dim_inner)

    affinity = model.net.BatchMatMul([theta, phi], prefix + "_affinity", trans_a=1)
    if cfg.NONLOCAL.USE_SOFTMAX is True:
        scaled = affinity
        if cfg.NONLOCAL.USE_SCALE is True:
            scaled = model.Scale(affinity, affinity, scale=dim_inner ** -.5)
        p = model.Softmax(scaled, affinity + "_prob", engine="CUDNN", axis=2)
    else:
        # dot-product variant (nonlocal_helper.py:107-119): the affinity divided by the number of keys -- a blob of ones
        # summed over the last axis, broadcast back, detached
        ones = model.net.ConstantFill([affinity], [affinity + "_ones"], value=1.)
        ones = model.net.ReduceBackSum([ones], [affinity + "_const"])
        zeros = model.net.ConstantFill([affinity], [affinity + "_zeros"], value=0.)
        denom = model.net.Add([zeros, ones], [affinity + "_denom"], broadcast=1, axis=0)
        model.StopGradient(denom, denom)
        p = model.net.Div([affinity, denom], [affinity + "_sc"])

    t = model.net.BatchMatMul([g, p], prefix + "_y", trans_b=1)
    t_re, _ = model.Reshape(
        [t, theta_shape_5d],
        [t if cfg.MODEL.ALLOW_INPLACE_RESHAPE else t + "_re", t + "_shape3d"])

    zero_or_gauss = (("ConstantFill", {"value": 0.}) if cfg.NONLOCAL.USE_ZERO_INIT_CONV else gauss)
    blob_out = _pointwise(model, t_re, prefix + "_out", dim_inner, dim_out, zero_or_gauss)
    if cfg.NONLOCAL.USE_BN:
        blob_out = model.SpatialBN(blob_out, prefix + "_bn", dim_out, epsilon=cfg.NONLOCAL.BN_EPSILON,
                                   momentum=cfg.NONLOCAL.BN_MOMENTUM, is_test=is_test)
        model.param_init_net.ConstantFill([prefix + "_bn_s"], prefix + "_bn_s", value=cfg.NONLOCAL.BN_INIT_GAMMA)
    if cfg.NONLOCAL.USE_AFFINE is True:
        blob_out = model.AffineNd(blob_out, prefix + "_bn", dim_out)
    return blob_out


def add_nonlocal(model, blob_in, dim_in, dim_out, batch_size, prefix, dim_inner):
    is_test = model.split in ["test", "val"]
    nl = spacetime_nonlocal(model, blob_in, dim_in, dim_out, batch_size, prefix, dim_inner, is_test)
    return model.net.Sum([blob_in, nl], prefix + "_sum")


def _to_groups(model, blob, shape):
    """(N, C, T, H, W) <-> (N*G, C, T/G, H, W) by transpose / reshape / transpose"""
    blob = model.Transpose(blob, blob + "_trans", axes=(0, 2, 1, 3, 4))
    if isinstance(shape, str):
        blob, shp = model.Reshape([blob, shape],
                                  [blob if cfg.MODEL.ALLOW_INPLACE_RESHAPE else blob + "_re", blob + "_shape5d"])
    else:
        blob, shp = model.Reshape(blob,
                                  [blob if cfg.MODEL.ALLOW_INPLACE_RESHAPE else blob + "_re", blob + "_shape5d"],
                                  shape=shape)
    return model.Transpose(blob, blob + "_trans", axes=(0, 2, 1, 3, 4)), shp


def add_nonlocal_group(model, blob_in, dim_in, dim_out, batch_size, pool_stride, height, width,
                       group_size, prefix, dim_inner):
    is_test = model.split in ["test", "val"]
    group_num = int(pool_stride / group_size)
    assert pool_stride % group_size == 0
    orig_shape = None
    if group_num > 1:
        blob_in, orig_shape = _to_groups(model, blob_in,
                                         (batch_size * group_num, group_size, dim_in, height, width))
    nl = spacetime_nonlocal(model, blob_in, dim_in, dim_out, batch_size * group_num, prefix,
                            dim_inner, is_test)
    blob_out = model.net.Sum([blob_in, nl], prefix + "_sum")
    if group_num > 1:
        blob_out, _ = _to_groups(model, blob_out, orig_shape)
    return blob_out
