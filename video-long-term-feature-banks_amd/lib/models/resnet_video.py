"""3-D ResNet (C2D / I3D, R50 / R101) with non-local blocks: the full per-GPU graph
(call-compatible with the reference's lib/models/resnet_video.py:33-351)."""
import logging

from core.config import config as cfg
from utils.misc import get_batch_size
import models.head_helper as head_helper
import models.resnet_helper as resnet_helper

logger = logging.getLogger(__name__)

BLOCK_CONFIG = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

# VIDEO_ARC_CHOICE -> (inflated?, depth)
_ARCS = {1: (False, 50), 2: (True, 50), 3: (False, 101), 4: (True, 101)}


def obtain_arc(arc_type):
    """Temporal kernel radius (1 -> kernel 3, 2 -> kernel 5) and temporal stride of every block,
    for conv1 and res2..res5.  I3D inflates conv1 to 5 frames, every res2 block, and every other
    block of res3/res4 (starting with the first) plus the middle block of res5."""
    if arc_type not in _ARCS:
        raise ValueError("unknown VIDEO_ARC_CHOICE {}".format(arc_type))
    inflated, depth = _ARCS[arc_type]
    n_blocks = (1,) + BLOCK_CONFIG[depth]
    if inflated:
        radius = [[2], [1] * n_blocks[1],
                  [1 - (i % 2) for i in range(n_blocks[2])],
                  [1 - (i % 2) for i in range(n_blocks[3])],
                  [0, 1, 0]]
    else:
        radius = [[0] * n for n in n_blocks]
    strides = [[1] * n for n in n_blocks]
    pool_stride = int(cfg.TRAIN.VIDEO_LENGTH / 2)
    return radius, strides, pool_stride


def _nonlocal_period(stage):
    """how often a non-local block follows a residual block in res3 / res4"""
    mod = cfg.NONLOCAL.LAYER_MOD
    if stage == 3:
        if cfg.MODEL.DEPTH == 101:
            mod = 2
        return mod if cfg.NONLOCAL.CONV3_NONLOCAL else 1000
    if cfg.MODEL.DEPTH == 101:
        mod = mod * 4 - 1
    return mod if cfg.NONLOCAL.CONV4_NONLOCAL else 1000


def create_model(model, data, labels, split, lfb_infer_only, suffix=''):
    """conv1 -> pool1 -> res2 -> pool2 -> res3 (+grouped NL) -> res4 (+NL) -> res5 -> head ->
    dropout -> FC -> sigmoid / loss."""
    cfg.DILATIONS = 1
    group = cfg.RESNETS.NUM_GROUPS
    width_per_group = cfg.RESNETS.WIDTH_PER_GROUP
    batch_size = get_batch_size(split)
    logger.info('ResNet-{} {}x{}d-{}, {}, {}, infer LFB? {}, suffix: "{}"'.format(
        cfg.MODEL.DEPTH, group, width_per_group, cfg.RESNETS.TRANS_FUNC, cfg.DATASET, split,
        lfb_infer_only, suffix))
    assert cfg.MODEL.DEPTH in BLOCK_CONFIG, 'Block config is not defined for specified model depth.'
    n1, n2, n3, n4 = BLOCK_CONFIG[cfg.MODEL.DEPTH]
    res_block = resnet_helper._generic_residual_block_3d
    dim_inner = group * width_per_group
    train_crop = split == 'train' and not lfb_infer_only
    crop_size = cfg.TRAIN.CROP_SIZE if train_crop else cfg.TEST.CROP_SIZE
    out_spatial_dim = crop_size // 16
    test_mode = split in ['test', 'val']

    radius, tstrides, pool_stride = obtain_arc(cfg.MODEL.VIDEO_ARC_CHOICE)
    logger.info("use_temp_convs_set: {}".format(radius))
    logger.info("temp_strides_set: {}".format(tstrides))

    # ---- stem ----------------------------------------------------------------------------------
    r1 = radius[0][0]
    stem = model.ConvNd(data, 'conv1', 3, 64, [1 + 2 * r1, 7, 7], strides=[tstrides[0][0], 2, 2],
                        pads=[r1, 3, 3] * 2, weight_init=('MSRAFill', {}),
                        bias_init=('ConstantFill', {'value': 0.}), no_bias=1)
    if cfg.MODEL.USE_AFFINE:
        stem = model.AffineNd(stem, 'res_conv1_bn', 64)
    else:
        stem = model.SpatialBN(stem, 'res_conv1_bn', 64, epsilon=cfg.MODEL.BN_EPSILON,
                               momentum=cfg.MODEL.BN_MOMENTUM, is_test=test_mode)
    stem = model.Relu(stem, stem)
    blob = model.MaxPool(stem, 'pool1', kernels=[1, 3, 3], strides=[1, 2, 2], pads=[0, 1, 1] * 2)

    if cfg.MODEL.DEPTH not in (50, 101):
        raise Exception("Unsupported network settings.")

    # ---- res2 ------------------------------------------------------------------------------------
    blob, dim = resnet_helper.res_stage_nonlocal(
        model, res_block, blob, 64, 256, stride=1, num_blocks=n1, prefix='res2',
        dim_inner=dim_inner, group=group, use_temp_convs=radius[1], temp_strides=tstrides[1])
    blob = model.MaxPool(blob, 'pool2', kernels=[2, 1, 1], strides=[2, 1, 1], pads=[0, 0, 0] * 2)

    # ---- res3: grouped non-local when BN is frozen (every shipped config) ------------------------
    res3_common = dict(stride=2, num_blocks=n2, prefix='res3', dim_inner=dim_inner * 2, group=group,
                       use_temp_convs=radius[2], temp_strides=tstrides[2], batch_size=batch_size,
                       nonlocal_name='nonlocal_conv3', nonlocal_mod=_nonlocal_period(3))
    if cfg.MODEL.USE_AFFINE:
        blob, dim = resnet_helper.res_stage_nonlocal_group(
            model, res_block, blob, dim, 512, pool_stride=pool_stride,
            spatial_dim=int(crop_size / 8), group_size=4, **res3_common)
    else:
        blob, dim = resnet_helper.res_stage_nonlocal(model, res_block, blob, dim, 512, **res3_common)

    # ---- res4 / res5 -----------------------------------------------------------------------------
    blob, dim = resnet_helper.res_stage_nonlocal(
        model, res_block, blob, dim, 1024, stride=2, num_blocks=n3, prefix='res4',
        dim_inner=dim_inner * 4, group=group, use_temp_convs=radius[3], temp_strides=tstrides[3],
        batch_size=batch_size, nonlocal_name='nonlocal_conv4', nonlocal_mod=_nonlocal_period(4))
    if cfg.MODEL.DILATIONS_AFTER_CONV5:
        cfg.DILATIONS = 2
    blob, dim = resnet_helper.res_stage_nonlocal(
        model, res_block, blob, dim, 2048, stride=1, num_blocks=n4, prefix='res5',
        dim_inner=dim_inner * 8, group=group, use_temp_convs=radius[4], temp_strides=tstrides[4])

    if cfg.MODEL.FREEZE_BACKBONE:
        model.StopGradient(blob, blob)

    # ---- head ------------------------------------------------------------------------------------
    if cfg.DATASET == 'ava':
        head_func = head_helper.add_roi_head
    elif cfg.DATASET in ['charades', 'epic']:
        head_func = head_helper.add_basic_head
    else:
        raise NotImplementedError('Unknown dataset {}'.format(cfg.DATASET))
    blob, dim = head_func(model, blob, dim, pool_stride, out_spatial_dim, suffix, lfb_infer_only, test_mode)
    if lfb_infer_only:
        return model, None, None

    if cfg.TRAIN.DROPOUT_RATE > 0 and not test_mode:
        blob = model.Dropout(blob, blob + '_dropout', ratio=cfg.TRAIN.DROPOUT_RATE, is_test=False)
    blob = model.FC(blob, 'pred', dim, cfg.MODEL.NUM_CLASSES,
                    weight_init=('GaussianFill', {'std': cfg.MODEL.FC_INIT_STD}),
                    bias_init=('ConstantFill', {'value': 0.}))

    scale = 1. / cfg.NUM_GPUS
    loss = None
    if split == 'train':
        if cfg.MODEL.MULTI_LABEL:
            prob = model.Sigmoid(blob, 'prob')
            loss = model.SigmoidCrossEntropyLoss([blob, labels], ['loss'], scale=scale)
        else:
            prob, loss = model.SoftmaxWithLoss([blob, labels], ['prob', 'loss'], scale=scale)
    elif cfg.MODEL.MULTI_LABEL:
        prob = model.Sigmoid(blob, 'prob', engine='CUDNN')
    else:
        prob = model.Softmax(blob, 'prob')
    return model, prob, loss
