"""Output heads: clip-level (Charades / EPIC) and box-level (AVA) pooling, optionally joined
with a feature-bank operator (call-compatible with lib/models/head_helper.py:32-123)."""
import logging

from core.config import config as cfg
import models.lfb_helper as lfb_helper

logger = logging.getLogger(__name__)


def _join_with_fbo(model, feat, dim_in, num_lfb_feat, suffix, lfb_infer_only, test_mode):
    heads, dims = [feat], [dim_in]
    if cfg.LFB.ENABLED and not lfb_infer_only:
        fbo_out, fbo_dim = lfb_helper.add_fbo_head(model, feat, dim_in, num_lfb_feat=num_lfb_feat,
                                                   test_mode=test_mode, suffix=suffix)
        heads.append(fbo_out)
        dims.append(fbo_dim)
    joined = model.net.Concat(heads, ['pool5', 'pool5_concat_info'], axis=1)[0]
    return joined, sum(dims)


def add_basic_head(model, blob_in, dim_in, pool_stride, out_spatial_dim, suffix, lfb_infer_only,
                   test_mode):
    """global space-time average -> (B, 2048, 1, 1, 1) [+ FBO]"""
    pooled = model.AveragePool(blob_in, blob_in + '_pooled',
                               kernels=[pool_stride, out_spatial_dim, out_spatial_dim],
                               strides=[1, 1, 1], pads=[0, 0, 0] * 2)
    return _join_with_fbo(model, pooled, dim_in, cfg.LFB.WINDOW_SIZE, suffix, lfb_infer_only, test_mode)


def add_roi_head(model, blob_in, dim_in, pool_stride, out_spatial_dim, suffix, lfb_infer_only,
                 test_mode):
    """per-box features (N_boxes, 2048, 1, 1, 1) [+ FBO over WINDOW_SIZE x boxes-per-step bank rows]"""
    roi_feat = roi_pool(model, blob_in, dim_in, out_spatial_dim, suffix)
    n_bank = cfg.LFB.WINDOW_SIZE * cfg.AVA.LFB_MAX_NUM_FEAT_PER_STEP
    return _join_with_fbo(model, roi_feat, dim_in, n_bank, suffix, lfb_infer_only, test_mode)


def roi_pool(model, blob_in, dim_in, out_spatial_dim, suffix):
    """temporal average -> RoIAlign -> spatial max -> `box_pooled`"""
    pooled = model.AveragePool(blob_in, 'blob_pooled', kernels=[cfg.TRAIN.VIDEO_LENGTH // 2, 1, 1],
                               strides=[1, 1, 1], pads=[0, 0, 0] * 2)
    pooled = model.Squeeze(pooled, pooled + '_4d', dims=[2])
    resolution = cfg.ROI.XFORM_RESOLUTION
    roi_feat = lfb_helper.RoIFeatureTransform(
        model, pooled, 'roi_feat_3d', spatial_scale=(1.0 / cfg.ROI.SCALE_FACTOR),
        resolution=resolution, blob_rois='proposals{}'.format(suffix))
    if resolution > 1:
        roi_feat = model.MaxPool(roi_feat, 'roi_feat_1d', kernels=[resolution, resolution],
                                 strides=[1, 1], pads=[0, 0] * 2)
    roi_feat, _ = model.Reshape(roi_feat, ['box_pooled', 'roi_feat_re2_shape'],
                                shape=(-1, dim_in, 1, 1, 1))
    return roi_feat
