"""RoIAlign (legacy Caffe2 operator, `aligned` = false, sampling_ratio = 0) restated twice.

Follows the call site lib/models/lfb_helper.py:130-152 (pooled 7x7, spatial_scale 1/16,
sampling_ratio 0) -> Caffe2 `RoIAlign` (caffe2/operators/roi_align_op.{cc,cu}, not in the
reference tree; semantics restated in SURVEY.md Appendix B).  All coordinate arithmetic is done
in float32 in the operator's own operation order, because the integer decisions (sampling grid
size, bilinear corner indices, the out-of-range predicate) must match the HIP kernel bit for bit.

`roi_align_loop` is the literal scalar restatement; `roi_align_vec` is an independent vectorised
one.  tests/test_oracle.py::test_two_roialign_implementations_agree_bit_exactly requires them to
agree exactly on every integer and to 1e-6 on values (SURVEY.md 8c iii).
"""
import numpy as np

f32 = np.float32


def _geom(roi, spatial_scale, pooled):
    s = f32(spatial_scale)
    batch = int(roi[0])
    start_w = f32(roi[1]) * s
    start_h = f32(roi[2]) * s
    end_w = f32(roi[3]) * s
    end_h = f32(roi[4]) * s
    roi_w = max(f32(end_w - start_w), f32(1.0))
    roi_h = max(f32(end_h - start_h), f32(1.0))
    bin_h = f32(roi_h / f32(pooled))
    bin_w = f32(roi_w / f32(pooled))
    grid_h = int(np.ceil(f32(roi_h / f32(pooled))))
    grid_w = int(np.ceil(f32(roi_w / f32(pooled))))
    return batch, start_w, start_h, bin_h, bin_w, grid_h, grid_w


def _bilinear(y, x, H, W):
    """returns (inside, y_low, x_low, y_high, x_high, w1..w4) with float32 arithmetic"""
    if y < f32(-1.0) or y > f32(H) or x < f32(-1.0) or x > f32(W):
        return False, -1, -1, -1, -1, f32(0), f32(0), f32(0), f32(0)
    if y <= 0:
        y = f32(0)
    if x <= 0:
        x = f32(0)
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = f32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = f32(x_low)
    else:
        x_high = x_low + 1
    ly, lx = f32(y - f32(y_low)), f32(x - f32(x_low))
    hy, hx = f32(f32(1.0) - ly), f32(f32(1.0) - lx)
    return True, y_low, x_low, y_high, x_high, f32(hy * hx), f32(hy * lx), f32(ly * hx), f32(ly * lx)


def roi_align_loop(feat, rois, pooled=7, spatial_scale=1.0 / 16):
    """feat: (N,C,H,W) float array; rois: (R,5) float32.
    Returns out (R,C,pooled,pooled) in feat's dtype and dbg (R,pooled,pooled,8) int32 holding
    {batch, grid_h, grid_w, y_low, x_low, y_high, x_high, inside} of the first sample per bin."""
    N, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, pooled, pooled), dtype=feat.dtype)
    dbg = np.zeros((R, pooled, pooled, 8), dtype=np.int32)
    for r in range(R):
        batch, start_w, start_h, bin_h, bin_w, grid_h, grid_w = _geom(rois[r], spatial_scale, pooled)
        count = grid_h * grid_w
        for ph in range(pooled):
            for pw in range(pooled):
                acc = np.zeros(C, dtype=feat.dtype)
                for iy in range(grid_h):
                    y = f32(f32(start_h + f32(f32(ph) * bin_h)) + f32(f32(f32(f32(iy) + f32(0.5)) * bin_h) / f32(grid_h)))
                    for ix in range(grid_w):
                        x = f32(f32(start_w + f32(f32(pw) * bin_w)) + f32(f32(f32(f32(ix) + f32(0.5)) * bin_w) / f32(grid_w)))
                        inside, yl, xl, yh, xh, w1, w2, w3, w4 = _bilinear(y, x, H, W)
                        if iy == 0 and ix == 0:
                            dbg[r, ph, pw] = [batch, grid_h, grid_w, yl, xl, yh, xh, int(inside)]
                        if not inside:
                            continue
                        f = feat[batch]
                        acc += (w1 * f[:, yl, xl] + w2 * f[:, yl, xh] + w3 * f[:, yh, xl] + w4 * f[:, yh, xh])
                out[r, :, ph, pw] = acc / count
    return out, dbg


def roi_decisions(rois, H, W, pooled=7, spatial_scale=1.0 / 16, max_grid=4):
    """integer decisions of EVERY bilinear sample: (R, pooled, pooled, max_grid, max_grid, 8) int32 =
    {batch, grid_h, grid_w, y_low, x_low, y_high, x_high, inside}; samples beyond the RoI's adaptive grid carry
    {.., -2, -2, -2, -2, -1} (the layout of vlfb_roi_align_decisions)"""
    R = rois.shape[0]
    dbg = np.zeros((R, pooled, pooled, max_grid, max_grid, 8), dtype=np.int32)
    for r in range(R):
        batch, start_w, start_h, bin_h, bin_w, grid_h, grid_w = _geom(rois[r], spatial_scale, pooled)
        assert grid_h <= max_grid and grid_w <= max_grid
        for ph in range(pooled):
            for pw in range(pooled):
                for iy in range(max_grid):
                    for ix in range(max_grid):
                        if iy >= grid_h or ix >= grid_w:
                            dbg[r, ph, pw, iy, ix] = [batch, grid_h, grid_w, -2, -2, -2, -2, -1]
                            continue
                        y = f32(f32(start_h + f32(f32(ph) * bin_h)) + f32(f32(f32(f32(iy) + f32(0.5)) * bin_h) / f32(grid_h)))
                        x = f32(f32(start_w + f32(f32(pw) * bin_w)) + f32(f32(f32(f32(ix) + f32(0.5)) * bin_w) / f32(grid_w)))
                        inside, yl, xl, yh, xh = _bilinear(y, x, H, W)[:5]
                        dbg[r, ph, pw, iy, ix] = [batch, grid_h, grid_w, yl, xl, yh, xh, int(inside)]
    return dbg


def roi_align_vec(feat, rois, pooled=7, spatial_scale=1.0 / 16):
    """Independent vectorised implementation (per RoI: all bins x samples at once)."""
    N, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, pooled, pooled), dtype=feat.dtype)
    dbg = np.zeros((R, pooled, pooled, 8), dtype=np.int32)
    s = f32(spatial_scale)
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = (f32(v) * s for v in rois[r, 1:5])
        rw = np.maximum(f32(x2 - x1), f32(1))
        rh = np.maximum(f32(y2 - y1), f32(1))
        bh, bw = f32(rh / f32(pooled)), f32(rw / f32(pooled))
        gh, gw = int(np.ceil(rh / f32(pooled))), int(np.ceil(rw / f32(pooled)))
        p = np.arange(pooled, dtype=np.float32)
        iy = np.arange(gh, dtype=np.float32)
        ix = np.arange(gw, dtype=np.float32)
        ys = ((y1 + p * bh)[:, None] + (((iy + f32(0.5)) * bh) / f32(gh))[None, :]).astype(np.float32)  # (P,gh)
        xs = ((x1 + p * bw)[:, None] + (((ix + f32(0.5)) * bw) / f32(gw))[None, :]).astype(np.float32)  # (P,gw)
        Y = np.broadcast_to(ys[:, None, :, None], (pooled, pooled, gh, gw)).copy()
        X = np.broadcast_to(xs[None, :, None, :], (pooled, pooled, gh, gw)).copy()
        inside = ~((Y < -1) | (Y > H) | (X < -1) | (X > W))
        Y = np.maximum(Y, f32(0))
        X = np.maximum(X, f32(0))
        yl = Y.astype(np.int32)
        xl = X.astype(np.int32)
        ycl = yl >= H - 1
        xcl = xl >= W - 1
        yl = np.where(ycl, H - 1, yl)
        xl = np.where(xcl, W - 1, xl)
        yh = np.where(ycl, H - 1, yl + 1)
        xh = np.where(xcl, W - 1, xl + 1)
        Y = np.where(ycl, yl.astype(np.float32), Y)
        X = np.where(xcl, xl.astype(np.float32), X)
        ly = (Y - yl.astype(np.float32)).astype(np.float32)
        lx = (X - xl.astype(np.float32)).astype(np.float32)
        hy, hx = (f32(1) - ly).astype(np.float32), (f32(1) - lx).astype(np.float32)
        w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
        f = feat[b]  # (C,H,W)
        val = (w1 * f[:, yl, xl] + w2 * f[:, yl, xh] + w3 * f[:, yh, xl] + w4 * f[:, yh, xh])  # (C,P,P,gh,gw)
        val = np.where(inside, val, 0)
        out[r] = val.sum(axis=(3, 4)) / (gh * gw)
        first_in = inside[:, :, 0, 0]
        dbg[r, :, :, 0] = b
        dbg[r, :, :, 1] = gh
        dbg[r, :, :, 2] = gw
        dbg[r, :, :, 3] = np.where(first_in, yl[:, :, 0, 0], -1)
        dbg[r, :, :, 4] = np.where(first_in, xl[:, :, 0, 0], -1)
        dbg[r, :, :, 5] = np.where(first_in, yh[:, :, 0, 0], -1)
        dbg[r, :, :, 6] = np.where(first_in, xh[:, :, 0, 0], -1)
        dbg[r, :, :, 7] = first_in
    return out, dbg


def roi_align_torch(feat, rois, pooled=7, spatial_scale=1.0 / 16):
    """Differentiable (w.r.t. feat) torch version used by the model oracle: the sampling
    indices / weights come from the same float32 arithmetic as above (numpy), the gather and
    the weighted sum are torch ops so autograd yields the RoIAlignGradient scatter."""
    import torch
    N, C, H, W = feat.shape
    outs = []
    s = f32(spatial_scale)
    rois = np.asarray(rois, dtype=np.float32)
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = (f32(v) * s for v in rois[r, 1:5])
        rw = np.maximum(f32(x2 - x1), f32(1))
        rh = np.maximum(f32(y2 - y1), f32(1))
        bh, bw = f32(rh / f32(pooled)), f32(rw / f32(pooled))
        gh, gw = int(np.ceil(rh / f32(pooled))), int(np.ceil(rw / f32(pooled)))
        acc = 0
        for iy in range(gh):
            for ix in range(gw):
                ws = np.zeros((4, pooled, pooled), dtype=np.float32)
                idx = np.zeros((4, pooled, pooled), dtype=np.int64)
                for ph in range(pooled):
                    for pw in range(pooled):
                        y = f32(f32(y1 + f32(f32(ph) * bh)) + f32(f32(f32(f32(iy) + f32(0.5)) * bh) / f32(gh)))
                        x = f32(f32(x1 + f32(f32(pw) * bw)) + f32(f32(f32(f32(ix) + f32(0.5)) * bw) / f32(gw)))
                        inside, yl, xl, yh, xh, w1, w2, w3, w4 = _bilinear(y, x, H, W)
                        if inside:
                            ws[:, ph, pw] = [w1, w2, w3, w4]
                            idx[:, ph, pw] = [yl * W + xl, yl * W + xh, yh * W + xl, yh * W + xh]
                fb = feat[b].reshape(C, H * W)
                for k in range(4):
                    g = fb[:, torch.from_numpy(idx[k].reshape(-1))].reshape(C, pooled, pooled)
                    acc = acc + g * torch.from_numpy(ws[k]).to(feat.dtype)
        outs.append(acc / (gh * gw))
    return torch.stack(outs, 0)
