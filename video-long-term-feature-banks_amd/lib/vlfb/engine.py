"""Execution engine: lowers the op list recorded by `ModelBuilder` into fused HIP kernel steps
and runs forward / backward / gradient all-reduce / solver on one MI355X per process.

Design (MI355X-first):
  * activations are channels-last matrices [positions][C] in bf16 (throughput) or fp32 (parity);
    every Reshape / Transpose / Squeeze of the reference graph that only regroups positions is a
    zero-cost view here (the grouped non-local block's transpose-reshape-transpose included);
  * Conv -> AffineNd -> ReLU -> Sum -> ReLU chains become ONE implicit-GEMM launch: the frozen
    affine scale is folded into the MFMA weight operand, the affine bias, the residual add and the
    ReLU into its epilogue;  the backward applies ReLU masks and gradient accumulation in the
    epilogue of whichever kernel contributes LAST to a tensor's gradient (decided at plan time);
  * parameters live in flat fp32 buckets ordered by backward completion, so gradient all-reduce
    (RCCL) can start per bucket on a side stream while backward continues, and the solver is one
    kernel per bucket;
  * shapes are static: every buffer is allocated once at plan time, nothing is allocated in the
    step loop.
There is no PyTorch/CPU fallback anywhere below: torch supplies memory, streams and the process
group only.
"""
import ctypes as C
import logging
import math
import os
import struct
from collections import OrderedDict

import numpy as np
import torch

from core.config import config as cfg
from vlfb import hip
from vlfb import dist
from vlfb.net import ssa_form
from vlfb.rng import dropout_seed

logger = logging.getLogger(__name__)


def _prod(xs):
    out = 1
    for x in xs:
        out *= int(x)
    return out


def _at(t, elems):
    """device address of element `elems` of a flat tensor (None stays None): the channel slice / weight block of a group"""
    if t is None:
        return None
    if isinstance(t, int):
        raise TypeError("_at needs the tensor (element size)")
    return t.data_ptr() + int(elems) * t.element_size()


# ================================================================================================
# values
# ================================================================================================
class Blob(object):
    """One SSA value.  Storage order = logical dims with the channel axis `caxis` moved last."""

    def __init__(self, name, shape, caxis, kind="act", root=None):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.caxis = caxis
        self.kind = kind              # 'act' (engine dtype) | 'f32' | 'i32'
        self.root = root or self
        self.tensor = None            # flat storage (root only)
        self.relu = False             # values are post-ReLU: the finished gradient gets masked
        self.needs_grad = False
        self.detached = False
        self.slot = None              # GradSlot (root only)
        self.producer = None
        self.grad_scale = 1.0         # the stored gradient is the true one times this power of two (fp16 range)
        self.planes = None            # "split" dtype: bf16 term planes [2][numel] of the values, written by the producing conv
        self.half = None              # "mix" dtype: fp16 copy of the values for the fp16 backward (root only; Engine.want_half)
        self.need_half = False
        self.grad_f32 = False         # "mix" dtype: the gradient of this blob is kept in fp32 (theta / phi / g of a non-local block)
        # "mix" dtype: the values are stored as TWO fp16 planes [2][numel] (hi = fp16(v), lo = fp16(v - hi): ~22 bits; vlfb.h
        # VLFB_F16PAIR) instead of fp32 -- what the two-plane forward convs (hip.MATH_F16X3) read without converting anything,
        # and whose hi plane IS the fp16 copy the backward reads (Blob.half aliases it).  Root only; Engine._plan_pairs.
        self.pair = False

    @property
    def numel(self):
        return _prod(self.shape)

    @property
    def C(self):
        return self.shape[self.caxis]

    @property
    def rows(self):
        return self.numel // self.C

    def storage(self):
        return self.root.tensor

    def bstorage(self):
        """the values as the BACKWARD pass reads them: the fp16 copy on the "mix" path, else the values themselves"""
        r = self.root
        return r.half if r.half is not None else r.tensor

    def bptr(self):
        return self.bstorage().data_ptr()

    def ptr(self):
        return self.root.tensor.data_ptr()

    def lo(self):
        """the low plane of a two-plane blob (a view of the second half of its storage)"""
        t = self.root.tensor
        return t[t.numel() // 2:]

    def view(self, name, shape, caxis):
        v = Blob(name, shape, caxis, self.kind, self.root)
        v.needs_grad = self.needs_grad
        v.detached = self.detached
        return v


class GradSlot(object):
    """Accumulates the gradient of one root blob.  Contributions arrive in backward order; the
    last one also applies the ReLU mask of the blob (its forward values) when the blob is
    post-ReLU, so no separate masking pass ever runs."""

    def __init__(self, engine, blob):
        self.engine = engine
        self.blob = blob
        self.expected = 0
        self.buf = None
        self.count = 0
        self.cur = None
        self.planes = None            # "split" dtype: bf16 term planes [2][numel] of the FINISHED gradient ...
        self.planes_valid = False     # ... valid when the last contribution came from a conv epilogue that wrote them
        # "mix" dtype, residual-stream gradients (a slot that receives the gradient of the next block's output as an alias and
        # adds a branch to it, block after block): kept as TWO fp16 terms -- cur + cur_lo -- so that the running sum is not
        # re-rounded to 11 bits sixteen times on its way down the stack.  The conv epilogues read / write the low term
        # (vlfb_conv_args R_lo / O_lo); a contributor that does not know about it passes it through unchanged.
        self.two_term = False
        self.buf_lo = None
        self.cur_lo = None
        self.add_lo = None            # handed to the contributor being called: low term of `add` ...
        self.out_lo = None            # ... and where it may leave the low term of what it writes
        # "mix" dtype, an fp32 slot with ONE contributor whose consumer is a 16-bit launch (the attention output / theta of a
        # non-local block): the contributing NT launch leaves the gradient rounded to fp16 here as well (vlfb_conv_args.O_lo
        # next to an fp32 output), so the consumer does not run a cast pass first (Engine.GRAD_HALF_COPY)
        self.half_buf = None
        self.half_valid = False

    def reset(self):
        self.count = 0
        self.cur = None
        self.cur_lo = None
        self.planes_valid = False
        self.half_valid = False

    def value_half(self):
        """the finished fp32 gradient rounded to the 16-bit backward type, if its contributor wrote that copy; else None"""
        return self.half_buf if (self.half_valid and self.count == self.expected and self.cur is self.buf) else None

    def half_dest(self, out, add, mask):
        """where the contributor being called may leave the 16-bit copy of what it writes (None: not this time)"""
        return self.half_buf if (out is self.buf and add is None and mask is None and self.expected == 1) else None

    def value_lo(self):
        """low term of the finished gradient, or None"""
        return self.cur_lo if self.count == self.expected else None

    def value_planes(self):
        """term planes of the finished gradient, or None (the consumer then splits the fp32 values itself)"""
        return self.planes if (self.planes_valid and self.cur is self.buf) else None

    def _flags(self):
        self.count += 1
        if self.count > self.expected:
            raise RuntimeError("too many gradient contributions for %s" % self.blob.name)
        last = self.count == self.expected
        # (the mask operand has the dtype of the gradient: the fp16 copy of the values on the 16-bit backward of "mix")
        mask = (self.blob.storage() if self.blob.grad_f32 else self.blob.bstorage()) if (last and self.blob.relu) else None
        return self.cur, mask

    def contribute(self, fn, supports_add=True, supports_mask=True, writes_planes=False):
        """fn(out, add, mask) launches a kernel writing out = value (+ add), masked by mask > 0.
        writes_planes: fn takes a fourth argument, the destination of the output's bf16 term planes (or None); it is
        handed this slot's planes when the contribution is the LAST one, i.e. when `out` is the finished gradient."""
        add, mask = self._flags()
        last = self.count == self.expected
        if writes_planes:
            want = self.planes if last else None
            # (a conv epilogue: it adds the low term of `add` and leaves the low term of its result, GradSlot.two_term)
            self.add_lo = self.cur_lo if add is not None else None
            self.out_lo = self.buf_lo if self.two_term else None
            fn(self.buf, add, mask, want)
            self.planes_valid = want is not None
            self.cur_lo = self.out_lo
        elif (add is None or supports_add) and (mask is None or supports_mask):
            fn(self.buf, add, mask)
            if mask is not None or add is None:
                self.cur_lo = None        # (a masked result would need a masked low term; a fresh value has none)
        else:
            tmp = self.engine.scratch_act(self.buf.numel(), self.buf.dtype)
            fn(tmp, None, None)
            hip.call("vlfb_add", hip.ptr(tmp), hip.ptr(add), hip.ptr(self.buf), hip.ptr(mask),
                     hip.dtype_code(self.buf.dtype), self.buf.numel(), 0)
            if mask is not None or add is None:
                self.cur_lo = None
        self.cur = self.buf

    def contribute_alias(self, t, t_lo=None):
        """the contribution is an existing tensor (identity branch of a Sum / ReLU); t_lo: its low term (two_term slots)"""
        add, mask = self._flags()
        if add is None and mask is None:
            self.cur = t
            self.cur_lo = t_lo if self.two_term else None
            return
        hip.call("vlfb_add", hip.ptr(t), hip.ptr(add), hip.ptr(self.buf), hip.ptr(mask),
                 hip.dtype_code(self.buf.dtype), self.buf.numel(), 0)
        self.cur = self.buf
        if mask is not None:
            self.cur_lo = None

    def value(self):
        if self.count != self.expected:
            raise RuntimeError("gradient of %s read after %d of %d contributions"
                               % (self.blob.name, self.count, self.expected))
        return self.cur


# ================================================================================================
# steps
# ================================================================================================
class Step(object):
    def __init__(self, eng):
        self.eng = eng
        self.inputs = []     # blobs whose gradient this step may produce
        self.outputs = []
        self.params = []     # trainable parameter names whose gradient this step produces

    def fwd(self):
        raise NotImplementedError

    def bwd(self):
        raise NotImplementedError

    def grad_inputs(self):
        """inputs that receive a gradient contribution from this step"""
        return [b for b in self.inputs if b.needs_grad and not b.detached]

    def out_grad(self, i=0):
        return self.outputs[i].root.slot.value()

    # "mix" dtype (Engine._plan_head_f32): a step of the head may find its output gradient and / or the slot it contributes to in
    # fp32 (Blob.grad_f32).  It computes in the dtype of the slot it WRITES; an incoming gradient of the other dtype goes
    # through a private buffer (fp32 -> fp16 is the one rounding where the gradient re-enters the fp16 backward).
    def gcode(self, blob):
        return hip.F32 if blob.root.grad_f32 else self.eng.bcode

    def g_as(self, g, blob_from, blob_to, key="_gcast"):
        """`g` (the finished gradient of blob_from) in the gradient dtype of blob_to"""
        if bool(blob_from.root.grad_f32) == bool(blob_to.root.grad_f32):
            return g
        buf = getattr(self, key, None)
        if buf is None or buf.numel() < g.numel():
            buf = torch.empty(g.numel(), device=self.eng.device, dtype=torch.float32 if blob_to.root.grad_f32 else self.eng.btdtype)
            setattr(self, key, buf)
        hip.call("vlfb_cast", hip.ptr(g), self.gcode(blob_from), hip.ptr(buf), self.gcode(blob_to), g.numel())
        return buf[:g.numel()]


class ConvStep(Step):
    """ConvNd [+ AffineNd] [+ residual Sum] [+ ReLU] as one implicit-GEMM launch."""

    def __init__(self, eng, x, out, wname, cbname, kernels, strides, pads, dils, group=1):
        """group > 1 (RESNETS.NUM_GROUPS, resnet_helper.py:56-63; no shipped config): output channel block g reads input
        channel block g -- G launches of the ordinary kernels on channel SLICES of the tensors (pointer offset + the
        descriptor's leading dimensions), one weight-operand block per group.  Correct, not tuned."""
        Step.__init__(self, eng)
        self.group = int(group)
        self.x, self.out = x, out
        self.inputs, self.outputs = [x], [out]
        self.wname, self.cbname = wname, cbname
        self.sname = self.bname = None
        self.relu = False
        self.residual = None
        self.k, self.s, self.p, self.d = kernels, strides, pads, dils
        self.stem = (x.C == 3)
        self.eff_bias = None

    def name(self):
        return "conv:" + self.out.name

    # -- geometry ------------------------------------------------------------------------------
    def _geom(self):
        k, s, p, d = self.k, self.s, self.p, self.d
        return dict(kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2],
                    dt=d[0], dh=d[1], dw=d[2])

    def setup(self):
        eng = self.eng
        N, Cin, T, H, W = self.x.shape
        _, Cout, To, Ho, Wo = self.out.shape
        G = self.group
        assert not (self.stem and G > 1)
        self.Cin_k = 4 if self.stem else Cin // G     # channels as the kernel sees them (one group's)
        self.Cog = Cout // G                          # output channels of one group
        # grouped: a launch works on a channel slice -- the tensors keep their row strides
        ld_f = dict(lda=Cin, ldo=Cout, ldr=Cout) if G > 1 else {}
        ld_d = dict(lda=Cout, ldo=Cin, ldr=Cin) if G > 1 else {}
        ld_w = dict(lda=Cin, ldp=Cout) if G > 1 else {}
        self._ld_d = ld_d
        self.pack = 8 if self.stem else 0
        code, bcode = eng.code, eng.bcode
        geom = self._geom()
        # the gradient arriving at `out` may be stored scaled (fp16 attention logits): divide it out in the
        # epilogues of the kernels that consume it
        self.gscale = float(self.out.root.grad_scale)
        if self.stem:
            # the data blob is stored with 4 zero pixels on both sides of every W row
            wpad = getattr(self.x.root, "pad_w", 0) or getattr(self.x, "pad_w", 0)
            assert wpad >= self.p[2] and wpad >= self.pack - self.k[2] + self.p[2], "stem needs a W-padded input"
            W = W + 2 * wpad
            geom["pw"] = self.p[2] - wpad
        # split-bf16 math on fp32 storage (Engine dtype "split"): the weight operand copies are bf16 term planes
        wshape = eng.kernel_shape(self.wname)
        mf, mb = eng.math_fwd, eng.math_bwd
        self.wblk = _prod(wshape) // G                # elements of one group's weight block
        planes = dict(b_pstride=self.wblk) if eng.split else {}
        bplanes = planes if mb != hip.MATH_NATIVE else {}          # ("mix": split forward, native fp16 backward)
        # "mix", two-plane forward (Engine._plan_pairs): x_pair -- the input is two fp16 planes (the stem makes the clip's per
        # pass): both operands pre-split, three fp16 MFMAs per product (hip.MATH_F16X3), fp32 or two-plane output; o_pair
        # without x_pair -- an fp32 input (the attention output of a non-local block) through the split-bf16 kernel that
        # writes two planes.  The residual, if any, has the format of the output.
        self.o_pair = bool(self.out.root.pair)
        self.x_pair = bool(self.x.root.pair or (self.stem and eng.pair_fwd and self.o_pair))
        assert not (self.o_pair or self.x_pair) or (G == 1 and eng.mix)
        assert self.residual is None or bool(self.residual.root.pair) == self.o_pair, "residual / output formats differ: %s" % self.out.name
        fkw = dict(N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T, Hs=H, Ws=W, Cs=self.Cin_k, Cn=self.Cog, pack_w=self.pack, relu=int(self.relu),
                   bias_mode=hip.BIAS_COL if self.has_bias() else hip.BIAS_NONE)
        if self.x_pair:
            n_in = (self.x.root.numel // self.x.root.C // self.x.root.shape[-1] * W * self.Cin_k) if self.stem else self.x.root.numel
            self.d_f = hip.conv_desc(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16 if self.o_pair else hip.F32, math=hip.MATH_F16X3,
                                     a_pstride=n_in, b_pstride=_prod(eng.kernel_shape(self.wname)), alpha=1.0 / hip.MIX_W2_SCALE,
                                     **fkw, **geom)
        else:
            self.d_f = hip.conv_desc(mode=hip.FPROP, dtype=code, out_dtype=hip.F16 if self.o_pair else code, math=mf,
                                     **fkw, **planes, **geom, **ld_f)
        self.d_d = None
        self.d_d_full = None
        self.w2 = False
        self.w2i = False                # (two-term DGRAD weights interleaved per k-tile: hip.MATH_F16W2 / MIX_W2I)
        self.bwd_split = False
        self.dx_f32 = False
        self.bwd_f32 = bool(eng.mix and self.out.root.grad_f32)
        if self.bwd_f32:
            eng.need_scratch_act(self.out.numel)
        if eng.mix and self.x.root.grad_f32:
            eng.need_scratch_f32(self.x.numel)           # GradSlot's add path for an fp32 slot
        if self.x.needs_grad and not self.x.detached:
            assert not self.stem
            dg = dict(geom)
            alpha = 1.0 / self.gscale
            rows = dict(N=N, Tr=T, Hr=H, Wr=W, Ts=To, Hs=Ho, Ws=Wo)
            # "mix", a conv BETWEEN fp32 gradient slots (the FBO head, Engine._plan_head_f32): its whole backward runs as
            # on the `split` dtype -- fp32 gradient in and out, three bf16 products per product, bf16 term planes of W
            self.bwd_split = bool(eng.mix and self.out.root.grad_f32 and self.x.root.grad_f32 and G == 1)
            w2 = self._w2_geometry(dg, rows) if (eng.mix and eng.MIX_W2 and not self.bwd_split) else None
            math_d = mb
            if w2 is not None:
                alpha /= hip.MIX_W2_SCALE
                self.w2 = True
                if self._w2_interleaved(w2, dg, rows):
                    # the same two-term product with ONE gradient tile per (Wh, Wl) pair of weight tiles (hip.MATH_F16W2): the
                    # conv's own DGRAD geometry, weight rows [tap][Cout / 64][term][64]
                    self.w2i = True
                    math_d = hip.MATH_F16W2
                else:
                    dg, rows = w2
            # ("mix": an input whose gradient slot is fp32 -- box_pooled, the attention output of a non-local block -- gets
            # the fp32 accumulators of the fp16 DGRAD as they are, not their fp16 rounding)
            self.dx_f32 = bool(eng.mix and self.x.root.grad_f32)
            if self.dx_f32 and not self.bwd_split:
                self.x.root.grad_half_src = True        # (this DGRAD can leave the fp16 rounding of its fp32 output: GradSlot.half_buf)
            self.d_d = hip.conv_desc(mode=hip.DGRAD, dtype=bcode, out_dtype=hip.F32 if self.dx_f32 else bcode, Cs=self.Cog,
                                     Cn=Cin // G, alpha=alpha, math=math_d, **rows, **bplanes, **dg, **ld_d)
            if self.bwd_split:
                self.d_d = hip.conv_desc(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32, Cs=self.Cog, Cn=Cin // G, alpha=alpha,
                                         math=hip.MATH_BF16X3, **rows, **planes, **dg, **ld_d)
            # (Engine._plan_sparse_shortcut_dgrads) the strided 1x1x1 shortcut as an in-place accumulate over the rows it
            # touches; d_d_full is the ordinary launch, for a pass in which this DGRAD is not an in-place second contribution
            if getattr(self, "sparse_dgrad", False) and not self.bwd_split and not self.dx_f32:
                sp = hip.ConvDesc.from_buffer_copy(bytes(self.d_d))
                sp.algo = hip.ALGO_CLASS0
                try:
                    hip.conv_workspace_bytes(sp)
                    self.d_d_full, self.d_d = self.d_d, sp
                except hip.VlfbError:
                    self.sparse_dgrad = False
        if self.bwd_f32 and self.d_d is not None and not self.bwd_split:
            self.out.root.grad_half = True      # (the fp16 DGRAD reads the fp32 output gradient rounded: GradSlot.half_buf)
        self.d_w = None
        if eng.is_trainable(self.wname):
            self.d_w = hip.conv_desc(mode=hip.WGRAD, dtype=bcode, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T,
                                     Hs=H, Ws=W, Cs=self.Cin_k, Cn=self.Cog, pack_w=self.pack, alpha=1.0 / self.gscale,
                                     math=mb, **geom, **ld_w)
            # "mix": the gradient arriving at `out` is fp32 (Blob.grad_f32): WGRAD as split-bf16 products on the fp32 operands
            self.bwd_f32 = bool(eng.mix and self.out.root.grad_f32)
            if self.bwd_f32:
                self.d_w = hip.conv_desc(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, N=N, Tr=To, Hr=Ho, Wr=Wo, Ts=T,
                                         Hs=H, Ws=W, Cs=self.Cin_k, Cn=self.Cog, pack_w=self.pack, alpha=1.0 / self.gscale,
                                         math=hip.MATH_BF16X3, **geom, **ld_w)
            # a conv with a trainable bias (the non-local / FBO convs): the WGRAD launch also produces the bias gradient --
            # db = alpha * s * column sums of the output gradient it reads anyway (vlfb_conv_run_wgrad_bias)
            if self.cbname and eng.is_trainable(self.cbname) and eng.FUSE_BIAS_GRAD:
                self.d_w.wgrad_bias = 1
            eng.need_workspace(hip.conv_workspace_bytes(self.d_w))
            if not self.bwd_f32:
                eng.want_half(self.x)
            elif self.x.root.pair:
                # the split-bf16 WGRAD reads fp32 operands: the two planes of the input are joined into a scratch tensor
                # right before it, on the parameter-gradient stream (whose launches run one after the other)
                eng.need_join_scratch(self.x.numel)
        # Pre-split operands ("split" dtype, Engine.PLANES): a conv epilogue can write the bf16 term planes of its output
        # next to the fp32 values (o_planes), and DGRAD / WGRAD launches that find their activation / gradient operands in
        # that form spend no VALU on the expansion (a_planes / p_planes).  Variants are built lazily (_pl_desc).
        self._pl = {}
        self._ran = {}
        # the stem reads the clip: its term planes are made once per forward pass by a split pass (3 planes for the
        # six-product FPROP, of which the WGRAD reads the first two next to a split pass over its output gradient)
        self.x_planes = self.g_planes = None
        if self.stem and self.x_pair:
            self.x_planes = torch.empty(2 * self.d_f.a_pstride, device=eng.device, dtype=torch.float16)
        elif self.stem and eng.split and (eng.PLANES or eng.mix) and eng.STEM_PLANES:
            n_in = self.x.root.numel // self.x.root.C // self.x.root.shape[-1] * W * self.Cin_k     # W is the padded width here
            self.x_npl = 3 if mf == hip.MATH_BF16X6 else 2
            self.x_planes = torch.empty(self.x_npl * n_in, device=eng.device, dtype=torch.bfloat16)
            if eng.is_trainable(self.wname) and not eng.mix:
                self.g_planes = torch.empty(2 * self.out.numel, device=eng.device, dtype=torch.bfloat16)
        unit = tuple(self.s) == (1, 1, 1)
        plain = unit and tuple(self.k) == (1, 1, 1) and tuple(self.p) == (0, 0, 0)
        self.dgrad_takes_planes = bool(eng.split and eng.PLANES and G == 1 and (plain or (unit and Cout % 32 == 0)))
        # FPROP reads its input as planes too when the forward products are the three-term ones (two planes per operand:
        # the LDS image of the plane DGRAD; six-term products would need three planes of both operands, 96 KiB per stage)
        self.fprop_takes_planes = bool(eng.split and eng.PLANES and eng.FPROP_PLANES and mf == hip.MATH_BF16X3 and G == 1 and
                                       not self.stem and (plain or self.Cin_k % 32 == 0))
        # operand copies (split math: 3 bf16 term planes for FPROP, 2 for DGRAD -- include/vlfb.h VLFB_SPLIT)
        if self.x_pair:
            self.w_f = torch.empty((2,) + tuple(wshape), device=eng.device, dtype=torch.float16)     # hi, lo of (w * s) * 2^10
        elif eng.split:
            self.w_f = torch.empty((3,) + tuple(wshape), device=eng.device, dtype=torch.bfloat16)
        else:
            self.w_f = torch.empty(wshape, device=eng.device, dtype=eng.tdtype)
        self.w_d = None
        if self.d_d is not None:
            if eng.mix and self.bwd_split:
                self.w_d = torch.empty(2 * _prod(wshape), device=eng.device, dtype=torch.bfloat16)
            elif eng.mix:
                self.w_d = torch.empty((2 if self.w2 else 1) * _prod(wshape), device=eng.device, dtype=torch.float16)
            else:
                self.w_d = (torch.empty(2 * _prod(wshape), device=eng.device, dtype=torch.bfloat16) if eng.split else
                            torch.empty(_prod(wshape), device=eng.device, dtype=eng.tdtype))
        # one group's weight-operand block: [planes][wblk] elements, group g at g * planes * wblk (vlfb_weight_prep* is run
        # per group, so the term planes of a group lie next to each other)
        self.wf_npl = 2 if self.x_pair else 3 if eng.split else 1
        self.wd_npl = (2 if (self.w2 or self.bwd_split) else 1) if eng.mix else (2 if eng.split else 1)
        # the format vlfb_weight_prep writes this conv's operand copies in ("mix": two-term or plain fp16 DGRAD copy, per conv --
        # ConvStep._w2_geometry; a conv without a DGRAD copy goes with the engine's default)
        self.wcode = eng.wcode if not eng.mix else (hip.SPLIT if self.bwd_split else
                                                    hip.MIX if (self.w_d is not None and not self.w2) else
                                                    hip.MIX_W2I if self.w2i else eng.wcode)
        if self.x_pair:
            assert not self.bwd_split, "a conv of the fp32 head with a two-plane input: %s" % self.out.name
            self.wcode = {hip.MIX: hip.MIXH, hip.MIX_W2: hip.MIXH_W2, hip.MIX_W2I: hip.MIXH_W2I}[self.wcode]
        self.half_by_copy = False       # (True: the fp16 copy of the output comes from a copy pass, Engine._plan_half_copies)
        if self.cbname and self.sname:
            self.eff_bias = torch.empty(Cout, device=eng.device, dtype=torch.float32)
        self.params = [n for n in (self.wname, self.cbname) if n and eng.is_trainable(n)]
        if self.cbname and eng.is_trainable(self.cbname):
            self.cb_tmp = torch.empty(Cout, device=eng.device, dtype=torch.float32)

    def _w2_geometry(self, dg, rows):
        """"mix", two-term fp16 DGRAD weights (hip.MIX_W2): the same convolution with a doubled OUTERMOST tap dimension of
        dilation 0, i.e. every tap is contracted with Wh and with Wl by the plain fp16 DGRAD kernels.  Returns (geometry,
        row dims) of that launch, or None when the doubled-tap form of THIS conv is not one the library plans (a k x 1 x 1
        conv that is strided / padded in H, W, a channel count the gathered kernels refuse, ...): the conv then keeps a
        plain fp16 DGRAD copy (11-bit weights) instead of failing the whole graph."""
        eng = self.eng
        k_, s_, p_, d_ = self.k, self.s, self.p, self.d
        N, _, T, H, W = self.x.shape
        _, _, To, Ho, Wo = self.out.shape
        dg = dict(dg)
        if k_[0] == 1:
            dg.update(kt=2, dt=0)
        elif k_[1] == 1 and k_[2] == 1 and tuple(s_) == (1, 1, 1) and p_[1] == 0 and p_[2] == 0:
            # k x 1 x 1 (the first conv of a bottleneck): T plays the role of H, H x W are one pointwise axis,
            # and the (size-1) T axis carries the term dimension
            dg.update(kt=2, kh=k_[0], kw=1, st=1, sh=1, sw=1, pt=0, ph=p_[0], pw=0, dt=0, dh=d_[0], dw=1)
            rows = dict(N=N, Tr=1, Hr=T, Wr=H * W, Ts=1, Hs=To, Ws=Ho * Wo)
        else:
            return None
        probe = hip.conv_desc(mode=hip.DGRAD, dtype=eng.bcode, out_dtype=eng.bcode, Cs=self.Cog, Cn=self.Cin_k,
                              alpha=1.0, math=hip.MATH_NATIVE, **rows, **dg, **self._ld_d)
        try:
            hip.conv_workspace_bytes(probe)        # (the planner is a pure host function of the descriptor)
        except hip.VlfbError:
            return None
        return dg, rows

    def _w2_interleaved(self, w2, dg, rows):
        """Can the two-term DGRAD of this conv run as hip.MATH_F16W2 (the gradient tile of a tap's 64 channels fetched and read
        once for both weight terms)?  Only where the doubled-tap form runs on the 128-row kernel with the tap cursor: the
        strided convs keep the class walk, and the launches the library gives to the 256-row pipelined kernel keep that
        (measured: scratch/r6/skipa_probe.py, profiles/r06_dgrad_shared_tile_probe.txt)."""
        eng = self.eng
        if not eng.MIX_W2I or self.group != 1 or self.Cog % 64 or self._ld_d:
            return False
        dg2, rows2 = w2
        doubled = hip.conv_desc(mode=hip.DGRAD, dtype=eng.bcode, out_dtype=eng.bcode, Cs=self.Cog, Cn=self.Cin_k,
                                alpha=1.0, math=hip.MATH_NATIVE, **rows2, **dg2)
        plan = hip.conv_plan(doubled)
        if not (plan.startswith("nt ") and " ut" in plan):
            return False
        probe = hip.conv_desc(mode=hip.DGRAD, dtype=eng.bcode, out_dtype=eng.bcode, Cs=self.Cog, Cn=self.Cin_k,
                              alpha=1.0, math=hip.MATH_F16W2, **rows, **dg)
        try:
            hip.conv_workspace_bytes(probe)
        except hip.VlfbError:
            return False
        return True

    def has_bias(self):
        return bool(self.cbname or self.bname)

    def taps(self):
        return self.k[0] * self.k[1] * (8 if self.stem else self.k[2])

    def refresh(self):
        """rebuild the MFMA operand copies from the fp32 masters (after a solver step / feed)"""
        eng = self.eng
        w = eng.param_tensor(self.wname)
        s = eng.param_tensor(self.sname) if self.sname else None
        for g in range(self.group):
            hip.call("vlfb_weight_prep", _at(w, g * self.wblk), _at(s, g * self.Cog), self.wf_ptr(g), self.wd_ptr(g),
                     self.wcode, self.Cog, self.taps(), self.Cin_k)
        self.refresh_bias()

    def wf_ptr(self, g):
        """FPROP weight operand of group g"""
        return _at(self.w_f, g * self.wf_npl * self.wblk)

    def wd_ptr(self, g):
        return _at(self.w_d, g * self.wd_npl * self.wblk)

    def refresh_bias(self):
        eng = self.eng
        if self.eff_bias is not None:   # s*cb + b
            Cout = self.out.shape[1]
            hip.call("vlfb_affine_nd_fwd", hip.ptr(eng.param_tensor(self.cbname)),
                     hip.ptr(eng.param_tensor(self.sname)),
                     hip.ptr(eng.param_tensor(self.bname)), hip.ptr(self.eff_bias), 1, Cout, 1)

    def bias_tensor(self):
        if self.eff_bias is not None:
            return self.eff_bias
        if self.cbname:
            return self.eng.param_tensor(self.cbname)
        if self.bname:
            return self.eng.param_tensor(self.bname)
        return None

    def _pl_desc(self, base, **planes):
        """copy of a conv descriptor with plane fields set (cached per combination)"""
        key = (id(base),) + tuple(sorted(planes.items()))
        d = self._pl.get(key)
        if d is None:
            d = hip.ConvDesc.from_buffer_copy(bytes(base))
            for k, v in planes.items():
                setattr(d, k, v)
            self._pl[key] = d
        self._ran[id(base)] = d           # the variant of `base` that was launched last (Engine.plan_table(launched=True))
        return d

    def fwd(self):
        R = self.residual.storage() if self.residual is not None else None
        op = self.out.root.planes
        if self.x_pair or self.o_pair:
            A = self.x.storage()
            if self.x_planes is not None:         # the stem: the clip's two fp16 planes, made per pass
                hip.call("vlfb_pair_split", self.x.ptr(), hip.ptr(self.x_planes), self.x_planes.numel() // 2)
                A = self.x_planes
            if self.o_pair:
                hip.conv_run(self.d_f, A, self.w_f, None, self.out.storage(), bias=self.bias_tensor(), R=R,
                             R_lo=self.residual.lo() if R is not None else None, O_lo=self.out.lo())
            else:
                # fp32 output (theta / phi / g of a non-local block); O_lo = the fp16 copy the backward reads, if it wants one
                assert R is None
                hip.conv_run(self.d_f, A, self.w_f, None, self.out.storage(), bias=self.bias_tensor(), O_lo=self.out.root.half)
            return
        if self.x_planes is not None:
            n = self.x_planes.numel() // self.x_npl
            hip.call("vlfb_split_planes", self.x.ptr(), hip.ptr(self.x_planes), self.x_npl, 1, n // 8, 8, 0)
            kw = dict(a_planes=self.x_npl, a_pstride=n)
            oh = self.out.root.half
            if oh is not None:
                kw.update(o_planes=1)
            hip.conv_run(self._pl_desc(self.d_f, **kw), self.x_planes, self.w_f, None, self.out.storage(),
                         bias=self.bias_tensor(), R=R, O_planes=oh)
            return
        xp = self.x.root.planes if self.fprop_takes_planes else None
        kw = {}
        if xp is not None:
            kw.update(a_planes=2, a_pstride=xp.numel() // 2)
        if op is not None:
            kw.update(o_planes=2, o_pstride=op.numel() // 2)
        elif self.out.root.half is not None:       # "mix": the fp16 copy the backward reads, written by this epilogue
            op = self.out.root.half
            kw.update(o_planes=1)
        d = self._pl_desc(self.d_f, **kw) if kw else self.d_f
        if self.group == 1:
            hip.conv_run(d, self.x.storage() if xp is None else xp, self.w_f, None,
                         self.out.storage(), bias=self.bias_tensor(), R=R, O_planes=op)
            return
        xs, os_, bt = self.x.storage(), self.out.storage(), self.bias_tensor()
        for g in range(self.group):                # channel slices: block g of the outputs reads block g of the inputs
            hip.conv_run(d, _at(xs, g * self.Cin_k), self.wf_ptr(g), None, _at(os_, g * self.Cog), bias=_at(bt, g * self.Cog),
                         R=_at(R, g * self.Cog), O_planes=_at(op, g * self.Cog))

    def bwd(self):
        eng = self.eng
        g = g_w = self.out_grad()
        if self.bwd_f32:
            # fp32 gradient ("mix", non-local theta / phi / g): WGRAD and the bias sum read it as it is, DGRAD its fp16 copy
            g_w = g
            if self.d_d is not None and not self.bwd_split:
                g = self.out.root.slot.value_half()       # (left by the launch that produced the gradient: GradSlot.half_buf)
                if g is None:
                    g = eng.scratch_act(self.out.numel)
                    hip.call("vlfb_cast", hip.ptr(g_w), hip.F32, hip.ptr(g), eng.bcode, self.out.numel)
        gp = self.out.root.slot.value_planes()         # term planes of the finished output gradient, or None
        if self.residual is not None and self.residual.needs_grad and not self.residual.detached:
            self.residual.root.slot.contribute_alias(g, self.out.root.slot.value_lo())
        if self.d_w is not None or (self.cbname and eng.is_trainable(self.cbname)):
            # weight / bias gradients are leaves of the backward graph: they run on the side stream
            # and overlap the dgrad chain (they only have to be finished before all-reduce / solver)
            eng.issue_param_grads(lambda: self._param_grads(g_w, gp))
        if self.d_d is not None and self.bwd_split:
            # (the FBO head) fp32 gradient operand, fp32 slot: the split-bf16 DGRAD with the slot's earlier contribution and
            # the fp32 values as the ReLU mask in its epilogue
            self.x.root.slot.contribute(lambda out, add, mask: hip.conv_run(self.d_d, g_w, self.w_d, None, out, R=add, mask=mask))
        elif self.d_d is not None and eng.mix and self.x.root.grad_f32:
            # (Engine._plan_head_f32, AttentionStep) the input's gradient slot is fp32: the fp16 DGRAD writes its fp32
            # accumulators there (out_dtype F32; GradSlot adds an earlier contribution in fp32)
            assert self.group == 1 and self.dx_f32
            xs = self.x.root.slot

            def dgrad_f32(out, add, mask):
                hb = xs.half_dest(out, add, mask)
                hip.conv_run(self.d_d, g, self.w_d, None, out, O_lo=hb)
                xs.half_valid = hb is not None
            xs.contribute(dgrad_f32, supports_add=False, supports_mask=False)
        elif self.d_d is not None:
            # the gradient operand as planes when it has them and this launch can take them (plain rows or taps that
            # span whole k-tiles at unit stride); the input gradient's planes when this is its last contribution
            a_pl = gp is not None and self.dgrad_takes_planes
            def dgrad(out, add, mask, planes):
                kw = {}
                if a_pl:
                    kw.update(a_planes=2, a_pstride=gp.numel() // 2)
                if planes is not None:
                    kw.update(o_planes=2, o_pstride=planes.numel() // 2)
                base = self.d_d
                xs = self.x.root.slot
                # the sparse launch leaves three quarters of the rows untouched: legal only as an IN-PLACE second contribution,
                # low term included (a two-term slot whose first contribution left no low term would keep stale rows of buf_lo)
                inplace = add is not None and add.data_ptr() == out.data_ptr() and mask is None and \
                    (xs.out_lo is None or (xs.add_lo is not None and xs.add_lo.data_ptr() == xs.out_lo.data_ptr()))
                if self.d_d_full is not None and not inplace:
                    base = self.d_d_full          # (not an in-place second contribution: the launch that writes every row)
                d = self._pl_desc(base, **kw) if kw else base
                if self.group == 1:
                    hip.conv_run(d, gp if a_pl else g, self.w_d, None, out, R=add, mask=mask, O_planes=planes,
                                 R_lo=xs.add_lo, O_lo=xs.out_lo)
                    return
                for gi in range(self.group):
                    c = gi * self.Cin_k
                    hip.conv_run(d, _at(g, gi * self.Cog), self.wd_ptr(gi), None, _at(out, c), R=_at(add, c), mask=_at(mask, c),
                                 O_planes=_at(planes, c), R_lo=_at(xs.add_lo, c), O_lo=_at(xs.out_lo, c))
            self.x.root.slot.contribute(dgrad, writes_planes=True)

    def _x_f32(self):
        """the input VALUES as an fp32 tensor (the operand of a split-bf16 WGRAD)"""
        if not self.x.root.pair:
            return self.x.storage()
        t = self.eng.join_scratch(self.x.numel)
        hip.call("vlfb_pair_join", self.x.ptr(), hip.ptr(t), self.x.numel)
        return t

    def _param_grads(self, g, gp=None):
        eng = self.eng
        if self.d_w is not None:
            s = eng.param_tensor(self.sname) if self.sname else None
            xp = self.x.root.planes
            if self.g_planes is not None:          # stem: both operands through a split pass (the clip's planes exist)
                n = self.x_planes.numel() // self.x_npl
                hip.call("vlfb_split_planes", hip.ptr(g), hip.ptr(self.g_planes), 2, 1, self.out.numel // 8, 8, 0)
                d = self._pl_desc(self.d_w, a_planes=self.x_npl, a_pstride=n, p_planes=2, p_pstride=self.out.numel)
                hip.conv_run(d, self.x_planes, None, self.g_planes, eng.grad_tensor(self.wname), rowscale=s, workspace=eng.workspace)
            elif gp is not None and xp is not None and not self.stem and self.group == 1:
                # both operands pre-split: DMA + transposed LDS reads, no VALU in the k-loop
                d = self._pl_desc(self.d_w, a_planes=2, a_pstride=xp.numel() // 2, p_planes=2, p_pstride=gp.numel() // 2, wgrad_bias=0)
                hip.conv_run(d, xp, None, gp, eng.grad_tensor(self.wname), rowscale=s, workspace=eng.workspace)
            elif self.group > 1:
                xsrc, gw = (self._x_f32() if self.bwd_f32 else self.x.bstorage()), eng.grad_tensor(self.wname)
                gb = eng.grad_tensor(self.cbname) if self.d_w.wgrad_bias else None
                for gi in range(self.group):
                    hip.conv_run(self.d_w, _at(xsrc, gi * self.Cin_k), None, _at(g, gi * self.Cog), _at(gw, gi * self.wblk),
                                 rowscale=_at(s, gi * self.Cog), workspace=eng.workspace, dbias=_at(gb, gi * self.Cog))
                if gb is not None:
                    return
            elif self.d_w.wgrad_bias:
                hip.conv_run(self.d_w, self._x_f32() if self.bwd_f32 else self.x.bstorage(), None, g,
                             eng.grad_tensor(self.wname), rowscale=s, workspace=eng.workspace, dbias=eng.grad_tensor(self.cbname))
                return
            else:
                hip.conv_run(self.d_w, self._x_f32() if self.bwd_f32 else self.x.bstorage(), None, g,
                             eng.grad_tensor(self.wname), rowscale=s, workspace=eng.workspace)
            if self.stem:   # keep the zero padding of the packed stem weight exactly zero
                gw = eng.grad_tensor(self.wname)
                hip.call("vlfb_add", hip.ptr(gw), None, hip.ptr(gw), hip.ptr(eng.stem_mask), hip.F32,
                         gw.numel(), 0)
        if self.cbname and eng.is_trainable(self.cbname):
            Cout = self.out.shape[1]
            gb = eng.grad_tensor(self.cbname)
            if self.sname:
                hip.call("vlfb_colsum", hip.ptr(g), hip.dtype_code(g.dtype), self.out.rows, Cout, Cout, hip.ptr(self.cb_tmp), 0)
                hip.call("vlfb_affine_nd_bwd", hip.ptr(self.cb_tmp), hip.ptr(eng.param_tensor(self.sname)),
                         hip.ptr(gb), 1, Cout, 1)
            else:
                hip.call("vlfb_colsum", hip.ptr(g), hip.dtype_code(g.dtype), self.out.rows, Cout, Cout, hip.ptr(gb), 0)
            if self.gscale != 1.0:
                hip.call("vlfb_scale_inplace", hip.ptr(gb), gb.numel(), 1.0 / self.gscale)


class PoolStep(Step):
    def __init__(self, eng, x, out, kernels, strides, pads, is_max):
        Step.__init__(self, eng)
        self.x, self.out = x, out
        self.inputs, self.outputs = [x], [out]
        self.k, self.s, self.p, self.is_max = kernels, strides, pads, is_max

    def name(self):
        return ("maxpool:" if self.is_max else "avgpool:") + self.out.name

    def setup(self):
        eng = self.eng
        N, Cc, T, H, W = self.x.shape
        _, _, To, Ho, Wo = self.out.shape
        # (a two-plane input: the max pool copies the selected element's planes, the average pool writes fp32)
        assert bool(self.x.root.pair) == bool(self.out.root.pair) or (self.x.root.pair and not self.is_max), self.out.name
        self.desc = hip.pool_desc(hip.F16PAIR if self.x.root.pair else eng.code, N, T, H, W, Cc, To, Ho, Wo, self.k, self.s, self.p)
        self.desc_b = self.desc if eng.bcode == eng.code else hip.pool_desc(eng.bcode, N, T, H, W, Cc, To, Ho, Wo, self.k, self.s, self.p)
        if self.is_max and eng.train and self.x.relu:
            eng.want_half(self.out)               # vlfb_maxpool_relu_bwd reads the pooled values as the ReLU mask
        self.argmax = None
        if self.is_max and eng.train:
            nbytes = hip.lib().vlfb_pool_argmax_bytes(C.byref(self.desc))
            self.argmax = torch.empty(self.out.numel * nbytes, device=eng.device, dtype=torch.uint8)

    def fwd(self):
        if self.is_max:
            hip.call("vlfb_maxpool_fwd", C.byref(self.desc), self.x.ptr(), self.out.ptr(), hip.ptr(self.argmax))
        else:
            hip.call("vlfb_avgpool_fwd", C.byref(self.desc), self.x.ptr(), self.out.ptr())

    def bwd(self):
        if not self.grad_inputs():
            return
        if getattr(self, "two_term_dx", False):
            # "mix": the fp32 pooled gradient -> a two-term fp16 input gradient (Engine._analyse_grads)
            g32, xs = self.out_grad(), self.x.root.slot

            def fn2(out, add, mask, planes):
                assert add is None and xs.out_lo is not None
                hip.call("vlfb_avgpool_bwd_two_term", C.byref(self.desc_b), hip.ptr(g32), hip.ptr(out), hip.ptr(xs.out_lo), hip.ptr(mask))
            xs.contribute(fn2, writes_planes=True)
            return
        g = self.g_as(self.out_grad(), self.out, self.x)      # ("mix": where the head's fp32 gradient re-enters fp16)
        if self.is_max:
            def fn(out, add, mask):
                if mask is not None and add is None:
                    # sole consumer of a ReLU output: the mask is `pooled value > 0` (see vlfb_maxpool_relu_bwd)
                    hip.call("vlfb_maxpool_relu_bwd", C.byref(self.desc_b), hip.ptr(g), hip.ptr(self.argmax),
                             self.out.bptr(), hip.ptr(out))
                else:
                    hip.call("vlfb_maxpool_bwd", C.byref(self.desc_b), hip.ptr(g), hip.ptr(self.argmax), hip.ptr(out),
                             hip.ptr(add), hip.ptr(mask))
        else:
            fn = lambda out, add, mask: hip.call("vlfb_avgpool_bwd", C.byref(self.desc_b), hip.ptr(g),
                                                 hip.ptr(out), hip.ptr(add), hip.ptr(mask))
        self.x.root.slot.contribute(fn)


class AttentionStep(Step):
    """BatchMatMul(trans_a) -> Scale -> Softmax(axis=2) -> BatchMatMul(trans_b)
    (nonlocal_helper.py:94-121, lfb_helper.py:223-234) with theta/phi/g stored [B][L][Ci]."""

    def __init__(self, eng, theta, phi, g, prob, out, scale, dot=False):
        """dot: the dot-product variant (NONLOCAL.USE_SOFTMAX False): prob = theta^T phi / L2, no softmax"""
        Step.__init__(self, eng)
        self.theta, self.phi, self.g, self.prob, self.out, self.scale = theta, phi, g, prob, out, scale
        self.dot = bool(dot)
        self.inputs, self.outputs = [theta, phi, g], [out]
        self.aux_outputs = [prob]

    def name(self):
        return "attention:" + self.out.name

    def setup(self):
        eng = self.eng
        B, Ci, L1 = self.theta.shape
        L2 = self.phi.shape[2]
        self.B, self.Ci, self.L1, self.L2 = B, Ci, L1, L2
        code, bcode = eng.code, eng.bcode
        self.single = (L1 == 1)
        # One bank per CLIP instead of one copy per RoI (inference): the `lfb` blob was planned with a row per
        # clip, lfb_1x1 and the phi / g convs of every FBO layer run on n_clips x K rows instead of R x K, and a query row
        # reads the bank of its clip -- the batch-index column of `proposals` (ava_data_input.py:175-192 writes it).
        self.kv_owner = None
        if self.phi.shape[0] != B:
            prop = [b for n, b in eng.env.items() if str(n).startswith("proposals")]
            if not (self.single and not eng.train and prop and self.phi.shape[0] == eng.plan_clips and self.g.shape[0] == self.phi.shape[0]):
                raise ValueError("attention %s: %d query rows against %d banks -- one bank per clip is an inference-mode plan of a "
                                 "RoI head (no dropout on the bank)" % (self.out.name, B, self.phi.shape[0]))
            self.kv_owner = prop[0].root
        if eng.train:
            for t in (self.theta, self.phi, self.g) + (() if self.single else (self.prob,)):
                eng.want_half(t)
        if self.single:
            self.ds_ws = torch.empty(B * L2, device=eng.device, dtype=torch.float32)
            return
        # split math: the B operands (phi, g^T, g, phi^T) are activations, expanded into bf16 term planes by
        # vlfb_split_planes right before the product that reads them (3 planes forward, 2 backward)
        mf, mb = eng.math_fwd, eng.math_bwd
        pl = dict(b_pstride=B * L2 * Ci) if eng.split else {}
        bpl = pl if mb != hip.MATH_NATIVE else {}                  # ("mix": split forward products, native fp16 backward)
        self.bsplit = bool(bpl)
        gemm = lambda dtype=code, **kw: hip.conv_desc(mode=hip.FPROP, dtype=dtype, N=1, Tr=1, Hr=1, Wr=L1, Ts=1, Hs=1,
                                                      Ws=L1, batch=B, **kw)
        self.d_s = gemm(out_dtype=hip.F32, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2, math=mf, **pl)
        # theta and phi as two fp16 planes (Engine._plan_pairs): the scores are a batched two-plane product (no plane pass over phi,
        # nothing converted in the loop); the backward reads their hi planes as before
        self.s_pair = bool(self.theta.root.pair and self.phi.root.pair)
        assert bool(self.theta.root.pair) == bool(self.phi.root.pair) and not self.g.root.pair
        if self.s_pair:
            self.d_s = gemm(dtype=hip.F16, out_dtype=hip.F32, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2,
                            math=hip.MATH_F16X3, a_pstride=B * L1 * Ci, b_pstride=B * L2 * Ci)
        self.o_pair = bool(self.out.root.pair)           # (Engine._plan_pairs: y leaves the P . g product as two fp16 planes)
        self.d_y = gemm(out_dtype=hip.F16 if self.o_pair else code, Cs=L2, Cn=Ci, a_bstride=L1 * L2, b_bstride=Ci * L2, o_bstride=L1 * Ci, math=mf, **pl)
        self.d_dp = gemm(dtype=bcode, out_dtype=hip.F32, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2, math=mb, **bpl)
        # "mix": dP = dY . g^T as split-bf16 products on fp32 operands (dY through a cast) and the softmax backward on the
        # fp32 probabilities -- what follows cancels the part of dP that is common to a row
        self.precise = bool(eng.mix and eng.MIX_NL_F32)
        if self.precise:
            self.d_dp = gemm(dtype=hip.F32, out_dtype=hip.F32, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2,
                             math=hip.MATH_BF16X3, **pl)
        if eng.split:
            eng.need_scratch_planes(3 * B * L2 * Ci)
        # fp16: dS = scale * P o (dP - <dP, P>) is ~ 1 / L2 of an activation gradient and would leave the fp16
        # range (6e-8) for long key axes (1568 keys in 64-frame clips); it is stored times a power of two, which
        # the two products that consume it divide out again in their epilogues (alpha).  Exact; 1 elsewhere.
        self.ds_scale = float(16 << max(L2 - 1, 1).bit_length()) if eng.btdtype == torch.float16 else 1.0
        # ... and the gradients of theta / phi themselves (again ~ 1 / L2 of an activation gradient) are stored
        # times Blob.grad_scale (set at lowering), which the theta / phi convs divide out (ConvStep.gscale)
        gs_th, gs_ph = float(self.theta.root.grad_scale), float(self.phi.root.grad_scale)
        o_th = hip.F32 if self.theta.root.grad_f32 else bcode
        o_ph = hip.F32 if self.phi.root.grad_f32 else bcode
        o_g = hip.F32 if self.g.root.grad_f32 else bcode
        self.d_dth = gemm(dtype=bcode, out_dtype=o_th, Cs=L2, Cn=Ci, a_bstride=L1 * L2, b_bstride=Ci * L2, o_bstride=L1 * Ci,
                          alpha=gs_th / self.ds_scale, math=mb, **bpl)
        if o_th == hip.F32 and bcode != hip.F32 and mb == hip.MATH_NATIVE:
            self.theta.root.grad_half_src = True        # (AttentionStep._dtheta: the fp16 rounding next to the fp32 output)
        # contract over L1: out[L2][Ci] = sum_l P[l][L2] * A[l][Ci]
        self.d_tn = hip.conv_desc(mode=hip.WGRAD, dtype=bcode, out_dtype=o_g, N=1, Tr=1, Hr=1, Wr=L1, Ts=1,
                                  Hs=1, Ws=L1, Cs=Ci, Cn=L2, batch=B, a_bstride=L1 * Ci, p_bstride=L1 * L2,
                                  o_bstride=L2 * Ci, splits=1, math=mb)
        self.d_tn_phi = hip.conv_desc(mode=hip.WGRAD, dtype=bcode, out_dtype=o_ph, N=1, Tr=1, Hr=1, Wr=L1, Ts=1,
                                      Hs=1, Ws=L1, Cs=Ci, Cn=L2, batch=B, a_bstride=L1 * Ci, p_bstride=L1 * L2,
                                      o_bstride=L2 * Ci, splits=1, alpha=gs_ph / self.ds_scale, math=mb)
        # 16-bit paths: scores + row softmax (and their backward) in one kernel each, the fp32 score matrix never
        # exists (csrc/vlfb_attn.hip) -- per direction, and only where the library reports the fused kernel as
        # measured faster; otherwise GEMM -> fp32 scratch -> softmax kernels
        self.fused_fwd = bool(hip.lib().vlfb_attn_scores_supported(code, L1, L2, Ci) & hip.ATTN_FWD_FASTER) and not self.dot
        self.fused_bwd = bool(hip.lib().vlfb_attn_scores_supported(bcode, L1, L2, Ci) & hip.ATTN_BWD_FASTER) and \
            not self.precise and not self.dot
        if self.dot:
            # p = theta^T phi / L2 comes straight out of the scores product (alpha), and so does dS = dP / L2 backward
            self.d_s = gemm(out_dtype=code, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci, o_bstride=L1 * L2, math=mf,
                            alpha=1.0 / L2, **pl)
            if self.precise:
                self.d_dp = gemm(dtype=hip.F32, out_dtype=hip.F32, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci,
                                 o_bstride=L1 * L2, math=hip.MATH_BF16X3, alpha=self.ds_scale / L2, **pl)
            else:
                self.d_dp = gemm(dtype=bcode, out_dtype=bcode, Cs=Ci, Cn=L2, a_bstride=L1 * Ci, b_bstride=L2 * Ci,
                                 o_bstride=L1 * L2, math=mb, alpha=self.ds_scale / L2, **bpl)
        if not (self.fused_fwd and self.fused_bwd):
            eng.need_scratch_f32(B * L1 * L2 + (B * L1 * Ci if self.precise else 0))
        self.dy_f32 = bool(self.out.root.grad_f32)         # ("mix": dY arrives in fp32; the fp16 products read a cast ...
        if self.dy_f32 and not self.single:
            self.out.root.grad_half = True                 # ... or the copy its producer leaves: GradSlot.half_buf)
        eng.need_scratch_act(B * L1 * L2 + B * Ci * L2 + (B * L1 * Ci if self.dy_f32 else 0))

    def _planes(self, src, nplanes, transpose):
        """bf16 term planes of a (B, L2, Ci) fp32 activation (optionally transposed per batch element): the B
        operand of a split-math product"""
        dst = self.eng.scratch_planes(nplanes * self.B * self.L2 * self.Ci)
        hip.call("vlfb_split_planes", hip.ptr(src), hip.ptr(dst), nplanes, self.B, self.L2, self.Ci, int(transpose))
        return dst

    def fwd(self):
        eng = self.eng
        B, Ci, L1, L2 = self.B, self.Ci, self.L1, self.L2
        if self.single and self.kv_owner is not None:
            hip.call("vlfb_fbo_attn_fwd_shared", self.theta.ptr(), self.phi.ptr(), self.g.ptr(), self.prob.ptr(),
                     self.out.ptr(), eng.code, B, L2, Ci, Ci, self.scale, self.kv_owner.ptr(), 5)
            return
        if self.single:
            hip.call("vlfb_fbo_attn_fwd", self.theta.ptr(), self.phi.ptr(), self.g.ptr(), self.prob.ptr(),
                     self.out.ptr(), eng.code, B, L2, Ci, Ci, self.scale)
            return
        if eng.split:
            nf = 3 if eng.math_fwd == hip.MATH_BF16X6 else 2
            if self.dot:
                hip.conv_run(self.d_s, self.theta.storage(), self._planes(self.phi.storage(), nf, False), None, self.prob.storage())
            else:
                S = eng.scratch_f32(B * L1 * L2)
                if self.s_pair:
                    hip.conv_run(self.d_s, self.theta.storage(), self.phi.storage(), None, S)
                else:
                    hip.conv_run(self.d_s, self.theta.storage(), self._planes(self.phi.storage(), nf, False), None, S)
                hip.call("vlfb_softmax_fwd", hip.ptr(S), self.prob.ptr(), eng.code, B * L1, L2, self.scale)
            hip.conv_run(self.d_y, self.prob.storage(), self._planes(self.g.storage(), nf, True), None, self.out.storage(),
                         O_lo=self.out.lo() if self.o_pair else None)
            return
        if self.dot:
            hip.conv_run(self.d_s, self.theta.storage(), self.phi.storage(), None, self.prob.storage())
        elif self.fused_fwd:
            hip.call("vlfb_attn_scores_fwd", self.theta.ptr(), self.phi.ptr(), self.prob.ptr(), eng.code, B, L1, L2, Ci,
                     self.scale)
        else:
            S = eng.scratch_f32(B * L1 * L2)
            hip.conv_run(self.d_s, self.theta.storage(), self.phi.storage(), None, S)
            hip.call("vlfb_softmax_fwd", hip.ptr(S), self.prob.ptr(), eng.code, B * L1, L2, self.scale)
        gT = eng.scratch_act(B * Ci * L2)
        hip.call("vlfb_transpose2d", self.g.ptr(), hip.ptr(gT), eng.code, B, L2, Ci)
        hip.conv_run(self.d_y, self.prob.storage(), gT, None, self.out.storage())

    def _dtheta(self, th, dS, phT):
        """d theta = dS . phi: into theta's slot; a 16-bit product with an fp32 output also leaves the result rounded to the
        operand type for the theta conv's DGRAD (GradSlot.half_buf)"""
        copy16 = self.d_dth.out_dtype == hip.F32 and self.d_dth.dtype != hip.F32 and self.d_dth.math == hip.MATH_NATIVE

        def fn(out, add, mask):
            hb = th.half_dest(out, add, mask) if copy16 else None
            hip.conv_run(self.d_dth, dS, phT, None, out, O_lo=hb)
            th.half_valid = hb is not None
        th.contribute(fn, supports_add=False, supports_mask=False)

    def bwd(self):
        eng = self.eng
        B, Ci, L1, L2 = self.B, self.Ci, self.L1, self.L2
        dY = self.out_grad()
        th, ph, gg = self.theta.root.slot, self.phi.root.slot, self.g.root.slot
        if self.single:
            # single consumer each: the three gradients are written in one launch
            for s in (th, ph, gg):
                s._flags()
                s.cur = s.buf
            if self.theta.root.grad_f32:
                # (the FBO head of "mix", Engine._plan_head_f32: fp32 gradients in and out, the fp32 forward values)
                assert self.phi.root.grad_f32 and self.g.root.grad_f32 and self.out.root.grad_f32
                hip.call("vlfb_fbo_attn_bwd", hip.ptr(dY), self.theta.ptr(), self.phi.ptr(), self.g.ptr(),
                         self.prob.ptr(), hip.ptr(th.buf), hip.ptr(ph.buf), hip.ptr(gg.buf), hip.ptr(self.ds_ws),
                         hip.F32, B, L2, Ci, Ci, self.scale)
                return
            hip.call("vlfb_fbo_attn_bwd", hip.ptr(dY), self.theta.bptr(), self.phi.bptr(), self.g.bptr(),
                     self.prob.ptr(), hip.ptr(th.buf), hip.ptr(ph.buf), hip.ptr(gg.buf), hip.ptr(self.ds_ws),
                     eng.bcode, B, L2, Ci, Ci, self.scale)
            return
        P = self.prob.bstorage()
        dY32 = None
        if self.dy_f32:
            # fp32 dY (the `out` conv's DGRAD accumulators): the split product dP reads it as it is, the fp16 products a cast
            dY32 = dY
            dY = self.out.root.slot.value_half()          # (left by the `out` conv's DGRAD: GradSlot.half_buf)
            if dY is None:
                dY = eng.scratch_act(B * L1 * L2 + B * Ci * L2 + B * L1 * Ci)[B * L1 * L2 + B * Ci * L2:]
                hip.call("vlfb_cast", hip.ptr(dY32), hip.F32, hip.ptr(dY), eng.bcode, B * L1 * Ci)
        if self.dot:
            # dot-product variant: dS = dP / L2 is the scores-gradient product itself (alpha), no softmax Jacobian
            gg.contribute(lambda out, add, mask: hip.conv_run(self.d_tn, dY, None, P, out),
                          supports_add=False, supports_mask=False)
            act = eng.scratch_act(B * L1 * L2 + B * Ci * L2)
            dS, phT = act[:B * L1 * L2], act[B * L1 * L2:]
            if self.precise:
                f32 = eng.scratch_f32(B * L1 * L2 + B * L1 * Ci)
                dP = f32[:B * L1 * L2]
                if dY32 is None:
                    dY32 = f32[B * L1 * L2:]
                    hip.call("vlfb_cast", hip.ptr(dY), eng.bcode, hip.ptr(dY32), hip.F32, B * L1 * Ci)
                hip.conv_run(self.d_dp, dY32, self._planes(self.g.storage(), 2, False), None, dP)
                hip.call("vlfb_cast", hip.ptr(dP), hip.F32, hip.ptr(dS), eng.bcode, B * L1 * L2)
            else:
                hip.conv_run(self.d_dp, dY, self._planes(self.g.storage(), 2, False) if self.bsplit else self.g.bstorage(), None, dS)
            if self.bsplit:
                phT = self._planes(self.phi.storage(), 2, True)
            else:
                hip.call("vlfb_transpose2d", self.phi.bptr(), hip.ptr(phT), eng.bcode, B, L2, Ci)
            self._dtheta(th, dS, phT)
            ph.contribute(lambda out, add, mask: hip.conv_run(self.d_tn_phi, self.theta.bstorage(), None, dS, out),
                          supports_add=False, supports_mask=False)
            return
        if self.precise:
            f32 = eng.scratch_f32(B * L1 * L2 + B * L1 * Ci)
            dP = f32[:B * L1 * L2]
            if dY32 is None:
                dY32 = f32[B * L1 * L2:]
                hip.call("vlfb_cast", hip.ptr(dY), eng.bcode, hip.ptr(dY32), hip.F32, B * L1 * Ci)
            hip.conv_run(self.d_dp, dY32, self._planes(self.g.storage(), 2, False), None, dP)
        elif not self.fused_bwd:
            dP = eng.scratch_f32(B * L1 * L2)
            hip.conv_run(self.d_dp, dY, self._planes(self.g.storage(), 2, False) if self.bsplit else self.g.bstorage(), None, dP)
        gg.contribute(lambda out, add, mask: hip.conv_run(self.d_tn, dY, None, P, out),
                      supports_add=False, supports_mask=False)
        act = eng.scratch_act(B * L1 * L2 + B * Ci * L2)
        dS = act[:B * L1 * L2]
        phT = act[B * L1 * L2:]
        if self.fused_bwd:
            hip.call("vlfb_attn_scores_bwd", hip.ptr(dY), self.g.bptr(), hip.ptr(P), hip.ptr(dS), eng.bcode, B, L1, L2, Ci,
                     self.scale * self.ds_scale)
        elif self.precise:
            hip.call("vlfb_softmax_bwd_p32", hip.ptr(dP), self.prob.ptr(), hip.ptr(dS), eng.bcode, B * L1, L2,
                     self.scale * self.ds_scale)
        else:
            hip.call("vlfb_softmax_bwd", hip.ptr(dP), hip.ptr(P), hip.ptr(dS), eng.bcode, B * L1, L2,
                     self.scale * self.ds_scale)
        if self.bsplit:
            phT = self._planes(self.phi.storage(), 2, True)
        else:
            hip.call("vlfb_transpose2d", self.phi.bptr(), hip.ptr(phT), eng.bcode, B, L2, Ci)
        self._dtheta(th, dS, phT)
        ph.contribute(lambda out, add, mask: hip.conv_run(self.d_tn_phi, self.theta.bstorage(), None, dS, out),
                      supports_add=False, supports_mask=False)


class AddStep(Step):
    """Sum([a, b]) [+ ReLU] that could not be folded into a conv epilogue"""

    def __init__(self, eng, a, b, out, relu):
        Step.__init__(self, eng)
        self.a, self.b, self.out, self.relu = a, b, out, relu
        self.inputs, self.outputs = [a, b], [out]

    def name(self):
        return "add:" + self.out.name

    def setup(self):
        pass

    def fwd(self):
        hip.call("vlfb_add", self.a.ptr(), self.b.ptr(), self.out.ptr(), None, self.eng.code,
                 self.out.numel, int(self.relu))

    def bwd(self):
        g = self.out_grad()
        g_lo = self.out.root.slot.value_lo()
        for x in self.grad_inputs():
            x.root.slot.contribute_alias(g, g_lo)


class ReluStep(Step):
    def __init__(self, eng, x, out):
        Step.__init__(self, eng)
        self.x, self.out = x, out
        self.inputs, self.outputs = [x], [out]

    def name(self):
        return "relu:" + self.out.name

    def setup(self):
        pass

    def fwd(self):
        hip.call("vlfb_relu_fwd", self.x.ptr(), self.out.ptr(), self.eng.code, self.out.numel)

    def bwd(self):
        if self.grad_inputs():
            self.x.root.slot.contribute_alias(self.out_grad())   # already masked by out > 0


class BNStep(Step):
    """SpatialBN (model_builder_video.py:186-190, resnet_video.py:185-188, nonlocal_helper.py:147-151): batch statistics of
    THIS GPU's rows in a training net (and the running statistics updated), the running statistics in a test / val net.
    vlfb_bn_fwd / vlfb_bn_bwd; scale and bias are trained, `_rm` / `_riv` are computed parameters."""

    def __init__(self, eng, x, out, sname, bname, rmname, rvname, eps, momentum, is_test):
        Step.__init__(self, eng)
        self.x, self.out = x, out
        self.sname, self.bname, self.rmname, self.rvname = sname, bname, rmname, rvname
        self.eps, self.momentum, self.is_test = float(eps), float(momentum), int(bool(is_test))
        self.inputs, self.outputs = [x], [out]

    def name(self):
        return "bn:" + self.out.name

    def setup(self):
        eng = self.eng
        if eng.mix:
            raise NotImplementedError("SpatialBN graphs are not built for the 'mix' dtype (use 'split' / 'fp32' / 16-bit)")
        self.rows, self.C = self.x.rows, self.x.C
        self.params = [n for n in (self.sname, self.bname) if eng.is_trainable(n)]
        nbytes = hip.query_workspace(hip.WS_BN, (eng.code, self.rows, self.C))
        self.ws = torch.empty(nbytes // 4, device=eng.device, dtype=torch.float32)
        self.save = None if self.is_test else torch.empty(2 * self.C, device=eng.device, dtype=torch.float32)

    def fwd(self):
        eng = self.eng
        P = eng.param_tensor
        mean = hip.ptr(self.save) if self.save is not None else None
        istd = hip.ptr(self.save) + 4 * self.C if self.save is not None else None
        hip.call("vlfb_bn_fwd", self.x.ptr(), self.out.ptr(), hip.ptr(P(self.sname)), hip.ptr(P(self.bname)),
                 hip.ptr(P(self.rmname)), hip.ptr(P(self.rvname)), mean, istd, hip.ptr(self.ws), self.ws.numel() * 4,
                 eng.code, self.rows, self.C, self.eps, self.momentum, self.is_test)

    def bwd(self):
        eng = self.eng
        assert self.save is not None, "backward through a test-mode SpatialBN"
        g = self.out_grad()
        dgamma = hip.ptr(eng.grad_tensor(self.sname)) if eng.is_trainable(self.sname) else None
        dbeta = hip.ptr(eng.grad_tensor(self.bname)) if eng.is_trainable(self.bname) else None

        def launch(out, add=None, mask=None):
            hip.call("vlfb_bn_bwd", hip.ptr(g), self.x.ptr(), hip.ptr(eng.param_tensor(self.sname)), hip.ptr(self.save),
                     hip.ptr(self.save) + 4 * self.C, hip.ptr(out), dgamma, dbeta, hip.ptr(self.ws), self.ws.numel() * 4,
                     eng.code, self.rows, self.C, 1.0)
        if self.grad_inputs():
            self.x.root.slot.contribute(launch, supports_add=False, supports_mask=False)
        elif dgamma is not None or dbeta is not None:
            launch(None)


class LayerNormStep(Step):
    def __init__(self, eng, x, out, eps):
        Step.__init__(self, eng)
        self.x, self.out, self.eps = x, out, eps
        self.inputs, self.outputs = [x], [out]

    def name(self):
        return "layernorm:" + self.out.name

    def setup(self):
        self.rows = self.x.shape[0]
        self.cols = self.x.numel // self.rows
        self.rstd = torch.empty(self.rows, device=self.eng.device, dtype=torch.float32)
        if self.eng.train:
            self.eng.want_half(self.out)      # the backward reads the normalised values

    def fwd(self):
        hip.call("vlfb_layernorm_fwd", self.x.ptr(), self.out.ptr(), hip.ptr(self.rstd), self.eng.code,
                 self.rows, self.cols, self.eps)

    def bwd(self):
        if not self.grad_inputs():
            return
        g = self.g_as(self.out_grad(), self.out, self.x)
        f32 = bool(self.x.root.grad_f32)              # ("mix", the FBO head: fp32 gradient, the fp32 normalised values)
        self.x.root.slot.contribute(
            lambda out, add, mask: hip.call("vlfb_layernorm_bwd", hip.ptr(g), self.out.ptr() if f32 else self.out.bptr(),
                                            hip.ptr(self.rstd), hip.ptr(out), hip.F32 if f32 else self.eng.bcode,
                                            self.rows, self.cols),
            supports_add=False, supports_mask=False)


class DropoutStep(Step):
    def __init__(self, eng, x, out, ratio):
        Step.__init__(self, eng)
        self.x, self.out, self.ratio = x, out, ratio
        self.inputs, self.outputs = [x], [out]

    def name(self):
        return "dropout:" + self.out.name

    def setup(self):
        self.mask = torch.empty(self.out.numel, device=self.eng.device, dtype=torch.uint8)
        shp = self.x.shape
        self.rows, self.ch = shp[0], shp[self.x.caxis]
        self.inner = self.x.numel // (self.rows * self.ch)

    def fwd(self):
        if self.eng._dev_scalars:      # captured step: the seed of the iteration is read from device memory
            hip.call("vlfb_dropout_fwd_dev", self.x.ptr(), self.out.ptr(), hip.ptr(self.mask), self.eng.code,
                     self.rows, self.inner, self.ch, self.ratio, hip.ptr(self.eng._scalars_dev) + 8 * self.seed_slot)
            return
        seed = dropout_seed(self.eng.base_seed, self.out.name, self.eng.iteration, self.eng.replica)
        hip.call("vlfb_dropout_fwd", self.x.ptr(), self.out.ptr(), hip.ptr(self.mask), self.eng.code, self.rows,
                 self.inner, self.ch, self.ratio, seed)

    def bwd(self):
        if not self.grad_inputs():
            return
        g = self.g_as(self.out_grad(), self.out, self.x)
        self.x.root.slot.contribute(
            lambda out, add, mask: hip.call("vlfb_dropout_bwd", hip.ptr(g), hip.ptr(self.mask), hip.ptr(out),
                                            self.gcode(self.x), self.out.numel, self.ratio),
            supports_add=False, supports_mask=False)


class RoiAlignMaxStep(Step):
    """RoIAlign + resolution x resolution MaxPool (head_helper.py:101-116)"""

    def __init__(self, eng, feat, rois, out, pooled, spatial_scale):
        Step.__init__(self, eng)
        self.feat, self.rois, self.out = feat, rois, out
        self.pooled, self.spatial_scale = pooled, spatial_scale
        self.inputs, self.outputs = [feat], [out]

    def name(self):
        return "roialign:" + self.out.name

    def setup(self):
        eng = self.eng
        self.N, self.Cc, self.H, self.W = self.feat.shape
        self.R = self.rois.shape[0]
        self.argbin = torch.empty(self.R * self.Cc, device=eng.device, dtype=torch.uint8)
        # integer decisions of every bilinear sample (thread 0 of a workgroup re-derives them serially):
        # only recorded when a test asks for them (Engine.debug_roi), never on the hot path
        self.dbg = (torch.zeros(self.R * self.pooled * self.pooled * 8, device=eng.device, dtype=torch.int32)
                    if eng.debug_roi else None)
        if eng.train:
            self.dfeat = torch.empty(self.feat.numel, device=eng.device, dtype=torch.float32)

    def fwd(self):
        hip.call("vlfb_roi_align_max_fwd", self.feat.ptr(), self.eng.code, self.rois.ptr(), self.out.ptr(),
                 hip.ptr(self.argbin), hip.ptr(self.dbg), self.N, self.H, self.W, self.Cc, self.R, self.pooled,
                 self.spatial_scale)

    def bwd(self):
        if not self.grad_inputs():
            return
        eng = self.eng
        g = self.out_grad()
        hip.call("vlfb_zero_f32", hip.ptr(self.dfeat), self.dfeat.numel())
        hip.call("vlfb_roi_align_max_bwd", hip.ptr(g), self.gcode(self.out), self.rois.ptr(), hip.ptr(self.argbin),
                 hip.ptr(self.dfeat), self.N, self.H, self.W, self.Cc, self.R, self.pooled, self.spatial_scale)
        self.feat.root.slot.contribute(
            lambda out, add, mask: hip.call("vlfb_cast", hip.ptr(self.dfeat), hip.F32, hip.ptr(out), self.gcode(self.feat),
                                            self.dfeat.numel()),
            supports_add=False, supports_mask=False)


class ConcatStep(Step):
    """Concat(axis=1) of (R, Ci, 1, 1, 1) blobs"""

    def __init__(self, eng, parts, out):
        Step.__init__(self, eng)
        self.parts, self.out = parts, out
        self.inputs, self.outputs = list(parts), [out]

    def name(self):
        return "concat:" + self.out.name

    def setup(self):
        self.rows = self.out.shape[0]
        self.total = self.out.shape[1]

    def fwd(self):
        off = 0
        es = self.eng.esize
        for p in self.parts:
            hip.call("vlfb_copy2d", p.ptr(), p.C, self.out.ptr() + off * es, self.total, self.eng.code,
                     self.rows, p.C)
            off += p.C

    def bwd(self):
        g = self.out_grad()
        gc = self.gcode(self.out)
        es = 4 if gc == hip.F32 else self.eng.besize
        off = 0
        for k, p in enumerate(self.parts):
            if p.needs_grad and not p.detached:
                o = off
                if self.gcode(p) == gc:
                    p.root.slot.contribute(
                        lambda out, add, mask, o=o, p=p: hip.call("vlfb_copy2d", hip.ptr(g) + o * es, self.total,
                                                                  hip.ptr(out), p.C, gc, self.rows, p.C),
                        supports_add=False, supports_mask=False)
                else:
                    # ("mix", fp32 head: a part whose producer keeps an fp16 gradient below an fp32 concat gradient, or the reverse)
                    key = "_part%d" % k
                    if getattr(self, key, None) is None:
                        setattr(self, key, torch.empty(self.rows * p.C, device=self.eng.device, dtype=g.dtype))
                    tmp = getattr(self, key)

                    def fn(out, add, mask, o=o, p=p, tmp=tmp):
                        hip.call("vlfb_copy2d", hip.ptr(g) + o * es, self.total, hip.ptr(tmp), p.C, gc, self.rows, p.C)
                        hip.call("vlfb_cast", hip.ptr(tmp), gc, hip.ptr(out), self.gcode(p), self.rows * p.C)
                    p.root.slot.contribute(fn, supports_add=False, supports_mask=False)
            off += p.C


class FCStep(Step):
    def __init__(self, eng, x, out, wname, bname):
        Step.__init__(self, eng)
        self.x, self.out, self.wname, self.bname = x, out, wname, bname
        self.inputs, self.outputs = [x], [out]

    def name(self):
        return "fc:" + self.out.name

    def setup(self):
        self.rows = self.x.shape[0]
        self.cin = self.x.numel // self.rows
        self.cout = self.out.shape[1]
        self.params = [n for n in (self.wname, self.bname) if self.eng.is_trainable(n)]
        if self.eng.train:
            self.eng.want_half(self.x)

    def fwd(self):
        eng = self.eng
        hip.call("vlfb_fc_fwd", self.x.ptr(), eng.code, hip.ptr(eng.param_tensor(self.wname)),
                 hip.ptr(eng.param_tensor(self.bname)), self.out.ptr(), self.rows, self.cin, self.cout)

    def bwd(self):
        eng = self.eng
        dl = self.out_grad()
        w = eng.param_tensor(self.wname)
        train = eng.is_trainable(self.wname)
        dw = eng.grad_tensor(self.wname) if train else None
        db = eng.grad_tensor(self.bname) if train else None
        f32 = bool(eng.mix and self.x.root.grad_f32)          # ("mix", fp32 head: fp32 input values and an fp32 input gradient)
        xp, xc = (self.x.ptr(), eng.code) if f32 else (self.x.bptr(), eng.bcode)
        if dw is not None:
            hip.call("vlfb_fc_bwd", xp, xc, hip.ptr(w), hip.ptr(dl), None, hip.ptr(dw), hip.ptr(db),
                     self.rows, self.cin, self.cout, 0)
        if self.grad_inputs():
            self.x.root.slot.contribute(
                lambda out, add, mask: hip.call("vlfb_fc_bwd", xp, self.gcode(self.x), hip.ptr(w), hip.ptr(dl),
                                                hip.ptr(out), None, None, self.rows, self.cin, self.cout, 0),
                supports_add=False, supports_mask=False)


LOSS_RING = 64


class LossStep(Step):
    """Sigmoid + SigmoidCrossEntropyLoss (multi-label: AVA, Charades; resnet_video.py:333-338) or
    Softmax / SoftmaxWithLoss (single-label: EPIC-Kitchens; :339-347); test mode: the probabilities only"""

    def __init__(self, eng, logits, labels, prob, loss, scale, kind="sigmoid"):
        Step.__init__(self, eng)
        self.logits, self.labels, self.prob, self.loss, self.scale = logits, labels, prob, loss, scale
        self.kernel = {"sigmoid": "vlfb_sigmoid_ce", "softmax": "vlfb_softmax_ce"}[kind]
        self.inputs = [logits]
        self.outputs = [b for b in (prob, loss) if b is not None]

    def name(self):
        return "loss"

    def setup(self):
        self.rows, self.cols = self.logits.shape[0], self.logits.shape[1]
        self.dlogits = None
        self.ring = None
        self.ring_pos = 0
        if self.loss is not None and self.eng.auto_loss_scale:
            # fp16: |dlogits| <= scale / normaliser (normaliser = #targets for the sigmoid loss, #rows for the
            # softmax loss); pick the power of two that puts that bound at 2^6, so that the gradients of a
            # production step (1/8 loss scale, ~1900 targets) and of a 2-clip test sit in the same fp16 range
            # DATA PARALLEL: gradients scaled by this factor are SUMMED across ranks, so every rank must choose the same
            # one.  The number of RoI rows differs from rank to rank and from step to step (lib/datasets/ava.py); the
            # number of clips per GPU does not (misc.py:68-72).  A RoI head therefore sizes the bound for a nominal
            # 2 RoIs per clip instead of the rows it happens to hold: the bound of |dlogits| then lies between 2^5 (4 RoIs
            # per clip) and 2^7 (1 per clip); 2 keeps the full-size parity configuration (1 clip, 2 RoIs) on the scale it
            # was validated at (the backbone's gradients sit 16+ binades below the fp16 maximum there).
            clips = self.eng.plan_clips
            rows = 2 * clips if (self.eng.plan_roi_rows and clips is not None) else self.rows
            norm = rows * self.cols if self.kernel == "vlfb_sigmoid_ce" else rows
            self.eng.loss_scale = float(2.0 ** round(math.log2(64.0 * norm / self.scale)))
        if self.loss is not None:
            self.dlogits = torch.empty(self.rows * self.cols, device=self.eng.device, dtype=torch.float32)
            # the last LOSS_RING losses stay on the device: the NaN guard (utils.misc.check_nan_losses)
            # reads them in one go every few iterations instead of syncing on `loss` every step
            self.ring = torch.zeros(LOSS_RING, device=self.eng.device, dtype=torch.float32)

    def fwd(self):
        hip.call(self.kernel, self.logits.ptr(), self.labels.ptr() if self.labels is not None else None,
                 self.prob.ptr() if self.prob is not None else None,
                 self.loss.ptr() if self.loss is not None else None, hip.ptr(self.dlogits), self.rows, self.cols,
                 self.scale)
        if self.dlogits is not None and self.eng.loss_scale != 1.0:
            hip.call("vlfb_scale_inplace", hip.ptr(self.dlogits), self.dlogits.numel(), self.eng.loss_scale)
        if not self.eng._dev_scalars:  # (a captured step pushes after the replay: the ring position is host state)
            self.ring_push()

    def ring_push(self):
        if self.ring is not None:
            self.ring.narrow(0, self.ring_pos % LOSS_RING, 1).copy_(self.loss.root.tensor.narrow(0, 0, 1), non_blocking=True)
            self.ring_pos += 1

    def recent_losses(self):
        """losses of the (at most LOSS_RING) forward passes since the previous call, oldest first; one sync"""
        n = min(self.ring_pos, LOSS_RING)
        if self.ring is None or n == 0:
            return []
        host = self.ring.cpu().numpy()
        start = self.ring_pos - n
        self.ring_pos = 0
        return [float(host[(start + i) % LOSS_RING]) for i in range(n)]

    def bwd(self):
        self.logits.root.slot.contribute_alias(self.dlogits)


# ================================================================================================
# lowering
# ================================================================================================
class Lowering(object):
    def __init__(self, eng, model, input_shapes):
        self.eng, self.model = eng, model
        self.env = {}
        self.steps = []
        self.uses = {}          # id(root blob) -> number of consumers seen so far
        self.blobs = OrderedDict()
        self.ssa = ssa_form(model.net.ops)
        # number of readers of every (name, version)
        self.readers = {}
        for op, ins, outs in self.ssa:
            for b in ins:
                self.readers[b] = self.readers.get(b, 0) + 1
        self.input_shapes = input_shapes
        self.shape_blobs = {}

    # ---- helpers -----------------------------------------------------------------------------
    def new_blob(self, name, shape, caxis, kind="act"):
        b = Blob(name, shape, caxis, kind)
        self.blobs[name + "#%d" % len(self.blobs)] = b
        return b

    def get(self, name):
        if name not in self.env:
            raise KeyError("blob %r is read before it is produced or fed" % name)
        return self.env[name]

    def add_step(self, step):
        for o in step.outputs:
            o.producer = step
        self.steps.append(step)
        return step

    def is_param(self, name):
        return name in self.model.param_init_net.fills and name in self.model.params

    # ---- main loop -----------------------------------------------------------------------------
    def run(self):
        eng = self.eng
        for name, shape in self.input_shapes.items():
            if name.startswith("data"):
                b = self.new_blob(name, shape, 1)            # stored [N][T][H][W + 2*4][4]
                b.pad_c = 4
                b.pad_w = 4
            elif name.startswith("labels"):
                b = self.new_blob(name, shape, len(shape) - 1, "i32")
            elif name.startswith("proposals"):
                b = self.new_blob(name, shape, 1, "f32")
            elif name.startswith("lfb"):
                b = self.new_blob(name, shape, 2)            # (R, K, D) row-major, D contiguous
            else:
                raise KeyError("unknown input blob %r" % name)
            b.is_input = True
            self.env[name] = b
        ops = self.ssa
        i = 0
        n = len(ops)
        while i < n:
            op, ins, outs = ops[i]
            handler = getattr(self, "lower_" + op.type, None)
            if handler is None:
                raise NotImplementedError("operator %s is not on the hot path (%r)" % (op.type, op))
            i = handler(i)
        return self.steps

    def sole_reader_is_next(self, i, out_ref, types):
        """the op after i reads out_ref, is the ONLY reader of it, and has one of `types`"""
        if i + 1 >= len(self.ssa):
            return False
        op2, ins2, _ = self.ssa[i + 1]
        return op2.type in types and out_ref in ins2 and self.readers.get(out_ref, 0) == 1

    # ---- operators -----------------------------------------------------------------------------
    def lower_StopGradient(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        v = x.view(op.outputs[0], x.shape, x.caxis)
        v.detached = True
        v.needs_grad = False
        if getattr(x, "is_input", False):
            v.is_input = True
            v.pad_c = getattr(x, "pad_c", None)
            v.pad_w = getattr(x, "pad_w", 0)
            v.root = x.root
        self.env[op.outputs[0]] = v
        return i + 1

    def lower_Conv(self, i):
        op, ins, outs = self.ssa[i]
        eng = self.eng
        x = self.get(op.inputs[0])
        wname = op.inputs[1]
        cbname = op.inputs[2] if len(op.inputs) > 2 else None
        a = op.args
        k, s, d = a["kernels"], a["strides"], a["dilations"]
        p = a["pads"][:3]
        assert list(a["pads"][:3]) == list(a["pads"][3:]), "asymmetric padding is not used by the builders"
        if len(x.shape) != 5 or x.caxis != 1:
            raise NotImplementedError("Conv input %s must be a 5-d channels-last blob" % x.name)
        N, Cin, T, H, W = x.shape
        wshape = self.model.param_init_net.fills[wname].shape
        Cout = wshape[0]
        group = int(a.get("group", 1))
        assert wshape[1] * group == Cin, "conv %s: weight expects %d x %d channels, input has %d" % (wname, group, wshape[1], Cin)
        dims = [(n_ + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for n_, kk, ss, pp, dd in zip((T, H, W), k, s, p, d)]
        out_name = op.outputs[0]
        step = ConvStep(eng, x, None, wname, cbname, k, s, p, d, group=group)
        j = i
        cur = outs[0]
        if self.sole_reader_is_next(j, cur, ("AffineNd",)):
            op2, ins2, outs2 = self.ssa[j + 1]
            step.sname, step.bname = op2.inputs[1], op2.inputs[2]
            out_name, cur, j = op2.outputs[0], outs2[0], j + 1
        if self.sole_reader_is_next(j, cur, ("Relu",)):
            op2, ins2, outs2 = self.ssa[j + 1]
            step.relu = True
            out_name, cur, j = op2.outputs[0], outs2[0], j + 1
        out = self.new_blob(out_name, (N, Cout) + tuple(dims), 1)
        out.relu = step.relu
        out.needs_grad = True
        step.out = out
        step.outputs = [out]
        self.add_step(step)
        self.env[out_name] = out
        return j + 1

    def lower_Sum(self, i):
        op, ins, outs = self.ssa[i]
        assert len(op.inputs) == 2, "Sum of two blobs expected"
        a, b = self.get(op.inputs[0]), self.get(op.inputs[1])
        if tuple(a.shape) != tuple(b.shape):
            # Caffe2's Sum enforces equal shapes at run time; here the plan refuses (e.g. a batch-norm graph with
            # cfg.DILATIONS = 2: Conv3dBN drops the dilation but keeps its pads, as in the reference)
            raise ValueError("Sum(%s, %s): operand shapes differ, %s vs %s" % (op.inputs[0], op.inputs[1],
                                                                             tuple(a.shape), tuple(b.shape)))
        out_name = op.outputs[0]
        j = i
        relu = False
        if j + 1 < len(self.ssa):
            op2, ins2, outs2 = self.ssa[j + 1]
            if op2.type == "Relu" and outs[0] in ins2 and (self.readers.get(outs[0], 0) == 1 or op2.outputs[0] == op2.inputs[0]):
                relu = True
                out_name, j = op2.outputs[0], j + 1
        # fold into the epilogue of the conv that produced one operand (first operand preferred)
        for cand, other, ref in ((a, b, ins[0]), (b, a, ins[1])):
            st = cand.producer
            if (isinstance(st, ConvStep) and cand is st.out and not st.relu and st.residual is None
                    and self.readers.get(ref, 0) == 1 and cand.shape == other.shape):
                self.steps.remove(st)
                cand.dead = True        # its storage is never allocated: the fused output replaces it
                st.residual = other
                st.relu = relu
                st.inputs = [st.x, other]
                out = self.new_blob(out_name, cand.shape, 1)
                out.relu = relu
                out.needs_grad = True
                st.out = out
                st.outputs = [out]
                self.add_step(st)
                self.env[out_name] = out
                return j + 1
        out = self.new_blob(out_name, a.shape, a.caxis)
        out.relu = relu
        out.needs_grad = a.needs_grad or b.needs_grad
        self.add_step(AddStep(self.eng, a, b, out, relu))
        self.env[out_name] = out
        return j + 1

    def lower_Relu(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        out = self.new_blob(op.outputs[0], x.shape, x.caxis)
        out.relu = True
        out.needs_grad = x.needs_grad
        self.add_step(ReluStep(self.eng, x, out))
        self.env[op.outputs[0]] = out
        return i + 1

    def lower_SpatialBN(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        if x.caxis != 1:
            raise NotImplementedError("SpatialBN input %s must be a channels-last blob" % x.name)
        out = self.new_blob(op.outputs[0], x.shape, x.caxis)
        out.needs_grad = True
        a = op.args
        self.add_step(BNStep(self.eng, x, out, op.inputs[1], op.inputs[2], op.inputs[3], op.inputs[4],
                             a["epsilon"], a["momentum"], a["is_test"]))
        self.env[op.outputs[0]] = out
        return i + 1

    def _pool(self, i, is_max):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        a = op.args
        k, s, p = list(a["kernels"]), list(a["strides"]), list(a["pads"])
        if len(x.shape) != 5 or x.caxis != 1:
            raise NotImplementedError("pool input %s must be a 5-d channels-last blob" % x.name)
        p = p[:3]
        N, Cc, T, H, W = x.shape
        dims = [(n_ + 2 * pp - kk) // ss + 1 for n_, kk, ss, pp in zip((T, H, W), k, s, p)]
        out = self.new_blob(op.outputs[0], (N, Cc) + tuple(dims), 1)
        out.needs_grad = x.needs_grad
        self.add_step(PoolStep(self.eng, x, out, k, s, p, is_max))
        self.env[op.outputs[0]] = out
        return i + 1

    def lower_MaxPool(self, i):
        return self._pool(i, True)

    def lower_AveragePool(self, i):
        return self._pool(i, False)

    # -- views -----------------------------------------------------------------------------------
    def lower_Transpose(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        axes = list(op.args["axes"])
        keep_old = [ax for ax in range(len(x.shape)) if ax != x.caxis]
        keep_new = [ax for ax in axes if ax != x.caxis]
        if keep_old != keep_new:
            raise NotImplementedError("Transpose %r of %s moves data (not a view in channels-last)" % (axes, x.name))
        shape = tuple(x.shape[ax] for ax in axes)
        v = x.view(op.outputs[0], shape, axes.index(x.caxis))
        self.env[op.outputs[0]] = v
        return i + 1

    def lower_Reshape(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        if len(op.inputs) == 2:
            shape = list(self.shape_blobs[op.inputs[1]])
        else:
            shape = list(op.args["shape"])
        if -1 in shape:
            known = _prod([s for s in shape if s != -1])
            shape[shape.index(-1)] = x.numel // known
        assert _prod(shape) == x.numel, "Reshape %s: %r -> %r" % (x.name, x.shape, shape)
        Cc = x.C
        cands = [ax for ax, s in enumerate(shape) if s == Cc]
        if not cands:
            raise NotImplementedError("Reshape of %s loses the channel axis" % x.name)
        if x.caxis in cands and len(shape) == len(x.shape):
            ca = x.caxis
        elif 1 in cands:
            ca = 1
        else:
            ca = cands[-1]
        # rows before the channel axis must regroup without crossing it: in channels-last storage
        # this is always a pure reinterpretation of the position index.
        v = x.view(op.outputs[0], shape, ca)
        self.env[op.outputs[0]] = v
        if len(op.outputs) > 1:
            self.shape_blobs[op.outputs[1]] = tuple(x.shape)
        return i + 1

    def lower_Squeeze(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        dims = sorted(op.args["dims"])
        assert all(x.shape[d_] == 1 for d_ in dims) and x.caxis not in dims
        shape = [s for ax, s in enumerate(x.shape) if ax not in dims]
        ca = x.caxis - sum(1 for d_ in dims if d_ < x.caxis)
        self.env[op.outputs[0]] = x.view(op.outputs[0], shape, ca)
        return i + 1

    # -- attention -------------------------------------------------------------------------------
    def lower_BatchMatMul(self, i):
        op, ins, outs = self.ssa[i]
        if not op.args.get("trans_a"):
            raise NotImplementedError("BatchMatMul outside the theta.phi / softmax / g pattern: %r" % op)
        theta, phi = self.get(op.inputs[0]), self.get(op.inputs[1])
        j = i + 1
        scale = 1.0
        dot = False
        if self.ssa[j][0].type == "ConstantFill":
            # dot-product variant (NONLOCAL.USE_SOFTMAX False, nonlocal_helper.py:107-119): ones -> ReduceBackSum (the
            # number of keys) -> zeros + broadcast Add -> StopGradient -> Div, i.e. p = affinity / L2
            seq = [self.ssa[j + k][0] for k in range(6)]
            aff = op.outputs[0]
            assert [o.type for o in seq] == ["ConstantFill", "ReduceBackSum", "ConstantFill", "Add", "StopGradient", "Div"] and \
                float(seq[0].args.get("value")) == 1.0 and float(seq[2].args.get("value")) == 0.0 and \
                seq[1].inputs == seq[0].outputs and seq[3].inputs == [seq[2].outputs[0], seq[1].outputs[0]] and \
                seq[5].inputs == [aff, seq[3].outputs[0]], "unexpected operators behind the affinity: %r" % (seq,)
            dot = True
            prob_name = seq[5].outputs[0]
            j += 6
        else:
            if self.ssa[j][0].type == "Scale":
                scale = float(self.ssa[j][0].args["scale"])
                j += 1
            sm = self.ssa[j][0]
            assert sm.type == "Softmax" and sm.args.get("axis") == 2, "expected Softmax(axis=2) after the affinity"
            prob_name = sm.outputs[0]
            j += 1
        mm = self.ssa[j][0]
        assert mm.type == "BatchMatMul" and mm.args.get("trans_b"), "expected BatchMatMul(trans_b=1)"
        g = self.get(mm.inputs[0])
        for t in (theta, phi, g):
            assert len(t.shape) == 3 and t.caxis == 1, "attention operand %s must be (B, C, L)" % t.name
        B, Ci, L1 = theta.shape
        L2 = phi.shape[2]
        single = (L1 == 1)
        if not single and self.eng.mix and self.eng.MIX_NL_F32:
            # "mix": the softmax Jacobian cancels the common part of dP, and the weight gradients of theta / phi sum those
            # differences over all positions -- the ill-conditioned tensors of the model (the worst of every path).  Their
            # gradients stay in fp32 and their weight gradients use split-bf16 products (ConvStep.bwd_f32).
            for t in (theta, phi, g):
                if isinstance(t.root.producer, ConvStep) and t.root.producer.out is t.root:
                    t.root.grad_f32 = True
        elif not single and self.eng.btdtype == torch.float16:
            # fp16: d theta and d phi are about 1 / L2 of a normal activation gradient (they pass the softmax
            # Jacobian): keep them times a power of two so that they stay in the fp16 normal range
            for t in (theta, phi):
                if isinstance(t.root.producer, ConvStep):
                    t.root.grad_scale = float(1 << max(L2 - 1, 1).bit_length())
        prob = self.new_blob(prob_name, (B, L1, L2), 2, "f32" if single else "act")
        out = self.new_blob(mm.outputs[0], (B, Ci, L1), 1)
        out.needs_grad = True
        if not single and self.eng.mix and self.eng.MIX_NL_F32:
            # ... and so does the gradient of the attention output: dP = dY . g^T feeds the softmax Jacobian, whose
            # cancellation amplifies the 11-bit rounding of a stored fp16 dY into the theta / phi weight gradients (they
            # moved between 2e-4 and 9e-4 from one rounding pattern to the next).  The `out` conv's fp16 DGRAD writes its
            # fp32 accumulators (ConvStep.dx_f32).
            out.root.grad_f32 = True
        assert not (dot and single), "the dot-product variant belongs to the space-time non-local block"
        self.add_step(AttentionStep(self.eng, theta, phi, g, prob, out, scale, dot=dot))
        self.env[prob_name] = prob
        self.env[mm.outputs[0]] = out
        return j + 1

    # -- head ------------------------------------------------------------------------------------
    def lower_LayerNorm(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        assert op.args.get("axis", 1) == 1
        out = self.new_blob(op.outputs[0], x.shape, x.caxis)
        out.needs_grad = x.needs_grad
        step = LayerNormStep(self.eng, x, out, float(op.args.get("epsilon", 1e-5)))
        j = i
        self.add_step(step)
        self.env[op.outputs[0]] = out
        return j + 1

    def lower_Dropout(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        if op.args.get("is_test"):
            self.env[op.outputs[0]] = x.view(op.outputs[0], x.shape, x.caxis)
            return i + 1
        out = self.new_blob(op.outputs[0], x.shape, x.caxis)
        out.needs_grad = x.needs_grad
        self.add_step(DropoutStep(self.eng, x, out, float(op.args["ratio"])))
        self.env[op.outputs[0]] = out
        return i + 1

    def lower_RoIAlign(self, i):
        op, ins, outs = self.ssa[i]
        feat, rois = self.get(op.inputs[0]), self.get(op.inputs[1])
        a = op.args
        assert a["pooled_w"] == a["pooled_h"] and a.get("sampling_ratio", 0) == 0
        res = a["pooled_w"]
        assert len(feat.shape) == 4 and feat.caxis == 1
        R = rois.shape[0]
        nxt = self.ssa[i + 1][0] if i + 1 < len(self.ssa) else None
        if res > 1:
            if not (nxt is not None and nxt.type == "MaxPool" and list(nxt.args["kernels"]) == [res, res]
                    and nxt.inputs[0] == op.outputs[0]):
                raise NotImplementedError("RoIAlign must be followed by the %dx%d MaxPool of the RoI head" % (res, res))
            out_name, j = nxt.outputs[0], i + 1
        else:
            out_name, j = op.outputs[0], i
        out = self.new_blob(out_name, (R, feat.shape[1], 1, 1), 1)
        out.needs_grad = feat.needs_grad
        self.add_step(RoiAlignMaxStep(self.eng, feat, rois, out, res, float(a["spatial_scale"])))
        self.env[out_name] = out
        return j + 1

    def lower_Concat(self, i):
        op, ins, outs = self.ssa[i]
        parts = [self.get(n) for n in op.inputs]
        assert op.args.get("axis", 1) == 1
        if len(parts) == 1:
            p = parts[0]
            self.env[op.outputs[0]] = p.view(op.outputs[0], p.shape, p.caxis)
            return i + 1
        for p in parts:
            assert p.caxis == 1 and p.rows == p.shape[0], "Concat parts must be (R, C, 1, 1, 1)"
        total = sum(p.C for p in parts)
        out = self.new_blob(op.outputs[0], (parts[0].shape[0], total) + tuple(parts[0].shape[2:]), 1)
        out.needs_grad = any(p.needs_grad for p in parts)
        self.add_step(ConcatStep(self.eng, parts, out))
        self.env[op.outputs[0]] = out
        return i + 1

    def lower_FC(self, i):
        op, ins, outs = self.ssa[i]
        x = self.get(op.inputs[0])
        wname, bname = op.inputs[1], op.inputs[2]
        cout = self.model.param_init_net.fills[wname].shape[0]
        out = self.new_blob(op.outputs[0], (x.shape[0], cout), 1, "f32")
        out.needs_grad = True
        self.add_step(FCStep(self.eng, x, out, wname, bname))
        self.env[op.outputs[0]] = out
        return i + 1

    def lower_Sigmoid(self, i):
        op, ins, outs = self.ssa[i]
        logits = self.get(op.inputs[0])
        prob = self.new_blob(op.outputs[0], logits.shape, 1, "f32")
        nxt = self.ssa[i + 1][0] if i + 1 < len(self.ssa) else None
        if nxt is not None and nxt.type == "SigmoidCrossEntropyLoss" and nxt.inputs[0] == op.inputs[0]:
            labels = self.get(nxt.inputs[1])
            loss = self.new_blob(nxt.outputs[0], (1,), 0, "f32")
            self.add_step(LossStep(self.eng, logits, labels, prob, loss, float(nxt.args["scale"])))
            self.env[nxt.outputs[0]] = loss
            self.env[op.outputs[0]] = prob
            return i + 2
        self.add_step(LossStep(self.eng, logits, None, prob, None, 1.0))
        self.env[op.outputs[0]] = prob
        return i + 1

    def lower_SoftmaxWithLoss(self, i):
        """[logits, labels] -> [prob, loss] (resnet_video.py:343-344); labels are class indices, one per row"""
        op, ins, outs = self.ssa[i]
        logits, labels = self.get(op.inputs[0]), self.get(op.inputs[1])
        assert labels.numel == logits.shape[0], "SoftmaxWithLoss: one label per row expected, got %r" % (labels.shape,)
        prob = self.new_blob(op.outputs[0], logits.shape, 1, "f32")
        loss = self.new_blob(op.outputs[1], (1,), 0, "f32")
        self.add_step(LossStep(self.eng, logits, labels, prob, loss, float(op.args["scale"]), kind="softmax"))
        self.env[op.outputs[0]] = prob
        self.env[op.outputs[1]] = loss
        return i + 1

    def lower_Softmax(self, i):
        """the test-mode head of single-label models (resnet_video.py:349-350); the softmaxes of the non-local
        blocks never get here (they are consumed by lower_BatchMatMul)"""
        op, ins, outs = self.ssa[i]
        logits = self.get(op.inputs[0])
        assert op.args.get("axis", 1) == 1 and len(logits.shape) == 2 and logits.kind == "f32", \
            "Softmax outside the classifier head: %r" % (op,)
        prob = self.new_blob(op.outputs[0], logits.shape, 1, "f32")
        self.add_step(LossStep(self.eng, logits, None, prob, None, 1.0, kind="softmax"))
        self.env[op.outputs[0]] = prob
        return i + 1

    def lower_SigmoidCrossEntropyLoss(self, i):
        op, ins, outs = self.ssa[i]
        logits, labels = self.get(op.inputs[0]), self.get(op.inputs[1])
        loss = self.new_blob(op.outputs[0], (1,), 0, "f32")
        self.add_step(LossStep(self.eng, logits, labels, None, loss, float(op.args["scale"])))
        self.env[op.outputs[0]] = loss
        return i + 1


# ================================================================================================
# engine
# ================================================================================================
class Engine(object):
    """One per-GPU replica: plan once, then forward()/backward()/allreduce()/sgd_step()."""

    def __init__(self, model, dtype="bf16", device=None, base_seed=None, dry_run=False, debug_roi=False,
                 share_params_with=None, side_stream=True, loss_scale=None):
        """share_params_with: another Engine of the same scope (the train net's, when this is the test / val
        net of the same process).  Caffe2 nets of one workspace share their parameter BLOBS
        (tools/train_net.py builds train_model and test_model in one workspace and evaluates the weights
        being trained); parameters with the same name and shape then alias the owner's storage here, and the
        MFMA operand copies of this engine are rebuilt whenever the owner's parameters have changed."""
        hip.lib()   # fail loudly if the native library is missing
        self.debug_roi = bool(debug_roi)
        self.use_side_stream = bool(side_stream)   # parameter-gradient kernels on a second HIP stream
        self.param_owner = share_params_with
        # [version] of the parameter storage, shared by the engines that alias it
        self._pstate = share_params_with._pstate if share_params_with is not None else [0]
        self._operand_version = -1
        self.dry_run = bool(dry_run)   # plan on the 'meta' device: shapes/fusion/bytes only, nothing runs
        if not self.dry_run and not torch.cuda.is_available():
            raise hip.VlfbError("vlfb.engine needs a GPU: there is no CPU fallback for the hot path")
        self.model = model
        self.tdtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16, "fp32": torch.float32,
                       "f32": torch.float32, "split": torch.float32, "mix": torch.float32}[dtype]
        self.code = hip.dtype_code(self.tdtype)
        # "mix": the FORWARD of "split" (fp32 storage, three bf16 products per product: activations within ~1e-5 of fp64, so
        # the ReLU / max-pool decisions are the parity path's) and the BACKWARD of "fp16" (fp16 gradient storage with the
        # static loss scale, one fp16 MFMA per product), joined by fp16 COPIES of the forward values the backward reads
        # (WGRAD operands, ReLU masks, attention operands: Blob.half, written by the producing conv epilogue or a copy
        # pass).  MIX_W2: the DGRAD weight operand as two fp16 terms (the rounding of W to 11 bits is the largest single
        # error class of an fp16 backward, scratch/r4/emu_hybrid.py).
        self.mix = dtype == "mix"
        self.btdtype = torch.float16 if self.mix else self.tdtype      # element type of the backward pass
        self.bcode = hip.dtype_code(self.btdtype)
        # "split": fp32 storage everywhere (as "fp32"), every contraction on the bf16 matrix cores with the operands
        # expanded into bf16 terms (csrc/vlfb_gemm_split.hip): six MFMAs per product forward (fp32-grade: ReLU / max-pool
        # decisions must match the oracle's), three backward.  The parity-grade path at several times the fp32-MFMA rate.
        self.split = dtype in ("split", "mix")
        self.math_fwd = self.SPLIT_MATH[0] if self.split else hip.MATH_NATIVE
        self.math_bwd = self.SPLIT_MATH[1] if (self.split and not self.mix) else hip.MATH_NATIVE
        if self.mix and self.math_fwd != hip.MATH_BF16X3:
            raise hip.VlfbError("the 'mix' dtype uses three-term forward products (VLFB_SPLIT_MATH=3,3)")
        # format of the MFMA weight operand copies
        self.wcode = (hip.MIX_W2 if self.MIX_W2 else hip.MIX) if self.mix else hip.SPLIT if self.split else self.code
        self.esize = 4 if self.tdtype == torch.float32 else 2
        self.besize = 4 if self.btdtype == torch.float32 else 2
        # fp16 storage (v_mfma_f32_16x16x32_f16; BASELINE.json configs[4]): 10 mantissa bits instead of bf16's 7,
        # but gradients of 1e-6 fall below the fp16 normal range, so the loss gradient is scaled by a power of two
        # (exact) and every parameter gradient carries that factor until the solver divides it out again
        # (lr / S, weight decay * S: lr/S * (S g + S wd p) = lr * (g + wd p)).  1 on the other paths.
        self.auto_loss_scale = loss_scale is None and self.btdtype == torch.float16   # set by LossStep.setup from the shapes
        self.plan_clips, self.plan_roi_rows = None, False
        self.loss_scale = float(loss_scale if loss_scale is not None else 1.0)
        self.device = torch.device("meta") if self.dry_run else torch.device(device or ("cuda:%d" % dist.local_rank()))
        self.train = bool(model.train and not model.force_fw_only and model.loss_blob is not None)
        self.base_seed = int(cfg.RNG_SEED if base_seed is None else base_seed)
        self.iteration = 0
        self.replica = dist.rank()    # folded into the dropout seeds: replicas draw independent masks
        self.trainable = [p for p in model.params if p in model.param_to_grad] if self.train else []
        self._trainable_set = set(self.trainable)
        self.steps = None
        self.workspace = None
        self._ws_bytes = 0
        self._sf32 = 0
        self._sact = 0
        self._spl = 0
        self.lr = float(model.current_lr)
        self.comm = None
        self.side = None
        self.solver_stream = None      # all-reduce hand-off + per-bucket solver (third HIP stream)
        self.solver_dirty = False
        self._eager_lr = None          # learning rate of the step whose buckets are solved during backward
        self._eager_next = 0
        self._eager_done = False
        self.side_dirty = False
        self._dev_scalars = False      # True while a step is being captured: per-iteration scalars come from memory
        self._scalars_dev = None
        self._graph = None
        self._graph_key = None
        self._graph_stream = None
        self._eager_steps = 0
        self._trace = None             # recorded step (STEP_TRACE)
        self._wq = []                  # parameter-gradient launches waiting for their lag (WGRAD_LAG)
        self._bwd_index = 0
        self._trace_key = None
        model.engine = self

    # (forward, backward) math of the "split" dtype: hip.MATH_BF16X6 / MATH_BF16X3 (see __init__)
    # Default: three products (hh hm mh, ~2^-17 each) in both directions.  Six forward products (VLFB_SPLIT_MATH=6,3:
    # ~2^-24, forward activations 1e-6 instead of 7e-6 off the fp64 oracle) cut the ReLU / max-pool decisions that differ
    # from the oracle's at full size from 307 to 25 units but NOT the gradient error: a single differing unit already moves
    # the raw comparison to ~1e-3, both settings measure the same raw table (median 1.5e-3 / 1.6e-3)
    # and 1.9e-5 / 3.8e-5 max on identical decisions -- at 151 against 171 clips/s on the same box.
    SPLIT_MATH = tuple(int(x) for x in os.environ.get("VLFB_SPLIT_MATH", "3,3").split(","))
    if len(SPLIT_MATH) != 2 or SPLIT_MATH[0] not in (3, 6) or SPLIT_MATH[1] != 3:
        raise ValueError("VLFB_SPLIT_MATH must be '3,3' or '6,3' (forward, backward terms), got %r"
                         % os.environ.get("VLFB_SPLIT_MATH"))
    # "mix" dtype: DGRAD contracts the fp16 gradient with TWO fp16 terms of the weight (22 bits; hip.MIX_W2) instead of one
    MIX_W2 = os.environ.get("VLFB_MIX_W2", "1") != "0"
    # ... as hip.MATH_F16W2 where the doubled-tap form would run on the 128-row kernel: one gradient tile in LDS per pair of
    # weight tiles (0 = every two-term DGRAD in the doubled-tap form, the round-5 arrangement; A/B switch)
    MIX_W2I = os.environ.get("VLFB_MIX_W2I", "1") != "0"
    # "mix": the NT launch that produces an fp32 gradient read next by a 16-bit launch (the attention output and theta of a
    # non-local block) also writes it rounded to fp16 (vlfb_conv_args.O_lo) instead of a cast pass in front of the reader
    GRAD_HALF_COPY = os.environ.get("VLFB_GRAD_HALF_COPY", "1") != "0"
    # "mix" dtype: gradients of theta / phi / g of the non-local blocks in fp32, their weight gradients and the dP product of
    # the attention backward as split-bf16 products, the softmax backward on the fp32 probabilities
    MIX_NL_F32 = os.environ.get("VLFB_MIX_NL_F32", "1") != "0"
    # "mix" dtype: the residual-stream gradient as two fp16 terms (GradSlot.two_term)
    MIX_TRUNK2 = os.environ.get("VLFB_MIX_TRUNK2", "1") != "0"
    # "split" dtype: conv epilogues also write the bf16 term planes of their outputs / input gradients, and the DGRAD / WGRAD
    # launches that find their operands in that form read them without expanding (ConvStep.bwd); tensors with more than
    # PLANES_MAX_NUMEL elements (the wide res2 / stem tensors: a second copy costs more HBM time than it saves) stay fp32-only
    PLANES = True
    PLANES_MAX_NUMEL = 52 << 20
    FPROP_PLANES = True         # three-term forward products: FPROP launches read existing input planes as well
    PLANES_SCOPE = "gathered"   # "all": planes around every conv, not only the gathered ones (measured)
    STEM_PLANES = True          # conv1: clip and output gradient through a split pass, FPROP / WGRAD on planes
    # bias gradients of the convs that carry a bias come out of their WGRAD launch (False: a column-sum pass per conv)
    FUSE_BIAS_GRAD = True

    # ---- side stream for parameter gradients ---------------------------------------------------
    class _Side(object):
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            eng = self.eng
            if eng.side is None:
                return self
            ev = eng.record_event()           # everything enqueued so far on the main stream
            eng.wait_event(eng.side, ev)
            self.ctx = torch.cuda.stream(eng.side)
            self.ctx.__enter__()
            eng.side_dirty = True
            return self

        def __exit__(self, *exc):
            if self.eng.side is not None:
                self.ctx.__exit__(*exc)
            return False

    def on_side_stream(self):
        return Engine._Side(self)

    # Fraction of the CUs the parameter-gradient stream may use (hipExtStreamCreateWithCUMask; 1.0 = an ordinary stream).
    # The idea: the main stream (forward, dgrad chain) is the critical path of a step and the parameter gradients finish
    # with ~6 ms of slack, so confining them to a part of the chip would leave the rest to the chain alone.  MEASURED
    # (round 4, one box, 8 clips): 1.0 -> 455.1 / 452.9 clips/s fp16; 0.875 -> 277.0, 0.75 -> 281.0, 0.625 -> 273.9,
    # 0.5 -> 273.2, 0.375 -> 245.8; mix 240.3 -> 173.1 (0.75) / 167.7 (0.5).  A masked queue costs far more than it frees
    # (the whole-row / streaming kernels are sized for one workgroup per CU of the full chip, and the two queues stop
    # overlapping the way unmasked ones do).  Off; kept as a switch so the measurement can be repeated.
    SIDE_CU_FRACTION = float(os.environ.get("VLFB_SIDE_CU_FRACTION", "1.0"))

    def _make_side_stream(self):
        frac = float(self.SIDE_CU_FRACTION)
        if frac >= 1.0:
            return torch.cuda.Stream(device=self.device)
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        # groups of 8 consecutive mask bits are taken or left whole, spread evenly: whatever the bit -> (XCD, CU) map is
        # (round-robin over the 8 XCDs or XCD-major), every XCD keeps the same share
        ngroups = (ncu + 7) // 8
        take = max(1, int(round(frac * ngroups)))
        words = [0] * ((ncu + 31) // 32)
        for k in range(take):
            gidx = (k * ngroups) // take
            for b in range(8 * gidx, min(8 * gidx + 8, ncu)):
                words[b // 32] |= 1 << (b % 32)
        rt = C.CDLL("libamdhip64.so")
        stream = C.c_void_p()
        mask = (C.c_uint32 * len(words))(*words)
        torch.cuda.set_device(self.device)
        rc = rt.hipExtStreamCreateWithCUMask(C.byref(stream), C.c_uint32(len(words)), mask)
        if rc != 0:
            raise hip.VlfbError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
        self._side_cu_mask = words
        return torch.cuda.ExternalStream(stream.value, device=self.device)

    # Parameter gradients are leaves of the backward graph: when they run does not matter as long as it is after their
    # output gradient exists (event) and before the all-reduce / solver.  WGRAD_LAG = k issues the parameter gradients of
    # backward step i only after the dgrad chain has advanced to step i + k, so that the (MFMA-bound) res5 / res4 wgrads run
    # beside the (HBM-bound) dgrads of the stages below instead of beside their own stage's dgrads.  0: issue at once.
    WGRAD_LAG = 0

    def issue_param_grads(self, fn):
        if self.side is None or not self.WGRAD_LAG or self.comm is not None or self._eager_lr is not None:
            with self.on_side_stream():
                fn()
            return
        self._wq.append((self._bwd_index, self.record_event(), fn))

    def _flush_param_grads(self, upto):
        while self._wq and self._wq[0][0] <= upto:
            _, ev, fn = self._wq.pop(0)
            self.wait_event(self.side, ev)
            with torch.cuda.stream(self.side):
                fn()
            self.side_dirty = True

    # stream-ordering primitives of a step: like the kernel launches (hip.call) they append themselves to hip.TRACE
    # while a step is being recorded, so that a replay re-issues the same edges between the same streams
    def record_event(self, stream=None):
        ev = torch.cuda.Event()
        if stream is None:
            stream = torch.cuda.current_stream()
        ev.record(stream)
        if hip.tracing() is not None:
            hip.tracing().append((ev.record, (stream,), "event record"))
        return ev

    def wait_event(self, stream, ev):
        stream.wait_event(ev)
        if hip.tracing() is not None:
            hip.tracing().append((stream.wait_event, (ev,), "stream wait"))

    def traced(self, fn, name):
        """run a HOST-side action of a step that is not a library call or a stream edge -- the gradient all-reduces torch issues
        (GradComm), their bookkeeping -- and, while a step is being recorded, append it to the call list so that a replayed
        step re-issues it at the same point between the same kernel launches (Engine.STEP_TRACE on data-parallel steps)"""
        fn()
        rec = hip.tracing()
        if rec is not None:
            def call():
                fn()
                return 0
            rec.append((call, (), name))

    def _issue_reductions(self, stream, i, wait):
        """the all-reduces of the gradient buckets that are final after backward step i, ordered behind `stream` (ProcessGroupNCCL
        = RCCL orders a collective behind the stream that is current at the call)"""
        ctx = torch.cuda.stream(stream) if stream is not None else None
        if ctx is not None:
            ctx.__enter__()
        try:
            works = self.comm.after_step(i)
            if wait:
                for w in works:
                    w.wait()
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def wait_stream(self, other):
        cur = torch.cuda.current_stream()
        cur.wait_stream(other)
        if hip.tracing() is not None:
            hip.tracing().append((cur.wait_stream, (other,), "stream join"))

    def join_side_stream(self):
        """make the main stream wait for every parameter-gradient kernel issued so far"""
        if self.side is not None and self.side_dirty:
            self.wait_stream(self.side)
            self.side_dirty = False
        if self.solver_stream is not None and self.solver_dirty:
            self.wait_stream(self.solver_stream)
            self.solver_dirty = False

    # ---- bookkeeping used by steps ------------------------------------------------------------
    def is_trainable(self, name):
        return name in self._trainable_set

    def want_half(self, blob):
        """the backward pass reads the VALUES of this blob: on the "mix" path it needs their fp16 copy"""
        if self.mix and self.train and blob.root.kind == "act":
            blob.root.need_half = True

    def need_workspace(self, nbytes):
        self._ws_bytes = max(self._ws_bytes, int(nbytes))

    def need_scratch_f32(self, n):
        self._sf32 = max(self._sf32, int(n))

    def need_scratch_act(self, n):
        self._sact = max(self._sact, int(n))

    def need_scratch_planes(self, n):
        self._spl = max(self._spl, int(n))

    def need_join_scratch(self, n):
        self._sjoin = max(getattr(self, "_sjoin", 0), int(n))

    def join_scratch(self, n):
        """fp32 scratch of the parameter-gradient stream (ConvStep._x_f32)"""
        assert n <= self._scratch_join.numel()
        return self._scratch_join[:n]

    def scratch_planes(self, n):
        assert n <= self._scratch_pl.numel()
        return self._scratch_pl[:n]

    def scratch_f32(self, n):
        assert n <= self._scratch_f32.numel()
        return self._scratch_f32[:n]

    def scratch_act(self, n, dtype=None):
        if dtype is not None and dtype != self._scratch_act.dtype:
            assert dtype == torch.float32
            return self.scratch_f32(n)
        assert n <= self._scratch_act.numel(), "scratch too small: %d > %d" % (n, self._scratch_act.numel())
        return self._scratch_act[:n]

    def kernel_shape(self, wname):
        """shape of a conv weight in kernel order [Cout][kt][kh][kw][Cin] (stem: [64][kt][kh][8][4])"""
        co, ci, kt, kh, kw = self.model.param_init_net.fills[wname].shape
        if ci == 3:
            return (co, kt, kh, 8, 4)
        return (co, kt, kh, kw, ci)

    def param_tensor(self, name):
        return self.param_views[name]

    def grad_tensor(self, name):
        return self.grad_views[name]

    # ---- planning -----------------------------------------------------------------------------
    def plan(self, input_shapes):
        """input_shapes: {blob name: shape in the REFERENCE layout} for data/labels/proposals/lfb."""
        if not self.dry_run:
            torch.cuda.set_device(self.device)
        self._trace = self._graph = None       # a recorded / captured step belongs to the buffers of ONE plan
        self._eager_steps = 0
        data = [tuple(v) for k, v in input_shapes.items() if str(k).startswith("data")]
        self.plan_clips = int(data[0][0]) if data else None     # clips per GPU of this plan (rank-independent)
        self.plan_roi_rows = any(str(k).startswith("proposals") for k in input_shapes)   # head rows are RoIs, not clips
        low = Lowering(self, self.model, OrderedDict(input_shapes))
        self.steps = low.run()
        self._plan_head_f32()
        self.env = low.env
        self.all_blobs = list(low.blobs.values())
        self._plan_pairs()
        self._plan_params()
        self._analyse_grads()
        for st in self.steps:
            st.setup()
        self._allocate()
        if not self.dry_run:
            self.refresh_operands(all_params=True)
            if self.train and self.use_side_stream:
                self.side = self._make_side_stream()
                self.solver_stream = torch.cuda.Stream(device=self.device)
        if self.train:
            self._plan_solver_buckets()
        self._plan_forward_branches()
        self._plan_half_copies()
        self._drop_steps = [st for st in self.steps if isinstance(st, DropoutStep)]
        for k, st in enumerate(self._drop_steps):
            st.seed_slot = 1 + k
        if not self.dry_run:
            self._scalars_dev = torch.zeros(1 + len(self._drop_steps), device=self.device, dtype=torch.int64)
        return self

    def _plan_head_f32(self):
        """"mix" dtype: fp32 gradients on the DIRECT path of the head -- classifier, dropout, concat, RoIAlign + max, the
        temporal / global average pool -- with one rounding to fp16 where the gradient enters res5 (PoolStep.bwd).  Those
        were five fp16 storages in series whose error is common to EVERY backbone gradient (measured at full size, round 5:
        identical-decisions median 4.0e-4 -> 3.0e-4, second-worst tensor 9.6e-4 -> 6.9e-4, at the same clips/s).  Marks the
        blobs whose gradient stays in fp32 (Blob.grad_f32): walk back from the classifier's input through dropout / concat /
        RoIAlign + max; the output of the average pool that reads res5 is the last one (its PoolStep casts down).  A blob
        is only marked when every step that contributes to its gradient can write fp32."""
        self.head_f32, self.head_f32_fbo = [], []
        if not (self.train and self.mix):
            return
        fcs = [st for st in self.steps if isinstance(st, FCStep)]
        if len(fcs) != 1:
            return
        writers = (FCStep, DropoutStep, ConcatStep, RoiAlignMaxStep, ConvStep)
        consumers = {}
        for st in self.steps:
            for b in st.inputs:
                consumers.setdefault(id(b.root), []).append(st)
        work, seen = [fcs[0].x.root], set()
        while work:
            b = work.pop()
            if id(b) in seen or b.kind != "act" or b.relu:
                continue
            seen.add(id(b))
            p = b.producer
            if isinstance(p, DropoutStep):
                nxt = [p.x.root]
            elif isinstance(p, ConcatStep):
                nxt = [q.root for q in p.parts]
            elif isinstance(p, RoiAlignMaxStep):
                nxt = [p.feat.root]
            elif isinstance(p, PoolStep) and not p.is_max:
                nxt = []
            else:
                continue
            if not all(isinstance(c, writers) for c in consumers.get(id(b), [])):
                continue
            if any(isinstance(c, ConvStep) and (c.group != 1 or c.x.root is not b) for c in consumers.get(id(b), [])):
                continue
            b.grad_f32 = True
            self.head_f32.append(b.name)
            work += nxt
        self._plan_fbo_f32(consumers)

    def _plan_fbo_f32(self, consumers):
        """... and the FBO branch of the head (lfb_helper.py:170-338: reduc conv, lfb_1x1, per layer theta / phi / g, the
        one-query attention, LayerNorm, ReLU, out conv, dropouts, the Sum) between the fp32 concat gradient and the fp32
        RoI features: a dozen fp16 storages in series that left a 2.2e-4 error on the gradient of `box_pooled` -- common
        to EVERY backbone gradient (scratch/r5/diag_mix.py, full-size clip) -- for < 2 % of the step's FLOPs.  Its steps
        run their backward in fp32 (ConvStep.bwd_split: the split-bf16 DGRAD / WGRAD of the `split` dtype; the other
        steps: their fp32 kernels on the fp32 forward values).  All or nothing: the branch is only marked when every blob
        between the concat and an fp32 slot / a fed blob is produced AND consumed by steps that can do that."""
        self.head_f32_fbo = []
        f32_ok = lambda st: isinstance(st, (AddStep, DropoutStep, ReluStep, LayerNormStep, ConcatStep)) or \
            (isinstance(st, AttentionStep) and st.theta.shape[2] == 1) or \
            (isinstance(st, ConvStep) and st.group == 1 and not st.stem)    # (incl. an out conv fused with the Sum)
        for cat in [st for st in self.steps if isinstance(st, ConcatStep) and st.out.root.grad_f32]:
            for part in cat.parts:
                cand, work, ok = {}, [part.root], True
                while work and ok:
                    b = work.pop()
                    if id(b) in cand or b.grad_f32 or b.producer is None:
                        continue                     # an fp32 slot already (box_pooled) or a fed blob (the bank)
                    st = b.producer
                    if b.kind != "act" or not f32_ok(st) or b.grad_scale != 1.0:
                        ok = False
                        break
                    cand[id(b)] = b
                    work += [i.root for i in st.inputs]
                for b in cand.values():                # every consumer writes into a slot of the set (or the concat)
                    for c in consumers.get(id(b), []):
                        outs_ok = all(id(o.root) in cand or o.root.grad_f32 for o in c.outputs)
                        ok = ok and f32_ok(c) and outs_ok
                if ok:
                    for b in cand.values():
                        b.grad_f32 = True
                        self.head_f32_fbo.append(b.name)

    # "mix" dtype: forward activations of the trunk as two fp16 planes, forward convs on them with three fp16 MFMAs per product
    # (hip.MATH_F16X3) -- VLFB_MIX_PAIR=0: the round-5 form (fp32 storage, split-bf16 products, separate fp16 copies)
    MIX_PAIR = os.environ.get("VLFB_MIX_PAIR", "1") != "0"
    PAIR_ATTENTION_OUT = os.environ.get("VLFB_PAIR_ATTN_OUT", "1") != "0"      # (A/B switch)
    PAIR_SCORES = os.environ.get("VLFB_PAIR_SCORES", "1") != "0"                # (A/B switch: theta / phi as two planes)

    def _plan_pairs(self):
        """Which activations are stored as two fp16 planes (Blob.pair).  A blob qualifies when a conv or a max pool over a
        two-plane blob produces it and EVERY consumer can read the format: a conv (as its input: plain rows or whole 32-channel
        k-tiles; or as the residual of its epilogue), a max pool (whose output then qualifies in turn), an average pool (fp32
        out: the head).  The formats of a conv's residual and output, and of a max pool's input and output, are tied; a conv
        with a two-plane input and an fp32 output takes no residual.  Everything else -- theta / phi / g and the internals of a
        non-local block, the head, the FBO branch -- stays fp32 and is read by the kernels it was read by before."""
        self.pair_fwd = bool(self.mix and self.MIX_PAIR)
        self.pair_blobs = []
        if not self.pair_fwd:
            return
        consumers = {}
        for st in self.steps:
            for b in st.inputs:
                consumers.setdefault(id(b.root), []).append((st, b))
        cand = {}
        for b in self.all_blobs:
            # (an fp32 gradient slot masks with the fp32 VALUES of a post-ReLU blob: those stay fp32)
            if b.root is not b or b.kind != "act" or getattr(b, "dead", False) or (b.grad_f32 and b.relu) or b.C % 8 or \
                    getattr(b, "is_input", False):
                continue
            st = b.producer
            if (isinstance(st, ConvStep) and st.group == 1 and st.out is b and (not b.grad_f32 or self.PAIR_SCORES)) or \
                    (isinstance(st, PoolStep) and st.is_max):
                cand[id(b)] = b            # (an fp32-gradient conv output -- theta / phi / g -- only survives as theta / phi below)
            elif isinstance(st, AttentionStep) and st.theta.shape[2] > 1 and not st.dot and st.out is b and self.PAIR_ATTENTION_OUT:
                # the attention output of a non-local block: its P . g product writes two planes (split-bf16 launch with a
                # two-plane output), so the `out` conv behind it is a two-plane conv as well
                cand[id(b)] = b

        def conv_reads(st, b):
            if st.group != 1:
                return False
            if st.residual is not None and st.residual.root is b and st.x.root is not b:
                return True
            plain = tuple(st.k) == (1, 1, 1) and tuple(st.s) == (1, 1, 1) and tuple(st.p) == (0, 0, 0)
            return st.x.root is b and not st.stem and (b.C % 32 == 0 or (plain and b.C % 8 == 0))

        changed = True
        while changed:
            changed = False
            for b in list(cand.values()):
                ok = True
                p = b.producer
                if isinstance(p, PoolStep):
                    ok = id(p.x.root) in cand
                elif isinstance(p, ConvStep) and b.grad_f32:
                    # a conv output with an fp32 gradient slot: only theta / phi of a space-time non-local block (the FBO
                    # head's convs run the split-bf16 backward on fp32 VALUES: they keep fp32 storage)
                    ok = all(isinstance(st, AttentionStep) for st, _ in consumers.get(id(b), []))
                for st, v in consumers.get(id(b), []):
                    if isinstance(st, ConvStep):
                        ok = ok and conv_reads(st, b)
                        if st.residual is not None and st.residual.root is b:        # residual and output: one format
                            ok = ok and id(st.out.root) in cand
                        if st.x.root is b and st.residual is not None and st.residual.root is not b:
                            ok = ok and id(st.residual.root) in cand       # (two-plane input + fp32 residual: no kernel)
                    elif isinstance(st, PoolStep):
                        ok = ok and (not st.is_max or id(st.out.root) in cand) and v.caxis == 1 and len(v.shape) == 5
                    elif isinstance(st, AttentionStep) and st.theta.shape[2] > 1 and not st.dot and self.PAIR_SCORES and \
                            (v.root is st.theta.root or v.root is st.phi.root) and v.root is not st.g.root:
                        # the scores product theta x phi^T of a non-local block as a batched two-plane product: both or neither
                        ok = ok and id(st.theta.root) in cand and id(st.phi.root) in cand
                    else:
                        ok = False
                if isinstance(p, ConvStep) and p.residual is not None:
                    ok = ok and id(p.residual.root) in cand
                if not ok:
                    del cand[id(b)]
                    changed = True
        for b in cand.values():
            b.pair = True
            self.pair_blobs.append(b.name)

    def _plan_params(self):
        """flat fp32 buckets: trainables ordered by backward completion; frozen ones separately"""
        fills = self.model.param_init_net.fills
        order = []
        seen = set()
        self.param_step = {}
        for st in reversed(self.steps):
            names = []
            if isinstance(st, ConvStep):
                names = [st.wname, st.cbname]
            elif isinstance(st, FCStep):
                names = [st.wname, st.bname]
            elif isinstance(st, BNStep):
                names = [st.sname, st.bname]
            for n in names:
                if n and self.is_trainable(n) and n not in seen:
                    seen.add(n)
                    order.append(n)
                    self.param_step[n] = st
        missing = [p for p in self.trainable if p not in seen]
        assert not missing, "trainable params without a producing step: %r" % missing
        self.train_order = order
        frozen = [p for p in list(self.model.params) + list(self.model.computed_params) if p not in seen]

        def sizes(names):
            out, off = OrderedDict(), 0
            for n in names:
                shp = fills[n].shape
                shape = self.kernel_shape(n) if len(shp) == 5 else tuple(shp)
                cnt = _prod(shape)
                out[n] = (off, cnt, shape)
                off += (cnt + 63) // 64 * 64      # keep every view 256-byte aligned
            return out, off
        self.train_layout, ntrain = sizes(order)
        self.frozen_layout, nfrozen = sizes(frozen)
        dev = self.device
        self.flat_param = torch.zeros(max(ntrain, 1), device=dev, dtype=torch.float32)
        self.flat_frozen = torch.zeros(max(nfrozen, 1), device=dev, dtype=torch.float32)
        self.param_views, self.grad_views = {}, {}
        for n, (off, cnt, shape) in self.train_layout.items():
            self.param_views[n] = self.flat_param[off:off + cnt].view(shape)
        for n, (off, cnt, shape) in self.frozen_layout.items():
            self.param_views[n] = self.flat_frozen[off:off + cnt].view(shape)
        self.shared_params = set()
        if self.param_owner is not None and not self.train:
            for n in list(self.param_views):
                o = self.param_owner.param_views.get(n)
                if o is not None and tuple(o.shape) == tuple(self.param_views[n].shape) and o.device == self.param_views[n].device:
                    self.param_views[n] = o
                    self.shared_params.add(n)
        # weight decay per parameter (model_builder_video.py:365-372): WEIGHT_DECAY_BN when the name
        # contains '_bn', else WEIGHT_DECAY; neighbours with the same value share one solver launch
        self.wd_ranges = []
        names = list(self.train_layout)
        for i, n in enumerate(names):
            off = self.train_layout[n][0]
            end = self.train_layout[names[i + 1]][0] if i + 1 < len(names) else ntrain
            wd = float(cfg.SOLVER.WEIGHT_DECAY_BN if "_bn" in n else cfg.SOLVER.WEIGHT_DECAY)
            if self.wd_ranges and self.wd_ranges[-1][2] == wd:
                self.wd_ranges[-1][1] = end
            else:
                self.wd_ranges.append([off, end, wd])
        if self.train:
            self.flat_grad = torch.zeros_like(self.flat_param)
            self.flat_mom = torch.zeros_like(self.flat_param)
            for n, (off, cnt, shape) in self.train_layout.items():
                self.grad_views[n] = self.flat_grad[off:off + cnt].view(shape)
            if "conv1_w" in self.train_layout:
                shape = self.train_layout["conv1_w"][2]
                m = torch.zeros(shape, dtype=torch.float32)
                m[:, :, :, :7, :3] = 1.0
                self.stem_mask = m.to(dev) if not self.dry_run else m

    def _analyse_grads(self):
        """needs_grad forward propagation + number of gradient contributions per root blob"""
        for b in self.all_blobs:
            b.slot = GradSlot(self, b)
        if not self.train:
            return
        for st in self.steps:
            names = []
            if isinstance(st, ConvStep):
                names = [st.wname, st.cbname]
            elif isinstance(st, FCStep):
                names = [st.wname, st.bname]
            has_trainable = any(n and self.is_trainable(n) for n in names)
            flows = any(b.needs_grad and not b.detached for b in st.inputs)
            for o in st.outputs:
                o.needs_grad = bool(has_trainable or flows) and o.kind != "i32"
        # views created during lowering copied needs_grad early; refresh them from their roots
        for name, b in self.env.items():
            if b.root is not b and not b.detached:
                b.needs_grad = b.root.needs_grad
        for st in self.steps:
            for b in st.inputs:
                if b.root is not b and not b.detached:
                    b.needs_grad = b.root.needs_grad
        # backward reachability from the loss
        loss = self.env.get(self.model.loss_blob)
        live = {id(loss.root)}
        self.bwd_steps = []
        for st in reversed(self.steps):
            if not any(id(o.root) in live for o in st.outputs):
                continue
            self.bwd_steps.append(st)
            for b in st.grad_inputs():
                live.add(id(b.root))
                b.root.slot.expected += 1
        for b in self.all_blobs:
            if b.root is b and b.grad_scale != 1.0:
                assert b.slot.expected <= 1, "a scaled gradient (%s) must have a single contributor" % b.name
        self._plan_sparse_shortcut_dgrads()
        if self.mix and self.MIX_TRUNK2:
            # the residual stream: a blob that is the identity operand of a conv's residual Sum receives that conv's output
            # gradient as an alias and adds its own branch to it (GradSlot.two_term)
            for st in self.bwd_steps:
                if isinstance(st, ConvStep) and st.residual is not None and st.residual.needs_grad and not st.residual.detached:
                    r = st.residual.root
                    if r.kind == "act" and not r.grad_f32 and r.slot.expected > 1:
                        r.slot.two_term = True
            # ... and the INPUT of a block with a projection shortcut (pool1, pool2, the last blob of res3 / res4): there is no
            # identity operand, the whole gradient is the sum of two conv DGRADs (branch1, branch2a).  Stored as one fp16
            # value it was rounded twice at full magnitude (the first contribution on its own, then the sum) -- the
            # error profile of the trunk gradient along the backward chain (scratch/r5/diag_mix.py, full-size clip) grows
            # by as much across each of these four blobs as across all identity blocks of a stage together.  The DGRAD
            # epilogues write hi + lo for nothing but the 2 B / element of the second store.
            for st in self.bwd_steps:
                if isinstance(st, ConvStep) and st.x.needs_grad and not st.x.detached:
                    r = st.x.root
                    if r.kind == "act" and not r.grad_f32 and r.slot.expected > 1 and r.grad_scale == 1.0:
                        r.slot.two_term = True
            # ... and where the fp32 gradient of the head re-enters the 16-bit backward: the average pool over res5 hands every
            # position of a channel the SAME value, so a one-term rounding is common to all positions (it was the median error
            # of every backbone gradient: 2.8e-4 on Charades' dense loss gradient).  PoolStep.bwd writes hi + lo.
            for st in self.bwd_steps:
                if isinstance(st, PoolStep) and not st.is_max and st.out.root.grad_f32 and st.grad_inputs():
                    r = st.x.root
                    if r.kind == "act" and not r.grad_f32 and r.slot.expected == 1 and r.C % 8 == 0:
                        r.slot.two_term = True
                        st.two_term_dx = True

    SPARSE_SHORTCUT_DGRAD = True

    def _plan_sparse_shortcut_dgrads(self):
        """16-bit backward: the DGRAD of a (1, 2, 2)-strided 1x1x1 projection shortcut (res3_0 / res4_0 branch1) reaches only
        the even (h, w) positions of the block input -- three quarters of its output rows are structural zeros that the
        gathered kernel still walks the whole k-loop for (172 / 140 us per step on fp16, 292 / 232 us with two-term weights).
        When the OTHER contribution to that gradient (branch2a's DGRAD) comes first, the shortcut can run as an in-place
        accumulate over the rows it touches (hip.ALGO_CLASS0: a quarter of the tiles, nothing else is read or written).
        So: mark those convs and move their backward step behind the other contributor's (they are independent: the
        shortcut's output gradient is the block-output gradient, finished before either runs)."""
        if not self.SPARSE_SHORTCUT_DGRAD or self.btdtype not in (torch.float16, torch.bfloat16) or (self.split and not self.mix):
            return
        for st in list(self.bwd_steps):
            if not (isinstance(st, ConvStep) and tuple(st.k) == (1, 1, 1) and tuple(st.s) == (1, 2, 2) and tuple(st.p) == (0, 0, 0)
                    and st.group == 1 and st.residual is None and not st.relu and st.x.needs_grad and not st.x.detached):
                continue
            r = st.x.root
            others = [o for o in self.bwd_steps if o is not st and any(b.root is r for b in o.grad_inputs())]
            if r.kind != "act" or r.grad_f32 or r.relu or r.slot.expected != 2 or len(others) != 1 or \
                    not isinstance(others[0], ConvStep) or others[0].x.root is not r or r.shape[3] % 2 or r.shape[4] % 2:
                continue
            i, j = self.bwd_steps.index(st), self.bwd_steps.index(others[0])
            if i < j:                                   # the shortcut ran first: move it right behind the other contributor
                self.bwd_steps.insert(j, self.bwd_steps.pop(i))
            st.sparse_dgrad = True

    def _allocate(self):
        dev = self.device
        for b in self.all_blobs:
            if b.root is not b or getattr(b, "dead", False):
                continue
            if b.kind == "act":
                n = b.numel
                if getattr(b, "pad_c", None):
                    wpad = getattr(b, "pad_w", 0)
                    n = b.numel // b.C // b.shape[-1] * (b.shape[-1] + 2 * wpad) * b.pad_c
                b.tensor = torch.zeros(2 * n if b.pair else n, device=dev, dtype=torch.float16 if b.pair else self.tdtype)
            elif b.kind == "f32":
                b.tensor = torch.zeros(max(b.numel, 1), device=dev, dtype=torch.float32)
            else:
                b.tensor = torch.zeros(max(b.numel, 1), device=dev, dtype=torch.int32)
            nval = b.tensor.numel() // 2 if b.pair else b.tensor.numel()
            if self.train and b.slot.expected > 0:
                gdt = self.btdtype if (b.kind == "act" and not b.grad_f32) else torch.float32 if b.pair else b.tensor.dtype
                b.slot.buf = torch.zeros(nval, device=dev, dtype=gdt)
                if b.slot.two_term:
                    b.slot.buf_lo = torch.zeros(nval, device=dev, dtype=gdt)
                if b.relu:
                    self.want_half(b)             # the finished gradient is masked by the sign of the values
                if self.mix and self.GRAD_HALF_COPY and getattr(b, "grad_half", False) and getattr(b, "grad_half_src", False) and \
                        b.grad_f32 and b.slot.expected == 1 and not b.relu:       # (a reader that wants it AND a writer that can)
                    b.slot.half_buf = torch.zeros(nval, device=dev, dtype=self.btdtype)       # (GradSlot.half_buf)
            if b.pair:
                b.half = b.tensor[:nval]          # the hi plane IS the fp16 copy the backward reads
            elif b.need_half:
                b.half = torch.zeros(b.tensor.numel(), device=dev, dtype=torch.float16)
            # "split" dtype: bf16 term planes next to the fp32 values of conv-produced tensors whose consumers are
            # MFMA-bound convs (the large res2 / stem tensors are HBM-bound: a second copy would only cost traffic)
            if self.split and not self.mix and self.PLANES and self.train and b.kind == "act" and isinstance(b.producer, ConvStep) and \
                    not getattr(b, "pad_c", None) and b.numel <= self.PLANES_MAX_NUMEL and b.C % 8 == 0:
                # ... and only around the gathered convs (3x3, 3x1x1): their WGRAD / DGRAD gain 1.4-1.7x from pre-split
                # operands, the 1x1x1 layers gain nothing that pays for writing a second copy of their (wide) tensors
                everywhere = self.PLANES_SCOPE == "all"
                gathered = lambda st: everywhere or st.k[0] * st.k[1] * st.k[2] > 1
                if any(isinstance(st, ConvStep) and st.d_w is not None and st.x.root is b and not st.stem and gathered(st)
                       for st in self.steps):
                    b.planes = torch.zeros(2 * b.numel, device=dev, dtype=torch.bfloat16)
                if b.slot.expected > 0 and gathered(b.producer):
                    b.slot.planes = torch.zeros(2 * b.numel, device=dev, dtype=torch.bfloat16)
        self.workspace = torch.empty(max(self._ws_bytes // 4, 4), device=dev, dtype=torch.float32)
        self._scratch_f32 = torch.empty(max(self._sf32, 4), device=dev, dtype=torch.float32)
        biggest = max([b.tensor.numel() for b in self.all_blobs
                       if b.root is b and b.kind == "act" and b.tensor is not None] + [4])
        self._scratch_act = torch.empty(max(self._sact, biggest), device=dev, dtype=self.btdtype)
        self._scratch_pl = torch.empty(max(self._spl, 8), device=dev, dtype=torch.bfloat16)
        self._scratch_join = torch.empty(max(getattr(self, "_sjoin", 0), 4), device=dev, dtype=torch.float32)

    # ---- parameters ---------------------------------------------------------------------------
    def _to_kernel_layout(self, name, arr):
        t = torch.as_tensor(np.asarray(arr), dtype=torch.float32)
        if t.dim() == 5:
            co, ci = t.shape[0], t.shape[1]
            k = t.permute(0, 2, 3, 4, 1).contiguous()
            if ci == 3:
                packed = torch.zeros(self.kernel_shape(name), dtype=torch.float32)
                packed[:, :, :, :7, :3] = k
                return packed
            return k
        return t

    def _from_kernel_layout(self, name, t):
        t = t.detach().float().cpu()
        if t.dim() == 5:
            ref = self.model.param_init_net.fills[name].shape
            if ref[1] == 3:
                t = t[:, :, :, :7, :3]
            return t.permute(0, 4, 1, 2, 3).contiguous().numpy()
        return t.numpy()

    def feed_params(self, params):
        """{name: array in the reference layout (Cout,Cin,kT,kH,kW) / (out,in) / (C,)}"""
        for name, arr in params.items():
            if name not in self.param_views:
                raise KeyError("unknown parameter %r" % name)
            self.param_views[name].copy_(self._to_kernel_layout(name, arr).to(self.device))
        self._pstate[0] += 1
        self.refresh_operands(all_params=True)

    def init_params(self, seed=None):
        """run the recorded fillers (MSRAFill / GaussianFill / ConstantFill) deterministically"""
        gen = np.random.default_rng(self.base_seed if seed is None else seed)
        out = {}
        for name in list(self.model.params) + list(self.model.computed_params):
            if name in getattr(self, "shared_params", ()):
                continue                      # owned (and initialised / trained) by the engine we share with
            f = self.model.param_init_net.fills[name]
            shape = f.shape
            if f.fill == "ConstantFill":
                v = np.full(shape, f.kwargs.get("value", 0.0), dtype=np.float32)
            elif f.fill == "GaussianFill":
                v = gen.standard_normal(shape) * f.kwargs.get("std", 1.0) + f.kwargs.get("mean", 0.0)
            elif f.fill == "MSRAFill":
                fan_out = shape[0] * _prod(shape[2:])
                v = gen.standard_normal(shape) * math.sqrt(2.0 / fan_out)
            else:
                raise NotImplementedError("filler %s" % f.fill)
            out[name] = v.astype(np.float32)
        self.feed_params(out)

    def fetch_param(self, name):
        return self._from_kernel_layout(name, self.param_views[name])

    def fetch_grad(self, name):
        """parameter gradient in the reference layout (the fp16 loss scale divided out)"""
        g = self._from_kernel_layout(name, self.grad_views[name])
        return g / np.float32(self.loss_scale) if self.loss_scale != 1.0 else g

    def fetch_momentum(self, name):
        off, cnt, shape = self.train_layout[name]
        return self._from_kernel_layout(name, self.flat_mom[off:off + cnt].view(shape))

    def feed_momentum(self, momenta):
        """{name: array in the reference layout}: the `<param>_momentum` blobs of a checkpoint"""
        for name, arr in momenta.items():
            if name not in self.train_layout:
                raise KeyError("parameter %r has no momentum buffer (not trainable)" % name)
            off, cnt, shape = self.train_layout[name]
            self.flat_mom[off:off + cnt].view(shape).copy_(self._to_kernel_layout(name, arr).to(self.device))

    def _wprep_table(self, convs):
        """device tables for vlfb_weight_prep_batched over these conv steps, one per operand format in use ("mix": two-term fp16
        DGRAD copies, and plain ones for the convs whose doubled-tap form the library does not plan): [(tensor, items, tiles, format), ...] or None"""
        codes = []
        for st in convs:
            if st.wcode not in codes:
                codes.append(st.wcode)
        tabs = [t for t in (self._wprep_table_of([st for st in convs if st.wcode == c], c) for c in codes) if t is not None]
        return tabs or None

    def _wprep_table_of(self, convs, wcode):
        items, tile = [], 0
        for st in convs:
            cout, taps, cin = st.Cog, st.taps(), st.Cin_k
            for g in range(st.group):              # (a grouped conv: one item per group, each with its own operand block)
                it = hip.WPrepItem()
                it.w = _at(self.param_tensor(st.wname), g * st.wblk)
                it.scale = _at(self.param_tensor(st.sname), g * cout) if st.sname else None
                it.w_fprop = st.wf_ptr(g)
                it.w_dgrad = st.wd_ptr(g) if st.w_d is not None else None
                it.cout, it.taps, it.cin, it.tile_begin = cout, taps, cin, tile
                tile += taps * ((cout + 31) // 32) * ((cin + 31) // 32)
                items.append(it)
        if not items:
            return None
        arr = (hip.WPrepItem * len(items))(*items)
        raw = bytes(memoryview(arr))
        dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        return (dev, len(items), tile, wcode)

    def _build_wprep_tables(self):
        """device tables for vlfb_weight_prep_batched: all convs / only those with trainable weights"""
        self._wprep = {}
        for key in ("all", "trainable"):
            self._wprep[key] = self._wprep_table([st for st in self.steps if isinstance(st, ConvStep) and
                                                  (key == "all" or self.is_trainable(st.wname))])

    def refresh_operands(self, all_params=False):
        """rebuild the MFMA operand copies (one batched launch) and the effective biases"""
        if getattr(self, "_wprep", None) is None:
            self._build_wprep_tables()
        tab = self._wprep["all" if all_params else "trainable"]
        for dev, n, tiles, wcode in tab or ():
            hip.call("vlfb_weight_prep_batched", hip.ptr(dev), n, tiles, wcode)
        for st in self.steps:
            if isinstance(st, ConvStep) and st.eff_bias is not None and (all_params or st.params):
                st.refresh_bias()
        self._operand_version = self._pstate[0]

    # ---- data ---------------------------------------------------------------------------------
    def feed(self, name, arr):
        """inputs in the reference layouts (data: (N,3,T,H,W) fp32, labels int32, proposals (R,5),
        lfb (R,K,D) fp32)"""
        b = self.env[name] if name in self.env else None
        if b is None or not getattr(b.root, "is_input", False) and not getattr(b, "is_input", False):
            raise KeyError("%r is not an input blob of this model" % name)
        root = b.root
        t = torch.as_tensor(np.asarray(arr))
        if tuple(t.shape) != tuple(root.shape) and self.plan_roi_rows and str(name).startswith(("labels", "proposals", "lfb")) \
                and t.dim() == len(root.shape) and tuple(t.shape[1:]) == tuple(root.shape[1:]) and 0 < t.shape[0] < root.shape[0]:
            # a RoI batch smaller than the plan (the number of AVA boxes changes from step to step): the missing rows are
            # padding -- labels -1 (ignored by the loss: no term, no gradient, not in the normaliser), box 0 of clip 0,
            # an empty bank window.  Loss and gradients are those of the unpadded batch ("Ragged batches").
            fill = -1 if str(name).startswith("labels") else 0
            pad = torch.full((root.shape[0] - t.shape[0],) + tuple(t.shape[1:]), fill, dtype=t.dtype)
            t = torch.cat([t, pad], dim=0)
        assert tuple(t.shape) == tuple(root.shape), "feed %s: shape %r, planned %r" % (name, tuple(t.shape), root.shape)
        if root.kind == "i32":
            root.tensor.copy_(t.to(torch.int32).reshape(-1).to(self.device))
        elif root.kind == "f32":
            root.tensor.copy_(t.to(torch.float32).reshape(-1).to(self.device))
        elif getattr(root, "pad_c", None):
            src = t.to(torch.float32).contiguous().to(self.device)
            N, Cc = root.shape[0], root.shape[1]
            wpad = getattr(root, "pad_w", 0)
            W = root.shape[-1]
            hip.call("vlfb_ncthw_to_nthwc_wpad", hip.ptr(src), root.ptr(), self.code, N, Cc,
                     _prod(root.shape[2:-1]), W, root.pad_c, wpad, W + 2 * wpad)
            torch.cuda.current_stream().synchronize()
        else:   # row-major activation input (lfb)
            src = t.to(torch.float32).contiguous().reshape(-1).to(self.device)
            hip.call("vlfb_cast", hip.ptr(src), hip.F32, root.ptr(), self.code, src.numel())
            torch.cuda.current_stream().synchronize()

    def blob_tensor(self, name):
        """(device tensor, dtype code) behind a blob, for device-side producers / consumers that skip
        the host (the feature bank appends `box_pooled` from it and samples into `lfb`).  Row-major
        blobs only: (rows, C[,1,1,1]) activations and (R, K, D) banks are stored exactly as shaped."""
        b = self.env[name].root
        if getattr(b, "dead", False):
            raise KeyError("blob %r was fused away" % name)
        if getattr(b, "pad_c", None):
            raise KeyError("blob %r is stored padded; use feed()/fetch()" % name)
        spatial = b.shape[2:] if b.caxis == 1 else ()
        if any(int(d) != 1 for d in spatial):
            raise KeyError("blob %r is stored channels-last; use feed()/fetch()" % name)
        t = b.tensor[:b.numel]
        return t, hip.dtype_code(t.dtype)

    def blob_padded(self, name):
        """the W-/channel-padded clip input (`data`) as a device tensor [N][T][H][W + 2*pad_w][pad_c] and
        its (pad_w, pad_c): the destination of datasets.data_input_helper.images_and_boxes_preprocessing.
        Padding pixels / the padding channel are zero and must stay zero."""
        b = self.env[name].root
        if not getattr(b, "pad_c", None):
            raise KeyError("blob %r is not stored padded" % name)
        wpad = getattr(b, "pad_w", 0)
        N, T, H, W = b.shape[0], b.shape[2], b.shape[3], b.shape[4]
        n = N * T * H * (W + 2 * wpad) * b.pad_c
        return b.tensor[:n].view(N, T, H, W + 2 * wpad, b.pad_c), (wpad, b.pad_c)

    def fetch(self, name):
        """blob (or its gradient with suffix '_grad') as a float32 numpy array in the reference layout"""
        grad = False
        if name not in self.env and name.endswith("_grad"):
            name, grad = name[:-5], True
        b = self.env[name]
        if getattr(b.root, "dead", False):
            raise KeyError("blob %r was fused away (its value only exists inside a kernel epilogue)" % name)
        src = b.root.slot.cur if grad else b.root.tensor
        t = src.detach().float().cpu()
        if b.root.pair and not grad:              # two fp16 planes: value = hi + lo
            t = t[:t.numel() // 2] + t[t.numel() // 2:]
        if grad and self.loss_scale != 1.0:
            t = t / self.loss_scale
        if grad and b.root.grad_scale != 1.0:     # fp16: theta / phi gradients are stored times a power of two
            t = t / b.root.grad_scale
        if getattr(b.root, "pad_c", None) and not grad:
            wpad = getattr(b.root, "pad_w", 0)
            W = b.shape[-1]
            t = t.view(-1, W + 2 * wpad, b.root.pad_c)[:, wpad:wpad + W, :b.C].reshape(-1)
        order = [ax for ax in range(len(b.shape)) if ax != b.caxis] + [b.caxis]
        stor = t[:b.numel].view([b.shape[ax] for ax in order])
        inv = [order.index(ax) for ax in range(len(b.shape))]
        return stor.permute(inv).contiguous().numpy()

    def discrete_decisions(self):
        """The discrete decisions of the last forward pass in the reference layout: the sign pattern of every ReLU output
        ({blob: bool array}), the selected window element of every max pool ({blob: tap index in t, h, w scan order}) and
        the arg-max bin of the RoI head ((R, C)).  Test support: the oracle can evaluate the same branches of the
        piecewise-linear network (oracle.model.run(decisions=...)), which separates arithmetic parity from ties at zero."""
        dec = {"relu": {}, "pool": {}, "roi_bin": None}
        for b in self.all_blobs:
            if b.root is b and b.relu and b.kind == "act" and b.tensor is not None and not getattr(b, "dead", False):
                dec["relu"][b.name] = self.fetch(b.name) > 0
        for st in self.steps:
            if isinstance(st, PoolStep) and st.is_max and st.argmax is not None:
                N, Cc = st.out.shape[0], st.out.shape[1]
                sp = tuple(st.out.shape[2:])
                raw = st.argmax.cpu().numpy()
                nb = raw.size // st.out.numel
                idx = raw.view(np.uint16) if nb == 2 else raw
                dec["pool"][st.out.name] = idx.reshape((N,) + sp + (Cc,)).transpose(0, 4, 1, 2, 3).astype(np.int64)
            elif isinstance(st, RoiAlignMaxStep):
                dec["roi_bin"] = st.argbin.cpu().numpy().reshape(st.R, st.Cc).astype(np.int64)
        return dec

    # ---- execution ----------------------------------------------------------------------------
    def forward(self):
        if self._operand_version != self._pstate[0]:
            # parameters were changed through another engine that shares them (the train net's solver)
            self.refresh_operands(all_params=True)
        if self._half_inputs:                       # "mix": fp16 copies of the fed blobs the backward reads (clip, bank)
            self._half_copies(self._half_inputs)
        if self.side is None or not self.FORWARD_BRANCHES or not self._fwd_side:
            for st in self.steps:
                self._fwd_step(st)
            return
        # The parameter-gradient stream is idle during forward: the projection shortcut of a stage's first block and
        # the pooled phi / g branch of a non-local block run on it, beside the bottleneck chain / the theta conv they
        # are independent of (20-150 us launches that leave most of the chip idle on their own).  Only the edges that
        # cross streams carry an event.
        main = torch.cuda.current_stream()
        done = {}
        if self._fwd_early:
            # the bank side of the FBO head (rule (c) of _plan_forward_branches): issued first, behind whatever fed the inputs
            self.wait_event(self.side, self.record_event())
            with torch.cuda.stream(self.side):
                for i in self._fwd_early:
                    self._fwd_step(self.steps[i])
                    if i in self._fwd_signal:
                        done[i] = self.record_event(self.side)
        for i, st in enumerate(self.steps):
            if i in self._fwd_early_set:
                continue
            on_side = i in self._fwd_side
            stream = self.side if on_side else main
            for j in self._fwd_wait[i]:
                self.wait_event(stream, done[j])
            if on_side:
                with torch.cuda.stream(self.side):
                    self._fwd_step(st)
            else:
                self._fwd_step(st)
            if i in self._fwd_signal:
                done[i] = self.record_event(stream)

    def _fwd_step(self, st):
        st.fwd()
        if st._half_post:                           # "mix": outputs whose fp16 copy no conv epilogue wrote
            self._half_copies(st._half_post)

    # "mix": the copy passes only feed the BACKWARD pass, so they do not have to sit in the forward chain: they run on the
    # parameter-gradient stream (idle during forward) behind an event of the producer; backward() joins that stream first
    HALF_COPIES_ON_SIDE = True

    def _half_copies(self, blobs):
        cur = torch.cuda.current_stream() if not self.dry_run else None
        # (a captured forward pass has to end with every forked stream joined: the copies stay in the chain there)
        if self.side is None or not self.HALF_COPIES_ON_SIDE or cur == self.side or self.STEP_GRAPH:
            for b in blobs:
                hip.call("vlfb_half_copy", b.ptr(), hip.ptr(b.half), b.half.numel())
            return
        self.wait_event(self.side, self.record_event())
        with torch.cuda.stream(self.side):
            for b in blobs:
                hip.call("vlfb_half_copy", b.ptr(), hip.ptr(b.half), b.half.numel())
        self._half_pending = True

    def _plan_half_copies(self):
        """which fp16 copies (Blob.half, "mix" dtype) are made by a copy pass: behind the step that produced the values
        (ConvStep writes them in its epilogue), or at the start of forward() for fed blobs"""
        seen = set()
        for st in self.steps:
            st._half_post = []
            if isinstance(st, ConvStep) and not st.half_by_copy:
                seen.update(id(o.root) for o in st.outputs)
                continue
            for o in list(st.outputs) + list(getattr(st, "aux_outputs", ())):
                r = o.root
                if r.pair:
                    seen.add(id(r))               # (two planes: the hi plane is the copy)
                if r.half is not None and id(r) not in seen:
                    seen.add(id(r))
                    st._half_post.append(r)
        self._half_inputs = [b for b in self.all_blobs if b.root is b and b.half is not None and id(b) not in seen]
        for b in self._half_inputs:
            assert getattr(b, "is_input", False), "blob %s has an fp16 copy that nothing writes" % b.name

    # independent forward branches on the second stream (Engine.forward)
    FORWARD_BRANCHES = True
    FORWARD_BANK_SIDE = os.environ.get("VLFB_FORWARD_BANK_SIDE", "1") != "0"       # (rule (c) below; A/B switch)

    def _plan_forward_branches(self):
        """which forward steps run on the second stream, and which cross-stream edges need an event"""
        steps = self.steps
        side = set()
        writer = {}                                   # id(root blob) -> index of the step that wrote it last
        writers_at = []                               # per step: {id(root): writer index} of its reads
        for i, st in enumerate(steps):
            reads = [b.root for b in st.inputs]
            if isinstance(st, ConvStep) and st.residual is not None:
                reads.append(st.residual.root)
            writers_at.append({id(r): writer[id(r)] for r in reads if id(r) in writer})
            for b in st.outputs:
                writer[id(b.root)] = i
        readers = {}
        for i, w in enumerate(writers_at):
            for j in set(w.values()):
                readers.setdefault(j, []).append(i)
        for i, st in enumerate(steps):
            # (a) projection shortcut: a conv whose output is only the residual operand of a later conv, and whose input
            #     feeds another chain as well
            if isinstance(st, ConvStep) and not st.stem and st.residual is None:
                cons = readers.get(i, [])
                if len(cons) == 1 and isinstance(steps[cons[0]], ConvStep) and steps[cons[0]].residual is not None and \
                        steps[cons[0]].residual.root is st.out.root and steps[cons[0]].x.root is not st.out.root:
                    side.add(i)
            # (b) non-local block: the max-pool and the phi / g convs it feeds (theta stays on the main stream)
            if isinstance(st, AttentionStep):
                prods = [writers_at[i].get(id(b.root)) for b in st.inputs]
                if all(p is not None and isinstance(steps[p], ConvStep) for p in prods):
                    for p in prods[1:]:
                        src = writers_at[p].get(id(steps[p].x.root))
                        if src is not None and isinstance(steps[src], PoolStep) and \
                                all(isinstance(steps[c], ConvStep) for c in readers.get(src, [])):
                            side.add(p)
                            side.add(src)
        # (c) the bank side of the FBO head: steps that do not depend on the clip at all -- lfb -> (dropout) -> lfb_1x1 -> the
        #     phi / g convs of the FBO blocks -- sit at the END of the step list (the head is built last) but can run from
        #     the start of the pass: on the second stream they finish under the stem instead of extending the chain
        early = []
        if self.FORWARD_BANK_SIDE:
            clip_dep = set()                          # steps that (transitively) read the clip
            for i, st in enumerate(steps):
                reads = [b.root for b in st.inputs]
                if isinstance(st, ConvStep) and st.residual is not None:
                    reads.append(st.residual.root)
                from_clip = any(getattr(r, "is_input", False) and r.name.startswith("data") for r in reads)
                if from_clip or any(j in clip_dep for j in writers_at[i].values()):
                    clip_dep.add(i)
            for i, st in enumerate(steps):
                reads = [b.root for b in st.inputs]
                if i not in clip_dep and reads and isinstance(st, (ConvStep, DropoutStep)) and \
                        all(getattr(r, "is_input", False) or writers_at[i].get(id(r)) in early for r in reads):
                    early.append(i)
            side.update(early)
        self._fwd_early = early                       # issued at the start of forward(), in this order
        self._fwd_early_set = set(early)
        self._fwd_side = side
        self._fwd_wait = []
        self._fwd_signal = set()
        for i in range(len(steps)):
            waits = sorted(j for j in set(writers_at[i].values()) if (j in side) != (i in side))
            self._fwd_wait.append(waits)
            self._fwd_signal.update(waits)

    def backward(self):
        assert self.train, "backward() on a forward-only engine"
        for b in self.all_blobs:
            if b.root is b and b.slot is not None:
                b.slot.reset()
        del self._wq[:]               # (parameter-gradient launches a failed backward() left queued: WGRAD_LAG)
        self._bwd_index = 0
        if getattr(self, "_half_pending", False):    # "mix": the fp16 copies made on the parameter-gradient stream
            self.wait_stream(self.side)
            self._half_pending = False
        if self.comm is not None:
            self.traced(self.comm.begin, "comm begin")
        eager = self._eager_lr is not None
        self._eager_next = 0
        buckets = self.sol_buckets
        for i, st in enumerate(self.bwd_steps):
            self._bwd_index = i
            st.bwd()
            self._flush_param_grads(i - self.WGRAD_LAG)
            if (self.comm is not None and self.comm.due(i)) or \
                    (eager and self._eager_next < len(buckets) and buckets[self._eager_next]["ready"] <= i and
                     (self.EAGER_SOLVER != "tail" or i >= len(self.bwd_steps) - 2)):
                self._bucket_ready(i)
        if eager and self._eager_next < len(buckets):
            self._bucket_ready(1 << 60)
        self._eager_done = eager
        self._flush_param_grads(1 << 60)
        self.join_side_stream()

    def _bucket_ready(self, i):
        """The buckets that became final with backward step i: all-reduce them and, in a train_step() (the
        learning rate of the step is known), run the solver and the operand refresh for them right away --
        WITHOUT stalling the dgrad chain or the parameter-gradient stream.  Their gradients were produced on
        the side stream (wgrads, bias column sums) or, for the classifier, on the main stream, so a third
        stream waits for both positions; ProcessGroupNCCL orders the collective behind the stream that is
        current at the call, and the solver kernels of the bucket follow the collective on that stream.  The
        main stream joins it once, at the end of backward (the forward of the next step reads the refreshed
        operand copies).  Safe because nothing later in backward reads a parameter of a finished bucket: the
        dgrad of a layer is enqueued (main stream) before the event this hand-off waits for."""
        eager = self._eager_lr is not None
        if self.side is None or self.BUCKET_HANDOFF == "join":
            # single-stream development mode, or the simple hand-off (Engine.BUCKET_HANDOFF = "join"): the main stream
            # joins the parameter-gradient stream and issues the collective itself -- no third stream, no cross-stream
            # events to get wrong; costs the overlap of the wgrad stream at every bucket boundary
            self.join_side_stream()
            if self.comm is not None:
                self.traced(lambda: self._issue_reductions(None, i, eager), "all-reduce buckets")
            if eager:
                self._solve_ready_buckets(i)
            return
        # position of the dgrad chain (main stream) and of the parameter-gradient stream; the third stream waits for
        # both -- neither of the two compute streams waits for anything here
        sol = self.solver_stream
        self.wait_event(sol, self.record_event())
        self.wait_event(sol, self.record_event(self.side))
        if self.comm is not None:
            self.traced(lambda: self._issue_reductions(sol, i, eager), "all-reduce buckets")
        with torch.cuda.stream(sol):
            if eager:
                self._solve_ready_buckets(i)
        self.side_dirty = True
        self.solver_dirty = True

    def _solve_ready_buckets(self, i):
        if self.EAGER_SOLVER == "tail" and i < len(self.bwd_steps) - 2:
            return        # "tail": the finished buckets are solved together beside the last (MFMA-bound) stem wgrad
        while self._eager_next < len(self.sol_buckets) and self.sol_buckets[self._eager_next]["ready"] <= i:
            self._solve_bucket(self.sol_buckets[self._eager_next], self._eager_lr)
            self._eager_next += 1

    def _solve_bucket(self, b, lr):
        """WeightedSum + MomentumSGDUpdate (model_builder_video.py:348-389) and the MFMA operand refresh for the
        parameters of one bucket, on the current stream"""
        sol = cfg.SOLVER
        S = self.loss_scale                       # gradients carry the fp16 loss scale: lr/S * (S g + S wd p)
        for off, end, wd in b["wd"]:
            self._sgd_launch(off, end, wd, lr)
        for dev, n, tiles, wcode in b["wprep"] or ():
            hip.call("vlfb_weight_prep_batched", hip.ptr(dev), n, tiles, wcode)
        for st in b["bias_steps"]:
            st.refresh_bias()

    def _sgd_launch(self, off, end, wd, lr):
        sol = cfg.SOLVER
        S = self.loss_scale                       # gradients carry the fp16 loss scale: lr/S * (S g + S wd p)
        args = (hip.ptr(self.flat_param) + 4 * off, hip.ptr(self.flat_grad) + 4 * off, hip.ptr(self.flat_mom) + 4 * off,
                end - off)
        if self._dev_scalars:                     # captured step: lr/S is slot 0 of the step scalars
            hip.call("vlfb_sgd_update_dev", *args, hip.ptr(self._scalars_dev), wd * S, float(sol.MOMENTUM),
                     int(bool(sol.NESTEROV)))
        else:
            hip.call("vlfb_sgd_update", *args, lr / S, wd * S, float(sol.MOMENTUM), int(bool(sol.NESTEROV)))

    def _plan_solver_buckets(self, bucket_mb=32):
        """buckets of the flat parameter buffer in backward-completion order (the all-reduce buckets), each with
        its weight-decay ranges, the batched operand-refresh table of its conv weights and its effective biases"""
        from vlfb.comm import make_buckets
        index = {id(st): i for i, st in enumerate(self.bwd_steps)}
        # (flat order = backward completion order as planned BEFORE _plan_sparse_shortcut_dgrads moved the strided shortcuts
        # behind their sibling DGRAD: a shortcut's weights are then ready 2-3 steps later than their flat neighbours.
        # make_buckets takes the MAX ready step of a bucket's segments, so a bucket is never issued early.)
        segs = []
        for n in self.train_order:
            off, cnt, _ = self.train_layout[n]
            segs.append((off, cnt, index[id(self.param_step[n])]))
        self._bucket_mb = int(bucket_mb)
        self.sol_buckets = []
        for start, end, ready in make_buckets(segs, int(bucket_mb) << 20, 4):
            names = [n for n in self.train_order if start <= self.train_layout[n][0] < end]
            wd = [[max(o, start), min(e, end), w] for o, e, w in self.wd_ranges if max(o, start) < min(e, end)]
            convs = [st for st in self.steps if isinstance(st, ConvStep) and st.wname in names]
            bias_steps = [st for st in self.steps if isinstance(st, ConvStep) and st.eff_bias is not None and
                          any(p in names for p in st.params)]
            self.sol_buckets.append({"start": start, "end": end, "ready": ready, "wd": wd, "names": names,
                                     "wprep": None if self.dry_run else self._wprep_table(convs),
                                     "bias_steps": bias_steps})

    def plan_table(self, launched=False):
        """[(step, role, launch tag, kernel family / tile / splits)] for every implicit-GEMM launch of a step, from the
        library's planner (a pure function of the descriptor: what the table says is what runs).  launched=True: with the
        operand-plane variants ("split" dtype) the last forward / backward pass actually used.  Test / bench support."""
        rows = []
        for st in self.steps:
            if isinstance(st, ConvStep):
                descs = (("fprop", st.d_f), ("dgrad", st.d_d), ("wgrad", st.d_w))
            elif isinstance(st, AttentionStep) and not st.single:
                descs = (("scores", st.d_s), ("p.g", st.d_y), ("dP", st.d_dp), ("dtheta", st.d_dth), ("dg", st.d_tn), ("dphi", st.d_tn_phi))
            else:
                continue
            for role, d in descs:
                if d is not None and (self.train or role in ("fprop", "scores", "p.g")):
                    if launched and isinstance(st, ConvStep):
                        d = st._ran.get(id(d), d)
                    rows.append((st.name(), role, hip.conv_tag(d), hip.conv_plan(d)))
        return rows

    def set_lr(self, lr):
        self.lr = float(lr)

    def enable_data_parallel(self, bucket_mb=32):
        """clip-level data parallel over the current process group: identical weights on every rank
        (broadcast from rank 0), bucketed sum-all-reduce of the flat gradient during backward"""
        import torch.distributed as td
        from vlfb.comm import GradComm
        if not (self.train and (dist.world_size() > 1 or (dist.forced() and dist.initialized()))):
            return
        # the loss is pre-scaled by 1/NUM_GPUS and the per-GPU batch is BATCH_SIZE/NUM_GPUS
        # (resnet_video.py:333-338, misc.py:68-72): a job of another size would silently mis-scale gradients
        if dist.world_size() != int(cfg.NUM_GPUS):
            raise hip.VlfbError("data parallel: %d ranks but cfg.NUM_GPUS = %d (loss scale and per-GPU batch come "
                                "from NUM_GPUS)" % (dist.world_size(), int(cfg.NUM_GPUS)))
        # the summed gradients carry the loss scale: it has to be the same number everywhere
        ls = torch.tensor([self.loss_scale, -self.loss_scale], device=self.device, dtype=torch.float64)
        td.all_reduce(ls, op=td.ReduceOp.MAX)
        if float(ls[0]) != self.loss_scale or float(ls[1]) != -self.loss_scale:
            raise hip.VlfbError("data parallel: the ranks chose different loss scales (%g here, %g .. %g in the job)"
                                % (self.loss_scale, -float(ls[1]), float(ls[0])))
        td.broadcast(self.flat_param, 0)
        td.broadcast(self.flat_frozen, 0)
        self.refresh_operands(all_params=True)
        if int(bucket_mb) != self._bucket_mb:
            self._plan_solver_buckets(bucket_mb)
        self.comm = GradComm(self.flat_grad, None, int(bucket_mb) << 20,
                             buckets=[(b["start"], b["end"], b["ready"]) for b in self.sol_buckets])

    def recent_losses(self):
        """losses since the last call (device ring of the loss step; a single host sync)"""
        out = []
        for st in self.steps:
            if isinstance(st, LossStep):
                out.extend(st.recent_losses())
        return out

    def scale_momentum(self, factor):
        hip.call("vlfb_scale_inplace", hip.ptr(self.flat_mom), self.flat_mom.numel(), float(factor))

    def sgd_step(self, lr=None):
        """WeightedSum + MomentumSGDUpdate over the flat bucket (model_builder_video.py:348-389)"""
        if lr is not None:
            self.lr = float(lr)
        if self.comm is not None:
            self.traced(self.comm.wait, "comm wait")      # (the current stream waits for every bucket's reduction)
        if self._eager_done:
            # train_step(): every bucket was solved and refreshed during backward, as soon as it was final
            assert self._eager_next == len(self.sol_buckets)
            self._eager_done = False
            self._pstate[0] += 1
            self._operand_version = self._pstate[0]
            self.iteration += 1
            return
        sol = cfg.SOLVER
        S = self.loss_scale                       # gradients carry the fp16 loss scale: lr/S * (S g + S wd p)
        for off, end, wd in self.wd_ranges:       # one launch unless a trainable '_bn' parameter exists
            self._sgd_launch(off, end, wd, self.lr)
        self._pstate[0] += 1
        self.refresh_operands()
        self.iteration += 1

    # True: train_step() solves every gradient bucket during backward, as soon as it is final (the learning rate
    # of the step is known before backward starts), on the third stream; False: one solver pass after backward.
    # Bit-identical (tests/test_eager_solver_gpu.py).  Measured on one MI355X at 8 clips: 425.8 vs 427.3 clips/s --
    # the solver's 0.9 GB of traffic then competes with the HBM-bound res2 / res3 backward instead of running
    # alone after it, which costs as much as the hidden tail saves -- so it is off unless a multi-GPU run wants the
    # solver of a bucket to follow its all-reduce directly.  "tail" = solve the finished buckets together beside the
    # last wgrad of backward (the MFMA-bound stem wgrad leaves HBM idle): 440.5 vs 440.4 clips/s, no gain either.
    EAGER_SOLVER = False

    # "streams": finished gradient buckets are handed to a third stream (all-reduce + optional per-bucket solver) by
    # two events, the compute streams never wait; "join": the main stream joins the parameter-gradient stream at every
    # bucket boundary and issues the collective itself (the conservative fallback; tests/test_dp_gpu.py runs both)
    BUCKET_HANDOFF = "streams"

    # True: from its second call on, train_step() replays the whole step (forward, backward on both streams,
    # solver, operand refresh) as ONE captured HIP graph; "forward": only the forward pass.  Nothing in the step
    # depends on the host -- shapes, plans and buffers are fixed at plan() -- except the learning rate and the dropout
    # seeds, which the captured kernels read from device memory (vlfb_store_scalars writes them in stream order before
    # each replay).  Bit-identical to the stream path (tests/test_step_graph_gpu.py).
    # OFF by default, because it measures slower on this runtime (ROCm 7.2, one MI355X, 8 clips, same box):
    #   streams 18.80 ms | whole-step graph 20.27 ms | forward-only graph 18.84 ms
    #   one stream only: streams 20.62 ms | graph 20.60 ms
    # A chain of empty kernels dispatches in 1.7 us per node from a graph against 4.5 us from a stream
    # (scratch/graph_probe.py), but behind real kernels the command processor already hides the dispatch of the next
    # packet, so the one-stream step gains nothing; and the graph executor runs the forked branches (wgrads beside
    # the dgrad chain) almost serially, which loses what the second stream buys.  The host is not the limit either:
    # enqueuing a step takes ~6 ms of the 18.8 ms it runs.  Data-parallel steps (collectives inside backward) and
    # profiled steps (hip.PROFILE) always use the stream path.
    STEP_GRAPH = False

    # True: the second train_step() records every call it makes into the library and every event / stream edge
    # (hip.TRACE), and later steps re-issue that list instead of walking the step objects: same kernels, same streams,
    # same order, same cross-stream edges -- only the Python between the calls (operand lookup, gradient-slot
    # bookkeeping, descriptor selection, ~450 ctypes argument conversions) is gone.  Unlike STEP_GRAPH the device
    # still sees ordinary stream launches, so the two-stream overlap is kept.  The per-iteration values (learning rate,
    # dropout seeds) are read from device memory as in a captured step.  Bit-identical to the stream path
    # (tests/test_step_graph_gpu.py).  Data-parallel steps are recorded too (round 6): the all-reduces torch issues and the
    # communicator's bookkeeping are host actions appended to the list at the point of the step they happen at
    # (Engine.traced), so a replay re-issues every collective between the same launches, behind the same stream edges.
    # Profiled steps use the stream path; the trace is re-recorded when the stream, the communicator or a class switch changes.
    # Measured on one MI355X (8 clips bf16, idle queue): enqueuing a step takes 6.1-6.3 ms through the step objects,
    # 1.9-2.8 ms from the trace (C3 frozen backbone: 2.5 -> 0.75 ms); the step itself is GPU-bound either way
    # (440.1 vs 440.1 clips/s), the point is the host thread the data loader shares.
    STEP_TRACE = True

    def train_step(self, lr=None):
        if lr is not None:
            self.lr = float(lr)
        if self.STEP_TRACE and not self.STEP_GRAPH and hip.PROFILE is None and self._eager_steps >= 1 and not self.dry_run:
            return self._trace_step()         # (data-parallel steps too: the collectives are recorded host actions, Engine.traced)
        if self.STEP_GRAPH and self.comm is None and hip.PROFILE is None and self._eager_steps >= 1 and \
                not self.dry_run:
            if self.STEP_GRAPH == "forward":
                self._graph_replay("forward")
                self._backward_and_solve()
                return
            return self._graph_replay("step")
        self._train_step_streams()
        self._eager_steps += 1

    def _backward_and_solve(self):
        self._eager_lr = self.lr if self.EAGER_SOLVER else None
        try:
            self.backward()
        finally:
            self._eager_lr = None
        self.sgd_step()

    def _train_step_streams(self):
        self.forward()
        self._backward_and_solve()

    def _store_step_scalars(self):
        """learning rate (fp32 bits, already divided by the loss scale) and the dropout seeds of this iteration"""
        vals = [struct.unpack("<I", struct.pack("<f", self.lr / self.loss_scale))[0]]
        vals += [dropout_seed(self.base_seed, st.out.name, self.iteration, self.replica) for st in self._drop_steps]
        for c in range(0, len(vals), 8):
            chunk = vals[c:c + 8]
            arr = (C.c_uint64 * len(chunk))(*chunk)
            hip.call("vlfb_store_scalars", hip.ptr(self._scalars_dev) + 8 * c, len(chunk), arr)

    def _trace_step(self):
        if self._operand_version != self._pstate[0]:
            self.refresh_operands(all_params=True)
        key = (torch.cuda.current_stream().cuda_stream, self.side is None, self.EAGER_SOLVER, self.FORWARD_BRANCHES,
               self.BUCKET_HANDOFF, self.WGRAD_LAG, id(self.comm))
        self._store_step_scalars()
        if self._trace is None or self._trace_key != key:
            self._trace, self._trace_key = None, None
            self._dev_scalars = True
            rec = hip.trace_begin()                  # (this thread's calls only)
            try:
                self._train_step_streams()           # runs the step for real while recording it
            finally:
                self._dev_scalars = False
                hip.trace_end()
            self._trace_losses = [st for st in self.steps if isinstance(st, LossStep)]
            for st in self._trace_losses:
                st.ring_push()
            self._trace, self._trace_key = rec, key
            return
        for fn, args, name in self._trace:
            rc = fn(*args)
            if rc:
                hip._check(rc, name)
        for st in self._trace_losses:
            st.ring_push()
        self._pstate[0] += 1
        self._operand_version = self._pstate[0]
        self.iteration += 1

    def _capture_step(self, key, scope):
        losses = [st for st in self.steps if isinstance(st, LossStep)]
        saved = (self.iteration, self._pstate[0], self._operand_version, [st.ring_pos for st in losses])
        if self._graph_stream is None:
            self._graph_stream = torch.cuda.Stream(device=self.device)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self._dev_scalars = True
        try:
            with torch.cuda.graph(graph, stream=self._graph_stream):
                if scope == "forward":
                    self.forward()
                else:
                    self._train_step_streams()
        finally:
            self._dev_scalars = False
        # capturing enqueues nothing: undo the host-side bookkeeping of the pass
        self.iteration, self._pstate[0], self._operand_version = saved[0], saved[1], saved[2]
        for st, pos in zip(losses, saved[3]):
            st.ring_pos = pos
        self._graph, self._graph_key, self._graph_losses = graph, key, losses

    def _graph_replay(self, scope):
        if self._operand_version != self._pstate[0]:
            self.refresh_operands(all_params=True)
        key = (scope, self.EAGER_SOLVER, self.FORWARD_BRANCHES)
        if self._graph is None or self._graph_key != key:
            self._capture_step(key, scope)
        self._store_step_scalars()
        self._graph.replay()
        for st in self._graph_losses:
            st.ring_push()
        if scope == "step":
            self._pstate[0] += 1
            self._operand_version = self._pstate[0]
            self.iteration += 1
