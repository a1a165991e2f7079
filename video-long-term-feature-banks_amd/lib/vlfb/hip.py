"""ctypes binding of libvlfb_hip.so (include/vlfb.h).

This is the whole Python<->native boundary: torch-ROCm tensors supply device memory
(`data_ptr()`) and the stream (`torch.cuda.current_stream().cuda_stream`); every compute call
lands in a hand-written HIP kernel.  There is deliberately NO fallback: if the shared library is
missing or a call fails, we raise.
"""
import ctypes as C
import os

import torch

F32, BF16, F16 = 0, 1, 2
SPLIT = 3                                           # weight-operand format of the split-bf16 path (vlfb.h VLFB_SPLIT)
MIX, MIX_W2 = 4, 5                                  # ... of the "mix" path (split FPROP copy, fp16 DGRAD copy: plain / two terms)
MIXH, MIXH_W2 = 6, 7                                # ... of the two-plane fp16 forward (two-term fp16 FPROP copy; DGRAD as MIX / MIX_W2)
MIX_W2I, MIXH_W2I = 9, 10                           # ... with the two-term DGRAD copy interleaved per 64-channel k-tile (MATH_F16W2)
F16PAIR = 8                                         # two-plane fp16 tensors (vlfb.h VLFB_F16PAIR; pool descriptors)
MIX_W2_SCALE = 1024.0                               # vlfb.h VLFB_MIX_W2_SCALE
MATH_NATIVE, MATH_BF16X3, MATH_BF16X6, MATH_F16X3 = 0, 3, 6, 13     # vlfb_conv_desc.math
MATH_F16W2 = 12                                     # ... 16-bit DGRAD on two-term weights interleaved per k-tile (MIX_W2I)
FPROP, DGRAD, WGRAD = 0, 1, 2
BIAS_NONE, BIAS_COL, BIAS_ROW = 0, 1, 2
ALGO_AUTO, ALGO_TILE128, ALGO_PIPE256, ALGO_STREAM, ALGO_CLASSES, ALGO_CLASS0 = 0, 1, 2, 3, 4, 5
ATTN_CAN_RUN, ATTN_FWD_FASTER, ATTN_BWD_FASTER = 1, 2, 4     # vlfb_attn_scores_supported flags

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvlfb_hip.so")

TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    if t == torch.float16:
        return F16
    raise TypeError("vlfb: unsupported dtype %r" % (t,))


class VlfbError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("N", C.c_int32), ("Tr", C.c_int32), ("Hr", C.c_int32), ("Wr", C.c_int32),
        ("Ts", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("Cs", C.c_int32),
        ("kt", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("st", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
        ("pt", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
        ("dt", C.c_int32), ("dh", C.c_int32), ("dw", C.c_int32),
        ("pack_w", C.c_int32), ("Cn", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32),
        ("ldp", C.c_int32),
        ("batch", C.c_int32),
        ("a_bstride", C.c_int64), ("b_bstride", C.c_int64), ("o_bstride", C.c_int64),
        ("r_bstride", C.c_int64), ("p_bstride", C.c_int64),
        ("alpha", C.c_float), ("relu", C.c_int32), ("bias_mode", C.c_int32),
        ("accumulate", C.c_int32), ("splits", C.c_int32), ("algo", C.c_int32),
        ("math", C.c_int32), ("b_pstride", C.c_int64),
        ("a_planes", C.c_int32), ("p_planes", C.c_int32), ("o_planes", C.c_int32), ("wgrad_bias", C.c_int32),
        ("a_pstride", C.c_int64), ("p_pstride", C.c_int64), ("o_pstride", C.c_int64),
    ]


class ConvArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("P", C.c_void_p), ("O", C.c_void_p), ("bias", C.c_void_p),
                ("rowscale", C.c_void_p), ("R", C.c_void_p), ("Mask", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("O_planes", C.c_void_p), ("dbias", C.c_void_p), ("R_lo", C.c_void_p),
                ("O_lo", C.c_void_p)]


class WPrepItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("w_fprop", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("cout", C.c_int32), ("taps", C.c_int32), ("cin", C.c_int32), ("tile_begin", C.c_int32)]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "N", "Ti", "Hi", "Wi", "C", "To", "Ho", "Wo",
        "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw")]


class ClipDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frames", "src_h", "src_w", "resized_h", "resized_w", "crop_h", "crop_w",
                                         "y0", "x0", "flip", "to_rgb", "w_left", "w_total", "c_pad")] + \
               [("mean", C.c_float * 3), ("std", C.c_float * 3)]


class LfbDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_videos", "n_steps", "capacity", "dim", "dtype", "step_base")]


_P = C.c_void_p
_I64 = C.c_int64
_SIGS = {
    "vlfb_last_error": (C.c_char_p, []),
    "vlfb_version": (C.c_int, []),
    "vlfb_dtype_size": (C.c_int, [C.c_int]),
    "vlfb_affine_nd_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "vlfb_affine_nd_bwd": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "vlfb_conv_desc_init": (None, [C.POINTER(ConvDesc)]),
    "vlfb_conv_workspace_bytes": (_I64, [C.POINTER(ConvDesc)]),
    "vlfb_query_workspace": (_I64, [C.c_int, C.c_void_p]),
    "vlfb_conv_plan_describe": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, _I64]),
    "vlfb_conv_run": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "vlfb_conv_run_planes": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "vlfb_conv_run_args": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvArgs), _P]),
    "vlfb_conv_run_wgrad_bias": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _I64, _P]),
    "vlfb_ncthw_to_nthwc": (C.c_int, [_P, _P, C.c_int, _I64, _I64, _I64, _I64, _P]),
    "vlfb_ncthw_to_nthwc_wpad": (C.c_int, [_P, _P, C.c_int, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P]),
    "vlfb_nthwc_to_ncthw": (C.c_int, [_P, _P, C.c_int, _I64, _I64, _I64, _P]),
    "vlfb_cast": (C.c_int, [_P, C.c_int, _P, C.c_int, _I64, _P]),
    "vlfb_half_copy": (C.c_int, [_P, _P, _I64, _P]),
    "vlfb_pair_split": (C.c_int, [_P, _P, _I64, _P]),
    "vlfb_pair_join": (C.c_int, [_P, _P, _I64, _P]),
    "vlfb_transpose2d": (C.c_int, [_P, _P, C.c_int, _I64, _I64, _I64, _P]),
    "vlfb_copy2d": (C.c_int, [_P, _I64, _P, _I64, C.c_int, _I64, _I64, _P]),
    "vlfb_zero_f32": (C.c_int, [_P, _I64, _P]),
    "vlfb_split_planes": (C.c_int, [_P, _P, C.c_int, _I64, _I64, _I64, C.c_int, _P]),
    "vlfb_weight_prep": (C.c_int, [_P, _P, _P, _P, C.c_int, _I64, _I64, _I64, _P]),
    "vlfb_weight_prep_batched": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "vlfb_pool_argmax_bytes": (C.c_int, [C.POINTER(PoolDesc)]),
    "vlfb_maxpool_fwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P, _P]),
    "vlfb_maxpool_bwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P, _P, _P, _P]),
    "vlfb_maxpool_relu_bwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P, _P, _P]),
    "vlfb_avgpool_fwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P]),
    "vlfb_avgpool_bwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P, _P, _P]),
    "vlfb_avgpool_bwd_two_term": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P, _P, _P]),
    "vlfb_softmax_fwd": (C.c_int, [_P, _P, C.c_int, _I64, _I64, C.c_float, _P]),
    "vlfb_softmax_bwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _I64, C.c_float, _P]),
    "vlfb_softmax_bwd_p32": (C.c_int, [_P, _P, _P, C.c_int, _I64, _I64, C.c_float, _P]),
    "vlfb_attn_scores_supported": (C.c_int, [C.c_int, _I64, _I64, _I64]),
    "vlfb_attn_scores_fwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _I64, _I64, _I64, C.c_float, _P]),
    "vlfb_attn_scores_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int, _I64, _I64, _I64, _I64, C.c_float, _P]),
    "vlfb_add": (C.c_int, [_P, _P, _P, _P, C.c_int, _I64, C.c_int, _P]),
    "vlfb_relu_fwd": (C.c_int, [_P, _P, C.c_int, _I64, _P]),
    "vlfb_relu_bwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _P]),
    "vlfb_colsum": (C.c_int, [_P, C.c_int, _I64, _I64, _I64, _P, C.c_int, _P]),
    "vlfb_bn_workspace_bytes": (_I64, [C.c_int, _I64, _I64]),
    "vlfb_bn_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.c_int, _I64, _I64, C.c_float, C.c_float, C.c_int, _P]),
    "vlfb_bn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.c_int, _I64, _I64, C.c_float, _P]),
    "vlfb_layernorm_fwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _I64, C.c_float, _P]),
    "vlfb_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int, _I64, _I64, _P]),
    "vlfb_dropout_fwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _I64, _I64, C.c_float, C.c_uint64, _P]),
    "vlfb_dropout_fwd_dev": (C.c_int, [_P, _P, _P, C.c_int, _I64, _I64, _I64, C.c_float, _P, _P]),
    "vlfb_dropout_bwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, C.c_float, _P]),
    "vlfb_fc_fwd": (C.c_int, [_P, C.c_int, _P, _P, _P, _I64, _I64, _I64, _P]),
    "vlfb_fc_bwd": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _I64, _I64, _I64, C.c_int, _P]),
    "vlfb_sigmoid_ce": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, C.c_float, _P]),
    "vlfb_softmax_ce": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, C.c_float, _P]),
    "vlfb_roi_align_max_fwd": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64,
                                         C.c_int, C.c_float, _P]),
    "vlfb_roi_align_max_bwd": (C.c_int, [_P, C.c_int, _P, _P, _P, _I64, _I64, _I64, _I64, _I64,
                                         C.c_int, C.c_float, _P]),
    "vlfb_roi_align_decisions": (C.c_int, [_P, _P, _I64, _I64, _I64, C.c_int, C.c_float, C.c_int, _P]),
    "vlfb_fbo_attn_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _I64, _I64, _I64, _I64, C.c_float, _P]),
    "vlfb_fbo_attn_fwd_shared": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _I64, _I64, _I64, _I64, C.c_float, _P, _I64, _P]),
    "vlfb_fbo_attn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _I64, _I64, _I64, _I64,
                                    C.c_float, _P]),
    "vlfb_sgd_update": (C.c_int, [_P, _P, _P, _I64, C.c_float, C.c_float, C.c_float, C.c_int, _P]),
    "vlfb_sgd_update_dev": (C.c_int, [_P, _P, _P, _I64, _P, C.c_float, C.c_float, C.c_int, _P]),
    "vlfb_store_scalars": (C.c_int, [_P, C.c_int, _P, _P]),
    "vlfb_scale_inplace": (C.c_int, [_P, _I64, C.c_float, _P]),
    "vlfb_clip_preprocess": (C.c_int, [C.POINTER(ClipDesc), _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "vlfb_lfb_bank_bytes": (_I64, [C.POINTER(LfbDesc)]),
    "vlfb_lfb_append": (C.c_int, [C.POINTER(LfbDesc), _P, _P, _P, C.c_int, _P, _I64, _P, _P]),
    "vlfb_lfb_sample_window": (C.c_int, [C.POINTER(LfbDesc), _P, _P, _P, _I64, C.c_int, C.c_int, C.c_uint64,
                                         _P, C.c_int, _P]),
    "vlfb_lfb_gather_slots": (C.c_int, [C.POINTER(LfbDesc), _P, _P, _P, _P, _I64, C.c_int, C.c_int, _P, C.c_int, _P]),
    "vlfb_lfb_sample_compact": (C.c_int, [C.POINTER(LfbDesc), _P, _P, _P, _I64, C.c_int, _P, C.c_int, _P]),
    "vlfb_lfb_sample_packed": (C.c_int, [C.POINTER(LfbDesc), _P, _P, _P, _I64, C.c_int, C.c_int, _P, C.c_int, _P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGS))

_lib = None


def lib():
    """Load libvlfb_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VlfbError(
                "libvlfb_hip.so is missing at %s -- run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or make -C video-long-term-feature-banks_amd/csrc). There is no "
                "CPU/PyTorch fallback for the hot path." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(rc, what):
    if rc != 0:
        raise VlfbError("%s failed (%d): %s" % (what, rc, lib().vlfb_last_error().decode()))


def ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# When a list, every stream-ordered call OF THE RECORDING THREAD is also appended to it as (function, arguments incl. the
# stream, name): Engine.train_step records one step this way and replays the list (Engine.STEP_TRACE).  Calls of other
# threads (a loader thread preprocessing the next clip, a bank sampler) are never recorded: replaying them with their frozen
# pointers every step would corrupt memory silently.  Use trace_begin / trace_end.
TRACE = None
_TRACE_THREAD = None


def trace_begin():
    """start recording the calls of THIS thread; returns the list"""
    global TRACE, _TRACE_THREAD
    import threading
    if TRACE is not None:
        raise VlfbError("a step is already being recorded (hip.TRACE)")
    _TRACE_THREAD = threading.get_ident()
    TRACE = []
    return TRACE


def trace_end():
    global TRACE, _TRACE_THREAD
    TRACE, _TRACE_THREAD = None, None


def tracing():
    """the list being recorded if the CALLING thread is the one recording, else None"""
    if TRACE is None:
        return None
    import threading
    return TRACE if threading.get_ident() == _TRACE_THREAD else None


def freeze_args(fn, args):
    """convert plain ints / floats to the ctypes objects of the signature once, so a replayed call converts nothing"""
    types = getattr(fn, "argtypes", None)
    if not types:
        return args
    return tuple(t(a) if isinstance(a, (int, float)) and issubclass(t, C._SimpleCData) else a
                 for t, a in zip(types, args))


def call(name, *args):
    """Call a status-returning entry point, appending the current HIP stream."""
    fn = getattr(lib(), name)
    args = args + (stream(),)
    rec = tracing()
    if rec is not None:
        rec.append((fn, freeze_args(fn, args), name))
    _check(fn(*args), name)


def conv_desc(**kw):
    d = ConvDesc()
    lib().vlfb_conv_desc_init(C.byref(d))
    for k, v in kw.items():
        if not hasattr(d, k):
            raise AttributeError("ConvDesc has no field %s" % k)
        setattr(d, k, v)
    return d


def conv_workspace_bytes(d):
    n = lib().vlfb_conv_workspace_bytes(C.byref(d))
    if n < 0:
        raise VlfbError("conv_workspace_bytes: %s" % lib().vlfb_last_error().decode())
    return n


def conv_plan(d):
    """kernel family / tile / splits the library runs for this descriptor (vlfb_conv_plan_describe)"""
    buf = C.create_string_buffer(128)
    _check(lib().vlfb_conv_plan_describe(C.byref(d), buf, 128), "vlfb_conv_plan_describe")
    return buf.value.decode()


WS_CONV, WS_MAXPOOL_ARGMAX, WS_FBO_ATTN_BWD, WS_ATTN_SCORES, WS_BN = 0, 1, 2, 3, 4


def query_workspace(op, arg):
    """bytes of caller-owned scratch for an entry point: arg = a ConvDesc / PoolDesc or a tuple of dimensions"""
    if isinstance(arg, (tuple, list)):
        arg = (C.c_int64 * len(arg))(*arg)
    n = lib().vlfb_query_workspace(op, C.cast(C.pointer(arg) if not isinstance(arg, C.Array) else arg, C.c_void_p))
    if n < 0:
        raise VlfbError("query_workspace: %s" % lib().vlfb_last_error().decode())
    return n


def conv_flops(d):
    """ALGORITHMIC flops (2*MAC of the convolution / GEMM the launch stands for; padding of the
    packed stem and masked-out taps of strided dgrads do not count)"""
    taps = d.kt * d.kh * d.kw
    if d.dt == 0:            # "mix" DGRAD with two-term weights: the doubled tap dimension is not algorithmic work
        taps //= 2
    batch = max(d.batch, 1)
    if d.mode == DGRAD:      # rows = conv input positions; source = conv output (N,Ts,Hs,Ws,Cs=Cout)
        return 2.0 * d.N * d.Ts * d.Hs * d.Ws * d.Cs * taps * d.Cn * batch
    cin = 3 if d.pack_w else d.Cs
    return 2.0 * d.N * d.Tr * d.Hr * d.Wr * d.Cn * taps * cin * batch


def conv_bytes(d, has_r=False, has_mask=False):
    """ALGORITHMIC HBM bytes of a launch: every operand read once, the output written once"""
    es = 4 if (d.dtype == F32 or d.math == MATH_F16X3) else 2          # (two fp16 planes = 4 bytes per value)
    os_ = 4 if (d.out_dtype == F32 or d.math not in (MATH_NATIVE, MATH_F16W2)) else 2
    batch = max(d.batch, 1)
    taps = d.kt * d.kh * d.kw
    k = d.kt * d.kh * d.pack_w * 4 if d.pack_w else taps * d.Cs
    m = d.N * d.Tr * d.Hr * d.Wr
    src = d.N * d.Ts * d.Hs * d.Ws * d.Cs * es
    if d.mode == WGRAD:
        return batch * (src + m * d.Cn * es + d.Cn * k * os_)
    out = m * d.Cn
    wterms = 2 if d.math == MATH_F16W2 else 1
    return batch * (src + d.Cn * k * es * wterms + out * os_ + (out * es if has_r else 0) + (out * es if has_mask else 0))


def conv_tag(d):
    return "%s M=%d Cs=%d Cn=%d k%d%d%d s%d%d%d d%d b%d src%dx%dx%d" % (
        ("fprop", "dgrad", "wgrad")[d.mode], d.N * d.Tr * d.Hr * d.Wr, d.Cs, d.Cn, d.kt, d.kh, d.kw,
        d.st, d.sh, d.sw, d.dh, max(d.batch, 1), d.Ts, d.Hs, d.Ws)


def conv_family(d):
    """(family, MFMA instructions per algorithmic product) of a launch, for the roofline records of bench.py: the kernels
    that run a split-bf16 product issue 3 (6) MFMAs per product, an fp16 DGRAD with two-term weights (the "mix" path: doubled
    outermost tap dimension of dilation 0, or MATH_F16W2) 2, everything else 1.  Families: nt_split / tn_split (csrc/vlfb_gemm_split.hip),
    nt_16 / tn_16 (the 16-bit families), nt_f32 / tn_f32 (exact-fp32 MFMA)."""
    side = "tn" if d.mode == WGRAD else "nt"
    if d.math in (MATH_BF16X3, MATH_BF16X6):
        return side + "_split", (6 if (d.math == MATH_BF16X6 and d.mode == FPROP) else 3)
    if d.math == MATH_F16X3:
        return side + "_pair", 3
    if d.dtype == F32:
        return side + "_f32", 1
    return side + "_16", (2 if (d.mode == DGRAD and ((d.dt == 0 and d.kt == 2) or d.math == MATH_F16W2)) else 1)



# When a list, every conv_run is bracketed by HIP events on the launch stream and appended as
# (mode, flops, start_event, end_event); bench.py uses it for the live roofline measurement.
PROFILE = None


def conv_run(d, A, B, P, O, bias=None, rowscale=None, R=None, mask=None, workspace=None, O_planes=None, dbias=None,
             R_lo=None, O_lo=None):
    """dbias (WGRAD descriptors with wgrad_bias = 1): destination of the bias gradient the launch produces as well;
    R_lo / O_lo: low terms of a two-term residual / output (vlfb_conv_args)"""
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    prof = PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if R_lo is not None or O_lo is not None:
        fn = lib().vlfb_conv_run_args
        a = ConvArgs(ptr(A), ptr(B), ptr(P), ptr(O), ptr(bias), ptr(rowscale), ptr(R), ptr(mask), ptr(workspace), ws_bytes,
                     ptr(O_planes), ptr(dbias), ptr(R_lo), ptr(O_lo))
        args = (C.byref(d), C.byref(a), stream())   # (a recorded step keeps `a` alive through the byref object in its call list)
    elif dbias is not None:
        fn = lib().vlfb_conv_run_wgrad_bias
        args = (C.byref(d), ptr(A), ptr(P), ptr(O), ptr(dbias), ptr(rowscale), ptr(workspace), ws_bytes, stream())
    else:
        fn = lib().vlfb_conv_run_planes
        args = (C.byref(d), ptr(A), ptr(B), ptr(P), ptr(O), ptr(bias), ptr(rowscale), ptr(R), ptr(mask), ptr(workspace),
                ws_bytes, ptr(O_planes), stream())
    rec = tracing()
    if rec is not None:
        rec.append((fn, freeze_args(fn, args), "vlfb_conv_run"))
    rc = fn(*args)
    if prof is not None:
        e1.record()
        prof.append((d.mode, conv_flops(d), e0, e1, conv_tag(d), conv_bytes(d, R is not None, mask is not None)) + conv_family(d))
    _check(rc, "vlfb_conv_run")


def pool_desc(dtype, N, Ti, Hi, Wi, Cc, To, Ho, Wo, k, s, p):
    d = PoolDesc()
    d.dtype, d.N, d.Ti, d.Hi, d.Wi, d.C, d.To, d.Ho, d.Wo = dtype, N, Ti, Hi, Wi, Cc, To, Ho, Wo
    d.kt, d.kh, d.kw = k
    d.st, d.sh, d.sw = s
    d.pt, d.ph, d.pw = p
    return d
