"""The C-ABI boundary: libvlfb_hip.so loads (no GPU needed) and exports exactly what
include/vlfb.h declares; the ctypes binding covers every declared entry point."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vlfb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vlfb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ["vlfb_affine_nd_fwd", "vlfb_affine_nd_bwd", "vlfb_conv_run", "vlfb_maxpool_fwd",
                 "vlfb_softmax_fwd", "vlfb_roi_align_max_fwd", "vlfb_fbo_attn_fwd", "vlfb_sgd_update",
                 "vlfb_sigmoid_ce", "vlfb_layernorm_fwd", "vlfb_dropout_fwd", "vlfb_last_error"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    import ctypes
    from vlfb import hip
    if not os.path.exists(hip.LIB_PATH):
        pytest.skip("libvlfb_hip.so not built (run __graft_entry__.build())")
    handle = ctypes.CDLL(hip.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(handle, s)]
    assert not missing, "declared in include/vlfb.h but not exported: %r" % missing


def test_binding_covers_the_header():
    from vlfb import hip
    declared = set(declared_symbols())
    bound = set(hip.EXPORTED_SYMBOLS)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_host_side_argument_checks_need_no_gpu():
    """error behaviour of the boundary: bad descriptors are rejected with a message, not a crash"""
    from vlfb import hip
    if not os.path.exists(hip.LIB_PATH):
        pytest.skip("libvlfb_hip.so not built")
    d = hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, out_dtype=hip.BF16, N=1, Wr=16, Ws=16, Cs=12, Cn=8)
    with pytest.raises(hip.VlfbError, match="multiple of 8"):
        hip.conv_workspace_bytes(d)
    d = hip.conv_desc(mode=hip.WGRAD, dtype=hip.BF16, out_dtype=hip.F32, N=1, Tr=4, Hr=8, Wr=8, Ts=4, Hs=8,
                      Ws=8, Cs=64, Cn=64, kh=3, kw=3, ph=1, pw=1, splits=4)
    assert hip.conv_workspace_bytes(d) == 4 * 64 * 9 * 64 * 4
    # one query for every caller-owned scratch buffer (vlfb_query_workspace)
    assert hip.query_workspace(hip.WS_CONV, d) == hip.conv_workspace_bytes(d)
    pd = hip.pool_desc(hip.BF16, 2, 4, 16, 16, 64, 4, 8, 8, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    assert hip.query_workspace(hip.WS_MAXPOOL_ARGMAX, pd) == 2 * 4 * 8 * 8 * 64 * hip.lib().vlfb_pool_argmax_bytes(pd)
    assert hip.query_workspace(hip.WS_FBO_ATTN_BWD, (33, 60)) == 33 * 60 * 4
    assert hip.query_workspace(hip.WS_ATTN_SCORES, (8, 3136, 784)) == 8 * 3136 * 784 * 4
    with pytest.raises(hip.VlfbError, match="unknown op"):
        hip.query_workspace(17, (1,))
    assert hip.lib().vlfb_dtype_size(hip.BF16) == 2 and hip.lib().vlfb_dtype_size(hip.F32) == 4
    assert hip.conv_flops(d) == 2.0 * 4 * 8 * 8 * 64 * 9 * 64


def test_the_step_recorder_only_sees_the_recording_thread():
    """hip.TRACE is what Engine.train_step replays with frozen pointers: a call of ANOTHER thread (a loader thread
    preprocessing the next clip, a bank sampler) must never land in it"""
    import threading
    from vlfb import hip
    rec = hip.trace_begin()
    try:
        assert hip.tracing() is rec
        seen = []
        t = threading.Thread(target=lambda: seen.append(hip.tracing()))
        t.start()
        t.join()
        assert seen == [None]
        import pytest
        with pytest.raises(hip.VlfbError):
            hip.trace_begin()               # one recording at a time
    finally:
        hip.trace_end()
    assert hip.tracing() is None and hip.TRACE is None
