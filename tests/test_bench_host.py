"""Host logic of the measurement code (no GPU): how bench.py classifies a launch into a kernel family and prices it, and how
tools/pmc_traffic.py maps rocprofv3 kernel names onto the same families."""
import collections
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_families_and_mfma_instructions_per_product():
    from vlfb import hip
    rows = dict(N=1, Tr=4, Hr=8, Wr=8, Ts=4, Hs=8, Ws=8, Cs=64, Cn=64, kt=1, kh=1, kw=1)
    d = lambda **kw: hip.conv_desc(**dict(rows, **kw))
    assert hip.conv_family(d(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16)) == ("nt_16", 1)
    assert hip.conv_family(d(mode=hip.WGRAD, dtype=hip.BF16, out_dtype=hip.F32)) == ("tn_16", 1)
    # the `mix` DGRAD with two-term fp16 weights: a doubled outermost tap dimension of dilation 0 -> two MFMAs per product,
    # and the doubled taps are not algorithmic work
    w2 = d(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, kt=2, dt=0)
    assert hip.conv_family(w2) == ("nt_16", 2)
    assert hip.conv_flops(w2) == hip.conv_flops(d(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16))
    # ... or the two terms interleaved per 64-channel k-tile (MATH_F16W2): the same two MFMAs per product on the conv's own
    # geometry; the weight operand is twice as many bytes, the fp16 output is not
    wi = d(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16W2)
    plain = d(mode=hip.DGRAD, dtype=hip.F16, out_dtype=hip.F16)
    assert hip.conv_family(wi) == ("nt_16", 2) and hip.conv_flops(wi) == hip.conv_flops(plain)
    assert hip.conv_bytes(wi) == hip.conv_bytes(plain) + 64 * 64 * 2 and hip.conv_plan(wi) == "nt f16 128x64 ut w2 pre"
    # split-bf16 products on fp32 storage: three MFMAs per product (six in a six-term forward)
    assert hip.conv_family(d(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, math=hip.MATH_BF16X3)) == ("nt_split", 3)
    assert hip.conv_family(d(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32, math=hip.MATH_BF16X6)) == ("nt_split", 6)
    # two fp16 planes per operand (the `mix` forward): three fp16 MFMAs per product, 4 bytes per stored value
    h2 = d(mode=hip.FPROP, dtype=hip.F16, out_dtype=hip.F16, math=hip.MATH_F16X3, a_pstride=4 * 8 * 8 * 64)
    assert hip.conv_family(h2) == ("nt_pair", 3) and hip.conv_bytes(h2) == hip.conv_bytes(d(mode=hip.FPROP, dtype=hip.F32, out_dtype=hip.F32))
    assert hip.conv_family(d(mode=hip.WGRAD, dtype=hip.F32, out_dtype=hip.F32, math=hip.MATH_BF16X3)) == ("tn_split", 3)
    assert hip.conv_family(d(mode=hip.DGRAD, dtype=hip.F32, out_dtype=hip.F32)) == ("nt_f32", 1)


def test_roofline_record_prices_a_family_against_the_mfma_peak_per_product():
    import bench
    fam = collections.OrderedDict()
    fam["nt_split"] = {"flops": 3.0e12, "mfma_flops": 9.0e12, "sec": 0.018, "n": 100, "bytes": 2.0e10, "att": 0.006}
    fam["nt_16"] = {"flops": 2.0e12, "mfma_flops": 3.5e12, "sec": 0.010, "n": 50, "bytes": 1.0e10, "att": 0.004}
    fam["tn_16"] = {"flops": 3.0e12, "mfma_flops": 3.0e12, "sec": 0.011, "n": 60, "bytes": 1.0e10, "att": 0.003}
    assert bench.dominant(fam, "nt") == "nt_split" and bench.dominant(fam, "tn") == "tn_16" and bench.dominant({}, "nt") is None
    r = bench.roof_record("nt_split", fam, {"nt_split": 4.1e8}, "test")
    assert r["kernel"].startswith("gemm_nt_sp_kernel") and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 3.0e12 / 0.018 / 1e12) < 0.01 and abs(r["peak"] - 2500.0 / 3) < 0.1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["mfma_per_product"] == 3.0
    assert r["traffic"] == 4.1e8 and r["launches_per_step"] == 100 and abs(r["frac_of_attainable"] - 0.006 / 0.018) < 1e-3
    assert abs(r["traffic_ratio"] - 4.1e8 / (2.0e10 / 100)) < 1e-3          # PMC bytes / algorithmic bytes per launch
    r = bench.roof_record("nt_16", fam, {}, "none", brief=True)
    assert r["traffic"] is None and r["traffic_ratio"] is None and abs(r["mfma_per_product"] - 1.75) < 1e-9 and abs(r["peak"] - 2500.0 / 1.75) < 0.1
    assert "source" not in r and bench.roof_record(None, fam, {}, "none") is None


def test_pmc_traffic_maps_kernel_names_onto_the_bench_families():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic as pt
    names = {
        "void vlfb::(anonymous namespace)::gemm_nt_sp_kernel<2, 128, true, false, false, false, false, 1>(vlfb::GP)": "nt_split",
        "void vlfb::(anonymous namespace)::gemm_nt_pl_kernel<2, 64, false, false>(vlfb::GP)": "nt_split",
        "vlfb::(anonymous namespace)::gemm_skinny_nt_sp_kernel(vlfb::GP)": "nt_split",
        "void vlfb::(anonymous namespace)::stem_fprop_pair_kernel<7, 7>(vlfb::GP, int, int, int)": "nt_pair",
        "void vlfb::(anonymous namespace)::stem_fprop_kernel<vlfb::f16_t, 7, 7>(vlfb::GP, int, int, int)": "nt_16",
        "void vlfb::(anonymous namespace)::gemm_skinny_nt_kernel<vlfb::f16_t, float>(vlfb::GP)": "nt_16",
        "void vlfb::(anonymous namespace)::gemm_tn_sp_kernel<128, 128, true, false, 8>(vlfb::GP)": "tn_split",
        "void vlfb::(anonymous namespace)::gemm_tn_tr_kernel<vlfb::bf16_t, float, 128, 128, false, false, 8, true>(vlfb::GP)": "tn_split",
        "void vlfb::(anonymous namespace)::gemm_tn_tr_kernel<vlfb::f16_t, float, 128, 128, false, false, 8, false>(vlfb::GP)": "tn_16",
        "void vlfb::(anonymous namespace)::gemm_nt_kernel<vlfb::f16_t, vlfb::f16_t, 128, 128, false, true, false, 128, false, 8, 2, true, false, false>(vlfb::GP)": "nt_16",
        "void vlfb::(anonymous namespace)::gemm_nt_kernel<vlfb::f16_t, vlfb::f16_t, 128, 128, false, true, false, 128, false, 8, 2, true, false, true>(vlfb::GP)": "nt_16",      # (W2I: the shared-tile two-term DGRAD)
        "void vlfb::(anonymous namespace)::gemm_nt_kernel<vlfb::f16_t, vlfb::f16_t, 128, 128, false, false, false, 128, false, 8, 2, true, true, false>(vlfb::GP)": "nt_pair",
        "void vlfb::(anonymous namespace)::gemm_nt_kernel<vlfb::f16_t, float, 128, 128, true, false, false, 128, false, 8, 2, false, true, false>(vlfb::GP)": "nt_pair",
        "void vlfb::(anonymous namespace)::gemm_nt_kernel<float, float, 128, 128, true, false, false, 128, false, 4, 2, false, false, false>(vlfb::GP)": "nt_f32",
        "void vlfb::(anonymous namespace)::gemm_nt8_kernel<vlfb::f16_t, vlfb::f16_t, 256, 2, false, 98, false>(vlfb::GP)": "nt_16",
        "void vlfb::(anonymous namespace)::gemm_nt8_kernel<vlfb::f16_t, vlfb::f16_t, 256, 1, false, 98, true>(vlfb::GP)": "nt_pair",
        "void vlfb::(anonymous namespace)::stem_wgrad_kernel<vlfb::f16_t>(vlfb::GP)": "tn_16",
        "void vlfb::(anonymous namespace)::wgrad_rows_fat_kernel<vlfb::f16_t, 4, 3>(vlfb::GP)": "tn_16",
        "void vlfb::(anonymous namespace)::conv_rows64_kernel<vlfb::f16_t, 1>(vlfb::GP)": "nt_16",
        "vlfb::sgd_kernel(float*, float*, float*, long long, float, float, float, int, float const*)": None,
        "half_copy_kernel(float const*, unsigned short*, long long, long long)": None,
    }
    for n, want in names.items():
        assert pt.family(n) == want, (n, pt.family(n), want)
