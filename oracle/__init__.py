"""CPU oracle of the R50/R101-I3D-NL + long-term-feature-bank hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under video-long-term-feature-banks_amd/ may import this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as
the checker / the timed CPU baseline -- never as the thing shipped.

PARITY UNPINNED BY THE REFERENCE: facebookresearch/video-long-term-feature-banks ships no tests,
no golden vectors and cannot run here (Caffe2 is gone, Python 2, CUDA-only builder, AffineNd has
no CPU kernel: caffe2_customized_ops/video/affine_nd_op.h:42-46).  The operator arithmetic lives
in Caffe2 (pytorch/pytorch `caffe2/` tree, version unpinned by INSTALL.md:24-31); it is restated
here from its published semantics with torch-CPU functional ops in fp64 /
fp32 plus a hand-written RoIAlign and sigmoid cross-entropy.  What pins this oracle instead:
structural checks against the reference's own builders (parameter names/shapes/counts, MAC
counts), two independent RoIAlign implementations that must agree bit-exactly on every integer
decision, closed forms of the losses and committed golden vectors produced by oracle/make_golden.py
(tests/golden/) -- all in tests/test_oracle.py; the one operator the reference ships itself, AffineNd,
is held bit for bit to the reference's own .cu compiled in place (oracle/build_ref.py,
tests/test_ref_affine_gpu.py).
"""
