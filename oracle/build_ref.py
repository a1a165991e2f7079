"""TEST INFRASTRUCTURE: builds oracle/_ref/libref_affine_nd.so from the reference's OWN sources.

The only native code of the reference on the hot path is the AffineNd operator
(caffe2_customized_ops/video/affine_nd_op.{h,cu}).  It cannot be built the reference's way (it is a
Caffe2 plug-in and Caffe2 no longer exists), but the two files only use a sliver of Caffe2, so this
recipe compiles them UNMODIFIED, from where they lie under /root/reference, against the mini-Caffe2 in
oracle/ref_shim/ with hipcc for gfx950.  Output goes to oracle/_ref/ only (git-ignored, NOT
gpurun-ignored: the .so travels to the GPU box, /root/reference does not).  No reference source is
copied into the repository.

    python oracle/build_ref.py            # no-op with a message when /root/reference is absent
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_OPS = "/root/reference/caffe2_customized_ops"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libref_affine_nd.so")


def build(verbose=True):
    src = os.path.join(REF_OPS, "video", "affine_nd_op.cu")
    if not os.path.exists(src):
        if verbose:
            print("oracle/build_ref: %s not present (GPU box?) -- keeping the prebuilt %s" % (src, OUT))
        return os.path.exists(OUT)
    os.makedirs(OUT_DIR, exist_ok=True)
    driver = os.path.join(HERE, "ref_affine_driver.cu")
    newest = max(os.path.getmtime(p) for p in (src, driver, os.path.join(REF_OPS, "video", "affine_nd_op.h"),
                                                os.path.join(HERE, "ref_shim", "caffe2", "core", "context_gpu.h")))
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return True
    env = dict(os.environ)
    if "/opt/rocm/bin" not in env.get("PATH", ""):
        env["PATH"] = "/opt/rocm/bin:" + env.get("PATH", "")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I" + os.path.join(HERE, "ref_shim"), "-I" + REF_OPS, driver, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, env=env)
    return True


if __name__ == "__main__":
    sys.exit(0 if build() else 1)
