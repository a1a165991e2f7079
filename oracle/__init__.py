"""CPU oracle of the R50/R101-I3D-NL + long-term-feature-bank hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under video-long-term-feature-banks_amd/ may import this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as
the checker / the timed CPU baseline -- never as the thing shipped.

WHAT THE REFERENCE PINS, AND WHAT IT CANNOT.  facebookresearch/video-long-term-feature-banks ships no tests and no
golden vectors, and its operator arithmetic lives in Caffe2 (pytorch/pytorch `caffe2/` tree, version unpinned by
INSTALL.md:24-31), which is gone: the network cannot run here.  But most of the reference is Python that never touches
Caffe2 arithmetic, and that part IS executed, unmodified, from /root/reference in the build container to produce
committed fixtures (generators beside this file; the fixtures travel to the GPU box, the reference does not):
  * oracle/make_ref_graph_golden.py -> tests/golden/ref_graphs.json.gz: lib/models/{resnet_video,resnet_helper,
    nonlocal_helper,lfb_helper,head_helper}.py and the composites of model_builder_video.py run against a recording
    model helper (oracle/graph_recorder.py) -- every operator call with its arguments, for all 26 configs (train / test /
    LFB-extraction graphs) and the option switches; plus lib/core/config.py's defaults and the effective configuration of
    every YAML.  tests/test_ref_graph.py holds THIS repo's builders, config defaults and presets to them, call for call.
  * oracle/make_ref_aux_golden.py -> tests/golden/ref_aux.npz: lr_policy.py and the ModelBuilder's lr-update / momentum-
    correction methods over a run, misc.py, config.py's merge_dicts / cfg_from_list rules, tools/lfb_loader.py and the
    datasets' LFB samplers, data_input_helper.py / image_processor.py (clip + box preprocessing), checkpoints.py (BN fold,
    inflation, classifier rule, momentum policy, save format, checkpoint discovery, the start-of-training policy),
    metrics.py's multi-crop merge, add_parameter_update_ops -- on seeded synthetic inputs.  tests/test_ref_aux.py holds
    the product's host logic and oracle/{lfb,preprocess,multicrop}.py to them, bit for bit; tests/test_ref_lfb_gpu.py
    holds the feature-bank kernels to the same banks and samples.
  * AffineNd, the one operator the reference ships itself, is held bit for bit to the reference's own .cu compiled in
    place (oracle/build_ref.py, tests/test_ref_affine_gpu.py).
STILL PARITY-UNPINNED: the arithmetic of the Caffe2 operators (Conv, MaxPool, AveragePool, BatchMatMul, Softmax,
RoIAlign, LayerNorm, Dropout, SpatialBN, FC, the two losses, MomentumSGDUpdate) -- restated in oracle/model.py from their
published semantics with torch-CPU functional ops in fp64 / fp32 plus a hand-written RoIAlign and
sigmoid cross-entropy -- and OpenCV's INTER_LINEAR resize (oracle/preprocess.py; not installed here).  What stands in for
a pin there: two independent RoIAlign implementations that must agree bit-exactly on every integer decision, closed forms
of the losses, MAC counts, and golden vectors of the oracle itself (oracle/make_golden.py, tests/golden/*.npz) that detect
drift -- all in tests/test_oracle.py.
"""
