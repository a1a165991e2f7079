#!/bin/bash
O=gpurun_out/${1:-r3h}; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "test_forward_backward_matches_oracle and not bf16" > $O/t_small.log 2>&1; echo "small rc=$?"
VLFB_PARITY_DIR=$O timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -s -k "test_full_size_clip_matches_oracle" > $O/t_full.log 2>&1; echo "full rc=$?"
grep "identical\|Error\|assert" $O/t_small.log | head -20; tail -3 $O/t_small.log; grep "same parameter\|parameter gradients:\|^==" $O/t_full.log; tail -5 $O/t_full.log
