"""upper bound of what a 1-bit ReLU mask could save: run the bench with the backward masks dropped
(WRONG gradients, timing only)"""
import sys, runpy
sys.path.insert(0, 'video-long-term-feature-banks_amd/lib')
from vlfb import hip
_orig = hip.conv_run
def patched(d, A, B, P, O, bias=None, rowscale=None, R=None, mask=None, workspace=None):
    return _orig(d, A, B, P, O, bias=bias, rowscale=rowscale, R=R, mask=None, workspace=workspace)
hip.conv_run = patched
sys.argv = ['bench.py', '--steps', '8', '--warmup', '2', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
