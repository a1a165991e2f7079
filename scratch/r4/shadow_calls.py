"""CPU tool: run Engine.forward() + backward() of a dry-run plan with hip.call / hip.conv_run replaced by recorders and dump the
kernel-call sequence (names, dtype codes, sizes; pointers anonymised).  Two uses: (1) prove that a Python-side change left the
default paths untouched (diff the dumps of two trees -- how Engine.MIX_HEAD_F32 was added without GPU time), (2) read what a
switch does to the sequence (VLFB_MIX_HEAD_F32=1 python scratch/r4/shadow_calls.py out.json).
Usage: python scratch/r4/shadow_calls.py out.json"""
import sys, json, os
sys.path.insert(0,'video-long-term-feature-banks_amd/lib'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from vlfb import hip
from test_lowering import plan
calls=[]
def norm(a):
    import ctypes
    if torch.is_tensor(a): return "T%d" % a.numel()
    if isinstance(a,(int,)) and a > 1<<20: return "P"
    if isinstance(a,float): return round(a,9)
    if hasattr(a,'_obj') or 'CArgObject' in type(a).__name__: return "ref"
    if a is None or isinstance(a,(int,str,bool)): return a
    return type(a).__name__
def fake_call(name,*args): calls.append([name]+[norm(a) for a in args]); return 0
def fake_conv_run(d,*args,**kw): calls.append(["conv_run", hip.conv_tag(d)]+[norm(a) for a in args]+sorted((k,norm(v)) for k,v in kw.items())); return 0
hip.call=fake_call; hip.conv_run=fake_conv_run
hip.ptr = lambda t: (None if t is None else (1<<30))
import vlfb.engine as E
E.hip.call=fake_call; E.hip.conv_run=fake_conv_run; E.hip.ptr=hip.ptr
out={}
for preset,dtype in [("ava_r50_lfb_nl","fp16"),("ava_r50_lfb_nl","mix"),("charades_r50_baseline","mix"),("ava_r50_lfb_nl","split"),("ava_r50_lfb_max","bf16")]:
    cfg,m,eng = plan(preset, ("NUM_GPUS",1,"TRAIN.BATCH_SIZE",2,"TRAIN.VIDEO_LENGTH",8,"TRAIN.CROP_SIZE",64), dtype=dtype)
    del calls[:]
    try:
        eng.forward(); eng.backward()
        out[preset+":"+dtype]=list(calls)
        print(preset,dtype,"ok",len(calls))
    except Exception as e:
        import traceback; traceback.print_exc()
        print(preset,dtype,"FAIL",repr(e)[:200])
json.dump(out, open(sys.argv[1],"w"))
