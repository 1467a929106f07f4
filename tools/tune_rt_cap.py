#!/usr/bin/env python3
"""Time the direct MFMA kernel on a geometry with 5 overlapping frames (NOFF = 6 instances) under a register-tile cap:
   for c in 3 2 1; do LEAF_FUSED_RT_CAP=$c python tools/tune_rt_cap.py; done     (round 2: 1.18 / 0.88 / 0.74 ms -> cap 1)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from leaf_pytorch_amd import _native
from tests.helpers import make_leaf
dev="cuda:0"
F,K,hop,T,B=64,321,80,16000,256
m=make_leaf(F,K,hop,True,None,dev); m._algo=_native.ALGO_MFMA
x=torch.randn(B,1,T,device=dev)
with torch.no_grad():
    for _ in range(20): m(x)
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): m(x)
    e.record(); e.synchronize()
print("cap",os.environ.get("LEAF_FUSED_RT_CAP"),"ms",s.elapsed_time(e)/20)
