#!/usr/bin/env python3
"""One projection GEMM shape in a loop (for rocprofv3 --pmc): python tools/gemm_one.py M K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fragnet_amd import _lib
from fragnet_amd.plan import _stream_ptr
M, K = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    _lib.call("fn_set_tuning", 1, int(sys.argv[3]))      # FN_TUNE_GEMM_SLOTS
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); bt = torch.randn(K, 128, device=dev) * 0.1; b = torch.randn(128, device=dev)
y = torch.empty(M, 128, device=dev)
st = _stream_ptr(dev)
for _ in range(30):
    _lib.call("fn_linear128_f32", x.data_ptr(), K, bt.data_ptr(), b.data_ptr(), y.data_ptr(), M, None, st)
torch.cuda.synchronize()
