"""Time fn_linear128_f32 for layer-0 widths: does the 4-byte-aligned 167-wide row layout cost against an aligned 168?"""
import sys, time
import torch
sys.path.insert(0, ".")
from fragnet_amd import _lib
from fragnet_amd.plan import _stream_ptr
dev = torch.device("cuda:0")
M = 13872
for K in (160, 164, 167, 168, 128, 17, 16):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(128, K, device=dev)
    bt = torch.empty(192 * 128, device=dev)
    b = torch.randn(128, device=dev)
    y = torch.empty(M, 128, device=dev)
    st = _stream_ptr(dev)
    _lib.call("fn_transpose_w_f32", w.data_ptr(), K, bt.data_ptr(), st)
    for _ in range(5):
        _lib.call("fn_linear128_f32", x.data_ptr(), K, bt.data_ptr(), b.data_ptr(), y.data_ptr(), M, None, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 50
    for _ in range(n):
        _lib.call("fn_linear128_f32", x.data_ptr(), K, bt.data_ptr(), b.data_ptr(), y.data_ptr(), M, None, st)
    e1.record(); torch.cuda.synchronize()
    ref = x @ w.t() + b
    print(f"K={K:4d}  {e0.elapsed_time(e1) / n * 1e3:7.2f} us   max err {float((y - ref).abs().max()):.2e}")
