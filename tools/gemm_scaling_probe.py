#!/usr/bin/env python3
"""Time of the K = 128 projection GEMM (fn_linear128_f32, one 64 x 64 tile per workgroup) against the row count: separates the
launch floor from the per-round cost (1024 workgroups are resident at once).  Back-to-back launches, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fragnet_amd import _lib
from fragnet_amd.plan import _stream_ptr

dev = "cuda:0"
st = _stream_ptr(torch.device(dev))


def timeit(fn, iters=100):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / iters


K = 128
w = torch.randn(128, K, device=dev) * 0.1
bt = w.t().contiguous()
bias = torch.randn(128, device=dev)
for M in (64, 1024, 4096, 8192, 13334, 16384, 26492, 32768, 45056, 65536, 131072):
    x = torch.randn(M, K, device=dev)
    y = torch.empty(M, 128, device=dev)
    t = timeit(lambda: _lib.call("fn_linear128_f32", x.data_ptr(), K, bt.data_ptr(), bias.data_ptr(), y.data_ptr(), M, None, st))
    tiles = 2 * ((M + 63) // 64)
    print(f"M={M:7d} workgroups={tiles:5d} rounds={tiles / 1024:5.2f}  {t:7.2f} us  {2 * M * K * 128 / t / 1e6:6.1f} TFLOP/s  "
          f"{(M * K + M * 128) * 4 / t / 1e3:7.1f} GB/s")
