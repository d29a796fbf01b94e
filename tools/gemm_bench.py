#!/usr/bin/env python3
"""Times the projection GEMM kernels against torch (rocBLAS/hipBLASLt) at the B=512 ESOL shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fragnet_amd import _lib, ops
from fragnet_amd.plan import _stream_ptr

dev = "cuda:0"

def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / iters

for M, K in ((26492, 128), (13334, 128), (2500, 128), (26492, 17), (13334, 167)):
    x = torch.randn(M, K, device=dev); w = torch.randn(128, K, device=dev) * 0.1; b = torch.randn(128, device=dev)
    g = torch.randn(M, 128, device=dev)
    st = _stream_ptr(torch.device(dev))
    bt = w.t().contiguous(); y = torch.empty(M, 128, device=dev)
    ws = torch.empty(_lib.load().fn_linear128_wgrad_ws(M, K), device=dev); gw = torch.empty_like(w); gb = torch.empty(128, device=dev)
    t_f = timeit(lambda: _lib.call("fn_linear128_f32", x.data_ptr(), K, bt.data_ptr(), b.data_ptr(), y.data_ptr(), M, None, st))
    t_w = timeit(lambda: _lib.call("fn_linear128_wgrad_f32", g.data_ptr(), x.data_ptr(), K, M, ws.data_ptr(), gw.data_ptr(), gb.data_ptr(), st))
    t_tf = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t_tw = timeit(lambda: (g.t() @ x, g.sum(0)))
    fl = 2 * M * K * 128 / 1e6
    print(f"M={M} K={K}: fwd {t_f:.1f} us ({fl/t_f:.1f} GF/ms... {fl/t_f/1e3:.1f} TF) torch {t_tf:.1f} | wgrad {t_w:.1f} us ({fl/t_w/1e3:.1f} TF) torch {t_tw:.1f}")
