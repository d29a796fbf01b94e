#!/usr/bin/env python3
"""Probe: what does an eager kernel between two hipGraph replays cost on this stack?  The training step is [eager staging launch] +
[graph replay]; the kernel trace shows ~8.6 us without a running kernel per step, all of it in front of the staging kernel.
(a) graph replays back to back, (b) one eager kernel in front of every replay, (c) that kernel captured as the graph's first node.
The graph is 30 element-wise kernels of ~10 us each, so the host is far ahead in every variant."""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(16 << 20, device=dev)          # 64 MB: an add_ takes ~25 us
y = torch.zeros(16 << 20, device=dev)


def body(with_stage):
    if with_stage:
        y.add_(1.0)
    for _ in range(30):
        x.add_(1.0)


def capture(with_stage):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(with_stage)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(with_stage)
    return g


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


g_plain, g_staged = capture(False), capture(True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(100):
    y.add_(1.0)
ev1.record()
torch.cuda.synchronize()
t_kernel = ev0.elapsed_time(ev1) * 10          # us per eager kernel, back to back
a = timed(lambda: g_plain.replay())
b = timed(lambda: (y.add_(1.0), g_plain.replay()))
c = timed(lambda: g_staged.replay())
print(f"one element-wise kernel, back to back: {t_kernel:.1f} us")
print(f"(a) graph of 30 kernels, replays back to back:            {a:.1f} us per iteration")
print(f"(b) one EAGER kernel in front of every replay:             {b:.1f} us  (+{b - a:.1f}; the kernel itself {t_kernel:.1f})")
print(f"(c) the same kernel captured as the graph's first node:    {c:.1f} us  (+{c - a:.1f})")
