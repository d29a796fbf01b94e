#!/usr/bin/env python3
"""Can the next batch's stage + plan build hide behind the current step?  Replays the captured training step on the main
stream while a second stream builds an (unused) graph plan for another batch, and compares with the serial order.
dev tool: python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fragnet_amd
from fragnet_amd import data, graphstep, parallel, synth
from fragnet_amd.model import FragNetFineTune
from fragnet_amd.plan import GraphPlan

dev = torch.device("cuda:0")
fragnet_amd.prefer_rocblas_for_dense_heads()
B = 512
batches = [data.batch_to(data.collate_fn(synth.synth_molecules(B, seed=10 + i)), dev) for i in range(4)]
shapes = graphstep.StaticShapes.from_batches(batches, margin=0.02)
torch.manual_seed(0)
model = FragNetFineTune(n_classes=1, num_layer=4, drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu").to(dev).train()
opt = parallel.FlatAdam.for_live_parameters(
    model, lambda: torch.nn.functional.mse_loss(model(dict(batches[0])).view(-1), batches[0]["y"]).backward(), lr=1e-4)
step = graphstep.GraphedTrainStep(model, opt, shapes, dict(batches[0]), loss="regr")
side = torch.cuda.Stream(dev)
other = graphstep.StaticBatch(shapes, dict(batches[1]))
other.load(dict(batches[1]))

def side_work():                       # what a prefetch would do: stage + plan of the next batch
    other.load(dict(batches[2]))
    GraphPlan.from_batch(other.t, n_mols=shapes.cap["mol"])

# capture the side work as its own graph
with torch.cuda.stream(side):
    side_work()
torch.cuda.synchronize()
g_side = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_side, stream=side):
    side_work()
torch.cuda.synchronize()

def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n

def main_only():
    step(dict(batches[0]))

def serial():
    step(dict(batches[0]))
    g_side.replay()

def overlapped():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g_side.replay()
    step(dict(batches[0]))
    torch.cuda.current_stream().wait_stream(side)

def side_only():
    g_side.replay()

def side_only_eager():
    side_work()

def overlapped_eager():                # the side work as ordinary launches (what an RCCL collective is), main = graph replay
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        side_work()
    step(dict(batches[0]))
    torch.cuda.current_stream().wait_stream(side)

big = torch.empty(64 << 20, dtype=torch.float32, device=dev)       # 256 MB: a bandwidth-bound stand-in for a collective

def copy_only():
    big[: 32 << 20].copy_(big[32 << 20:])

def overlapped_copy():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        big[: 32 << 20].copy_(big[32 << 20:])
    step(dict(batches[0]))
    torch.cuda.current_stream().wait_stream(side)

tiny = torch.zeros(4, device=dev)

def sync_only():                       # cross-stream dependencies around a trivial kernel on the side stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tiny.add_(1.0)
    step(dict(batches[0]))
    torch.cuda.current_stream().wait_stream(side)

def sync_after():                      # the pattern of a collective after the step: side waits for main, main waits for side
    step(dict(batches[0]))
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tiny.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)

for name, fn in (("sync only", sync_only), ("sync after", sync_after), ("main only", main_only), ("side only", side_only), ("serial", serial), ("overlapped", overlapped),
                 ("side eager", side_only_eager), ("ovl eager", overlapped_eager), ("copy only", copy_only), ("ovl copy", overlapped_copy)):
    print(f"{name:12s} {timeit(fn):.3f} ms")
