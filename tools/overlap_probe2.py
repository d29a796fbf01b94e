#!/usr/bin/env python3
"""Stand-in for the N > 1 gradient exchange on one GPU: a bandwidth-bound side-stream kernel of ~100 us plays the head's
all-reduce, a ~30 us one the encoder's.  Compares (P1) whole step then one exchange with (P2) the two-graph step of
graphstep.GraphedTrainStep(overlap=True) with the first exchange beside the second graph.  dev tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fragnet_amd
from fragnet_amd import data, graphstep, parallel, synth
from fragnet_amd.model import FragNetFineTune

dev = torch.device("cuda:0")
fragnet_amd.prefer_rocblas_for_dense_heads()
B = 512
batches = [data.batch_to(data.collate_fn(synth.synth_molecules(B, seed=10 + i)), dev) for i in range(4)]
shapes = graphstep.StaticShapes.from_batches(batches, margin=0.02)
torch.manual_seed(0)
model = FragNetFineTune(n_classes=1, num_layer=4, drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu").to(dev).train()
opt = parallel.FlatAdam.for_live_parameters(
    model, lambda: torch.nn.functional.mse_loss(model(dict(batches[0])).view(-1), batches[0]["y"]).backward(), lr=1e-4)
step = graphstep.GraphedTrainStep(model, opt, shapes, dict(batches[0]), loss="regr", overlap=True)
assert step.split
side = torch.cuda.Stream(dev)
big = torch.empty(96 << 20, dtype=torch.float32, device=dev)

def fake_exchange(n_copies):
    for _ in range(n_copies):
        big[: 48 << 20].copy_(big[48 << 20:])           # ~45 us each

def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n

cur = torch.cuda.current_stream()

def compute_only():
    step.replay(dict(batches[0]))          # staging + both graphs as a pair
    opt.apply_gathered(reduced=True)

def p1_after():
    step.replay(dict(batches[0]))          # staging + both graphs as a pair
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fake_exchange(3)
    cur.wait_stream(side)
    opt.apply_gathered(reduced=True)

def p2_beside():
    step.static.load(dict(batches[0]))
    step.graph.replay()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fake_exchange(2)
    step.graph_b.replay()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fake_exchange(1)
    cur.wait_stream(side)
    opt.apply_gathered(reduced=True)

def exchange_only():
    fake_exchange(3)

for name, fn in (("compute only", compute_only), ("exchange only", exchange_only), ("P1 after", p1_after), ("P2 beside", p2_beside)):
    print(f"{name:14s} {timeit(fn):.3f} ms")
