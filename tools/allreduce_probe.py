#!/usr/bin/env python3
"""Cost of torch.distributed's stream handling around a collective, measurable with a 1-rank RCCL group on one GPU:
the captured step alone, followed by a blocking all_reduce of the flat gradient buffer, and with async_op + wait.
dev tool: python tools/allreduce_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import fragnet_amd
from fragnet_amd import data, graphstep, parallel, synth
from fragnet_amd.model import FragNetFineTune

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
fragnet_amd.prefer_rocblas_for_dense_heads()
B = 512
batches = [data.batch_to(data.collate_fn(synth.synth_molecules(B, seed=10 + i)), dev) for i in range(2)]
shapes = graphstep.StaticShapes.from_batches(batches, margin=0.02)
torch.manual_seed(0)
model = FragNetFineTune(n_classes=1, num_layer=4, drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu").to(dev).train()
opt = parallel.FlatAdam.for_live_parameters(
    model, lambda: torch.nn.functional.mse_loss(model(dict(batches[0])).view(-1), batches[0]["y"]).backward(), lr=1e-4)
step = graphstep.GraphedTrainStep(model, opt, shapes, dict(batches[0]), loss="regr", overlap=False)

def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n

def plain():
    step(dict(batches[0]))

def blocking():
    step.replay(dict(batches[0]))          # staging + replay as a pair (counters, plan workspaces)
    dist.all_reduce(opt.grad, op=dist.ReduceOp.AVG)
    opt.apply_gathered(reduced=True)

def asynchronous():
    step.replay(dict(batches[0]))          # staging + replay as a pair (counters, plan workspaces)
    w = dist.all_reduce(opt.grad, op=dist.ReduceOp.AVG, async_op=True)
    w.wait()
    opt.apply_gathered(reduced=True)

for name, fn in (("step", plain), ("+ all_reduce", blocking), ("+ async/wait", asynchronous)):
    print(f"{name:14s} {timeit(fn):.3f} ms")
dist.destroy_process_group()
