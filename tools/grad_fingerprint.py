#!/usr/bin/env python3
"""Bit-level fingerprint of one training step's gradients (eager and static-shape padded) on the small test model:
prints a SHA-1 per parameter tensor.  Run in several processes / on several boxes and diff the output: every line must
be identical (no float atomics, fixed reduction orders) -- this is how run-to-run nondeterminism is located."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from fragnet_amd import data, graphstep, parallel      # noqa: E402
import tests.test_graphstep as T                        # noqa: E402

dev = torch.device("cuda:0")
batches = [data.batch_to(b, dev) for b in T._batches(4, 48, seed=21)]
shapes = graphstep.StaticShapes.from_batches(batches, margin=0.05)
model, par, lr = T._make(dev)


def run_eager(b):
    torch.nn.functional.mse_loss(model(dict(b)).view(-1), b["y"]).backward()


opt = par.FlatAdam.for_live_parameters(model, lambda: run_eager(batches[0]), lr=lr)
names = {id(p): n for n, p in model.named_parameters()}
sb = graphstep.StaticBatch(shapes, batches[0])


def fp(tag):
    opt.gather_grads()
    g = opt.grad.detach().cpu().numpy()
    for p, off in zip(opt.params, opt.offsets):
        print(tag, names[id(p)], hashlib.sha1(g[off: off + p.numel()].tobytes()).hexdigest()[:12])


for k in range(2):
    opt.zero_grad()
    run_eager(batches[k])
    fp(f"eager{k}")
    assert sb.load(batches[k])
    opt.zero_grad()
    t = sb.t
    t.pop("_fragnet_plan", None)
    graphstep.masked_regr_loss(model(t), t["y"], t[graphstep.MASK_KEY]).backward()
    fp(f"padded{k}")

# ---- six optimiser steps: eager FlatAdam vs the captured step (same sequence as tests/test_graphstep.py)
import copy                                               # noqa: E402

model_a, par, lr = T._make(dev)
model_b = copy.deepcopy(model_a)


def probe(m):
    return lambda: torch.nn.functional.mse_loss(m(dict(batches[0])).view(-1), batches[0]["y"]).backward()


opt_a = par.FlatAdam.for_live_parameters(model_a, probe(model_a), lr=lr)
opt_b = par.FlatAdam.for_live_parameters(model_b, probe(model_b), lr=lr)
step_b = graphstep.GraphedTrainStep(model_b, opt_b, shapes, dict(batches[0]), loss="regr")
for i in range(6):
    b = batches[i % 4]
    opt_a.zero_grad()
    torch.nn.functional.mse_loss(model_a(dict(b)).view(-1), b["y"]).backward()
    opt_a.step()
    step_b(dict(b))
    torch.cuda.synchronize()
    print(f"step{i} eager-weights", hashlib.sha1(opt_a.flat.detach().cpu().numpy().tobytes()).hexdigest()[:12],
          "graph-weights", hashlib.sha1(opt_b.flat.detach().cpu().numpy().tobytes()).hexdigest()[:12],
          "graph-grads", hashlib.sha1(opt_b.grad.detach().cpu().numpy().tobytes()).hexdigest()[:12],
          "max|dw|", float((opt_a.flat - opt_b.flat).abs().max()))
