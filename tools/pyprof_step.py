#!/usr/bin/env python3
"""Host-side profile of the bench step (where does the Python time go)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fragnet_amd import parallel
from fragnet_amd.model import FragNetFineTune
from fragnet_amd.plan import PLAN_KEY

import fragnet_amd; fragnet_amd.prefer_rocblas_for_dense_heads()
dev = torch.device("cuda", 0)
pool = bench.make_pool(4, 0, dev)
torch.manual_seed(0)
model = FragNetFineTune(**bench.MODEL_CFG).to(dev)
model.train()

def fwd_bwd(batch):
    batch.pop(PLAN_KEY, None)
    loss = torch.nn.functional.mse_loss(model(batch).view(-1), batch["y"])
    loss.backward()
    return loss

opt = parallel.FlatAdam.for_live_parameters(model, lambda: fwd_bwd(pool[0]), lr=1e-4)
def step(i):
    opt.zero_grad(); loss = fwd_bwd(pool[i % 4]); opt.step(); return loss
for i in range(10): step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(50): step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_enq/50*1e3:.3f} ms/step, with sync {t_all/50*1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(50): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
