import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, bench, fragnet_amd
from fragnet_amd import parallel
from fragnet_amd.model import FragNetFineTune
from fragnet_amd.plan import PLAN_KEY
fragnet_amd.prefer_rocblas_for_dense_heads()
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
def timeit(tag):
    for i in range(10): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(60): l = step(i)
    torch.cuda.synchronize(); print(tag, (time.perf_counter()-t0)/60*1e3, "ms/step loss", float(l))
timeit("eager head")
sample = torch.randn(512, 256, device=dev, requires_grad=True)
model.fthead = torch.cuda.make_graphed_callables(model.fthead, (sample,))
timeit("graphed head")
