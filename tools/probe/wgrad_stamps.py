"""dev: phase stamps of wgrad128_block inside k_wgrad_all (needs the instrumented build)"""
import sys, torch
sys.path.insert(0, ".")
from fragnet_amd import _lib, data, synth
from fragnet_amd.model import FragNetFineTune
DEV = "cuda:0"
b = data.batch_to(data.collate_fn(synth.synth_molecules(512, seed=1000, profile="esol")), DEV)
torch.manual_seed(0)
m = FragNetFineTune(n_classes=1, num_layer=4, drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu", fthead="FTHead3").to(DEV).train()
for it in range(3):
    m.zero_grad(set_to_none=True)
    b.pop("_fragnet_plan", None)
    out = m(b)
    loss = torch.nn.functional.mse_loss(out.view(-1), b["y"])
    if it == 2:
        buf = torch.zeros(16 * 1024, dtype=torch.int64, device=DEV)
        _lib.call("fn_debug_set_stamps", buf.data_ptr(), buf.numel())
    loss.backward()
torch.cuda.synchronize()
_lib.call("fn_debug_set_stamps", None, 0)
s = buf.view(1024, 16).cpu().double()
s = s[s[:, 12] > 0][:256]
ns = s[:, 12]
print("blocks", s.shape[0], "slots median", float(ns.median()), "min", float(ns.min()), "max", float(ns.max()))
wall = (s[:, 15] - s[:, 14]) * 0.01
rate = ((s[:, 11] - s[:, 0]) / wall).median()
print("wall median", float(wall.median()), "max", float(wall.max()), "ticks/us", float(rate))
full = s[ns == ns.median()]
t = lambda i: (full[:, i] - full[:, 0]) / rate
prev = 0
for i in range(1, int(ns.median()) + 1):
    print(f"slot {i-1} landed at {float(t(i).median()):6.2f} (+{float((t(i)-prev).median()):5.2f})"); prev = t(i)
print(f"loop done  {float(t(10).median()):6.2f} (+{float((t(10)-prev).median()):5.2f})")
print(f"end        {float(t(11).median()):6.2f} (+{float((t(11)-t(10)).median()):5.2f})")
