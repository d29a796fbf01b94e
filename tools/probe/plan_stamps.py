"""Phase stamps of k_plan_mol (csrc/mol_plan.hip): median over the molecule workgroups, microseconds."""
import sys
import torch
sys.path.insert(0, ".")
from fragnet_amd import _lib, data, plan as P, synth
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = data.batch_to(data.collate_fn(synth.synth_molecules(B, seed=1000, profile="esol")), DEV)
for _ in range(3):
    b.pop(P.PLAN_KEY, None); P.GraphPlan.from_batch(b)
buf = torch.zeros(B * 16, dtype=torch.int64, device=DEV)
_lib.call("fn_debug_set_stamps", buf.data_ptr(), buf.numel())
b.pop(P.PLAN_KEY, None); pl = P.GraphPlan.from_batch(b)
torch.cuda.synchronize()
_lib.call("fn_debug_set_stamps", None, 0)
s = buf.view(B, 16).cpu().double()
wall = (s[:, 15] - s[:, 14]) * 0.01            # 100 MHz
ticks = s[:, 1:14] - s[:, 0:1]
n = int((s[0, 1:14] > 0).sum())
rate = (ticks[:, n - 1] / wall).median()       # ticks per us
print("ticks/us", float(rate), "wall median", float(wall.median()), "max", float(wall.max()), "start spread us", float((s[:, 14].max() - s[:, 14].min()) * 0.01))
names = ["ext", "setup", "keys", "hist", "scan", "fill", "rank", "cross"]
prev = torch.zeros(B, dtype=torch.double)
for i in range(n):
    t = ticks[:, i] / rate
    print(f"{names[i] if i < len(names) else i:10s} +{float((t - prev).median()):6.2f} us   (at {float(t.median()):6.2f})")
    prev = t
