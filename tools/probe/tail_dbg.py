import torch, sys
sys.path.insert(0, ".")
from fragnet_amd import data, synth, _lib
from fragnet_amd.model import FragNetPreTrain, FragNetFineTune
DEV = "cuda:0"
def poison():
    # fill the caching allocator's free blocks with NaN so that unwritten gradient elements show
    xs = [torch.full((n,), float("nan"), device=DEV) for n in (64, 256, 1024, 4096, 16384, 65536, 1 << 20, 1 << 22) for _ in range(6)]
    del xs
def run(kind, layers, B, key20):
    _lib.call("fn_set_tuning", 20, key20)
    pt = kind == "pt"
    mols = synth.synth_molecules(B, seed=9, profile="esol", pretrain_targets=pt)
    coll = data.batch_to((data.collate_fn_pt if pt else data.collate_fn)(mols), DEV)
    torch.manual_seed(5)
    model = (FragNetPreTrain(num_layer=layers, drop_ratio=0.1, edge_features=17) if pt else FragNetFineTune(n_classes=1, num_layer=layers, drop_ratio=0.1, h1=64, h2=64, h3=64, h4=32, act="relu", fthead="FTHead3")).to(DEV).train()
    res = []
    for it in range(4):
        model.zero_grad(set_to_none=True)
        model.pretrain.rng.offset = 5
        coll.pop("_fragnet_plan", None)
        outs = model(coll)
        outs = outs if isinstance(outs, tuple) else (outs,)
        loss = sum(o.square().mean() for o in outs if o is not None)
        poison()
        loss.backward()
        torch.cuda.synchronize()
        res.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    bad = []
    for n in res[0]:
        nan = any(not torch.isfinite(r[n]).all() for r in res)
        diff = max(float((res[0][n] - r[n]).abs().nan_to_num(1e9).max()) for r in res[1:])
        if nan or diff > 0:
            bad.append((n, nan, diff, tuple(res[0][n].shape)))
    print(kind, "layers", layers, "B", B, "tail", key20, "->", bad if bad else "deterministic, all written")
for kind in ("ft", "pt"):
    for layers in (1, 2, 4):
        for key20 in (0, 1):
            run(kind, layers, 48, key20)
