import torch, sys
sys.path.insert(0, ".")
from fragnet_amd import data, synth, _lib
from fragnet_amd.model import FragNetPreTrain, FragNetFineTune
DEV = "cuda:0"
def poison():
    xs = [torch.full((n,), float("nan"), device=DEV) for n in (64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 22) for _ in range(8)]
    del xs
layers, B = int(sys.argv[1]), int(sys.argv[2])
mols = synth.synth_molecules(B, seed=9, profile="esol")
coll = data.batch_to(data.collate_fn(mols), DEV)
torch.manual_seed(5)
model = FragNetFineTune(n_classes=1, num_layer=layers, drop_ratio=0.1, h1=64, h2=64, h3=64, h4=32, act="relu", fthead="FTHead3").to(DEV).train()
for key20 in (0, 1, 1):
    _lib.call("fn_set_tuning", 20, key20)
    model.zero_grad(set_to_none=True)
    model.pretrain.rng.offset = 5
    coll.pop("_fragnet_plan", None)
    poison()
    xa, xf, xb, xfb = model.pretrain(coll)
    from fragnet_amd.model import pooled
    pl = pooled(xa, xf, coll)
    torch.cuda.synchronize()
    print("tail", key20, "fwd finite:", [bool(torch.isfinite(t).all()) for t in (xa, xf, xb, xfb, pl)], "sums", [float(t.double().sum()) for t in (xa, xf, pl)])
    loss = pl.square().mean()
    poison()
    loss.backward()
    torch.cuda.synchronize()
    coll["_fragnet_plan"].check()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("   bwd non-finite:", bad, " |grad| sum", sum(float(p.grad.double().abs().nan_to_num(0).sum()) for p in model.parameters() if p.grad is not None))
