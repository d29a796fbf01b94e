"""Dev probe: host time of FlatMolStore.collate per batch (no synchronisation) against its device time, by batch size."""
import sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, ".")
from fragnet_amd import synth
from fragnet_amd.dataset import FlatMolStore
from fragnet_amd.train import StoreLoader

dev = torch.device("cuda:0")
base = FlatMolStore.from_records(synth.synth_molecules(4096, seed=7000, profile="synth40")).to(dev)
store = base.replicate(32)
for B in (512, 2048, 8192):
    loader = StoreLoader(store, B, shuffle=True, drop_last=True, seed=11)
    it = iter(loader.sampler)
    idx = [next(it) for _ in range(12)]
    for i in idx[:4]:
        store.collate(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx[4:]:
        store.collate(i)
    host = (time.perf_counter() - t0) / 8
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 8
    print(f"B={B}: host {host*1e3:.3f} ms per collate, host+device {total*1e3:.3f} ms")
    if B == 8192:
        pr = cProfile.Profile(); pr.enable()
        for i in idx[4:]:
            store.collate(i)
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3000])
