"""Dev tool: per-phase cycle counts of the fused molecule kernel (fn_debug_set_stamps), averaged over workgroups.
    python tools/mol_phase_times.py [--batch 512]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fragnet_amd import _lib, data, synth          # noqa: E402
from fragnet_amd.model import FragNetFineTune        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--profile", default="esol")
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    model = FragNetFineTune(n_classes=1, num_layer=4, drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu", fthead="FTHead3").to(dev)
    model.train()
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(args.batch, seed=1000, profile=args.profile)), dev)
    S = 128
    for _ in range(3):
        batch.pop("_fragnet_plan", None)
        with torch.no_grad():
            model.pretrain(batch)
    stamps = torch.zeros(args.batch * S, dtype=torch.int64, device=dev)
    _lib.call("fn_debug_set_stamps", stamps.data_ptr(), stamps.numel())
    batch.pop("_fragnet_plan", None)
    with torch.no_grad():
        model.pretrain(batch)
    torch.cuda.synchronize()
    _lib.call("fn_debug_set_stamps", None, 0)
    st = stamps.view(args.batch, S).cpu().double()
    n_st = int((st[0] != 0).sum())
    st = st[:, :n_st]
    t0 = st[:, 0].min()
    print(f"{n_st} stamps per workgroup; kernel span (first start -> last end) {float(st[:, -1].max() - t0):.0f} ticks")
    print(f"workgroup start spread {float(st[:, 0].max() - t0):.0f}; workgroup lifetime mean {float((st[:, -1] - st[:, 0]).mean()):.0f} max {float((st[:, -1] - st[:, 0]).max()):.0f}")
    d = (st[:, 1:] - st[:, :-1])
    names = ["ext+csr staging"]
    per_level = ["fold (level start)", "stage X", "mfma", "store tiles", "scalars+copy-out", "attend", "drain+join"]
    # stamps: 0 start, 1 after csr; then per level: hit0 .. hit6
    labels = ["csr staging"]
    lv_names = ["bond", "atom", "fbond"]
    k = 0
    while len(labels) < n_st - 1:
        lv = lv_names[(k // 7) % 3]
        layer = k // 21
        ph = k % 7
        labels.append(f"L{layer} {lv:5s} {['(prev join->start)', 'fold+stage X', 'mfma', 'store tiles', 'scalars+copy-out', 'attend', 'drain+join'][ph]}")
        k += 1
    tot = d.mean(0).sum()
    agg = {}
    for i, lab in enumerate(labels[: d.shape[1]]):
        m = float(d[:, i].mean())
        key = lab.split(" ", 1)[1] if lab.startswith("L") else lab
        agg[key] = agg.get(key, 0.0) + m
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{key:32s} {v:10.0f} ticks  {100 * v / float(tot):5.1f} %")
    print("per-layer totals:", [round(float(d[:, 1 + 21 * l: 1 + 21 * (l + 1)].mean(0).sum())) for l in range(4)])


if __name__ == "__main__":
    main()
