"""Dev tool: fused molecule-resident encoder (FN_TUNE_FUSED = 1) vs the per-level engine (0) on the same batch, same
weights, same Philox stream: outputs, loss and parameter gradients, plus forward / forward+backward timing.

    python tools/fused_check.py [--batch 512] [--profile esol] [--drop 0.1] [--layers 4]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fragnet_amd import _lib, data, synth          # noqa: E402
from fragnet_amd.model import FragNetFineTune        # noqa: E402

FN_TUNE_FUSED = 4


def run(model, batch, fused, seed_off):
    _lib.call("fn_set_tuning", FN_TUNE_FUSED, int(fused))
    model.zero_grad(set_to_none=True)
    model.pretrain.rng.offset = seed_off
    batch.pop("_fragnet_plan", None)
    x_atoms, x_frags, x_bond, x_fbond = model.pretrain(batch)
    loss = (x_atoms.square().mean() + x_frags.square().mean() + x_bond.square().mean() * 0.5
            + (x_fbond.square().mean() * 0.25 if x_fbond is not None else 0.0))
    loss.backward()
    torch.cuda.synchronize()
    batch["_fragnet_plan"].check()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    outs = [t.detach().clone() for t in (x_atoms, x_frags, x_bond, x_fbond) if t is not None]
    return outs, loss.item(), grads


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--profile", default="esol")
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--variant", default="gat2")
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--tune", action="append", default=[], help="KEY=VALUE for fn_set_tuning, timed after the comparison (fused on)")
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    cls = FragNetFineTune
    kw = dict(n_classes=1, num_layer=args.layers, drop_ratio=args.drop, h1=128, h2=1024, h3=1024, h4=512, act="relu", fthead="FTHead3")
    if args.variant != "gat2":
        from fragnet_amd import model as M
        cls = {"gat2_lite": M.FragNetFineTuneLite, "gat2_edge": M.FragNetFineTuneEdge}[args.variant]
    model = cls(**kw).to(dev)
    model.train()
    mols = synth.synth_molecules(args.batch, seed=args.seed, profile=args.profile)
    batch = data.batch_to(data.collate_fn(mols), dev)
    if args.variant == "gat2_edge" and batch["cnx_attr"].shape[1] < 8:
        batch["cnx_attr"] = torch.nn.functional.pad(batch["cnx_attr"], (0, 8 - batch["cnx_attr"].shape[1]))
    print({k: tuple(v.shape) for k, v in batch.items() if torch.is_tensor(v) and k in ("x_atoms", "node_features_bonds", "edge_attr_bonds", "x_frags", "node_features_fbonds", "edge_attr_fbonds")})
    o0, l0, g0 = run(model, batch, 0, 12345)
    o1, l1, g1 = run(model, batch, 1, 12345)
    names = ["atoms", "frags", "bond", "fbond"]
    ok = True
    for n, a, b in zip(names, o0, o1):
        d = (a - b).abs().max().item()
        print(f"out {n:6s} max|diff| {d:.3e}  (max|ref| {a.abs().max().item():.3e}) nan={bool(torch.isnan(b).any())}")
        ok &= d < 1e-4
    print(f"loss {l0:.8f} vs {l1:.8f}")
    worst = 0.0
    for n in g0:
        d = (g0[n] - g1[n]).abs().max().item()
        s = g0[n].abs().max().item()
        worst = max(worst, d / (s + 1e-12))
        if d > 1e-5 * max(s, 1.0):
            print(f"grad {n}: max|diff| {d:.3e} (max|ref| {s:.3e})")
    print(f"worst relative grad diff {worst:.3e}; missing grads: {sorted(set(g0) ^ set(g1))}")
    print("OK" if ok and worst < 1e-3 else "MISMATCH")

    def fwd():
        batch.pop("_fragnet_plan", None)
        with torch.no_grad():
            model.pretrain(batch)

    def fwdbwd():
        batch.pop("_fragnet_plan", None)
        model.zero_grad(set_to_none=True)
        a, f, b, fb = model.pretrain(batch)
        (a.sum() + f.sum()).backward()

    def graphed(fn):
        """GPU time of fn() replayed as a hipGraph (no host launch overhead), ms."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    for fused in (0, 1):
        _lib.call("fn_set_tuning", FN_TUNE_FUSED, fused)
        print(f"fused={fused}: forward {timeit(fwd):.3f} ms, forward+backward {timeit(fwdbwd):.3f} ms (eager, host-inclusive); "
              f"graphed forward {graphed(fwd):.3f} ms")
    for spec in args.tune:
        k, v = spec.split("=")
        _lib.call("fn_set_tuning", int(k), int(v))
        print(f"tune {k}={v}: graphed forward {graphed(fwd):.3f} ms")


if __name__ == "__main__":
    main()
