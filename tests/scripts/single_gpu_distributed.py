"""Run by tests/test_gpu_distributed.py in a subprocess: the N>1 training-step sequence on ONE GPU.

A 1-rank "nccl" (RCCL) process group + ``force_distributed=True`` makes GraphedTrainStep take the branch that ranks of
a multi-GPU job take: hipGraph replay (stage + plan + forward + loss + backward + gradient gather) -> RCCL all-reduce
(AVG) of the flat gradient buffer -> fn_adam_f32 outside the graph, with the rank loss weights of the pretrain loss
coming through an all-reduce.  The result must equal the single-rank step (Adam captured inside the graph) bit for bit:
averaging over one rank is the identity.  Also covers the two-graph overlapped exchange and an eager fallback in it.
"""
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fragnet_amd import data, graphstep, parallel, synth          # noqa: E402
from fragnet_amd.model import FragNetFineTune, FragNetPreTrain      # noqa: E402

DEV = torch.device("cuda", 0)


def batches(n, B, seed, pretrain=False):
    coll = data.collate_fn_pt if pretrain else data.collate_fn
    return [data.batch_to(coll(synth.synth_molecules(B, seed=seed + i, profile="esol", pretrain_targets=pretrain)), DEV) for i in range(n)]


def run(kind, drop, overlap=False):
    pre = kind == "pretrain"
    bs = batches(3, 40, 300, pre)
    shapes = graphstep.StaticShapes.from_batches(bs, margin=0.05)
    torch.manual_seed(5)
    if pre:
        model_a = FragNetPreTrain(num_layer=2, drop_ratio=drop, num_heads=4, emb_dim=128, atom_features=167, frag_features=167, edge_features=17).to(DEV)
        from fragnet_amd.train import pretrain_loss
        loss_of = lambda m, b: pretrain_loss(m(b), b)
    else:
        model_a = FragNetFineTune(n_classes=1, num_layer=2, drop_ratio=drop, h1=32, h2=64, h3=64, h4=32, act="relu", fthead="FTHead3").to(DEV)
        loss_of = lambda m, b: torch.nn.functional.mse_loss(m(b).view(-1), b["y"])
    model_a.train()
    model_b = copy.deepcopy(model_a)
    opts = []
    for m in (model_a, model_b):
        opts.append(parallel.FlatAdam.for_live_parameters(m, lambda m=m: loss_of(m, dict(bs[0])).backward(), lr=1e-3))
    step_a = graphstep.GraphedTrainStep(model_a, opts[0], shapes, dict(bs[0]), loss=kind)
    step_b = graphstep.GraphedTrainStep(model_b, opts[1], shapes, dict(bs[0]), loss=kind, force_distributed=True, overlap=overlap)
    assert step_a.adam_in_graph and not step_b.adam_in_graph
    assert step_b.split == (overlap and not pre)
    for i in range(4):
        la = step_a(dict(bs[i % 3])).clone()
        lb = step_b(dict(bs[i % 3])).clone()
        assert torch.equal(la, lb), (kind, i, float(la), float(lb))
    assert step_b.replays == 4 and step_b.fallbacks == 0
    assert torch.equal(opts[0].flat.detach(), opts[1].flat.detach()), f"{kind}: weights differ after 4 steps"
    if overlap:        # a batch beyond the capacities: the eager fallback of the two-graph step issues the same two slice exchanges
        big = batches(1, 80, 900, pre)[0]
        before = opts[1].flat.detach().clone()
        step_b(dict(big))
        assert step_b.fallbacks == 1 and not torch.equal(before, opts[1].flat.detach())
        step_a(dict(big))
        torch.testing.assert_close(opts[1].flat.detach(), opts[0].flat.detach(), atol=1e-6, rtol=1e-5)
    print(f"ok {kind} drop={drop} overlap={overlap}", flush=True)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    try:
        run("regr", 0.1)
        run("pretrain", 0.0)
        run("regr", 0.0, overlap=True)
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    print("ALL OK", flush=True)


if __name__ == "__main__":
    main()
