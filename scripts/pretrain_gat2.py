#!/usr/bin/env python3
"""Counterpart of the reference's fragnet/train/pretrain/pretrain_gat2.py: same CLI and YAML schema
(exps/pt/unimol_exp1s4/config.yaml), data parallel over the GPUs of one node with one flat all-reduce per step."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fragnet_amd import parallel, train
from fragnet_amd.dataset import FlatMolStore
from fragnet_amd.model import FragNetPreTrain

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config.yaml")
    cli = ap.parse_args()
    args = train.load_config(cli.config, config=cli.config)
    train.seed_everything(args.seed)
    import fragnet_amd
    fragnet_amd.prefer_rocblas_for_dense_heads()
    fragnet_amd.tune_library_gemms()
    rank, local_rank, world = parallel.init_distributed()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    exp_dir = args["exp_dir"]
    os.makedirs(exp_dir, exist_ok=True)
    pt = args.pretrain
    model = FragNetPreTrain(num_layer=pt.num_layer, drop_ratio=pt.drop_ratio, num_heads=pt.num_heads, emb_dim=pt.emb_dim,
                            atom_features=args.atom_features, frag_features=args.frag_features,
                            edge_features=args.edge_features, fedge_in=args.fedge_in, fbond_edge_in=args.fbond_edge_in)
    if pt.get("saved_checkpoint"):
        model.load_state_dict(torch.load(pt.saved_checkpoint, map_location="cpu"))
    model.to(device)
    model.pretrain.rng.rank = rank
    # pretrain.data: list of directories holding train.pt / val.pt flat stores (the reference splits 90/10 itself,
    # pretrain_gat2.py:141; the synthetic stores come pre-split)
    d0 = pt.data[0]
    train_store = FlatMolStore.load(os.path.join(d0, "train.pt"), device=device)
    val_store = FlatMolStore.load(os.path.join(d0, "val.pt"), device=device)
    train_loader = train.StoreLoader(train_store, pt.batch_size, shuffle=True, drop_last=True, pretrain=True, seed=args.seed,
                                     rank=rank, world=world)
    val_loader = train.StoreLoader(val_store, pt.batch_size, pretrain=True)
    trainer = train.PretrainTrainer()
    probe = next(iter(train_loader))
    optimizer = train.make_optimizer(model, float(pt.lr), probe, lambda mdl, b: train.pretrain_loss(mdl(b), b))
    graph_step = None
    if pt.get("graph_step", True):      # whole-step hipGraph over static shapes; `pretrain.graph_step: false` = eager step
        from fragnet_amd import graphstep
        sample = [probe] + [b for _, b in zip(range(7), iter(train_loader))]
        shapes = graphstep.StaticShapes.from_batches(sample, margin=0.05, heads=pt.num_heads)
        model.train()
        graph_step = graphstep.GraphedTrainStep(model, optimizer, shapes, probe, loss="pretrain")
    stopper = train.EarlyStopping(patience=pt.es_patience, verbose=rank == 0, chkpoint_name=pt.chkpoint_name)
    every = int(pt.get("valdiate_every", 5))          # sic: the reference's key
    log = open(os.path.join(exp_dir, "log.jsonl"), "a") if rank == 0 else None
    for epoch in range(pt.n_epochs):
        train_loss = trainer.train(model, train_loader, optimizer, graph_step=graph_step)
        rec = {"epoch": epoch, "Loss/train": train_loss}
        if epoch % every == 0:
            val_loss = trainer.validate(val_loader, model)
            rec["Loss/val"] = val_loss
            if rank == 0:
                stopper(val_loss, model)
        if rank == 0:
            print(rec)
            log.write(json.dumps(rec) + "\n")
            log.flush()
        stop = torch.tensor([int(stopper.early_stop)], device=device)
        if world > 1:
            torch.distributed.broadcast(stop, 0)
        if int(stop):
            break
