#!/usr/bin/env python3
"""Counterpart of the reference's fragnet/train/finetune/finetune_gat2.py (model_version gat2): same CLI
(--config X.yaml), same YAML schema (exps/ft/esol/e1pt4.yaml), same checkpoint format (plain state_dict), on the
MI355X path.  Datasets are flat stores (fragnet_amd/dataset.py) instead of pickled torch_geometric Data lists.

    python scripts/finetune_gat2.py --config exps/ft/esol_synth/config.yaml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/finetune_gat2.py --config ...
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fragnet_amd import parallel, train
from fragnet_amd.dataset import FlatMolStore
from fragnet_amd.model import FragNetFineTune, FragNetPreTrain

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
    ft, m = args.finetune, args.finetune.model
    if args.model_version not in ("gat2", "gat2_lite", "gat2_edge"):
        raise SystemExit("model_version gat2, gat2_lite and gat2_edge are on the accelerated path")
    model = FragNetFineTune(n_classes=m.n_classes, atom_features=args.atom_features, frag_features=args.frag_features,
                            edge_features=args.edge_features, num_layer=m.num_layer, drop_ratio=m.drop_ratio,
                            num_heads=m.num_heads, emb_dim=m.emb_dim, h1=m.h1, h2=m.h2, h3=m.h3, h4=m.h4, act=m.act,
                            fthead=m.fthead, variant=args.model_version)
    pt = args.pretrain
    if pt.get("chkpoint_name") and os.path.exists(str(pt.chkpoint_name)):
        modelpt = FragNetPreTrain(num_layer=pt.num_layer, drop_ratio=pt.drop_ratio, num_heads=pt.num_heads, emb_dim=pt.emb_dim,
                                  atom_features=args.atom_features, frag_features=args.frag_features,
                                  edge_features=args.edge_features, fedge_in=args.fedge_in, fbond_edge_in=args.fbond_edge_in)
        modelpt.load_state_dict(torch.load(pt.chkpoint_name, map_location="cpu"))
        model.pretrain.load_state_dict(modelpt.pretrain.state_dict())
        print("loaded pretrained encoder", pt.chkpoint_name)
    model.to(device)
    model.pretrain.rng.rank = rank
    stores = {k: FlatMolStore.load(ft[k].path, device=device) for k in ("train", "val", "test")}
    train_loader = train.StoreLoader(stores["train"], ft.batch_size, shuffle=True, drop_last=True, seed=args.seed, rank=rank, world=world)
    val_loader = train.StoreLoader(stores["val"], 64)
    test_loader = train.StoreLoader(stores["test"], 64)
    trainer = train.TrainerFineTune(target_type=ft.target_type)
    probe = next(iter(train_loader))
    optimizer = train.make_optimizer(model, float(ft.lr), probe, lambda mdl, b: trainer._loss(mdl, b))
    # whole-step hipGraph over static shapes (fragnet_amd/graphstep.py); `finetune.graph_step: false` in the YAML keeps
    # the launch-by-launch step (then only the prediction head is graph-captured)
    graph_step = None
    if ft.get("graph_step", True) and ft.target_type in ("regr", "clsf"):
        from fragnet_amd import graphstep
        sample = [probe] + [b for _, b in zip(range(7), iter(train_loader))]
        shapes = graphstep.StaticShapes.from_batches(sample, margin=0.05, heads=m.num_heads)
        model.train()
        graph_step = graphstep.GraphedTrainStep(model, optimizer, shapes, probe, loss=ft.target_type)
    else:
        fragnet_amd.graph_capture_head(model, len(probe["y"]))      # training batches have a fixed size (drop_last)
    scheduler = None
    if ft.get("use_schedular"):
        class _Linear:          # LinearLR(start_factor=1.0, end_factor=0.5, total_iters=30), finetune_gat2.py:258-259
            def __init__(self, opt): self.opt, self.base, self.t = opt, float(ft.lr), 0
            def step(self):
                self.t += 1
                self.opt.hyper["lr"] = self.base * (1.0 - 0.5 * min(self.t, 30) / 30)
        scheduler = _Linear(optimizer)
    stopper = train.EarlyStopping(patience=ft.es_patience, verbose=rank == 0, chkpoint_name=ft.chkpoint_name)
    log = open(os.path.join(exp_dir, "log.jsonl"), "a") if rank == 0 else None
    for epoch in range(ft.n_epochs):
        train_loss = trainer.train(model=model, loader=train_loader, optimizer=optimizer, scheduler=scheduler, graph_step=graph_step)
        val_loss, _, _ = trainer.test(model=model, loader=val_loader)
        if rank == 0:
            print("epoch: ", epoch, train_loss, val_loss, "" if graph_step is None else f"(graph replays {graph_step.replays}, eager fallbacks {graph_step.fallbacks})")
            log.write(json.dumps({"epoch": epoch, "Loss/train": train_loss, "Loss/val": val_loss}) + "\n")
            log.flush()
        stopper(val_loss, model) if rank == 0 else None       # clsf: test() returns -AUC, a loss like the reference's test_clsf_bce
        stop = torch.tensor([int(stopper.early_stop)], device=device)
        if world > 1:
            torch.distributed.broadcast(stop, 0)
        if int(stop):
            print("Early stopping")
            break
    if rank == 0:
        model.load_state_dict(torch.load(ft.chkpoint_name, map_location=device))
        for name, loader in (("val_res", val_loader), ("test_res", test_loader)):
            acc = train.save_predictions(trainer, loader, model, exp_dir, name, ft.loss, args.seed)
            print(f"{name} {'rmse' if ft.loss == 'mse' else '-auc'}: {acc}")
