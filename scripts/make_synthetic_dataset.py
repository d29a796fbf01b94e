#!/usr/bin/env python3
"""Writes flat molecule stores (fragnet_amd.dataset.FlatMolStore) of synthetic ESOL-/Tox21-shape molecules.
There is no RDKit in the build image, so this stands in for the reference's data_create/*.py pipelines."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fragnet_amd import synth
from fragnet_amd.dataset import FlatMolStore

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True, help="output directory")
ap.add_argument("--profile", default="esol", choices=list(synth.PROFILES))
ap.add_argument("--n", type=int, nargs=3, default=[902, 113, 113], metavar=("TRAIN", "VAL", "TEST"))
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--pretrain-targets", action="store_true")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)
for split, n, s in zip(("train", "val", "test"), args.n, (0, 1, 2)):
    mols = synth.synth_molecules(n, seed=args.seed * 3 + s, profile=args.profile, pretrain_targets=args.pretrain_targets)
    FlatMolStore.from_records(mols).save(os.path.join(args.out, f"{split}.pt"))
    print(f"{split}: {n} molecules -> {os.path.join(args.out, split + '.pt')}")
