#!/usr/bin/env python3
"""Checks the per-molecule extents table (fn_mol_extents: MolExt) of a collated batch against the plan it was derived from:
for every molecule and level, the row range must map to the edge range in BOTH CSR orders, and every edge of the range must
stay inside the molecule's rows (the closure the molecule-resident backward relies on).   python tools/probe/dbg_ext.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import molbwd_check as mc                              # noqa: E402
from fragnet_amd import data, synth                     # noqa: E402
from fragnet_amd.plan import GraphPlan                  # noqa: E402

dev = torch.device("cuda", 0)
batch = data.batch_to(data.collate_fn(synth.synth_molecules(64, seed=1000, profile="esol")), dev)
plan = GraphPlan.from_batch(batch)
ext = mc.mol_extents(plan, dev)
torch.cuda.synchronize()
e = ext.cpu()
print(e[:4])


def view(t, ptr, n):
    off = (ptr - t.data_ptr()) // 4
    return t[off: off + n].cpu()


for name, (r0i, e0i) in {"bond": (2, 8), "atom": (0, 10), "fbond": (6, 12), "frag": (4, 14)}.items():
    lv = plan.levels[name]
    c = lv.c
    rpd = view(plan.rowptr, c.rowptr_d, lv.n + 1) - c.pos_base_d
    rps = view(plan.rowptr, c.rowptr_s, lv.n + 1) - c.pos_base_s
    dst_s = view(plan.aux_a, c.dst_s, lv.m)
    dpos_s = view(plan.aux_b, c.dpos_s, lv.m)
    bad = 0
    for k in range(e.shape[0]):
        r0, nr, e0, me = int(e[k, r0i]), int(e[k, r0i + 1]), int(e[k, e0i]), int(e[k, e0i + 1])
        if rpd[r0] != e0 or rpd[r0 + nr] != e0 + me or rps[r0] != e0 or rps[r0 + nr] != e0 + me:
            bad += 1
            continue
        ds, dp = dst_s[e0:e0 + me], dpos_s[e0:e0 + me]
        if me and (ds.min() < r0 or ds.max() >= r0 + nr or dp.min() < e0 or dp.max() >= e0 + me):
            bad += 1
    print(name, "n", lv.n, "m", lv.m, "m_real", lv.m_real, "bad molecules", bad)
