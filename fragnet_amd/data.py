"""Batch assembly: per-molecule records -> the 16/19-key batch dict the model consumes.

Counterpart of the reference's ``collate_fn`` / ``collate_fn_pt``
(fragnet/dataset/data.py:877-948, 951-1032) and its five ``get_incr_*`` helpers
(data.py:11-113).  Same keys, dtypes and integer values (bit-exact; pinned by
tests/golden/collate_*.npz), but the per-molecule offsets are one ``cumsum`` +
``repeat_interleave`` per index space instead of Python loops over ``torch.cat``.

Index spaces and what offsets them:
  atoms        edge_index              += sum of previous x_atoms.size(0)
  fragments    frag_index, a2f         += sum of previous n_frags
  bond nodes   edge_index_bonds_graph  += sum of previous node_features_bonds.size(0)
  fbond nodes  edge_index_fbonds       += sum of previous node_feautures_fbondg.size(0)

The reference accumulates the first three offsets in float32 and casts to int64 at the
end (data.py:883,933-942); that is exact only below 2**24 nodes, so this collate refuses
larger batches instead of reproducing index collisions.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .plan import CollatedBatch, mol_offsets

_F32_EXACT = 1 << 24


def _offsets(counts: Sequence[int], reps: Sequence[int]) -> torch.Tensor:
    """For molecule i: (sum of counts[:i]) repeated reps[i] times, int64."""
    c = torch.as_tensor(list(counts), dtype=torch.long)
    r = torch.as_tensor(list(reps), dtype=torch.long)
    start = torch.cumsum(c, 0) - c
    return torch.repeat_interleave(start, r)


def _collate_common(data_list: List) -> Dict[str, torch.Tensor]:
    if len(data_list) == 0:
        raise ValueError("collate_fn: empty data_list")
    n_atoms = [int(d.x_atoms.size(0)) for d in data_list]
    n_frags = [int(d.n_frags.item()) for d in data_list]
    n_bnodes = [int(d.node_features_bonds.size(0)) for d in data_list]
    n_fbnodes = [int(d.node_feautures_fbondg.size(0)) for d in data_list]
    if max(sum(n_atoms), sum(n_frags), sum(n_bnodes)) >= _F32_EXACT:
        raise ValueError("batch exceeds 2**24 nodes: the reference's float32 offsets are inexact there")

    e = [int(d.edge_index.shape[1]) for d in data_list]
    ef = [int(d.frag_index.shape[1]) for d in data_list]
    eb = [int(d.edge_index_bonds.shape[1]) for d in data_list]
    efb = [int(d.edge_index_fbondg.shape[1]) for d in data_list]

    cat0 = lambda name: torch.cat([getattr(d, name) for d in data_list], dim=0)
    cat1 = lambda name: torch.cat([getattr(d, name) for d in data_list], dim=1)

    mol_id = torch.arange(len(data_list), dtype=torch.long)
    out = CollatedBatch({
        "x_atoms": cat0("x_atoms"),
        "edge_index": cat1("edge_index").to(torch.long) + _offsets(n_atoms, e),
        "frag_index": cat1("frag_index").to(torch.long) + _offsets(n_frags, ef),
        "x_frags": cat0("x_frags"),
        "edge_attr": cat0("edge_attr"),
        "cnx_attr": cat0("cnx_attr"),
        "batch": torch.repeat_interleave(mol_id, torch.as_tensor(n_atoms)),
        "frag_batch": torch.repeat_interleave(mol_id, torch.as_tensor(n_frags)),
        "atom_to_frag_ids": cat0("atom_id_frag_id").to(torch.long) + _offsets(n_frags, n_atoms),
        "node_features_bonds": cat0("node_features_bonds"),
        "edge_index_bonds_graph": cat1("edge_index_bonds").to(torch.long) + _offsets(n_bnodes, eb),
        "edge_attr_bonds": cat0("edge_attr_bonds"),
        "node_features_fbonds": cat0("node_feautures_fbondg"),
        "edge_index_fbonds": cat1("edge_index_fbondg").to(torch.long) + _offsets(n_fbnodes, efb),
        "edge_attr_fbonds": cat0("edge_attr_fbondg"),
    })
    # the cumulative counts as a table (plan.CollatedBatch): the one-launch graph-plan builder is driven by it
    per_mol = {"atom": n_atoms, "edge": e, "bedge": eb, "frag": n_frags, "fedge": ef, "fbedge": efb}
    out.offsets = mol_offsets({k: torch.as_tensor(v, dtype=torch.long) for k, v in per_mol.items()})
    out.max_per_mol = {k: max(v) for k, v in per_mol.items()}
    out.max_per_mol["mol"] = 1
    return out


def collate_fn(data_list: List) -> Dict[str, torch.Tensor]:
    """Finetune batch dict (16 keys) -- reference data.py:877-948."""
    out = _collate_common(data_list)
    out["y"] = torch.cat([d.y for d in data_list], dim=0).type(torch.float)
    return out


def collate_fn_pt(data_list: List) -> Dict[str, torch.Tensor]:
    """Pretrain batch dict (19 keys) -- reference data.py:951-1032."""
    out = _collate_common(data_list)
    out["bnd_lngth"] = torch.cat([d.bnd_lngth for d in data_list], dim=0)
    out["bnd_angl"] = torch.cat([d.bnd_angl for d in data_list], dim=0)
    out["dh_angl"] = torch.cat([d.dh_angl for d in data_list], dim=0)
    out["y"] = torch.cat([d.y for d in data_list], dim=0).type(torch.float)
    return out


BATCH_KEYS_FT = (
    "x_atoms", "edge_index", "frag_index", "x_frags", "edge_attr", "cnx_attr", "batch",
    "frag_batch", "atom_to_frag_ids", "node_features_bonds", "edge_index_bonds_graph",
    "edge_attr_bonds", "node_features_fbonds", "edge_index_fbonds", "edge_attr_fbonds", "y",
)
BATCH_KEYS_PT = BATCH_KEYS_FT[:-1] + ("bnd_lngth", "bnd_angl", "dh_angl", "y")


def batch_to(batch: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """``batch[k] = batch[k].to(device)`` for every key -- reference train/utils.py:335-336.  A batch collated from a
    store without its bond-graph index (dataset.FlatMolStore.without_bond_graph_index) gets ``edge_index_bonds_graph``
    rebuilt from ``edge_index`` once it is on the GPU (ops.bond_graph)."""
    out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    if isinstance(batch, CollatedBatch):         # the layout promise (and the offsets table) travels with the batch
        out = batch.like(out)
        if out.offsets is not None:
            out.offsets = out.offsets.to(device)
    if "edge_index_bonds_graph" not in out and "edge_index" in out and out["edge_index"].is_cuda:
        from . import ops
        out["edge_index_bonds_graph"] = ops.bond_graph(out["edge_index"], out["batch"], int(out["y"].shape[0]))
    if "edge_index_fbonds" not in out and "frag_index" in out and out["frag_index"].is_cuda:
        from . import ops
        eifb = ops.bond_graph(out["frag_index"], out["frag_batch"], int(out["y"].shape[0]), fragments=True)
        out["edge_index_fbonds"] = eifb
        out["edge_attr_fbonds"] = out["node_features_fbonds"][eifb[0]] + out["node_features_fbonds"][eifb[1]]
    return out
