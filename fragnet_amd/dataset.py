"""Flat molecule store + vectorised batch assembly.

The reference keeps a dataset as a pickled list of torch_geometric ``Data`` objects and assembles every
batch with Python loops over ``torch.cat`` (fragnet/dataset/dataset.py:273-292, fragnet/dataset/data.py:877-948).
Here a dataset is ONE set of concatenated tensors plus per-molecule offsets (a CSR of molecules); a batch is
assembled by a handful of ragged gathers and offset additions, on the CPU or directly on the GPU when the
store lives there (no host->device copy per step).  ``FlatMolStore.collate(indices)`` returns exactly the dict
``fragnet_amd.data.collate_fn([records[i] for i in indices])`` returns (tests/test_dataset.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .plan import CollatedBatch, mol_offsets

# field -> (ragged axis, index space that offsets its VALUES or None)
_ROW_FIELDS = {           # concatenated along dim 0
    "x_atoms": "atom", "edge_attr": "edge", "cnx_attr": "fedge", "x_frags": "frag", "atom_id_frag_id": "atom",
    "node_features_bonds": "edge", "edge_attr_bonds": "bedge", "node_feautures_fbondg": "fedge",
    "edge_attr_fbondg": "fbedge", "bnd_lngth": "edge", "bnd_angl": "atom", "dh_angl": "edge",
}
_COL_FIELDS = {"edge_index": "edge", "frag_index": "fedge", "edge_index_bonds": "bedge", "edge_index_fbondg": "fbedge"}
_COUNT_OF = {"atom": "x_atoms", "edge": "edge_attr", "fedge": "cnx_attr", "frag": "x_frags", "bedge": "edge_attr_bonds",
             "fbedge": "edge_attr_fbondg"}


FUSED_COLLATE = True      # GPU stores, host indices: the batch in one launch (fn_collate_store); False: the torch path (A/B, tests)


def _ragged_rows(offsets: torch.Tensor, idx: torch.Tensor, total: Optional[int] = None):
    """Row indices of the concatenation of segments idx[0], idx[1], ...; also the per-segment lengths.  ``total`` = the
    number of rows when the caller knows it on the host (else it is read back from the device: a synchronisation)."""
    start = offsets[idx]
    length = offsets[idx + 1] - start
    if total is None:
        total = int(length.sum())
    seg = torch.repeat_interleave(torch.arange(idx.numel(), device=idx.device), length, output_size=total)
    first = torch.cumsum(length, 0) - length
    rows = start[seg] + (torch.arange(total, device=idx.device) - first[seg])
    return rows, length, seg


class FlatMolStore:
    """Concatenated per-molecule tensors (attribute names of the reference ``Data`` item, data.py:437-480)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], offsets: Dict[str, torch.Tensor], y: torch.Tensor,
                 smiles: Optional[List[str]] = None):
        self.t, self.off, self.y, self.smiles = tensors, offsets, y, smiles
        self.n = int(y.shape[0])
        self.has_pretrain_targets = "bnd_lngth" in tensors

    def __len__(self):
        return self.n

    @property
    def device(self):
        return self.y.device

    @classmethod
    def from_records(cls, records: Sequence) -> "FlatMolStore":
        names = [f for f in list(_ROW_FIELDS) + list(_COL_FIELDS) if getattr(records[0], f, None) is not None]
        tensors = {}
        for f in names:
            dim = 1 if f in _COL_FIELDS else 0
            tensors[f] = torch.cat([getattr(r, f) for r in records], dim=dim)
            if f in _COL_FIELDS:
                tensors[f] = tensors[f].to(torch.long).contiguous()
        offsets = {}
        for space, field in _COUNT_OF.items():
            counts = torch.tensor([int(getattr(r, field).shape[0]) for r in records], dtype=torch.long)
            offsets[space] = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
        y = torch.cat([r.y.reshape(1, -1) if r.y.dim() > 1 else r.y.reshape(1) for r in records], dim=0).to(torch.float)
        smiles = [getattr(r, "smiles", "") for r in records]
        return cls(tensors, offsets, y, smiles)

    def to(self, device) -> "FlatMolStore":
        return FlatMolStore({k: v.to(device) for k, v in self.t.items()}, {k: v.to(device) for k, v in self.off.items()},
                            self.y.to(device), self.smiles)

    def save(self, path: str):
        torch.save({"tensors": {k: v.cpu() for k, v in self.t.items()}, "offsets": {k: v.cpu() for k, v in self.off.items()},
                    "y": self.y.cpu(), "smiles": self.smiles, "format": "fragnet_amd.flat.v1"}, path)

    @classmethod
    def load(cls, path: str, device=None) -> "FlatMolStore":
        blob = torch.load(path, map_location="cpu", weights_only=False)
        if blob.get("format") != "fragnet_amd.flat.v1":
            raise ValueError(f"{path}: not a fragnet_amd flat store")
        store = cls(blob["tensors"], blob["offsets"], blob["y"], blob.get("smiles"))
        return store.to(device) if device is not None else store

    def replicate(self, k: int) -> "FlatMolStore":
        """A store holding ``k`` back-to-back copies of every molecule (molecule i of copy j = index j * n + i): builds the
        million-molecule stores of the throughput sweeps from a few thousand distinct synthetic molecules."""
        if k < 1:
            raise ValueError("replicate: k must be >= 1")
        tensors = {f: (v.repeat(1, k) if f in _COL_FIELDS else v.repeat((k,) + (1,) * (v.dim() - 1))) for f, v in self.t.items()}
        offsets = {}
        for space, off in self.off.items():
            total = off[-1]
            offsets[space] = torch.cat([off[:1]] + [off[1:] + j * total for j in range(k)])
        y = self.y.repeat((k,) + (1,) * (self.y.dim() - 1))
        return FlatMolStore(tensors, offsets, y, None if self.smiles is None else self.smiles * k)

    DERIVED = ("edge_index_bonds", "edge_index_fbondg", "edge_attr_fbondg")

    def without_bond_graph_index(self) -> "FlatMolStore":
        """The same store minus the tensors that are pure functions of the rest: ``edge_index_bonds`` (its largest tensor, 16
        bytes per bond-graph edge), ``edge_index_fbondg`` and ``edge_attr_fbondg``.  ``collate`` then rebuilds
        ``edge_index_bonds_graph`` from ``edge_index`` and ``edge_index_fbonds`` / ``edge_attr_fbonds`` from ``frag_index``
        and the connection features on the GPU (ops.bond_graph, SURVEY §8 row f4: the reference's pair rules in the
        reference's order, so the stored cos(theta) rows ``edge_attr_bonds`` still line up)."""
        return FlatMolStore({k: v for k, v in self.t.items() if k not in self.DERIVED}, self.off, self.y, self.smiles)

    def bond_graph_edges(self) -> torch.Tensor:
        """Per-molecule bond-graph edge counts: the dominant cost, used to balance shards (parallel.shard_indices)."""
        return (self.off["bedge"][1:] - self.off["bedge"][:-1]).cpu()

    def collate(self, indices, pretrain: bool = False) -> Dict[str, torch.Tensor]:
        dev = self.device
        # Row totals of the batch from a HOST copy of the per-molecule lengths when the indices arrive on the host (a sampler's do):
        # reading them back from the device was seven synchronisations per batch, each waiting for the training step enqueued
        # before it -- collate and step ran strictly one after the other (1.75 ms per step where the step alone is 0.78)
        totals = {}
        idx = None
        if dev.type == "cuda" and not (torch.is_tensor(indices) and indices.is_cuda):
            host_idx = torch.as_tensor(indices, dtype=torch.long)
            if host_idx.numel() == 0:
                raise ValueError("collate: empty batch")
            ix = host_idx.numpy()          # numpy, not torch: a CPU gather of 8192 elements goes through torch's thread pool, and
            # (numpy would wrap a negative index round to the store's end -- offs[-1] is the total row count -- and the kernel
            # would then read past every store tensor; the torch path below raised for it, so does this one)
            if int(ix.min()) < 0 or int(ix.max()) >= len(self):
                raise IndexError(f"collate: molecule index out of range for a store of {len(self)} molecules")
            # ONE gather of the molecules' records (lengths | first store rows of every index space, a row of <= 128 bytes per molecule):
            # fourteen separate gathers out of 8-MB arrays were ~1.3 ms of cache misses a batch of 8192 on a store of a million molecules
            cols, table = self._host_table()
            rec = table[ix]
            totals = {space: int(rec[:, cols[space]].sum()) for space in _COUNT_OF}      # waking it cost up to 90 ms a batch beside the GPU work
            if FUSED_COLLATE:
                fused = self._collate_fused(ix, pretrain, rec, cols)      # reads the host indices only: nothing is uploaded for it
                if fused is not None:
                    return fused
            # through pinned memory (torch's caching host allocator): an asynchronous copy from pageable memory of 64 KB and more
            # is pinned on the fly by the runtime, ~100 ms a time with a 66-GB store mapped (batches of 8192 molecules)
            idx = host_idx.pin_memory().to(dev, non_blocking=True)
        else:
            idx = torch.as_tensor(indices, dtype=torch.long, device=dev)
        if idx.numel() == 0:
            raise ValueError("collate: empty batch")
        rows, length, seg = {}, {}, {}
        for space in _COUNT_OF:
            rows[space], length[space], seg[space] = _ragged_rows(self.off[space], idx, totals.get(space))
        base = {s: torch.cumsum(length[s], 0) - length[s] for s in ("atom", "frag", "edge", "fedge")}
        t = self.t
        out = CollatedBatch({
            "x_atoms": t["x_atoms"][rows["atom"]],
            "edge_index": t["edge_index"][:, rows["edge"]] + base["atom"][seg["edge"]],
            "frag_index": t["frag_index"][:, rows["fedge"]] + base["frag"][seg["fedge"]],
            "x_frags": t["x_frags"][rows["frag"]],
            "edge_attr": t["edge_attr"][rows["edge"]],
            "cnx_attr": t["cnx_attr"][rows["fedge"]],
            "batch": seg["atom"],
            "frag_batch": seg["frag"],
            "atom_to_frag_ids": t["atom_id_frag_id"][rows["atom"]] + base["frag"][seg["atom"]],
            "node_features_bonds": t["node_features_bonds"][rows["edge"]],
            "edge_index_bonds_graph": None,      # filled below (stored index, or rebuilt on the device)
            "edge_attr_bonds": t["edge_attr_bonds"][rows["bedge"]],
            "node_features_fbonds": t["node_feautures_fbondg"][rows["fedge"]],
            "edge_index_fbonds": None,           # filled below too
            "edge_attr_fbonds": None,
        })
        if "edge_index_bonds" in t:
            out["edge_index_bonds_graph"] = t["edge_index_bonds"][:, rows["bedge"]] + base["edge"][seg["bedge"]]
        elif dev.type == "cuda":          # topology rebuilt on the device; a CPU store leaves it to data.batch_to(batch, gpu)
            from . import ops
            out["edge_index_bonds_graph"] = ops.bond_graph(out["edge_index"], out["batch"], int(idx.numel()))
        else:
            del out["edge_index_bonds_graph"]
        if "edge_index_fbondg" in t:
            out["edge_index_fbonds"] = t["edge_index_fbondg"][:, rows["fbedge"]] + base["fedge"][seg["fbedge"]]
            out["edge_attr_fbonds"] = t["edge_attr_fbondg"][rows["fbedge"]]
        elif dev.type == "cuda":
            from . import ops
            eifb = ops.bond_graph(out["frag_index"], out["frag_batch"], int(idx.numel()), fragments=True)
            out["edge_index_fbonds"] = eifb
            out["edge_attr_fbonds"] = out["node_features_fbonds"][eifb[0]] + out["node_features_fbonds"][eifb[1]]      # data.py:291-303
        else:
            del out["edge_index_fbonds"], out["edge_attr_fbonds"]
        if pretrain:
            out["bnd_lngth"] = t["bnd_lngth"][rows["edge"]]
            out["bnd_angl"] = t["bnd_angl"][rows["atom"]]
            out["dh_angl"] = t["dh_angl"][rows["edge"]]
        out["y"] = self.y[idx]
        # the per-molecule offsets of the batch (plan.CollatedBatch.offsets): cumulative lengths, already on the device;
        # molecule extents are bounded by the store's own maxima (no device synchronisation here)
        if all(s in length for s in ("atom", "edge", "bedge", "frag", "fedge", "fbedge")):
            out.offsets = mol_offsets(length)
            out.max_per_mol = self.max_per_mol()
        return out

    # output key -> (store tensor, index space) of the feature rows; index tensors: output key -> (store tensor, space, space pointed into)
    _FUSED_ROWS = {"x_atoms": ("x_atoms", "atom"), "x_frags": ("x_frags", "frag"), "edge_attr": ("edge_attr", "edge"),
                   "cnx_attr": ("cnx_attr", "fedge"), "node_features_bonds": ("node_features_bonds", "edge"),
                   "edge_attr_bonds": ("edge_attr_bonds", "bedge"), "node_features_fbonds": ("node_feautures_fbondg", "fedge"),
                   "edge_attr_fbonds": ("edge_attr_fbondg", "fbedge")}
    _FUSED_PT = {"bnd_lngth": ("bnd_lngth", "edge"), "bnd_angl": ("bnd_angl", "atom"), "dh_angl": ("dh_angl", "edge")}
    _FUSED_IDS = {"edge_index": ("edge_index", "edge", "atom"), "frag_index": ("frag_index", "fedge", "frag"),
                  "edge_index_bonds_graph": ("edge_index_bonds", "bedge", "edge"), "edge_index_fbonds": ("edge_index_fbondg", "fbedge", "fedge"),
                  "atom_to_frag_ids": ("atom_id_frag_id", "atom", "frag")}

    def _collate_fused(self, ix, pretrain: bool, rec, cols):
        """The whole batch in ONE launch (fn_collate_store): the batch's offsets table and the molecules' first store rows are built
        on the host from the store's lengths (numpy, microseconds) and copied over in two small transfers; None when a tensor of
        the store is not laid out the way the kernel reads it (the torch path below then builds the batch)."""
        import ctypes as C
        import numpy as np
        from . import _lib
        from .plan import SPACES, _stream_ptr
        t, dev, B = self.t, self.device, int(ix.shape[0])
        rows_spec = dict(self._FUSED_ROWS)
        if pretrain:
            rows_spec.update(self._FUSED_PT)
        need = [v[0] for v in rows_spec.values()] + [v[0] for v in self._FUSED_IDS.values()]
        if any(k not in t for k in need):
            return None
        for key, (f, _) in rows_spec.items():
            if t[f].element_size() != 4 or not t[f].is_contiguous():
                return None
        for key, (f, _, _) in self._FUSED_IDS.items():
            if t[f].dtype != torch.long or not t[f].is_contiguous():
                return None
        if self.y.element_size() != 4 or not self.y.is_contiguous():
            return None
        S, nsp = len(SPACES), len(cols)
        # one pinned buffer [starts int64 [S, B] | offsets int32 [S, B + 1]] and one device buffer of the same layout: the collate's
        # own first launch copies it (no copy-engine transfer in the step's stream)
        n_st, n_off = S * B, S * (B + 1)
        tab_h, slot = self._pinned_tables(2 * n_st + n_off)
        tab_d = torch.empty(2 * n_st + n_off, dtype=torch.int32, device=dev)
        sn = tab_h[: 2 * n_st].view(torch.long).view(S, B).numpy()
        on = tab_h[2 * n_st:].view(S, B + 1).numpy()
        total = {}
        for s, name in enumerate(SPACES):
            on[s, 0] = 0
            if name == "mol":
                on[s, 1:] = np.arange(1, B + 1)
                sn[s] = ix
                total[name] = B
            else:
                on[s, 1:] = np.cumsum(rec[:, cols[name]])
                sn[s] = rec[:, nsp + cols[name]]
                total[name] = int(on[s, B])
        st_d, off_d = tab_d[: 2 * n_st].view(torch.long).view(S, B), tab_d[2 * n_st:].view(S, B + 1)
        out = CollatedBatch()
        fields = (_lib.CollateField * _lib.FN_MAX_COLLATE_FIELDS)()
        n = 0

        bound = self.max_per_mol()          # a molecule's largest extent per index space: how many chunks its segment is cut into

        def add(dst, src, rows, src_rows, width, space, kind, rebase=0):
            nonlocal n
            fields[n] = _lib.CollateField(None if src is None else src.data_ptr(), dst.data_ptr(), rows, src_rows, width, SPACES.index(space), kind,
                                          SPACES.index(rebase) if rebase else 0, 0, int(bound.get(space, 0)))
            n += 1
        for key, (f, space) in rows_spec.items():
            src = t[f]
            width = src.numel() // max(1, src.shape[0])
            out[key] = torch.empty((total[space],) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
            add(out[key], src, total[space], 0, width, space, _lib.COLLATE_ROWS)
        for key, (f, space, points_into) in self._FUSED_IDS.items():
            src = t[f]
            width = 1 if src.dim() == 1 else int(src.shape[0])
            out[key] = torch.empty(((total[space],) if src.dim() == 1 else (width, total[space])), dtype=torch.long, device=dev)
            add(out[key], src, total[space], int(src.shape[-1]), width, space, _lib.COLLATE_IDS, points_into)
        for key, space in (("batch", "atom"), ("frag_batch", "frag")):
            out[key] = torch.empty((total[space],), dtype=torch.long, device=dev)
            add(out[key], None, total[space], 0, 1, space, _lib.COLLATE_BATCH)
        out["y"] = torch.empty((B,) + tuple(self.y.shape[1:]), dtype=self.y.dtype, device=dev)
        add(out["y"], self.y, B, 0, max(1, self.y.numel() // max(1, self.y.shape[0])), "mol", _lib.COLLATE_ROWS)
        timing = self.__dict__.get("_collate_events")       # measurement hook (bench.py): a list -> (start, end) events around the launches
        if timing is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(dev))
        _lib.call("fn_collate_store", fields, n, st_d.data_ptr(), off_d.data_ptr(), S, B, tab_h.data_ptr(), _stream_ptr(dev))
        if timing is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(torch.cuda.current_stream(dev))
            timing.append((ev0, ev1))
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))      # the pinned buffer is free again once the launch above has read it
        out._keep = (tab_d,)
        out.offsets = off_d
        out.max_per_mol = self.max_per_mol()
        return out

    _PIN_RING = 8

    def _pinned_tables(self, words: int):
        """A pinned int32 buffer of ``words`` for the collate's tables, out of a ring of eight per size: a kernel reads it some time
        after this call returns, so a buffer is handed out again only once the event recorded behind that launch has passed (the
        host then waits -- it is eight batches ahead of the GPU)."""
        ring = self.__dict__.setdefault("_pin_rings", {}).setdefault(words, {"next": 0, "slots": []})
        if len(ring["slots"]) < self._PIN_RING:
            ring["slots"].append([torch.empty(words, dtype=torch.int32, pin_memory=True), None])
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"]]
            ring["next"] = (ring["next"] + 1) % self._PIN_RING
            if slot[1] is not None:
                slot[1].synchronize()
        return slot[0], slot

    def _host_offsets(self):
        """First store row of every molecule in every index space as numpy arrays on the host (copied once)."""
        if getattr(self, "_off_cpu", None) is None:
            self._off_cpu = {s: o.to("cpu", torch.long).numpy() for s, o in self.off.items()}
        return self._off_cpu

    def _host_table(self):
        """(column of every index space, int64 [n_molecules, 2 x spaces]): a molecule's extents in all spaces, then its first store rows --
        one row per molecule, so that a batch's records are one gather (copied once; the store is immutable)."""
        if getattr(self, "_tab_cpu", None) is None:
            import numpy as np
            lens, offs = self._host_lengths(), self._host_offsets()
            names = list(self.off)
            tab = np.empty((len(self), 2 * len(names)), dtype=np.int64)
            for j, s in enumerate(names):
                tab[:, j] = lens[s]
                tab[:, len(names) + j] = offs[s][:-1]
            self._tab_cpu = ({s: j for j, s in enumerate(names)}, tab)
        return self._tab_cpu

    def _host_lengths(self):
        """Per-molecule extent in every index space as numpy arrays on the host (copied once; the store is immutable)."""
        if getattr(self, "_len_cpu", None) is None:
            self._len_cpu = {s: (o[1:] - o[:-1]).to("cpu", torch.long).numpy() for s, o in self.off.items()}
        return self._len_cpu

    def max_per_mol(self) -> Dict[str, int]:
        """Largest extent of one molecule in every index space over the whole store (computed once)."""
        if getattr(self, "_max_per_mol", None) is None:
            m = {s: int((o[1:] - o[:-1]).max()) if o.numel() > 1 else 0 for s, o in self.off.items()}
            m["mol"] = 1
            self._max_per_mol = m
        return self._max_per_mol


class BatchSampler:
    """``DataLoader(shuffle=..., drop_last=...)`` semantics over molecule indices, optionally one shard per rank."""

    def __init__(self, n: int, batch_size: int, shuffle: bool, drop_last: bool, seed: int = 0, rank: int = 0, world: int = 1):
        if world > 1 and batch_size < world:
            raise ValueError(f"batch_size {batch_size} < world size {world}: some ranks would get empty batches")
        self.n, self.bs, self.shuffle, self.drop_last = n, batch_size, shuffle, drop_last
        self.gen = torch.Generator().manual_seed(seed)
        self.rank, self.world = rank, world

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.gen) if self.shuffle else torch.arange(self.n)
        for b in range(0, self.n, self.bs):
            chunk = order[b: b + self.bs]
            if chunk.numel() < self.bs and self.drop_last:
                break
            if self.world > 1 and chunk.numel() < self.world:
                break           # a trailing chunk that cannot give every rank a molecule is dropped on ALL ranks (no rank may skip a collective)
            yield chunk[self.rank:: self.world] if self.world > 1 else chunk

    def __len__(self):
        if self.drop_last:
            return self.n // self.bs
        full, rest = divmod(self.n, self.bs)
        return full + (1 if rest and not (self.world > 1 and rest < self.world) else 0)


# ---------------------------------------------------------------------------------------- pickled datasets
def load_pickle_dataset(path):
    """A pickled list of per-molecule records (the reference's on-disk format, dataset/dataset.py:273-277): entries the
    featuriser rejected are stored as None / empty and dropped here.  Unpickling torch_geometric ``Data`` items needs
    torch_geometric installed; plain records (synth.MolRecord) need nothing."""
    import pickle
    with open(path, "rb") as f:
        items = pickle.load(f)
    return [it for it in items if it]


def load_data_parts(path, select_name=None):
    """Concatenation of every pickled part file in ``path`` whose name contains ``select_name`` (dataset/dataset.py:280-292)."""
    import os
    names = sorted(n for n in os.listdir(path) if not select_name or select_name in n)
    out = []
    for n in names:
        out += load_pickle_dataset(os.path.join(path, n))
    return out
