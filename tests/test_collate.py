"""fragnet_amd.data.collate_fn / collate_fn_pt against the reference's collate (golden fixture).

Integer index maps must be bit-exact (BASELINE.json north_star); float tensors are plain concatenations.
"""
import os

import numpy as np
import pytest
import torch

from fragnet_amd import data as fdata
from fragnet_amd.synth import MolRecord
from tests.helpers import GOLDEN

FIELDS = ["x_atoms", "edge_index", "edge_attr", "frag_index", "cnx_attr", "x_frags", "atom_id_frag_id", "n_frags",
          "node_features_bonds", "edge_index_bonds", "edge_attr_bonds", "node_feautures_fbondg",
          "edge_index_fbondg", "edge_attr_fbondg", "y", "bnd_lngth", "bnd_angl", "dh_angl"]


def _records(z):
    recs = []
    for i in range(int(z["n_mols"])):
        recs.append(MolRecord(**{f: torch.from_numpy(z[f"mol{i}/{f}"]) for f in FIELDS}))
    return recs


@pytest.fixture(scope="module")
def fixture():
    return np.load(os.path.join(GOLDEN, "collate_edge6.npz"))


@pytest.mark.parametrize("kind", ["ft", "pt", "one"])
def test_collate_bit_exact(fixture, kind):
    recs = _records(fixture)
    if kind == "ft":
        got = fdata.collate_fn(recs)
    elif kind == "pt":
        got = fdata.collate_fn_pt(recs)
    else:
        got = fdata.collate_fn(recs[:1])
    want = {k[len(kind) + 1:]: fixture[k] for k in fixture.files if k.startswith(kind + "/")}
    assert list(got.keys()) == list(want.keys())
    for k, w in want.items():
        g = got[k].numpy()
        assert g.dtype == w.dtype, (k, g.dtype, w.dtype)
        assert g.shape == w.shape, k
        assert np.array_equal(g, w), k


def test_collate_rejects_empty():
    with pytest.raises(ValueError):
        fdata.collate_fn([])


def test_every_last_node_has_an_incoming_edge():
    """The reference's .view(num_nodes, -1) after scatter_add needs index.max()+1 == num_nodes."""
    from fragnet_amd import synth
    for profile in ("esol", "tox21", "synth40"):
        b = fdata.collate_fn(synth.synth_molecules(16, seed=77, profile=profile))
        assert int(b["edge_index_bonds_graph"][0].max()) + 1 == b["node_features_bonds"].shape[0]
        assert int(b["edge_index"].max()) + 1 == b["x_atoms"].shape[0]
        assert int(b["edge_index_fbonds"][0].max()) + 1 == b["node_features_fbonds"].shape[0]
        assert int(b["frag_index"][1].max()) + 1 == b["x_frags"].shape[0]
        assert int(b["atom_to_frag_ids"].max()) + 1 == b["x_frags"].shape[0]


def test_collated_batch_carries_the_layout_promise_and_the_offsets_table(fixture):
    """collate_fn's result is a dict with the reference's keys (above) whose TYPE says "molecules are concatenated" and whose
    attributes hold the cumulative per-molecule counts; copies through dict() drop the promise, .like() / batch_to keep it."""
    from fragnet_amd.plan import SPACES, CollatedBatch
    recs = _records(fixture)
    b = fdata.collate_fn(recs)
    assert isinstance(b, CollatedBatch) and b.mol_contiguous
    off = b.offsets
    assert off.dtype == torch.int32 and tuple(off.shape) == (len(SPACES), len(recs) + 1)
    sizes = {"atom": "x_atoms", "edge": "node_features_bonds", "bedge": "edge_attr_bonds", "frag": "x_frags",
             "fedge": "node_features_fbonds", "fbedge": "edge_attr_fbonds", "mol": "y"}
    for s, name in enumerate(SPACES):
        assert int(off[s, 0]) == 0 and int(off[s, -1]) == b[sizes[name]].shape[0], name
        assert bool((off[s, 1:] >= off[s, :-1]).all())
        assert b.max_per_mol[name] == int((off[s, 1:] - off[s, :-1]).max()), name
    # every atom of molecule i lies in its offsets range, every bond joins atoms of one molecule
    mol_of_atom = b["batch"]
    for i in range(len(recs)):
        a0, a1 = int(off[0, i]), int(off[0, i + 1])
        assert bool((mol_of_atom[a0:a1] == i).all())
        e0, e1 = int(off[1, i]), int(off[1, i + 1])
        ei = b["edge_index"][:, e0:e1]
        assert ei.numel() == 0 or (int(ei.min()) >= a0 and int(ei.max()) < a1)
    assert not isinstance(dict(b), CollatedBatch)
    c = fdata.batch_to(b, "cpu")
    assert isinstance(c, CollatedBatch) and torch.equal(c.offsets, off) and c.max_per_mol == b.max_per_mol
    assert isinstance(b.like(dict(b)), CollatedBatch)
