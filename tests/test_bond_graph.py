"""Bond-graph topology: oracle/bond_graph_ref.py against the reference's own functions (golden
fixture written by tests/golden/make_golden.py bond_graph), and the HIP builder (fn_bond_graph_count / _fill through
ops.bond_graph) against the oracle and against the collated batches' own edge_index_bonds_graph -- bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import bond_graph_ref as ref

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "bond_graph_cases.npz")


def _cases():
    z = np.load(GOLD)
    return {str(n): (int(z[f"{n}/n_atoms"]), z[f"{n}/ends"], z[f"{n}/pairs"]) for n in z["names"]}


def _fcases():
    z = np.load(GOLD)
    return {str(n): (z[f"{n}/ends"], z[f"{n}/pairs"]) for n in z["fnames"]}


def test_oracle_matches_the_reference_functions():
    for name, (n_atoms, ends, pairs) in _cases().items():
        got = ref.bond_graph_one_molecule(n_atoms, [tuple(map(int, e)) for e in ends])
        assert np.array_equal(got, pairs), name
    for name, (ends, pairs) in _fcases().items():
        got = ref.fbond_graph_one_molecule([tuple(map(int, e)) for e in ends])
        assert np.array_equal(got, pairs), name


def _batch_of(cases):
    """Concatenates molecules the way collate_fn does: atom ids and bond ids offset molecule after molecule."""
    src, dst, batch, want, a0, e0 = [], [], [], [], 0, 0
    for m, (n_atoms, ends, pairs) in enumerate(cases):
        src += [int(u) + a0 for u, _ in ends]
        dst += [int(v) + a0 for _, v in ends]
        batch += [m] * n_atoms
        want.append(pairs + e0)
        a0 += n_atoms
        e0 += len(ends)
    ei = np.asarray([src, dst], dtype=np.int64).reshape(2, -1)
    return ei, np.asarray(batch, dtype=np.int64), np.concatenate(want, axis=1) if want else np.zeros((2, 0), np.int64)


def test_oracle_batched_matches_concatenated_golden_and_synthetic_collate():
    cases = list(_cases().values())
    ei, batch, want = _batch_of(cases)
    assert np.array_equal(ref.bond_graph_batch(ei, batch, len(cases)), want)
    from fragnet_amd import data, synth
    b = data.collate_fn(synth.synth_molecules(24, seed=12))
    got = ref.bond_graph_batch(b["edge_index"].numpy(), b["batch"].numpy(), int(b["y"].shape[0]))
    assert np.array_equal(got, b["edge_index_bonds_graph"].numpy())


@gpu
def test_hip_builder_matches_golden_cases_bit_exact():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fragnet_amd import ops
    dev = torch.device("cuda:0")
    cases = list(_cases().values())
    for order in (cases, cases[::-1], cases[:2], [cases[1]], cases[1:2] * 3 + cases[:1]):      # incl. single-atom molecules first / only
        ei, batch, want = _batch_of(order)
        got = ops.bond_graph(torch.from_numpy(ei).to(dev), torch.from_numpy(batch).to(dev), len(order))
        assert got.dtype == torch.int64 and tuple(got.shape) == want.shape
        assert np.array_equal(got.cpu().numpy(), want)


def _fbatch_of(cases):
    src, dst, batch, want, f0, e0 = [], [], [], [], 0, 0
    for m, (ends, pairs) in enumerate(cases):
        n_frag = int(ends.max()) + 1 if len(ends) else 1
        src += [int(u) + f0 for u, _ in ends]
        dst += [int(v) + f0 for _, v in ends]
        batch += [m] * n_frag
        want.append(pairs + e0)
        f0 += n_frag
        e0 += len(ends)
    return (np.asarray([src, dst], dtype=np.int64).reshape(2, -1), np.asarray(batch, dtype=np.int64),
            np.concatenate(want, axis=1) if want else np.zeros((2, 0), np.int64))


@gpu
def test_hip_builder_fragment_mode_matches_golden_and_collate():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fragnet_amd import data, ops, synth
    dev = torch.device("cuda:0")
    cases = list(_fcases().values())
    for order in (cases, cases[::-1], cases[:1], cases[1:2]):
        fi, fb, want = _fbatch_of(order)
        assert np.array_equal(ref.bond_graph_batch(fi, fb, len(order), fragments=True), want)
        got = ops.bond_graph(torch.from_numpy(fi).to(dev), torch.from_numpy(fb).to(dev), len(order), fragments=True)
        assert np.array_equal(got.cpu().numpy(), want)
    b = data.collate_fn(synth.synth_molecules(300, seed=14, p_salt=0.2))
    got = ops.bond_graph(b["frag_index"].to(dev), b["frag_batch"].to(dev), int(b["y"].shape[0]), fragments=True)
    assert torch.equal(got.cpu(), b["edge_index_fbonds"])
    attr = b["node_features_fbonds"].to(dev)[got[0]] + b["node_features_fbonds"].to(dev)[got[1]]      # data.py:291-303
    assert torch.equal(attr.cpu(), b["edge_attr_fbonds"])


@gpu
@pytest.mark.parametrize("n_mols,profile,seed", [(64, "esol", 1), (512, "esol", 2), (256, "tox21", 3)])
def test_hip_builder_reproduces_the_collated_bond_graph(n_mols, profile, seed):
    """Full size (ESOL batch 512: 200 k pairs): the builder's output equals the batch's own edge_index_bonds_graph."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fragnet_amd import data, ops, synth
    dev = torch.device("cuda:0")
    b = data.collate_fn(synth.synth_molecules(n_mols, seed=seed, profile=profile))
    got = ops.bond_graph(b["edge_index"].to(dev), b["batch"].to(dev), int(b["y"].shape[0]))
    assert torch.equal(got.cpu(), b["edge_index_bonds_graph"])


@gpu
def test_store_without_bond_graph_index_collates_the_same_batches():
    """dataset.FlatMolStore.without_bond_graph_index(): GPU-resident and CPU-resident stores give the batches the full
    store gives (every key, bit-exact), with the bond-graph index rebuilt by the HIP builder."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fragnet_amd import data, dataset, synth
    dev = torch.device("cuda:0")
    full = dataset.FlatMolStore.from_records(synth.synth_molecules(96, seed=31, p_salt=0.15))
    lean = full.without_bond_graph_index()
    assert not any(k in lean.t for k in ("edge_index_bonds", "edge_index_fbondg", "edge_attr_fbondg"))
    assert lean.t["edge_attr_bonds"].shape == full.t["edge_attr_bonds"].shape
    idx = [5, 90, 17, 3, 44, 45, 46, 0]
    want = full.collate(idx)
    got_gpu = lean.to(dev).collate(idx)
    got_cpu = data.batch_to(lean.collate(idx), dev)
    for got in (got_gpu, got_cpu):
        assert set(got) == set(want)
        for k, v in want.items():
            assert torch.equal(got[k].cpu(), v), k
