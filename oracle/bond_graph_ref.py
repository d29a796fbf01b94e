"""ORACLE -- test infrastructure only (see oracle/scatter_ref.py for who may import this).

CPU restatement of the reference's bond-graph TOPOLOGY construction: the
``edge_index_bonds_graph`` of a batch as a pure function of its ``edge_index``.

Follows (no code copied):
  pairs of directed bonds sharing exactly one atom, i-major, j ascending   fragnet/dataset/data.py:116-127
  bond id = position of the directed bond in edge_index                    fragnet/dataset/data.py:381-403
  two-atom connected components ("one-bond fragments", RDKit GetMolFrags order = lowest atom first) append
  (id(a->b), id(b->a)) and (id(b->a), id(a->b)), a < b, after the molecule's pairs   data.py:157-182, 407-410
  batching: bond ids of molecule k are offset by the bonds of the molecules before it   data.py:877-948 (collate_fn)
  fragment-bond graph: a molecule with exactly two connection nodes pairs those whose [begin, end] lists differ, any
  other molecule uses the share-exactly-one rule, no extras                        fragnet/dataset/data.py:131-154

PARITY STATUS: pinned against the reference's own ``get_bond_pair_bond_graph``, ``get_bond_pair_fbond_graph`` and
``add_one_bond_frag_nodes_to_index`` (imported in the build container) on the molecules frozen in
tests/golden/bond_graph_cases.npz (tests/test_bond_graph.py).  ``get_one_bond_frags`` itself is RDKit
(Chem.GetMolFrags) and is restated, not executed: components of exactly two atoms, ordered by lowest atom.
The cos(theta) edge attribute needs 3-D coordinates (data.py:185-211) and is not part of this function.
"""
from __future__ import annotations

import numpy as np


def one_bond_fragments(n_atoms: int, ends) -> list:
    """Connected components with exactly two atoms, as (a, b) with a < b, ordered by a."""
    deg = np.zeros(n_atoms, dtype=np.int64)
    for u, _ in ends:
        deg[u] += 1
    out = []
    for u, v in ends:
        if u < v and deg[u] == 1 and deg[v] == 1:
            out.append((int(u), int(v)))
    return sorted(out)


def bond_graph_one_molecule(n_atoms: int, ends) -> np.ndarray:
    """``ends``: the molecule's directed bonds [(u, v), ...] in edge_index order -> [2, Eb] local bond ids."""
    n = len(ends)
    res = [[], []]
    for i in range(n):
        for j in range(n):
            if len(set(ends[i]) & set(ends[j])) == 1:
                res[0].append(i)
                res[1].append(j)
    ids = {(int(u), int(v)): k for k, (u, v) in enumerate(ends)}
    for a, b in one_bond_fragments(n_atoms, ends):
        id1, id2 = ids[(a, b)], ids[(b, a)]
        res[0] += [id1, id2]
        res[1] += [id2, id1]
    return np.asarray(res, dtype=np.int64).reshape(2, -1)


def fbond_graph_one_molecule(ends) -> np.ndarray:
    """``ends``: the molecule's fragment connections [(begin, end), ...] in frag_index order -> [2, Efb] local ids."""
    n = len(ends)
    res = [[], []]
    for i in range(n):
        for j in range(n):
            hit = (list(ends[i]) != list(ends[j])) if n == 2 else (len(set(ends[i]) & set(ends[j])) == 1)
            if hit:
                res[0].append(i)
                res[1].append(j)
    return np.asarray(res, dtype=np.int64).reshape(2, -1)


def bond_graph_batch(edge_index: np.ndarray, atom_batch: np.ndarray, n_mols: int, fragments: bool = False) -> np.ndarray:
    """Batched ``edge_index`` [2, E] (molecule-contiguous, global atom ids) -> ``edge_index_bonds_graph`` [2, Eb] with
    global bond ids, molecule after molecule.  ``fragments``: frag_index / frag_batch -> edge_index_fbonds."""
    edge_index = np.asarray(edge_index, dtype=np.int64)
    atom_batch = np.asarray(atom_batch, dtype=np.int64)
    mol_of_edge = atom_batch[edge_index[0]] if edge_index.shape[1] else np.zeros(0, dtype=np.int64)
    atom_start = np.searchsorted(atom_batch, np.arange(n_mols + 1))
    out, first = [], 0
    for m in range(n_mols):
        k = int((mol_of_edge == m).sum())
        a0 = int(atom_start[m])
        ends = [(int(u) - a0, int(v) - a0) for u, v in edge_index[:, first:first + k].T]
        out.append((fbond_graph_one_molecule(ends) if fragments else bond_graph_one_molecule(int(atom_start[m + 1]) - a0, ends)) + first)
        first += k
    return np.concatenate(out, axis=1) if out else np.zeros((2, 0), dtype=np.int64)
