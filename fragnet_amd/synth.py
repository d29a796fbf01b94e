"""Deterministic synthetic molecule generator (SURVEY.md §8d "Concrete synthetic inputs").

There is no RDKit and no dataset in the build image, so "ESOL-shape" is *defined* by
this generator: a random valence-capped heavy-atom tree, 0-2 ring closures, explicit H
atoms, the bond graph / fragment graph / fragment-bond graph built by the same rules the
reference's offline featuriser uses, and random features of the reference's widths.

Topology rules restated from the reference (no code copied):
  * directed ``edge_index``: every bond twice, (a->b, b->a) interleaved
    -- reference fragnet/dataset/feature_utils.py:285-296
  * bond graph: ordered pairs (i, j) of directed bonds sharing exactly one atom
    -- reference fragnet/dataset/data.py:116-128; two-atom components get the pair of
    mutual edges with attribute 1 -- data.py:157-182,192-195
  * fragments numbered by smallest atom index (RDKit ``GetMolFrags`` order);
    ``atom_id_frag_id`` in atom order -- fragnet/dataset/fragments.py:205-252
  * one connection per cut bond, both directions; a single-fragment molecule gets the
    one self connection (0, 0) -- data.py:505-538, fragments.py:230-234;
    disconnected components are linked by extra connections -- fragments.py:236-240,274-300
  * fragment-bond graph: 2-node special case, otherwise "share exactly one fragment"
    -- data.py:131-154; its edge attribute is the sum of the two node features -- data.py:291-303

A record is a ``MolRecord`` with the attribute names of the reference's per-molecule
``Data`` object (data.py:437-480) so that the reference's own ``collate_fn`` can consume
it when golden vectors are generated.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

ATOM_F = 167   # exps/ft/esol/e1pt4.yaml:8
EDGE_F = 17    # exps/ft/esol/e1pt4.yaml:10
CNX_F = 6      # fedge_in / fbond_edge_in, e1pt4.yaml:11-12

# name -> (mu heavy atoms, p_cut, n_tasks)   -- SURVEY.md §8d profiles
PROFILES = {
    "esol": (10.5, 0.35, 0),
    "tox21": (14.5, 0.35, 12),
    "synth40": (16.0, 0.78, 0),
}


@dataclass
class MolRecord:
    """Per-molecule tensors, attribute names as in the reference ``Data`` item."""
    x_atoms: torch.Tensor                # [n, 167] f32
    edge_index: torch.Tensor             # [2, e]  i64   row0=src,row1=dst
    edge_attr: torch.Tensor              # [e, 17] f32
    frag_index: torch.Tensor             # [2, ef] i64
    cnx_attr: torch.Tensor               # [ef, 6] f32
    x_frags: torch.Tensor                # [f, 167] f32
    atom_id_frag_id: torch.Tensor        # [n] i64
    n_frags: torch.Tensor                # [1] i64
    node_features_bonds: torch.Tensor    # [e, 17] f32
    edge_index_bonds: torch.Tensor       # [2, eb] i32
    edge_attr_bonds: torch.Tensor        # [eb, 1] f32
    node_feautures_fbondg: torch.Tensor  # [ef, 6] f32   (sic: reference spelling)
    edge_index_fbondg: torch.Tensor      # [2, efb] i32
    edge_attr_fbondg: torch.Tensor       # [efb, 6] f32
    y: torch.Tensor                      # [1] or [1, T] f32
    bnd_lngth: Optional[torch.Tensor] = None   # [e, 1]
    bnd_angl: Optional[torch.Tensor] = None    # [n, 1]
    dh_angl: Optional[torch.Tensor] = None     # [e, 1]
    smiles: str = ""


def bond_graph_pairs(ends: np.ndarray) -> np.ndarray:
    """Ordered pairs (i, j), i-major, of rows of ``ends`` ([k, 2]) sharing exactly one id."""
    k = ends.shape[0]
    if k == 0:
        return np.zeros((2, 0), dtype=np.int64)
    a = ends[:, 0][:, None]
    b = ends[:, 1][:, None]
    c = ends[:, 0][None, :]
    d = ends[:, 1][None, :]
    # |set(b_i) & set(b_j)| for 2-element lists (elements of one row may coincide: self edge)
    same_i = (ends[:, 0] == ends[:, 1])
    n_common = np.zeros((k, k), dtype=np.int64)
    # distinct elements of row i that also occur in row j
    in_j_a = (a == c) | (a == d)
    in_j_b = (b == c) | (b == d)
    n_common = in_j_a.astype(np.int64) + (in_j_b & ~same_i[:, None]).astype(np.int64)
    ii, jj = np.nonzero(n_common == 1)
    return np.stack([ii, jj]).astype(np.int64)


def fbond_graph_pairs(ends: np.ndarray) -> np.ndarray:
    """Fragment-bond graph: the reference's 2-node special case, else 'share exactly one'."""
    k = ends.shape[0]
    if k == 2:
        rows = [[], []]
        for i in range(2):
            for j in range(2):
                if not (ends[i, 0] == ends[j, 0] and ends[i, 1] == ends[j, 1]):
                    rows[0].append(i)
                    rows[1].append(j)
        return np.asarray(rows, dtype=np.int64).reshape(2, -1)
    return bond_graph_pairs(ends)


def _components(n: int, bonds: List[tuple]) -> np.ndarray:
    """Connected-component label per atom, components numbered by smallest atom index."""
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in bonds:
        ra, rb = find(a), find(b)
        if ra != rb:
            if ra < rb:
                parent[rb] = ra
            else:
                parent[ra] = rb
    roots = np.array([find(i) for i in range(n)])
    _, labels = np.unique(roots, return_inverse=True)   # roots are the smallest members
    return labels.astype(np.int64)


def _multi_hot(rng, rows: int, width: int, ones: int) -> np.ndarray:
    out = np.zeros((rows, width), dtype=np.float32)
    if rows:
        cols = np.argsort(rng.random((rows, width)), axis=1)[:, :ones]
        np.put_along_axis(out, cols, 1.0, axis=1)
    return out


def make_topology(rng, mu: float, p_cut: float, p_salt: float = 0.0):
    """Returns (n_atoms, bonds[list of (a,b)], cut[list of bool per bond])."""
    n_heavy = max(2, int(rng.poisson(mu)))
    cap = rng.choice([4, 3, 2, 1], size=n_heavy, p=[0.70, 0.10, 0.15, 0.05])
    cap[0] = max(cap[0], 1)
    deg = np.zeros(n_heavy, dtype=np.int64)
    bonds: List[tuple] = []
    for i in range(1, n_heavy):
        free = np.nonzero(deg[:i] < cap[:i])[0]
        if free.size == 0:           # every earlier atom is saturated: widen the newest one
            j = i - 1
            cap[j] = deg[j] + 1
        else:
            j = int(free[rng.integers(free.size)])
        bonds.append((j, i))
        deg[j] += 1
        deg[i] += 1
        cap[i] = max(cap[i], 1)
    n_tree = len(bonds)
    in_ring = [False] * n_tree
    # 0-2 ring closures between non-adjacent atoms with free valence
    for _ in range(int(rng.integers(0, 3))):
        free = np.nonzero(deg < cap)[0]
        if free.size < 2:
            break
        a, b = (int(v) for v in rng.choice(free, size=2, replace=False))
        if a > b:
            a, b = b, a
        if (a, b) in bonds or (b, a) in bonds:
            continue
        # tree path a..b becomes cyclic: those bonds may no longer be cut
        par = {}
        for k, (p, c) in enumerate(bonds[:n_tree]):
            par[c] = (p, k)
        anc_a = {a: None}
        x = a
        while x in par:
            x = par[x][0]
            anc_a[x] = None
        y = b
        path_b = []
        while y not in anc_a:
            p, k = par[y]
            path_b.append(k)
            y = p
        x = a
        while x != y:
            p, k = par[x]
            in_ring[k] = True
            x = p
        for k in path_b:
            in_ring[k] = True
        bonds.append((a, b))
        in_ring.append(True)
        deg[a] += 1
        deg[b] += 1
    n_heavy_bonds = len(bonds)
    cut = [(not in_ring[k]) and (rng.random() < p_cut) for k in range(n_heavy_bonds)]
    # explicit hydrogens fill the remaining valence
    n = n_heavy
    for a in range(n_heavy):
        for _ in range(int(cap[a] - deg[a])):
            bonds.append((a, n))
            cut.append(False)
            n += 1
    salt_atoms = 0
    if p_salt > 0 and rng.random() < p_salt:
        if rng.random() < 0.5:        # lone counter-ion (no bonds)
            # must not be the last atom: the reference rejects a molecule whose last atom
            # has no bond (data.py:368-371).  Insert by swapping ids with the last atom.
            salt_atoms = 1
        else:                          # two-atom component (one-bond fragment)
            bonds.append((n, n + 1))
            cut.append(False)
            n += 2
    if salt_atoms:
        # new atom takes the id of the current last atom; the last atom moves to the end
        last = n - 1
        ion = last
        moved = n
        bonds = [((moved if a == last else a), (moved if b == last else b)) for a, b in bonds]
        n += 1
        _ = ion
    return n, bonds, cut


def make_molecule(rng, mu: float = 10.5, p_cut: float = 0.35, n_tasks: int = 0,
                  pretrain_targets: bool = False, p_salt: float = 0.0,
                  topology=None) -> MolRecord:
    if topology is None:
        n, bonds, cut = make_topology(rng, mu, p_cut, p_salt)
    else:
        n, bonds, cut = topology
    nb = len(bonds)
    e = 2 * nb
    src = np.empty(e, dtype=np.int64)
    dst = np.empty(e, dtype=np.int64)
    barr = np.asarray(bonds, dtype=np.int64).reshape(nb, 2)
    src[0::2], dst[0::2] = barr[:, 0], barr[:, 1]
    src[1::2], dst[1::2] = barr[:, 1], barr[:, 0]
    edge_index = np.stack([src, dst])
    ends = edge_index.T.copy()                      # bond-graph node k = directed edge k

    # ---- bond graph
    eib = bond_graph_pairs(ends)
    cos = rng.uniform(-1.0, 1.0, size=eib.shape[1])
    mol_comp = _components(n, bonds)
    comp_size = np.bincount(mol_comp, minlength=mol_comp.max() + 1)
    extra = [[], []]
    for k in range(nb):
        a, b = bonds[k]
        if comp_size[mol_comp[a]] == 2:
            extra[0] += [2 * k, 2 * k + 1]
            extra[1] += [2 * k + 1, 2 * k]
    if extra[0]:
        eib = np.concatenate([eib, np.asarray(extra, dtype=np.int64)], axis=1)
        cos = np.concatenate([cos, np.ones(len(extra[0]))])

    # ---- fragments
    kept = [bonds[k] for k in range(nb) if not cut[k]]
    a2f = _components(n, kept)
    n_frags = int(a2f.max()) + 1
    cnx: List[tuple] = [(int(a2f[bonds[k][0]]), int(a2f[bonds[k][1]])) for k in range(nb) if cut[k]]
    if not cnx and n_frags == 1:
        cnx = [(0, 0)]
    n_mol_comp = int(mol_comp.max()) + 1
    if n_mol_comp > 1:
        frag_comp = np.zeros(n_frags, dtype=np.int64)
        frag_comp[a2f] = mol_comp
        have = {tuple(sorted(c)) for c in cnx}
        for ci in range(n_mol_comp):
            for cj in range(ci + 1, n_mol_comp):
                for fi in np.nonzero(frag_comp == ci)[0]:
                    for fj in np.nonzero(frag_comp == cj)[0]:
                        if tuple(sorted((int(fi), int(fj)))) not in have:
                            cnx.append((int(fi), int(fj)))
    one_hot_c = np.zeros((len(cnx), CNX_F), dtype=np.float32)
    one_hot_c[np.arange(len(cnx)), rng.integers(0, CNX_F, size=len(cnx))] = 1.0
    if n_frags == 1 and len(cnx) == 1 and cnx[0] == (0, 0):
        frag_index = np.asarray([[0], [0]], dtype=np.int64)
        cnx_attr = one_hot_c
    else:
        fs, fd, rows = [], [], []
        for k, (a, b) in enumerate(cnx):
            fs += [a, b]
            fd += [b, a]
            rows += [k, k]
        frag_index = np.asarray([fs, fd], dtype=np.int64)
        cnx_attr = one_hot_c[rows]
    fends = frag_index.T.copy()
    eifb = fbond_graph_pairs(fends)
    ea_fb = cnx_attr[eifb[0]] + cnx_attr[eifb[1]] if eifb.shape[1] else np.zeros((0, CNX_F), np.float32)

    # ---- features
    x_atoms = _multi_hot(rng, n, ATOM_F, 10)
    bond_feat = _multi_hot(rng, nb, EDGE_F, 5)
    edge_attr = np.repeat(bond_feat, 2, axis=0)
    x_frags = np.zeros((n_frags, ATOM_F), dtype=np.float32)
    np.add.at(x_frags, a2f, x_atoms)
    if n_tasks:
        y = rng.integers(-1, 2, size=(1, n_tasks)).astype(np.float32)
    else:
        y = rng.normal(size=(1,)).astype(np.float32)

    t = torch.from_numpy
    rec = MolRecord(
        x_atoms=t(x_atoms), edge_index=t(edge_index), edge_attr=t(edge_attr),
        frag_index=t(frag_index), cnx_attr=t(cnx_attr.astype(np.float32)), x_frags=t(x_frags),
        atom_id_frag_id=t(a2f), n_frags=torch.tensor([n_frags], dtype=torch.long),
        node_features_bonds=t(edge_attr.copy()),
        edge_index_bonds=t(eib.astype(np.int32)),
        edge_attr_bonds=t(cos.astype(np.float32).reshape(-1, 1)),
        node_feautures_fbondg=t(cnx_attr.astype(np.float32).copy()),
        edge_index_fbondg=t(eifb.astype(np.int32)),
        edge_attr_fbondg=t(ea_fb.astype(np.float32)),
        y=t(y),
    )
    if pretrain_targets:
        rec.bnd_lngth = t(rng.uniform(0.8, 2.5, size=(e, 1)).astype(np.float32))
        rec.bnd_angl = t(rng.uniform(0.0, 4.0, size=(n, 1)).astype(np.float32))
        rec.dh_angl = t(rng.uniform(-1.0, 1.0, size=(e, 1)).astype(np.float32))
    return rec


def synth_molecules(B: int, seed: int, profile: str = "esol", pretrain_targets: bool = False,
                    p_salt: float = 0.0) -> List[MolRecord]:
    mu, p_cut, n_tasks = PROFILES[profile]
    rng = np.random.default_rng(seed)
    return [make_molecule(rng, mu, p_cut, n_tasks, pretrain_targets, p_salt) for _ in range(B)]


# ----------------------------------------------------------------------------------------
# The one molecule whose fragmentation the reference publishes:
# CC[NH+](CCCl)CCOc1cccc2ccccc12.[Cl-] with explicit H (41 atoms, 7 fragments),
# fragnet/notebooks/FragNet.ipynb cell 34 (atom -> fragment map), cell 28 (bond list),
# cells 43/46 (11 connections and the fragment-bond node order).
# ----------------------------------------------------------------------------------------
NOTEBOOK_ATOMS_IN_FRAGS = {
    0: [0, 1, 20, 21, 22, 23, 24],
    1: [2, 25],
    2: [3, 4, 5, 26, 27, 28, 29],
    3: [6, 7, 30, 31, 32, 33],
    4: [8],
    5: [9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 34, 35, 36, 37, 38, 39, 40],
    6: [19],
}
# heavy-atom bonds 0..19 then X-H bonds (notebook cell 28 lists (index, atom1, atom2))
NOTEBOOK_BONDS = [
    (0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (2, 6), (6, 7), (7, 8), (8, 9), (9, 10),
    (10, 11), (11, 12), (12, 13), (13, 14), (14, 15), (15, 16), (16, 17), (17, 18),
    (18, 9), (18, 13),
    (0, 20), (0, 21), (0, 22), (1, 23), (1, 24), (2, 25), (3, 26), (3, 27), (4, 28),
    (4, 29), (6, 30), (6, 31), (7, 32), (7, 33), (10, 34), (11, 35), (12, 36), (14, 37),
    (15, 38), (16, 39), (17, 40),
]
# fragment-bond nodes 0,2,4,...,20 (begin, end) -- notebook cell 46; odd nodes are the reverses
NOTEBOOK_CONNECTIONS = [(4, 3), (4, 5), (0, 1), (2, 1), (3, 1), (0, 6), (1, 6), (2, 6), (3, 6), (4, 6), (5, 6)]


def notebook_molecule(seed: int = 7, pretrain_targets: bool = False) -> MolRecord:
    """Topology of the notebook molecule with random features (no RDKit here)."""
    rng = np.random.default_rng(seed)
    n = 41
    a2f = np.empty(n, dtype=np.int64)
    for f, atoms in NOTEBOOK_ATOMS_IN_FRAGS.items():
        a2f[atoms] = f
    bonds = list(NOTEBOOK_BONDS)
    # a bond is "cut" iff its atoms sit in different fragments (5 BRICS bonds)
    cut = [bool(a2f[a] != a2f[b]) for a, b in bonds]
    rec = make_molecule(rng, n_tasks=0, pretrain_targets=pretrain_targets, topology=(n, bonds, cut))
    # impose the published connection order (BRICS bonds first, then the iso links)
    fs, fd = [], []
    for a, b in NOTEBOOK_CONNECTIONS:
        fs += [a, b]
        fd += [b, a]
    frag_index = np.asarray([fs, fd], dtype=np.int64)
    one_hot = np.zeros((len(NOTEBOOK_CONNECTIONS), CNX_F), dtype=np.float32)
    one_hot[np.arange(len(NOTEBOOK_CONNECTIONS)), rng.integers(0, CNX_F, size=len(NOTEBOOK_CONNECTIONS))] = 1.0
    cnx_attr = np.repeat(one_hot, 2, axis=0)
    eifb = fbond_graph_pairs(frag_index.T.copy())
    rec.frag_index = torch.from_numpy(frag_index)
    rec.cnx_attr = torch.from_numpy(cnx_attr)
    rec.node_feautures_fbondg = torch.from_numpy(cnx_attr.copy())
    rec.edge_index_fbondg = torch.from_numpy(eifb.astype(np.int32))
    rec.edge_attr_fbondg = torch.from_numpy(cnx_attr[eifb[0]] + cnx_attr[eifb[1]])
    assert torch.equal(rec.atom_id_frag_id, torch.from_numpy(a2f))
    return rec
