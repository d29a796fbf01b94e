"""ORACLE -- test infrastructure only (see oracle/scatter_ref.py for who may import this).

Pure-torch CPU restatement of the reference's four-level message-passing stack in
*reference-faithful* form: every per-edge tensor the reference materialises
(index_select -> cat -> mul -> sum -> LeakyReLU -> scatter_softmax -> mul -> scatter_add)
is materialised here too, so this module doubles as the "reference CPU path" timed by
bench.py's cpu_baseline leg.

Follows (structure and arithmetic; no code copied):
  FragNetLayerA   fragnet/model/gat/gat2.py:40-330   (ctor order :59-119 = RNG order)
  FragNet         fragnet/model/gat/gat2.py:333-442
  FTHead3/FTHead4 fragnet/model/gat/gat2.py:678-725 / 640-675
  FTHead1/2/5     fragnet/model/gat/gat2.py:569-587 / 727-751 / 590-637
  FragNetFineTune fragnet/model/gat/gat2.py:758-826
  PretrainTask    fragnet/model/gat/pretrain_heads.py:8-102
  FragNetPreTrain fragnet/model/gat/pretrain_heads.py:105-141
  variant="gat2_lite"  fragnet/model/gat/gat2_lite.py:13-217, 467-510 (same constructors; layers stop after L3)
  variant="gat2_edge"  fragnet/model/gat/gat2_edge.py:13-228, 520-561 (no fragment-bond graph; the fragment graph's
                       edge term is Linear(8 -> 128)(cnx_attr))

PARITY STATUS: pinned against the reference's own Python, imported in the build
container with stub third-party modules, on the cases frozen in tests/golden/*.npz
(tests/test_oracle_golden.py).  Unpinned at the torch-scatter boundary (scatter_ref.py).

Module tree, parameter names and construction order are the reference's, including the
parameters that never reach ``forward``, so the same ``torch.manual_seed``
gives the same weights and reference ``state_dict``s load unchanged.  The reference leaves
``bias`` as uninitialised memory (gat2.py:81); it is dead, and is zero-filled here.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .scatter_ref import add_self_loops, scatter_add, scatter_softmax

_ACTS = {
    "relu": nn.ReLU, "silu": nn.SiLU, "gelu": nn.GELU, "celu": nn.CELU, "selu": nn.SELU,
    "rrelu": nn.RReLU, "relu6": nn.ReLU6, "prelu": nn.PReLU, "leakyrelu": nn.LeakyReLU,
}


def _mlp2(width: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(width, 2 * width), nn.ReLU(), nn.Linear(2 * width, width))


def gat_level_materialised(h, edge_vec, att, dst, src, heads, neg_slope=0.2):
    """One attention level exactly as gat2.py:146-169 spells it.

    h [n, heads, d]; edge_vec [m, w] (shared by all heads); att [heads, d + w + d] laid out
    [dst | edge | src]; returns (out [rows, heads, d], probs [m, heads], attn_by_src)."""
    per_head_edge = edge_vec.repeat(heads, 1, 1).permute(1, 0, 2)
    h_src = torch.index_select(h, 0, src)
    h_dst = torch.index_select(h, 0, dst)
    message = torch.cat([h_dst, per_head_edge, h_src], dim=-1)
    logits = torch.nn.functional.leaky_relu(torch.sum(message * att, dim=2), neg_slope)
    probs = scatter_softmax(logits, dst, dim=0)
    h_src_again = torch.index_select(h, 0, src)
    out = scatter_add(probs[..., None] * h_src_again, dst, dim=0)
    attn_by_src = scatter_add(probs, src, dim=0)
    return out, probs, attn_by_src


class FragNetLayerA(nn.Module):
    def __init__(self, atom_in=128, atom_out=128, frag_in=128, frag_out=128, edge_in=128, edge_out=128,
                 fedge_in=128, num_heads=2, bond_edge_in=1, fbond_edge_in=8, return_attentions=False,
                 add_frag_self_loops=False, bond_mask=None, frag_bond_mask=None, atom_mask_individual=None):
        super().__init__()
        self.add_frag_self_loops = add_frag_self_loops
        self.return_attentions = return_attentions
        self.edge_out = edge_out
        self.num_heads = num_heads
        # --- never used by forward, kept for RNG order / state-dict compatibility
        self.atom_embed = nn.Linear(atom_in, atom_out)
        self.frag_embed = nn.Linear(frag_in, frag_out)
        self.edge_embed = nn.Linear(edge_in, edge_out)
        self.bond_edge_embed = nn.Linear(edge_in, edge_out)
        self.frag_message_mlp = nn.Linear(2 * atom_out, atom_out)
        self.atom_mlp = _mlp2(atom_out)
        self.frag_mlp = _mlp2(atom_out)
        self.bias = nn.Parameter(torch.zeros(atom_out))
        self.leakyrelu = nn.LeakyReLU(0.2)
        self.edge_attr_bond_embed2 = nn.Linear(edge_out, edge_out)
        # --- live
        d_e = edge_out // num_heads
        self.projection_b = nn.Linear(edge_in, d_e * num_heads)
        self.projection_fb = nn.Linear(fedge_in, d_e * num_heads)
        self.edge_attr_bond_embed = nn.Linear(bond_edge_in, d_e)
        self.edge_attr_fbond_embed = nn.Linear(fbond_edge_in, d_e)
        d_a = atom_out // num_heads
        self.projection_a = nn.Linear(atom_in, d_a * num_heads)
        self.a_b = nn.Parameter(torch.empty(num_heads, 3 * d_e))
        self.a = nn.Parameter(torch.empty(num_heads, 2 * d_a + d_e * num_heads))
        self.f = nn.Parameter(torch.empty(num_heads, 2 * d_a + d_e * num_heads))
        self.f_a_b = nn.Parameter(torch.empty(num_heads, 3 * d_e))
        for t in (self.projection_b.weight, self.a_b, self.a, self.f, self.f_a_b):
            nn.init.xavier_uniform_(t.data, gain=1.414)
        self.bond_mask = bond_mask
        self.frag_bond_mask = frag_bond_mask
        self.atom_mask_individual = atom_mask_individual

    lite = False

    def forward(self, x_atoms, edge_index, edge_attr, frag_index, x_frags, atom_to_frag_ids,
                bond_nodes, bond_graph_index, bond_graph_attr, fbond_nodes, fbond_graph_index, fbond_graph_attr):
        H = self.num_heads
        # L1: bond graph (row 0 of its index is the destination) -- gat2.py:137-169
        dst, src = bond_graph_index
        h_b = self.projection_b(bond_nodes).view(bond_nodes.size(0), H, -1)
        out_b, _, attn_b = gat_level_materialised(h_b, self.edge_attr_bond_embed(bond_graph_attr), self.a_b, dst, src, H)
        new_bond = out_b.view(bond_nodes.size(0), -1)
        if self.bond_mask is not None:                       # gat2.py:173-176
            with torch.no_grad():
                new_bond[self.bond_mask:self.bond_mask + 2, :] = 0.0

        # L2: atom graph with self loops; edge attribute = L1 output, zeros on the loops -- :179-224
        looped, _ = add_self_loops(edge_index)
        loop_attr = torch.zeros(x_atoms.size(0), self.edge_out, dtype=torch.long).to(edge_attr)
        attr_a = torch.cat((new_bond, loop_attr), dim=0)
        src, dst = looped
        h_a = self.projection_a(x_atoms)
        n_a = h_a.size(0)
        out_a, _, attn_a = gat_level_materialised(h_a.view(n_a, H, -1), attr_a, self.a, dst, src, H)
        atoms_new = out_a.view(n_a, -1)
        if self.atom_mask_individual is not None:            # :227-231
            with torch.no_grad():
                atoms_new[self.atom_mask_individual, :] = 0.0

        # L3: atom -> fragment sum (the incoming x_frags is overwritten) -- :234
        frags = scatter_add(atoms_new, atom_to_frag_ids, dim=0)
        if self.lite:            # model_version gat2_lite stops here (gat2_lite.py:65-150): no fragment(-bond) graphs
            if self.return_attentions:
                return atoms_new, frags, new_bond, None, attn_a, None, attn_b, None
            return atoms_new, frags, new_bond, None

        # L4a: fragment-bond graph -- :239-272
        dst, src = fbond_graph_index
        h_fb = self.projection_fb(fbond_nodes).view(fbond_nodes.size(0), H, -1)
        out_fb, _, attn_fb = gat_level_materialised(h_fb, self.edge_attr_fbond_embed(fbond_graph_attr), self.f_a_b, dst, src, H)
        new_fbond = out_fb.view(fbond_nodes.size(0), -1)
        if self.frag_bond_mask is not None:                  # :275-278
            with torch.no_grad():
                new_fbond[2 * self.frag_bond_mask, :] = 0.0
                new_fbond[2 * self.frag_bond_mask + 1, :] = 0.0

        # L4b: fragment graph on the un-projected fragment sums -- :283-316
        src, dst = frag_index
        n_f = frags.size(0)
        out_f, _, attn_f = gat_level_materialised(frags.view(n_f, H, -1), new_fbond, self.f, dst, src, H)
        frags_new = out_f.view(n_f, -1)

        if self.return_attentions:
            return atoms_new, frags_new, new_bond, new_fbond, attn_a, attn_f, attn_b, attn_fb
        return atoms_new, frags_new, new_bond, new_fbond


class FragNetLayerEdge(nn.Module):
    """model_version gat2_edge: fragnet/model/gat/gat2_edge.py:13-177.  Constructor order as there (RNG / state dict)."""

    def __init__(self, atom_in=128, atom_out=128, frag_in=128, frag_out=128, edge_in=128, edge_out=128, num_heads=2,
                 bond_edge_in=1, return_attentions=False, add_frag_self_loops=False):
        super().__init__()
        self.add_frag_self_loops = add_frag_self_loops
        self.return_attentions = return_attentions
        self.edge_out = edge_out
        self.atom_embed = nn.Linear(atom_in, atom_out)           # :22-35 never used by forward
        self.frag_embed = nn.Linear(frag_in, frag_out)
        self.edge_embed = nn.Linear(edge_in, edge_out)
        self.bond_edge_embed = nn.Linear(edge_in, edge_out)
        self.frag_message_mlp = nn.Linear(2 * atom_out, atom_out)
        self.atom_mlp = _mlp2(atom_out)
        self.frag_mlp = _mlp2(atom_out)
        self.bias = nn.Parameter(torch.zeros(atom_out))
        self.leakyrelu = nn.LeakyReLU(0.2)
        self.num_heads = num_heads
        self.edge_attr_bond_embed2 = nn.Linear(edge_out, edge_out)
        d_e = edge_out // num_heads
        self.projection_b = nn.Linear(edge_in, d_e * num_heads)
        self.edge_attr_bond_embed = nn.Linear(bond_edge_in, d_e)
        self.cnx_attr_transform = nn.Linear(8, self.edge_out)    # :46
        d_a = atom_out // num_heads
        self.projection_a = nn.Linear(atom_in, d_a * num_heads)
        self.a_b = nn.Parameter(torch.empty(num_heads, 3 * d_e))
        self.a = nn.Parameter(torch.empty(num_heads, 2 * d_a + d_e * num_heads))
        self.f = nn.Parameter(torch.empty(num_heads, 2 * d_a + d_e * num_heads))
        for t in (self.projection_b.weight, self.a_b, self.a, self.f):      # :54-58
            nn.init.xavier_uniform_(t.data, gain=1.414)

    def forward(self, x_atoms, edge_index, edge_attr, frag_index, x_frags, atom_to_frag_ids, bond_nodes,
                bond_graph_index, bond_graph_attr, cnx_attr):
        H = self.num_heads
        dst, src = bond_graph_index                                              # L1 :74-103
        h_b = self.projection_b(bond_nodes).view(bond_nodes.size(0), H, -1)
        out_b, _, attn_b = gat_level_materialised(h_b, self.edge_attr_bond_embed(bond_graph_attr), self.a_b, dst, src, H)
        new_bond = out_b.view(bond_nodes.size(0), -1)
        looped, _ = add_self_loops(edge_index)                                   # L2 :108-141
        loop_attr = torch.zeros(x_atoms.size(0), self.edge_out, dtype=torch.long).to(edge_attr)
        attr_a = torch.cat((new_bond, loop_attr), dim=0)
        src, dst = looped
        h_a = self.projection_a(x_atoms)
        n_a = h_a.size(0)
        out_a, _, attn_a = gat_level_materialised(h_a.view(n_a, H, -1), attr_a, self.a, dst, src, H)
        atoms_new = out_a.view(n_a, -1)
        frags = scatter_add(atoms_new, atom_to_frag_ids, dim=0)                   # L3 :142
        if self.add_frag_self_loops:
            frag_index, _ = add_self_loops(frag_index)
        src, dst = frag_index                                                    # L4 :148-172
        n_f = frags.size(0)
        out_f, _, attn_f = gat_level_materialised(frags.view(n_f, H, -1), self.cnx_attr_transform(cnx_attr), self.f, dst, src, H)
        frags_new = out_f.view(n_f, -1)
        if self.return_attentions:
            return atoms_new, frags_new, new_bond, attn_a, attn_f, attn_b
        return atoms_new, frags_new, new_bond


class FragNet(nn.Module):
    def __init__(self, num_layer, drop_ratio=0.2, emb_dim=128, atom_features=167, frag_features=167,
                 edge_features=17, fedge_in=6, fbond_edge_in=6, num_heads=4, variant="gat2"):
        super().__init__()
        if variant not in ("gat2", "gat2_lite", "gat2_edge"):
            raise ValueError(variant)
        self.variant = variant           # "gat2_lite": fragnet/model/gat/gat2_lite.py (same parameters, levels L1-L3)
        self.num_layer = num_layer
        self.dropout = nn.Dropout(p=drop_ratio)
        self.act = nn.ReLU()
        self.layers = nn.ModuleList()
        if variant == "gat2_edge":       # gat2_edge.py:190-196
            self.layers.append(FragNetLayerEdge(atom_in=atom_features, atom_out=emb_dim, frag_in=frag_features, frag_out=emb_dim,
                                                edge_in=edge_features, edge_out=emb_dim, num_heads=num_heads))
            for _ in range(num_layer - 1):
                self.layers.append(FragNetLayerEdge(atom_in=emb_dim, atom_out=emb_dim, frag_in=emb_dim, frag_out=emb_dim,
                                                    edge_in=emb_dim, edge_out=emb_dim, num_heads=num_heads))
            return
        self.layers.append(FragNetLayerA(atom_in=atom_features, atom_out=emb_dim, frag_in=frag_features,
                                         frag_out=emb_dim, edge_in=edge_features, fedge_in=fedge_in,
                                         fbond_edge_in=fbond_edge_in, edge_out=emb_dim, num_heads=num_heads))
        for _ in range(num_layer - 1):
            self.layers.append(FragNetLayerA(atom_in=emb_dim, atom_out=emb_dim, frag_in=emb_dim, frag_out=emb_dim,
                                             edge_in=emb_dim, edge_out=emb_dim, fedge_in=emb_dim,
                                             fbond_edge_in=fbond_edge_in, num_heads=num_heads))

    def forward(self, batch, trace=None):
        drop_act = lambda t: self.act(self.dropout(t))
        if self.variant == "gat2_edge":                          # gat2_edge.py:198-236
            x_atoms = self.dropout(batch["x_atoms"])
            x_frags = self.dropout(batch["x_frags"])
            e_attr = bond_nodes = None
            for i, layer in enumerate(self.layers):
                x_atoms, x_frags, bond_nodes = layer(
                    x_atoms, batch["edge_index"], batch["edge_attr"] if i == 0 else e_attr, batch["frag_index"], x_frags,
                    batch["atom_to_frag_ids"], batch["node_features_bonds"] if i == 0 else bond_nodes,
                    batch["edge_index_bonds_graph"], batch["edge_attr_bonds"], batch["cnx_attr"])
                if trace is not None:
                    trace.append((x_atoms, x_frags, bond_nodes, None))
                x_atoms, x_frags = drop_act(x_atoms), drop_act(x_frags)
                bond_nodes = drop_act(bond_nodes)
                e_attr = bond_nodes
            return x_atoms, x_frags, bond_nodes, None
        lite = self.variant == "gat2_lite"
        for layer in self.layers:
            layer.lite = lite
        x_atoms = self.dropout(batch["x_atoms"])
        x_frags = self.dropout(batch["x_frags"])
        e_attr = batch["edge_attr"]
        bond_nodes = batch["node_features_bonds"]
        fbond_nodes = batch["node_features_fbonds"]
        for layer in self.layers:
            x_atoms, x_frags, bond_nodes, fbond_nodes = layer(
                x_atoms, batch["edge_index"], e_attr, batch["frag_index"], x_frags, batch["atom_to_frag_ids"],
                bond_nodes, batch["edge_index_bonds_graph"], batch["edge_attr_bonds"],
                fbond_nodes, batch["edge_index_fbonds"], batch["edge_attr_fbonds"])
            if trace is not None:
                trace.append((x_atoms, x_frags, bond_nodes, fbond_nodes))
            x_atoms, x_frags = drop_act(x_atoms), drop_act(x_frags)
            bond_nodes = drop_act(bond_nodes)
            fbond_nodes = None if lite else drop_act(fbond_nodes)     # gat2_lite.py:197-214
            e_attr = bond_nodes          # layers 1.. receive the bond features as edge_attr too (gat2.py:424)
        return x_atoms, x_frags, bond_nodes, fbond_nodes


class _PredictorHead(nn.Sequential):
    """FTHead3: Linear stack with act(dropout(linear(x))) between layers -- gat2.py:678-725."""

    def __init__(self, input_dim=128, h1=128, h2=1024, h3=1024, h4=512, drop_ratio=0.2, n_classes=1, act="relu"):
        super().__init__()
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = _ACTS[act]()
        self.hidden_dims = [h1, h2, h3, h4]
        dims = [input_dim * 2] + self.hidden_dims + [n_classes]
        self.predictor = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])

    def forward(self, enc):
        for lin in self.predictor[:-1]:
            enc = self.activation(self.dropout(lin(enc)))
        return self.predictor[-1](enc)


FTHead3 = _PredictorHead


class FTHead4(nn.Module):
    """dropout -> dense -> act -> dropout -> out_proj -- gat2.py:640-675."""

    def __init__(self, input_dim=128, h1=128, act="relu", n_classes=1, drop_ratio=0.2):
        super().__init__()
        self.activation = _ACTS[act]()
        self.dense = nn.Linear(input_dim * 2, h1)
        self.dropout = nn.Dropout(p=drop_ratio)
        self.out_proj = nn.Linear(h1, n_classes)

    def forward(self, x):
        return self.out_proj(self.dropout(self.activation(self.dense(self.dropout(x)))))


class FTHead1(nn.Sequential):
    """dropout -> lin1 -> relu -> dropout -> out, dropout rate fixed by the constructor default -- gat2.py:569-587."""

    def __init__(self, emb_dim=128, h1=128, drop_ratio=0.2, n_classes=1):
        super().__init__()
        self.lin1 = nn.Linear(emb_dim * 2, h1)
        self.out = nn.Linear(h1, n_classes)
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = nn.ReLU()

    def forward(self, enc):
        return self.out(self.dropout(self.activation(self.lin1(self.dropout(enc)))))


class FTHead2(nn.Sequential):
    """relu(dropout(linear)) stack 256 -> 1024 -> 1024 -> 512 -> n_classes with p = 0.1; ``lin1`` / ``out`` are constructed
    (RNG order, state dict) but never used -- gat2.py:727-751."""

    def __init__(self, input_dim=128, h1=128, drop_ratio=0.2, n_classes=1):
        super().__init__()
        self.lin1 = nn.Linear(input_dim * 2, h1)
        self.out = nn.Linear(h1, n_classes)
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = nn.ReLU()
        dims = [input_dim * 2, 1024, 1024, 512, n_classes]
        self.predictor = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.dropout = nn.Dropout(p=0.1)

    def forward(self, enc):
        for lin in self.predictor[:-1]:
            enc = torch.relu(self.dropout(lin(enc)))
        return self.predictor[-1](enc)


class FTHead5(nn.Sequential):
    """act(dropout(linear)) twice, then a plain Linear (``h4`` is accepted and ignored) -- gat2.py:590-637."""

    def __init__(self, input_dim=128, h1=128, h2=1024, h4=512, drop_ratio=0.2, n_classes=1, act="relu"):
        super().__init__()
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = _ACTS[act]()
        dims = [input_dim * 2, h1, h2, n_classes]
        self.predictor = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])

    def forward(self, enc):
        for lin in self.predictor[:-1]:
            enc = self.activation(self.dropout(lin(enc)))
        return self.predictor[-1](enc)


def pool_cat(x_atoms, x_frags, batch):
    """cat(sum-pool atoms by molecule, sum-pool fragments by molecule) -- gat2.py:820-823."""
    frags = scatter_add(x_frags, batch["frag_batch"], dim=0)
    atoms = scatter_add(x_atoms, batch["batch"], dim=0)
    return torch.cat((atoms, frags), 1)


class InjectedDropout(nn.Module):
    """Test hook: a drop-in for ``nn.Dropout`` that multiplies its input by SUPPLIED scaled keep masks (mask / (1 - p)) in
    call order instead of drawing them, so that this restatement can be run with the masks another implementation drew
    (tests/test_gpu_dropout_parity.py regenerates the HIP path's Philox masks).  ``None`` in the list = identity (a dropout whose
    result the reference never reads, e.g. gat2.py:397 on x_frags)."""

    def __init__(self, masks, p: float = 0.0):
        super().__init__()
        self.masks, self.cursor, self.p = list(masks), 0, p

    def forward(self, x):
        if self.cursor >= len(self.masks):
            raise IndexError(f"InjectedDropout: call {self.cursor} has no mask ({len(self.masks)} supplied)")
        m = self.masks[self.cursor]
        self.cursor += 1
        if m is None:
            return x
        if m.numel() != x.numel():
            raise ValueError(f"InjectedDropout: call {self.cursor - 1} got a mask of {m.numel()} elements for an input of {x.numel()}")
        return x * m.reshape(x.shape)


def inject_dropout(model: nn.Module, masks) -> InjectedDropout:
    """Replaces EVERY ``nn.Dropout`` of ``model`` by one shared InjectedDropout, so the masks are consumed in the order the
    forward pass calls dropout (gat2.py:396-397, then per layer :414-418, then the head :721-722 / :668-675)."""
    inj = InjectedDropout(masks)
    for mod in model.modules():
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.Dropout):
                setattr(mod, name, inj)
    return inj


class FragNetFineTune(nn.Module):
    def __init__(self, n_classes=1, atom_features=167, frag_features=167, edge_features=17, num_layer=4,
                 num_heads=4, drop_ratio=0.15, h1=256, h2=256, h3=256, h4=256, act="celu", emb_dim=128,
                 fthead="FTHead3", variant="gat2"):
        super().__init__()
        self.pretrain = FragNet(num_layer=num_layer, drop_ratio=drop_ratio, num_heads=num_heads, emb_dim=emb_dim,
                                atom_features=atom_features, frag_features=frag_features, edge_features=edge_features,
                                variant=variant)
        if fthead == "FTHead3":
            self.fthead = FTHead3(n_classes=n_classes, input_dim=emb_dim, h1=h1, h2=h2, h3=h3, h4=h4,
                                  drop_ratio=drop_ratio, act=act)
        elif fthead == "FTHead4":
            self.fthead = FTHead4(n_classes=n_classes, h1=h1, drop_ratio=drop_ratio, act=act)
        elif fthead == "FTHead1":                      # gat2.py:795-799: constructor defaults except n_classes
            self.fthead = FTHead1(n_classes=n_classes)
        elif fthead == "FTHead2":
            self.fthead = FTHead2(n_classes=n_classes)
        else:
            raise ValueError(f"FragNetFineTune selects FTHead1-4 (gat2.py:795-814), got {fthead}")

    def forward(self, batch):
        x_atoms, x_frags, _, _ = self.pretrain(batch)
        return self.fthead(pool_cat(x_atoms, x_frags, batch))


class PretrainTask(nn.Module):
    def __init__(self, dim_in=128, dim_out=1, L=2):
        super().__init__()
        halving = lambda w: nn.ModuleList(
            [nn.Linear(w // 2 ** l, w // 2 ** (l + 1)) for l in range(L)] + [nn.Linear(w // 2 ** L, dim_out)])
        self.bl_reduce_layer = nn.Linear(dim_in * 3, dim_in)
        self.bl_layers = halving(dim_in)
        self.ba_layers = halving(dim_in)
        self.da_layers = halving(dim_in)
        self.FC_layers = halving(dim_in * 2)
        self.L = L
        self.activation = nn.ReLU()

    def _tower(self, layers, x):
        for lin in layers[:-1]:
            x = self.activation(lin(x))
        return layers[-1](x)

    def forward(self, x_atoms, x_frags, edge_attr, batch):
        ends = x_atoms[batch["edge_index"].T]                       # [E, 2, D] -- pretrain_heads.py:67-70
        bl = self.bl_reduce_layer(torch.cat((ends[:, 0, :], ends[:, 1, :], edge_attr), dim=1))
        for lin in self.bl_layers:                                  # activation BEFORE each layer -- :72-74
            bl = lin(self.activation(bl))
        ba = self._tower(self.ba_layers, x_atoms)
        da = self._tower(self.da_layers, edge_attr)
        graph_rep = self._tower(self.FC_layers, pool_cat(x_atoms, x_frags, batch))
        return bl, ba, da, graph_rep


class FragNetPreTrain(nn.Module):
    def __init__(self, num_layer=4, drop_ratio=0.15, num_heads=4, emb_dim=128, atom_features=167,
                 frag_features=167, edge_features=16, fedge_in=6, fbond_edge_in=6):
        super().__init__()
        self.pretrain = FragNet(num_layer=num_layer, drop_ratio=drop_ratio, num_heads=num_heads, emb_dim=emb_dim,
                                atom_features=atom_features, frag_features=frag_features,
                                edge_features=edge_features, fedge_in=fedge_in, fbond_edge_in=fbond_edge_in)
        self.head = PretrainTask(128, 1)

    def forward(self, batch):
        x_atoms, x_frags, e_edge, _ = self.pretrain(batch)
        return self.head(x_atoms, x_frags, e_edge, batch)


# ------------------------------------------------------------------ trainer-step semantics
def finetune_regr_loss(out, y):
    """MSELoss(model(batch).view(-1), y) -- fragnet/train/utils.py:337-341."""
    return torch.nn.functional.mse_loss(out.view(-1), y)


def finetune_bce_loss(out, y):
    """BCE-with-logits masked by y > -0.5, divided by the valid count -- train/utils.py:297-304,420-430."""
    valid = y > -0.5
    mat = torch.nn.functional.binary_cross_entropy_with_logits(out, y.view(out.shape), reduction="none")
    return torch.where(valid, mat, torch.zeros_like(mat)).sum() / valid.sum()


def pretrain_loss(outputs, batch):
    """2*MSE(dihedral) + MSE(angle) + MSE(energy): the bond-length term is overwritten before use
    -- fragnet/train/pretrain/pretrain_utils.py:21-26."""
    _, ba, da, graph_rep = outputs
    mse = torch.nn.functional.mse_loss
    l_dh = mse(da, batch["dh_angl"])
    return l_dh + mse(ba, batch["bnd_angl"]) + l_dh + mse(graph_rep.view(-1), batch["y"])
