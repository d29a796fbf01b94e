"""FragNet's four-level message-passing stack on the MI355X kernels.

Module API, constructor signatures, parameter names and construction order (= RNG order = state-dict
order) are the reference's, so the same seed gives the same weights and reference checkpoints load:
  FragNetLayerA    fragnet/model/gat/gat2.py:40-330
  FragNet          fragnet/model/gat/gat2.py:333-442
  FTHead1..5       fragnet/model/gat/gat2.py:569-751
  FragNetFineTune  fragnet/model/gat/gat2.py:758-826
  PretrainTask     fragnet/model/gat/pretrain_heads.py:8-102
  FragNetPreTrain  fragnet/model/gat/pretrain_heads.py:105-141 (twin: gat2_pretrain.py:7-27)

``forward`` is not the reference's op list.  Each attention level is the decomposed form of SURVEY.md
§3.3 -- two scalars per node and head, one scalar per edge and head, a destination-sorted segmented
softmax-aggregate -- run by libfragnet_hip.so (fragnet_amd/ops.py); no [E, H, 3d] message tensor exists.
Dense projections are the fp32-MFMA kernels of the library (ops.linear128; inside the engine they ride in the attention launches),
ReLU heads the hand-written dense kernels (ops.mlp_head); only heads with other activations fall to torch.nn.functional.linear.
The parameters the reference constructs but never reads are constructed too and stay
without gradient, exactly as there.  GPU tensors only: there is no CPU fallback.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine, ops
from ._lib import FN_D
from .plan import LIVE_MOLS_KEY, plan_for

# dev switch for A/B measurements: the finetune models ask the encoder for its bond / fragment-bond outputs although they read neither
_KEEP_EDGE_OUTPUTS = os.environ.get("FRAGNET_KEEP_EDGE_OUTPUTS", "0") == "1"

_ACTS = {
    "relu": nn.ReLU, "silu": nn.SiLU, "gelu": nn.GELU, "celu": nn.CELU, "selu": nn.SELU,
    "rrelu": nn.RReLU, "relu6": nn.ReLU6, "prelu": nn.PReLU, "leakyrelu": nn.LeakyReLU,
}


_LAYER_PLANS = {}      # tiny cache for direct layer calls: index-tensor identity -> GraphPlan


def _plan_from_indices(N, E, F_, EF, edge_index, frag_index, a2f, eib, eifb):
    from .plan import GraphPlan
    key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in (edge_index, frag_index, a2f, eib, eifb))
    plan = _LAYER_PLANS.get(key)
    if plan is None:
        if len(_LAYER_PLANS) >= 4:
            _LAYER_PLANS.clear()
        plan = GraphPlan([
            dict(kind="gat", name="bond", dst=eib[0], src=eib[1], n=E, n_loops=0),
            dict(kind="gat", name="atom", dst=edge_index[1], src=edge_index[0], n=N, n_loops=N),
            dict(kind="gat", name="fbond", dst=eifb[0], src=eifb[1], n=EF, n_loops=0),
            dict(kind="gat", name="frag", dst=frag_index[1], src=frag_index[0], n=F_, n_loops=0),
            dict(kind="seg", name="a2f", key=a2f, n_seg=F_),
        ], edge_index.device)
        _LAYER_PLANS[key] = plan
    return plan


def _two_layer(width: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(width, 2 * width), nn.ReLU(), nn.Linear(2 * width, width))


def _project(x, lin: nn.Linear):
    """``lin(x)`` for the 128-wide node projections (gat2.py:138,186,241): the hand-written fp32-MFMA kernel with its own
    backward (ops.linear128) wherever it applies -- GPU rows, 128 outputs, K <= 168, an input gradient only for K == 128 --
    so that the per-level path (return_attentions, masks, per-op tests) issues no library GEMM either."""
    w = lin.weight
    if x.is_cuda and lin.bias is not None and w.shape[0] == FN_D and w.shape[1] <= 168 and (w.shape[1] == FN_D or not x.requires_grad):
        return ops.linear128(x, w, lin.bias)
    return F.linear(x, w, lin.bias)


class FragNetLayerA(nn.Module):
    def __init__(self, atom_in=128, atom_out=128, frag_in=128, frag_out=128, edge_in=128, edge_out=128,
                 fedge_in=128, num_heads=2, bond_edge_in=1, fbond_edge_in=8, return_attentions=False,
                 add_frag_self_loops=False, bond_mask=None, frag_bond_mask=None, atom_mask_individual=None):
        super().__init__()
        if atom_out != 128 or edge_out != 128:
            raise ValueError("the gfx950 kernels are specialised for emb_dim = 128 (every reference config)")
        if num_heads not in (1, 2, 4, 8):
            raise ValueError("num_heads must be 1, 2, 4 or 8")
        self.add_frag_self_loops = add_frag_self_loops
        self.return_attentions = return_attentions
        self.edge_out = edge_out
        self.num_heads = num_heads
        # constructed-but-unused block (kept for RNG order and checkpoint compatibility)
        self.atom_embed = nn.Linear(atom_in, atom_out)
        self.frag_embed = nn.Linear(frag_in, frag_out)
        self.edge_embed = nn.Linear(edge_in, edge_out)
        self.bond_edge_embed = nn.Linear(edge_in, edge_out)
        self.frag_message_mlp = nn.Linear(atom_out * 2, atom_out)
        self.atom_mlp = _two_layer(atom_out)
        self.frag_mlp = _two_layer(atom_out)
        self.bias = nn.Parameter(torch.zeros(atom_out))      # reference: uninitialised, never read
        self.leakyrelu = nn.LeakyReLU(0.2)
        self.edge_attr_bond_embed2 = nn.Linear(edge_out, edge_out)
        # live block
        d = edge_out // num_heads
        self.projection_b = nn.Linear(edge_in, d * num_heads)
        self.projection_fb = nn.Linear(fedge_in, d * num_heads)
        self.edge_attr_bond_embed = nn.Linear(bond_edge_in, d)
        self.edge_attr_fbond_embed = nn.Linear(fbond_edge_in, d)
        self.projection_a = nn.Linear(atom_in, (atom_out // num_heads) * num_heads)
        self.a_b = nn.Parameter(torch.empty(num_heads, 3 * d))
        self.a = nn.Parameter(torch.empty(num_heads, 2 * d + edge_out))
        self.f = nn.Parameter(torch.empty(num_heads, 2 * d + edge_out))
        self.f_a_b = nn.Parameter(torch.empty(num_heads, 3 * d))
        for w in (self.projection_b.weight, self.a_b, self.a, self.f, self.f_a_b):
            nn.init.xavier_uniform_(w.data, gain=1.414)
        self.bond_mask = bond_mask
        self.frag_bond_mask = frag_bond_mask
        self.atom_mask_individual = atom_mask_individual

    lite = False      # set by FragNet(variant="gat2_lite")

    def forward(self, x_atoms, edge_index, edge_attr, frag_index, x_frags, atom_to_frag_ids,
                node_feautures_bond_graph, edge_index_bonds_graph, edge_attr_bond_graph,
                node_feautures_fbond_graph, edge_index_fbond_graph, edge_attr_fbond_graph):
        """The reference's 12-argument layer signature (gat2.py:121-135).  ``edge_attr`` and ``x_frags`` are
        accepted and ignored, as they are overwritten before use there (gat2.py:183, 234)."""
        plan = _plan_from_indices(x_atoms.shape[0], node_feautures_bond_graph.shape[0], x_frags.shape[0],
                                  node_feautures_fbond_graph.shape[0], edge_index, frag_index, atom_to_frag_ids,
                                  edge_index_bonds_graph, edge_index_fbond_graph)
        return self.run(x_atoms, node_feautures_bond_graph, node_feautures_fbond_graph, edge_attr_bond_graph,
                        edge_attr_fbond_graph, plan)

    def run(self, x_atoms, bond_nodes, fbond_nodes, bond_cos, fbond_attr, plan):
        """x_atoms [N, atom_in]; bond_nodes [E, edge_in]; fbond_nodes [EF, fedge_in];
        bond_cos [Eb, 1]; fbond_attr [EFB, fbond_edge_in]; plan: GraphPlan of the batch."""
        H, d = self.num_heads, self.edge_out // self.num_heads
        want = self.return_attentions
        L = plan.levels

        # L1 bond graph (gat2.py:137-169): affine-in-cos edge term folded in-kernel
        r = ops.gat_level(_project(bond_nodes, self.projection_b), self.a_b, L["bond"], H,
                          x_sorted=plan.sorted_attr("bond", bond_cos), embW=self.edge_attr_bond_embed.weight,
                          embb=self.edge_attr_bond_embed.bias,
                          want_probs=want)
        new_bond, p_bond = (r[0], r[2]) if want else (r, None)
        if self.bond_mask is not None:
            with torch.no_grad():
                new_bond[self.bond_mask:self.bond_mask + 2, :] = 0.0

        # L2 atom graph with self loops (gat2.py:179-224): edge term = <new_bond[e], a[:, d:d+128]>, 0 on loops
        s_edge = ops.row_dots_sorted(new_bond, self.a, d, L["atom"])
        r = ops.gat_level(_project(x_atoms, self.projection_a), self.a, L["atom"], H,
                          s_sorted=s_edge, want_probs=want)
        atoms_new, p_atom = (r[0], r[2]) if want else (r, None)
        if self.atom_mask_individual is not None:
            with torch.no_grad():
                atoms_new[self.atom_mask_individual, :] = 0.0

        # L3 atom -> fragment sum (gat2.py:234)
        frags = ops.segment_sum(atoms_new, plan.segs["a2f"], plan)
        if self.lite:            # model_version gat2_lite (gat2_lite.py:65-150): levels L1-L3 only
            if want:
                return (atoms_new, frags, new_bond, None, ops.attn_by_src(p_atom, L["atom"], H), None,
                        ops.attn_by_src(p_bond, L["bond"], H), None)
            return atoms_new, frags, new_bond, None

        # L4a fragment-bond graph (gat2.py:239-272)
        r = ops.gat_level(_project(fbond_nodes, self.projection_fb), self.f_a_b,
                          L["fbond"], H, x_sorted=plan.sorted_attr("fbond", fbond_attr), embW=self.edge_attr_fbond_embed.weight,
                          embb=self.edge_attr_fbond_embed.bias, want_probs=want)
        new_fbond, p_fbond = (r[0], r[2]) if want else (r, None)
        if self.frag_bond_mask is not None:
            with torch.no_grad():
                new_fbond[2 * self.frag_bond_mask, :] = 0.0
                new_fbond[2 * self.frag_bond_mask + 1, :] = 0.0

        # L4b fragment graph on the raw fragment sums (gat2.py:283-316)
        s_edge_f = ops.row_dots_sorted(new_fbond, self.f, d, L["frag"])
        r = ops.gat_level(frags, self.f, L["frag"], H, s_sorted=s_edge_f, want_probs=want)
        frags_new, p_frag = (r[0], r[2]) if want else (r, None)

        if want:
            return (atoms_new, frags_new, new_bond, new_fbond,
                    ops.attn_by_src(p_atom, L["atom"], H), ops.attn_by_src(p_frag, L["frag"], H),
                    ops.attn_by_src(p_bond, L["bond"], H), ops.attn_by_src(p_fbond, L["fbond"], H))
        return atoms_new, frags_new, new_bond, new_fbond


class FragNetLayerEdge(nn.Module):
    """model_version gat2_edge (fragnet/model/gat/gat2_edge.py:13-177): bond graph, atom graph and atom -> fragment sum as
    in gat2; no fragment-bond graph -- the fragment graph's edge term is <Linear(8 -> 128)(cnx_attr), f[:, d:d+128]>,
    folded in-kernel like the bond graph's cos term (fn_edge_term mode 2 with K = 8, d_e = 128).  Same constructor
    order and state-dict keys as the reference."""

    def __init__(self, atom_in=128, atom_out=128, frag_in=128, frag_out=128, edge_in=128, edge_out=128, num_heads=2,
                 bond_edge_in=1, return_attentions=False, add_frag_self_loops=False):
        super().__init__()
        if atom_out != 128 or edge_out != 128:
            raise ValueError("the gfx950 kernels are specialised for emb_dim = 128 (every reference config)")
        if num_heads not in (1, 2, 4, 8):
            raise ValueError("num_heads must be 1, 2, 4 or 8")
        if add_frag_self_loops:
            # the reference cannot run this option either: gat2_edge.py:144-156 appends the loop edges to frag_index but not to
            # the connection attributes, and its forward dies with a RuntimeError (shape mismatch in torch.cat; recorded in
            # tests/golden/head5_direct.npz).  NotImplementedError is a RuntimeError.
            raise NotImplementedError("gat2_edge with add_frag_self_loops=True fails in the reference too (gat2_edge.py:144-156: "
                                      "no connection attributes for the loop edges); never set by its drivers")
        self.add_frag_self_loops = add_frag_self_loops
        self.return_attentions = return_attentions
        self.edge_out = edge_out
        self.atom_embed = nn.Linear(atom_in, atom_out)           # constructed-but-unused block (gat2_edge.py:22-35)
        self.frag_embed = nn.Linear(frag_in, frag_out)
        self.edge_embed = nn.Linear(edge_in, edge_out)
        self.bond_edge_embed = nn.Linear(edge_in, edge_out)
        self.frag_message_mlp = nn.Linear(atom_out * 2, atom_out)
        self.atom_mlp = _two_layer(atom_out)
        self.frag_mlp = _two_layer(atom_out)
        self.bias = nn.Parameter(torch.zeros(atom_out))
        self.leakyrelu = nn.LeakyReLU(0.2)
        self.num_heads = num_heads
        self.edge_attr_bond_embed2 = nn.Linear(edge_out, edge_out)
        d = edge_out // num_heads
        self.projection_b = nn.Linear(edge_in, d * num_heads)
        self.edge_attr_bond_embed = nn.Linear(bond_edge_in, d)
        self.cnx_attr_transform = nn.Linear(8, edge_out)
        self.projection_a = nn.Linear(atom_in, (atom_out // num_heads) * num_heads)
        self.a_b = nn.Parameter(torch.empty(num_heads, 3 * d))
        self.a = nn.Parameter(torch.empty(num_heads, 2 * d + edge_out))
        self.f = nn.Parameter(torch.empty(num_heads, 2 * d + edge_out))
        for w in (self.projection_b.weight, self.a_b, self.a, self.f):
            nn.init.xavier_uniform_(w.data, gain=1.414)

    def engine_placeholders(self):
        """Zero stand-ins for the fragment-bond parameters fn_layer_weights has slots for (never read for variant 2)."""
        ph = getattr(self, "_engine_ph", None)
        dev = self.f.device
        if ph is None or ph["f_a_b"].device != dev:
            # [128, 168]: covers every width the transposing prologue may read for this slot; never multiplied
            ph = self._engine_ph = {"proj_fb_w": torch.zeros(self.edge_out, 168, device=dev), "proj_fb_b": torch.zeros(self.edge_out, device=dev),
                                    "f_a_b": torch.zeros(self.num_heads, 3 * (self.edge_out // self.num_heads), device=dev)}
        return ph

    def run(self, x_atoms, bond_nodes, bond_cos, cnx_attr, plan):
        H, d = self.num_heads, self.edge_out // self.num_heads
        want = self.return_attentions
        L = plan.levels
        r = ops.gat_level(_project(bond_nodes, self.projection_b), self.a_b, L["bond"], H,
                          x_sorted=plan.sorted_attr("bond", bond_cos), embW=self.edge_attr_bond_embed.weight,
                          embb=self.edge_attr_bond_embed.bias, want_probs=want)                      # L1 :74-103
        new_bond, p_bond = (r[0], r[2]) if want else (r, None)
        s_edge = ops.row_dots_sorted(new_bond, self.a, d, L["atom"])                                  # L2 :108-141
        r = ops.gat_level(_project(x_atoms, self.projection_a), self.a, L["atom"], H,
                          s_sorted=s_edge, want_probs=want)
        atoms_new, p_atom = (r[0], r[2]) if want else (r, None)
        frags = ops.segment_sum(atoms_new, plan.segs["a2f"], plan)                                    # L3 :142
        r = ops.gat_level(frags, self.f, L["frag"], H, x_sorted=plan.sorted_attr("frag", cnx_attr),   # L4 :148-172
                          embW=self.cnx_attr_transform.weight, embb=self.cnx_attr_transform.bias, want_probs=want)
        frags_new, p_frag = (r[0], r[2]) if want else (r, None)
        if want:
            return (atoms_new, frags_new, new_bond, ops.attn_by_src(p_atom, L["atom"], H),
                    ops.attn_by_src(p_frag, L["frag"], H), ops.attn_by_src(p_bond, L["bond"], H))
        return atoms_new, frags_new, new_bond


class FragNet(nn.Module):
    def __init__(self, num_layer, drop_ratio=0.2, emb_dim=128, atom_features=167, frag_features=167,
                 edge_features=17, fedge_in=6, fbond_edge_in=6, num_heads=4, variant="gat2"):
        super().__init__()
        if variant not in ("gat2", "gat2_lite", "gat2_edge"):
            raise ValueError(f"model_version {variant!r}: gat2, gat2_lite and gat2_edge are on the accelerated path")
        self.variant = variant      # gat2_lite = fragnet/model/gat/gat2_lite.py: same parameters, levels L1-L3 only
        self.num_layer = num_layer
        self.dropout = nn.Dropout(p=drop_ratio)
        self.act = nn.ReLU()
        self.layers = nn.ModuleList()
        self.rng = ops.PhiloxStream()
        self.use_engine = True      # False: one autograd node per level (same kernels, used by the per-op tests)
        if variant == "gat2_edge":  # gat2_edge.py:190-196; per-level path only (fn_encoder_* knows gat2 / gat2_lite)
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

    def forward(self, batch, edge_outputs: bool = True):
        """``edge_outputs=False`` (what the finetune models pass: their heads pool atoms and fragments only, gat2.py:816-826): the engine does
        not store the last layer's activated bond / fragment-bond rows; the third and fourth result are then empty tensors."""
        plan = plan_for(batch)
        p, train = self.dropout.p, self.training
        if self.variant == "gat2_edge" and self.use_engine and not any(l.return_attentions for l in self.layers):
            # whole encoder in two C calls (fn_encoder.variant = 2); the placeholder fragment-bond input is never multiplied
            outs = engine.encoder_forward(self.layers, plan, batch["x_atoms"], batch["node_features_bonds"],
                                          batch["node_features_fbonds"], plan.sorted_attr("bond", batch["edge_attr_bonds"], defer=True),
                                          plan.sorted_attr("frag", batch["cnx_attr"], defer=True), self.layers[0].num_heads,
                                          p, train, self.rng, variant=2, edge_outputs=edge_outputs)
            return outs[0], outs[1], outs[2], None
        if self.variant == "gat2_edge":                      # gat2_edge.py:198-236: three tensors travel between layers
            x_atoms = ops.dropout_act(batch["x_atoms"], p, train, False, self.rng)
            bond_nodes, x_frags = batch["node_features_bonds"], None
            for layer in self.layers:
                x_atoms, x_frags, bond_nodes = layer.run(x_atoms, bond_nodes, batch["edge_attr_bonds"], batch["cnx_attr"], plan)[:3]
                x_atoms = ops.dropout_act(x_atoms, p, train, True, self.rng)
                x_frags = ops.dropout_act(x_frags, p, train, True, self.rng)
                bond_nodes = ops.dropout_act(bond_nodes, p, train, True, self.rng)
            return x_atoms, x_frags, bond_nodes, None
        lite = self.variant == "gat2_lite"
        for layer in self.layers:
            layer.lite = lite
        if self.use_engine and not any(l.return_attentions or l.bond_mask is not None or l.frag_bond_mask is not None
                                       or l.atom_mask_individual is not None for l in self.layers):
            # whole encoder in two C calls (fragnet_amd/engine.py); masks / attention outputs use the per-level path
            outs = engine.encoder_forward(self.layers, plan, batch["x_atoms"], batch["node_features_bonds"],
                                          batch["node_features_fbonds"], plan.sorted_attr("bond", batch["edge_attr_bonds"], defer=True),
                                          plan.sorted_attr("fbond", batch["edge_attr_fbonds"], defer=True), self.layers[0].num_heads,
                                          p, train, self.rng, variant=1 if lite else 0, edge_outputs=edge_outputs)
            if outs[4].numel():      # the fused fragment tail also produced the readout: pooled() below hands it out
                # (with the versions of both tensors at this point: an in-place edit of either -- a mask, an attribution hook --
                # between the encoder and pooled() must not be answered with the readout of the unedited rows)
                outs[0]._fragnet_readout = (outs[1], outs[4], outs[0]._version, outs[1]._version)
            return (outs[0], outs[1], outs[2], None) if lite else outs[:4]
        x_atoms = ops.dropout_act(batch["x_atoms"], p, train, False, self.rng)
        # batch["x_frags"] is dead in the reference too: every layer overwrites it with the atom->fragment
        # sum before first use (gat2.py:234); its dropout mask is drawn and discarded there (gat2.py:397).
        bond_nodes, fbond_nodes = batch["node_features_bonds"], batch["node_features_fbonds"]
        x_frags = None
        for layer in self.layers:
            x_atoms, x_frags, bond_nodes, fbond_nodes = layer.run(
                x_atoms, bond_nodes, fbond_nodes, batch["edge_attr_bonds"], batch["edge_attr_fbonds"], plan)[:4]
            x_atoms = ops.dropout_act(x_atoms, p, train, True, self.rng)
            x_frags = ops.dropout_act(x_frags, p, train, True, self.rng)
            bond_nodes = ops.dropout_act(bond_nodes, p, train, True, self.rng)
            if not lite:
                fbond_nodes = ops.dropout_act(fbond_nodes, p, train, True, self.rng)
        return x_atoms, x_frags, bond_nodes, (None if lite else fbond_nodes)


# ------------------------------------------------------------------------------------ heads
class FTHead1(nn.Sequential):
    def __init__(self, emb_dim=128, h1=128, drop_ratio=0.2, n_classes=1):
        super().__init__()
        self.lin1 = nn.Linear(emb_dim * 2, h1)
        self.out = nn.Linear(h1, n_classes)
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = nn.ReLU()

    def forward(self, enc):
        return self.out(self.dropout(self.activation(self.lin1(self.dropout(enc)))))


class _PredictorStack(nn.Sequential):
    """act(dropout(linear(x))) between layers, plain last layer -- gat2.py:631-637, 719-725, 745-751."""

    rng = None       # the encoder's Philox stream when the owning model attaches it (FragNetFineTune does)
    live_rows = None  # set per call by the owning model: input rows >= live_rows are padding, their outputs are 0
    loss_spec = None  # (kind, target, row weights) set per call by a training step that backpropagates the loss with gradient 1 right
                      # away (graphstep.GraphedTrainStep): the last Linear, the loss and that Linear's backward then share a launch
                      # (ops.mlp_head); the returned predictions carry the loss as ``_fragnet_loss``

    def _run(self, enc):
        fused = self.rng is not None and isinstance(self.activation, nn.ReLU) and enc.is_cuda
        if fused and all(l.bias is not None for l in self.predictor) and all(l.out_features % 4 == 0 for l in self.predictor[:-1]):
            if self.loss_spec is not None:
                out, loss = ops.mlp_head(enc, list(self.predictor), self.dropout.p, self.training, self.rng, self.live_rows, loss=self.loss_spec)
                if loss is not None:
                    out._fragnet_loss = (loss, self.loss_spec[1], self.loss_spec[2])
                return out
            return ops.mlp_head(enc, list(self.predictor), self.dropout.p, self.training, self.rng, self.live_rows)     # one autograd node
        for lin in self.predictor[:-1]:
            if fused:    # relu(dropout(.)) as one kernel each way instead of two (same op as between encoder layers)
                enc = ops.dropout_act(lin(enc), self.dropout.p, self.training, True, self.rng)
            else:
                enc = self.activation(self.dropout(lin(enc)))
        return self.predictor[-1](enc)


class FTHead5(_PredictorStack):
    def __init__(self, input_dim=128, h1=128, h2=1024, h4=512, drop_ratio=0.2, n_classes=1, act="relu"):
        super().__init__()
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = _ACTS[act]()
        self.hidden_dims = [h1, h2]
        dims = [input_dim * 2] + self.hidden_dims + [n_classes]
        self.predictor = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])

    def forward(self, enc):
        return self._run(enc)


class FTHead4(nn.Module):
    def __init__(self, input_dim=128, h1=128, act="relu", n_classes=1, drop_ratio=0.2):
        super().__init__()
        self.activation = _ACTS[act]()
        self.dense = nn.Linear(input_dim * 2, h1)
        self.dropout = nn.Dropout(p=drop_ratio)
        self.out_proj = nn.Linear(h1, n_classes)

    def forward(self, x):
        return self.out_proj(self.dropout(self.activation(self.dense(self.dropout(x)))))


class FTHead3(_PredictorStack):
    def __init__(self, input_dim=128, h1=128, h2=1024, h3=1024, h4=512, drop_ratio=0.2, n_classes=1, act="relu"):
        super().__init__()
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = _ACTS[act]()
        self.hidden_dims = [h1, h2, h3, h4]
        dims = [input_dim * 2] + self.hidden_dims + [n_classes]
        self.predictor = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])

    def forward(self, enc):
        return self._run(enc)


class FTHead2(_PredictorStack):
    def __init__(self, input_dim=128, h1=128, drop_ratio=0.2, n_classes=1):
        super().__init__()
        self.lin1 = nn.Linear(input_dim * 2, h1)
        self.out = nn.Linear(h1, n_classes)
        self.dropout = nn.Dropout(p=drop_ratio)
        self.activation = nn.ReLU()
        self.hidden_dims = [1024, 1024, 512]
        dims = [input_dim * 2] + self.hidden_dims + [n_classes]
        self.predictor = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.dropout = nn.Dropout(p=0.1)

    def forward(self, enc):
        return self._run(enc)


def pooled(x_atoms, x_frags, batch):
    """cat(sum of atoms per molecule, sum of fragments per molecule) -- gat2.py:820-823.  For the encoder's own outputs on a
    molecule-contiguous batch the engine has already produced it (inside the fused fragment tail, csrc/mol_tail.inc)."""
    ready = getattr(x_atoms, "_fragnet_readout", None)
    if ready is not None and ready[0] is x_frags and x_atoms._version == ready[2] and x_frags._version == ready[3]:
        return ready[1]
    return ops.pool_cat(x_atoms, x_frags, plan_for(batch))


class FragNetFineTune(nn.Module):
    def __init__(self, n_classes=1, atom_features=167, frag_features=167, edge_features=17, num_layer=4,
                 num_heads=4, drop_ratio=0.15, h1=256, h2=256, h3=256, h4=256, act="celu", emb_dim=128,
                 fthead="FTHead3", variant="gat2"):
        super().__init__()
        self.pretrain = FragNet(num_layer=num_layer, drop_ratio=drop_ratio, num_heads=num_heads, emb_dim=emb_dim,
                                atom_features=atom_features, frag_features=frag_features, edge_features=edge_features,
                                variant=variant)
        if fthead == "FTHead1":
            self.fthead = FTHead1(n_classes=n_classes)
        elif fthead == "FTHead2":
            self.fthead = FTHead2(n_classes=n_classes)
        elif fthead == "FTHead3":
            self.fthead = FTHead3(n_classes=n_classes, input_dim=emb_dim, h1=h1, h2=h2, h3=h3, h4=h4,
                                  drop_ratio=drop_ratio, act=act)
        elif fthead == "FTHead4":
            self.fthead = FTHead4(n_classes=n_classes, h1=h1, drop_ratio=drop_ratio, act=act)

        if isinstance(self.fthead, _PredictorStack):
            self.fthead.rng = self.pretrain.rng

    def forward(self, batch):
        x_atoms, x_frags, _, _ = self.pretrain(batch, edge_outputs=_KEEP_EDGE_OUTPUTS)
        self.fthead.live_rows = batch.get(LIVE_MOLS_KEY)     # static-shape batches: the rows behind it are padding molecules
        return self.fthead(pooled(x_atoms, x_frags, batch))


class FragNetFineTuneEdge(FragNetFineTune):
    """fragnet.model.gat.gat2_edge.FragNetFineTune (finetune_gat2.py:165-166, model_version "gat2_edge")."""

    def __init__(self, *args, **kw):
        kw["variant"] = "gat2_edge"
        super().__init__(*args, **kw)


class FragNetFineTuneLite(FragNetFineTune):
    """fragnet.model.gat.gat2_lite.FragNetFineTune (finetune_gat2.py:141-146, model_version "gat2_lite"): the same
    parameters and state-dict keys as gat2, each layer stops after the atom -> fragment sum."""

    def __init__(self, *args, **kw):
        kw["variant"] = "gat2_lite"
        super().__init__(*args, **kw)


class PretrainTask(nn.Module):
    def __init__(self, dim_in=128, dim_out=1, L=2):
        super().__init__()

        def tower(width):
            return nn.ModuleList([nn.Linear(width // 2 ** l, width // 2 ** (l + 1)) for l in range(L)]
                                 + [nn.Linear(width // 2 ** L, dim_out)])
        self.bl_reduce_layer = nn.Linear(dim_in * 3, dim_in)
        self.bl_layers = tower(dim_in)
        self.ba_layers = tower(dim_in)
        self.da_layers = tower(dim_in)
        self.FC_layers = tower(dim_in * 2)
        self.L = L
        self.activation = nn.ReLU()

    fused_towers = True          # False: the tall towers through ops.mlp_head (library GEMMs + element-wise launches; A/B and tests)
    need_bond_length = True      # False: skip the bond-length tower (graphstep / trainers: the reference's loss never reads it,
                                 # pretrain_utils.py:17-24 overwrites that term) and return None in its place
    _no_rng = ops.PhiloxStream(seed=0)

    def _tower(self, layers, x):
        if x.is_cuda:            # Linear -> ReLU stack as one autograd node (ops.mlp_head with p = 0)
            return ops.mlp_head(x, list(layers), 0.0, self.training, self._no_rng)
        for lin in layers[:-1]:
            x = self.activation(lin(x))
        return layers[-1](x)

    def forward(self, x_atoms, x_frags, edge_attr, batch):
        plan = plan_for(batch, edge_ends=self.need_bond_length)
        bl = None
        if self.need_bond_length:
            # pretrain_heads.py:67-76: reduce -> [lin(act(.))]*: the same stack with the reduce layer in front
            bl = self._tower([self.bl_reduce_layer] + list(self.bl_layers), ops.edge_concat(x_atoms, edge_attr, batch["edge_index"], plan))
        if self.fused_towers and ops.tower_ok(x_atoms, list(self.ba_layers)) and ops.tower_ok(edge_attr, list(self.da_layers)):
            # the two tall towers (every atom, every directed bond) as one launch each way (csrc/tower.hip)
            ba, da = ops.towers([(x_atoms, list(self.ba_layers)), (edge_attr, list(self.da_layers))])
        else:
            ba = self._tower(self.ba_layers, x_atoms)
            da = self._tower(self.da_layers, edge_attr)
        graph_rep = self._tower(self.FC_layers, pooled(x_atoms, x_frags, batch))
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
        plan_for(batch, edge_ends=self.head.need_bond_length)      # build the plan once, with the bond-length head's CSRs
        x_atoms, x_frags, e_edge, _ = self.pretrain(batch)
        return self.head(x_atoms, x_frags, e_edge, batch)
