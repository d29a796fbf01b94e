"""The whole encoder (reference FragNet.forward, gat2.py:381-442) as one autograd node backed by two C calls:
fn_encoder_forward / fn_encoder_backward walk the layers inside libfragnet_hip.so and enqueue every kernel
(MFMA projections, node scalars, segmented softmax-aggregate, segment sum, dropout+ReLU) on the current
stream.  Python does not see the per-level ops, so a training step costs two ctypes calls for the encoder
instead of ~250 autograd nodes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import torch

from . import _lib, ops
from ._lib import Encoder, FN_D, LAYER_FIELDS, LayerWeights, SegPlan
from .plan import GraphPlan, _stream_ptr

# dev switch for A/B measurements: evaluation passes save everything a backward pass would read even when none can follow
_EVAL_SAVES = os.environ.get("FRAGNET_EVAL_SAVES", "0") == "1"


def layer_param_list(layer) -> List[torch.nn.Parameter]:
    """Parameters of one FragNetLayerA in the order of fn_layer_weights."""
    if hasattr(layer, "cnx_attr_transform"):        # gat2_edge layer: no fragment-bond parameters, the cnx_attr Linear sits in emb_fb_*
        ph = layer.engine_placeholders()
        return [layer.projection_b.weight, layer.projection_b.bias, layer.projection_a.weight, layer.projection_a.bias,
                ph["proj_fb_w"], ph["proj_fb_b"], layer.edge_attr_bond_embed.weight, layer.edge_attr_bond_embed.bias,
                layer.cnx_attr_transform.weight, layer.cnx_attr_transform.bias, layer.a_b, layer.a, layer.f, ph["f_a_b"]]
    return [layer.projection_b.weight, layer.projection_b.bias, layer.projection_a.weight, layer.projection_a.bias,
            layer.projection_fb.weight, layer.projection_fb.bias, layer.edge_attr_bond_embed.weight,
            layer.edge_attr_bond_embed.bias, layer.edge_attr_fbond_embed.weight, layer.edge_attr_fbond_embed.bias,
            layer.a_b, layer.a, layer.f, layer.f_a_b]


NP = len(LAYER_FIELDS)
F_IDX = LAYER_FIELDS.index("f")
LITE_DEAD = {LAYER_FIELDS.index(n) for n in ("proj_fb_w", "proj_fb_b", "emb_fb_w", "emb_fb_b", "f", "f_a_b")}
EDGE_DEAD = {LAYER_FIELDS.index(n) for n in ("proj_fb_w", "proj_fb_b", "f_a_b")}          # placeholders of a gat2_edge layer
EDGE_LAST_ONLY = {LAYER_FIELDS.index(n) for n in ("emb_fb_w", "emb_fb_b", "f")}             # read by the last layer's fragment graph only


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.FragnetHipError(f"{name}: the encoder engine needs GPU tensors (got {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _describe(plan: GraphPlan, x_atoms, bond_nodes, fbond_nodes, cos_sorted, fattr_sorted, params: Sequence[torch.Tensor],
              n_layers: int, heads: int, drop_p: float, training: bool, seed: int, offset: int, offset_dev=None,
              variant: int = 0, no_backward: bool = False) -> Encoder:
    e = Encoder()
    e.n_layers, e.heads = n_layers, heads
    e.k_atom0, e.k_bond0, e.k_fbond0 = x_atoms.shape[1], bond_nodes.shape[1], fbond_nodes.shape[1]
    e.k_fattr = fattr_sorted.shape[0]
    e.training, e.drop_p = int(training), float(drop_p)
    e.no_backward = int(bool(no_backward) and not training)       # an evaluation pass nobody differentiates saves nothing for a backward pass
    e.variant = int(variant)
    e.seed, e.offset = seed, offset
    e.offset_dev = None if offset_dev is None else offset_dev.data_ptr()
    L = plan.levels
    e.N, e.E, e.F, e.EF = L["atom"].n, L["bond"].n, L["frag"].n, L["fbond"].n
    e.bond, e.atom, e.fbond, e.frag = L["bond"].c, L["atom"].c, L["fbond"].c, L["frag"].c
    s = plan.segs["a2f"]
    e.a2f = SegPlan(s.rowptr.data_ptr(), s.perm.data_ptr(), s.index.data_ptr(), s.n_seg, s.n_items, s.pos_base, 0)
    e.x_atoms, e.bond_nodes, e.fbond_nodes = x_atoms.data_ptr(), bond_nodes.data_ptr(), fbond_nodes.data_ptr()
    e.cos_sorted, e.fattr_sorted = cos_sorted.data_ptr(), fattr_sorted.data_ptr()
    raw_b, raw_f = plan.pending.get("bond"), plan.pending.get("frag" if variant == 2 else "fbond")      # sorted copies not filled yet: the forward does it
    e.cos_raw = None if raw_b is None else raw_b.data_ptr()
    e.fattr_raw = None if raw_f is None else raw_f.data_ptr()
    for l in range(n_layers):
        for k, name in enumerate(LAYER_FIELDS):
            setattr(e.w[l], name, params[l * NP + k].data_ptr())
    # molecule CSRs: with them fn_encoder_* runs the molecule-resident fused kernels (include/fragnet_hip.h, fn_encoder.n_mols)
    ma, mf = plan.segs.get("mol_atoms"), plan.segs.get("mol_frags")
    if ma is not None and mf is not None and ma.n_seg == mf.n_seg:
        e.mol_atoms = SegPlan(ma.rowptr.data_ptr(), ma.perm.data_ptr(), ma.index.data_ptr(), ma.n_seg, ma.n_items, ma.pos_base, 0)
        e.mol_frags = SegPlan(mf.rowptr.data_ptr(), mf.perm.data_ptr(), mf.index.data_ptr(), mf.n_seg, mf.n_items, mf.pos_base, 0)
        e.n_mols = ma.n_seg
        counts = getattr(plan, "real_mols", None)
        e.counts_dev = None if counts is None else counts.data_ptr()
        e.status = plan._status.data_ptr()
        e.mol_contiguous = int(bool(getattr(plan, "mol_contiguous", False)))       # CollatedBatch: collate_fn's layout
    return e


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_atoms, bond_nodes, fbond_nodes, cos_sorted, fattr_sorted, plan, n_layers, heads, drop_p, training,
                seed, offset, offset_dev, variant, no_backward, edge_outputs, *params):
        x_atoms, bond_nodes, fbond_nodes = _f32(x_atoms, "x_atoms"), _f32(bond_nodes, "node_features_bonds"), _f32(fbond_nodes, "node_features_fbonds")
        ctx.param_objs, ctx.slots = params, [ops.grad_slot(p) for p in params]
        params = tuple(_f32(p, "parameter") for p in params)
        dev = x_atoms.device
        lib = _lib.load()
        e = _describe(plan, x_atoms, bond_nodes, fbond_nodes, cos_sorted, fattr_sorted, params, n_layers, heads, drop_p,
                      training, seed, offset, offset_dev, variant, no_backward)
        ws = torch.empty(lib.fn_encoder_ws_floats(C.byref(e)), dtype=torch.float32, device=dev)
        e.ws, e.ws_floats = ws.data_ptr(), ws.numel()
        # edge_outputs = False: the caller reads neither the bond nor the fragment-bond output (a finetune head): the library gets NULL for
        # them and does not store the last layer's activated rows; two empty, non-differentiable tensors stand in
        outs = [torch.empty((n if (edge_outputs or k < 2) else 0, FN_D), dtype=torch.float32, device=dev) for k, n in enumerate((e.N, e.F, e.E, e.EF))]
        # molecule-contiguous batches: the last layer's fragment tail is one molecule-resident launch that also writes the
        # readout cat(scatter_add(x_atoms, batch), scatter_add(x_frags, frag_batch)) (gat2.py:820-823) -- a fifth output
        pooled = None
        if lib.fn_encoder_fused_tail(C.byref(e)):
            pooled = torch.empty((e.n_mols, 2 * FN_D), dtype=torch.float32, device=dev)
            e.pooled = pooled.data_ptr()
        _lib.check(lib.fn_encoder_forward(C.byref(e), *((o.data_ptr() if (edge_outputs or k < 2) else None) for k, o in enumerate(outs)),
                                          _stream_ptr(dev)), "fn_encoder_forward")
        e.pooled = None
        plan.pending.pop("bond", None)
        plan.pending.pop("frag" if variant == 2 else "fbond", None)
        e.cos_raw = e.fattr_raw = None           # the backward pass reads the sorted copies only
        ctx.desc = e
        ctx.keep = (plan, x_atoms, bond_nodes, fbond_nodes, cos_sorted, fattr_sorted, ws, offset_dev)
        ctx.n_layers = n_layers
        ctx.variant = int(variant)
        ctx.edge_outputs = bool(edge_outputs)
        ctx.save_for_backward(*params, *outs)
        ctx.set_materialize_grads(False)
        dead = [] if edge_outputs else [outs[2], outs[3]]
        if pooled is None:                   # placeholder: the caller pools with ops.pool_cat
            pooled = torch.empty(0, dtype=torch.float32, device=dev)
            dead.append(pooled)
        if dead:
            ctx.mark_non_differentiable(*dead)           # (at most one call per forward)
        return tuple(outs) + (pooled,)

    @staticmethod
    def backward(ctx, g_atoms, g_frags, g_bond, g_fbond, g_pooled):
        n_layers = ctx.n_layers
        saved = ctx.saved_tensors
        params, outs = saved[: n_layers * NP], saved[n_layers * NP:]
        e = ctx.desc
        dev = outs[0].device
        lib = _lib.load()
        gs = [None if g is None else _f32(g, "grad") for g in (g_atoms, g_frags, g_bond, g_fbond)]
        g_pooled = None if g_pooled is None else _f32(g_pooled, "grad")
        e.g_pooled = None if g_pooled is None else g_pooled.data_ptr()       # only a fused-tail forward has a differentiable readout
        grads = [ops.grad_buffer(p, slot) for p, slot in zip(ctx.param_objs, ctx.slots)]      # FlatAdam slots where they exist
        gw = (LayerWeights * n_layers)()
        for l in range(n_layers):
            for k, name in enumerate(LAYER_FIELDS):
                setattr(gw[l], name, grads[l * NP + k].data_ptr())
        scratch = torch.empty(lib.fn_encoder_bwd_ws_floats(C.byref(e)), dtype=torch.float32, device=dev)
        rider = take_adam_rider()                # an optimiser slice that rides in this pass's last launch (arm_adam_rider)
        e.adam_rider = None if rider is None else C.addressof(rider)
        if rider is not None:
            rider.launched = 0
        try:
            _lib.check(lib.fn_encoder_backward(C.byref(e), *((o.data_ptr() if (ctx.edge_outputs or k < 2) else None) for k, o in enumerate(outs)),
                                               *(None if g is None else g.data_ptr() for g in gs), gw, scratch.data_ptr(),
                                               scratch.numel(), _stream_ptr(dev)), "fn_encoder_backward")
        finally:
            e.adam_rider = None
            if rider is not None:
                _ADAM_RIDER[1] = bool(rider.launched)        # the library's word that a launch carried the slice, not ours that we offered it
        e.g_pooled = None
        out = []
        have_frags = gs[1] is not None or g_pooled is not None
        any_grad = have_frags or any(g is not None for g in gs)
        for l in range(n_layers):
            for k in range(NP):
                live = any_grad and not (k == F_IDX and not (l == n_layers - 1 and have_frags))
                if ctx.variant == 1 and k in LITE_DEAD:          # gat2_lite never reads the fragment(-bond) parameters
                    live = False
                if ctx.variant == 2 and (k in EDGE_DEAD or (k in EDGE_LAST_ONLY and not (l == n_layers - 1 and have_frags))):
                    live = False
                out.append(grads[l * NP + k] if live else None)
        return (None,) * 16 + tuple(out)


_ADAM_RIDER = [None, False]       # [armed fn_adam_slice, was it handed to a backward pass]


def arm_adam_rider(slice_struct) -> None:
    """The next encoder backward pass of this process carries ``slice_struct`` (``_lib.AdamSlice``: parameters whose gradients are
    complete before that pass starts) in its last launch.  The caller checks ``adam_rider_taken()`` afterwards and updates
    whatever did not ride itself (graphstep.GraphedTrainStep)."""
    _ADAM_RIDER[0], _ADAM_RIDER[1] = slice_struct, False


def take_adam_rider():
    r = _ADAM_RIDER[0]
    if r is not None:
        _ADAM_RIDER[0], _ADAM_RIDER[1] = None, False
    return r


def adam_rider_taken() -> bool:
    """True if a launch of a backward pass carried the armed slice (fn_adam_slice.launched, set by fn_encoder_backward); disarms
    either way.  The state is per process (one training step at a time), not per thread."""
    taken = _ADAM_RIDER[1]
    _ADAM_RIDER[0], _ADAM_RIDER[1] = None, False
    return taken


def encoder_forward(layers, plan: GraphPlan, x_atoms, bond_nodes, fbond_nodes, cos_sorted, fattr_sorted, heads: int,
                    drop_p: float, training: bool, rng, variant: int = 0, edge_outputs: bool = True) -> tuple:
    """Runs all ``layers`` (FragNetLayerA modules) + the inter-layer act(dropout(.)); returns the four outputs and, fifth,
    the readout [n_mols, 256] when the fused fragment tail produced it (an empty tensor otherwise)."""
    params = [p for layer in layers for p in layer_param_list(layer)]
    n_layers = len(layers)
    # the C side takes widths from the batch and pointers from the parameters: a model built for other feature widths would
    # read and write past its layer-0 weights.  Fail the way the reference's nn.Linear does.
    for name, x, k in (("projection_b", bond_nodes, 0), ("projection_a", x_atoms, 2)) + \
            ((("projection_fb", fbond_nodes, 4),) if variant == 0 else ()):
        w = params[k]
        if w.dim() != 2 or x.dim() != 2 or w.shape[1] != x.shape[1] or w.shape[0] != FN_D:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.shape[0]}x{x.shape[-1]} and {w.shape[-1]}x{w.shape[0]}): "
                               f"layer 0 {name} expects {w.shape[-1]} input features, the batch has {x.shape[-1]}")
    p_eff = float(drop_p) if training else 0.0
    if p_eff > 0.0:
        # reserve the Philox offsets the engine will consume (fn_encoder_rng_blocks): x_atoms + 4 tensors per layer
        N, E = x_atoms.shape[0], bond_nodes.shape[0]
        F, EF = plan.levels["frag"].n, fbond_nodes.shape[0]
        blocks = (N * x_atoms.shape[1] + 3) // 4 + n_layers * sum((n * FN_D + 3) // 4 for n in (N, F, E, EF))
        seed, offset = rng.take(4 * blocks)
    else:
        seed, offset = 0, 0
    # under torch.no_grad(), or when nothing it reads requires a gradient, no backward pass can follow: an evaluation pass then
    # stores nothing for one (fn_encoder.no_backward).  (Inside Function.forward grad mode is always off: it is read here.)
    no_backward = not _EVAL_SAVES and not (torch.is_grad_enabled() and any(t.requires_grad for t in (x_atoms, bond_nodes, fbond_nodes, *params)))
    return _EncoderFn.apply(x_atoms, bond_nodes, fbond_nodes, cos_sorted, fattr_sorted, plan, n_layers, heads, p_eff,
                            bool(training), seed, offset, rng.dev if p_eff > 0.0 else None, int(variant), no_backward, bool(edge_outputs), *params)
