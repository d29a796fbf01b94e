"""Static-shape training step replayed as ONE hipGraph.

An eager step of the hot path is ~200 kernel launches plus the Python around them and is bound by the host
(2.0 ms wall for 1.6 ms of GPU work at ESOL batch 512).  Kernel sizes, however, depend on the batch only through
six counts (atoms, directed bonds, bond-graph edges, fragments, fragment edges, fragment-bond-graph edges), so a
step over buffers of FIXED capacity can be captured once -- graph plan, encoder, pooling, head, loss, backward,
gradient gather -- and replayed with one launch per step.  Every batch (the dict of the reference's
``collate_fn``, dataset/data.py:931-948) is copied into the fixed buffers by one staging kernel
(``fn_stage_padded``) which also fills the tails with PADDING:

* padding rows carry zero features;
* padding index values point at the last ``slack`` slots of the target index space (atoms, bond nodes,
  fragments, fragment edges, molecules), which are guaranteed to be padding themselves, spread round-robin so
  that no padding node collects a large in-degree.

Padding therefore forms extra, disconnected "molecules" behind the real ones: real rows of every level see
exactly the edges they had, the loss is weighted by a 1/0 molecule mask, and padding rows receive zero
upstream gradient -- outputs, loss and parameter gradients of the real batch are unchanged
(tests/test_graphstep.py checks this against the oracle on CPU and against the eager path on the GPU).
Batches that do not fit the capacities fall back to the eager step, sharing optimiser and RNG state.

The all-reduce and the Adam update stay outside the graph (one RCCL call + one kernel) so the N>1 path is the
same collective as in the eager step.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, Iterable, Optional

import torch

import os

from . import _lib
from .plan import LIVE_MOLS_KEY, PLAN_KEY, REAL_MOLS_KEY, SPACES, CollatedBatch, prezeroed_plans

# field -> (index space of the ragged axis, layout, index space its VALUES point into)
FIELDS = {
    "x_atoms": ("atom", "rows", None), "batch": ("atom", "ids", "mol"), "atom_to_frag_ids": ("atom", "ids", "frag"),
    "edge_index": ("edge", "cols", "atom"), "edge_attr": ("edge", "rows", None), "node_features_bonds": ("edge", "rows", None),
    "edge_index_bonds_graph": ("bedge", "cols", "edge"), "edge_attr_bonds": ("bedge", "rows", None),
    "x_frags": ("frag", "rows", None), "frag_batch": ("frag", "ids", "mol"),
    "frag_index": ("fedge", "cols", "frag"), "cnx_attr": ("fedge", "rows", None), "node_features_fbonds": ("fedge", "rows", None),
    "edge_index_fbonds": ("fbedge", "cols", "fedge"), "edge_attr_fbonds": ("fbedge", "rows", None),
    "y": ("mol", "rows", None),
    # pretrain targets (collate_fn_pt, dataset/data.py:1028-1030)
    "bnd_lngth": ("edge", "rows", None), "bnd_angl": ("atom", "rows", None), "dh_angl": ("edge", "rows", None),
}
COUNT_FIELD = {"atom": "x_atoms", "edge": "node_features_bonds", "bedge": "edge_attr_bonds", "frag": "x_frags",
               "fedge": "node_features_fbonds", "fbedge": "edge_attr_fbonds", "mol": "y"}
MASK_KEY = "mol_weight"           # float32 [mol cap]: 1 for real molecules, 0 for padding
MASKS = {"mol": MASK_KEY, "atom": "atom_weight", "edge": "edge_weight"}     # 1/0 weights of the mean losses
SCALE_KEY = "loss_scales"         # float32 [2]: (per-edge, per-atom) rank weights of the pretrain loss, 1 on one GPU


# "thread_local": HIP calls of OTHER threads (the collective library's watchdog, a data-loader thread) do not invalidate
# a capture in progress; everything the captured step itself does runs on the capturing thread
_CAPTURE_MODE = "thread_local"


def _gat_limit(heads: int) -> int:
    """Padding in-degree that stays on the kernels' one-pass path (in-degree <= 2 * 32/heads, self loop included)."""
    return max(2, min(12, (2 * (32 // heads) * 3) // 4))


def _pairs(heads: int):
    g = _gat_limit(heads)
    # node space -> [(item space whose values point into it, max padding items per padding node)]
    return [("edge", [("bedge", g)]), ("fedge", [("fbedge", g)]), ("atom", [("edge", g)]),
            ("frag", [("fedge", g), ("atom", 32)]), ("mol", [("atom", 64), ("frag", 64)])]


def batch_counts(batch: Dict[str, torch.Tensor]) -> Dict[str, int]:
    return {sp: int(batch[f].shape[0]) for sp, f in COUNT_FIELD.items()}


def _up(x: int, q: int = 64) -> int:
    return (x + q - 1) // q * q


class StaticShapes:
    """Capacities per index space + the number of trailing slots of each node space reserved for padding."""

    def __init__(self, cap: Dict[str, int], slack: Dict[str, int], heads: int = 4, lower: Optional[Dict[str, int]] = None,
                 max_per_mol: Optional[Dict[str, int]] = None):
        self.cap, self.slack, self.heads = dict(cap), dict(slack), heads
        self.lower = dict(lower) if lower else {sp: 0 for sp in cap}       # fewest real items a batch is expected to bring
        # largest molecule the one-launch plan builder's LDS tile is sized for (None: batches carry no such bound)
        self.max_per_mol = dict(max_per_mol) if max_per_mol else None

    @classmethod
    def from_counts(cls, counts: Iterable[Dict[str, int]], margin: float = 0.03, heads: int = 4, spread_sigmas: float = 4.0) -> "StaticShapes":
        counts = list(counts)
        mx = {sp: max(c[sp] for c in counts) for sp in COUNT_FIELD}
        mn = {sp: min(c[sp] for c in counts) for sp in COUNT_FIELD}
        # a small index space varies far more from batch to batch than the margin allows for (fragment-bond-graph edges of ESOL-shape
        # batches of 512: 8.5 k ... 10.4 k, +-5 % one sigma): with three or more sample batches the bounds also cover mean +- 4 sigma
        # (spread_sigmas = 0: the sample IS the data that will run -- a benchmark pool sized from itself -- so max / min are exact)
        if len(counts) >= 3 and spread_sigmas > 0:
            for sp in COUNT_FIELD:
                v = [float(c[sp]) for c in counts]
                mean = sum(v) / len(v)
                sd = (sum((x - mean) ** 2 for x in v) / (len(v) - 1)) ** 0.5
                mx[sp] = max(mx[sp], int(math.ceil((mean + spread_sigmas * sd) / (1.0 + margin)))) if sp != "mol" else mx[sp]
                mn[sp] = min(mn[sp], max(0, int((mean - spread_sigmas * sd) / (1.0 - margin)))) if sp != "mol" else mn[sp]
        lower = {sp: int(mn[sp] * (1.0 - margin)) for sp in COUNT_FIELD}
        cap, slack = {}, {}
        for sp in ("bedge", "fbedge"):
            cap[sp] = _up(int(math.ceil(mx[sp] * (1.0 + margin))))
        for node, sources in _pairs(heads):
            need = max(-(-(cap[item] - lower[item]) // lim) for item, lim in sources)
            slack[node] = max(8, need)
            grow = 0.0 if node == "mol" else margin
            cap[node] = _up(int(math.ceil(mx[node] * (1.0 + grow))) + slack[node], 8)
            if node == "mol":       # the batch size is exact: every slot the rounding added is padding too (the head skips them)
                slack[node] = cap[node] - mx[node]
        return cls(cap, slack, heads, lower)

    @classmethod
    def from_batches(cls, batches, margin: float = 0.03, heads: int = 4, spread_sigmas: float = 4.0) -> "StaticShapes":
        batches = list(batches)
        shapes = cls.from_counts([batch_counts(b) for b in batches], margin, heads, spread_sigmas)
        bounds = [getattr(b, "max_per_mol", None) for b in batches]
        if bounds and all(m is not None for m in bounds):       # CollatedBatches: room for molecules half as large again
            shapes.max_per_mol = {sp: (1 if sp == "mol" else _up(int(1.5 * max(m[sp] for m in bounds)) + 8, 8)) for sp in SPACES}
        return shapes

    def fits(self, counts: Dict[str, int], max_per_mol: Optional[Dict[str, int]] = None) -> bool:
        if self.max_per_mol is not None and max_per_mol is not None and any(max_per_mol[sp] > self.max_per_mol[sp] for sp in SPACES):
            return False        # a molecule larger than the plan builder's tile was sized for
        for sp, n in counts.items():
            if n > self.cap[sp] - self.slack.get(sp, 0):
                return False
        for node, sources in _pairs(self.heads):
            for item, lim in sources:
                if self.cap[item] - counts[item] > lim * self.slack[node]:
                    return False
        return True

    def pad_rule(self, target: str):
        """(pad_hi, pad_mod): padding position i of a field pointing into ``target`` holds pad_hi - (i - n_real) % pad_mod."""
        return self.cap[target] - 1, self.slack[target]

    def pad_info(self) -> Dict[str, Dict[str, int]]:
        """CollatedBatch.pad of a batch staged into these shapes (plan.GraphPlan.from_batch -> fn_plan_build_mol)."""
        return {"cap": dict(self.cap), "mod": dict(self.slack),
                "hint": {sp: self.cap[sp] - self.lower.get(sp, 0) for sp in self.cap}}

    def __repr__(self):
        return f"StaticShapes(cap={self.cap}, slack={self.slack})"


def pad_batch(batch: Dict[str, torch.Tensor], shapes: StaticShapes) -> Dict[str, torch.Tensor]:
    """Reference implementation of the staging kernel with torch ops (any device): the padded batch dict."""
    counts = batch_counts(batch)
    if not shapes.fits(counts):
        raise ValueError(f"batch {counts} does not fit {shapes}")
    out = {}
    for name, (space, layout, target) in FIELDS.items():
        if name not in batch:
            continue
        src, cap, n = batch[name], shapes.cap[space], counts[space]
        if layout == "rows":
            dst = src.new_zeros((cap,) + tuple(src.shape[1:]))
            dst[:n] = src
        else:
            hi, mod = shapes.pad_rule(target)
            pad = hi - (torch.arange(cap, device=src.device, dtype=torch.long) - n) % mod
            if layout == "ids":
                dst = pad.clone()
                dst[:n] = src
            else:
                dst = pad.repeat(2, 1)
                dst[:, :n] = src
        out[name] = dst
    for space, key in MASKS.items():
        w = torch.zeros(shapes.cap[space], dtype=torch.float32, device=batch["y"].device)
        w[: counts[space]] = 1.0
        out[key] = w
    if isinstance(batch, CollatedBatch) and batch.offsets is not None:      # the staged offsets table (FN_STAGE_OFFSETS)
        out = batch.like(out)
        B, cap = counts["mol"], shapes.cap["mol"]
        off = batch.offsets.new_empty((len(SPACES), cap + 1))
        off[:, : B + 1] = batch.offsets
        off[:, B + 1:] = batch.offsets[:, B:]
        out.offsets, out.pad = off, shapes.pad_info()
        if shapes.max_per_mol is not None:
            out.max_per_mol = shapes.max_per_mol
    return out


ADAM_RIDER = os.environ.get("FRAGNET_ADAM_RIDER", "1") != "0"      # the head's Adam slice rides in the encoder backward's last launch (A/B: 0)


def masked_regr_loss(out: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """MSELoss()(out.view(-1), y) of train/utils.py:341 restricted to the molecules with weight 1."""
    if out.is_cuda:
        from . import ops
        return ops.masked_mse(out, y, w)              # loss + its gradient from one kernel
    d = out.reshape(w.shape[0], -1) - y.reshape(w.shape[0], -1)
    return (d * d * w[:, None]).sum() / (w.sum() * d.shape[1])


def masked_bce_loss(out: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """compute_bce_loss (train/utils.py:297-304): targets <= -0.5 are missing labels; padding molecules too."""
    if out.is_cuda:
        from . import ops
        return ops.masked_bce(out, y, w)
    y = y.reshape(out.shape)
    valid = (y > -0.5) & (w[:, None] > 0)
    mat = torch.nn.functional.binary_cross_entropy_with_logits(out, y.clamp(min=0.0), reduction="none")
    return torch.where(valid, mat, torch.zeros_like(mat)).sum() / valid.sum()


def masked_pretrain_loss(outputs, sb: Dict[str, torch.Tensor]) -> torch.Tensor:
    """The reference's pretrain loss (pretrain_utils.py:9-31: 2*MSE(dihedral) + MSE(angle) + MSE(energy); the
    bond-length term is overwritten there before use) over a padded batch, with the rank weights of
    parallel.weighted_loss_scale applied to the per-edge and per-atom means."""
    from . import ops
    _, ba, da, graph_rep = outputs
    if da.is_cuda:       # 2 * sc0 * MSE(dihedral) + sc1 * MSE(angle) + MSE(energy) and its gradients: two launches
        return ops.masked_mse_multi([(2.0, 0), (1.0, 1), (1.0, -1)], sb[SCALE_KEY],
                                    da, sb["dh_angl"], sb[MASKS["edge"]], ba, sb["bnd_angl"], sb[MASKS["atom"]],
                                    graph_rep, sb["y"], sb[MASK_KEY])
    sc = sb[SCALE_KEY]
    l_dh = ops.masked_mse(da, sb["dh_angl"], sb[MASKS["edge"]]) * sc[0]
    return l_dh + ops.masked_mse(ba, sb["bnd_angl"], sb[MASKS["atom"]]) * sc[1] + l_dh \
        + ops.masked_mse(graph_rep, sb["y"], sb[MASK_KEY])


class StaticBatch:
    """Fixed-capacity device buffers for one batch + the single-launch staging call."""

    def __init__(self, shapes: StaticShapes, example: Dict[str, torch.Tensor]):
        dev = example["x_atoms"].device
        if dev.type != "cuda":
            raise _lib.FragnetHipError("StaticBatch stages batches on the GPU (fn_stage_padded); there is no CPU path")
        self.shapes, self.device = shapes, dev
        # the padded batch keeps the example's layout promise for its real molecules (the fused kernels treat the padding
        # molecules, whose items point round-robin at the reserved slots, separately: REAL_MOLS_KEY)
        self.collated = isinstance(example, CollatedBatch)
        self.t: Dict[str, torch.Tensor] = CollatedBatch() if self.collated else {}
        # per-molecule offsets table of the staged batch: drives the one-launch plan builder (fn_plan_build_mol)
        self.with_offsets = self.collated and example.offsets is not None and example.offsets.is_cuda and shapes.max_per_mol is not None
        if self.with_offsets:
            self.t.offsets = torch.zeros((len(SPACES), shapes.cap["mol"] + 1), dtype=torch.int32, device=dev)
            self.t.max_per_mol, self.t.pad = dict(shapes.max_per_mol), shapes.pad_info()
        self._desc = []
        for name, (space, layout, target) in FIELDS.items():
            if name not in example:
                continue
            src, cap = example[name], shapes.cap[space]
            if layout == "rows":
                if src.dtype != torch.float32:
                    raise TypeError(f"{name}: expected float32 rows, got {src.dtype}")
                self.t[name] = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=torch.float32, device=dev)
                width = int(src[0].numel()) if src.dim() > 1 else 1
                self._desc.append((name, space, _lib.STAGE_ROWS, max(width, 1), 0, 1))
            else:
                if src.dtype != torch.int64:
                    raise TypeError(f"{name}: expected int64 indices, got {src.dtype}")
                hi, mod = shapes.pad_rule(target)
                self.t[name] = torch.zeros((cap,) if layout == "ids" else (2, cap), dtype=torch.int64, device=dev)
                self._desc.append((name, space, _lib.STAGE_IDS if layout == "ids" else _lib.STAGE_COLS, 1, hi, mod))
        for space, key in MASKS.items():
            self.t[key] = torch.zeros(shapes.cap[space], dtype=torch.float32, device=dev)
        self.t[SCALE_KEY] = torch.ones(2, dtype=torch.float32, device=dev)
        self._fields = (_lib.StageField * _lib.FN_MAX_STAGE_FIELDS)()
        for i, (name, space, kind, width, hi, mod) in enumerate(self._desc):
            f = self._fields[i]
            f.dst, f.cap, f.width, f.kind, f.pad_hi, f.pad_mod = self.t[name].data_ptr(), shapes.cap[space], width, kind, hi, mod
        self._mask_spaces = list(MASKS)
        for q, space in enumerate(self._mask_spaces):
            m = self._fields[len(self._desc) + q]
            m.dst, m.cap, m.width, m.kind, m.pad_hi, m.pad_mod = self.t[MASKS[space]].data_ptr(), shapes.cap[space], 1, _lib.STAGE_MASK, 0, 1
        # device-side copy of the number of real molecules: the fused encoder kernels skip the padding behind them
        self.t[REAL_MOLS_KEY] = torch.zeros(1, dtype=torch.int32, device=dev)
        c = self._fields[len(self._desc) + len(self._mask_spaces)]
        c.dst, c.cap, c.width, c.kind, c.pad_hi, c.pad_mod = self.t[REAL_MOLS_KEY].data_ptr(), 1, 1, _lib.STAGE_COUNT, 0, 1
        self.n_fields = len(self._desc) + len(self._mask_spaces) + 1
        if self.with_offsets:
            o = self._fields[self.n_fields]
            o.dst, o.cap, o.width, o.kind, o.pad_hi, o.pad_mod = self.t.offsets.data_ptr(), shapes.cap["mol"], len(SPACES), _lib.STAGE_OFFSETS, 0, 1
            self._off_field = self.n_fields
            self.n_fields += 1
        # molecule rows behind capacity - slack are padding in every batch that fits: the prediction head skips them
        self.t[LIVE_MOLS_KEY] = shapes.cap["mol"] - shapes.slack["mol"]
        if self.n_fields > _lib.FN_MAX_STAGE_FIELDS:
            raise ValueError("too many batch fields for one staging launch")
        self.counts: Optional[Dict[str, int]] = None
        self._n_static = self.n_fields

    def set_bumps(self, bumps, zeros=()):
        """Extra work of the staging launch in front of every replay of a captured step.  ``bumps`` = [(int64 device tensor
        [1], increment)]: counters it advances (FN_STAGE_BUMP: the step's Philox block counter and optimiser step count);
        ``zeros`` = [(device pointer, int32 count)]: workspaces it zeroes (FN_STAGE_ZERO: those of the plans built inside the
        graph).  The graph then needs no launch of its own for either."""
        self.n_fields = self._n_static
        extra = [(None, t.data_ptr(), int(inc), 1, _lib.STAGE_BUMP) for t, inc in bumps] + \
                [(None, int(ptr), 0, int(n), _lib.STAGE_ZERO) for ptr, n in zeros]
        for src, dst, n_real, cap, kind in extra:
            if self.n_fields >= _lib.FN_MAX_STAGE_FIELDS:
                raise ValueError("too many batch fields for one staging launch")
            f = self._fields[self.n_fields]
            f.src, f.dst, f.n_real, f.cap, f.width, f.kind, f.pad_hi, f.pad_mod = src, dst, n_real, cap, 1, kind, 0, 1
            self.n_fields += 1

    def load(self, batch: Dict[str, torch.Tensor]) -> bool:
        """Stage ``batch`` (GPU tensors); False when it does not fit the capacities (nothing is written then)."""
        counts = batch_counts(batch)
        if not self.shapes.fits(counts, getattr(batch, "max_per_mol", None)):
            return False
        if self.collated and not isinstance(batch, CollatedBatch):
            return False        # no layout promise: the caller's eager step takes the general kernels
        if self.with_offsets:
            off = batch.offsets
            if off is None or not off.is_cuda or batch.max_per_mol is None:
                return False
            off = off if off.is_contiguous() else off.contiguous()
            o = self._fields[self._off_field]
            o.src, o.n_real = off.data_ptr(), counts["mol"]
        keep = []
        for i, (name, space, kind, width, hi, mod) in enumerate(self._desc):
            src = batch[name]
            if not src.is_cuda:
                raise _lib.FragnetHipError(f"{name} must live on the GPU to be staged (got {src.device})")
            if not src.is_contiguous():
                src = src.contiguous()
                keep.append(src)
            f = self._fields[i]
            f.src, f.n_real = src.data_ptr(), counts[space]
        for q, space in enumerate(self._mask_spaces):
            m = self._fields[len(self._desc) + q]
            m.src, m.n_real = None, counts[space]
        c = self._fields[len(self._desc) + len(self._mask_spaces)]
        c.src, c.n_real = None, counts["mol"]
        _lib.call("fn_stage_padded", self._fields, self.n_fields, torch.cuda.current_stream(self.device).cuda_stream)
        self.counts = counts
        self._keep = (keep, batch.offsets if self.with_offsets else None)       # alive until the launch has run
        return True


class GraphedTrainStep:
    """zero_grad + forward + loss + backward + gradient gather as one hipGraph; all-reduce + Adam after it.

    ``model``: FragNetFineTune / FragNetPreTrain (fragnet_amd.model) in train mode; ``opt``: parallel.FlatAdam over its
    live parameters; ``loss``: "regr" (MSE, train/utils.py:341), "clsf" (masked BCE, train/utils.py:297) or "pretrain"
    (pretrain_utils.py:9-31 on the collate_fn_pt batch).
    """

    def __init__(self, model, opt, shapes: StaticShapes, example: Dict[str, torch.Tensor], loss: str = "regr",
                 group=None, warmup: int = 3, overlap: Optional[bool] = None, capture_adam: bool = True,
                 force_distributed: bool = False):
        """``overlap=True``: capture the step as TWO graphs -- (forward + loss + head backward) and (encoder backward) -- and
        start an asynchronous all-reduce of the head's gradients (83 % of the bytes for FTHead3) between them, so that it
        runs on RCCL's stream beside the encoder's backward pass; the encoder's own, small slice follows the second graph.
        Off by default: on this stack an async collective costs ~95 us of stream fork/join against ~20 us for a blocking
        one (tools/allreduce_probe.py), which eats what the overlap hides at 8 MB of gradients (DESIGN.md section 7 and HISTORY.md section 7)."""
        self.model, self.opt, self.shapes, self.group = model, opt, shapes, group
        # force_distributed: run the N>1 step sequence (graph replay -> RCCL all-reduce -> Adam outside the graph, rank loss
        # weights through an all-reduce) even in a 1-rank process group, so that one GPU can test it
        self.force_distributed = bool(force_distributed)
        if self.force_distributed:
            opt.force_collective = True
        self.loss_kind = loss
        if loss not in ("regr", "clsf", "pretrain"):
            raise ValueError(f"unknown loss kind {loss!r}")
        self._masked = {"regr": masked_regr_loss, "clsf": masked_bce_loss, "pretrain": None}[loss]
        self.static = StaticBatch(shapes, example)
        self.device = self.static.device
        if loss == "pretrain" and "dh_angl" not in example:
            raise ValueError("pretrain step needs the collate_fn_pt batch (bnd_lngth, bnd_angl, dh_angl)")
        if loss == "pretrain" and hasattr(getattr(model, "head", None), "need_bond_length"):
            model.head.need_bond_length = False      # the loss never reads the bond-length prediction (pretrain_utils.py:17-24)
        self.rng = model.pretrain.rng
        # [Philox blocks consumed, Adam steps done]: one captured vector add moves both on every replay
        self._counters = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.rng.dev = self._counters[0:1]
        import torch.distributed as dist
        single = (not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1) and not self.force_distributed
        # one rank, fused HIP Adam: the update is captured too (step count and learning rate live in device memory)
        self.adam_in_graph = bool(capture_adam) and single and getattr(opt, "opt", 1) is None
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._lr_host = None
        self._dev_steps = -1
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.loss: Optional[torch.Tensor] = None
        self.replays = self.fallbacks = 0
        # Philox bookkeeping of the captured step (set by _capture): the host offset the capture started from, the blocks one
        # step draws, and whether the staging launch advances the device counters (single-graph step).  Invariant: between
        # replays the device counter is ONE STEP BEHIND when _stage_bumps is set -- the staging launch in front of every
        # replay moves it on -- so anything that replays without staging (or stages without replaying) must go through replay()
        self._rng_base, self._per_step, self._stage_bumps = 0, 0, False
        from . import ops
        self._unit = ops.unit_grad(self.static.device)
        self.graph_b: Optional[torch.cuda.CUDAGraph] = None
        self.head_off = self._head_offset()
        self.split = bool(overlap) and self.head_off is not None
        self._rider_lo: Optional[int] = None         # set by the warm-up steps of _capture: where the Adam slice that may ride starts
        self._capture(example, warmup)

    # -- pieces
    def _head_offset(self) -> Optional[int]:
        """Offset of the head's parameters in the optimiser's flat buffer if they are its contiguous tail (model =
        encoder ``pretrain`` + ``fthead``, regression / classification loss), else None (single-graph step)."""
        head = getattr(self.model, "fthead", None)
        if self.loss_kind == "pretrain" or head is None or not hasattr(self.model, "pretrain"):
            return None
        ids = {id(p) for p in head.parameters()}
        first = None
        for p, off in zip(self.opt.params, self.opt.offsets):
            if id(p) in ids:
                if first is None:
                    first = off
            elif first is not None:
                return None                                  # an encoder parameter after a head parameter
        return first

    def _head_grads_in_place(self) -> bool:
        base, es, ids = self.opt.grad.data_ptr(), self.opt.grad.element_size(), {id(p) for p in self.model.fthead.parameters()}
        for p, off in zip(self.opt.params, self.opt.offsets):
            if id(p) in ids and (p.grad is None or p.grad.data_ptr() != base + off * es):
                return False
        return True

    def _part_a(self):
        """forward, loss, and the backward pass of loss + head; leaves d loss / d pooled in ``leaf.grad``."""
        from .model import _KEEP_EDGE_OUTPUTS, pooled
        sb = self.static.t
        sb.pop(PLAN_KEY, None)
        x_atoms, x_frags, _, _ = self.model.pretrain(sb, edge_outputs=_KEEP_EDGE_OUTPUTS)
        pooled_t = pooled(x_atoms, x_frags, sb)
        leaf = pooled_t.detach().requires_grad_(True)
        self.model.fthead.live_rows = sb.get(LIVE_MOLS_KEY)
        loss = self._head_loss(lambda: self.model.fthead(leaf), sb["y"], sb[MASK_KEY])
        loss.backward(gradient=self._unit)
        return loss, pooled_t, leaf

    def _head_loss(self, run, y, w):
        """predictions -> loss.  A predictor-stack head is told the targets first (``loss_spec``): its last Linear, the loss and that
        Linear's backward then share one launch (ops.mlp_head) and the predictions come back with the loss attached; every other
        head, and a stack the fused launch does not cover, goes through the loss kernel of its own."""
        head = getattr(self.model, "fthead", None)
        armed = head is not None and hasattr(head, "loss_spec") and y.is_cuda
        if armed:
            from . import _lib
            head.loss_spec = (_lib.LOSS_MSE if self.loss_kind == "regr" else _lib.LOSS_BCE, y, w)
        try:
            out = run()
        finally:
            if armed:
                head.loss_spec = None
        fused = getattr(out, "_fragnet_loss", None)
        if fused is not None and fused[1] is y and fused[2] is w:
            return fused[0]
        return self._masked(out, y, w)

    def _part_b(self, pooled_t, leaf):
        pooled_t.backward(leaf.grad)
        self.opt.gather_grads()

    def _fwd_bwd_static(self, ride_adam: bool = False):
        """forward, loss, backward, gradients gathered.  ``ride_adam`` (the captured single-graph step with Adam inside): the
        head's slice of the Adam update -- its gradients are complete once the head's backward has run -- rides in the encoder
        backward's last launch; returns (loss, first element of the flat buffer that rode, or None)."""
        sb = self.static.t
        sb.pop(PLAN_KEY, None)                      # every step builds its own graph plan (inside the graph)
        if self.loss_kind == "pretrain":
            loss = masked_pretrain_loss(self.model(sb), sb)
        else:
            loss = self._head_loss(lambda: self.model(sb), sb["y"], sb[MASK_KEY])
        rode = None
        if ride_adam and ADAM_RIDER and self._rider_lo is not None:
            from . import engine
            keep = self.opt.adam_slice(self._counters[1:2], self._lr_dev, self._rider_lo)
            engine.arm_adam_rider(keep)
            try:
                loss.backward(gradient=self._unit)
            finally:
                if engine.adam_rider_taken():
                    rode = self._rider_lo
        else:
            loss.backward(gradient=self._unit)      # persistent 1.0: no fill launch, masked_mse skips the multiply
        self.opt.gather_grads()
        return loss, rode

    def _capture(self, example, warmup: int):
        if not self.model.training:
            raise RuntimeError("GraphedTrainStep captures a TRAINING step: call model.train() first")
        if not self.static.load(example):
            raise ValueError(f"the example batch {batch_counts(example)} does not fit {self.shapes}")
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        per_step = 0
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.opt.zero_grad()
                before = self.rng.offset
                if self.split:
                    _, pooled_t, leaf = self._part_a()
                    if not self._head_grads_in_place():      # e.g. a non-ReLU head (torch's own backward): one graph
                        self.split = False
                    self._part_b(pooled_t, leaf)
                    del pooled_t, leaf
                else:
                    self._fwd_bwd_static()
                    # may the head's Adam slice ride in the encoder's backward?  Only if its gradients sit in the flat buffer already
                    # when that backward starts (the fused head writes them there) and its parameters are the buffer's tail
                    off = self._head_offset()
                    self._rider_lo = off if (off is not None and off % 4 == 0 and self._head_grads_in_place()) else None
                per_step = self.rng.offset - before              # Philox blocks one step draws (fixed by the static shapes)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.opt.zero_grad()
        if self.split:
            self.adam_in_graph = False
        # what one replay adds to [Philox blocks consumed, Adam steps done]; built here because a captured region cannot copy from the host
        inc = self._inc = torch.tensor([per_step, 1 if self.adam_in_graph else 0], dtype=torch.int64, device=self.device)      # kept alive: replays read it
        self._sync_adam_state()
        graph = torch.cuda.CUDAGraph()
        off0 = self._rng_base = self.rng.offset
        if self.split:
            pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode=_CAPTURE_MODE):
                loss, pooled_t, leaf = self._part_a()
                consumed = self.rng.offset - off0
                if consumed:
                    self.rng.advance_device(consumed)
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=pool, capture_error_mode=_CAPTURE_MODE):
                self._part_b(pooled_t, leaf)
            del pooled_t, leaf
        else:
            # plans built inside the capture skip their zeroing launch: the staging launch zeroes their workspaces (below)
            with prezeroed_plans() as pz, torch.cuda.graph(graph, capture_error_mode=_CAPTURE_MODE):
                loss, rode = self._fwd_bwd_static(ride_adam=self.adam_in_graph)
                if self.rng.offset - off0 != per_step:
                    raise RuntimeError("the captured step drew a different number of Philox blocks than the warm-up steps")
                # fresh dropout masks and the next Adam step number on every replay: the staging launch in front of the replay
                # advances both counters (FN_STAGE_BUMP), so the graph has no launch of its own for them
                if self.adam_in_graph:
                    self.opt.adam_in_graph(self._counters[1:2], self._lr_dev, 0, rode)      # what did not ride in the backward
        self.graph, self.loss = graph, loss.detach()
        # single-graph step: the staging launch in front of every replay advances the counters.  The Philox counter is
        # therefore "one step behind" between replays (it starts at -per_step so that the first replay sees 0).
        self._per_step, self._stage_bumps = int(per_step), not self.split
        if self._stage_bumps:
            bumps = []
            if per_step:
                bumps.append((self._counters[0:1], per_step))
                self._counters[0:1] -= per_step
            if self.adam_in_graph:
                bumps.append((self._counters[1:2], 1))
            self._captured_plans = pz.plans          # keep their arenas (the regions below) alive with the graph
            self.static.set_bumps(bumps, zeros=[r for plan in pz.plans for r in plan.zero_regions])

    def _sync_adam_state(self):
        """Device copies of the optimiser's step count and learning rate follow the host values (an eager fallback step,
        a scheduler or a restored checkpoint may have moved them)."""
        if not self.adam_in_graph:
            return
        if self._dev_steps != self.opt.steps:
            self._counters[1:2].fill_(self.opt.steps)
            self._dev_steps = self.opt.steps
        lr = float(self.opt.hyper["lr"])
        if self._lr_host != lr:
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    def _rank_scales(self, batch) -> Optional[torch.Tensor]:
        """(per-edge, per-atom) loss weights local_count * world / global_count as a device tensor (no host sync)."""
        import torch.distributed as dist
        if self.loss_kind != "pretrain" or not (dist.is_available() and dist.is_initialized()) or \
                (dist.get_world_size(self.group) == 1 and not self.force_distributed):
            return None
        local = torch.tensor([float(batch["dh_angl"].shape[0]), float(batch["bnd_angl"].shape[0])], device=self.device)
        total = local.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return (local * dist.get_world_size(self.group) / total).to(torch.float32)

    def _eager(self, batch):
        self.opt.zero_grad()
        # draw from the same base offsets the captured step has baked in (the device counter, which the eager kernels add as
        # well, makes them this step's own), then move the device counter past what was drawn: fallback steps and replays
        # never share Philox blocks
        host_after = self.rng.offset
        self.rng.offset = self._rng_base
        behind = self._per_step if self._stage_bumps else 0      # see _capture: the staging launch bumps the counter
        if behind:
            self._counters[0:1] += behind
        if self.loss_kind == "pretrain":
            from .train import pretrain_loss
            sc = self._rank_scales(batch)
            loss = pretrain_loss(self.model(batch), batch, (1.0, 1.0) if sc is None else (sc[0], sc[1]))
        else:
            out = self.model(batch)
            w = torch.ones(batch["y"].shape[0], dtype=torch.float32, device=out.device)
            loss = self._masked(out, batch["y"], w)
        loss.backward()
        if self.split:
            # the ranks that replay the two graphs issue two slice all-reduces (head, then encoder): a rank that fell back to
            # the eager step must issue the same collectives, in the same order and sizes
            self.opt.gather_grads()
            self.opt.all_reduce_slice(self.head_off, None, self.group)
            self.opt.all_reduce_slice(0, self.head_off, self.group)
            self.opt.apply_gathered(self.group, reduced=True)
        else:
            self.opt.step(self.group)
        # Philox blocks this eager step drew are gone for the replays too: without this the fallback step and the replay after
        # next would share dropout random numbers (the graph adds the device counter to offsets baked in at capture)
        drawn = self.rng.offset - self._rng_base
        if drawn - behind:
            self._counters[0:1] += drawn - behind
        self.rng.offset = host_after
        return loss.detach()

    def replay(self, batch: Dict[str, torch.Tensor]) -> bool:
        """Stage ``batch`` and replay the captured graph(s) -- always as a pair: the staging launch zeroes the plan workspaces
        and advances the Philox / step counters the graph reads.  No collective, no optimiser step outside the graph (tools
        that time or profile the captured part use this instead of ``static.load`` + ``graph.replay``).  False: the batch does
        not fit the capacities (nothing ran)."""
        self._sync_adam_state()
        if not self.static.load(batch):
            return False
        self.graph.replay()
        if self.graph_b is not None:
            self.graph_b.replay()
        if self.adam_in_graph:
            self.opt.steps += 1
            self._dev_steps += 1
        self.replays += 1
        return True

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """One optimiser step on ``batch``.  Returns the loss (a tensor that the next call overwrites)."""
        self._sync_adam_state()                      # before the staging launch: it bumps the step counter
        if not self.static.load(batch):
            self.fallbacks += 1
            return self._eager(batch)
        sc = self._rank_scales(batch)
        if sc is not None:
            self.static.t[SCALE_KEY].copy_(sc)
        self.graph.replay()
        if self.adam_in_graph:                       # the update ran inside the graph
            self.opt.steps += 1
            self._dev_steps += 1
        elif self.split:
            head = self.opt.all_reduce_slice(self.head_off, None, self.group, async_op=True)     # beside graph_b
            self.graph_b.replay()
            rest = self.opt.all_reduce_slice(0, self.head_off, self.group, async_op=True)
            for work in (head, rest):
                if work is not None:
                    work.wait()                     # stream-side wait, the host does not block
            self.opt.apply_gathered(self.group, reduced=True)
        else:
            self.opt.apply_gathered(self.group)
        self.replays += 1
        return self.loss


class GraphedForward:
    """Inference counterpart of GraphedTrainStep: stage + plan build + ``model(batch)`` (eval mode, no autograd) as one
    hipGraph over static shapes.  ``__call__`` returns the predictions of the real molecules (a view of a buffer the next
    call overwrites); batches beyond the capacities run eagerly.  Useful when the host is the bottleneck (small batches,
    busy CPU); on an idle host the eager forward is already GPU-bound and the staging copy makes this path 5-10 % slower
    (745 k vs 784 k molecules/s at 512 molecules, 1.15 M vs 1.27 M at 8192), so bench.py's forward sweep stays eager."""

    def __init__(self, model, shapes: StaticShapes, example: Dict[str, torch.Tensor], warmup: int = 2):
        if model.training:
            raise RuntimeError("GraphedForward captures an inference pass: call model.eval() first")
        self.model, self.shapes = model, shapes
        self.static = StaticBatch(shapes, example)
        self.device = self.static.device
        self.replays = self.fallbacks = 0
        if not self.static.load(example):
            raise ValueError(f"the example batch {batch_counts(example)} does not fit {self.shapes}")
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self.out = self._run()

    def _run(self):
        sb = self.static.t
        sb.pop(PLAN_KEY, None)
        return self.model(sb)

    def __call__(self, batch: Dict[str, torch.Tensor]):
        n = batch[COUNT_FIELD["mol"]].shape[0] if COUNT_FIELD["mol"] in batch else int(batch["batch"].max()) + 1
        if not self.static.load(batch):
            self.fallbacks += 1
            batch.pop(PLAN_KEY, None)
            with torch.no_grad():
                return self.model(batch)
        self.graph.replay()
        self.replays += 1
        out = self.out
        return tuple(o[:n] if torch.is_tensor(o) and o.shape[0] == self.shapes.cap["mol"] else o for o in out) \
            if isinstance(out, tuple) else out[:n]

