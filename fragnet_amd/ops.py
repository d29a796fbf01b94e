"""torch.autograd wrappers around the C-ABI kernels (include/fragnet_hip.h).

PyTorch is plumbing here: it owns the device memory, the current HIP stream and the autograd tape.
Every numeric step of the message-passing path runs in libfragnet_hip.so; CPU tensors are rejected.

Operator surface mirrored from torch-scatter (the reference's fragnet/model/gat/gat2.py:5):
    scatter_add(src, index, dim=0, out=None, dim_size=None)
    scatter_softmax(src, index, dim=0, dim_size=None)
plus the fused per-level op the model uses (gat_level) and the small epilogues around it.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import torch

from . import _lib
from ._lib import EdgeTerm, FN_D, FN_MAX_PART
from .plan import GraphPlan, Level, Segments, _stream_ptr

NEG_SLOPE = 0.2   # nn.LeakyReLU(0.2), reference gat2.py:83
BWD_ONE_PASS = True   # per-level operator path: the attention backward as one source-owner pass (False: destination + source pass)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.FragnetHipError(f"{name}: fragnet_amd kernels need GPU tensors (got {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def grad_slot(param):
    """The (flat gradient buffer, offset) parallel.FlatAdam assigned to ``param``, or None."""
    return getattr(param, "_fn_grad_slot", None)


def grad_buffer(param, slot):
    """Where a hand-written backward writes d loss / d param: the parameter's slot of the optimiser's flat gradient
    buffer when there is one, nothing has been accumulated yet and no other node of this backward pass has taken it
    (autograd then adopts the returned view as ``param.grad`` and FlatAdam.gather_grads finds it in place), else a
    fresh tensor.  The slot is handed out once per pass: a parameter that feeds two custom nodes (the model called on two
    batches before one ``backward()``) would otherwise have both kernels write the same memory and autograd add the
    tensor to its own alias (2 x the second gradient).  FlatAdam.zero_grad / gather_grads release the claims."""
    if slot is not None and param.grad is None and not getattr(param, "_fn_slot_claimed", False):
        flat, off = slot
        param._fn_slot_claimed = True
        return flat[off: off + param.numel()].view(param.shape)
    return torch.empty_like(param)


def _part_rows(n: int) -> int:
    return max(1, min((n + 7) // 8, FN_MAX_PART))


# ======================================================================================
# one attention level
# ======================================================================================
class _GatLevel(torch.autograd.Function):
    """out[n,128] = sum_e softmax_dst(LeakyReLU(s_dst + s_src + s_edge))_e * h[src_e].

    Inputs (differentiable): h [n,128]; att [H, att_w]; then either s_sorted [H,m] (mode 0: the edge term in
    destination-sorted order, head-major, from row_dots_sorted) or x_sorted [K,m] (mode 2: the raw edge attribute
    in destination-sorted order, not differentiated) with embW [d,K], embb [d]."""

    @staticmethod
    def forward(ctx, h, att, s_sorted, x_sorted, embW, embb, level: Level, heads: int, dst_off: int, mid_off: int,
                src_off: int, want_probs: bool):
        h = _f32c(h, "h")
        att = _f32c(att, "att")
        dev = h.device
        n, m = level.n, level.m
        if h.shape != (n, FN_D):
            raise ValueError(f"h must be [{n}, {FN_D}], got {tuple(h.shape)}")
        att_w = att.shape[1]
        mode = 0 if x_sorted is None else 2
        if mode == 0:
            s_sorted = _f32c(s_sorted, "s_sorted")
            if s_sorted.shape != (heads, m):
                raise ValueError(f"s_sorted must be head-major [{heads}, {m}], got {tuple(s_sorted.shape)}")
            et = EdgeTerm(0, 0, 0, 0, s_sorted.data_ptr(), None, None, None)
        else:
            x_sorted, embW, embb = _f32c(x_sorted, "x_sorted"), _f32c(embW, "embW"), _f32c(embb, "embb")
            K, d_e = x_sorted.shape[0], embW.shape[0]     # d_e = head_dim (gat2's a_b / f_a_b) or 128 (gat2_edge's f)
            if x_sorted.shape[1] != m or embW.shape[1] != K or embb.shape[0] != d_e or mid_off + d_e > src_off:
                raise ValueError("edge attribute / embedding shapes do not match the plan and the attention vector")
            et = EdgeTerm(2, K, d_e, mid_off, None, x_sorted.data_ptr(), embW.data_ptr(), embb.data_ptr())
        st = _stream_ptr(dev)
        s_dst = torch.empty((n, heads), dtype=torch.float32, device=dev)
        s_src = torch.empty((n, heads), dtype=torch.float32, device=dev)
        _lib.call("fn_node_scalars_f32", h.data_ptr(), att.data_ptr(), att_w, dst_off, src_off, s_dst.data_ptr(),
                  s_src.data_ptr(), n, heads, st)
        out = torch.empty((n, FN_D), dtype=torch.float32, device=dev)
        p_sorted = torch.empty((heads, m), dtype=torch.float32, device=dev)      # head-major
        probs = torch.empty((m, heads), dtype=torch.float32, device=dev) if want_probs else None
        # the one-pass backward (fn_gat_bwd_one_f32) needs the forward's second output row and its per-head weight sum
        # (levels beyond the one-pass kernel's 32-bit byte offsets keep the destination + source passes)
        one = BWD_ONE_PASS and any(ctx.needs_input_grad) and n <= (1 << 23) and m * heads <= (1 << 28)
        out2 = torch.empty((n, FN_D), dtype=torch.float32, device=dev) if one else None
        sigma = torch.empty((n, heads), dtype=torch.float32, device=dev) if one else None
        _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w,
                  C.byref(et), C.byref(level.c), NEG_SLOPE, out.data_ptr(), p_sorted.data_ptr(), _ptr(probs), _ptr(out2), _ptr(sigma),
                  0, None, heads, st)
        ctx.level, ctx.heads, ctx.mode = level, heads, mode
        ctx.offs = (dst_off, mid_off, src_off)
        ctx.one = one
        if one:
            ctx.save_for_backward(h, att, p_sorted, x_sorted, embW, embb, out, out2, sigma)
        else:
            ctx.save_for_backward(h, att, p_sorted, x_sorted, embW, embb)
        ctx.set_materialize_grads(False)
        if want_probs:
            ctx.mark_non_differentiable(probs, p_sorted)
            return out, probs, p_sorted
        return out

    @staticmethod
    def backward(ctx, g_out, *unused):
        h, att, p_sorted, x_sorted, embW, embb = ctx.saved_tensors[:6]
        level, heads, mode = ctx.level, ctx.heads, ctx.mode
        dst_off, mid_off, src_off = ctx.offs
        if g_out is None:
            return (None,) * 12
        g_out = _f32c(g_out, "g_out")
        dev = h.device
        n, m = level.n, level.m
        att_w = att.shape[1]
        st = _stream_ptr(dev)
        if mode == 0:
            et = EdgeTerm(0, 0, 0, 0, None, None, None, None)
            part_e = None
        else:
            K = x_sorted.shape[0]
            et = EdgeTerm(2, K, embW.shape[0], mid_off, None, x_sorted.data_ptr(), embW.data_ptr(), embb.data_ptr())
            part_e = torch.empty((FN_MAX_PART, heads * (K + 1)), dtype=torch.float32, device=dev)
        dz = torch.empty((heads, m), dtype=torch.float32, device=dev) if mode == 0 else None
        g_s_dst = torch.empty((n, heads), dtype=torch.float32, device=dev)
        n_e, n_a = C.c_int(0), C.c_int(0)
        g_h = torch.empty((n, FN_D), dtype=torch.float32, device=dev)
        part_a = torch.empty((FN_MAX_PART, 2 * FN_D), dtype=torch.float32, device=dev)
        if ctx.one:
            # one source-owner pass: c = <g, out> and g_s_dst = <g, out2> - c sigma are node-local (csrc/gat_bwd_one.inc)
            out, out2, sigma = ctx.saved_tensors[6:]
            cdot = torch.empty((n, heads), dtype=torch.float32, device=dev)
            _lib.call("fn_gat_cu_f32", g_out.data_ptr(), out.data_ptr(), out2.data_ptr(), sigma.data_ptr(), 1.0, cdot.data_ptr(),
                      g_s_dst.data_ptr(), n, heads, st)
            _lib.call("fn_gat_bwd_one_f32", g_out.data_ptr(), h.data_ptr(), p_sorted.data_ptr(), cdot.data_ptr(), g_s_dst.data_ptr(),
                      C.byref(et), att.data_ptr(), att_w, dst_off, src_off, C.byref(level.c), NEG_SLOPE, g_h.data_ptr(), _ptr(dz), None,
                      part_a.data_ptr(), C.byref(n_a), _ptr(part_e), C.byref(n_e), 0, None, heads, st)
        else:
            pz = torch.empty((heads, m, 2), dtype=torch.float32, device=dev)
            _lib.call("fn_gat_bwd_dst_f32", g_out.data_ptr(), h.data_ptr(), p_sorted.data_ptr(), C.byref(et),
                      C.byref(level.c), NEG_SLOPE, _ptr(dz), None, pz.data_ptr(), g_s_dst.data_ptr(), _ptr(part_e), C.byref(n_e), heads, st)
            _lib.call("fn_gat_bwd_src_f32", g_out.data_ptr(), h.data_ptr(), pz.data_ptr(), g_s_dst.data_ptr(), att.data_ptr(), att_w, dst_off, src_off, C.byref(level.c), g_h.data_ptr(),
                      part_a.data_ptr(), C.byref(n_a), heads, st)
        g_att = torch.zeros_like(att)
        g_embW = torch.empty_like(embW) if mode == 2 else None
        g_embb = torch.empty_like(embb) if mode == 2 else None
        _lib.call("fn_gat_bwd_finalize_f32", part_a.data_ptr(), n_a.value, _ptr(part_e), n_e.value, C.byref(et),
                  att.data_ptr(), att_w, dst_off, src_off, g_att.data_ptr(), _ptr(g_embW), _ptr(g_embb), heads, st)
        # dL/ds_sorted is dz itself (mode 0)
        return g_h, g_att, (dz if mode == 0 else None), None, g_embW, g_embb, None, None, None, None, None, None


def gat_level(h, att, level: Level, heads: int, *, s_sorted=None, x_sorted=None, embW=None, embb=None, want_probs=False):
    """``att`` = [dst(d) | edge | src(d)] per head, the reference's a_b / a / f / f_a_b layout."""
    d = FN_D // heads
    att_w = att.shape[1]
    return _GatLevel.apply(h, att, s_sorted, x_sorted, embW, embb, level, heads, 0, d, att_w - d, want_probs)


def attn_by_src(p_sorted: torch.Tensor, level: Level, heads: int) -> torch.Tensor:
    """scatter_add(attn_probs, source): attention mass per source node (gat2.py:165,219,268,312)."""
    out = torch.empty((level.n, heads), dtype=torch.float32, device=p_sorted.device)
    _lib.call("fn_attn_by_src_f32", p_sorted.data_ptr(), C.byref(level.c), out.data_ptr(), heads, _stream_ptr(out.device))
    return out


# ======================================================================================
# full-width edge term, destination-sorted: s_sorted[pos, j] = <feat[eid(pos), :], A[j, off:off+128]>
# ======================================================================================
class _RowDotsSorted(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, A, off: int, level: Level):
        feat, A = _f32c(feat, "feat"), _f32c(A, "A")
        J = A.shape[0]
        if feat.shape != (level.m_real, FN_D):
            raise ValueError(f"feat must be [{level.m_real}, {FN_D}], got {tuple(feat.shape)}")
        s = torch.empty((J, level.m), dtype=torch.float32, device=feat.device)      # head-major
        _lib.call("fn_row_dots_sorted_f32", feat.data_ptr(), A.data_ptr(), A.shape[1], off, J, C.byref(level.c), s.data_ptr(),
                  _stream_ptr(feat.device))
        ctx.off, ctx.level = off, level
        ctx.save_for_backward(feat, A)
        return s

    @staticmethod
    def backward(ctx, g_s):
        feat, A = ctx.saved_tensors
        level = ctx.level
        g_s = _f32c(g_s, "g_s")
        J = A.shape[0]
        st = _stream_ptr(feat.device)
        g_feat = torch.empty_like(feat)
        g_A = torch.zeros_like(A)
        if level.m_real:
            part = torch.empty((FN_MAX_PART, J * FN_D), dtype=torch.float32, device=feat.device)
            n_part = C.c_int(0)
            _lib.call("fn_row_dots_sorted_bwd_f32", g_s.data_ptr(), feat.data_ptr(), A.data_ptr(), A.shape[1], ctx.off, J,
                      C.byref(level.c), g_feat.data_ptr(), part.data_ptr(), C.byref(n_part), st)
            _lib.call("fn_colsum_f32", part.data_ptr(), n_part.value, J * FN_D, g_A.data_ptr(), A.shape[1], ctx.off, st)
        return g_feat, g_A, None, None


def row_dots_sorted(feat, A, off: int, level: Level):
    return _RowDotsSorted.apply(feat, A, off, level)


# ======================================================================================
# node projections on the fp32 matrix cores
# ======================================================================================
class _Linear128(torch.autograd.Function):
    """y = x @ W.T + b with W [128, K] (nn.Linear(K, 128)); K <= 168.  Input gradient only for K == 128."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight, bias = _f32c(x, "x"), _f32c(weight, "weight"), _f32c(bias, "bias")
        M, K = x.shape
        if weight.shape != (FN_D, K) or bias.shape != (FN_D,):
            raise ValueError("linear128: weight must be [128, K], bias [128]")
        st = _stream_ptr(x.device)
        bt = torch.empty((K, FN_D), dtype=torch.float32, device=x.device)
        _lib.call("fn_transpose_w_f32", weight.data_ptr(), K, bt.data_ptr(), st)
        y = torch.empty((M, FN_D), dtype=torch.float32, device=x.device)
        _lib.call("fn_linear128_f32", x.data_ptr(), K, bt.data_ptr(), bias.data_ptr(), y.data_ptr(), M, None, st)
        ctx.save_for_backward(x, weight)
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _f32c(g, "g")
        M, K = x.shape
        st = _stream_ptr(x.device)
        gx = None
        if ctx.x_needs_grad:
            if K != FN_D:
                raise NotImplementedError("linear128: input gradient is implemented for K == 128 (layers >= 1)")
            gx = torch.empty_like(x)
            _lib.call("fn_linear128_f32", g.data_ptr(), FN_D, weight.data_ptr(), None, gx.data_ptr(), M, None, st)
        ws = torch.empty(_lib.load().fn_linear128_wgrad_ws(M, K), dtype=torch.float32, device=x.device)
        gw = torch.empty_like(weight)
        gb = torch.empty(FN_D, dtype=torch.float32, device=x.device)
        _lib.call("fn_linear128_wgrad_f32", g.data_ptr(), x.data_ptr(), K, M, ws.data_ptr(), gw.data_ptr(), gb.data_ptr(), st)
        return gx, gw, gb


def linear128(x, weight, bias):
    return _Linear128.apply(x, weight, bias)


# ======================================================================================
# segment sum (scatter_add along dim 0) and its gather backward
# ======================================================================================
class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, seg: Segments, keep_plan):
        src = _f32c(src, "src")
        if src.shape[0] != seg.n_items:
            raise ValueError(f"src has {src.shape[0]} rows, index has {seg.n_items}")
        width = 1
        for extent in src.shape[1:]:
            width *= int(extent)
        alloc = torch.zeros if seg.n_items == 0 else torch.empty
        out = alloc((seg.n_seg,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
        if seg.n_seg and width and seg.n_items:
            _lib.call("fn_segment_sum_f32", src.data_ptr(), width, seg.rowptr.data_ptr(), seg.perm.data_ptr(),
                      seg.pos_base, out.data_ptr(), seg.n_seg, width, seg.n_items, _stream_ptr(src.device))
        ctx.seg, ctx.width, ctx.keep = seg, width, keep_plan
        return out

    @staticmethod
    def backward(ctx, g_out):
        seg, width = ctx.seg, ctx.width
        g_out = _f32c(g_out, "g_out")
        g_src = torch.empty((seg.n_items,) + tuple(g_out.shape[1:]), dtype=torch.float32, device=g_out.device)
        if seg.n_items and width:
            _lib.call("fn_gather_rows_f32", g_out.data_ptr(), seg.index.data_ptr(), g_src.data_ptr(), seg.n_items, width,
                      _stream_ptr(g_out.device))
        return g_src, None, None


def segment_sum(src, seg: Segments, plan=None):
    return _SegmentSum.apply(src, seg, plan)


def _seg_struct(seg: Segments):
    return _lib.SegPlan(seg.rowptr.data_ptr(), seg.perm.data_ptr(), seg.index.data_ptr(), seg.n_seg, seg.n_items, seg.pos_base, 0)


class _PoolCat(torch.autograd.Function):
    """cat(scatter_add(x_atoms, batch), scatter_add(x_frags, frag_batch), dim=1) -> [B, 256] (gat2.py:820-823): one
    launch forward, one launch backward (two segment sums + cat, and two strided copies + two gathers, otherwise)."""

    @staticmethod
    def forward(ctx, x_atoms, x_frags, plan):
        x_atoms, x_frags = _f32c(x_atoms, "x_atoms"), _f32c(x_frags, "x_frags")
        sa, sf = plan.segs["mol_atoms"], plan.segs["mol_frags"]
        if x_atoms.shape[0] != sa.n_items or x_frags.shape[0] != sf.n_items or x_atoms.shape[1] != FN_D or x_frags.shape[1] != FN_D:
            raise ValueError("pool_cat: feature tables do not match the batch / frag_batch index")
        out = torch.empty((sa.n_seg, 2 * FN_D), dtype=torch.float32, device=x_atoms.device)
        a, f = _seg_struct(sa), _seg_struct(sf)
        _lib.call("fn_pool_cat_f32", x_atoms.data_ptr(), x_frags.data_ptr(), C.byref(a), C.byref(f), out.data_ptr(),
                  _stream_ptr(x_atoms.device))
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        sa, sf = plan.segs["mol_atoms"], plan.segs["mol_frags"]
        g = _f32c(g, "g")
        g_atoms = torch.empty((sa.n_items, FN_D), dtype=torch.float32, device=g.device)
        g_frags = torch.empty((sf.n_items, FN_D), dtype=torch.float32, device=g.device)
        _lib.call("fn_pool_cat_bwd_f32", g.data_ptr(), sa.index.data_ptr(), sf.index.data_ptr(), g_atoms.data_ptr(),
                  g_frags.data_ptr(), sa.n_items, sf.n_items, _stream_ptr(g.device))
        return g_atoms, g_frags, None


def pool_cat(x_atoms, x_frags, plan):
    return _PoolCat.apply(x_atoms, x_frags, plan)


class _MaskedMSE(torch.autograd.Function):
    """sum_i w_i |out_i - y_i|^2 / (sum_i w_i * T): loss and d loss / d out from one single-block kernel."""

    @staticmethod
    def forward(ctx, out, y, w):
        B = w.shape[0]
        out2, y2 = _f32c(out, "out").reshape(B, -1), _f32c(y, "y").reshape(B, -1)
        if out2.shape != y2.shape:
            raise ValueError(f"masked_mse: prediction {tuple(out.shape)} vs target {tuple(y.shape)}")
        w = _f32c(w, "w")
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        g = torch.empty_like(out2)
        _lib.call("fn_masked_mse_f32", out2.data_ptr(), y2.data_ptr(), w.data_ptr(), B, out2.shape[1], loss.data_ptr(),
                  g.data_ptr(), _stream_ptr(out.device))
        ctx.save_for_backward(g)
        ctx.shape = out.shape
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (g,) = ctx.saved_tensors
        unit = _UNIT_GRAD.get(g.device)
        if unit is not None and g_loss.data_ptr() == unit.data_ptr():      # d loss / d loss = 1: nothing to multiply
            return g.reshape(ctx.shape), None, None
        return (g * g_loss).reshape(ctx.shape), None, None


def masked_mse(out, y, w):
    return _MaskedMSE.apply(out, y, w)


class _MaskedBCE(torch.autograd.Function):
    """compute_bce_loss (train/utils.py:297-304) over a padded batch: loss and gradient from one single-block kernel."""

    @staticmethod
    def forward(ctx, out, y, w):
        B = w.shape[0]
        out2, y2 = _f32c(out, "out").reshape(B, -1), _f32c(y, "y").reshape(B, -1)
        if out2.shape != y2.shape:
            raise ValueError(f"masked_bce: prediction {tuple(out.shape)} vs target {tuple(y.shape)}")
        w = _f32c(w, "w")
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        g = torch.empty_like(out2)
        _lib.call("fn_masked_bce_f32", out2.data_ptr(), y2.data_ptr(), w.data_ptr(), B, out2.shape[1], loss.data_ptr(),
                  g.data_ptr(), _stream_ptr(out.device))
        ctx.save_for_backward(g)
        ctx.shape = out.shape
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (g,) = ctx.saved_tensors
        unit = _UNIT_GRAD.get(g.device)
        if unit is not None and g_loss.data_ptr() == unit.data_ptr():
            return g.reshape(ctx.shape), None, None
        return (g * g_loss).reshape(ctx.shape), None, None


def masked_bce(out, y, w):
    return _MaskedBCE.apply(out, y, w)


class _MaskedMSEMulti(torch.autograd.Function):
    """sum_k coef_k * masked_mse(out_k, y_k, w_k), coef_k = c_k * (scale[idx_k] if idx_k >= 0 else 1): loss and all
    gradients in two launches (fn_masked_mse_multi_f32)."""

    @staticmethod
    def forward(ctx, spec, scale, *tensors):
        n = len(spec)
        dev = tensors[0].device
        tasks = (_lib.MseTask * n)()
        keep, grads, shapes = [], [], []
        for k, (c, idx) in enumerate(spec):
            out, y, w = tensors[3 * k: 3 * k + 3]
            B = w.shape[0]
            out2, y2, w = _f32c(out, "out").reshape(B, -1), _f32c(y, "y").reshape(B, -1), _f32c(w, "w")
            if out2.shape != y2.shape:
                raise ValueError(f"masked_mse_multi: prediction {tuple(out.shape)} vs target {tuple(y.shape)}")
            g = torch.empty_like(out2)
            tasks[k] = _lib.MseTask(out2.data_ptr(), y2.data_ptr(), w.data_ptr(), g.data_ptr(), B, out2.shape[1], int(idx), float(c), 0.0)
            keep += [out2, y2, w]
            grads.append(g)
            shapes.append(out.shape)
        scale = None if scale is None else _f32c(scale, "scale")
        ws = torch.empty(_lib.load().fn_masked_mse_multi_ws(n), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.call("fn_masked_mse_multi_f32", tasks, n, _ptr(scale), ws.data_ptr(), loss.data_ptr(), _stream_ptr(dev))
        ctx.save_for_backward(*grads)
        ctx.shapes = shapes
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        unit = _UNIT_GRAD.get(g_loss.device)
        one = unit is not None and g_loss.data_ptr() == unit.data_ptr()
        out = [None, None]
        for g, shape in zip(ctx.saved_tensors, ctx.shapes):
            out += [(g if one else g * g_loss).reshape(shape), None, None]
        return tuple(out)


def masked_mse_multi(spec, scale, *triples):
    """``spec`` = [(c_k, scale_index_k or -1), ...]; ``triples`` = out_1, y_1, w_1, out_2, ...; ``scale`` a device vector."""
    return _MaskedMSEMulti.apply(tuple(spec), scale, *triples)


_UNIT_GRAD = {}


def unit_grad(device):
    """A persistent scalar 1.0 to seed ``loss.backward(gradient=...)`` with: autograd then launches no fill kernel, and
    ``masked_mse`` recognises it and skips the multiply (two launches of a captured step)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device not in _UNIT_GRAD:
        _UNIT_GRAD[device] = torch.ones((), dtype=torch.float32, device=device)
    return _UNIT_GRAD[device]


class _GatherRows(torch.autograd.Function):
    """index_select(table, 0, index) with a segment-sum backward (needs the CSR of ``index``)."""

    @staticmethod
    def forward(ctx, table, seg: Segments, keep_plan):
        table = _f32c(table, "table")
        width = table.shape[1]
        out = torch.empty((seg.n_items, width), dtype=torch.float32, device=table.device)
        if seg.n_items:
            _lib.call("fn_gather_rows_f32", table.data_ptr(), seg.index.data_ptr(), out.data_ptr(), seg.n_items, width,
                      _stream_ptr(table.device))
        ctx.seg, ctx.keep, ctx.rows = seg, keep_plan, table.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        seg = ctx.seg
        g = _f32c(g, "g")
        width = g.shape[1]
        out = torch.empty((ctx.rows, width), dtype=torch.float32, device=g.device)
        _lib.call("fn_segment_sum_f32", g.data_ptr(), width, seg.rowptr.data_ptr(), seg.perm.data_ptr(), seg.pos_base,
                  out.data_ptr(), seg.n_seg, width, seg.n_items, _stream_ptr(g.device))
        return out, None, None


# ======================================================================================
# torch-scatter operator surface
# ======================================================================================
def _dim0_only(dim, src):
    if dim not in (0, -src.dim()):
        raise NotImplementedError("fragnet_amd implements scatter along dim 0 (all the reference's call sites)")


def _n_seg(index: torch.Tensor, dim_size) -> int:
    if dim_size is not None:
        return int(dim_size)
    if index.numel() == 0:
        return 0
    return int(index.max()) + 1      # synchronises, exactly like torch-scatter does


def scatter_add(src, index, dim: int = 0, out=None, dim_size=None):
    """Drop-in for torch_scatter.scatter_add(src, index, dim=0): 1-D int64 ``index`` over the rows of ``src``."""
    _dim0_only(dim, src)
    if out is not None:
        raise NotImplementedError("out= is not used on the reference path")
    if index.dim() != 1 or index.numel() != src.shape[0]:
        raise RuntimeError("index must be 1-D with one entry per row of src")
    plan = GraphPlan.segments_only(index, _n_seg(index, dim_size))
    return segment_sum(src, plan.segs["s"], plan)


class _SegmentSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, seg: Segments, keep_plan):
        src = _f32c(src, "src")
        width = 1
        for extent in src.shape[1:]:
            width *= int(extent)
        probs = torch.empty_like(src)
        if src.numel():
            _lib.call("fn_segment_softmax_f32", src.data_ptr(), seg.rowptr.data_ptr(), seg.perm.data_ptr(), seg.pos_base,
                      probs.data_ptr(), seg.n_seg, width, _stream_ptr(src.device))
        ctx.seg, ctx.width, ctx.keep = seg, width, keep_plan
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, g):
        (probs,) = ctx.saved_tensors
        seg = ctx.seg
        g = _f32c(g, "g")
        out = torch.empty_like(probs)
        if probs.numel():
            _lib.call("fn_segment_softmax_bwd_f32", probs.data_ptr(), g.data_ptr(), seg.rowptr.data_ptr(),
                      seg.perm.data_ptr(), seg.pos_base, out.data_ptr(), seg.n_seg, ctx.width, _stream_ptr(g.device))
        return out, None, None


def scatter_softmax(src, index, dim: int = 0, dim_size=None):
    """Drop-in for torch_scatter.scatter_softmax(src, index, dim=0)."""
    _dim0_only(dim, src)
    if index.dim() != 1 or index.numel() != src.shape[0]:
        raise RuntimeError("index must be 1-D with one entry per row of src")
    plan = GraphPlan.segments_only(index, _n_seg(index, dim_size))
    return _SegmentSoftmax.apply(src, plan.segs["s"], plan)


# ======================================================================================
# act(dropout(x)) epilogue
# ======================================================================================
class PhiloxStream:
    """Counter-based dropout stream: (seed, running offset). One Philox block = 4 floats."""

    def __init__(self, seed: Optional[int] = None, rank: int = 0):
        self.seed = None if seed is None else (int(seed) & 0xFFFFFFFFFFFFFFFF)
        self.rank = rank
        self.offset = 0
        self.dev = None          # optional device-resident counter added to every offset when the kernels run

    def use_device_counter(self, device):
        """hipGraph mode: offsets handed out while a step is captured are baked into the graph, so the per-replay
        part of the counter lives in device memory and ``advance_device`` (captured too) moves it on."""
        if self.dev is None or self.dev.device != torch.device(device):
            self.dev = torch.zeros(1, dtype=torch.int64, device=device)
        return self.dev

    def advance_device(self, blocks: int):
        self.dev += int(blocks)

    def dev_ptr(self):
        return None if self.dev is None else self.dev.data_ptr()

    def take(self, numel: int):
        if self.seed is None:
            self.seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self.rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        off = self.offset
        self.offset += (numel + 3) // 4
        return self.seed, off


class _DropoutAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p: float, relu: bool, seed: int, offset: int, dev):
        x = _f32c(x, "x")
        y = torch.empty_like(x)
        _lib.call("fn_dropout_act_f32", x.data_ptr(), y.data_ptr(), x.numel(), float(p), seed, offset, _ptr(dev), int(relu),
                  _stream_ptr(x.device))
        ctx.args = (float(p), bool(relu), seed, offset, dev)
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        p, relu, seed, offset, dev = ctx.args
        y = ctx.saved_tensors[0] if relu else None
        g = _f32c(g, "g")
        gx = torch.empty_like(g)
        _lib.call("fn_dropout_act_bwd_f32", g.data_ptr(), _ptr(y), gx.data_ptr(), g.numel(), p, seed, offset, _ptr(dev),
                  int(relu), _stream_ptr(g.device))
        return gx, None, None, None, None, None


def dropout_act(x, p: float, training: bool, relu: bool, rng: PhiloxStream):
    """relu(dropout(x)) (gat2.py:414-418) or plain dropout (gat2.py:396-397) as one elementwise kernel."""
    p_eff = float(p) if training else 0.0
    if p_eff == 0.0 and not relu:
        return x
    seed, off = rng.take(x.numel()) if p_eff > 0.0 else (0, 0)
    return _DropoutAct.apply(x, p_eff, relu, seed, off, rng.dev if p_eff > 0.0 else None)


# ======================================================================================
# prediction head: Linear -> relu(dropout(.)) stack with a hand-written backward
# ======================================================================================
SMALL_LINEAR_MAX = 16        # FN_SMALL_LINEAR_MAX


DENSE_MAX_ROWS = 4096        # FN_DENSE_MAX_ROWS
DENSE_HEAD = True            # hidden layers through fn_dense_fwd/bwd_f32 (False: library GEMMs + the element-wise kernels)


def _dense_ok(rows: int, W) -> bool:
    return DENSE_HEAD and rows <= DENSE_MAX_ROWS and W.shape[0] % 4 == 0 and W.shape[1] % 4 == 0 and W.is_contiguous()


def _scratch(n_floats: int, device):
    return torch.empty(n_floats, dtype=torch.float32, device=device) if n_floats else None


class _MLPHead(torch.autograd.Function):
    """FTHead1-5's predictor stack (gat2.py:631-637, 745-751) as one autograd node.

    On molecule-sized inputs (``_dense_ok``) every hidden layer is ONE launch each way (csrc/dense_head.inc, fp32 matrix
    cores): forward = product + bias + relu(dropout(.)); backward = weight gradient + bias gradient + input gradient, the
    latter already through the backward of the layer below's relu(dropout(.)) (the saved output encodes the mask, so no
    Philox replay).  The last Linear (n_classes outputs) is one launch each way, too.  Taller inputs (the pretrain towers
    run on every atom / bond) keep library GEMMs (addmm / mm) with the element-wise work fused around them
    (``fn_dropout_act_f32`` in place, ``fn_gate_colsum_f32``).  ``draws`` = the (seed, offset) of each hidden layer's mask,
    taken from the model's Philox stream in the same order as the unfused path, so both paths produce identical numbers.
    ``live``: input rows >= live are padding (static-shape batches): they are not computed, their outputs and input
    gradients are 0.
    """

    @staticmethod
    def forward(ctx, x, p: float, draws, dev, live, loss, *params):
        n = len(params) // 2
        st = _stream_ptr(x.device)
        x = _f32c(x, "x")
        M = x.shape[0]
        live = M if live is None else max(0, min(int(live), M))
        h = x[:live]
        acts = [h]
        dense = all(_dense_ok(live, params[2 * i]) for i in range(n - 1)) and params[-2].shape[0] <= SMALL_LINEAR_MAX \
            and params[-2].shape[1] % 4 == 0
        for i in range(n - 1):
            W, b = params[2 * i], params[2 * i + 1]
            seed, off = draws[i]
            if dense:
                y = torch.empty((live, W.shape[0]), dtype=torch.float32, device=h.device)
                act = _lib.ActEpilogue(y.data_ptr(), float(p), 1, seed, off, _ptr(dev) if p > 0.0 else None)
                _lib.call("fn_dense_fwd_f32", h.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), live, W.shape[1], W.shape[0],
                          C.byref(act), st)
            else:
                y = torch.addmm(b, h, W.t())
                _lib.call("fn_dropout_act_f32", y.data_ptr(), y.data_ptr(), y.numel(), float(p), seed, off,
                          _ptr(dev) if p > 0.0 else None, 1, st)
            h = y
            acts.append(h)
        W, b = params[-2], params[-1]
        C_out, K = W.shape
        ctx.p, ctx.dense, ctx.rows = float(p), dense, (M, live)
        ctx.params, ctx.slots = params, [grad_slot(q) for q in params]
        ctx.fused_loss = False
        if loss is not None and dense and n > 1 and live > 0 and K <= _lib.SMALL_LINEAR_LOSS_MAX_K and loss[2].shape[0] == M \
                and loss[1].numel() == M * C_out:
            # last Linear + loss + its input gradient in one launch; dW / db / the loss value ride in the backward's first launch
            kind, tgt, row_w = loss
            tgt, row_w = _f32c(tgt, "y"), _f32c(row_w, "w")
            out = torch.empty((M, C_out), dtype=torch.float32, device=h.device)
            g = torch.empty((live, C_out), dtype=torch.float32, device=h.device)
            gz = torch.empty_like(h)
            parts = torch.empty(_lib.load().fn_small_linear_loss_ws(M), dtype=torch.float32, device=h.device)
            loss_t = torch.empty((), dtype=torch.float32, device=h.device)
            scale = 1.0 / (1.0 - p) if 0.0 < p < 1.0 else 1.0
            _lib.call("fn_small_linear_loss_f32", h.data_ptr(), _f32c(W, "W").data_ptr(), b.data_ptr(), tgt.data_ptr(), row_w.data_ptr(),
                      int(kind), out.data_ptr(), g.data_ptr(), gz.data_ptr(), scale, parts.data_ptr(), live, K, C_out, M, st)
            ctx.fused_loss = True
            ctx.save_for_backward(*acts, *params[0::2], g, gz, parts, loss_t)
            ctx.mark_non_differentiable(out)
            ctx.set_materialize_grads(False)            # no zero-filled gradient for the predictions (a fill launch per step)
            return out, loss_t
        if C_out <= SMALL_LINEAR_MAX and K % 4 == 0:
            out = torch.empty((M, C_out), dtype=torch.float32, device=h.device)        # the kernel zeroes the padding rows
            _lib.call("fn_small_linear_f32", h.data_ptr(), _f32c(W, "W").data_ptr(), b.data_ptr(), out.data_ptr(), live, K, C_out, M, st)
        else:
            out = torch.addmm(b, h, W.t())
            if live < M:
                out = torch.cat([out, out.new_zeros((M - live, C_out))])
        ctx.save_for_backward(*acts, *params[0::2])
        return (out, None) if loss is not None else out

    @staticmethod
    def backward(ctx, g, g_loss=None):
        saved = ctx.saved_tensors
        fused = None
        if ctx.fused_loss:
            saved, fused = saved[:-4], saved[-4:]
        n = len(saved) // 2
        acts, Ws = saved[:n], saved[n:]
        M, live = ctx.rows
        if fused is None:
            g = _f32c(g, "g")[:live]
        st = _stream_ptr(acts[0].device)
        grads = [None] * (2 * n)
        W, h = Ws[-1], acts[-1]
        C_out, K = W.shape
        P, slots = ctx.params, ctx.slots
        dense, need_x = ctx.dense, ctx.needs_input_grad[0]
        # gate scale of the fused backward of relu(dropout(.)): the kernels read 0 as "no gate", so p >= 1 (every element dropped:
        # the saved outputs are all 0 and the gate lets nothing through) passes a positive value, not 1 / (1 - p) = inf or 0
        scale = 1.0 / (1.0 - ctx.p) if 0.0 < ctx.p < 1.0 else 1.0
        dW, db = grad_buffer(P[-2], slots[-2]), grad_buffer(P[-1], slots[-1])

        def input_grad(like, last, zeroed_by_kernel=False):
            """buffer of d loss / d (input of a layer); the head's own input gradient has all M rows (padding rows 0)"""
            if not last or live == M:
                return torch.empty_like(like)
            full = torch.empty((M, like.shape[1]), dtype=torch.float32, device=like.device)
            if not zeroed_by_kernel:
                full[live:].zero_()
            return full

        tail = None
        if fused is not None:
            # the forward's fused launch left g = d loss / d out and gz (for d loss / d loss = 1); the sums over rows ride below
            g, gz, parts, loss_t = fused
            unit = _UNIT_GRAD.get(g.device)
            if g_loss is not None and not (unit is not None and g_loss.data_ptr() == unit.data_ptr()):
                g, gz = g * g_loss, gz * g_loss
            tail = _lib.SmallDw(g.data_ptr(), h.data_ptr(), dW.data_ptr(), db.data_ptr(), parts.data_ptr(), loss_t.data_ptr(),
                                parts.numel(), live, K, C_out)
            ctx.tail_keep = (g, gz)
        elif C_out <= SMALL_LINEAR_MAX and K % 4 == 0:
            gz = input_grad(h, n == 1) if (n > 1 or need_x) else torch.empty_like(h)
            ws = _scratch(_lib.load().fn_small_linear_bwd_ws(live, K, C_out), g.device)
            # dense path: gz leaves gated by h > 0 (the backward of the top hidden layer's relu(dropout(.)))
            _lib.call("fn_small_linear_bwd_f32", g.data_ptr(), h.data_ptr(), W.data_ptr(), gz.data_ptr(), dW.data_ptr(), db.data_ptr(),
                      live, K, C_out, scale if (dense and n > 1) else 0.0, _ptr(ws), st)
        else:
            gz = g @ W
            torch.mm(g.t(), h, out=dW)
            torch.sum(g, 0, out=db)
            if n == 1 and live < M:
                gz = torch.cat([gz, gz.new_zeros((M - live, K))])
        grads[-2], grads[-1] = dW, db
        # (running the weight-gradient GEMMs on a side stream beside the gate -> input-gradient chain was measured: a
        # two-branch hipGraph replays 9 % slower on ROCm 7.2 than the serial one, HISTORY.md section 4)
        for i in range(n - 2, -1, -1):
            z, h_in, W = acts[i + 1], acts[i], Ws[i]
            dW, db = grad_buffer(P[2 * i], slots[2 * i]), grad_buffer(P[2 * i + 1], slots[2 * i + 1])
            grads[2 * i], grads[2 * i + 1] = dW, db
            need_gx = i > 0 or need_x
            if dense:                                   # gz is d loss / d (pre-activation) already: one launch for the layer
                gx = input_grad(h_in, i == 0, zeroed_by_kernel=True) if need_gx else None
                if tail is not None:
                    _lib.call("fn_dense_bwd_tail_f32", gz.data_ptr(), h_in.data_ptr(), W.data_ptr(), _ptr(gx), scale if i > 0 else 0.0,
                              dW.data_ptr(), db.data_ptr(), live, W.shape[1], W.shape[0], M if i == 0 else live, C.byref(tail), st)
                    tail = None
                else:
                    _lib.call("fn_dense_bwd_f32", gz.data_ptr(), h_in.data_ptr(), W.data_ptr(), _ptr(gx), scale if i > 0 else 0.0, dW.data_ptr(),
                              db.data_ptr(), live, W.shape[1], W.shape[0], M if i == 0 else live, st)
                gz = gx
                continue
            gy = torch.empty_like(z)
            ws = _scratch(_lib.load().fn_gate_colsum_ws(z.shape[0], z.shape[1]), z.device)
            _lib.call("fn_gate_colsum_f32", gz.data_ptr(), z.data_ptr(), gy.data_ptr(), db.data_ptr(), z.shape[0], z.shape[1], scale,
                      _ptr(ws), st)
            torch.mm(gy.t(), h_in, out=dW)
            if need_gx:
                gz = gy @ W
                if i == 0 and live < M:
                    gz = torch.cat([gz, gz.new_zeros((M - live, gz.shape[1]))])
        return (gz if need_x else None, None, None, None, None, None, *grads)


FUSED_HEAD_LOSS = os.environ.get("FRAGNET_FUSED_HEAD_LOSS", "1") != "0"       # False: last Linear, loss and the Linear's backward as three launches (A/B and tests)


def mlp_head(x, linears, p: float, training: bool, rng: "PhiloxStream", live=None, loss=None):
    """Runs ``linears`` (nn.Linear modules; relu(dropout(.)) after all but the last) through ``_MLPHead``.

    ``loss = (kind, target, row_weights)`` (kind: ``_lib.LOSS_MSE`` / ``_lib.LOSS_BCE``): the caller is a training step that will call
    ``backward`` on the loss with gradient 1 right away.  Returns ``(out, loss)``; ``loss`` is None when the fused launch does not
    apply (the caller then computes it from ``out``), else a scalar whose VALUE is complete once backward has run (the sum of the
    partials rides in the backward's first launch) and ``out`` carries no gradient."""
    p_eff = float(p) if training else 0.0
    draws = []
    for lin in linears[:-1]:
        draws.append(rng.take(x.shape[0] * lin.out_features) if p_eff > 0.0 else (0, 0))
    params = []
    for lin in linears:
        if lin.bias is None or (lin.out_features % 4 != 0 and lin is not linears[-1]):
            raise ValueError("mlp_head: Linear layers need a bias, hidden ones an output width that is a multiple of 4")
        params += [lin.weight, lin.bias]
    if loss is not None and not (FUSED_HEAD_LOSS and training and x.requires_grad and linears[-1].out_features <= SMALL_LINEAR_MAX
                                 and linears[-1].in_features % 4 == 0):
        return _MLPHead.apply(x, p_eff, tuple(draws), rng.dev if p_eff > 0.0 else None, live, None, *params), None
    return _MLPHead.apply(x, p_eff, tuple(draws), rng.dev if p_eff > 0.0 else None, live, loss, *params)


# ======================================================================================
# PretrainTask's tall towers: Linear(128 -> 64) -> ReLU -> Linear(64 -> 32) -> ReLU -> Linear(32 -> 1) on every atom / bond
# ======================================================================================
TOWER_SHAPES = ((64, 128), (32, 64), (1, 32))


def tower_ok(x, linears) -> bool:
    """The fused tower kernels take exactly the reference's `PretrainTask(128, 1)` shape (pretrain_heads.py:33-58) on GPU rows."""
    return x.is_cuda and x.dim() == 2 and x.shape[1] == FN_D and len(linears) == 3 and \
        all(tuple(lin.weight.shape) == shp and lin.bias is not None for lin, shp in zip(linears, TOWER_SHAPES))


class _Towers(torch.autograd.Function):
    """Up to FN_MAX_TOWERS towers as ONE launch each way (csrc/tower.hip): forward keeps h1 / h2, backward writes the input
    gradients and reduces the weight-gradient partials straight into the parameters' gradient buffers (FlatAdam slots)."""

    @staticmethod
    def forward(ctx, n, *args):
        xs, params = [_f32c(x, "x") for x in args[:n]], args[n:]
        dev = xs[0].device
        tw = (_lib.Tower * n)()
        keep, outs = [], []
        for i, x in enumerate(xs):
            M = x.shape[0]
            w1, b1, w2, b2, w3, b3 = (_f32c(q, "tower parameter") for q in params[6 * i: 6 * i + 6])
            h1 = torch.empty((M, 64), dtype=torch.float32, device=dev)
            h2 = torch.empty((M, 32), dtype=torch.float32, device=dev)
            out = torch.empty((M, 1), dtype=torch.float32, device=dev)
            t = tw[i]
            t.x, t.w1, t.b1, t.w2, t.b2, t.w3, t.b3 = (q.data_ptr() for q in (x, w1, b1, w2, b2, w3, b3))
            t.h1, t.h2, t.out, t.M = h1.data_ptr(), h2.data_ptr(), out.data_ptr(), M
            keep += [x, h1, h2, w1, b1, w2, b2, w3, b3]
            outs.append(out)
        _lib.call("fn_tower_fwd_f32", tw, n, _stream_ptr(dev))
        ctx.n = n
        ctx.params, ctx.slots = params, [grad_slot(q) for q in params]
        ctx.save_for_backward(*keep)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, saved = ctx.n, ctx.saved_tensors
        dev = saved[0].device
        tw = (_lib.Tower * n)()
        gxs, grads, keep = [], [], []
        for i in range(n):
            x, h1, h2, w1, b1, w2, b2, w3, b3 = saved[9 * i: 9 * i + 9]
            M = x.shape[0]
            g = gs[i]
            g = torch.zeros((M, 1), dtype=torch.float32, device=dev) if g is None else _f32c(g, "g")
            gx = torch.empty_like(x) if ctx.needs_input_grad[1 + i] else None
            pg = [grad_buffer(ctx.params[6 * i + k], ctx.slots[6 * i + k]) for k in range(6)]
            t = tw[i]
            t.x, t.w1, t.b1, t.w2, t.b2, t.w3, t.b3 = (q.data_ptr() for q in (x, w1, b1, w2, b2, w3, b3))
            t.h1, t.h2, t.M, t.g_out, t.g_x = h1.data_ptr(), h2.data_ptr(), M, g.data_ptr(), _ptr(gx)
            t.g_w1, t.g_b1, t.g_w2, t.g_b2, t.g_w3, t.g_b3 = (q.data_ptr() for q in pg)
            gxs.append(gx)
            grads += pg
            keep.append(g)
        ws = torch.empty(_lib.load().fn_tower_bwd_ws(tw, n), dtype=torch.float32, device=dev)
        _lib.call("fn_tower_bwd_f32", tw, n, ws.data_ptr(), _stream_ptr(dev))
        return (None, *gxs, *grads)


def towers(pairs):
    """``pairs`` = [(x [M,128], [Linear(128,64), Linear(64,32), Linear(32,1)]), ...] -> list of [M,1] outputs; every pair must pass
    ``tower_ok``."""
    if not 1 <= len(pairs) <= _lib.FN_MAX_TOWERS:
        raise ValueError("towers: 1..FN_MAX_TOWERS towers per call")
    params = [q for _, lins in pairs for lin in lins for q in (lin.weight, lin.bias)]
    return list(_Towers.apply(len(pairs), *[x for x, _ in pairs], *params))


# ======================================================================================
# pretrain bond-length head input: cat(x[src], x[dst], e_attr)
# ======================================================================================
class _EdgeConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, e_attr, edge_index, plan: GraphPlan):
        x, e_attr = _f32c(x, "x"), _f32c(e_attr, "e_attr")
        E = edge_index.shape[1]
        edge_index = edge_index.contiguous()
        out = torch.empty((E, 3 * FN_D), dtype=torch.float32, device=x.device)
        _lib.call("fn_edge_concat_f32", x.data_ptr(), e_attr.data_ptr(), edge_index.data_ptr(), out.data_ptr(), E,
                  _stream_ptr(x.device))
        ctx.plan, ctx.n = plan, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        plan, n = ctx.plan, ctx.n
        g = _f32c(g, "g")
        st = _stream_ptr(g.device)
        gx = torch.empty((n, FN_D), dtype=torch.float32, device=g.device)
        tmp = torch.empty_like(gx)
        s, d = plan.segs["edge_src"], plan.segs["edge_dst"]
        ld = 3 * FN_D
        _lib.call("fn_segment_sum_f32", g.data_ptr(), ld, s.rowptr.data_ptr(), s.perm.data_ptr(), s.pos_base,
                  gx.data_ptr(), n, FN_D, s.n_items, st)
        _lib.call("fn_segment_sum_f32", g.data_ptr() + 4 * FN_D, ld, d.rowptr.data_ptr(), d.perm.data_ptr(), d.pos_base,
                  tmp.data_ptr(), n, FN_D, d.n_items, st)
        gx += tmp
        return gx, g[:, 2 * FN_D:].contiguous(), None, None


def edge_concat(x, e_attr, edge_index, plan: GraphPlan):
    return _EdgeConcat.apply(x, e_attr, edge_index, plan)


# ======================================================================================
# bond-graph topology from edge_index (dataset side)
# ======================================================================================
def bond_graph(edge_index: torch.Tensor, atom_batch: torch.Tensor, n_mols: int, fragments: bool = False) -> torch.Tensor:
    """``edge_index_bonds_graph`` [2, Eb] (int64, global bond ids) of a collated batch from its ``edge_index`` [2, E] and
    ``batch`` vector -- the reference's get_bond_pair_bond_graph + one-bond-fragment rule (dataset/data.py:116-127,
    157-182) on the GPU, in the reference's order, so a dataset can ship without its largest tensor (the per-pair
    cos(theta) attribute still has to be stored: it needs 3-D coordinates).  ``fragments=True``: ``edge_index_fbonds`` from
    ``frag_index`` and ``frag_batch`` (get_bond_pair_fbond_graph, data.py:131-154).  One host sync to learn Eb."""
    if not edge_index.is_cuda or edge_index.dtype != torch.int64 or atom_batch.dtype != torch.int64:
        raise _lib.FragnetHipError("bond_graph: int64 GPU tensors expected (there is no CPU fallback)")
    edge_index, atom_batch = edge_index.contiguous(), atom_batch.contiguous()
    E, N, dev = edge_index.shape[1], atom_batch.shape[0], edge_index.device
    st, mode = _stream_ptr(dev), int(bool(fragments))
    ws = torch.empty(_lib.load().fn_bond_graph_ws(E, n_mols), dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.call("fn_bond_graph_count", edge_index.data_ptr(), atom_batch.data_ptr(), E, N, n_mols, mode, ws.data_ptr(), total.data_ptr(), st)
    n = int(total.item())
    out = torch.empty((2, n), dtype=torch.int64, device=dev)
    _lib.call("fn_bond_graph_fill", edge_index.data_ptr(), atom_batch.data_ptr(), E, N, n_mols, mode, ws.data_ptr(), out.data_ptr(), n, st)
    return out
