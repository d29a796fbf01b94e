"""ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product path in fragnet_amd/).

CPU restatement of the three third-party functions the reference hot path calls.  Their
source is NOT under /root/reference (un-vendored wheels):
  * torch-scatter (version unpinned by the reference: install_cpu.sh:5, README.md:44;
    the torch-2.4.0 wheel index carried 2.1.2) -- ``scatter_add`` (= ``scatter_sum``) and
    ``scatter_softmax``; call sites fragnet/model/gat/gat2.py:153,162,165,210,216,219,234,
    257,265,268,303,309,312,820,821 and fragnet/model/gat/pretrain_heads.py:93-94
  * torch_geometric==2.6.1 (requirements.txt:20) -- ``add_self_loops``; call site gat2.py:179

PARITY STATUS: *unpinned at this third-party boundary* -- the reference ships no tests,
golden vectors or known answers for these functions.  The semantics
below are the published ones (torch-scatter README / docs: "scatter_sum", "composite
softmax"; PyG docs: add_self_loops).  What IS pinned: the reference's own Python
(gat2.py, pretrain_heads.py, data.py) executed in the build container on top of stub
modules with these semantics, frozen as tests/golden/*.npz (see tests/golden/make_golden.py).
"""
from __future__ import annotations

import torch


def _out_rows(index: torch.Tensor, dim_size) -> int:
    if dim_size is not None:
        return int(dim_size)
    if index.numel() == 0:
        return 0
    return int(index.max()) + 1


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = 0, out=None, dim_size=None) -> torch.Tensor:
    """out[index[i], ...] += src[i, ...]; rows = dim_size or index.max()+1 (torch-scatter scatter_sum)."""
    if dim != 0:
        raise NotImplementedError("the reference hot path only scatters along dim 0")
    rows = _out_rows(index, dim_size)
    if out is None:
        out = torch.zeros((rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_add_(0, idx, src)


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim_size=None) -> torch.Tensor:
    rows = _out_rows(index, dim_size)
    out = torch.full((rows,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype, device=src.device)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_reduce_(0, idx, src, reduce="amax", include_self=True)


def scatter_softmax(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size=None) -> torch.Tensor:
    """exp(src - segmax) / segsum(exp(src - segmax)), per trailing column (torch-scatter composite)."""
    if dim != 0:
        raise NotImplementedError
    seg_max = scatter_max(src, index, dim_size)   # differentiable, as torch-scatter's scatter_max is
    shifted = src - seg_max.index_select(0, index)
    e = shifted.exp()
    seg_sum = scatter_add(e, index, dim_size=dim_size)
    return e / seg_sum.index_select(0, index)


def add_self_loops(edge_index: torch.Tensor):
    """Appends (i, i) for i in range(edge_index.max()+1) AFTER the real edges (PyG 2.6.1)."""
    n = int(edge_index.max()) + 1
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loops, loops])], dim=1), None
