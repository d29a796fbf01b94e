"""The attention backward as ONE source-owner pass (csrc/gat_bwd_one.inc).

Three kinds of checks: (1) one pass == the destination + source passes on every output (same inputs, independent kernels: a
self-consistency check between two HIP paths, heads 1 / 2 / 4 / 8 x edge classes x hubs); (2) DIRECTLY against the oracle's autograd
of the materialised level (oracle.fragnet_ref.gat_level_materialised = gat2.py:146-169) on the cases the golden fixtures do not
reach -- a source of out-degree far above 2 * LPH (the kernel's serial path), a destination hub, a single-edge level, a level with
nodes and no edges; (3) the forward's second output (out2, sigma) against its definition, and the DEFERRED form of the pass
(no g_s_dst read: dz at destination-order slots, g_s_dst and dL/da_dst from fn_gat_gsd_f32) against the non-deferred pass and the
oracle.  (The golden fixtures, the B = 512 graph step and the dropout-parity tests also run the one-pass path: it is the default.)
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fragnet_amd import _lib
    from fragnet_amd.build import build_lib
    build_lib()
    _lib.load()


def _graph(n, m, seed, hub):
    g = torch.Generator().manual_seed(seed)
    dst = torch.randint(0, n, (m,), generator=g)
    src = torch.randint(0, n, (m,), generator=g)
    if hub == "dst":
        dst[: m // 3] = 1
    elif hub == "src":                      # out-degree far above 2*LPH: the serial path of the source-owner pass
        src[: m // 3] = 2
    elif hub == "both":
        dst[: m // 4] = 1
        src[m // 4: m // 2] = 1
    dst[-1] = n - 1
    src[-1] = n - 1
    return dst, src, g


def _run_level(one_pass, heads, mode, dst, src, n, loops, g_seed, K=6):
    from fragnet_amd import ops
    from fragnet_amd.plan import GraphPlan
    d = 128 // heads
    m = dst.numel()
    g = torch.Generator().manual_seed(g_seed)
    h = torch.randn(n, 128, generator=g).to(DEV).requires_grad_(True)
    w_out = torch.randn(n, 128, generator=g).to(DEV)
    plan = GraphPlan([dict(kind="gat", name="l", dst=dst.to(DEV), src=src.to(DEV), n=n, n_loops=n if loops else 0)], DEV)
    lv = plan.levels["l"]
    old = ops.BWD_ONE_PASS
    ops.BWD_ONE_PASS = one_pass
    try:
        if mode == 2:
            att = (torch.randn(heads, 3 * d, generator=g) * 0.3).to(DEV).requires_grad_(True)
            x = torch.randn(m, K, generator=g).to(DEV)
            embW = (torch.randn(d, K, generator=g) * 0.5).to(DEV).requires_grad_(True)
            embb = (torch.randn(d, generator=g) * 0.5).to(DEV).requires_grad_(True)
            out = ops.gat_level(h, att, lv, heads, x_sorted=plan.sorted_attr("l", x), embW=embW, embb=embb)
            leaves = (h, att, embW, embb)
        else:
            att = (torch.randn(heads, 2 * d + 128, generator=g) * 0.3).to(DEV).requires_grad_(True)
            feat = torch.randn(m, 128, generator=g).to(DEV).requires_grad_(True)
            s_edge = ops.row_dots_sorted(feat, att, d, lv)
            out = ops.gat_level(h, att, lv, heads, s_sorted=s_edge)
            leaves = (h, att, feat)
        (out * w_out).sum().backward()
        torch.cuda.synchronize()
        plan.check()
    finally:
        ops.BWD_ONE_PASS = old
    return out.detach(), [t.grad.detach().clone() for t in leaves]


@pytest.mark.parametrize("heads", [1, 2, 4, 8])
@pytest.mark.parametrize("mode,K", [(0, 0), (2, 1), (2, 6)])
@pytest.mark.parametrize("hub", [None, "src", "both"])
def test_one_pass_equals_two_pass(heads, mode, K, hub):
    n, m = 301, 1700
    dst, src, _ = _graph(n, m, seed=heads * 100 + mode * 10 + K, hub=hub)
    loops = mode == 0
    out1, g1 = _run_level(True, heads, mode, dst, src, n, loops, 77, K=max(K, 1))
    out2, g2 = _run_level(False, heads, mode, dst, src, n, loops, 77, K=max(K, 1))
    assert torch.equal(out1, out2)                  # the forward's first output does not depend on the second
    for a, b, nm in zip(g1, g2, ("h", "att", "embW/feat", "embb")):
        scale = max(1.0, float(b.abs().max()))
        torch.testing.assert_close(a, b, atol=2e-5 * scale, rtol=1e-4, msg=lambda s: f"grad {nm}: {s}")


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (5, 0), (9, 2), (64, 3)])
def test_one_pass_tiny_levels(n, m):
    """single-edge and empty levels take the serial path / no path at all"""
    g = torch.Generator().manual_seed(n * 7 + m)
    dst = torch.randint(0, n, (m,), generator=g)
    src = torch.randint(0, n, (m,), generator=g)
    for mode, K in ((0, 0), (2, 1)):
        if m == 0 and mode == 2:
            continue
        out1, g1 = _run_level(True, 4, mode, dst, src, n, mode == 0, 5, K=max(K, 1))
        out2, g2 = _run_level(False, 4, mode, dst, src, n, mode == 0, 5, K=max(K, 1))
        assert torch.equal(out1, out2)
        for a, b in zip(g1, g2):
            assert torch.isfinite(a).all()
            if b.numel():
                torch.testing.assert_close(a, b, atol=2e-5 * max(1.0, float(b.abs().max())), rtol=1e-4)


@pytest.mark.parametrize("heads", [2, 4, 8])
def test_forward_second_output_matches_its_definition(heads):
    """out2[t] = sum_e lambda_e p_e h[src_e], sigma[t] = sum_e lambda_e p_e with lambda = 1 (z > 0) or the slope; out = sum_e p_e h[src_e]"""
    from fragnet_amd import _lib
    from fragnet_amd.plan import GraphPlan, _stream_ptr
    n, m, d = 211, 1300, 128 // heads
    dst, src, g = _graph(n, m, seed=heads, hub="dst")
    h = torch.randn(n, 128, generator=g).to(DEV)
    att = (torch.randn(heads, 2 * d + 128, generator=g) * 0.4).to(DEV)
    s_edge = (torch.randn(heads, m, generator=g) * 0.5).to(DEV)
    plan = GraphPlan([dict(kind="gat", name="l", dst=dst.to(DEV), src=src.to(DEV), n=n, n_loops=0)], DEV)
    lv = plan.levels["l"]
    st = _stream_ptr(torch.device(DEV))
    f32 = dict(dtype=torch.float32, device=DEV)
    s_dst, s_src = torch.empty(n, heads, **f32), torch.empty(n, heads, **f32)
    out, out2, sigma = torch.empty(n, 128, **f32), torch.empty(n, 128, **f32), torch.empty(n, heads, **f32)
    p_sorted = torch.empty(heads, m, **f32)
    et = _lib.EdgeTerm(0, 0, 0, 0, s_edge.data_ptr(), None, None, None)
    _lib.call("fn_node_scalars_f32", h.data_ptr(), att.data_ptr(), att.shape[1], 0, att.shape[1] - d, s_dst.data_ptr(), s_src.data_ptr(), n, heads, st)
    _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att.shape[1], C.byref(et), C.byref(lv.c),
              0.2, out.data_ptr(), p_sorted.data_ptr(), None, out2.data_ptr(), sigma.data_ptr(), 0, None, heads, st)
    torch.cuda.synchronize()
    # from the kernel's own signed probabilities in destination order
    _, role, item_base, seg_base, items, segs = plan.task_meta[0]        # the level's destination task: slices of the plan arena
    assert role == _lib.ROLE_DST and items == m and segs == n
    src_d = plan.aux_a[item_base: item_base + m].long()
    rowptr = plan.rowptr[seg_base: seg_base + n + 1].long() - item_base
    dst_of = torch.repeat_interleave(torch.arange(n, device=DEV), rowptr[1:] - rowptr[:-1])
    p = p_sorted.abs()                                          # [H, m]
    lam = torch.where(torch.signbit(p_sorted), torch.full_like(p, 0.2), torch.ones_like(p))
    hs = h.view(n, heads, d)[src_d]                             # [m, H, d]
    want = torch.zeros(n, heads, d, **f32).index_add_(0, dst_of, p.t().unsqueeze(-1) * hs)
    want2 = torch.zeros(n, heads, d, **f32).index_add_(0, dst_of, (p * lam).t().unsqueeze(-1) * hs)
    wsig = torch.zeros(n, heads, **f32).index_add_(0, dst_of, (p * lam).t())
    torch.testing.assert_close(out, want.view(n, 128), atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(out2, want2.view(n, 128), atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(sigma, wsig, atol=2e-6, rtol=1e-5)


def _oracle_level(heads, dst, src, n, loops, g_seed):
    """oracle autograd of the same level as _run_level(mode 0): returns out and the gradients of (h, att, feat)"""
    from oracle.fragnet_ref import gat_level_materialised
    d = 128 // heads
    m = dst.numel()
    g = torch.Generator().manual_seed(g_seed)
    h = torch.randn(n, 128, generator=g).requires_grad_(True)
    w_out = torch.randn(n, 128, generator=g)
    att = (torch.randn(heads, 2 * d + 128, generator=g) * 0.3).requires_grad_(True)
    feat = torch.randn(m, 128, generator=g).requires_grad_(True)
    e_dst, e_src, edge_vec = dst, src, feat
    if loops:                                                   # add_self_loops: identity edges with a zero edge vector (gat2.py:179-186)
        ar = torch.arange(n)
        e_dst, e_src = torch.cat([dst, ar]), torch.cat([src, ar])
        edge_vec = torch.cat([feat, torch.zeros(n, 128)])
    out, _, _ = gat_level_materialised(h.view(n, heads, d), edge_vec, att.view(heads, 1, -1).squeeze(1), e_dst, e_src, heads)
    if out.shape[0] < n:                                        # scatter_add sizes its output by the largest index
        out = torch.cat([out, torch.zeros(n - out.shape[0], heads, d)])
    out = out.reshape(n, 128)
    (out * w_out).sum().backward()
    return out.detach(), [h.grad, att.grad, feat.grad]


@pytest.mark.parametrize("heads", [1, 4, 8])
@pytest.mark.parametrize("case", ["src_hub", "dst_hub", "single_edge", "both_hubs"])
def test_one_pass_matches_the_oracle_on_the_paths_the_goldens_do_not_reach(heads, case):
    if case == "single_edge":
        n, dst, src = 7, torch.tensor([3]), torch.tensor([5])
    else:
        n, m = 150, 900
        dst, src, _ = _graph(n, m, seed=heads * 31 + len(case), hub={"src_hub": "src", "dst_hub": "dst", "both_hubs": "both"}[case])
    out, grads = _run_level(True, heads, 0, dst, src, n, True, 123)
    want, wgrads = _oracle_level(heads, dst, src, n, True, 123)
    torch.testing.assert_close(out.cpu(), want, atol=1e-4, rtol=1e-4)
    for a, b, nm in zip(grads, wgrads, ("h", "att", "feat")):
        torch.testing.assert_close(a.cpu(), b, atol=1e-4 * max(1.0, float(b.abs().max())), rtol=1e-4, msg=lambda t: f"grad {nm} ({case}): {t}")


@pytest.mark.parametrize("mode", [0, 2])
def test_one_pass_level_with_nodes_and_no_edges(mode):
    """n > 0, m == 0, no self loops (a batch of single-fragment molecules: the fragment-bond graph): the per-edge arrays are empty --
    their pointers may be null -- and the kernel's clamped loads must not touch them; every output row is zero, dL/dh = 0"""
    n = 5
    empty = torch.zeros(0, dtype=torch.long)
    out, grads = _run_level(True, 4, mode, empty, empty, n, False, 9, K=1)
    assert torch.equal(out, torch.zeros_like(out))
    for gr in grads:
        assert torch.isfinite(gr).all() and float(gr.abs().max()) == 0.0 if gr.numel() else True


@pytest.mark.parametrize("hub", [None, "src", "both"])
@pytest.mark.parametrize("mode,K", [(0, 0), (2, 1), (2, 6)])
def test_deferred_form_of_the_pass_equals_the_plain_one(hub, mode, K):
    """fn_gat_bwd_one_f32 with dz_em (four heads): g_h + g_s_dst a_dst, the attention-vector partials after fn_gat_gsd_f32 and the edge
    outputs equal the non-deferred pass's, and g_s_dst (a segment sum of dz) equals <g, out2> - c sigma of round 4's form"""
    from fragnet_amd import _lib
    from fragnet_amd.plan import GraphPlan, _stream_ptr
    H, d, n, m = 4, 32, 301, 1700
    dst, src, g = _graph(n, m, seed=17 + mode + K, hub=hub)
    loops = n if mode == 0 else 0
    plan = GraphPlan([dict(kind="gat", name="l", dst=dst.to(DEV), src=src.to(DEV), n=n, n_loops=loops)], DEV)
    lv = plan.levels["l"]
    M = lv.m
    st = _stream_ptr(torch.device(DEV))
    f32 = dict(dtype=torch.float32, device=DEV)
    h, gout = torch.randn(n, 128, generator=g).to(DEV), torch.randn(n, 128, generator=g).to(DEV)
    att_w = 3 * d if mode == 2 else 2 * d + 128
    att = (torch.randn(H, att_w, generator=g) * 0.3).to(DEV)
    if mode == 2:
        x = plan.sorted_attr("l", torch.randn(m, K, generator=g).to(DEV))
        embW, embb = (torch.randn(d, K, generator=g) * 0.5).to(DEV), (torch.randn(d, generator=g) * 0.5).to(DEV)
        et = _lib.EdgeTerm(2, K, d, d, None, x.data_ptr(), embW.data_ptr(), embb.data_ptr())
    else:
        s_edge = (torch.randn(H, M, generator=g) * 0.5).to(DEV)
        et = _lib.EdgeTerm(0, 0, 0, 0, s_edge.data_ptr(), None, None, None)
    s_dst, s_src = torch.empty(n, H, **f32), torch.empty(n, H, **f32)
    out, out2, sigma, p_em = torch.empty(n, 128, **f32), torch.empty(n, 128, **f32), torch.empty(n, H, **f32), torch.empty(M, H, **f32)
    _lib.call("fn_node_scalars_f32", h.data_ptr(), att.data_ptr(), att_w, 0, att_w - d, s_dst.data_ptr(), s_src.data_ptr(), n, H, st)
    _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w, C.byref(et), C.byref(lv.c),
              0.2, out.data_ptr(), p_em.data_ptr(), None, out2.data_ptr(), sigma.data_ptr(), 1, None, H, st)
    cdot, gsd = torch.empty(n, H, **f32), torch.empty(n, H, **f32)
    _lib.call("fn_gat_cu_f32", gout.data_ptr(), out.data_ptr(), out2.data_ptr(), sigma.data_ptr(), 1.0, cdot.data_ptr(), gsd.data_ptr(), n, H, st)

    def run(deferred):
        g_h, dz_s = torch.empty(n, 128, **f32), (torch.zeros(H, M, **f32) if mode == 0 else None)
        part_e = torch.zeros(_lib.FN_MAX_PART, H * (max(K, 1) + 1), **f32)
        part_a = torch.zeros(256, _lib.FN_MAX_PART, **f32)                      # column-major [256][FN_MAX_PART]
        dz_em = torch.zeros(M, H, **f32) if deferred else None
        n_a, n_e = C.c_int(0), C.c_int(0)
        _lib.call("fn_gat_bwd_one_f32", gout.data_ptr(), h.data_ptr(), p_em.data_ptr(), cdot.data_ptr(), None if deferred else gsd.data_ptr(),
                  C.byref(et), att.data_ptr(), att_w, 0, att_w - d, C.byref(lv.c), 0.2, g_h.data_ptr(), None if dz_s is None else dz_s.data_ptr(),
                  None, part_a.data_ptr(), C.byref(n_a), part_e.data_ptr() if mode == 2 else None, C.byref(n_e), 1,
                  None if dz_em is None else dz_em.data_ptr(), H, st)
        gsd_seg = None
        if deferred:
            gsd_seg = torch.empty(n, H, **f32)
            _lib.call("fn_gat_gsd_f32", dz_em.data_ptr(), C.byref(lv.c), h.data_ptr(), gsd_seg.data_ptr(), part_a.data_ptr(), n_a.value, st)
            a_dst = att[:, :d].reshape(1, 128)                                   # head h's block of the attention vector, as columns
            g_h = g_h + gsd_seg.repeat_interleave(d, dim=1) * a_dst
        torch.cuda.synchronize()
        plan.check()
        return g_h, part_a[:, : n_a.value].sum(1), (part_e[: n_e.value].sum(0) if mode == 2 else dz_s), gsd_seg

    g_h0, pa0, e0, _ = run(False)
    g_h1, pa1, e1, gsd1 = run(True)
    torch.testing.assert_close(gsd1, gsd, atol=2e-5 * max(1.0, float(gsd.abs().max())), rtol=1e-4)
    torch.testing.assert_close(g_h1, g_h0, atol=2e-5 * max(1.0, float(g_h0.abs().max())), rtol=1e-4)
    torch.testing.assert_close(pa1, pa0, atol=2e-5 * max(1.0, float(pa0.abs().max())), rtol=1e-4)
    torch.testing.assert_close(e1, e0, atol=2e-5 * max(1.0, float(e0.abs().max())), rtol=1e-4)
