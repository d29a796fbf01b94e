"""Property test of one attention level: ``ops.gat_level`` (HIP, through the C-ABI) against the oracle's materialised-message
restatement of gat2.py:146-169 over RANDOM graphs -- next to the hand-picked cases of test_gpu_parity.py.

hypothesis draws the head count (1, 2, 4, 8), the edge-term form (a stored per-edge row, gat2.py:196-219, or the folded
``Linear(K -> d)`` of a raw attribute, gat2.py:146-164), ``add_self_loops`` (gat2.py:190-194), and the graph: isolated nodes,
duplicate edges, explicit self loops, hubs whose in-degree exceeds the 2 * (32 / heads) edges a half-wave takes in its fast path
(serial three-pass walk), and levels without any edge.  Forward rows, probabilities, the by-source attention read-out and the
gradient of every input are compared; tolerances as in test_gpu_parity.py (2e-5 absolute on O(1) rows, gradients 5e-5 x scale).
"""
import pytest
import torch

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

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


@st.composite
def level_cases(draw):
    heads = draw(st.sampled_from([1, 2, 4, 8]))
    mode = draw(st.sampled_from([0, 2]))
    loops = draw(st.booleans())
    n = draw(st.integers(min_value=1, max_value=70))
    shape = draw(st.sampled_from(["empty", "sparse", "dense", "hub", "two_hubs", "self_loops", "chain"]))
    seed = draw(st.integers(min_value=0, max_value=2 ** 31 - 1))
    g = torch.Generator().manual_seed(seed)
    lph2 = 2 * (32 // heads)                         # edges of a row the half-wave's fast path takes
    if shape == "empty":
        m = 0
    elif shape == "sparse":
        m = draw(st.integers(min_value=1, max_value=max(1, n // 2)))          # most nodes isolated
    elif shape == "chain":
        m = max(1, n - 1)
    else:
        m = draw(st.integers(min_value=1, max_value=260))
    dst = torch.randint(0, n, (m,), generator=g)
    src = torch.randint(0, n, (m,), generator=g)
    if shape == "chain" and n > 1:
        dst, src = torch.arange(1, n), torch.arange(0, n - 1)
    if shape in ("hub", "two_hubs") and m > 0:
        k = min(m, lph2 + draw(st.integers(min_value=1, max_value=40)))       # in-degree beyond the fast path
        dst[:k] = draw(st.integers(min_value=0, max_value=n - 1))
        if shape == "two_hubs" and m > k:
            dst[k:k + min(m - k, lph2 + 3)] = draw(st.integers(min_value=0, max_value=n - 1))
    if shape == "self_loops" and m > 0:
        src[: (m + 1) // 2] = dst[: (m + 1) // 2]                             # explicit loops among the real edges
    K = draw(st.integers(min_value=1, max_value=8)) if mode == 2 else 0
    return dict(heads=heads, mode=mode, loops=loops, n=n, m=int(dst.numel()), dst=dst, src=src, K=K, seed=seed, shape=shape)


def _run(case):
    from fragnet_amd import ops
    from fragnet_amd.plan import GraphPlan
    from oracle.fragnet_ref import gat_level_materialised
    heads, mode, loops, n, m, dst, src, K = (case[k] for k in ("heads", "mode", "loops", "n", "m", "dst", "src", "K"))
    d = 128 // heads
    g = torch.Generator().manual_seed(case["seed"] ^ 0x5bd1e995)
    h = torch.randn(n, 128, generator=g)
    w_out = torch.randn(n, 128, generator=g)
    if mode == 2:
        att = torch.randn(heads, 3 * d, generator=g) * 0.3
        x = torch.randn(m, K, generator=g)
        embW = torch.randn(d, K, generator=g) * 0.5
        embb = torch.randn(d, generator=g) * 0.5
        inputs = (h, att, embW, embb)
    else:
        att = torch.randn(heads, 2 * d + 128, generator=g) * 0.3
        feat = torch.randn(m, 128, generator=g)
        inputs = (h, att, feat)

    # ---- oracle: the messages materialised, torch autograd
    leaves = [t.clone().requires_grad_(True) for t in inputs]
    edge_vec = torch.nn.functional.linear(x, leaves[2], leaves[3]) if mode == 2 else leaves[2]
    rdst, rsrc = dst, src
    if loops:
        # add_self_loops: one loop per node behind the real edges.  Stored-row form (the atom graph, the only level the reference adds
        # loops to): the loop's row is zero.  Folded-Linear form: the loop's raw ATTRIBUTE is zero (x_sorted is 0 at loop positions,
        # include/fragnet_hip.h), so its edge vector is the Linear of zero = the bias.
        rdst = torch.cat([dst, torch.arange(n)])
        rsrc = torch.cat([src, torch.arange(n)])
        loop_vec = torch.nn.functional.linear(torch.zeros(n, K), leaves[2], leaves[3]) if mode == 2 else torch.zeros(n, edge_vec.shape[1])
        edge_vec = torch.cat([edge_vec, loop_vec])
    if rdst.numel():
        want, want_p, want_attn = gat_level_materialised(leaves[0].view(n, heads, d), edge_vec, leaves[1], rdst, rsrc, heads)
        if want.shape[0] < n:                        # scatter_add sizes its result by the largest destination id
            want = torch.cat([want, torch.zeros(n - want.shape[0], heads, d)])
        if want_attn.shape[0] < n:
            want_attn = torch.cat([want_attn, torch.zeros(n - want_attn.shape[0], heads)])
        want = want.reshape(n, 128)
        (want * w_out).sum().backward()
    else:                                            # no edge at all: zero rows, no gradient
        want, want_p, want_attn = torch.zeros(n, 128), torch.zeros(0, heads), torch.zeros(n, heads)

    # ---- HIP
    plan = GraphPlan([dict(kind="gat", name="l", dst=dst.to(DEV), src=src.to(DEV), n=n, n_loops=n if loops else 0)], DEV)
    lv = plan.levels["l"]
    dl = [t.detach().clone().to(DEV).requires_grad_(True) for t in inputs]
    if mode == 2:
        out, probs, p_sorted = ops.gat_level(dl[0], dl[1], lv, heads, x_sorted=plan.sorted_attr("l", x.to(DEV)), embW=dl[2], embb=dl[3],
                                             want_probs=True)
    else:
        out, probs, p_sorted = ops.gat_level(dl[0], dl[1], lv, heads, s_sorted=ops.row_dots_sorted(dl[2], dl[1], d, lv), want_probs=True)
    (out * w_out.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    plan.check()
    note = f"{case['shape']} H={heads} mode={mode} loops={loops} n={n} m={m} K={K}"
    assert torch.isfinite(out).all(), note
    torch.testing.assert_close(out.detach().cpu(), want.detach(), atol=2e-5, rtol=1e-4, msg=lambda s: f"out [{note}]: {s}")
    torch.testing.assert_close(probs.cpu().reshape(want_p.shape), want_p.detach(), atol=2e-6, rtol=1e-4, msg=lambda s: f"probs [{note}]: {s}")
    torch.testing.assert_close(ops.attn_by_src(p_sorted, lv, heads).cpu(), want_attn.detach(), atol=2e-5, rtol=1e-4,
                               msg=lambda s: f"attention by source [{note}]: {s}")
    for got, ref, nm in zip(dl, leaves, ("h", "att", "embW / feat", "embb")):
        rg = ref.grad if ref.grad is not None else torch.zeros_like(ref)
        gg = got.grad.cpu() if got.grad is not None else torch.zeros_like(ref)
        scale = max(1.0, float(rg.abs().max())) if rg.numel() else 1.0
        torch.testing.assert_close(gg, rg, atol=5e-5 * scale, rtol=1e-4, msg=lambda s: f"grad {nm} [{note}]: {s}")


@settings(max_examples=240, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(level_cases())
def test_gat_level_equals_the_materialised_oracle_on_random_graphs(case):
    _run(case)


@pytest.mark.parametrize("heads", [1, 2, 4, 8])
def test_gat_level_hub_just_beyond_the_fast_path(heads):
    """in-degree 2 * (32 / heads) is the last row the half-wave takes in its fast path, + 1 the first it walks serially."""
    for extra in (0, 1):
        k = 2 * (32 // heads) + extra
        n = 9
        dst = torch.cat([torch.full((k,), 3), torch.tensor([0, 8])])
        src = torch.cat([torch.arange(k) % n, torch.tensor([5, 8])])
        _run(dict(heads=heads, mode=0, loops=False, n=n, m=int(dst.numel()), dst=dst, src=src, K=0, seed=heads + extra, shape=f"hub{k}"))
