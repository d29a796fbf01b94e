"""Property tests of the torch-scatter operator surface -- ``ops.scatter_add`` / ``ops.scatter_softmax`` (gat2.py:5; call sites
gat2.py:153-165, 210-219, 234, 257-268, 303-312, 820-821) -- and of the segment plan behind them, against ``oracle/scatter_ref.py``
and a stable argsort, over RANDOM index vectors: unsorted, with empty segments (also at the end: ``dim_size`` beyond the largest
id), a single segment holding everything, single items, no items at all, widths 1 ... 130 (the 128-wide row kernels and the generic
ones), trailing shapes [H, d].  Values and the gradients of both operators; the absolute tolerance of a segment sum grows with the
segment (2e-5 + 1e-6 per item on N(0, 9) data: a long segment is summed by several lanes, a deterministic order but not torch's),
probabilities to 2e-6 / 1e-4.
"""
import numpy as np
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
def scatter_cases(draw):
    n_seg = draw(st.integers(min_value=1, max_value=60))
    kind = draw(st.sampled_from(["random", "one_segment", "sorted", "sparse", "empty", "singletons"]))
    items = 0 if kind == "empty" else draw(st.integers(min_value=1, max_value=400))
    seed = draw(st.integers(min_value=0, max_value=2 ** 31 - 1))
    g = torch.Generator().manual_seed(seed)
    if kind == "one_segment":
        index = torch.full((items,), draw(st.integers(min_value=0, max_value=n_seg - 1)), dtype=torch.int64)
    elif kind == "sparse":                                   # most segments empty
        ids = torch.randint(0, n_seg, (max(1, n_seg // 8),), generator=g)
        index = ids[torch.randint(0, ids.numel(), (items,), generator=g)]
    elif kind == "singletons":
        items = min(items, n_seg)
        index = torch.randperm(n_seg, generator=g)[:items]
    else:
        index = torch.randint(0, n_seg, (items,), generator=g)
        if kind == "sorted":
            index = index.sort().values
    shape = draw(st.sampled_from([(1,), (4,), (32,), (128,), (130,), (4, 32), (8, 16), (2, 3)]))
    explicit = draw(st.booleans())                           # dim_size given (may exceed the largest id + 1) or inferred
    return dict(index=index, shape=shape, seed=seed, dim_size=n_seg if explicit else None, kind=kind)


@settings(max_examples=200, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(scatter_cases())
def test_scatter_add_and_softmax_equal_the_oracle_on_random_indices(case):
    from fragnet_amd import ops
    from oracle import scatter_ref as ref
    index, shape, dim_size = case["index"], case["shape"], case["dim_size"]
    g = torch.Generator().manual_seed(case["seed"] ^ 0x2545F491)
    src = torch.randn((index.numel(),) + shape, generator=g) * 3.0
    note = f"{case['kind']} items={index.numel()} shape={shape} dim_size={dim_size}"
    rows = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    longest = int(torch.bincount(index).max()) if index.numel() else 0
    atol_sum = 2e-5 + 1e-6 * longest
    w_add = torch.randn((rows,) + shape, generator=g)
    w_sm = torch.randn_like(src)

    a = src.clone().requires_grad_(True)
    want_add = ref.scatter_add(a, index, dim=0, dim_size=dim_size)
    (want_add * w_add).sum().backward()
    b = src.clone().requires_grad_(True)
    want_sm = ref.scatter_softmax(b, index, dim=0, dim_size=dim_size)
    (want_sm * w_sm).sum().backward()

    da = src.to(DEV).requires_grad_(True)
    got_add = ops.scatter_add(da, index.to(DEV), dim=0, dim_size=dim_size)
    (got_add * w_add.to(DEV)).sum().backward()
    db = src.to(DEV).requires_grad_(True)
    got_sm = ops.scatter_softmax(db, index.to(DEV), dim=0, dim_size=dim_size)
    (got_sm * w_sm.to(DEV)).sum().backward()
    torch.cuda.synchronize()

    assert tuple(got_add.shape) == tuple(want_add.shape), note
    torch.testing.assert_close(got_add.detach().cpu(), want_add.detach(), atol=atol_sum, rtol=1e-5, msg=lambda s: f"scatter_add [{note}]: {s}")
    torch.testing.assert_close(da.grad.cpu(), a.grad, atol=2e-5, rtol=1e-5, msg=lambda s: f"scatter_add grad [{note}]: {s}")
    torch.testing.assert_close(got_sm.detach().cpu(), want_sm.detach(), atol=2e-6, rtol=1e-4, msg=lambda s: f"scatter_softmax [{note}]: {s}")
    torch.testing.assert_close(db.grad.cpu(), b.grad, atol=atol_sum, rtol=1e-4, msg=lambda s: f"scatter_softmax grad [{note}]: {s}")
    if index.numel():                                        # every segment's probabilities sum to one, per trailing column
        sums = ref.scatter_add(got_sm.detach().cpu(), index, dim_size=rows)
        occupied = torch.zeros(rows, dtype=torch.bool)
        occupied[index] = True
        torch.testing.assert_close(sums[occupied], torch.ones_like(sums[occupied]), atol=1e-5 + 1e-7 * longest, rtol=0)


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(min_value=1, max_value=300), st.integers(min_value=0, max_value=3000), st.integers(min_value=0, max_value=2 ** 31 - 1))
def test_segment_plan_is_the_stable_argsort_of_its_keys(n_seg, items, seed):
    """The CSR behind the operators: rowptr = exclusive prefix sums of the per-segment counts, perm = the stable argsort of the keys
    (ascending original id inside a segment = the reference's sequential scatter order) -- bit-exact."""
    from fragnet_amd.plan import GraphPlan
    g = torch.Generator().manual_seed(seed)
    keys = torch.randint(0, n_seg, (items,), generator=g)
    plan = GraphPlan.segments_only(keys.to(DEV), n_seg)
    seg = plan.segs["s"]
    torch.cuda.synchronize()
    plan.check()
    rowptr = seg.rowptr.cpu().numpy().astype(np.int64)[: n_seg + 1] - seg.pos_base
    perm = seg.perm.cpu().numpy()[:items]
    k = keys.numpy()
    assert np.array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(k, minlength=n_seg))]))
    assert np.array_equal(perm, np.argsort(k, kind="stable"))
