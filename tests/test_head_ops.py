"""Prediction-head small ops (include/fragnet_hip.h: fn_gate_colsum_f32, fn_small_linear(_bwd)_f32) and the fused
``ops.mlp_head`` autograd node against plain torch and against the unfused layer-by-layer path."""
import copy

import pytest
import torch

gpu = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@gpu
@pytest.mark.parametrize("rows,cols", [(528, 1024), (37, 128), (0, 64), (1, 4), (300, 36), (28104, 64), (2049, 32)])
def test_gate_colsum_matches_torch(rows, cols):
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    dev = _dev()
    torch.manual_seed(rows + cols)
    g = torch.randn(rows, cols, device=dev)
    y = torch.relu(torch.randn(rows, cols, device=dev))            # about half zeros, like relu(dropout(.))
    gx, cs = torch.full_like(g, 7.0), torch.full((cols,), 7.0, device=dev)
    n_ws = _lib.load().fn_gate_colsum_ws(rows, cols)
    ws = torch.empty(n_ws, device=dev) if n_ws else None
    assert (n_ws > 0) == (rows > 2048)
    _lib.call("fn_gate_colsum_f32", g.data_ptr(), y.data_ptr(), gx.data_ptr(), cs.data_ptr(), rows, cols, 1.25,
              None if ws is None else ws.data_ptr(), _stream_ptr(dev))
    ref = torch.where(y > 0, g * 1.25, torch.zeros_like(g))
    assert torch.equal(gx, ref)
    torch.testing.assert_close(cs, ref.double().sum(0).float(), atol=2e-4 * max(1.0, rows / 500) ** 0.5, rtol=1e-5)


@gpu
@pytest.mark.parametrize("M,K,C", [(528, 512, 1), (528, 512, 12), (5, 128, 3), (0, 64, 2), (100, 36, 16), (28104, 32, 1), (13872, 32, 3)])
def test_small_linear_matches_torch(M, K, C):
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    dev = _dev()
    torch.manual_seed(M + K + C)
    x, w, b = torch.randn(M, K, device=dev), torch.randn(C, K, device=dev) * 0.1, torch.randn(C, device=dev)
    g = torch.randn(M, C, device=dev)
    y = torch.empty(M, C, device=dev)
    st = _stream_ptr(dev)
    _lib.call("fn_small_linear_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, C, M, st)
    torch.testing.assert_close(y, (x.double() @ w.double().t() + b.double()).float(), atol=2e-5, rtol=1e-5)
    gx, dW, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    n_ws = _lib.load().fn_small_linear_bwd_ws(M, K, C)
    ws = torch.empty(n_ws, device=dev) if n_ws else None
    _lib.call("fn_small_linear_bwd_f32", g.data_ptr(), x.data_ptr(), w.data_ptr(), gx.data_ptr(), dW.data_ptr(), db.data_ptr(), M, K, C,
              0.0, None if ws is None else ws.data_ptr(), st)
    torch.testing.assert_close(gx, (g.double() @ w.double()).float(), atol=2e-5, rtol=1e-5)
    tol = 2e-4 * max(1.0, M / 500) ** 0.5
    torch.testing.assert_close(dW, (g.double().t() @ x.double()).float(), atol=tol, rtol=1e-5)
    torch.testing.assert_close(db, g.double().sum(0).float(), atol=tol, rtol=1e-5)
    gx2 = torch.empty_like(x)                                        # gated: the backward of the relu(dropout(.)) that produced x
    _lib.call("fn_small_linear_bwd_f32", g.data_ptr(), x.data_ptr(), w.data_ptr(), gx2.data_ptr(), dW.data_ptr(), db.data_ptr(), M, K, C,
              1.25, None if ws is None else ws.data_ptr(), st)
    assert torch.equal(gx2, torch.where(x > 0, gx * 1.25, torch.zeros_like(gx)))


@gpu
@pytest.mark.parametrize("M,K,N", [(528, 256, 1024), (528, 1024, 1024), (33, 128, 512), (1, 4, 4), (200, 20, 36), (0, 64, 32), (4096, 132, 68)])
def test_dense_layer_kernels_match_torch(M, K, N):
    """fn_dense_fwd_f32 / fn_dense_bwd_f32 (fp32 matrix cores, fused bias + dropout + ReLU / gate + bias gradient) against
    float64 products and the standalone fn_dropout_act_f32 Philox stream."""
    import ctypes as C
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    dev = _dev()
    st = _stream_ptr(dev)
    torch.manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
    lin = (x.double() @ w.double().t() + b.double()).float()
    y = torch.full((M, N), 7.0, device=dev)
    _lib.call("fn_dense_fwd_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, None, st)
    torch.testing.assert_close(y, lin, atol=2e-5, rtol=1e-5)
    off_dev = torch.tensor([40], dtype=torch.int64, device=dev)
    act = _lib.ActEpilogue(None, 0.25, 1, 99, 1000, off_dev.data_ptr())
    z = torch.full((M, N), 7.0, device=dev)
    _lib.call("fn_dense_fwd_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), M, K, N, C.byref(act), st)
    want = torch.empty_like(y)
    if M:
        _lib.call("fn_dropout_act_f32", y.data_ptr(), want.data_ptr(), y.numel(), 0.25, 99, 1040, None, 1, st)
        assert torch.equal(z > 0, want > 0) or ((z > 0) != (want > 0)).sum() <= 2          # pre-activations within rounding of 0
        torch.testing.assert_close(z, want, atol=3e-5, rtol=1e-5)
    g = torch.randn(M, N, device=dev)
    xr = torch.relu(x)                                               # a layer input that is itself relu(dropout(.)): about half zeros
    for gate in (4.0 / 3.0, 0.0):
        gx, dW, db = torch.full_like(x, 7.0), torch.full_like(w, 7.0), torch.full_like(b, 7.0)
        _lib.call("fn_dense_bwd_f32", g.data_ptr(), xr.data_ptr(), w.data_ptr(), gx.data_ptr(), gate, dW.data_ptr(), db.data_ptr(),
                  M, K, N, M, st)
        tol = 2e-4 * max(1.0, M / 500) ** 0.5
        want = g.double() @ w.double()
        if gate:
            want = torch.where(xr > 0, want * gate, torch.zeros_like(want))
        torch.testing.assert_close(gx, want.float(), atol=3e-5, rtol=1e-5)
        torch.testing.assert_close(dW, (g.double().t() @ xr.double()).float(), atol=tol, rtol=1e-5)
        torch.testing.assert_close(db, g.double().sum(0).float(), atol=tol, rtol=1e-5)
    dW2 = torch.full_like(w, 7.0)                                    # first layer: no input gradient, no bias gradient
    _lib.call("fn_dense_bwd_f32", g.data_ptr(), x.data_ptr(), w.data_ptr(), None, 0.0, dW2.data_ptr(), None, M, K, N, M, st)
    torch.testing.assert_close(dW2, (g.double().t() @ x.double()).float(), atol=2e-4 * max(1.0, M / 500) ** 0.5, rtol=1e-5)
    if 0 < M <= 4000:                                                # padding rows behind M: written as 0 by the same launch
        gx3 = torch.full((M + 40, K), 7.0, device=dev)
        _lib.call("fn_dense_bwd_f32", g.data_ptr(), xr.data_ptr(), w.data_ptr(), gx3.data_ptr(), 0.0, dW2.data_ptr(), None, M, K, N, M + 40, st)
        assert torch.equal(gx3[:M], gx) and not gx3[M:].any()
        y3 = torch.full((M + 5, N if N <= 16 else 1), 7.0, device=dev)
        Cs = y3.shape[1]
        _lib.call("fn_small_linear_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), y3.data_ptr(), M, K, Cs, M + 5, st)
        torch.testing.assert_close(y3[:M], lin[:, :Cs], atol=2e-5, rtol=1e-5)
        assert not y3[M:].any()


@gpu
@pytest.mark.parametrize("M,K,N", [(2048, 1024, 1024), (2100, 128, 1024), (1600, 64, 1000), (4096, 32, 512), (1537, 96, 1024)])
def test_dense_forward_on_workgroup_shared_tiles_matches_float64_and_the_per_wave_kernel(M, K, N):
    """fn_dense_fwd_f32 on tall inputs (>= 192 tiles of 64 x 128, K a multiple of 32: k_dense_fwd_tiles, FN_TUNE_DENSE_TILES): ragged row
    and column edges, reductions shorter than the ring (K = 32, 64), against float64 products, against the standalone Philox stream
    (bias + dropout + ReLU epilogue) and against the per-wave-operand kernel the same call runs with the key off."""
    import ctypes as C
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    dev = _dev()
    st = _stream_ptr(dev)
    assert -(-M // 64) * -(-N // 128) >= 192 and K % 32 == 0
    torch.manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
    lin = (x.double() @ w.double().t() + b.double()).float()
    y = torch.full((M + 3, N), 7.0, device=dev)                      # three guard rows behind the output
    _lib.call("fn_dense_fwd_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, None, st)
    torch.testing.assert_close(y[:M], lin, atol=2e-5, rtol=1e-5)
    assert bool((y[M:] == 7.0).all())
    off_dev = torch.tensor([40], dtype=torch.int64, device=dev)
    act = _lib.ActEpilogue(None, 0.25, 1, 99, 1000, off_dev.data_ptr())
    z = torch.full((M, N), 7.0, device=dev)
    _lib.call("fn_dense_fwd_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), M, K, N, C.byref(act), st)
    want = torch.empty((M, N), device=dev)
    plain = y[:M].contiguous()
    _lib.call("fn_dropout_act_f32", plain.data_ptr(), want.data_ptr(), plain.numel(), 0.25, 99, 1040, None, 1, st)
    torch.testing.assert_close(z, want, atol=3e-5, rtol=1e-5)
    try:
        _lib.call("fn_set_tuning", 32, 0)
        z0 = torch.full((M, N), 7.0, device=dev)
        _lib.call("fn_dense_fwd_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), z0.data_ptr(), M, K, N, C.byref(act), st)
    finally:
        _lib.call("fn_set_tuning", 32, 1)
    assert ((z > 0) != (z0 > 0)).sum() <= 2                          # the same masks; pre-activations within rounding of 0 may flip
    torch.testing.assert_close(z, z0, atol=3e-5, rtol=1e-5)


@gpu
def test_dense_layer_kernels_reject_bad_shapes():
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    dev = _dev()
    t = torch.zeros(64, 64, device=dev)
    for M, K, N in [(8, 6, 8), (8, 8, 6), (4097, 8, 8)]:
        with pytest.raises(Exception, match="fn_dense_fwd_f32"):
            _lib.call("fn_dense_fwd_f32", t.data_ptr(), t.data_ptr(), None, t.data_ptr(), M, K, N, None, _stream_ptr(dev))


@gpu
@pytest.mark.parametrize("p,n_classes", [(0.0, 1), (0.1, 1), (0.2, 12)])
def test_fused_head_equals_layer_by_layer_path(p, n_classes):
    """Same Philox draws in the same order: the fused node must reproduce the unfused path (outputs and all grads)."""
    from fragnet_amd import ops
    from fragnet_amd.model import FTHead3
    dev = _dev()
    torch.manual_seed(3)
    head_a = FTHead3(input_dim=128, drop_ratio=p, n_classes=n_classes).to(dev).train()
    head_b = copy.deepcopy(head_a)
    x_a = torch.randn(64, 256, device=dev, requires_grad=True)
    x_b = x_a.detach().clone().requires_grad_(True)
    head_a.rng, head_b.rng = ops.PhiloxStream(seed=11), ops.PhiloxStream(seed=11)
    out_a = head_a(x_a)                                            # fused (ops.mlp_head)

    h = x_b                                                        # the loop _PredictorStack._run falls back to
    for lin in head_b.predictor[:-1]:
        h = ops.dropout_act(lin(h), p, True, True, head_b.rng)
    out_b = head_b.predictor[-1](h)
    torch.testing.assert_close(out_a, out_b, atol=1e-5, rtol=1e-5)
    t = torch.randn_like(out_a)
    (out_a * t).sum().backward()
    (out_b * t).sum().backward()
    torch.testing.assert_close(x_a.grad, x_b.grad, atol=1e-5, rtol=1e-4)
    for (n, pa), (_, pb) in zip(head_a.named_parameters(), head_b.named_parameters()):
        torch.testing.assert_close(pa.grad, pb.grad, atol=2e-5, rtol=1e-4, msg=lambda m, n=n: f"{n}: {m}")
    assert head_a.rng.offset == head_b.rng.offset


@gpu
def test_fused_head_skips_padding_rows():
    """live_rows (static-shape batches): rows behind it are not computed -- outputs and input gradients 0 there, everything
    else as if the input ended at live_rows."""
    from fragnet_amd import ops
    from fragnet_amd.model import FTHead3
    dev = _dev()
    torch.manual_seed(5)
    head_a = FTHead3(input_dim=128, drop_ratio=0.0, n_classes=1).to(dev).train()
    head_b = copy.deepcopy(head_a)
    x_a = torch.randn(80, 256, device=dev, requires_grad=True)
    x_b = x_a.detach()[:64].clone().requires_grad_(True)
    head_a.rng, head_b.rng = ops.PhiloxStream(seed=11), ops.PhiloxStream(seed=11)
    head_a.live_rows = 64
    out_a, out_b = head_a(x_a), head_b(x_b)
    assert out_a.shape == (80, 1) and torch.equal(out_a[:64], out_b) and not out_a[64:].any()
    t = torch.randn_like(out_a)
    (out_a * t).sum().backward()
    (out_b * t[:64]).sum().backward()
    assert torch.equal(x_a.grad[:64], x_b.grad) and not x_a.grad[64:].any()
    for (n, pa), (_, pb) in zip(head_a.named_parameters(), head_b.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), n


@gpu
def test_fused_head_matches_plain_torch_without_dropout():
    from fragnet_amd import ops
    from fragnet_amd.model import FTHead3
    dev = _dev()
    torch.manual_seed(4)
    head = FTHead3(input_dim=128, drop_ratio=0.0, n_classes=1).to(dev).train()
    head.rng = ops.PhiloxStream(seed=1)
    x = torch.randn(33, 256, device=dev, requires_grad=True)
    out = head(x)
    h = x.detach().double()
    for lin in head.predictor[:-1]:
        h = torch.relu(h @ lin.weight.double().t() + lin.bias.double())
    ref = h @ head.predictor[-1].weight.double().t() + head.predictor[-1].bias.double()
    torch.testing.assert_close(out, ref.float(), atol=1e-5, rtol=1e-5)


@gpu
def test_parameter_gradients_are_written_into_the_flat_buffer():
    """Encoder and fused head write d loss / d param straight into FlatAdam's flat gradient buffer: after backward
    every live parameter's .grad is its slot (gather_grads has nothing to copy) and holds the same numbers as a
    run without slots."""
    from fragnet_amd import data, parallel, synth, train
    from fragnet_amd.model import FragNetFineTune
    dev = _dev()
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(24, seed=5)), dev)
    torch.manual_seed(2)
    model = FragNetFineTune(n_classes=1, num_layer=2, drop_ratio=0.0, act="relu").to(dev).train()     # ReLU head: fused node
    ref = copy.deepcopy(model)

    def run(m):
        loss = train.compute_mse_loss(m(dict(batch)), batch["y"]) if hasattr(train, "compute_mse_loss") else \
            torch.nn.functional.mse_loss(m(dict(batch)).reshape(-1), batch["y"].reshape(-1).float())
        loss.backward()

    opt = parallel.FlatAdam.for_live_parameters(model, lambda: run(model), lr=1e-3)
    opt.zero_grad()
    run(model)
    names = {id(q): n for n, q in model.named_parameters()}
    base, elsewhere = opt.grad.data_ptr(), []
    for p, off in zip(opt.params, opt.offsets):
        assert off % 4 == 0                                        # every slot starts on a 16-byte boundary
        if p.grad.data_ptr() != base + 4 * off:
            elsewhere.append(names[id(p)])
    assert not elsewhere, f"gradients not in place: {elsewhere}"
    opt.gather_grads()
    run(ref)                                                       # no optimiser attached: ordinary .grad tensors
    live = [q for q in ref.parameters() if q.grad is not None]
    assert len(live) == len(opt.params)
    for q, p, off in zip(live, opt.params, opt.offsets):
        torch.testing.assert_close(opt.grad[off: off + p.numel()].view_as(q), q.grad, atol=1e-6, rtol=1e-5)


@gpu
def test_masked_mse_multi_matches_the_sum_of_single_losses():
    from fragnet_amd import ops
    dev = _dev()
    torch.manual_seed(6)
    sizes = [(28104, 1), (13872, 1), (528, 3)]
    outs = [torch.randn(b, t, device=dev, requires_grad=True) for b, t in sizes]
    ys = [torch.randn(b, t, device=dev) for b, t in sizes]
    ws = [(torch.rand(b, device=dev) > 0.1).float() for b, _ in sizes]
    scale = torch.tensor([0.9, 1.3], device=dev)
    loss = ops.masked_mse_multi([(2.0, 0), (1.0, 1), (1.0, -1)], scale, outs[0], ys[0], ws[0], outs[1], ys[1], ws[1], outs[2], ys[2], ws[2])
    loss.backward()
    refs = [o.detach().double().requires_grad_(True) for o in outs]

    def mse(o, y, w):
        return (w.double()[:, None] * (o - y.double()) ** 2).sum() / (w.double().sum() * o.shape[1])
    want = 2 * 0.9 * mse(refs[0], ys[0], ws[0]) + 1.3 * mse(refs[1], ys[1], ws[1]) + mse(refs[2], ys[2], ws[2])
    want.backward()
    assert abs(float(loss) - float(want)) < 1e-5 * max(1.0, abs(float(want)))
    for o, r in zip(outs, refs):
        torch.testing.assert_close(o.grad, r.grad.float(), atol=1e-7, rtol=1e-4)


@gpu
def test_masked_bce_matches_the_torch_formula():
    from fragnet_amd import ops
    dev = _dev()
    torch.manual_seed(8)
    B, T = 1040, 12
    out = torch.randn(B, T, device=dev) * 3
    out.requires_grad_(True)
    y = torch.randint(-1, 2, (B, T), device=dev).float()             # -1 = missing label
    w = (torch.arange(B, device=dev) < 1024).float()                  # padded molecules at the end
    loss = ops.masked_bce(out, y, w)
    loss.backward()
    ref = out.detach().double().requires_grad_(True)
    valid = (y > -0.5) & (w[:, None] > 0)
    mat = torch.nn.functional.binary_cross_entropy_with_logits(ref, y.clamp(min=0).double(), reduction="none")
    want = torch.where(valid, mat, torch.zeros_like(mat)).sum() / valid.sum()
    want.backward()
    assert abs(float(loss) - float(want)) < 2e-6
    torch.testing.assert_close(out.grad, ref.grad.float(), atol=1e-8, rtol=1e-4)


@gpu
def test_dense_bwd_with_no_input_rows_zeroes_the_padding_rows_without_reading_null():
    """M = 0 with g_x and M_out > 0: g_y / X may be NULL then; the input gradient's rows are all padding = 0,
    dW and db are 0."""
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    dev = _dev()
    K, N, M_out = 64, 32, 24
    w = torch.randn(N, K, device=dev)
    gx = torch.full((M_out, K), 7.0, device=dev)
    dW, db = torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
    _lib.call("fn_dense_bwd_f32", None, None, w.data_ptr(), gx.data_ptr(), 1.25, dW.data_ptr(), db.data_ptr(), 0, K, N, M_out, _stream_ptr(dev))
    torch.cuda.synchronize()
    assert float(gx.abs().max()) == 0.0 and float(dW.abs().max()) == 0.0 and float(db.abs().max()) == 0.0


@gpu
def test_fused_head_with_drop_probability_one_passes_no_gradient():
    """p = 1 drops everything: outputs are the last bias, every gradient below the last layer is exactly 0 (the gate scale
    must not be read as "no gate")."""
    from fragnet_amd import ops
    dev = _dev()
    torch.manual_seed(0)
    lins = [torch.nn.Linear(64, 32).to(dev), torch.nn.Linear(32, 32).to(dev), torch.nn.Linear(32, 2).to(dev)]
    x = torch.randn(20, 64, device=dev, requires_grad=True)
    out = ops.mlp_head(x, lins, 1.0, True, ops.PhiloxStream(seed=3))
    torch.testing.assert_close(out, lins[-1].bias.detach().expand(20, 2))
    out.sum().backward()
    assert float(x.grad.abs().max()) == 0.0
    for lin in lins[:-1]:
        assert float(lin.weight.grad.abs().max()) == 0.0 and float(lin.bias.grad.abs().max()) == 0.0
    torch.testing.assert_close(lins[-1].bias.grad, torch.full((2,), 20.0, device=dev))


@gpu
@pytest.mark.parametrize("rows", [(0, 5), (1, 31), (33, 64), (1000, 13337), (27001, 13872)])
def test_fused_towers_match_torch(rows):
    """ops.towers (csrc/tower.hip: PretrainTask's Linear(128->64)->ReLU->Linear(64->32)->ReLU->Linear(32->1) towers, two towers in
    one launch each way) against plain torch: outputs, input gradients, all six parameter gradients per tower; ragged tile
    tails, an empty tower, tall inputs (several tiles per workgroup in the backward)."""
    from fragnet_amd import ops
    dev = _dev()
    torch.manual_seed(sum(rows))
    mk = lambda: [torch.nn.Linear(128, 64).to(dev), torch.nn.Linear(64, 32).to(dev), torch.nn.Linear(32, 1).to(dev)]
    ta, tb = mk(), mk()
    xa = torch.randn(rows[0], 128, device=dev, requires_grad=True)
    xb = torch.randn(rows[1], 128, device=dev, requires_grad=True)
    assert ops.tower_ok(xa, ta) and ops.tower_ok(xb, tb)
    oa, ob = ops.towers([(xa, ta), (xb, tb)])
    ga, gb = torch.randn_like(oa), torch.randn_like(ob)
    torch.autograd.backward([oa, ob], [ga, gb])
    got = [xa.grad.clone(), xb.grad.clone()] + [q.grad.clone() for lin in ta + tb for q in (lin.weight, lin.bias)]
    for q in [xa, xb] + [q for lin in ta + tb for q in (lin.weight, lin.bias)]:
        q.grad = None
    ref = lambda x, lins: lins[2](torch.relu(lins[1](torch.relu(lins[0](x)))))
    ra, rb = ref(xa, ta), ref(xb, tb)
    torch.testing.assert_close(oa, ra.detach(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(ob, rb.detach(), atol=2e-5, rtol=1e-5)
    torch.autograd.backward([ra, rb], [ga, gb])
    want = [xa.grad, xb.grad] + [q.grad for lin in ta + tb for q in (lin.weight, lin.bias)]
    for i, (g, w) in enumerate(zip(got, want)):
        scale = max(1.0, float(w.abs().max())) if w.numel() else 1.0
        torch.testing.assert_close(g, w, atol=3e-5 * scale, rtol=1e-4, msg=lambda m, i=i: f"tensor {i}: {m}")


@gpu
def test_pretrain_task_fused_towers_equal_the_generic_path():
    """FragNetPreTrain with the fused towers against the same model with `fused_towers = False` (ops.mlp_head on the tall inputs):
    the four outputs and every gradient."""
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetPreTrain
    from fragnet_amd.train import pretrain_loss
    dev = _dev()
    b = data.batch_to(data.collate_fn_pt(synth.synth_molecules(24, seed=8, profile="esol", pretrain_targets=True)), dev)
    torch.manual_seed(2)
    model = FragNetPreTrain(num_layer=2, drop_ratio=0.0, edge_features=17).to(dev).train()
    res = []
    for fused in (True, False):
        model.head.fused_towers = fused
        for q in model.parameters():
            q.grad = None
        outs = model(dict(b))
        pretrain_loss(outs, b).backward()
        res.append(([o.detach().clone() for o in outs if o is not None], {n: q.grad.clone() for n, q in model.named_parameters() if q.grad is not None}))
    for a, c in zip(*[r[0] for r in res]):
        torch.testing.assert_close(a, c, atol=2e-5, rtol=1e-5)
    assert res[0][1].keys() == res[1][1].keys()
    for n in res[0][1]:
        torch.testing.assert_close(res[0][1][n], res[1][1][n], atol=2e-5, rtol=1e-3, msg=lambda m, n=n: f"{n}: {m}")


@gpu
@pytest.mark.parametrize("kind", ["mse", "bce"])
@pytest.mark.parametrize("M,M_out,K,C", [(512, 552, 512, 1), (1000, 1024, 512, 12), (5, 8, 128, 3), (64, 64, 1024, 4), (3, 3, 36, 16), (0, 4, 64, 2)])
def test_last_linear_with_loss_in_one_launch_equals_the_three_launches(kind, M, M_out, K, C):
    """fn_small_linear_loss_f32 + the riders of fn_dense_bwd_tail_f32 against fn_small_linear_f32 -> fn_masked_mse/bce_f32 ->
    fn_small_linear_bwd_f32: predictions, d loss / d out and the input gradient bit for bit (same operation order), the sums over
    rows (dW, db, the loss value) to rounding; padding rows (M .. M_out) have weight 0 and prediction 0."""
    from fragnet_amd import _lib
    from fragnet_amd.plan import _stream_ptr
    import ctypes as C_
    dev = _dev()
    torch.manual_seed(M + K + C)
    st = _stream_ptr(dev)
    x = torch.relu(torch.randn(M, K, device=dev))
    w, b = torch.randn(C, K, device=dev) * 0.1, torch.randn(C, device=dev)
    tgt = torch.randn(M_out, C, device=dev)
    if kind == "bce":
        tgt = (tgt > 0).float()
        tgt[torch.rand(M_out, C, device=dev) < 0.2] = -1.0                  # missing labels
    row_w = torch.zeros(M_out, device=dev)
    row_w[:M] = 1.0
    if M > 4:
        row_w[3] = 0.0                                                      # a real row the caller masks out
    if M == 0:                                                              # padding rows only: predictions 0, nothing to reduce
        y1, parts = torch.full((M_out, C), 7.0, device=dev), torch.empty(_lib.load().fn_small_linear_loss_ws(M_out), device=dev)
        _lib.call("fn_small_linear_loss_f32", None, w.data_ptr(), b.data_ptr(), tgt.data_ptr(), row_w.data_ptr(),
                  _lib.LOSS_MSE if kind == "mse" else _lib.LOSS_BCE, y1.data_ptr(), None, None, 1.25, parts.data_ptr(), 0, K, C, M_out, st)
        assert not y1.any()
        return
    # the three launches
    y0 = torch.empty(M_out, C, device=dev)
    _lib.call("fn_small_linear_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), y0.data_ptr(), M, K, C, M_out, st)
    loss0, g0 = torch.empty((), device=dev), torch.empty(M_out, C, device=dev)
    _lib.call("fn_masked_mse_f32" if kind == "mse" else "fn_masked_bce_f32", y0.data_ptr(), tgt.data_ptr(), row_w.data_ptr(), M_out, C,
              loss0.data_ptr(), g0.data_ptr(), st)
    gx0, dW0, db0 = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    _lib.call("fn_small_linear_bwd_f32", g0.data_ptr(), x.data_ptr(), w.data_ptr(), gx0.data_ptr(), dW0.data_ptr(), db0.data_ptr(), M, K, C,
              1.25, None, st)
    # one launch + riders in a dense backward launch (a 64 -> 32 layer whose own results must not change)
    y1, g1, gx1 = torch.full((M_out, C), 7.0, device=dev), torch.full((M, C), 7.0, device=dev), torch.full_like(x, 7.0)
    n_part = _lib.load().fn_small_linear_loss_ws(M_out)
    assert n_part == (M_out + 3) // 4
    parts = torch.full((n_part,), 7.0, device=dev)
    _lib.call("fn_small_linear_loss_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), tgt.data_ptr(), row_w.data_ptr(),
              _lib.LOSS_MSE if kind == "mse" else _lib.LOSS_BCE, y1.data_ptr(), g1.data_ptr(), gx1.data_ptr(), 1.25, parts.data_ptr(), M, K, C,
              M_out, st)
    assert torch.equal(y1, y0) and torch.equal(g1, g0[:M]) and torch.equal(gx1, gx0)
    gy, X, W2 = torch.randn(40, 32, device=dev), torch.relu(torch.randn(40, 64, device=dev)), torch.randn(32, 64, device=dev)
    ref = [torch.empty(40, 64, device=dev), torch.empty(32, 64, device=dev), torch.empty(32, device=dev)]
    _lib.call("fn_dense_bwd_f32", gy.data_ptr(), X.data_ptr(), W2.data_ptr(), ref[0].data_ptr(), 1.25, ref[1].data_ptr(), ref[2].data_ptr(),
              40, 64, 32, 40, st)
    got = [torch.empty_like(t) for t in ref]
    dW1, db1, loss1 = torch.full_like(w, 7.0), torch.full_like(b, 7.0), torch.full((), 7.0, device=dev)
    tail = _lib.SmallDw(g1.data_ptr(), x.data_ptr(), dW1.data_ptr(), db1.data_ptr(), parts.data_ptr(), loss1.data_ptr(), n_part, M, K, C)
    _lib.call("fn_dense_bwd_tail_f32", gy.data_ptr(), X.data_ptr(), W2.data_ptr(), got[0].data_ptr(), 1.25, got[1].data_ptr(), got[2].data_ptr(),
              40, 64, 32, 40, C_.byref(tail), st)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    tol = 2e-4 * max(1.0, M / 500) ** 0.5 * float(g0.abs().max().clamp_min(1e-3))
    torch.testing.assert_close(dW1, dW0, atol=tol, rtol=1e-5)
    torch.testing.assert_close(db1, db0, atol=tol, rtol=1e-5)
    torch.testing.assert_close(loss1, loss0, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(dW1, (g0[:M].double().t() @ x.double()).float(), atol=tol, rtol=1e-5)


@gpu
@pytest.mark.parametrize("loss_kind,n_classes,p", [("regr", 1, 0.1), ("clsf", 12, 0.2), ("regr", 3, 0.0)])
def test_head_with_fused_loss_equals_head_then_loss(loss_kind, n_classes, p):
    """ops.mlp_head(loss=...) -- what GraphedTrainStep arms the head with -- against the head followed by the loss kernel: predictions
    and the input gradient bit for bit, loss and parameter gradients to rounding; padding rows behind live_rows."""
    from fragnet_amd import _lib, ops
    from fragnet_amd.graphstep import masked_bce_loss, masked_regr_loss
    from fragnet_amd.model import FTHead3
    dev = _dev()
    torch.manual_seed(9)
    head_a = FTHead3(input_dim=128, drop_ratio=p, n_classes=n_classes).to(dev).train()
    head_b = copy.deepcopy(head_a)
    x_a = torch.randn(80, 256, device=dev, requires_grad=True)
    x_b = x_a.detach().clone().requires_grad_(True)
    y = torch.randn(80, n_classes, device=dev)
    if loss_kind == "clsf":
        y = (y > 0).float()
        y[torch.rand_like(y) < 0.2] = -1.0
    w = torch.zeros(80, device=dev)
    w[:70] = 1.0
    head_a.rng, head_b.rng = ops.PhiloxStream(seed=11), ops.PhiloxStream(seed=11)
    head_a.live_rows = head_b.live_rows = 70
    unit = ops.unit_grad(dev)
    head_a.loss_spec = (_lib.LOSS_MSE if loss_kind == "regr" else _lib.LOSS_BCE, y, w)
    out_a = head_a(x_a)
    head_a.loss_spec = None
    loss_a, y_seen, w_seen = out_a._fragnet_loss
    assert y_seen is y and w_seen is w and not out_a.requires_grad
    loss_a.backward(gradient=unit)
    out_b = head_b(x_b)
    loss_b = (masked_regr_loss if loss_kind == "regr" else masked_bce_loss)(out_b, y, w)
    loss_b.backward(gradient=unit)
    assert torch.equal(out_a, out_b) and torch.equal(x_a.grad, x_b.grad)
    torch.testing.assert_close(loss_a, loss_b, atol=1e-6, rtol=1e-5)
    for (n, pa), (_, pb) in zip(head_a.named_parameters(), head_b.named_parameters()):
        torch.testing.assert_close(pa.grad, pb.grad, atol=1e-6, rtol=1e-5, msg=lambda m, n=n: f"{n}: {m}")
    # a gradient other than the persistent 1 scales everything but the loss value
    head_c = copy.deepcopy(head_b)
    head_c.zero_grad()
    head_c.rng = ops.PhiloxStream(seed=11)
    x_c = x_a.detach().clone().requires_grad_(True)
    head_c.loss_spec = (_lib.LOSS_MSE if loss_kind == "regr" else _lib.LOSS_BCE, y, w)
    loss_c = head_c(x_c)._fragnet_loss[0]
    head_c.loss_spec = None
    (loss_c * 3.0).backward()
    torch.testing.assert_close(x_c.grad, 3.0 * x_b.grad, atol=1e-7, rtol=1e-5)
    torch.testing.assert_close(loss_c, loss_b, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(head_c.predictor[-1].weight.grad, 3.0 * head_b.predictor[-1].weight.grad, atol=1e-6, rtol=1e-5)
