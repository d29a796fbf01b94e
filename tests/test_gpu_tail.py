"""The molecule-resident fragment tail (csrc/mol_tail.inc: fragment sums + fragment graph + readout in one launch, its backward
in one more) against the reference's golden vectors, against the separate launches, and on padded static batches.

It runs for batches that carry collate_fn's layout promise (plan.CollatedBatch); a plain dict takes the general kernels."""
import pytest
import torch

from tests.helpers import check_grads, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _fused(model, batch) -> bool:
    """Did the encoder hand out the readout (i.e. did the fused tail run)?"""
    model.pretrain.rng.offset = 0
    with torch.no_grad():
        return hasattr(model.pretrain(batch)[0], "_fragnet_readout")


@pytest.mark.parametrize("case", ["ft_esol_b8", "ft_tox21_b4", "ft_edge_b6"])
def test_fused_tail_matches_reference_golden(case):
    """The reference's own collate output, marked as such: logits, loss and every gradient of the golden fixture (drop 0)."""
    from fragnet_amd import data
    from fragnet_amd.model import FragNetFineTune
    from fragnet_amd.plan import CollatedBatch
    from oracle import fragnet_ref as ref
    cfg, batch, out, grads, pkeys, psums = load_case(case)
    torch.manual_seed(cfg["seed"])
    model = FragNetFineTune(**cfg["ctor"]).to(DEV).train()
    b = data.batch_to(CollatedBatch(batch), DEV)
    assert isinstance(b, CollatedBatch)
    logits = model(b)
    torch.testing.assert_close(logits.detach().cpu(), torch.from_numpy(out["logits"]), atol=ATOL, rtol=1e-4)
    loss = torch.nn.functional.mse_loss(logits.view(-1), b["y"]) if cfg["loss"] == "mse" else ref.finetune_bce_loss(logits, b["y"])
    assert abs(loss.item() - float(out["loss"])) < ATOL
    loss.backward()
    torch.cuda.synchronize()
    check_grads(model, grads, atol=ATOL, rtol=1e-4)
    assert _fused(model, b) and not _fused(model, data.batch_to(dict(batch), DEV))


@pytest.mark.parametrize("path", [1, 2], ids=["lds", "global"])
@pytest.mark.parametrize("heads,drop", [(4, 0.1), (2, 0.0), (8, 0.1)])
def test_fused_tail_equals_the_separate_launches(heads, drop, path):
    """Same weights, same Philox stream: a CollatedBatch (fused tail) and the same tensors as a plain dict (k_frag_tail, k_gat_fwd,
    k_pool_cat; five launches backward) agree to summation-order round-off -- logits and every gradient, with single-fragment,
    fully cut and salt molecules in the batch; both the LDS path and the global-memory path of oversize molecules."""
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetFineTune
    mols = synth.synth_molecules(70, seed=41, profile="esol", p_salt=0.2) + [synth.notebook_molecule()]
    coll = data.batch_to(data.collate_fn(mols), DEV)
    plain = dict(coll)
    torch.manual_seed(3)
    model = FragNetFineTune(n_classes=1, num_layer=2, num_heads=heads, drop_ratio=drop, h1=64, h2=64, h3=64, h4=32, act="relu",
                            fthead="FTHead3").to(DEV).train()
    res = []
    from fragnet_amd import _lib
    try:
        _lib.call("fn_set_tuning", 20, path)        # 2: every molecule on the global-memory path of oversize molecules
        for b in (coll, plain):
            model.zero_grad(set_to_none=True)
            model.pretrain.rng.offset = 77
            out = model(b)
            torch.nn.functional.mse_loss(out.view(-1), b["y"]).backward()
            res.append((out.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        assert _fused(model, coll) and not _fused(model, plain)
    finally:
        _lib.call("fn_set_tuning", 20, 1)
    torch.testing.assert_close(res[0][0], res[1][0], atol=1e-5, rtol=1e-5)
    assert set(res[0][1]) == set(res[1][1])
    for n, g in res[1][1].items():
        torch.testing.assert_close(res[0][1][n], g, atol=1e-5 * max(1.0, float(g.abs().max())), rtol=1e-4, msg=lambda m, n=n: f"{n}: {m}")


def test_fused_tail_with_gradients_on_atoms_fragments_and_readout():
    """FragNetPreTrain reads the atoms (bond-angle tower), the bonds and the readout (graph-energy tower): dL/d(out_atoms) and
    dL/d(readout) both enter the tail's backward.  Fused vs separate launches, every head output and gradient."""
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetPreTrain
    mols = synth.synth_molecules(48, seed=9, profile="esol", pretrain_targets=True)
    coll = data.batch_to(data.collate_fn_pt(mols), DEV)
    plain = dict(coll)
    torch.manual_seed(5)
    model = FragNetPreTrain(num_layer=2, drop_ratio=0.1, edge_features=17).to(DEV).train()
    res = []
    for b in (coll, plain):
        model.zero_grad(set_to_none=True)
        model.pretrain.rng.offset = 5
        outs = model(b)
        sum(o.square().mean() for o in outs if o is not None).backward()
        res.append(([o.detach().clone() for o in outs if o is not None],
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    for a, c in zip(res[0][0], res[1][0]):
        torch.testing.assert_close(a, c, atol=1e-5, rtol=1e-5)
    assert set(res[0][1]) == set(res[1][1])
    for n, g in res[1][1].items():
        torch.testing.assert_close(res[0][1][n], g, atol=1e-5 * max(1.0, float(g.abs().max())), rtol=1e-4, msg=lambda m, n=n: f"{n}: {m}")


def test_fused_tail_on_a_padded_static_batch_matches_the_unpadded_batch():
    """Static-shape batch (graphstep.pad_batch layout: padding items point round-robin at reserved slots, REAL_MOLS_KEY on the
    device): the padding molecules' workgroups write zeros, the real molecules' logits and the parameter gradients equal the
    unpadded run's and everything stays finite."""
    from fragnet_amd import data, graphstep, synth
    from fragnet_amd.model import FragNetFineTune
    from fragnet_amd.plan import CollatedBatch, LIVE_MOLS_KEY, REAL_MOLS_KEY
    mols = synth.synth_molecules(40, seed=23, profile="esol")
    coll = data.batch_to(data.collate_fn(mols), DEV)
    shapes = graphstep.StaticShapes.from_batches([coll], margin=0.1)
    padded = CollatedBatch(graphstep.pad_batch(coll, shapes))
    padded[REAL_MOLS_KEY] = torch.tensor([40], dtype=torch.int32, device=DEV)
    w = padded[graphstep.MASK_KEY]
    torch.manual_seed(11)
    model = FragNetFineTune(n_classes=1, num_layer=2, drop_ratio=0.0, h1=64, h2=64, h3=64, h4=32, act="relu", fthead="FTHead3").to(DEV).train()
    res = []
    for b, rows in ((coll, 40), (padded, 40)):
        model.zero_grad(set_to_none=True)
        out = model(b)
        assert torch.isfinite(out).all()
        y = b["y"].view(-1)
        wt = w if b is padded else torch.ones(40, device=DEV)
        (((out.view(-1) - y) ** 2) * wt).sum().div(40).backward()
        res.append((out.detach()[:rows].clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert _fused(model, padded)
    torch.testing.assert_close(res[1][0], res[0][0], atol=1e-5, rtol=1e-5)
    for n, g in res[0][1].items():
        assert torch.isfinite(res[1][1][n]).all(), n
        torch.testing.assert_close(res[1][1][n], g, atol=1e-5 * max(1.0, float(g.abs().max())), rtol=1e-4, msg=lambda m, n=n: f"{n}: {m}")


def test_feature_width_mismatch_fails_like_the_reference():
    """A model built for 16 bond features on a batch with 17 (FragNetPreTrain's default against the ESOL featuriser): the
    reference's nn.Linear raises a shape error; the engine, which only sees pointers, must not read past the weights."""
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetPreTrain
    b = data.batch_to(data.collate_fn_pt(synth.synth_molecules(6, seed=2, profile="esol", pretrain_targets=True)), DEV)
    model = FragNetPreTrain(num_layer=1, drop_ratio=0.0).to(DEV)       # edge_features = 16
    with pytest.raises(RuntimeError, match="shapes cannot be multiplied"):
        model(b)


def _poison():
    """Fill the caching allocator's free blocks with NaN: whatever a kernel reads without having written it shows up."""
    xs = [torch.full((n,), float("nan"), device=DEV) for n in (64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 22) for _ in range(6)]
    del xs


@pytest.mark.parametrize("tail", [1, 2, 0], ids=["fused_tail_lds", "fused_tail_global", "separate_launches"])
@pytest.mark.parametrize("kind,layers", [("ft", 1), ("ft", 2), ("ft", 4), ("pt", 1), ("pt", 2), ("pt", 4)])
def test_training_step_reads_nothing_it_did_not_write(kind, layers, tail):
    """Workspaces come from torch.empty.  With the allocator's free memory poisoned with NaN before the forward and before the
    backward pass, four runs of the same step give finite, bit-identical gradients -- i.e. no kernel reads a buffer (or a
    masked lane multiplies a row) that this pass has not written.  (Found on the way: a lane without an edge gathered a row of
    another molecule and multiplied it by 0; the molecule-resident tail made that row possibly unwritten.)"""
    from fragnet_amd import _lib, data, synth
    from fragnet_amd.model import FragNetFineTune, FragNetPreTrain
    pt = kind == "pt"
    mols = synth.synth_molecules(48, seed=9, profile="esol", pretrain_targets=pt)
    b = data.batch_to((data.collate_fn_pt if pt else data.collate_fn)(mols), DEV)
    torch.manual_seed(5)
    model = (FragNetPreTrain(num_layer=layers, drop_ratio=0.1, edge_features=17) if pt else
             FragNetFineTune(n_classes=1, num_layer=layers, drop_ratio=0.1, h1=64, h2=64, h3=64, h4=32, act="relu", fthead="FTHead3")).to(DEV).train()
    res = []
    try:
        _lib.call("fn_set_tuning", 20, tail)
        for _ in range(4):
            model.zero_grad(set_to_none=True)
            model.pretrain.rng.offset = 5
            b.pop("_fragnet_plan", None)
            _poison()
            outs = model(b)
            outs = outs if isinstance(outs, tuple) else (outs,)
            loss = sum(o.square().mean() for o in outs if o is not None)
            _poison()
            loss.backward()
            torch.cuda.synchronize()
            res.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        _lib.call("fn_set_tuning", 20, 1)
    for n, g in res[0].items():
        assert torch.isfinite(g).all(), n
        for r in res[1:]:
            assert torch.equal(r[n], g), n
