"""The molecule-resident single-pass backward of an attention level (csrc/mol_bwd.hip, fn_gat_bwd_mol_f32 / FN_TUNE_BWD_MOL):
against the two-pass kernels on the same inputs per level, against the oracle through the encoder engine, on padded
static-shape batches, and its refusal of batches that are not molecule-contiguous.  Off by default (it measured slower,
DESIGN.md section 4c); these tests keep it correct."""
import ctypes as C
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None
ATOL = 1e-4
FN_TUNE_BWD_MOL, FN_TUNE_BWD_MOL_FORCE_SLOW = 18, 19


@pytest.fixture
def tune():
    from fragnet_amd import _lib
    touched = []

    def set_(key, value):
        touched.append(key)
        _lib.call("fn_set_tuning", key, value)

    yield set_
    for key in touched:
        _lib.call("fn_set_tuning", key, 0)


@pytest.mark.parametrize("variant", ["fast", "slow", "large_class"])
@pytest.mark.parametrize("profile,n_mols", [("esol", 96), ("synth40", 40)])
def test_single_pass_level_matches_the_two_pass_kernels(profile, n_mols, variant, tune):
    """Every level (bond: folded K = 1 edge term; atom: stored edge term + self loops; fragment-bond: K = 6; fragment) --
    g_h, dz in original edge order, and the reduced partials of dL/da and of the edge-embedding sums."""
    import molbwd_check as mc
    from fragnet_amd import data, synth
    from fragnet_amd.plan import GraphPlan
    if variant != "fast":
        tune(FN_TUNE_BWD_MOL_FORCE_SLOW, 1 if variant == "slow" else 2)
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(n_mols, seed=77, profile=profile)), DEV)
    plan = GraphPlan.from_batch(batch)
    ext = mc.mol_extents(plan, DEV)
    for name, (which, per) in {"bond": (0, 1), "atom": (1, 2), "fbond": (2, 8), "frag": (3, 16)}.items():
        r = mc.run_level(batch, plan, ext, name, which, per, 1, DEV)
        assert r["status"] == 0, r
        assert r["g_h_maxdiff"] < 2e-5 * max(1.0, r["g_h_scale"]), r
        assert r["part_a_maxdiff"] < 1e-5 * max(1.0, r["part_a_scale"]) + 1e-4, r
        if "dz_maxdiff" in r:
            assert r["dz_maxdiff"] < 2e-5, r
        else:
            assert r["part_e_maxdiff"] < 1e-5 * max(1.0, r["part_e_scale"]) + 1e-4, r


@pytest.mark.parametrize("n_layers,n_mols,p_cut,version", [(1, 12, 0.35, "gat2"), (3, 9, 0.35, "gat2"), (4, 1, 0.35, "gat2"), (3, 10, 0.0, "gat2"),
                                                           (3, 7, 1.0, "gat2"), (6, 6, 0.35, "gat2"), (2, 8, 0.35, "gat2_lite"), (2, 8, 0.35, "gat2_edge")])
def test_engine_with_single_pass_backward_matches_the_oracle(n_layers, n_mols, p_cut, version, tune):
    """fn_encoder_backward with FN_TUNE_BWD_MOL = 1 (two launches per layer: the three levels' passes, then the input-gradient
    products + edge-term partials) against the oracle: logits and every gradient; one / several layers, a single molecule,
    one-fragment and fully cut molecules, the gat2_lite and gat2_edge variants."""
    import numpy as np
    from fragnet_amd import data, synth
    from fragnet_amd import model as M
    from oracle import fragnet_ref as ref
    tune(FN_TUNE_BWD_MOL, 1)
    rng = np.random.default_rng(5200 + 10 * n_layers + n_mols)
    batch = data.collate_fn([synth.make_molecule(rng, 10.5, p_cut, 0) for _ in range(n_mols)])
    cfg = dict(n_classes=1, num_layer=n_layers, num_heads=4, drop_ratio=0.0, h1=64, h2=64, h3=64, h4=32, act="relu", edge_features=17)
    if version == "gat2_edge":
        batch["cnx_attr"] = torch.nn.functional.pad(batch["cnx_attr"], (0, 8 - batch["cnx_attr"].shape[1]))
    torch.manual_seed(n_layers)
    gold = ref.FragNetFineTune(**cfg, variant=version).train()
    want = gold(batch)
    torch.nn.functional.mse_loss(want.view(-1), batch["y"]).backward()
    model = M.FragNetFineTune(**cfg, variant=version)
    model.load_state_dict(gold.state_dict())
    model = model.to(DEV).train()
    model.pretrain.use_engine = True
    b = data.batch_to(batch, DEV)
    got = model(b)
    torch.testing.assert_close(got.detach().cpu(), want.detach(), atol=ATOL, rtol=1e-4)
    torch.nn.functional.mse_loss(got.view(-1), b["y"]).backward()
    torch.cuda.synchronize()
    b["_fragnet_plan"].check()
    for (n, p), (_, q) in zip(model.named_parameters(), gold.named_parameters()):
        if q.grad is not None:
            assert p.grad is not None, n
            torch.testing.assert_close(p.grad.cpu(), q.grad, atol=ATOL, rtol=2e-3, msg=lambda m, n=n: f"{n}: {m}")


def test_padded_graph_step_with_single_pass_backward_equals_the_two_pass_step(tune):
    """Static-shape hipGraph step (padding molecules behind the real ones: their rows get zero gradients from extra
    workgroups) with the single-pass backward: loss and the whole flat gradient against the default step."""
    from fragnet_amd import data, graphstep, parallel, synth
    from fragnet_amd.model import FragNetFineTune
    cfg = dict(n_classes=1, num_layer=3, num_heads=4, drop_ratio=0.0, h1=64, h2=128, h3=64, h4=32, act="relu", fthead="FTHead3")
    bs = [data.batch_to(data.collate_fn(synth.synth_molecules(48, seed=500 + i, profile="esol")), DEV) for i in range(3)]
    shapes = graphstep.StaticShapes.from_batches(bs, margin=0.05)
    grads, losses = [], []
    for on in (0, 1):
        tune(FN_TUNE_BWD_MOL, on)
        torch.manual_seed(3)
        model = FragNetFineTune(**cfg).to(DEV).train()
        opt = parallel.FlatAdam.for_live_parameters(
            model, lambda: torch.nn.functional.mse_loss(model(dict(bs[0])).view(-1), bs[0]["y"]).backward(), lr=0.0)
        step = graphstep.GraphedTrainStep(model, opt, shapes, dict(bs[0]), loss="regr")
        losses.append(float(step(dict(bs[1]))))
        torch.cuda.synchronize()
        assert step.replays == 1 and step.fallbacks == 0
        grads.append(opt.grad.detach().clone())
    assert abs(losses[0] - losses[1]) < 1e-6
    torch.testing.assert_close(grads[1], grads[0], atol=ATOL, rtol=2e-3)


def test_batch_that_is_not_molecule_contiguous_sets_the_status_bit(tune):
    """Two molecules' atoms interleaved (not what collate_fn produces): the single-pass backward clamps every index into its
    workgroup's rows (no out-of-bounds access) and flags the plan; plan.check() raises."""
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetFineTune
    tune(FN_TUNE_BWD_MOL, 1)
    batch = data.collate_fn(synth.synth_molecules(6, seed=9, profile="esol"))
    N = batch["x_atoms"].shape[0]
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(N)
    batch["x_atoms"] = batch["x_atoms"][perm]
    batch["batch"] = batch["batch"][perm]
    batch["atom_to_frag_ids"] = batch["atom_to_frag_ids"][perm]
    batch["edge_index"] = inv[batch["edge_index"]]
    b = data.batch_to(batch, DEV)
    model = FragNetFineTune(n_classes=1, num_layer=2, num_heads=4, drop_ratio=0.0, h1=32, h2=32, h3=32, h4=16).to(DEV).train()
    model.pretrain.use_engine = True
    torch.nn.functional.mse_loss(model(b).view(-1), b["y"]).backward()
    torch.cuda.synchronize()
    with pytest.raises(IndexError, match="molecule-contiguous"):
        b["_fragnet_plan"].check()
