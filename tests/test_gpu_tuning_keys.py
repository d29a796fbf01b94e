"""The alternative kernels behind the tuning keys (include/fragnet_hip.h FN_TUNE_*) reproduce the default path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _encoder_run(model, batch, offset):
    model.zero_grad(set_to_none=True)
    model.pretrain.rng.offset = offset
    batch.pop("_fragnet_plan", None)
    outs = model.pretrain(batch)
    loss = sum(t.square().mean() for t in outs if t is not None)
    loss.backward()
    torch.cuda.synchronize()
    batch["_fragnet_plan"].check()
    return [t.detach().clone() for t in outs if t is not None], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("key,value", [(8, 0), (7, 0), (14, 0), (20, 0), (29, 1), (29, 2), (22, 0), (33, 0)])
def test_alternative_kernels_behind_tuning_keys_stay_parity_green(key, value):
    """The alternative paths that stay selectable (include/fragnet_hip.h FN_TUNE_*): 8 = 0 the LDS-staged grouped weight-gradient
    kernel, 7 = 0 the separate row-dots launch, 14 = 0 the projection GEMMs as launches of their own instead of riding with the
    attention launches, 20 = 0 the last layer's fragment tail as separate launches instead of the
    molecule-resident pair, 29 = 1 the one-pass backward in its deferred form (no second forward output: the g_s_dst term is added by the
    consumers of g_h; 29 = 2 its mixed form: layer 0 keeps the second output), 22 = 0 the destination + source passes, 33 = 0 the general kernel instances instead of the engine-constant ones (round 6).  Each must reproduce the default path's outputs and gradients (same Philox
    stream) on a training step with dropout."""
    from fragnet_amd import _lib, data, model as M, synth
    torch.manual_seed(0)
    net = M.FragNetFineTune(n_classes=1, num_layer=3, drop_ratio=0.1, h1=32, h2=32, h3=32, h4=32, act="relu", fthead="FTHead3").to(DEV)
    net.train()
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(96, seed=17, profile="esol")), DEV)
    default = {7: 1, 8: 1, 14: 2, 20: 1, 29: 0, 22: 1, 33: 1}[key]
    try:
        o0, g0 = _encoder_run(net, batch, 999)
        _lib.call("fn_set_tuning", key, value)
        o1, g1 = _encoder_run(net, batch, 999)
    finally:
        _lib.call("fn_set_tuning", key, default)
    for a, b in zip(o0, o1):
        torch.testing.assert_close(b, a, atol=1e-5, rtol=1e-5)
    assert set(g0) == set(g1)
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], atol=1e-5, rtol=1e-4, msg=lambda m: f"{n}: {m}")


def test_launch_geometry_keys_of_the_plain_forward_do_not_change_a_bit():
    """FN_TUNE_FWD_BLOCKS_EVAL (10), FN_TUNE_FWD_BLOCKS_EVAL_LARGE (30) and FN_TUNE_FWD_TAIL_ROWS (31) only decide how many rows a
    half-wave of the attention forward walks; a row's arithmetic does not depend on it.  Forced into play on a small batch (8 workgroups
    per level -> many rows per half-wave; then the large-level count and the second level's cap), the inference outputs stay bit-identical."""
    from fragnet_amd import _lib, data, model as M, synth
    torch.manual_seed(1)
    net = M.FragNetFineTune(n_classes=1, num_layer=3, drop_ratio=0.1, h1=32, h2=32, h3=32, h4=32, act="relu", fthead="FTHead3").to(DEV)
    net.eval()
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(96, seed=23, profile="esol")), DEV)

    def run():
        batch.pop("_fragnet_plan", None)
        with torch.no_grad():
            out = net(batch)
        torch.cuda.synchronize()
        return out.detach().clone()

    base = run()
    try:
        for settings in ({10: 8, 30: 0, 31: 0}, {10: 8, 30: 64, 31: 0}, {10: 8, 30: 0, 31: 1}, {10: 8, 30: 64, 31: 1}):
            for k, v in settings.items():
                _lib.call("fn_set_tuning", k, v)
            assert torch.equal(run(), base), settings
    finally:
        for k, v in {10: 1792, 30: 6144, 31: 1}.items():
            _lib.call("fn_set_tuning", k, v)


def test_engine_constant_instances_do_not_change_a_bit():
    """FN_TUNE_ENGINE_CONST (33): the attention launches of the engine run kernel instances whose uniform run-time flags are
    compile-time constants (forward kinds 2 / 3, the one-pass backward's EN instances).  Same arithmetic: an inference pass and a
    training step with dropout (outputs and every gradient) are BIT-identical with the key off."""
    from fragnet_amd import _lib, data, model as M, synth
    torch.manual_seed(2)
    net = M.FragNetFineTune(n_classes=1, num_layer=3, drop_ratio=0.1, h1=32, h2=32, h3=32, h4=32, act="relu", fthead="FTHead3").to(DEV)
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(96, seed=29, profile="esol")), DEV)

    def infer():
        net.eval()
        batch.pop("_fragnet_plan", None)
        with torch.no_grad():
            out = net(batch)
        torch.cuda.synchronize()
        return out.detach().clone()

    def train():
        net.train()
        return _encoder_run(net, batch, 777)

    try:
        _lib.call("fn_set_tuning", 33, 0)
        e0, (o0, g0) = infer(), train()
        _lib.call("fn_set_tuning", 33, 1)
        e1, (o1, g1) = infer(), train()
    finally:
        _lib.call("fn_set_tuning", 33, 1)
    assert torch.equal(e0, e1)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert set(g0) == set(g1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
