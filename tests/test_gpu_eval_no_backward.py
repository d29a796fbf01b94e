"""fn_encoder.no_backward: an evaluation pass that nobody differentiates (torch.no_grad(), or nothing requires a gradient) stores
nothing for a backward pass -- no attention probabilities, no raw bond rows (include/fragnet_hip.h).  Its outputs are BIT-identical
to those of the evaluation pass that does save them, the latter can still be differentiated (gradients against the training-mode
pass without dropout, the reference's own semantics: gat2.py:381-442 is the same graph in both modes when p = 0), and the library
refuses a backward pass on a descriptor whose forward saved nothing.
"""
from unittest import mock

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _net_and_batch(drop=0.0, n=96, seed=31, variant="gat2"):
    from fragnet_amd import data, model as M, synth
    torch.manual_seed(4)
    kw = dict(n_classes=1, num_layer=3, drop_ratio=drop, h1=32, h2=32, h3=32, h4=32, act="relu", fthead="FTHead3", variant=variant)
    net = M.FragNetFineTune(**kw).to(DEV)
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(n, seed=seed, profile="esol")), DEV)
    if variant == "gat2_edge":      # gat2_edge.py:46 wants 8 connection features, the featuriser writes 6
        batch["cnx_attr"] = torch.nn.functional.pad(batch["cnx_attr"], (0, 8 - batch["cnx_attr"].shape[1]))
    return net, batch


def _encoder_outputs(net, batch):
    batch.pop("_fragnet_plan", None)
    outs = net.pretrain(batch)
    torch.cuda.synchronize()
    return [t for t in outs if t is not None]


@pytest.mark.parametrize("engine_const,variant", [(1, "gat2"), (0, "gat2"), (1, "gat2_lite"), (1, "gat2_edge")])
def test_an_evaluation_pass_without_a_backward_pass_returns_the_same_bits(engine_const, variant):
    from fragnet_amd import _lib
    net, batch = _net_and_batch(variant=variant)
    net.eval()
    try:
        _lib.call("fn_set_tuning", 33, engine_const)
        saved = [t.detach().clone() for t in _encoder_outputs(net, batch)]           # grad mode on, parameters require gradients: everything saved
        batch.pop("_fragnet_plan", None)
        logit_saved = net(batch).detach().clone()
        with torch.no_grad():
            lean = [t.clone() for t in _encoder_outputs(net, batch)]
            batch.pop("_fragnet_plan", None)
            logit_lean = net(batch).clone()
        for p in net.parameters():
            p.requires_grad_(False)
        frozen = [t.clone() for t in _encoder_outputs(net, batch)]                    # grad mode on, nothing requires a gradient
    finally:
        _lib.call("fn_set_tuning", 33, 1)
    assert len(saved) == len(lean) == len(frozen) >= 3
    for a, b, c in zip(saved, lean, frozen):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(logit_saved, logit_lean)
    assert all(not t.requires_grad for t in lean + frozen)


def test_an_evaluation_pass_can_still_be_differentiated():
    """model.eval() with gradients enabled keeps everything a backward pass reads; the gradients equal those of the training-mode pass of a
    model without dropout (same function) within the repo's gradient tolerance |got - ref| <= 1e-4 + 1e-4 |ref|."""
    net, batch = _net_and_batch(drop=0.0)

    def grads(train):
        net.train(train)
        net.zero_grad(set_to_none=True)
        outs = _encoder_outputs(net, batch)
        sum(t.square().mean() for t in outs).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    g_train, g_eval = grads(True), grads(False)
    assert set(g_train) == set(g_eval) and len(g_eval) > 20
    for n in g_train:
        torch.testing.assert_close(g_eval[n], g_train[n], atol=1e-4, rtol=1e-4, msg=lambda s, n=n: f"{n}: {s}")


def test_the_library_refuses_a_backward_pass_after_a_forward_that_saved_nothing():
    from fragnet_amd import _lib, engine
    net, batch = _net_and_batch()
    net.eval()
    batch.pop("_fragnet_plan", None)
    # the host decides from torch's grad mode; lie to it for the forward only, so that autograd still records a backward node
    with mock.patch.object(engine.torch, "is_grad_enabled", return_value=False):
        outs = [t for t in net.pretrain(batch) if t is not None]
    assert any(t.requires_grad for t in outs)
    with pytest.raises(_lib.FragnetHipError, match="no_backward"):
        sum(t.square().mean() for t in outs).backward()
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant", ["gat2", "gat2_lite", "gat2_edge"])
def test_a_finetune_model_without_the_edge_outputs_it_never_reads_is_the_same_model(variant):
    """FragNetFineTune pools atoms and fragments only (gat2.py:816-826): its encoder call passes edge_outputs=False and the library does not
    store the last layer's activated bond / fragment-bond rows (out_bond = out_fbond = NULL).  Logits and every gradient of a training
    step with dropout are BIT-identical to the head applied to the encoder's full result; the two unwanted outputs come back empty."""
    from fragnet_amd import model as M
    net, batch = _net_and_batch(drop=0.1, variant=variant)
    net.train()

    def step(full):
        net.zero_grad(set_to_none=True)
        net.pretrain.rng.offset = 4242
        batch.pop("_fragnet_plan", None)
        if full:
            outs = net.pretrain(batch)                         # all outputs stored
            assert outs[2].shape[0] > 0
            net.fthead.live_rows = None
            logit = net.fthead(M.pooled(outs[0], outs[1], batch))
        else:
            logit = net(batch)
        logit.square().mean().backward()
        torch.cuda.synchronize()
        return logit.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    l_full, g_full = step(True)
    l_lean, g_lean = step(False)
    assert torch.equal(l_full, l_lean)
    assert set(g_full) == set(g_lean) and len(g_lean) > 20
    for n in g_full:
        assert torch.equal(g_full[n], g_lean[n]), n
    batch.pop("_fragnet_plan", None)
    lean = net.pretrain(batch, edge_outputs=False)
    assert lean[2].numel() == 0 and (lean[3] is None or lean[3].numel() == 0) and not lean[2].requires_grad
