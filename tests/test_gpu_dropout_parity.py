"""Oracle-level parity WITH dropout on: the HIP path's masks are regenerated from its Philox stream
(fn_dropout_act_f32 on ones with the recorded seed / offsets), injected into the oracle in the reference's call order
(gat2.py:396-397, 414-418, 436-440; head :721-722, :668-675) and loss, logits and every live gradient must agree to 1e-4.
Covers the engine and the captured static-shape graph step (the path bench.py times, drop 0.1), FTHead3 (Philox in the fused
head) and FTHead4 (dropout BEFORE dense).  The backward relies on "saved relu(dropout(x)) > 0 encodes the mask" in four
epilogues -- this is the check that it equals the reference's semantics, not just another HIP path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None
ATOL = 1e-4


def philox_mask(numel, p, seed, offset):
    """mask / (1 - p) of the HIP dropout stream for a tensor of ``numel`` elements drawn at (seed, offset)."""
    from fragnet_amd import _lib
    ones = torch.ones(numel, dtype=torch.float32, device=DEV)
    y = torch.empty_like(ones)
    _lib.call("fn_dropout_act_f32", ones.data_ptr(), y.data_ptr(), numel, float(p), seed, offset, None, 0,
              torch.cuda.current_stream(DEV).cuda_stream)
    return y


class TakeLog:
    """Records every PhiloxStream.take of a model's stream."""

    def __init__(self, rng):
        self.rng, self.calls, self._orig = rng, [], rng.take

    def __enter__(self):
        def take(numel):
            seed, off = self._orig(numel)
            self.calls.append((seed, off, numel))
            return seed, off
        self.rng.take = take
        return self

    def __exit__(self, *exc):
        self.rng.take = self._orig


def encoder_masks(call, p, counts, n_layers, k_atom0, rows_real, extra_offset=0):
    """The engine draws one block range for the whole encoder (fn_encoder_rng_blocks): dropout(x_atoms), then per layer the
    atoms, fragments, bond nodes and fragment-bond nodes outputs.  Returns the oracle's call-order list: x_atoms, x_frags (dead:
    None), then per layer atoms, frags, bond, fbond -- cut to the real rows (padding rows are a suffix)."""
    seed, off, _ = call
    off += extra_offset
    N, F, E, EF = counts                       # rows of the (possibly padded) tensors the masks were drawn for
    n, f, e, ef = rows_real
    blocks = lambda numel: (numel + 3) // 4
    out = [philox_mask(N * k_atom0, p, seed, off).view(N, k_atom0)[:n].cpu(), None]
    off += blocks(N * k_atom0)
    for _ in range(n_layers):
        for rows, real in ((N, n), (F, f), (E, e), (EF, ef)):
            out.append(philox_mask(rows * 128, p, seed, off).view(rows, 128)[:real].cpu())
            off += blocks(rows * 128)
    return out


def head3_masks(calls, p, dims, rows, rows_real, extra_offset=0):
    return [philox_mask(rows * d, p, seed, off + extra_offset).view(rows, d)[:rows_real].cpu() for (seed, off, _), d in zip(calls, dims)]


def _compare(model_grads, gold, logits, want_logits, loss, want_loss):
    torch.testing.assert_close(logits, want_logits, atol=ATOL, rtol=1e-4)
    assert abs(loss - want_loss) < ATOL, (loss, want_loss)
    checked = 0
    for name, q in gold.named_parameters():
        if q.grad is None:
            continue
        torch.testing.assert_close(model_grads[name], q.grad, atol=ATOL, rtol=1e-4, msg=lambda m, name=name: f"{name}: {m}")
        checked += 1
    return checked


@pytest.mark.parametrize("case", ["slice64", "ft_esol_b8"])
def test_engine_with_dropout_matches_the_oracle_under_the_same_masks(case):
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetFineTune
    from oracle import fragnet_ref as ref
    from tests.helpers import load_case
    p = 0.1
    if case == "slice64":
        cfg = dict(n_classes=1, num_layer=4, num_heads=4, drop_ratio=p, h1=128, h2=256, h3=256, h4=64, act="relu", fthead="FTHead3")
        batch = data.collate_fn(synth.synth_molecules(64, seed=1000, profile="esol"))
        torch.manual_seed(0)
        gold = ref.FragNetFineTune(**cfg)
    else:
        c, batch, _, _, _, _ = load_case("ft_esol_b8")
        cfg = dict(c["ctor"], drop_ratio=p)
        torch.manual_seed(c["seed"])
        gold = ref.FragNetFineTune(**cfg)
    gold.train()
    model = FragNetFineTune(**cfg)
    model.load_state_dict(gold.state_dict())
    model = model.to(DEV).train()
    model.pretrain.rng.seed = 0x1234567
    b = data.batch_to(batch, DEV)
    with TakeLog(model.pretrain.rng) as log:
        logits = model(b)
        loss = torch.nn.functional.mse_loss(logits.view(-1), b["y"])
        loss.backward()
    torch.cuda.synchronize()
    N, F = b["x_atoms"].shape[0], b["x_frags"].shape[0]
    E, EF = b["node_features_bonds"].shape[0], b["node_features_fbonds"].shape[0]
    L = cfg["num_layer"]
    assert len(log.calls) == 1 + 4                            # one draw for the encoder, one per hidden layer of FTHead3
    dims = [cfg["h1"], cfg["h2"], cfg["h3"], cfg["h4"]]
    B = b["y"].shape[0]
    masks = encoder_masks(log.calls[0], p, (N, F, E, EF), L, b["x_atoms"].shape[1], (N, F, E, EF)) + head3_masks(log.calls[1:], p, dims, B, B)
    kept = torch.cat([m.reshape(-1) for m in masks if m is not None])
    assert 0.85 < float((kept > 0).float().mean()) < 0.95     # these really are p = 0.1 masks
    inj = ref.inject_dropout(gold, masks)
    want = gold(batch)
    want_loss = torch.nn.functional.mse_loss(want.view(-1), batch["y"])
    want_loss.backward()
    assert inj.cursor == len(masks)
    grads = {n: q.grad.detach().cpu() for n, q in model.named_parameters() if q.grad is not None}
    assert _compare(grads, gold, logits.detach().cpu(), want.detach(), float(loss), float(want_loss)) >= 40


def test_graph_step_with_dropout_matches_the_oracle_under_the_same_masks():
    """The benchmarked path: static-shape staging + whole-step hipGraph replay at drop 0.1.  The masks of the replay are the
    offsets baked in at capture plus the device counter, drawn over the PADDED tensors; the oracle runs the unpadded batch."""
    from fragnet_amd import data, graphstep, parallel, synth
    from fragnet_amd.model import FragNetFineTune
    from oracle import fragnet_ref as ref
    p = 0.1
    cfg = dict(n_classes=1, num_layer=3, num_heads=4, drop_ratio=p, h1=128, h2=256, h3=128, h4=64, act="relu", fthead="FTHead3")
    cpu_batches = [data.collate_fn(synth.synth_molecules(64, seed=700 + i, profile="esol")) for i in range(2)]
    bs = [data.batch_to(cb, DEV) for cb in cpu_batches]
    torch.manual_seed(0)
    gold = ref.FragNetFineTune(**cfg).train()
    model = FragNetFineTune(**cfg)
    model.load_state_dict(gold.state_dict())
    model = model.to(DEV).train()
    model.pretrain.rng.seed = 0x7654321
    opt = parallel.FlatAdam.for_live_parameters(
        model, lambda: torch.nn.functional.mse_loss(model(dict(bs[0])).view(-1), bs[0]["y"]).backward(), lr=0.0)
    shapes = graphstep.StaticShapes.from_batches(bs, margin=0.05)
    with TakeLog(model.pretrain.rng) as log:
        step = graphstep.GraphedTrainStep(model, opt, shapes, dict(bs[0]), loss="regr")
    captured = log.calls[-5:]                                 # the draws made inside the capture (the last forward of __init__)
    loss = float(step(dict(bs[1])))
    torch.cuda.synchronize()
    assert step.replays == 1 and step.fallbacks == 0
    counter = int(step._counters[0].item())                   # what the replay added to every baked-in offset
    cap = shapes.cap
    real = tuple(int(bs[1][k].shape[0]) for k in ("x_atoms", "x_frags", "node_features_bonds", "node_features_fbonds"))
    masks = encoder_masks(captured[0], p, (cap["atom"], cap["frag"], cap["edge"], cap["fedge"]), cfg["num_layer"],
                          bs[1]["x_atoms"].shape[1], real, extra_offset=counter)
    B = int(bs[1]["y"].shape[0])
    rows = captured[1][2] // cfg["h1"]                        # molecule rows the head drew for (capacity)
    masks += head3_masks(captured[1:], p, [cfg["h1"], cfg["h2"], cfg["h3"], cfg["h4"]], rows, B, extra_offset=counter)
    inj = ref.inject_dropout(gold, masks)
    want = gold(cpu_batches[1])
    want_loss = torch.nn.functional.mse_loss(want.view(-1), cpu_batches[1]["y"])
    want_loss.backward()
    assert inj.cursor == len(masks)
    assert abs(loss - float(want_loss)) < ATOL, (loss, float(want_loss))
    gold_params = dict(gold.named_parameters())
    checked = 0
    for name, q in model.named_parameters():
        slot = getattr(q, "_fn_grad_slot", None)
        if slot is None or gold_params[name].grad is None:
            continue
        flat, off = slot
        got = flat[off: off + q.numel()].view(q.shape).cpu()
        torch.testing.assert_close(got, gold_params[name].grad, atol=ATOL, rtol=1e-4, msg=lambda m, name=name: f"{name}: {m}")
        checked += 1
    assert checked >= 40


def test_fthead4_dropout_before_dense_matches_the_oracle_under_the_same_masks():
    """FTHead4 (gat2.py:640-675: dropout -> dense -> act -> dropout -> out_proj) on the Tox21-shape golden batch: the encoder's
    masks come from the Philox stream, the head's two (torch dropout in this head) are injected on both sides."""
    from fragnet_amd import data
    from fragnet_amd.model import FragNetFineTune
    from oracle import fragnet_ref as ref
    from tests.helpers import load_case
    p = 0.1
    c, batch, _, _, _, _ = load_case("ft_tox21_b4")
    cfg = dict(c["ctor"], drop_ratio=p)
    assert cfg["fthead"] == "FTHead4"
    torch.manual_seed(c["seed"])
    gold = ref.FragNetFineTune(**cfg).train()
    model = FragNetFineTune(**cfg)
    model.load_state_dict(gold.state_dict())
    model = model.to(DEV).train()
    model.pretrain.rng.seed = 0xABCDEF
    b = data.batch_to(batch, DEV)
    B = b["y"].shape[0]
    g = torch.Generator().manual_seed(5)
    head_masks = [(torch.rand(B, 256, generator=g) >= p).float() / (1 - p), (torch.rand(B, cfg["h1"], generator=g) >= p).float() / (1 - p)]
    ref.inject_dropout(model.fthead, [m.to(DEV) for m in head_masks])           # the same hook class works on GPU tensors
    with TakeLog(model.pretrain.rng) as log:
        logits = model(b)
        loss = ref.finetune_bce_loss(logits, b["y"])
        loss.backward()
    torch.cuda.synchronize()
    assert len(log.calls) == 1
    N, F = b["x_atoms"].shape[0], b["x_frags"].shape[0]
    E, EF = b["node_features_bonds"].shape[0], b["node_features_fbonds"].shape[0]
    masks = encoder_masks(log.calls[0], p, (N, F, E, EF), cfg["num_layer"], b["x_atoms"].shape[1], (N, F, E, EF)) + head_masks
    inj = ref.inject_dropout(gold, masks)
    want = gold(batch)
    want_loss = ref.finetune_bce_loss(want, batch["y"])
    want_loss.backward()
    assert inj.cursor == len(masks)
    grads = {n: q.grad.detach().cpu() for n, q in model.named_parameters() if q.grad is not None}
    assert _compare(grads, gold, logits.detach().cpu(), want.detach(), float(loss), float(want_loss)) >= 30
