"""Shared loaders for the golden fixtures (tests/golden/*.npz, written by make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    batch = {k[len("batch/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch/")}
    out = {k[len("out/"):]: z[k] for k in z.files if k.startswith("out/")}
    grads = {"full": {k[len("gfull/"):]: z[k] for k in z.files if k.startswith("gfull/")},
             "samp": {k[len("gsamp/"):]: z[k] for k in z.files if k.startswith("gsamp/")},
             "sum": {k[len("gsum/"):]: z[k] for k in z.files if k.startswith("gsum/")}}
    pkeys = json.loads(str(z["pkeys"])) if "pkeys" in z.files else []
    psums = z["psums"] if "psums" in z.files else None
    return cfg, batch, out, grads, pkeys, psums


def check_params_match(model, pkeys, psums, rtol=1e-12):
    """Same seed + same construction order => same weights as the reference had (checked by checksum)."""
    sd = model.state_dict()
    assert list(sd.keys()) == pkeys, "state_dict key order differs from the reference"
    for k, (s, a) in zip(pkeys, psums):
        v = sd[k].double()
        assert abs(float(v.sum()) - s) <= rtol * max(1.0, abs(a)), k
        assert abs(float(v.abs().sum()) - a) <= rtol * max(1.0, abs(a)), k


def sample_like_golden(g):
    g = g.reshape(-1)
    pick = torch.linspace(0, g.numel() - 1, 1024).long()
    return g[pick]


def check_grads(model, grads, atol, rtol):
    live = set(grads["sum"].keys())
    for name, p in model.named_parameters():
        if name not in live:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, f"{name}: reference has no gradient here"
            continue
        assert p.grad is not None, f"{name}: missing gradient"
        g = p.grad.detach().cpu()
        if name in grads["full"]:
            ref = torch.from_numpy(grads["full"][name])
            torch.testing.assert_close(g, ref, atol=atol, rtol=rtol, msg=lambda m: f"{name}: {m}")
        else:
            ref = torch.from_numpy(grads["samp"][name])
            torch.testing.assert_close(sample_like_golden(g), ref, atol=atol, rtol=rtol, msg=lambda m: f"{name}: {m}")
            s, a = grads["sum"][name]
            assert abs(float(g.double().sum()) - s) <= 1e-3 * max(1.0, a), name
