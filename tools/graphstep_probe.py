#!/usr/bin/env python3
"""Probe: capture one whole training step (plan + fwd + loss + bwd + grad gather) of a FIXED batch in a hipGraph
and time replays against the eager step.  Answers "what would a static-shape graph step buy" before building it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import fragnet_amd  # noqa: E402
from fragnet_amd import parallel  # noqa: E402
from fragnet_amd.model import FragNetFineTune  # noqa: E402
from fragnet_amd.plan import PLAN_KEY  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fragnet_amd.prefer_rocblas_for_dense_heads()
    batch = bench.make_pool(1, 0, dev)[0]
    model = FragNetFineTune(**bench.MODEL_CFG).to(dev)
    model.train()

    if os.environ.get("PROBE_NO_ENGINE"):
        model.pretrain.use_engine = False

    def fwd_bwd():
        if not os.environ.get("PROBE_KEEP_PLAN"):
            batch.pop(PLAN_KEY, None)
        loss = torch.nn.functional.mse_loss(model(batch).view(-1), batch["y"])
        loss.backward()
        return loss

    opt = parallel.FlatAdam.for_live_parameters(model, fwd_bwd, lr=1e-4)
    rng = model.pretrain.rng
    rng.use_device_counter(dev)

    def eager_step():
        opt.zero_grad()
        loss = fwd_bwd()
        opt.step()
        return loss

    for _ in range(5):
        eager_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eager_step()
    torch.cuda.synchronize()
    print(f"eager step: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            opt.zero_grad()
            fwd_bwd()
            opt.gather_grads()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print("side-stream warm-up ok", flush=True)
    stage = os.environ.get("PROBE_STAGE", "full")
    if stage == "fwd":
        gf = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf):
            batch.pop(PLAN_KEY, None)
            with torch.no_grad():
                out = model(batch)
        torch.cuda.synchronize()
        print("fwd captured", flush=True)
        gf.replay()
        torch.cuda.synchronize()
        print("fwd replay ok", float(out.sum()), flush=True)
        return
    g = torch.cuda.CUDAGraph()
    opt.zero_grad()
    off0 = rng.offset
    with torch.cuda.graph(g):
        loss = fwd_bwd()
        opt.gather_grads()
        if not os.environ.get('PROBE_NO_ADV'):
            rng.advance_device(rng.offset - off0)
    torch.cuda.synchronize()
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("first replay ok", flush=True)
    for i in range(10):
        g.replay()
        torch.cuda.synchronize()
    print("10 synchronised replays ok", flush=True)
    for i in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("10 back-to-back replays ok", flush=True)

    def graph_step():
        g.replay()
        opt.all_reduce()
        opt.steps += 1
        h = opt.hyper
        from fragnet_amd import _lib
        _lib.call("fn_adam_f32", opt.flat.data_ptr(), opt.grad.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(),
                  opt.flat.numel(), float(h["lr"]), float(h["betas"][0]), float(h["betas"][1]), float(h["eps"]),
                  float(h["weight_decay"]), opt.steps, torch.cuda.current_stream(dev).cuda_stream)

    for i in range(5):
        graph_step()
        torch.cuda.synchronize()
        print("graph_step", i, "ok", flush=True)
    t0 = time.perf_counter()
    for _ in range(50):
        graph_step()
    torch.cuda.synchronize()
    print(f"graph step: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms   loss {float(loss):.5f}  rng dev {int(rng.dev)}")
    # host cost of the replay alone
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    t_host = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    print(f"host time per replay call (not synchronised): {t_host:.3f} ms")


if __name__ == "__main__":
    main()
