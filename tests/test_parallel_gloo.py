"""N > 1 path on CPU: world_size-2 gloo processes exercising the flat-bucket gradient exchange, the
sharding helper and the count-weighted loss scale.  The model itself needs a GPU, so a
small stand-in module plays its part; what is tested is the exchange logic bench.py / the trainers use."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fragnet_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l1 = torch.nn.Linear(6, 8)
        self.dead = torch.nn.Linear(3, 3)      # constructed, never used: must stay out of the bucket
        self.l2 = torch.nn.Linear(8, 1)

    def forward(self, x):
        return self.l2(torch.relu(self.l1(x)))


def _net():
    torch.manual_seed(0)
    return _Net()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_distributed("gloo")
    torch.manual_seed(123)
    x = torch.randn(16, 6)
    y = torch.randn(16)
    mine = parallel.shard_indices(16, rank, world)
    model = _net()
    opt = parallel.FlatAdam.for_live_parameters(
        model, lambda: torch.nn.functional.mse_loss(model(x[mine]).view(-1), y[mine]).backward(), lr=1e-2)
    n_live = len(opt.params)
    for it in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(model(x[mine]).view(-1), y[mine]).backward()
        if it == 1:      # the exchange of graphstep.GraphedTrainStep(overlap=True): "head" slice first, the rest after
            opt.gather_grads()
            k = model.l1.weight.numel() + model.l1.bias.numel()
            works = [opt.all_reduce_slice(k, None, async_op=True), opt.all_reduce_slice(0, k, async_op=True)]
            for w in works:
                if w is not None:
                    w.wait()
            opt.apply_gathered(reduced=True)
        else:
            opt.step()
    # count-weighted scale: ranks hold 5 and 11 "atoms"
    scale = parallel.weighted_loss_scale(5 if rank == 0 else 11, torch.device("cpu"))
    q.put((rank, n_live, [p.detach().numpy().tolist() for p in model.parameters()], scale))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch, stock Adam over ALL parameters (dead ones get no grad and are skipped)
    torch.manual_seed(123)
    x = torch.randn(16, 6)
    y = torch.randn(16)
    model = _net()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(model(x).view(-1), y).backward()
        opt.step()
    for rank, n_live, params, scale in res:
        assert n_live == 4                                  # the dead Linear is not in the bucket
        for got, want in zip(params, model.parameters()):
            torch.testing.assert_close(torch.tensor(got), want.detach(), atol=1e-6, rtol=1e-5)
    assert abs(res[0][3] - 5 * 2 / 16) < 1e-12 and abs(res[1][3] - 11 * 2 / 16) < 1e-12


def test_shard_indices_cover_and_balance():
    w = [float((i * 7919) % 13 + 1) for i in range(101)]
    seen = []
    loads = []
    for r in range(4):
        idx = parallel.shard_indices(101, r, 4, weights=w)
        seen += idx
        loads.append(sum(w[i] for i in idx))
    assert sorted(seen) == list(range(101))
    assert max(loads) - min(loads) <= max(w)
    assert parallel.shard_indices(10, 1, 4) == [1, 5, 9]


def test_flat_grad_bucket_aliases_and_survives_backward():
    model = _net()
    x = torch.randn(4, 6)
    bucket = parallel.FlatGradBucket.for_live_parameters(model, lambda: model(x).sum().backward())
    assert len(bucket.params) == 4 and bucket.nbytes == 4 * sum(p.numel() for p in bucket.params)
    bucket.zero()
    model(x).sum().backward()
    assert bucket.intact()
    assert float(bucket.flat.abs().sum()) > 0
