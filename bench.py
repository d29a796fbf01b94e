#!/usr/bin/env python3
"""Headline benchmark: molecules/s of one full training step of FragNetFineTune on ESOL-shape batches
of 512 molecules per GPU (BASELINE.json configs[1]; `e1pt4.yaml` model: 4 layers, 4 heads, emb 128,
FTHead3 128/1024/1024/512, drop 0.1, fp32), synthetic data already resident in HBM.

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts the N ranks itself, as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
    python bench.py --gpus 1 --spawn                        (one child, 1-rank RCCL group: the N>1 step sequence on one GPU)

A step = graph plan build + zero grads + forward + MSE loss + backward + (N>1: one flat-bucket
gradient all-reduce over RCCL) + Adam step, on a batch the model has not seen in the previous step
(a pool of pre-collated batches, seeds 1000+i).  N > 1 (`--scaling both`, the default): the headline is STRONG scaling --
the GLOBAL batch is 512 molecules (what SURVEY.md §8d quotes the metric on), split over the ranks by
parallel.shard_indices (rank r owns molecules r::N of the same 512) -- and the same line carries `weak` (every rank owns 512
molecules per step), `overlap_on` (two-graph step, the head's gradient exchange beside the encoder backward),
`allreduce_alone` (the flat-buffer all-reduce by itself), `nccl_ranks` and the fastest rank's step time; `--scaling weak` /
`strong` measure only that one.  At N = 1 the two are the same run.

Default mode "graph": the batch is staged into fixed-capacity buffers (one kernel; the few % of padding are
disconnected dummy rows, see fragnet_amd/graphstep.py) and plan + forward + loss + backward + gradient gather
replay as ONE hipGraph; all-reduce and Adam follow on the stream.  `--eager` runs the same step launch by launch
(hipGraph only around the prediction head).  Both count only the 512 real molecules per step.

Besides the contract fields the JSON line carries
  roofline      for the dominant scatter work (bond-graph level, the largest of the four): algorithmic bytes (SURVEY.md
                §8d closed forms B_agg / B_agg' on this batch's n, m -- the backward counted ONCE for its two passes
                together) / launch time, measured here with HIP events on the launch stream over back-to-back launches
                on a resident batch ("standalone"); "in_graph" uses the per-kernel durations of the replayed step from the
                committed rocprof summary named in `in_graph.source`; `traffic` comes from the PMC file named in
                `traffic_source` (not collected in this run)
  cpu_baseline  the oracle (reference-faithful pure-torch restatement) timed on this host's cores (BASELINE.md §3:
                1 thread and all cores, 2 warm-up + 5 timed steps, median, CPU model named).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODEL_CFG = dict(n_classes=1, atom_features=167, frag_features=167, edge_features=17, num_layer=4, num_heads=4,
                 drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu", emb_dim=128, fthead="FTHead3")
PER_GPU_BATCH = 512
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec peak


def level_bytes(n, m, H=4, D=128):
    """SURVEY.md §8d compulsory-traffic model for one attention level, fp32 + int32 indices."""
    fwd = 4 * ((n + 1) + m + m * H + 2 * n * H + n * D + n * D + m * H)
    bwd = 4 * (2 * n * D + 2 * m * H + 2 * m + n * D + m * H + 2 * n * H)
    return fwd, bwd


def step_bytes(batch, variant="gat2"):
    """Whole-step algorithmic bytes (4 layers x 4 levels fwd+bwd + L3 + pooling + projections), §8d.  ``variant``: gat2_lite has no
    fragment-bond and no fragment graph (its layers stop after the atom -> fragment sum), gat2_edge no fragment-bond graph."""
    N = batch["x_atoms"].shape[0]
    E = batch["node_features_bonds"].shape[0]
    Eb = batch["edge_index_bonds_graph"].shape[1]
    F = batch["x_frags"].shape[0]
    EF = batch["node_features_fbonds"].shape[0]
    EFB = batch["edge_index_fbonds"].shape[1]
    B = batch["y"].shape[0]
    D, H = 128, 4
    total = 0
    for layer in range(4):
        k_b, k_a, k_fb = (17, 167, 6) if layer == 0 else (128, 128, 128)
        levels = ((E, Eb, k_b), (N, E + N, k_a), (EF, EFB, k_fb), (F, EF, None))
        if variant == "gat2_lite":
            levels = levels[:2]
        elif variant == "gat2_edge":
            levels = (levels[0], levels[1], levels[3])
        for n, m, K in levels:
            f, b = level_bytes(n, m)
            total += f + b
            if K is not None:
                total += 4 * (n * K + n * D) + 4 * (n * D + 2 * n * K)      # projection fwd + bwd
            total += 2 * 4 * 2 * n * H                                       # node-scalar epilogue fwd+bwd
        total += 2 * 4 * (E * D + (E + N) * H)                               # full-width edge term of the atom graph (L2)
        if variant == "gat2":
            total += 2 * 4 * (EF * D + EF * H)                               # ... and of the fragment graph (L4b)
        total += 2 * 4 * (N * D + N + F * D)                                  # L3 each way
    total += 2 * 4 * (N * D + N + B * D) + 2 * 4 * (F * D + F + B * D)       # pooling each way
    return total


def forward_bytes(batch):
    """The forward half of step_bytes (eval mode: no dropout): 4 layers x 4 levels, projections, node scalars, edge terms, L3, readout."""
    N = batch["x_atoms"].shape[0]
    E = batch["node_features_bonds"].shape[0]
    Eb = batch["edge_index_bonds_graph"].shape[1]
    F = batch["x_frags"].shape[0]
    EF = batch["node_features_fbonds"].shape[0]
    EFB = batch["edge_index_fbonds"].shape[1]
    B = batch["y"].shape[0]
    D, H = 128, 4
    total = 0
    for layer in range(4):
        k_b, k_a, k_fb = (17, 167, 6) if layer == 0 else (128, 128, 128)
        for n, m, K in ((E, Eb, k_b), (N, E + N, k_a), (EF, EFB, k_fb), (F, EF, None)):
            total += level_bytes(n, m)[0]
            if K is not None:
                total += 4 * (n * K + n * D)                                  # projection
            total += 4 * 2 * n * H                                           # node-scalar epilogue
        total += 4 * (E * D + (E + N) * H) + 4 * (EF * D + EF * H)           # full-width edge terms (L2, L4b)
        total += 4 * (N * D + N + F * D)                                      # L3
    total += 4 * (N * D + N + B * D) + 4 * (F * D + F + B * D)               # readout
    return total


def batch_bytes(batch):
    """Bytes of a collated batch (every tensor once): what a collate reads from the store and writes."""
    return sum(v.numel() * v.element_size() for v in batch.values() if torch.is_tensor(v))


def sweep_roofline(alg_bytes, seconds_per_step, what):
    gbps = alg_bytes / seconds_per_step / 1e9
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
            "algorithmic_bytes_per_step": int(alg_bytes), "what": what}


def make_pool(n_batches, rank, device, batch=PER_GPU_BATCH, world=1, scaling="weak"):
    """weak: ``batch`` molecules per rank, rank-specific seeds.  strong: the same ``batch`` molecules on every rank (seeds do
    not depend on the rank), of which this rank collates its shard parallel.shard_indices(batch, rank, world)."""
    from fragnet_amd import data, parallel, synth
    pool = []
    for i in range(n_batches):
        if scaling == "strong" and world > 1:
            mols = synth.synth_molecules(batch, seed=1000 + i, profile="esol")
            mols = [mols[j] for j in parallel.shard_indices(len(mols), rank, world)]
        else:
            mols = synth.synth_molecules(batch, seed=1000 + 97 * rank + i, profile="esol")
        pool.append(data.batch_to(data.collate_fn(mols), device))
    return pool


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """Physical cores of the host (distinct (physical id, core id) pairs of /proc/cpuinfo); falls back to os.cpu_count()."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(budget_s=150.0):
    """The oracle's full training step (reference-faithful materialised form, oracle/fragnet_ref.py) on this host's cores, on the
    metric's own batch: ESOL-shape, 512 molecules.  BASELINE.md section 3: warm-up + up to 5 timed steps at all PHYSICAL cores, at 32,
    at 8 and at one thread (torch's intra-op threading loses to a single core at all cores on this memory-bound path: the counts in
    between say where the best is); `value` is the fastest count's median, `cores` the threads it used.  The leg is bounded
    (budget_s, shared out evenly): a thread count whose steps do not fit reports the timed steps it did."""
    from fragnet_amd import data, synth
    from oracle import fragnet_ref as ref
    torch.manual_seed(0)
    model = ref.FragNetFineTune(n_classes=1, num_layer=4, drop_ratio=0.1, h1=128, h2=1024, h3=1024, h4=512, act="relu", fthead="FTHead3")
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step(batch):
        opt.zero_grad()
        loss = ref.finetune_regr_loss(model(batch), batch["y"])
        loss.backward()
        opt.step()

    t_all = time.perf_counter()
    default_threads, phys = torch.get_num_threads(), _physical_cores()
    batch = data.collate_fn(synth.synth_molecules(PER_GPU_BATCH, seed=1000, profile="esol"))
    runs = {}
    counts = sorted({c for c in (phys, 32, 8, 1) if 1 <= c <= phys}, reverse=True)
    shares = {c: (i + 1) / len(counts) for i, c in enumerate(counts)}        # cumulative share of the time bound each count may use up
    for threads in counts:
        share = shares[threads]
        torch.set_num_threads(threads)
        stop_at = t_all + budget_s * share
        t0 = time.perf_counter()
        step(batch)
        first = time.perf_counter() - t0
        fit = int(max(0.0, stop_at - time.perf_counter()) / first)         # further steps of that length the share still holds
        warm = 2 if fit >= 6 else (1 if fit >= 2 else 0)                   # the first step is a warm-up step when any fit
        times = [] if warm else [first]
        for _ in range(warm - 1):
            step(batch)
        for _ in range(min(5, fit - (warm - 1)) if warm else 0):
            t0 = time.perf_counter()
            step(batch)
            times.append(time.perf_counter() - t0)
        med = statistics.median(times)
        runs[f"threads_{threads}"] = {"threads": threads, "warmup_steps": warm, "timed_steps": len(times), "median_s_per_step": round(med, 3),
                                      "min_s_per_step": round(min(times), 3), "molecules_per_s": round(PER_GPU_BATCH / med, 2)}
    torch.set_num_threads(default_threads)
    best = max(runs.values(), key=lambda r: r["molecules_per_s"])
    return {"value": best["molecules_per_s"], "unit": "molecules/s", "cores": best["threads"], "kind": "port",
            "timed_steps": best["timed_steps"],
            "sample": f"full training steps (forward + MSE + backward + Adam) of ONE ESOL-shape batch of {PER_GPU_BATCH} molecules, the metric's own "
                      f"workload: at {', '.join(str(c) for c in counts)} threads ({phys} = physical cores), up to 2 warm-up + 5 timed steps each "
                      f"inside a {budget_s:.0f}-s bound shared out evenly (see `runs` for what fitted); value = the fastest thread count's median; "
                      f"oracle/fragnet_ref.py, torch {torch.__version__} CPU",
            "runs": runs, "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "physical_cores": phys}


def kernel_roofline(batch, model, iters=50, alternatives=False):
    """Dominant scatter kernels at the bond-graph level, timed back to back with HIP events on the launch stream: the pair the engine
    runs by default -- the training forward with its second output (`k_gat_fwd(+out2)`) and the one-pass backward (`k_gat_bwd_one`).
    ``alternatives`` (the dev loop, --kernels-only): also the plain forward, the dots kernel, the deferred form of the pass
    (FN_TUNE_DEFER_GSD, HISTORY.md 4h) and the general two-pass backward."""
    import ctypes as C
    from fragnet_amd import _lib
    from fragnet_amd.plan import GraphPlan, _stream_ptr
    dev = batch["x_atoms"].device
    plan = GraphPlan.from_batch(batch)
    lv = plan.levels["bond"]
    layer = model.pretrain.layers[1]
    n, m, H = lv.n, lv.m, 4
    g = torch.Generator(device="cpu").manual_seed(0)
    h = torch.randn(n, 128, generator=g).to(dev)
    gout = torch.randn(n, 128, generator=g).to(dev)
    att = layer.a_b.detach().contiguous()
    embW, embb = layer.edge_attr_bond_embed.weight.detach().contiguous(), layer.edge_attr_bond_embed.bias.detach().contiguous()
    x = plan.sorted_attr("bond", batch["edge_attr_bonds"])
    et = _lib.EdgeTerm(2, 1, 32, 32, None, x.data_ptr(), embW.data_ptr(), embb.data_ptr())
    f32 = dict(dtype=torch.float32, device=dev)
    s_dst, s_src = torch.empty(n, H, **f32), torch.empty(n, H, **f32)
    out, p_sorted, dz = torch.empty(n, 128, **f32), torch.empty(m, H, **f32), torch.empty(m, H, **f32)
    g_s_dst, g_h = torch.empty(n, H, **f32), torch.empty(n, 128, **f32)
    pz = torch.empty(H, m, 2, **f32)
    part_e, part_a = torch.empty(4096, H * 2, **f32), torch.empty(4096, 256, **f32)
    n_e, n_a = C.c_int(0), C.c_int(0)
    st = _stream_ptr(dev)
    _lib.call("fn_node_scalars_f32", h.data_ptr(), att.data_ptr(), 96, 0, 64, s_dst.data_ptr(), s_src.data_ptr(), n, H, st)

    def fwd():
        _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), 96, C.byref(et),
                  C.byref(lv.c), 0.2, out.data_ptr(), p_sorted.data_ptr(), None, None, None, 0, None, H, st)

    def bwd_dst():
        _lib.call("fn_gat_bwd_dst_f32", gout.data_ptr(), h.data_ptr(), p_sorted.data_ptr(), C.byref(et), C.byref(lv.c), 0.2,
                  dz.data_ptr(), None, pz.data_ptr(), g_s_dst.data_ptr(), part_e.data_ptr(), C.byref(n_e), H, st)

    def bwd_src():
        _lib.call("fn_gat_bwd_src_f32", gout.data_ptr(), h.data_ptr(), pz.data_ptr(), g_s_dst.data_ptr(),
                  att.data_ptr(), 96, 0, 64, C.byref(lv.c), g_h.data_ptr(), part_a.data_ptr(), C.byref(n_a), H, st)

    # the engine's default backward: ONE source-owner pass (csrc/gat_bwd_one.inc).  It needs the forward's second output (out2, sigma:
    # `k_gat_fwd(+out2)` is that forward) and the two node-local dots c, g_s_dst, which inside the step ride in the epilogue of the
    # input-gradient GEMM that produces the gradient rows; `k_gat_cu` is the stand-alone kernel for them (last layer / operator path).
    # The deferred form (dz_em) needs neither out2 nor g_s_dst: its consumers add the term (HISTORY.md 4h)
    out2, sigma, p_em = torch.empty(n, 128, **f32), torch.empty(n, H, **f32), torch.empty(m, H, **f32)
    cdot, g_s_dst1, g_h1 = torch.empty(n, H, **f32), torch.empty(n, H, **f32), torch.empty(n, 128, **f32)
    part_e1, part_a1 = torch.empty(4096, H * 2, **f32), torch.empty(4096, 256, **f32)
    dz_em = torch.empty(m, H, **f32)
    x_src = torch.empty(1, m, **f32)
    _lib.call("fn_sort_edge_attr_src_f32", batch["edge_attr_bonds"].contiguous().data_ptr(), 1, C.byref(lv.c), x_src.data_ptr(), st)
    et1 = _lib.EdgeTerm(2, 1, 32, 32, None, x.data_ptr(), embW.data_ptr(), embb.data_ptr(), x_src.data_ptr())
    n_e1, n_a1 = C.c_int(0), C.c_int(0)

    def fwd_em():          # the training forward: probabilities edge-major, as the one-pass backward gathers them
        _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), 96, C.byref(et),
                  C.byref(lv.c), 0.2, out.data_ptr(), p_em.data_ptr(), None, None, None, 1, None, H, st)

    def fwd_o2():
        _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), 96, C.byref(et),
                  C.byref(lv.c), 0.2, out.data_ptr(), p_em.data_ptr(), None, out2.data_ptr(), sigma.data_ptr(), 1, None, H, st)

    def cu():
        _lib.call("fn_gat_cu_f32", gout.data_ptr(), out.data_ptr(), out2.data_ptr(), sigma.data_ptr(), 1.0, cdot.data_ptr(),
                  g_s_dst1.data_ptr(), n, H, st)

    def bwd_one(deferred=True):
        _lib.call("fn_gat_bwd_one_f32", gout.data_ptr(), h.data_ptr(), p_em.data_ptr(), cdot.data_ptr(), g_s_dst1.data_ptr(), C.byref(et1),
                  att.data_ptr(), 96, 0, 64, C.byref(lv.c), 0.2, g_h1.data_ptr(), None, None, part_a1.data_ptr(), C.byref(n_a1),
                  part_e1.data_ptr(), C.byref(n_e1), 1, dz_em.data_ptr() if deferred else None, H, st)

    D = 128
    fwd_b, bwd_b = level_bytes(n, m, H, D)         # B_agg (forward), B_agg' (the WHOLE backward of the level)
    res = {}
    fwd_o2()
    cu()           # (cdot / g_s_dst hold real dots for every variant below)
    runs = [("k_gat_fwd(+out2)", fwd_o2, fwd_b), ("k_gat_bwd_one", lambda: bwd_one(False), bwd_b)]
    if alternatives:
        # B_agg' counts g_out and h once.  The two passes both read them, so per-pass figures are only a split of B_agg' for
        # orientation (dst pass: g_out, h, probs, idx -> dz, g_s_dst; src pass: the rest).  The dots kernel reads three row tables and
        # writes two [n, H] tables (not part of B_agg').
        bwd_dst_b = 4 * (2 * n * D + m * H + m + m * H + n * H)
        cu_b = 4 * (3 * n * D + n * H + 2 * n * H)
        fwd()
        runs += [("k_gat_fwd", fwd_em, fwd_b), ("k_gat_bwd_one(deferred form)", lambda: bwd_one(True), bwd_b), ("k_gat_cu", cu, cu_b),
                 ("k_gat_bwd_dst", bwd_dst, bwd_dst_b), ("k_gat_bwd_src", bwd_src, bwd_b - bwd_dst_b)]
    for name, fn, nbytes in runs:
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1000.0 / iters
        res[name] = {"us_per_launch": round(us, 2), "algorithmic_bytes": nbytes, "GBps": round(nbytes / us / 1e3, 1),
                     "n": n, "m": m}
    if alternatives:
        us_bwd = res["k_gat_bwd_dst"]["us_per_launch"] + res["k_gat_bwd_src"]["us_per_launch"]
        res["k_gat_bwd(dst+src)"] = {"us_per_launch": round(us_bwd, 2), "algorithmic_bytes": bwd_b, "GBps": round(bwd_b / us_bwd / 1e3, 1),
                                     "n": n, "m": m}
    return res

def forward_sweep(rank, world, dev, args):
    """Forward-only (eval mode) molecules/s on synthetic 40-atom / 12-fragment molecules: every
    rank runs its own shard, no collective at all.  Plan build + forward per step, batches resident in HBM."""
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetFineTune
    from fragnet_amd.plan import PLAN_KEY
    torch.manual_seed(0)
    model = FragNetFineTune(**MODEL_CFG).to(dev).eval()
    for B in [int(v) for v in args.sweep_batches.split(",")]:
        pool = [data.batch_to(data.collate_fn(synth.synth_molecules(B, seed=5000 + 31 * rank + i, profile="synth40")), dev)
                for i in range(2)]
        with torch.no_grad():
            for i in range(24):          # warm-up: the caching allocator grows into the new batch size over the first ~20 steps
                                         # (hipMalloc inside a step: 3.4-3.7 ms instead of 1.8 at 2048 molecules when only 8 were run)
                pool[i % 2].pop(PLAN_KEY, None)
                model(pool[i % 2])
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                pool[i % 2].pop(PLAN_KEY, None)
                model(pool[i % 2])
            torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            sec = float(el.item())
            print(json.dumps({"metric": "molecules/sec forward only (eval), synthetic 40-atom/12-fragment molecules",
                              "value": round(B * world * args.steps / sec, 1), "unit": "molecules/s", "n_gpus": world,
                              "per_gpu_batch": B, "steps": args.steps, "ms_per_step": round(sec / args.steps * 1e3, 3),
                              "atoms": int(pool[0]["x_atoms"].shape[0]),
                              "bond_graph_edges": int(pool[0]["edge_index_bonds_graph"].shape[1]), "dtype": "f32",
                              "data": "synthetic", "scaling": "weak",
                              "roofline": sweep_roofline(forward_bytes(pool[0]), sec / args.steps,
                                                         "the WHOLE forward step (plan build + 4 layers + head) against the forward half of the "
                                                         "SURVEY 8d compulsory-traffic model (forward_bytes); per GPU")}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def forward_sweep_store(rank, world, dev, args):
    """BASELINE configs[4] end to end: a FlatMolStore of ``--store`` synthetic 40-atom / 12-fragment molecules resident in
    HBM; every step = GPU collate of a fresh batch (StoreLoader, shuffled) + plan build + forward (eval mode).  Reports
    molecules/s for the whole pipeline and the share of the time spent in collate (timed alone over the same index lists).
    The store is ``distinct`` generated molecules replicated (the generator is Python: ~1.5 ms per molecule)."""
    from fragnet_amd import synth
    from fragnet_amd.dataset import FlatMolStore
    from fragnet_amd.model import FragNetFineTune
    from fragnet_amd.train import StoreLoader
    torch.manual_seed(0)
    model = FragNetFineTune(**MODEL_CFG).to(dev).eval()
    distinct = min(args.store, 8192)
    reps = max(1, args.store // distinct)
    t0 = time.perf_counter()
    base = FlatMolStore.from_records(synth.synth_molecules(distinct, seed=7000 + rank, profile="synth40")).to(dev)
    store = base.replicate(reps)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    store_gb = (sum(v.numel() * v.element_size() for v in store.t.values()) + sum(v.numel() * 8 for v in store.off.values())) / 1e9
    for B in (2048, 8192):
        steps = max(4, min(args.steps, len(store) // B))
        loader = StoreLoader(store, B, shuffle=True, drop_last=True, seed=11 + rank)
        it = iter(loader.sampler)
        WU = 24                # batches differ in size: the caching allocator needs a few steps to stop calling hipMalloc
        steps = max(4, min(steps, len(store) // B - WU))
        idx_lists = [next(it) for _ in range(steps + WU)]          # host indices: FlatMolStore.collate sizes the batch from its host-side lengths (no read-back)
        with torch.no_grad():
            probe = store.collate(idx_lists[0])
            alg_fwd, alg_coll = forward_bytes(probe), 2 * batch_bytes(probe)          # (batches differ by a fraction of a per cent)
            del probe
            for idx in idx_lists[:WU]:
                model(store.collate(idx))
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for idx in idx_lists[WU:]:
                model(store.collate(idx))
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
            # the collate alone over the same index lists: wall time of the loop (host-bound when the host's table work is longer than
            # the kernel) and, from events around every call, the time the GPU spends on it -- that is its share of the pipeline above,
            # where the host works on batch k + 1 while the GPU runs batch k
            evs = []
            store._collate_events = evs                     # (start, end) events right around the collate's two launches
            t0 = time.perf_counter()
            for idx in idx_lists[WU:]:
                store.collate(idx)
            torch.cuda.synchronize()
            coll = time.perf_counter() - t0
            store._collate_events = None
            # (no events: the collate did not take the fused one-launch path -- unsupported layout or FUSED_COLLATE off -- and nothing timed
            # its launches; the loop's wall time stands in and the collate gets no roofline line of its own)
            coll_timed = len(evs) > 0
            coll_dev = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 if coll_timed else coll
        if world > 1:
            torch.distributed.barrier()
        el = torch.tensor([total, coll, coll_dev, 0.0 if coll_timed else 1.0], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            total, coll, coll_dev, coll_timed_all = float(el[0]), float(el[1]), float(el[2]), float(el[3]) == 0.0
            print(json.dumps({"metric": "molecules/sec forward only (eval) from a resident store: collate + plan + forward",
                              "value": round(B * world * steps / total, 1), "unit": "molecules/s", "n_gpus": world,
                              "per_gpu_batch": B, "steps": steps, "ms_per_step": round(total / steps * 1e3, 3),
                              "collate_ms_per_step": round(coll_dev / steps * 1e3, 3), "collate_share": round(coll_dev / total, 3),
                              "collate_loop_ms_per_step": round(coll / steps * 1e3, 3),
                              "collate_note": "collate_ms_per_step / collate_share: GPU time of the collate's launches (events right around them); "
                                              "collate_loop_ms_per_step: wall time of a loop of collates alone (host + GPU, whichever is longer)",
                              "store_molecules": len(store), "store_distinct_molecules": distinct, "store_GB": round(store_gb, 2),
                              "store_build_s": round(build_s, 1), "dtype": "f32", "data": "synthetic (synth40 profile)", "scaling": "weak",
                              "roofline": sweep_roofline(alg_fwd + alg_coll, total / steps,
                                                         "collate (the batch's bytes once in, once out) + the whole forward step (forward_bytes); per GPU"),
                              "roofline_collate": (sweep_roofline(alg_coll, coll_dev / steps, "fn_collate_store alone (GPU time): the batch's bytes once in, once out")
                                                   if coll_timed_all else None)}),
                  flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def spawn_ranks(n, argv, child_cmd=None, poll_s=0.2):
    """Parent of an N-rank run started as plain `python bench.py --gpus N` (no RANK in the environment): starts N CHILD
    processes -- never exec, and before this process has imported torch.cuda or touched a GPU -- one per GPU, with the torchrun
    environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT = a free port), relays rank 0's
    stdout and prints rank 0's last JSON line as ITS OWN last stdout line.  Returns the exit code: 0 only if every child
    exited 0; the first failing child ends the others (by PID).  `child_cmd` (a list; tests) replaces `python bench.py ...`.
    Reference precedent for the data-parallel launch: train/finetune/finetune_gat2_pl.py:230-248 (Fabric starts the ranks)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), FRAGNET_BENCH_CHILD="1")
        # rank 0's stdout is the result channel; the other ranks' stdout joins stderr (their stderr passes through)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    import threading
    lines = []

    def pump():
        for ln in procs[0].stdout:
            lines.append(ln.rstrip("\n"))

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = 0
    live = set(range(n))
    # nothing waits for ever: an overall deadline (FRAGNET_BENCH_DEADLINE_S, default 1500 s -- the driver allows 1800), and once a
    # rank has failed or the deadline has passed the others get terminate(), then 10 s later kill() -- on the exact PIDs started here
    deadline = time.monotonic() + float(os.environ.get("FRAGNET_BENCH_DEADLINE_S", "1500"))
    kill_at = None
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench parent] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()                 # exact PIDs of our own children
                kill_at = time.monotonic() + 10.0
        if live and kill_at is None and time.monotonic() > deadline:
            rc = rc or 124
            print(f"[bench parent] deadline passed with ranks {sorted(live)} still running; stopping them", file=sys.stderr, flush=True)
            for q in live:
                procs[q].terminate()
            kill_at = time.monotonic() + 10.0
        if live and kill_at is not None and time.monotonic() > kill_at:
            for q in live:
                procs[q].kill()
            kill_at = time.monotonic() + 3600.0          # killed processes are reaped by the polls above
        if live:
            time.sleep(poll_s)
    th.join(timeout=10)
    result = None
    for ln in lines:
        if ln.startswith("{") and ln.endswith("}"):
            result = ln
        else:
            print(ln, flush=True)                        # anything else rank 0 printed, in order
    if result is not None:
        print(result, flush=True)                        # the JSON line is the LAST line of stdout
    elif rc == 0:
        print("[bench parent] rank 0 printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pool", type=int, default=4, help="distinct pre-collated batches cycled through")
    ap.add_argument("--scaling", choices=("weak", "strong", "both"), default="both",
                    help="N>1 only.  both (default): the headline is the GLOBAL batch of 512 sharded over the ranks (strong, what SURVEY.md "
                         "8d quotes the metric on) and the line carries a `weak` sub-object (512 molecules per rank); weak / strong: "
                         "only that one")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (one child, 1-rank RCCL group, the N>1 step sequence)")
    ap.add_argument("--child-cmd", default=None, help=argparse.SUPPRESS)      # tests: JSON list replacing the child command line
    ap.add_argument("--shard-of", type=int, default=0, metavar="R",
                    help="dev measurement on ONE GPU: time what rank 0 of an R-GPU strong-scaling job does per step (its 512/R-molecule "
                         "shard, no collective); the line is marked config.emulated_shard_of and is not a contract line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch-by-launch step instead of the whole-step hipGraph")
    ap.add_argument("--overlap", choices=("on", "off", "both"), default=None,
                    help="on: two-graph step with the head's gradient all-reduce (async) beside the encoder backward; N>1 default: "
                         "both (headline off, `overlap_on` sub-object)")
    ap.add_argument("--margin", type=float, default=0.05, help="capacity head-room of the static shapes over the batches they are sized from "
                                                               "(5 % as scripts/finetune_gat2.py)")
    ap.add_argument("--shape-sample", type=int, default=16,
                    help="N = 1: size the static shapes from this many batches SAMPLED from a resident store of other molecules (as the "
                         "drivers do) instead of from the timed pool itself, and run the epoch sample (--epoch-batches); 0: from the pool")
    ap.add_argument("--no-round3-shapes", action="store_true", help="skip the extra run under pool-sized shapes (same_code_round3_shapes)")
    ap.add_argument("--epoch-batches", type=int, default=200, help="shuffled batches of the epoch sample (with --shape-sample)")
    ap.add_argument("--store-molecules", type=int, default=8192, help="distinct synthetic molecules in the sample's store")
    ap.add_argument("--eager-head", action="store_true", help="(--eager) do not HIP-graph-capture the prediction head")
    ap.add_argument("--kernels-only", action="store_true", help="only time the bond-level scatter kernels (dev loop)")
    ap.add_argument("--kbatch", type=int, default=PER_GPU_BATCH, help="molecules per batch for --kernels-only")
    ap.add_argument("--model-version", default="gat2", choices=["gat2", "gat2_lite", "gat2_edge"],
                    help="gat2 (the headline metric), gat2_lite (SURVEY f3: levels L1-L3) or gat2_edge (f3: no fragment-bond "
                         "graph, per-level launches; cnx_attr widened to the 8 columns that model version expects)")
    ap.add_argument("--forward-sweep", action="store_true",
                    help="extra (BASELINE configs[4]): forward-only eval throughput on 40-atom/12-fragment molecules, one line per batch size")
    ap.add_argument("--sweep-batches", default="512,2048,8192", help="molecules per batch of --forward-sweep")
    ap.add_argument("--store", type=int, default=0,
                    help="with --forward-sweep: molecules in a FlatMolStore resident in HBM (1048576 = config[4] as written); every "
                         "step then collates a fresh shuffled batch on the GPU before plan + forward")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="A/B: library heuristics for the head GEMMs instead of TunableOp")
    ap.add_argument("--library-head", action="store_true", help="A/B: head layers as library GEMMs + element-wise kernels instead of fn_dense_*")
    ap.add_argument("--tune", action="append", default=None, help="A/B: KEY=VALUE for fn_set_tuning (include/fragnet_hip.h FN_TUNE_*)")
    ap.add_argument("--scatter-blocks", type=int, default=None, help="A/B: resident workgroups of the scatter kernels (FN_TUNE_FWD_BLOCKS)")
    return ap.parse_args(argv)


def in_step_launch_times(args, rank, world, dev, scaling, overlap, shape_batches, replays=40):
    """Durations of layer 0's bond-graph launches INSIDE the captured step, measured in this run: a second capture of the same
    step with four external HIP event-record nodes around them (fn_debug_set_profile_events; the library records them with
    hipEventRecordWithFlags(..., hipEventRecordExternal) on the capturing stream), `replays` replays, hipEventElapsedTime after
    each.  The timed headline loop runs WITHOUT these nodes.  Returns {"fwd_us", "bwd_us", "n", ...} or None (with a reason)."""
    import ctypes as C
    from fragnet_amd import _lib
    # plain timing events: torch only creates the handles and reads the elapsed time -- the LIBRARY records them, as external
    # event-record nodes (torch's own external=True events are refused on ROCm builds)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in evs:
        e.record()                                        # creates the handles
    torch.cuda.synchronize()
    arr = (C.c_void_p * 4)(*[int(e.cuda_event) for e in evs])
    run = None
    try:
        _lib.call("fn_debug_set_profile_events", arr)
        run = StepRun(args, rank, world, dev, scaling, overlap, shape_batches=shape_batches)
    except Exception as exc:      # noqa: BLE001
        return None, f"capture with event nodes failed: {type(exc).__name__}: {exc}"
    finally:
        _lib.call("fn_debug_set_profile_events", None)    # (the captured graph keeps its event nodes; nothing else records them)
    if run.gstep is None:
        why = run.capture_note
        run.release()
        return None, f"no captured step ({why})"
    fwd, bwd = [], []
    try:
        for i in range(5 + replays):
            run.step(i)
            torch.cuda.synchronize()
            if i >= 5:
                fwd.append(evs[0].elapsed_time(evs[1]) * 1e3)
                bwd.append(evs[2].elapsed_time(evs[3]) * 1e3)
    except Exception as exc:      # noqa: BLE001 -- e.g. the runtime refuses elapsed time between graph event nodes
        run.release()
        return None, f"{type(exc).__name__}: {exc}"
    run.release()
    med = statistics.median
    if not fwd or min(fwd) <= 0 or min(bwd) <= 0:
        return None, "event-node timestamps not usable (non-positive elapsed time)"
    return {"fwd_us": round(med(fwd), 2), "bwd_us": round(med(bwd), 2), "fwd_min_us": round(min(fwd), 2), "bwd_min_us": round(min(bwd), 2),
            "n": len(fwd),
            "method": "median over %d replays of a second capture of the same step with external HIP event-record nodes right before / behind "
                      "k_gat_fwd_pair (layer 0: bond + fragment-bond levels) and layer 0's k_gat_bwd_one3; hipEventElapsedTime after every replay "
                      "(the interval holds the launch and the event nodes' own hand-over, so it is an UPPER bound on the kernel's duration)" % len(fwd)}, None


class StepRun:
    """One configuration of the training step (scaling x overlap): its own model, optimiser, batch pool and captured graph."""

    def __init__(self, args, rank, world, dev, scaling, overlap, force_distributed=False, shape_batches=None):
        import fragnet_amd
        from fragnet_amd import parallel
        from fragnet_amd.model import FragNetFineTune
        from fragnet_amd.plan import PLAN_KEY
        self.args, self.rank, self.world, self.dev, self.scaling, self.overlap = args, rank, world, dev, scaling, overlap
        if args.shard_of > 1 and world == 1:
            pool = make_pool(args.pool, 0, dev, PER_GPU_BATCH, args.shard_of, "strong")
        else:
            pool = make_pool(args.pool, rank, dev, PER_GPU_BATCH, world, scaling)
        self.pool = pool
        self.local_batch = int(pool[0]["y"].shape[0])
        self.global_batch = PER_GPU_BATCH if scaling == "strong" else PER_GPU_BATCH * world
        if args.model_version == "gat2_edge":      # gat2_edge.py:46 wants 8 connection features, the featuriser writes 6
            for b in pool:
                b["cnx_attr"] = torch.nn.functional.pad(b["cnx_attr"], (0, 8 - b["cnx_attr"].shape[1]))
        torch.manual_seed(0)
        model = self.model = FragNetFineTune(**MODEL_CFG, variant=args.model_version).to(dev)
        model.train()
        model.pretrain.rng.rank = rank

        def fwd_bwd(batch):
            batch.pop(PLAN_KEY, None)                      # every step pays for its own graph plan
            loss = torch.nn.functional.mse_loss(model(batch).view(-1), batch["y"])
            loss.backward()
            return loss

        opt = self.opt = parallel.FlatAdam.for_live_parameters(model, lambda: fwd_bwd(pool[0]), lr=1e-4)
        self.gstep, self.capture_note, self.graphed_head = None, None, False
        eager = args.eager
        if not eager:
            from fragnet_amd import graphstep
            try:
                shapes = graphstep.StaticShapes.from_batches(shape_batches if shape_batches else pool, margin=args.margin, heads=MODEL_CFG["num_heads"],
                                                             spread_sigmas=4.0 if shape_batches else 0.0)
                self.gstep = graphstep.GraphedTrainStep(model, opt, shapes, pool[0], loss="regr", overlap=overlap,
                                                        force_distributed=force_distributed)
                torch.cuda.synchronize()
            except Exception as exc:      # never lose the measurement to a capture problem: run the same step eagerly
                self.gstep, self.capture_note = None, f"hipGraph capture failed ({type(exc).__name__}: {exc}); eager step"
                print(f"[bench rank {rank}] {self.capture_note}", file=sys.stderr, flush=True)
            if world > 1:                 # all ranks take the same path
                ok = torch.tensor([int(self.gstep is not None)], device=dev)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
                if not int(ok.item()) and self.gstep is not None:
                    self.gstep, self.capture_note = None, "another rank could not capture the step; eager step"
        self.eager = self.gstep is None
        if self.eager:
            if force_distributed:
                opt.force_collective = True
            self.graphed_head = fragnet_amd.graph_capture_head(model, self.local_batch) if not args.eager_head else False

            def step(i):
                opt.zero_grad()
                loss = fwd_bwd(pool[i % len(pool)])
                opt.step()                                   # one cat + (N>1: one all-reduce) + one fused Adam
                return loss
        else:
            gstep = self.gstep

            def step(i):
                return gstep(pool[i % len(pool)])            # stage (1 kernel) + graph replay + (all-reduce) + Adam
        self.step = step

    def timed(self, steps, warmup):
        """W untimed steps, barrier + synchronize, EXACTLY K steps, synchronize + barrier; seconds of this rank, the max and the
        min over ranks, the last loss."""
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        for i in range(warmup):
            self.step(i)
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = self.step(warmup + i)
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        mine = time.perf_counter() - t0
        hi = torch.tensor([mine], dtype=torch.float64, device=self.dev)
        lo = hi.clone()
        if dist_on:
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        return float(hi.item()), float(lo.item()), float(loss.item())

    def mode(self):
        if self.eager:
            return "eager launches, head " + ("hipGraph-captured" if self.graphed_head else "eager")
        if self.gstep.split:
            return ("two hipGraphs over static shapes (stage+plan+fwd+mse+head bwd | encoder bwd); the head's gradient "
                    "all-reduce runs beside the second")
        return "whole-step hipGraph over static shapes (stage+plan+fwd+mse+bwd+grad gather in the graph)"

    def release(self):
        self.gstep = self.step = self.model = self.opt = self.pool = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()


def allreduce_alone(opt, dev, iters=20):
    """The step's only collective by itself: blocking all-reduce (AVG) of the flat gradient buffer, back to back between
    synchronisations; microseconds per call, max over ranks."""
    for _ in range(3):
        opt.all_reduce()
    torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        opt.all_reduce()
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / iters * 1e6], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def epoch_sample(run, store, n_batches, dev):
    """A shuffled epoch through the captured step: every batch is collated on the GPU from the resident store (other molecules than
    the shapes were sized from are in it too: 16 of its batches were the sample), staged into the static buffers and replayed -- or
    run eagerly when it does not fit.  Reports the fallbacks and what the padding costs."""
    from fragnet_amd import graphstep
    from fragnet_amd.train import StoreLoader
    g = run.gstep
    f0, r0 = g.fallbacks, g.replays
    cap = g.shapes.cap
    pad_rows, tot_rows, n = 0, 0, 0
    # the loader the training drivers use: host indices from its sampler, one-launch collate in front of the step on the same stream
    warm = StoreLoader(store, PER_GPU_BATCH, shuffle=True, drop_last=True, seed=99)
    for _, b in zip(range(3), warm):
        g(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ep = 0
    while n < n_batches:
        for b in StoreLoader(store, PER_GPU_BATCH, shuffle=True, drop_last=True, seed=100 + ep):
            cnt = graphstep.batch_counts(b)
            pad_rows += sum(cap[sp] - cnt[sp] for sp in cnt)
            tot_rows += sum(cap[sp] for sp in cnt)
            n += 1
            g(b)
            if n == n_batches:
                break
        ep += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"batches": n, "molecules_in_store": len(store), "ms_per_step_incl_gpu_collate": round(dt / n * 1e3, 4),
            "molecules_per_s_incl_gpu_collate": round(PER_GPU_BATCH * n / dt, 1),
            "eager_fallbacks": g.fallbacks - f0 - 0, "graph_replays": g.replays - r0,
            "padded_row_fraction": round(pad_rows / max(1, tot_rows), 4),
            "what": f"{n} shuffled batches of {PER_GPU_BATCH} out of train.StoreLoader over a resident FlatMolStore (one-launch GPU collate, no device "
                    "read-back); static shapes sized from a "
                    "16-batch sample of the same store at the bench's margin; padded_row_fraction = (capacity - real items) / capacity "
                    "summed over the index spaces"}


def main():
    args = parse_args()
    # ---- N > 1 (or --spawn) without a launcher: this process becomes the PARENT of the ranks.  Nothing above or below this
    # point has initialised a GPU yet (torch is imported, torch.cuda is not initialised; device_count is not even asked).
    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ:
        child = json.loads(args.child_cmd) if args.child_cmd else None
        argv = [a for a in sys.argv[1:] if a != "--spawn"]
        sys.exit(spawn_ranks(args.gpus, argv, child))

    import fragnet_amd
    from fragnet_amd import parallel
    fragnet_amd.prefer_rocblas_for_dense_heads()
    if args.library_head:
        from fragnet_amd import ops
        ops.DENSE_HEAD = False
    if args.library_head and not args.no_gemm_tuning:
        fragnet_amd.tune_library_gemms()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    for kv in args.tune or []:         # A/B of any fn_set_tuning key: --tune 3=1024
        k, v = kv.split("=")
        from fragnet_amd import _lib
        _lib.call("fn_set_tuning", int(k), int(v))
    if args.scatter_blocks is not None:
        from fragnet_amd import _lib
        _lib.call("fn_set_tuning", 0, args.scatter_blocks)
    spawned_single = os.environ.get("FRAGNET_BENCH_CHILD") == "1" and int(os.environ.get("WORLD_SIZE", "1")) == 1
    # FRAGNET_BENCH_BACKEND=gloo (tests): the N > 1 code path with several ranks on ONE GPU -- RCCL refuses two ranks per device,
    # gloo moves the CUDA buffers through the host; ranks then share device LOCAL_RANK % device_count
    rank, local_rank, world = parallel.init_distributed(backend=os.environ.get("FRAGNET_BENCH_BACKEND") or None, force=spawned_single)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    dist_on = world > 1 or spawned_single            # the N>1 step sequence: graph replay -> RCCL all-reduce -> Adam outside the graph

    if args.forward_sweep:
        (forward_sweep_store if args.store > 0 else forward_sweep)(rank, world, dev, args)
        return
    if args.kernels_only:
        from fragnet_amd.model import FragNetFineTune
        torch.manual_seed(0)
        model = FragNetFineTune(**MODEL_CFG, variant=args.model_version).to(dev)
        print(json.dumps(kernel_roofline(make_pool(1, rank, dev, args.kbatch)[0], model, alternatives=True)))
        return

    # headline configuration: one GPU = the 512-molecule batch; N > 1 = the global batch of 512 sharded over the ranks
    # (SURVEY.md 8d quotes the metric on a global batch of 512) unless --scaling weak
    head_scaling = "weak" if (world == 1 or args.scaling == "weak") else "strong"
    overlap_sel = args.overlap if args.overlap is not None else ("both" if dist_on else "off")
    head_overlap = overlap_sel == "on"
    # N = 1: the static shapes come from a 16-batch SAMPLE of a resident store of other molecules at 5 % (scripts/finetune_gat2.py sizes
    # them that way), not from the timed pool: a pool batch that does not fit shows up as an eager fallback in the line
    store, shape_batches = None, None
    if world == 1 and args.shard_of <= 1 and args.shape_sample > 0 and not args.eager and args.model_version == "gat2":
        from fragnet_amd import synth
        from fragnet_amd.dataset import BatchSampler, FlatMolStore
        store = FlatMolStore.from_records(synth.synth_molecules(args.store_molecules, seed=9000, profile="esol")).to(dev)
        sampler = iter(BatchSampler(len(store), PER_GPU_BATCH, True, True, seed=5))
        shape_batches = [store.collate(next(sampler)) for _ in range(args.shape_sample)]
    run = StepRun(args, rank, world, dev, head_scaling, head_overlap, force_distributed=spawned_single, shape_batches=shape_batches)
    elapsed, fastest, final_loss = run.timed(args.steps, args.warmup)
    # the contract's K steps are the FIRST timed loop; four more loops of K steps give the spread
    rep_ms = [elapsed / args.steps * 1e3]
    for _ in range(4):
        hi, _, _ = run.timed(args.steps, 0)
        rep_ms.append(hi / args.steps * 1e3)
    extras = {}
    extras["ms_per_step_repeats"] = {"n": len(rep_ms), "steps_each": args.steps, "min": round(min(rep_ms), 4),
                                     "median": round(statistics.median(rep_ms), 4), "max": round(max(rep_ms), 4),
                                     "note": "ms_per_step / value are the first loop's (the contract's K steps)"}
    if store is not None and run.gstep is not None and args.epoch_batches > 0:
        extras["epoch_sample"] = epoch_sample(run, store, args.epoch_batches, dev)
    if store is not None and run.gstep is not None and not args.no_round3_shapes and args.epoch_batches > 0:      # (--epoch-batches 0: the traced runs, one configuration only)
        # the same code under the shapes rounds 1-3 quoted their headline on (capacities from the timed pool itself + 2 %): what the
        # switch to sampled shapes costs, and the like-for-like figure against BENCH_r03
        import copy
        a3 = copy.copy(args)
        a3.margin = 0.02
        r3 = StepRun(a3, rank, world, dev, head_scaling, head_overlap, force_distributed=spawned_single)
        hi3, _, _ = r3.timed(args.steps, args.warmup)
        extras["same_code_round3_shapes"] = {"ms_per_step": round(hi3 / args.steps * 1e3, 4), "value": round(r3.global_batch * args.steps / hi3, 1),
                                             "unit": "molecules/s", "static_capacity": r3.gstep.shapes.cap, "eager_fallbacks": r3.gstep.fallbacks,
                                             "what": "static shapes sized from the four timed pool batches at 2 % head-room, as BENCH_r01-r03 were measured "
                                                     "(646 k molecules/s, 0.792 ms in round 3); NOT the headline: a dataset's batches do not fit such shapes"}
        r3.release()
    sub_steps, sub_warm = max(5, min(args.steps, 20)), max(2, min(args.warmup, 5))
    if dist_on:
        extras["allreduce_alone"] = {"bytes": run.opt.nbytes, "us_per_call": round(allreduce_alone(run.opt, dev), 1),
                                     "what": "blocking all-reduce (AVG) of the flat gradient buffer, 20 back-to-back calls, max over ranks"}
        try:
            import torch.cuda.nccl as _nccl
            ver = ".".join(str(v) for v in _nccl.version())
        except Exception:      # noqa: BLE001 -- version string only
            ver = "?"
        extras["nccl_ranks"] = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                                "library_version": ver, "devices": torch.cuda.device_count()}
    head = {"local_batch": run.local_batch, "global_batch": run.global_batch, "mode": run.mode(), "capture_note": run.capture_note,
            "cap": None if run.gstep is None else run.gstep.shapes.cap, "replays": None if run.gstep is None else run.gstep.replays,
            "fallbacks": None if run.gstep is None else run.gstep.fallbacks, "nbytes": run.opt.nbytes, "eager": run.eager,
            "pool0": run.pool[0]}

    def sub_run(scaling, overlap):
        r = StepRun(args, rank, world, dev, scaling, overlap, force_distributed=spawned_single)
        hi, lo, _ = r.timed(sub_steps, sub_warm)
        out = {"value": round(r.global_batch * sub_steps / hi, 1), "unit": "molecules/s", "scaling": scaling, "overlap": "on" if overlap else "off",
               "per_gpu_batch": r.local_batch, "global_batch": r.global_batch, "steps": sub_steps, "warmup": sub_warm,
               "ms_per_step": round(hi / sub_steps * 1e3, 3), "ms_per_step_fastest_rank": round(lo / sub_steps * 1e3, 3), "mode": r.mode()}
        r.release()
        return out

    if dist_on and not args.eager and args.shard_of <= 1:
        if overlap_sel == "both":
            extras["overlap_on"] = sub_run(head_scaling, True)
        if world > 1 and args.scaling == "both":
            extras["weak"] = sub_run("weak", False)
            if overlap_sel == "both":
                extras["weak"]["overlap_on"] = sub_run("weak", True)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = head["global_batch"] * args.steps / elapsed
        if args.shard_of > 1 and world == 1:      # dev line: one rank's shard of a strong-scaling job, measured alone
            value = head["local_batch"] * args.steps / elapsed
        sb = step_bytes(head["pool0"], args.model_version)
        line = {
            "metric": "molecules/sec fwd+bwd (full training step), ESOL-shape batch=512 " + ("per GPU" if head_scaling == "weak" else "global"),
            "value": round(value, 1), "unit": "molecules/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": head_scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"model_version": args.model_version,
                       "workload": "ESOL finetune batch=512 fp32 (BASELINE configs[1]): FragNetFineTune 4 layers x 4 heads, "
                                   "emb 128, FTHead3 128/1024/1024/512, drop 0.1; synthetic ESOL-shape molecules (synth.py)",
                       "per_gpu_batch": head["local_batch"], "global_batch": head["global_batch"], "parallelism": f"dp{world}",
                       **({"emulated_shard_of": args.shard_of, "note": "NOT a contract line: rank 0's shard of a strong-scaling job on "
                           "one GPU, no collective; value = this shard's molecules/s"} if args.shard_of > 1 and world == 1 else {}),
                       **({"spawned": "one child process, 1-rank RCCL group, N>1 step sequence (all-reduce + Adam outside the graph)"}
                          if spawned_single else {}),
                       "mode": head["mode"], "overlap": "on" if head_overlap else "off",
                       "capture_note": head["capture_note"],
                       "static_capacity": head["cap"],
                       "graph_replays": head["replays"],
                       "eager_fallbacks": head["fallbacks"],
                       "step": "plan+zero_grad+fwd+mse+bwd" + ("+allreduce(flat %.1f MB)" % (head["nbytes"] / 1e6) if dist_on else "") + "+adam",
                       "atoms": int(head["pool0"]["x_atoms"].shape[0]), "bond_graph_edges": int(head["pool0"]["edge_index_bonds_graph"].shape[1])},
            "final_loss": round(final_loss, 6),
            "ms_per_step_fastest_rank": round(fastest / args.steps * 1e3, 3),
            "step_algorithmic_GB": round(sb * (world if head_scaling == "strong" else 1) / 1e9, 4),
        }
        # whole job: bytes all ranks moved per step / the step time (strong: rank 0's shard x world approximates the global batch)
        job_bytes = sb * world
        line["step_achieved_GBps"] = round(job_bytes / (ms * 1e-3) / 1e9, 1)
        line["step_frac_of_hbm_peak"] = round(job_bytes / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), 4)
        line.update(extras)
        if not args.no_roofline:
            # the roofline object is quoted on the metric's configuration: a full 512-molecule batch (a rank of a strong-scaling
            # job holds only its shard)
            roof_batch = head["pool0"] if head["local_batch"] == PER_GPU_BATCH else make_pool(1, 0, dev, PER_GPU_BATCH)[0]
            one_pass = not any(kv.replace(" ", "") == "22=0" for kv in (args.tune or []))       # FN_TUNE_BWD_ONE (the default)
            kr = kernel_roofline(roof_batch, run.model, alternatives=not one_pass)
            fwd_k, bwd_k = ("k_gat_fwd(+out2)", "k_gat_bwd_one") if one_pass else ("k_gat_fwd", "k_gat_bwd(dst+src)")
            nb, mb = kr[fwd_k]["n"], kr[fwd_k]["m"]
            f1, b1 = level_bytes(nb, mb)
            sa_us = kr[fwd_k]["us_per_launch"] + kr[bwd_k]["us_per_launch"]
            standalone = {fwd_k: {"us": kr[fwd_k]["us_per_launch"], "GBps": kr[fwd_k]["GBps"], "frac": round(kr[fwd_k]["GBps"] / HBM_PEAK_GBPS, 4)},
                          bwd_k: {"us": kr[bwd_k]["us_per_launch"], "GBps": kr[bwd_k]["GBps"], "frac": round(kr[bwd_k]["GBps"] / HBM_PEAK_GBPS, 4)},
                          "fwd+bwd": {"us": round(sa_us, 2), "GBps": round((f1 + b1) / sa_us / 1e3, 1),
                                      "frac": round((f1 + b1) / sa_us / 1e3 / HBM_PEAK_GBPS, 4)},
                          "method": "50 back-to-back launches of each kernel on one resident batch, HIP events on the launch stream (hot cache)"}
            # ---- inside the replayed step: durations from the committed rocprofv3 trace of this command (profiles/in_graph_kernels.json,
            # tools/rocpd_summary.py --json), used only when that trace was taken with THIS library's sources.  The launches of the
            # bond-graph level that carry nothing else are layer 0's: k_gat_fwd_pair (bond + fragment-bond levels, forward) and the
            # smallest k_gat_bwd_one3 grid (the same two levels, backward; two-pass: k_gat_bwd_dst_pair + k_gat_bwd_src_pair)
            inside, ig_source = {}, None
            ig = os.path.join(ROOT, "profiles", "in_graph_kernels.json")
            if os.path.exists(ig):
                from fragnet_amd import build
                gj = json.load(open(ig))
                fresh = gj.get("source_digest") == build.source_digest()
                ks, bg = (gj.get("kernels", {}), gj.get("by_grid", {})) if fresh else ({}, {})
                ig_source = "profiles/in_graph_kernels.json <- " + gj.get("source", "?") + (
                    "" if fresh else " -- STALE: traced with other kernel sources than this library, figures dropped")
                fb_n, fb_m = int(head["pool0"]["node_features_fbonds"].shape[0]), int(head["pool0"]["edge_index_fbonds"].shape[1])
                f2, b2 = level_bytes(fb_n, fb_m)
                # the replayed step's launch, not the mean over all dispatches of that name: the warm-up / capture passes run once under
                # other shapes and the very first launch is a cold one (134 us)
                fcand = {int(k.split("@")[1]): v for k, v in bg.items() if k.startswith("k_gat_fwd_pair@")}
                t_f = max(fcand.values(), key=lambda v: v["calls"])["avg_us"] if fcand else ks.get("k_gat_fwd_pair", {}).get("avg_us")
                t_b = None
                if one_pass:
                    cand = {int(k.split("@")[1]): v for k, v in bg.items() if k.startswith("k_gat_bwd_one3@")}
                    most = max((v["calls"] for v in cand.values()), default=0)
                    # layer 0's launch is the smallest grid of the REPLAYED step (the warm-up / capture passes run once under other shapes)
                    grids = sorted((g, v["avg_us"]) for g, v in cand.items() if v["calls"] * 10 >= most)
                    if grids:
                        t_b, bwd_name = grids[0][1], f"k_gat_bwd_one3@{grids[0][0]} workgroups (layer 0: bond + fragment-bond levels)"
                if t_f:
                    inside["k_gat_fwd_pair"] = {"us": t_f, "bytes": f1 + f2, "frac": round((f1 + f2) / t_f / 1e3 / HBM_PEAK_GBPS, 4)}
                if t_b:
                    inside[bwd_name] = {"us": round(t_b, 2), "bytes": b1 + b2, "frac": round((b1 + b2) / t_b / 1e3 / HBM_PEAK_GBPS, 4)}
                if t_f and t_b:
                    inside["fwd+bwd"] = {"us": round(t_f + t_b, 2), "bytes": f1 + f2 + b1 + b2,
                                         "GBps": round((f1 + f2 + b1 + b2) / (t_f + t_b) / 1e3, 1),
                                         "frac": round((f1 + f2 + b1 + b2) / (t_f + t_b) / 1e3 / HBM_PEAK_GBPS, 4)}
            traffic, traffic_source, traffic_by_kernel = None, None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
            if os.path.exists(pmc):
                from fragnet_amd import build
                pj = json.load(open(pmc))
                parts = (bwd_k,) if one_pass else ("k_gat_bwd_dst", "k_gat_bwd_src")
                if pj.get("source_digest") not in (None, build.source_digest()):
                    traffic_source = "profiles/pmc_per_launch.json -- STALE: collected with other kernel sources than this library, figures dropped"
                elif all(k in pj for k in parts):
                    traffic = sum(pj[k]["hbm_bytes_per_launch"] for k in parts)
                    traffic_by_kernel = {k: v["hbm_bytes_per_launch"] for k, v in pj.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
                    traffic_source = ("profiles/pmc_per_launch.json (" + pj.get("_collected", "?") + "): separate rocprofv3 --pmc FETCH_SIZE / "
                                      "WRITE_SIZE passes over `bench.py --kernels-only`, NOT this run; `traffic` = the backward of the level (" +
                                      " + ".join(parts) + "), to be held against algorithmic_bytes_per_launch of the backward")
            # ---- the same two launches timed IN THIS RUN (external event-record nodes in a second capture of the step)
            measured, measured_why = None, "not attempted (needs the single-rank captured one-pass step)"
            if one_pass and world == 1 and run.gstep is not None and args.shard_of <= 1 and head["local_batch"] == PER_GPU_BATCH:
                measured, measured_why = in_step_launch_times(args, rank, world, dev, head_scaling, head_overlap, shape_batches)
            live = {}
            if measured:
                fb_n, fb_m = int(head["pool0"]["node_features_fbonds"].shape[0]), int(head["pool0"]["edge_index_fbonds"].shape[1])
                f2, b2 = level_bytes(fb_n, fb_m)
                t_f, t_b = measured["fwd_us"], measured["bwd_us"]
                live["k_gat_fwd_pair"] = {"us": t_f, "bytes": f1 + f2, "frac": round((f1 + f2) / t_f / 1e3 / HBM_PEAK_GBPS, 4)}
                live["k_gat_bwd_one3 (layer 0: bond + fragment-bond levels)"] = {"us": t_b, "bytes": b1 + b2, "frac": round((b1 + b2) / t_b / 1e3 / HBM_PEAK_GBPS, 4)}
                live["fwd+bwd"] = {"us": round(t_f + t_b, 2), "bytes": f1 + f2 + b1 + b2, "GBps": round((f1 + f2 + b1 + b2) / (t_f + t_b) / 1e3, 1),
                                   "frac": round((f1 + f2 + b1 + b2) / (t_f + t_b) / 1e3 / HBM_PEAK_GBPS, 4)}
            # what the headline launches MOVE inside the step (whole-step PMC table, profiles/rNN_pmc_step.json) next to what a plain copy
            # of that size gets on this part (profiles/r06_cold_stream_probe.txt): context for `frac`, not a metric
            moved = None
            # (the newest committed table; dropped -- like the trace's durations -- when it was collected with other kernel sources)
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_step.json")))
            ps = cands[-1] if cands else ""
            psj = json.load(open(ps)) if ps else {}
            if ps and psj.get("source_digest") is not None:
                from fragnet_amd import build
                if psj["source_digest"] != build.source_digest():
                    psj = {}
            base = (live or inside).get("fwd+bwd")
            if one_pass and base and psj:
                seq = psj.get("sequence", [])
                fw = [e for e in seq if e["kernel"] == "k_gat_fwd_pair"]
                bw_ = sorted((e for e in seq if e["kernel"] == "k_gat_bwd_one3"), key=lambda e: e["workgroups"])
                if fw and bw_:
                    mb = fw[0]["hbm_MB"] + bw_[0]["hbm_MB"]
                    moved = {"hbm_bytes_per_launch_pair": int(mb * 1e6), "GBps": round(mb * 1e3 / base["us"], 1),
                             "over_algorithmic": round(mb * 1e6 / base["bytes"], 2),
                             "source": "profiles/" + os.path.basename(ps) + " (FETCH_SIZE x 2 + WRITE_SIZE of k_gat_fwd_pair and layer 0's backward launch in "
                                       "one replayed step, another run; same kernel sources by digest where the table carries one) over the in-step durations above",
                             "copy_reference": "profiles/r06_cold_stream_probe.txt (device timestamps): a plain copy of 32-64 MB per direction moves 7.0-7.5 TB/s "
                                               "hot AND cold behind a read-only evicting pass; 3.3-3.7 TB/s only behind a FILL that left the memory-side "
                                               "cache dirty (the 3.7-4.1 TB/s 'cold' figure of r04-r06_hbm_cold_stream.md is that case)"}
            # headline = what the step obeys: the bond-graph level's forward + backward bytes over its in-step durations -- measured in
            # this run when the event nodes gave usable times, else read from the committed trace when it describes this library, else the
            # stand-alone launches (and the line says which)
            use = live.get("fwd+bwd") or inside.get("fwd+bwd")
            headline = use if use else standalone["fwd+bwd"]
            line["roofline"] = {"bound": "hbm",
                                "kernel": (fwd_k + " + " + bwd_k + "<4> @ bond-graph level, ") + ("inside the replayed step (layer 0's launches: + fragment-bond level)"
                                                                                                  if use else "stand-alone launches"),
                                "achieved": headline["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": headline["frac"],
                                "basis": ("measured in this run: in-step durations of layer 0's two launches from external HIP event-record nodes "
                                          "(`in_graph_measured`); the committed rocprofv3 trace of the same command is the cross-check (`in_graph`)") if live
                                         else (("in_graph: durations read from the committed rocprofv3 trace " + str(ig_source) + ", not measured in this "
                                                "run (event nodes: " + str(measured_why) + "); `standalone` below IS measured in this run") if use
                                               else "standalone: measured in this run (no committed trace of this library's sources; event nodes: "
                                                    + str(measured_why) + ")"),
                                # the same quantity on the bases earlier rounds quoted (round 3's headline was the stand-alone backward: 0.235)
                                "frac_backward_standalone": standalone[bwd_k]["frac"],
                                "frac_forward_standalone": standalone[fwd_k]["frac"],
                                "frac_backward_in_graph": next((v["frac"] for k, v in (live or inside).items() if k.startswith("k_gat_bwd")), None),
                                "frac_forward_in_graph": (live or inside).get("k_gat_fwd_pair", {}).get("frac"),
                                "in_graph_measured": ({**live, **{k: measured[k] for k in ("fwd_min_us", "bwd_min_us", "n", "method")}} if live
                                                      else {"unavailable": measured_why}),
                                "traffic": traffic, "traffic_source": traffic_source, "traffic_by_kernel": traffic_by_kernel,
                                "moved_in_graph": moved,
                                "us_per_launch": headline["us"],
                                "algorithmic_bytes_per_launch": headline.get("bytes", f1 + b1),
                                "algorithmic_bytes_backward_bond_level": b1,
                                "bytes_model": "SURVEY.md 8d: B_agg = 4[(n+1)+m+mH+2nH+nD+nD+mH] forward; B_agg' = 4[2nD+2mH+2m+nD+mH+2nH] for "
                                               "the whole backward of the level; the one-pass backward is priced against the same B_agg' (its "
                                               "second forward output and the dots it reads are extra traffic, not extra algorithmic bytes)",
                                "standalone": standalone,
                                "in_graph": {"source": ig_source, **inside} if ig_source else None,
                                "all": kr}
            # extra evidence (not part of the contract): the same kernels on a 2048-molecule batch, where a launch
            # is long enough for the per-launch fixed cost (~4 us) not to dominate
            big = make_pool(1, rank, dev, 2048)[0]
            kb = kernel_roofline(big, run.model, iters=20, alternatives=not one_pass)
            line["roofline"]["at_batch_2048"] = {k: {"us_per_launch": v["us_per_launch"], "GBps": v["GBps"],
                                                     "frac": round(v["GBps"] / HBM_PEAK_GBPS, 4)} for k, v in kb.items() if isinstance(v, dict)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
