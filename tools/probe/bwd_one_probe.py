"""Dev probe (round 4): the one-pass attention backward (csrc/gat_bwd_one.inc) beside the two passes, stand-alone launches.

Bond level (edge class 1) and atom level (class 0, self loops) of ESOL-shape batches; HIP events on the launch stream, hot cache.
    python tools/probe/bwd_one_probe.py [--batch 512] [--iters 50] [--blocks N]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


_BIG = None


def timeit_cold(fn, iters=12):
    """one launch at a time, a 1-GB streaming pass in front of each (evicts the L2s and the 256-MB memory-side cache)"""
    global _BIG
    if _BIG is None:
        _BIG = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
    tot = 0.0
    for _ in range(iters):
        _BIG.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b) * 1000.0
    return tot / iters


def timeit(fn, iters):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--profile", default="esol")
    ap.add_argument("--tune", action="append", default=[], help="KEY=VALUE for fn_set_tuning")
    ap.add_argument("--stamps", action="store_true", help="phase stamps of the one-pass kernel (s_memtime, median / p90 over waves)")
    ap.add_argument("--cold", action="store_true", help="also time every kernel launch by launch behind a 1-GB streaming pass (cold caches)")
    ap.add_argument("--pem", type=int, default=1, help="probabilities edge-major for the one-pass kernel")
    ap.add_argument("--xsrc", type=int, default=1, help="raw edge attribute in source order for the one-pass kernel")
    args = ap.parse_args()
    from fragnet_amd import _lib, data, synth
    from fragnet_amd.plan import GraphPlan, _stream_ptr
    dev = torch.device("cuda:0")
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(args.batch, seed=1000, profile=args.profile)), dev)
    plan = GraphPlan.from_batch(batch)
    st = _stream_ptr(dev)
    H, D, d = 4, 128, 32
    PEM = args.pem
    if args.blocks:
        _lib.call("fn_set_tuning", 23, args.blocks)
    for kv in args.tune:
        _lib.call("fn_set_tuning", int(kv.split("=")[0]), int(kv.split("=")[1]))
    f32 = dict(dtype=torch.float32, device=dev)
    res = {}
    for name in ("bond", "atom"):
        lv = plan.levels[name]
        n, m = lv.n, lv.m
        g = torch.Generator().manual_seed(0)
        h = torch.randn(n, D, generator=g).to(dev)
        gout = torch.randn(n, D, generator=g).to(dev)
        if name == "bond":
            att = (torch.randn(H, 3 * d, generator=g) * 0.3).to(dev)
            embW, embb = (torch.randn(d, 1, generator=g) * 0.5).to(dev), (torch.randn(d, generator=g) * 0.5).to(dev)
            x = plan.sorted_attr("bond", batch["edge_attr_bonds"])
            et = _lib.EdgeTerm(2, 1, d, d, None, x.data_ptr(), embW.data_ptr(), embb.data_ptr())
            att_w, src_off = 3 * d, 2 * d
        else:
            att = (torch.randn(H, 2 * d + D, generator=g) * 0.3).to(dev)
            s_sorted = (torch.randn(H, m, generator=g) * 0.5).to(dev)
            et = _lib.EdgeTerm(0, 0, 0, 0, s_sorted.data_ptr(), None, None, None)
            att_w, src_off = 2 * d + D, d + D
        et_b = _lib.EdgeTerm(et.mode, et.K, et.d_e, et.mid_off, None, et.x_sorted, et.embW, et.embb)
        et_1 = _lib.EdgeTerm(et.mode, et.K, et.d_e, et.mid_off, None, et.x_sorted, et.embW, et.embb)
        if name == "bond" and args.xsrc:
            x_src = torch.empty(1, m, dtype=torch.float32, device=dev)
            _lib.call("fn_sort_edge_attr_src_f32", batch["edge_attr_bonds"].contiguous().data_ptr(), 1, C.byref(lv.c), x_src.data_ptr(), st)
            et_1.x_src = x_src.data_ptr()
        p_one = torch.empty(H, m, dtype=torch.float32, device=dev)
        s_dst, s_src = torch.empty(n, H, **f32), torch.empty(n, H, **f32)
        out, out2, sigma = torch.empty(n, D, **f32), torch.empty(n, D, **f32), torch.empty(n, H, **f32)
        p_sorted = torch.empty(H, m, **f32)
        pz = torch.empty(H, m, 2, **f32)
        dz, dz1 = torch.zeros(m, H, **f32), torch.zeros(m, H, **f32)
        g_s_dst, g_s_dst1, cdot = torch.empty(n, H, **f32), torch.empty(n, H, **f32), torch.empty(n, H, **f32)
        g_h, g_h1 = torch.empty(n, D, **f32), torch.empty(n, D, **f32)
        part_e, part_a = torch.zeros(4096, H * 2, **f32), torch.zeros(256, 4096, **f32)
        part_e1, part_a1 = torch.zeros(4096, H * 2, **f32), torch.zeros(256, 4096, **f32)
        n_e, n_a, n_e1, n_a1 = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.call("fn_node_scalars_f32", h.data_ptr(), att.data_ptr(), att_w, 0, src_off, s_dst.data_ptr(), s_src.data_ptr(), n, H, st)
        orig = name == "atom"

        def fwd():
            _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w, C.byref(et), C.byref(lv.c),
                      0.2, out.data_ptr(), p_sorted.data_ptr(), None, None, None, 0, None, H, st)

        def fwd2():
            _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w, C.byref(et), C.byref(lv.c),
                      0.2, out.data_ptr(), p_one.data_ptr(), None, out2.data_ptr(), sigma.data_ptr(), PEM, None, H, st)

        y_act = torch.empty_like(out)
        act_relu = _lib.ActEpilogue(y_act.data_ptr(), 0.0, 1, 7, 0, None)
        act_drop = _lib.ActEpilogue(y_act.data_ptr(), 0.1, 1, 7, 0, None)

        def fwd2_relu():        # + the relu epilogue (second store of the row)
            _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w, C.byref(et), C.byref(lv.c),
                      0.2, out.data_ptr(), p_one.data_ptr(), None, out2.data_ptr(), sigma.data_ptr(), PEM, C.byref(act_relu), H, st)

        def fwd2_drop():        # + Philox dropout in it (what the training step runs)
            _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w, C.byref(et), C.byref(lv.c),
                      0.2, out.data_ptr(), p_one.data_ptr(), None, out2.data_ptr(), sigma.data_ptr(), PEM, C.byref(act_drop), H, st)

        def bwd_dst():
            _lib.call("fn_gat_bwd_dst_f32", gout.data_ptr(), h.data_ptr(), p_sorted.data_ptr(), C.byref(et_b), C.byref(lv.c), 0.2,
                      None, dz.data_ptr() if orig else None, pz.data_ptr(), g_s_dst.data_ptr(), part_e.data_ptr(), C.byref(n_e), H, st)

        def bwd_src():
            _lib.call("fn_gat_bwd_src_f32", gout.data_ptr(), h.data_ptr(), pz.data_ptr(), g_s_dst.data_ptr(), att.data_ptr(), att_w, 0, src_off,
                      C.byref(lv.c), g_h.data_ptr(), part_a.data_ptr(), C.byref(n_a), H, st)

        def cu():
            _lib.call("fn_gat_cu_f32", gout.data_ptr(), out.data_ptr(), out2.data_ptr(), sigma.data_ptr(), 1.0, cdot.data_ptr(),
                      g_s_dst1.data_ptr(), n, H, st)

        def one():
            _lib.call("fn_gat_bwd_one_f32", gout.data_ptr(), h.data_ptr(), p_one.data_ptr(), cdot.data_ptr(), g_s_dst1.data_ptr(), C.byref(et_1),
                      att.data_ptr(), att_w, 0, src_off, C.byref(lv.c), 0.2, g_h1.data_ptr(), None, dz1.data_ptr() if orig else None,
                      part_a1.data_ptr(), C.byref(n_a1), part_e1.data_ptr(), C.byref(n_e1), PEM, None, H, st)

        fwd(); fwd2(); bwd_dst(); bwd_src(); cu(); one()
        torch.cuda.synchronize()
        err = {"g_h": float((g_h - g_h1).abs().max()), "g_h_scale": float(g_h.abs().max()),
               "g_s_dst": float((g_s_dst - g_s_dst1).abs().max()),
               "part_a": float((part_a[:, :n_a.value].sum(1) - part_a1[:, :n_a1.value].sum(1)).abs().max()),
               "part_a_scale": float(part_a[:, :n_a.value].sum(1).abs().max())}
        if orig:
            err["dz"] = float((dz - dz1).abs().max())
        else:
            err["part_e"] = float((part_e[:n_e.value].sum(0) - part_e1[:n_e1.value].sum(0)).abs().max())
        r = {"n": n, "m": m, "err_vs_two_pass": err, "blocks_one": n_a1.value}
        for nm, fn in (("fwd", fwd), ("fwd+out2", fwd2), ("fwd+out2+relu", fwd2_relu), ("fwd+out2+relu(dropout)", fwd2_drop), ("bwd_dst", bwd_dst), ("bwd_src", bwd_src), ("cu", cu), ("bwd_one", one)):
            r[nm + "_us"] = round(timeit(fn, args.iters), 2)
            if args.cold:
                r[nm + "_cold_us"] = round(timeit_cold(fn), 2)
        bwd_b = 4 * (2 * n * D + 2 * m * H + 2 * m + n * D + m * H + 2 * n * H)
        r["B_agg_bwd"] = bwd_b
        r["two_pass_us"] = round(r["bwd_dst_us"] + r["bwd_src_us"], 2)
        r["frac_two_pass"] = round(bwd_b / r["two_pass_us"] / 1e6 / 8.0, 4)
        r["frac_one"] = round(bwd_b / r["bwd_one_us"] / 1e6 / 8.0, 4)
        r["frac_one+cu"] = round(bwd_b / (r["bwd_one_us"] + r["cu_us"]) / 1e6 / 8.0, 4)
        if args.stamps:
            nw = n_a1.value * 4
            buf = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            _lib.call("fn_debug_set_stamps", buf.data_ptr(), buf.numel())
            one()
            torch.cuda.synchronize()
            _lib.call("fn_debug_set_stamps", None, 0)
            t = buf.view(nw, 16).double().cpu()
            # s_memtime counts shader cycles and is NOT comparable across XCDs: phases are differences inside one wave
            names = ["entry -> loop start (two dependent round trips)"] + \
                    [f"row{i}: {w}" for i in range(4) for w in ("previous end -> loads issued", "issued -> data arrived", "arrived -> row done")] + \
                    ["last stamped row -> loop end (rows beyond the fourth)", "loop end -> exit (LDS reduction, partial rows)"]
            ph = {}
            for k, nm in enumerate(names):        # a phase counts for the waves that stamped both of its ends (short levels stamp fewer rows)
                a, b = t[:, k], t[:, k + 1]
                if k == 13:                       # loop end follows the LAST stamped row
                    a = torch.stack([t[:, q] for q in (4, 7, 10, 13)], 1).max(dim=1).values
                both = (a > 0) & (b > 0)
                if int(both.sum()):
                    v = (b - a)[both]
                    ph[nm] = [int(v.median()), int(v.quantile(0.9)), int(both.sum())]
            r["phase_cycles(median,p90,waves)"] = ph
            life = (t[:, 15] - t[:, 0])[(t[:, 15] > 0) & (t[:, 0] > 0)]
            r["wave_lifetime_cycles(median,p90,max)"] = [int(life.median()), int(life.quantile(0.9)), int(life.max())]
        res[name] = r
    print(json.dumps({"batch": args.batch, "profile": args.profile, "levels": res}, indent=1))


if __name__ == "__main__":
    main()
