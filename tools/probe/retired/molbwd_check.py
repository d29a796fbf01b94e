#!/usr/bin/env python3
"""Dev loop for csrc/mol_bwd.hip: the molecule-resident single-pass backward of an attention level against the two-pass
kernels (fn_gat_bwd_dst_f32 + fn_gat_bwd_src_f32) on the same inputs -- outputs compared, both timed back to back.
    python tools/molbwd_check.py [--batch 512] [--profile esol] [--iters 50] [--slow]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fragnet_amd import _lib, data, synth            # noqa: E402
from fragnet_amd.plan import GraphPlan, _stream_ptr  # noqa: E402
from fragnet_amd._lib import SegPlan                 # noqa: E402


def seg(s):
    return SegPlan(s.rowptr.data_ptr(), s.perm.data_ptr(), s.index.data_ptr(), s.n_seg, s.n_items, s.pos_base, 0)


def mol_extents(plan, dev):
    ext = torch.zeros(plan.n_mols, 16, dtype=torch.int32, device=dev)
    L = plan.levels
    _lib.call("fn_mol_extents", C.byref(seg(plan.segs["mol_atoms"])), C.byref(seg(plan.segs["mol_frags"])), C.byref(L["bond"].c),
              C.byref(L["atom"].c), C.byref(L["fbond"].c), C.byref(L["frag"].c), plan.n_mols, ext.data_ptr(), _stream_ptr(dev))
    return ext


def timed(fn, iters):
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


def run_level(batch, plan, ext, name, which, per_unit, iters, dev, seed=0, stamps=False):
    lv = plan.levels[name]
    n, m, H = lv.n, lv.m, 4
    g = torch.Generator(device="cpu").manual_seed(seed)
    f32 = dict(dtype=torch.float32, device=dev)
    h = torch.randn(n, 128, generator=g).to(dev)
    gout = torch.randn(n, 128, generator=g).to(dev)
    mode2 = name in ("bond", "fbond")
    if mode2:
        att_w, src_off = 96, 64
        att = (torch.randn(H, att_w, generator=g) * 0.3).to(dev)
        raw = batch["edge_attr_bonds"] if name == "bond" else batch["edge_attr_fbonds"]
        x = plan.sorted_attr(name, raw)
        K = x.shape[0]
        embW, embb = (torch.randn(32, K, generator=g) * 0.3).to(dev), (torch.randn(32, generator=g) * 0.3).to(dev)
        et = _lib.EdgeTerm(2, K, 32, 32, None, x.data_ptr(), embW.data_ptr(), embb.data_ptr())
        keep = (x, embW, embb)
    else:
        att_w, src_off = 192, 160
        att = (torch.randn(H, att_w, generator=g) * 0.3).to(dev)
        s_sorted = (torch.randn(H, m, generator=g) * 0.5).to(dev)
        et = _lib.EdgeTerm(0, 0, 0, 0, s_sorted.data_ptr(), None, None, None)
        K = 0
        keep = (s_sorted,)
    st = _stream_ptr(dev)
    s_dst, s_src = torch.empty(n, H, **f32), torch.empty(n, H, **f32)
    out, p_sorted = torch.empty(n, 128, **f32), torch.empty(H, m, **f32)
    _lib.call("fn_node_scalars_f32", h.data_ptr(), att.data_ptr(), att_w, 0, src_off, s_dst.data_ptr(), s_src.data_ptr(), n, H, st)
    _lib.call("fn_gat_fwd_f32", h.data_ptr(), s_dst.data_ptr(), s_src.data_ptr(), att.data_ptr(), att_w, C.byref(et), C.byref(lv.c), 0.2,
              out.data_ptr(), p_sorted.data_ptr(), None, None, None, 0, None, H, st)
    # ---- two-pass reference
    ne = H * (K + 1)
    dz_orig = torch.zeros(lv.m_real, H, **f32)
    pz, g_s_dst, g_h = torch.empty(H, m, 2, **f32), torch.empty(n, H, **f32), torch.empty(n, 128, **f32)
    part_e, part_a = torch.zeros(4096, max(ne, 1), **f32), torch.zeros(256, 4096, **f32)
    n_e, n_a = C.c_int(0), C.c_int(0)

    def old():
        _lib.call("fn_gat_bwd_dst_f32", gout.data_ptr(), h.data_ptr(), p_sorted.data_ptr(), C.byref(et), C.byref(lv.c), 0.2,
                  None, None if mode2 else dz_orig.data_ptr(), pz.data_ptr(), g_s_dst.data_ptr(), part_e.data_ptr() if mode2 else None, C.byref(n_e), H, st)
        _lib.call("fn_gat_bwd_src_f32", gout.data_ptr(), h.data_ptr(), pz.data_ptr(), g_s_dst.data_ptr(), att.data_ptr(), att_w, 0, src_off,
                  C.byref(lv.c), g_h.data_ptr(), part_a.data_ptr(), C.byref(n_a), H, st)

    # ---- single pass
    dz2 = torch.zeros(lv.m_real, H, **f32)
    g_h2 = torch.empty(n, 128, **f32)
    part_e2, part_a2 = torch.zeros(4096, max(ne, 1), **f32), torch.zeros(256, 4096, **f32)
    scratch = torch.empty(H * m + n * H, **f32)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    n_p = C.c_int(0)

    def new():
        _lib.call("fn_gat_bwd_mol_f32", gout.data_ptr(), h.data_ptr(), p_sorted.data_ptr(), C.byref(et), att.data_ptr(), att_w, 0, src_off,
                  C.byref(lv.c), 0.2, ext.data_ptr(), plan.n_mols, which, per_unit, None, g_h2.data_ptr(), None if mode2 else dz2.data_ptr(),
                  part_a2.data_ptr(), part_e2.data_ptr() if mode2 else None, C.byref(n_p), scratch.data_ptr(), status.data_ptr(), H, st)

    torch.cuda.synchronize()
    print(f"[{name}] forward ok", file=sys.stderr, flush=True)
    old()
    torch.cuda.synchronize()
    print(f"[{name}] two-pass ok", file=sys.stderr, flush=True)
    new()
    torch.cuda.synchronize()
    print(f"[{name}] single-pass ok", file=sys.stderr, flush=True)
    res = {"level": name, "n": n, "m": m, "status": int(status.item()), "n_part_new": n_p.value, "n_part_old": (n_a.value, n_e.value)}
    res["g_h_maxdiff"] = float((g_h - g_h2).abs().max())
    res["g_h_scale"] = float(g_h.abs().max())
    if not mode2:
        res["dz_maxdiff"] = float((dz_orig - dz2).abs().max())
    else:
        a, b = part_e[: n_e.value].sum(0), part_e2[: n_p.value].sum(0)
        res["part_e_maxdiff"] = float((a - b).abs().max())
        res["part_e_scale"] = float(a.abs().max())
    a, b = part_a[:, : n_a.value].sum(1), part_a2[:, : n_p.value].sum(1)
    res["part_a_maxdiff"] = float((a - b).abs().max())
    res["part_a_scale"] = float(a.abs().max())
    D = 128
    bwd_b = 4 * (2 * n * D + 2 * m * H + 2 * m + n * D + m * H + 2 * n * H)
    t_old, t_new = timed(old, iters), timed(new, iters)
    res["us_two_pass"], res["us_single_pass"] = round(t_old, 2), round(t_new, 2)
    res["frac_two_pass"], res["frac_single_pass"] = round(bwd_b / t_old / 1e3 / 8000, 4), round(bwd_b / t_new / 1e3 / 8000, 4)
    res["algorithmic_MB"] = round(bwd_b / 1e6, 2)
    if stamps:
        nwg = n_p.value
        buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
        _lib.call("fn_debug_set_stamps", buf.data_ptr(), buf.numel())
        new()
        torch.cuda.synchronize()
        _lib.call("fn_debug_set_stamps", None, 0)
        t = buf.view(4096, 16)[:nwg].cpu().double()
        t = t[t[:, 14] > 0]                                      # workgroups that stamped (the fast ones)
        starts = (t[:, 14] - t[:, 14].min()) / 100.0
        res["wg_start_us_hist"] = {f"<{b}": int((starts < b).sum()) for b in (1, 2, 5, 10, 15, 20, 30, 40)}
        res["stamped_wgs"] = int(t.shape[0])
        ends = (t[:, 15] - t[:, 14].min()) / 100.0
        res["wg_end_us_hist"] = {f"<{b}": int((ends < b).sum()) for b in (10, 20, 30, 40, 60, 80, 100, 120, 140, 160)}
        res["wg_start_us_hist2"] = {f"<{b}": int((starts < b).sum()) for b in (20, 40, 60, 80, 100, 120, 140, 160)}
        names = ["extents", "loaded", "passA", "passB", "passC", "unit_sums", "partials"]
        wall = (t[:, 15] - t[:, 14]) / 100.0                     # us (wall_clock64 = 100 MHz)
        ticks = t[:, 7] - t[:, 0]
        rate = float((ticks / wall.clamp(min=0.01)).median())    # s_memtime ticks per us
        d = (t[:, 1:8] - t[:, 0:7]) / rate
        res["tick_per_us"] = round(rate, 1)
        res["wg_us_median"] = {nm: round(float(d[:, i].median()), 2) for i, nm in enumerate(names)}
        res["wg_us_p90"] = {nm: round(float(d[:, i].quantile(0.9)), 2) for i, nm in enumerate(names)}
        res["wg_total_us"] = {"median": round(float(wall.median()), 2), "p90": round(float(wall.quantile(0.9)), 2), "max": round(float(wall.max()), 2)}
        res["launch_span_us"] = round(float((t[:, 15].max() - t[:, 14].min()) / 100.0), 2)      # wall_clock64 is chip-wide
    del keep
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--profile", default="esol")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--slow", action="store_true", help="force the global-memory path of every unit (FN_TUNE_BWD_MOL_FORCE_SLOW)")
    ap.add_argument("--levels", default="bond,atom,fbond,frag")
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--per", type=int, default=0, help="molecules per workgroup unit (0: the engine's choice per level)")
    ap.add_argument("--stamps", action="store_true", help="phase time stamps of the single-pass kernel (median over workgroups)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.slow:
        _lib.call("fn_set_tuning", 19, 1)
    if args.dbg:
        _lib.call("fn_set_tuning", 19, args.dbg)
    batch = data.batch_to(data.collate_fn(synth.synth_molecules(args.batch, seed=1000, profile=args.profile)), dev)
    plan = GraphPlan.from_batch(batch)
    ext = mol_extents(plan, dev)
    per = {"bond": (0, 1), "atom": (1, 2), "fbond": (2, 8), "frag": (3, 16)}
    for name in args.levels.split(","):
        which, pu = per[name]
        pu = args.per or pu
        res = run_level(batch, plan, ext, name, which, pu, args.iters, dev, stamps=args.stamps)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
