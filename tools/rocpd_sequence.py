#!/usr/bin/env python3
"""Kernel sequence of ONE training step from a rocprofv3 results .db (rocpd sqlite): the dispatches between two consecutive
k_stage_padded launches (the step of median wall time among the last nine), in start order, with duration and the gap to the
previous kernel's end.

usage: tools/rocpd_sequence.py <..._results.db> [anchor-kernel-substring]"""
import re
import sqlite3
import sys


def short(n):
    m = re.match(r"_ZN\d+_GLOBAL__N_1\d+(k_[a-z0-9_]+)", n)
    if m:
        return m.group(1)
    if "copyBuffer" in n:
        return "COPY(copyBuffer)"
    if "FillFunctor" in n:
        return "FILL(torch)"
    if "Cijk" in n:
        return "library GEMM"
    return re.sub(r"^void ", "", n)[:48]


def main():
    con = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_stage_padded"
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [c[1] for c in cur.execute(f"pragma table_info({sym})")]
    namecol = "kernel_name" if "kernel_name" in cols else "display_name"
    names = {r[0]: r[1] for r in cur.execute(f"select id,{namecol} from {sym}")}
    rows = list(cur.execute(f"select kernel_id,start,end,grid_size_x,workgroup_size_x from {disp} order by start"))
    seq = [(short(names[k]), s, e, g // max(1, w)) for k, s, e, g, w in rows]
    idx = [i for i, x in enumerate(seq) if anchor in x[0]]
    # the step of MEDIAN wall time among the last (up to) nine complete ones of the same length: a single sample now and then
    # catches a stall between two kernels
    cands = sorted((seq[q][1] - seq[p][1], p, q) for p, q in zip(idx[-10:-1], idx[-9:]) if q - p == idx[-1] - idx[-2])
    _, a, b = cands[len(cands) // 2] if cands else (0, idx[-2], idx[-1])
    prev_end, busy = None, 0
    print(f"# one step = dispatches between two consecutive {anchor} launches (median of the last {max(len(cands), 1)} steps): "
          f"{b - a} kernels, {(seq[b][1] - seq[a][1]) / 1e3:.1f} us wall")
    for i in range(a, b):
        n, s, e, g = seq[i]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        busy += e - s
        print(f"{i - a:3d} {n:30s} blocks {g:6d} dur {(e - s) / 1e3:7.2f} us  gap {gap:6.2f} us")
        prev_end = e
    print(f"# GPU busy {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
