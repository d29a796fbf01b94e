#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV into a per-kernel (and per-grid-size) table.

usage: tools/rocprof_summary.py <..._kernel_trace.csv> [steps] > profiles/rNN_summary.md
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>]+?)(\(|$)", name)
    base = name.split("(")[0]
    if base.startswith("Cijk_"):
        mt = re.search(r"MT\d+x\d+x\d+", base)
        return "rocBLAS/hipBLASLt GEMM " + (mt.group(0) if mt else "") + (" Alik" if "Alik" in base[:12] else " Ailk") + ("_Bljk" if "Bljk" in base[:18] else "_Bjlk")
    if base.startswith("at::native::"):
        base = base[len("at::native::"):]
        f = re.search(r"(CUDAFunctor_add|FillFunctor|sum_functor|MulFunctor|[A-Za-z]+Functor[A-Za-z_]*|multi_tensor_apply_kernel|mse|threshold)", name)
        return "torch " + base.split("<")[0] + (":" + f.group(1) if f else "")
    return base


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(csv.DictReader(open(path)))
    per = defaultdict(list)
    per_grid = defaultdict(list)
    for r in rows:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        per[k].append(dur)
        grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        per_grid[(k, grid)].append(dur)
    total = sum(sum(v) for v in per.values())
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    print(f"# {path}\n")
    print(f"dispatches: {len(rows)}; GPU busy {total/1e6:.2f} ms of {(t1-t0)/1e6:.2f} ms traced wall\n")
    print("| kernel | calls | total ms | avg us | % busy |")
    print("|---|---|---|---|---|")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"| {k} | {len(v)} | {sum(v)/1e6:.3f} | {sum(v)/len(v)/1e3:.2f} | {100*sum(v)/total:.1f} |")
    print("\n## fragnet kernels by launch grid (workgroups)\n")
    print("| kernel | grid | calls | avg us | min us | max us |")
    print("|---|---|---|---|---|---|")
    for (k, g), v in sorted(per_grid.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        if k.startswith("k_"):
            print(f"| {k} | {g} | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} |")


if __name__ == "__main__":
    main()
