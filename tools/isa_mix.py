#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop(s), from the device assembly of the shipped sources.

    python tools/isa_mix.py gat_fwd 'k_gat_fwd_pair<4, 1, 8, true, true>' [--loops 2] [--dump]

Compiles fragnet_amd/csrc/<unit>.hip with the library's own flags to assembly (hipcc -S --cuda-device-only; cross-compiles
without a GPU), finds the kernel by its demangled name, takes its loops (a backward branch to a label) largest body first and
counts the instructions of each body by class.  One iteration of the row loops = the two rows of a wave (one per half-wave).
Static counts: a tier's loads inside a wave-uniform branch count once, whether the wave takes it or not.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = [
    ("vmem load", r"^(global_load|buffer_load|flat_load|scratch_load)"),
    ("vmem store", r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic)"),
    ("lds / bpermute", r"^ds_"),
    ("mfma", r"^v_mfma"),
    ("dpp / lane", r"(_dpp$|^v_readlane|^v_readfirstlane|^v_writelane|^v_permlane|^v_mov_b32_dpp)"),
    ("transcendental", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"),
    ("fp32 fma / mul / add", r"^v_(fma|fmac|mul|add|sub|mac|pk_fma|pk_mul|pk_add)_f32"),
    ("fp32 max / min / cmp-select", r"^v_(max|min|max3|min3|med3)_f32|^v_cmp.*_f32|^v_cndmask"),
    ("int multiply (Philox, offsets)", r"^v_(mul_hi_u32|mul_lo_u32|mul_u32_u24|mul_i32_i24|mad_u64_u32|mad_u32_u24|mad_i32_i24)"),
    ("int add / shift / logic (addresses, Philox xor)", r"^v_(add|sub|subrev|lshl|lshr|ashr|and|or|xor|xad|not|bfe|bfi|lshl_add|add_lshl|lshl_or|and_or|or3|add3|addc|add_co|min_[iu]|max_[iu])"),
    ("int cmp", r"^v_cmp"),
    ("v_mov / cvt / other VALU", r"^v_"),
    ("s_waitcnt", r"^s_waitcnt"),
    ("s_nop / barrier / branch", r"^s_(nop|barrier|cbranch|branch|sleep|setprio|sethalt)"),
    ("scalar memory", r"^s_(load|buffer_load|store)"),
    ("scalar ALU", r"^s_"),
]


def classify(op, dpp):
    if dpp and op.startswith("v_"):
        return "dpp / lane"
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other"


def assembly(unit):
    from fragnet_amd import build
    src = os.path.join(build.HERE, "csrc", unit + ".hip")
    out = os.path.join("/tmp", f"isa_mix_{unit}.s")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(p) for p in build.SOURCES + build.INCLUDED):
        cmd = [build._hipcc(), *build.FLAGS, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(build.HERE, "csrc"),
               "-S", "--cuda-device-only", src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read().splitlines()


def kernel_body(lines, want):
    labels = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(lines[i].split(":")[0] for i in labels), capture_output=True, text=True).stdout.splitlines()
    for i, nm in zip(labels, names):
        if want in nm.replace("(anonymous namespace)::", ""):
            end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm"))
            return nm, lines[i:end + 1]
    raise SystemExit(f"kernel {want!r} not found; have e.g. {names[:5]}")


def loops(body):
    """[(start, end)] of loop bodies: label .. the backward branch to it."""
    pos = {}
    out = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            pos[m.group(1)] = i
        m = re.match(r"^\s+s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in pos:
            out.append((pos[m.group(1)], i))
    # largest first; a loop that shares more than half of its lines with a larger one already kept is the same loop seen through
    # another back edge (or an inner loop of it: the hub walks, which the static count of the outer body already holds)
    keep = []
    for s, e in sorted(set(out), key=lambda se: se[0] - se[1]):
        if all(min(e, e2) - max(s, s2) < 0.5 * (e - s) for s2, e2 in keep):
            keep.append((s, e))
    return keep


def mix(body, start, end):
    c = collections.Counter()
    vm_bytes = collections.Counter()
    for l in body[start:end + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", l)
        if not m or m.group(1).startswith("."):
            continue
        op, rest = m.group(1), m.group(2)
        cls = classify(op, "row_" in rest or "quad_perm" in rest or "row_mirror" in rest or "row_half_mirror" in rest or "wave_" in rest)
        c[cls] += 1
        w = re.search(r"(dwordx(\d)|dword|b(\d+)|ushort|ubyte|short|byte)", op)
        if cls.startswith("vmem"):
            width = {"dword": 4}.get(w.group(1), None) if w else None
            if w and w.group(2):
                width = 4 * int(w.group(2))
            vm_bytes[cls + f" x{width or '?'}B"] += 1
    return c, vm_bytes


def main():
    unit, want = sys.argv[1], sys.argv[2]
    n_loops = int(sys.argv[sys.argv.index("--loops") + 1]) if "--loops" in sys.argv else 2
    lines = assembly(unit)
    name, body = kernel_body(lines, want)
    n_all = sum(1 for l in body if re.match(r"^\s+[a-z]", l))
    short = name.replace("(anonymous namespace)::", "")
    print(f"## `{short}` ({unit}.hip): {n_all} instructions in all")
    meta = [l.strip() for l in lines if want.split("<")[0] in l and ("vgpr_count" in l or "sgpr_count" in l)]
    for k, (s, e) in enumerate(loops(body)[:n_loops]):
        c, vb = mix(body, s, e)
        tot = sum(c.values())
        valu = sum(v for k2, v in c.items() if k2 not in ("vmem load", "vmem store", "lds / bpermute", "s_waitcnt", "s_nop / barrier / branch", "scalar memory", "scalar ALU", "mfma"))
        print(f"\nloop {k + 1}: lines {s}..{e} of the kernel, {tot} instructions per iteration, {valu} of them vector-ALU\n")
        print("| class | per iteration | share |\n|---|---|---|")
        for cls, _ in CLASSES + [("other", "")]:
            if c.get(cls):
                print(f"| {cls} | {c[cls]} | {c[cls] / tot:.1%} |")
        print("\n" + ", ".join(f"{k2}: {v}" for k2, v in sorted(vb.items())))
        if "--dump" in sys.argv:
            print("\n".join(body[s:e + 1]))


if __name__ == "__main__":
    main()
