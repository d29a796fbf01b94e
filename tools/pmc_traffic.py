#!/usr/bin/env python3
"""HBM-side bytes per launch of the scatter kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --kernels-only
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --kernels-only
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_per_launch.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB;
FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is exact.  Separate passes because
the two counters do not fit one TCC pass."""
import collections
import csv
import json
import sys

# key in the output -> (kernel base name, wanted value of the forward's O2 template flag or None).  k_gat_fwd<H, KL, O2>: the plain
# forward and the one that also writes out2 / sigma (the one-pass backward's forward) share a base name.
KERNELS = {"k_gat_fwd": ("k_gat_fwd", False), "k_gat_fwd(+out2)": ("k_gat_fwd", True),
           # k_gat_bwd_one<H, KL, RB, DF>: the form the engine runs (DF = false) and the deferred form share a base name too
           "k_gat_bwd_one": ("k_gat_bwd_one", False), "k_gat_bwd_one(deferred form)": ("k_gat_bwd_one", True),
           "k_gat_cu": ("k_gat_cu", None), "k_gat_bwd_dst": ("k_gat_bwd_dst", None), "k_gat_bwd_src": ("k_gat_bwd_src", None)}


# which template argument carries the variant, and which of its spellings mean "on":
#   k_gat_fwd<H, KL, KIND>           KIND 0 plain | 1 second output | 2 second output, engine-constant instance (round 6; bool until then)
#   k_gat_bwd_one<H, KL, RB, DF, EN> DF = the deferred form (EN: engine-constant instance of the same pass, round 6)
_VARIANT_ARG = {"k_gat_fwd": 2, "k_gat_bwd_one": 3}
_ON = ("true", "1", "2")


def _template_args(name, base):
    """The kernel's template arguments as strings ('4', '1', 'true', ...), from a demangled or an Itanium-mangled name; None: another kernel."""
    import re
    if base + "<" in name:                          # demangled
        return [a.strip() for a in name.split(base + "<", 1)[1].split(">", 1)[0].split(",")]
    m = re.search(base + r"I((?:L[ib]\d+E)+)E", name)       # mangled: template arguments as Li4ELi1ELb1E
    if not m:
        return None
    return [("true" if v != "0" else "false") if t == "b" else v for t, v in re.findall(r"L([ib])(\d+)E", m.group(1))]


def _match(name, base, o2):
    args = _template_args(name, base)
    if args is None:
        return False
    if o2 is None:
        return True
    i = _VARIANT_ARG.get(base, len(args) - 1)
    on = i < len(args) and args[i] in _ON
    return on == o2


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{path}/p_counter_collection.csv")):
        if r["Counter_Name"] != counter:
            continue
        for key, (base, o2) in KERNELS.items():
            if _match(r["Kernel_Name"], base, o2):
                agg[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in KERNELS:
        if k in fetch and k in write:
            f, w = fetch[k] * 1024 * 2, write[k] * 1024
            out[k] = {"hbm_bytes_per_launch": int(f + w), "fetch_bytes_corrected_x2": int(f), "write_bytes": int(w),
                      "raw_FETCH_SIZE_KiB": round(fetch[k], 1), "raw_WRITE_SIZE_KiB": round(write[k], 1)}
    import datetime
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fragnet_amd import build
    out["_collected"] = "tools/final_artifacts.sh, " + datetime.date.today().isoformat()
    out["source_digest"] = build.source_digest()          # the kernel sources these counters belong to
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --kernels-only` (bond-graph level, "
                    "B=512 ESOL shape), averaged over the launches of each kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                    "section HBM; tools/pmc_traffic.py")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
