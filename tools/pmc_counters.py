#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters (CSV output), optionally restricted to kernels whose name contains a pattern.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \\
              SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_sq -o p \\
              -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU \\
              --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_ta -o p -- python3 $REPO/bench.py ...(same)
    python tools/pmc_counters.py gpurun_out/pmc_sq k_linear128 > profiles/pmc_mfma_gemm.json
    python tools/pmc_counters.py gpurun_out/pmc_sq,gpurun_out/pmc_ta k_gat_ > profiles/pmc_ta_scatter.json

(--pmc runs carry --kernel-trace only, never a runtime / sys trace: the GPU pool refuses that combination.)
Units as documented in MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles; SQ_BUSY_CYCLES is summed over the shader engines; GRBM_GUI_ACTIVE over the 8 XCDs."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def main():
    dirs = sys.argv[1].split(",")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                if pat and pat not in k:
                    continue
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r and r.get("End_Timestamp"):
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, cs in sorted(agg.items()):
        e = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
        e["dispatches_averaged"] = max(len(v) for v in cs.values())
        if dur[k]:
            e["avg_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]) / 1e3, 2)
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in e:
                    e[c + "/SQ_WAVE_CYCLES"] = round(e[c] / e["SQ_WAVE_CYCLES"], 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
            e["SQ_VALU_MFMA_BUSY_CYCLES/SQ_BUSY_CYCLES"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"], 3)
        out[k] = e
    out["_note"] = ("per-dispatch means; collected with rocprofv3 --pmc (+ --kernel-trace only) over bench.py's replayed training step "
                    "(ESOL-shape batch 512); tools/pmc_counters.py; counter units: MI355X_MICROARCH.md 'Per-instruction cycle constants' "
                    "and 'rocprofv3 PMC slots'")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
