#!/usr/bin/env python3
"""profiles/rNN_pmc_sq_ta_step.json (tools/pmc_step_sq_ta.sh -> tools/pmc_counters.py) as a table:
    python tools/pmc_sq_ta_md.py profiles/r06_pmc_sq_ta_step.json r06 > profiles/r06_pmc_sq_ta_step.md"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    tag = sys.argv[2] if len(sys.argv) > 2 else "?"
    print(f"# SQ / TA counters of every kernel of the replayed training step ({tag}, final code)\n")
    print(f"`bash tools/pmc_step_sq_ta.sh {tag}` on the GPU box: two separate `rocprofv3 --pmc ... --kernel-trace` passes over `bench.py --steps 8 --warmup 2 "
          "--no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes`, means per kernel name (`tools/pmc_counters.py`; raw means in the "
          ".json beside this file).  Durations are under the counter pass (a few % above the plain trace).\n")
    print("* `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the launch's cycles, GRBM_GUI_ACTIVE / 8): the share of SIMD-cycles the matrix pipe is busy")
    print("* `waiting` = SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a resident wave's cycles spent waiting for anything (memory, barriers)")
    print("* `wait inst` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting for an instruction-issue slot / dependency)")
    print("* `VALU` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave (x resident waves per SIMD = pipe utilisation)")
    print("* `TA busy` = TA_BUSY_avr / the launch's cycles")
    print("* `VALU insts`, `SALU`, `VMEM`, `LDS` = SQ_INSTS_* per launch (wave-instructions, thousands)\n")
    print("| kernel | launches averaged | µs | MFMA busy | waiting | wait inst | VALU per wave | TA busy | VALU insts k | SALU k | VMEM k | LDS k |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    rows = []
    for k, v in d.items():
        if not isinstance(v, dict) or "SQ_WAVE_CYCLES" not in v:
            continue
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
        wc = v["SQ_WAVE_CYCLES"] or 1
        rows.append((v.get("avg_us_under_pmc", 0) * v.get("dispatches_averaged", 1), k, v, cyc, wc))
    for _, k, v, cyc, wc in sorted(rows, reverse=True):
        print(f"| `{k}` | {v.get('dispatches_averaged')} | {v.get('avg_us_under_pmc', 0):.1f} | {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):.2f} | "
              f"{v.get('SQ_WAIT_ANY', 0) / wc:.2f} | {v.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.2f} | "
              f"{v.get('TA_BUSY_avr', 0) / cyc:.2f} | {v.get('SQ_INSTS_VALU', 0) / 1e3:.0f} | {v.get('SQ_INSTS_SALU', 0) / 1e3:.0f} | "
              f"{v.get('SQ_INSTS_VMEM', 0) / 1e3:.0f} | {v.get('SQ_INSTS_LDS', 0) / 1e3:.0f} |")


if __name__ == "__main__":
    main()
