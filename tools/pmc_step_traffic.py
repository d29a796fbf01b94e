#!/usr/bin/env python3
"""HBM-side bytes of every kernel of ONE replayed training step, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
the default bench command, held against the step's algorithmic bytes.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ps_fetch -o p -- python3 $REPO/bench.py --steps 8 --warmup 2 \\
              --no-cpu-baseline --no-roofline --epoch-batches 0
    rocprofv3 --pmc WRITE_SIZE ... -d /tmp/ps_write ...          (separate passes: the two counters do not fit one TCC pass)
    python tools/pmc_step_traffic.py /tmp/ps_fetch /tmp/ps_write <step_algorithmic_GB> [step_sequence.txt] > profiles/r04_pmc_step.json

With the step sequence of a kernel trace of the same command (tools/rocpd_sequence.py) every kernel also gets its duration there and
the bandwidth it MOVED (hbm_MB / us): what the launch actually pulled through the fabric, to be held against what a streaming kernel
of that size gets (profiles/r06_cold_stream_probe.txt: 7+ TB/s hot or clean-cold, 3.3-3.7 behind a pass that left the memory-side
cache dirty -- the case profiles/r04_hbm_cold_stream.md measured), not against the 8 TB/s of the roofline.
`--join <existing.json> <step_sequence.txt>` adds those two columns to a table collected earlier.

One step = the dispatches between the last two k_stage_padded launches of the run (dispatch-id order).  Units and the gfx950
correction as MI355X_MICROARCH.md prescribes: both counters are KiB; FETCH_SIZE counts 128-byte requests at 64 bytes -> doubled;
WRITE_SIZE is taken as read (exact for 16-byte-per-lane stores; scattered 4-byte stores are uncalibrated and read high)."""
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:48]


def one_step(path, counter):
    rows = []
    for f in glob.glob(f"{path}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])),
                             float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[1] == "k_stage_padded"]
    a, b = marks[-2], marks[-1]
    return rows[a:b]


def join_sequence(out, seq_path):
    rows = [l.split() for l in open(seq_path) if l.strip() and l.strip()[0].isdigit()]
    durs = [(r[1], int(r[3]), float(r[5])) for r in rows]                 # "<i> <kernel> blocks <n> dur <us> us ..."
    if [(k, g) for k, g, _ in durs] != [(e["kernel"], e["workgroups"]) for e in out["sequence"]]:
        out["durations_from"] = f"{seq_path}: kernel sequence differs from the counter passes, not joined"
        return out
    tot = 0.0
    for e, (_, _, us) in zip(out["sequence"], durs):
        e["us_in_step"] = us
        e["moved_TBps"] = round(e["hbm_MB"] / us, 3) if us > 0 else None           # MB / us = TB/s
        tot += us
    out["durations_from"] = seq_path + " (rocprofv3 --kernel-trace of the same command, another run)"
    out["moved_TBps_whole_step"] = round(out["hbm_GB"] * 1e3 / tot, 3)
    return out


def main():
    if sys.argv[1] == "--join":
        print(json.dumps(join_sequence(json.load(open(sys.argv[2])), sys.argv[3]), indent=1))
        return
    fetch = one_step(sys.argv[1], "FETCH_SIZE")
    write = one_step(sys.argv[2], "WRITE_SIZE")
    algo_gb = float(sys.argv[3]) if len(sys.argv) > 3 else None
    assert [r[1:3] for r in fetch] == [r[1:3] for r in write], "the two passes saw different kernel sequences"
    seq, tot_f, tot_w = [], 0.0, 0.0
    for (_, k, g, fv), (_, _, _, wv) in zip(fetch, write):
        f, w = fv * 1024 * 2, wv * 1024
        tot_f += f
        tot_w += w
        seq.append({"kernel": k, "workgroups": g, "fetch_MB_x2": round(f / 1e6, 2), "write_MB": round(w / 1e6, 2), "hbm_MB": round((f + w) / 1e6, 2)})
    out = {"what": "HBM-side traffic per kernel of one replayed training step (ESOL-shape batch of 512, the default bench command), "
                   "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950), KiB -> bytes",
           "kernels_per_step": len(seq), "fetch_GB_x2": round(tot_f / 1e9, 4), "write_GB": round(tot_w / 1e9, 4),
           "hbm_GB": round((tot_f + tot_w) / 1e9, 4), "step_algorithmic_GB": algo_gb,
           "traffic_over_algorithmic": round((tot_f + tot_w) / 1e9 / algo_gb, 3) if algo_gb else None, "sequence": seq}
    if len(sys.argv) > 4:
        out = join_sequence(out, sys.argv[4])
    # the kernel sources these counters belong to (bench.py marks the table STALE when its library was built from others)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fragnet_amd import build
    out["source_digest"] = build.source_digest()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
