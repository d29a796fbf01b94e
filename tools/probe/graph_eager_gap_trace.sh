#!/bin/bash
# gaps between kernels of tools/probe/graph_eager_gap_probe.py under rocprofv3 --kernel-trace: inside a replay, replay -> replay, replay -> eager kernel -> replay
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gg
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gg -o t -- python3 $R/tools/probe/graph_eager_gap_probe.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, statistics
f = glob.glob("/tmp/gg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])) for r in csv.DictReader(open(f))))
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
durs = [(e - s) / 1e3 for s, e, _ in rows]
print("kernels", len(rows), "median duration %.2f us" % statistics.median(durs))
big = sorted(g for g in gaps if g > 0.5)
small = [g for g in gaps if g <= 0.5]
print("gaps <= 0.5 us: %d (median %.3f us);  gaps > 0.5 us: %d" % (len(small), statistics.median(small) if small else 0, len(big)))
import collections
hist = collections.Counter(round(g) for g in big)
print("histogram of the larger gaps (us -> count):", sorted(hist.items())[:25])
PY
