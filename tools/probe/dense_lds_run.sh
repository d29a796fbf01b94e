#!/bin/bash
# bash tools/probe/dense_lds_run.sh <tag>: the head re-tile probe on a GPU box -> gpurun_out/dense_lds_<tag>*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-a}
B=$R/tools/probe/dense_lds_probe.bin
O=$R/gpurun_out
$B > $O/dense_lds_$tag.txt 2>&1
$B cold > $O/dense_lds_${tag}_cold.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dl_trace /tmp/dl_pmc1 /tmp/dl_pmc2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dl_trace -o t -- $B cold > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $(find /tmp/dl_trace -name '*kernel_trace.csv' | head -1) > $O/dense_lds_${tag}_trace_cold.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/dl_pmc1 -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/dl_pmc2 -o p -- $B > /dev/null 2>&1
python3 - <<PY > $O/dense_lds_${tag}_pmc.md 2>&1
import csv, glob, collections
for d in ("/tmp/dl_pmc1", "/tmp/dl_pmc2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60] + " @" + str(int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n =", len(next(iter(cs.values()))))
PY
