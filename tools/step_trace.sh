#!/bin/bash
# Kernel sequence of one replayed training step:  bash tools/step_trace.sh <out-file> [bench.py args...]   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$(dirname "$OUT")"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd_st
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pd_st -o d -- python3 $R/bench.py --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
DB=$(ls /tmp/pd_st/*/*.db /tmp/pd_st/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_sequence.py $DB > $OUT 2>&1
