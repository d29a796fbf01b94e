#!/bin/bash
# A/B runs of tuning keys + the GPU parity tests of the alternative kernels (dev loop)
set -u
mkdir -p gpurun_out/r02e
rm -f gpurun_out/r02e/ab.txt
timeout 900 python3 -m pytest tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/r02e/tests.txt 2>&1; tail -3 gpurun_out/r02e/tests.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8"
run() { echo "== $*" >> gpurun_out/r02e/ab.txt; timeout 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/r02e/ab.txt 2>&1; }
for a in "$@"; do run $a; done
cat gpurun_out/r02e/ab.txt
if [ "${SEQ:-0}" = "1" ]; then
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pd
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pd -o d -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline > /dev/null 2>&1
DB=$(ls /tmp/pd/*/*.db /tmp/pd/*.db 2>/dev/null | head -1)
python3 /root/repo/tools/rocpd_sequence.py $DB > /root/repo/gpurun_out/r02e/step_sequence.txt 2>&1
fi
