#!/bin/bash
# A/B of FN_TUNE_GEMM_COLAUNCH (projection GEMM workgroups inside the attention launches; layer-0 merges) + the GPU parity tests
set -u
mkdir -p gpurun_out/r02e
rm -f gpurun_out/r02e/ab.txt
timeout 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02e/tests.txt 2>&1; tail -3 gpurun_out/r02e/tests.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8"
run() { echo "== $*" >> gpurun_out/r02e/ab.txt; timeout 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/r02e/ab.txt 2>&1; }
run --tune 14=0
run
run --tune 14=0
run
cat gpurun_out/r02e/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pd -o d -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline > /dev/null 2>&1
DB=$(ls /tmp/pd/*/*.db /tmp/pd/*.db 2>/dev/null | head -1)
python3 /root/repo/tools/rocpd_sequence.py $DB > /root/repo/gpurun_out/r02e/step_sequence.txt 2>&1
python3 /root/repo/tools/gemm_scaling_probe.py > /root/repo/gpurun_out/r02e/gemm_scaling.txt 2>&1
