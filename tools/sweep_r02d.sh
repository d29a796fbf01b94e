#!/bin/bash
# launch-shape sweep of the projection GEMM and weight-gradient kernels + a refresh of the HBM-traffic counters (round 2, final shapes)
set -u
mkdir -p gpurun_out/r02d
B="python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8"
run() { echo "== $*" >> gpurun_out/r02d/sweep.txt; timeout 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/r02d/sweep.txt 2>&1; }
run
run --tune 1=1024
run --tune 1=768
run --tune 1=2048
run --tune 3=128
run --tune 3=192
run --tune 3=384
run
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -o p -- python3 $R/bench.py --kernels-only > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -o p -- python3 $R/bench.py --kernels-only > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write > $R/gpurun_out/r02d/pmc_per_launch.json 2> $R/gpurun_out/r02d/pmc_err.txt
