#!/bin/bash
# launch-shape sweep of the attention passes with the projection GEMMs riding along (round 2, after the co-launch change)
set -u
mkdir -p gpurun_out/r02d
rm -f gpurun_out/r02d/sweep2.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8"
run() { echo "== $*" >> gpurun_out/r02d/sweep2.txt; timeout 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/r02d/sweep2.txt 2>&1; }
run
for v in 512 640 896 1024; do run --tune 0=$v; done
for v in 1024 1280 2048; do run --tune 11=$v; done
for v in 384 640 768; do run --tune 12=$v; done
for v in 128 384; do run --tune 13=$v; done
for v in 192 320; do run --tune 3=$v; done
run
cat gpurun_out/r02d/sweep2.txt
