#!/bin/bash
# A/B of a tuning key in one gpurun call: bash tools/probe/ab_r06_tune.sh KEY VAL_A VAL_B   (interleaved three times; step time, then both step sequences)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
K=$1; A=$2; B=$3
for rep in 1 2 3; do for v in $A $B; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes --tune $K=$v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$K=$v ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'], 'loss', d['final_loss'])"
done; done
for v in $A $B; do
  bash tools/step_trace.sh gpurun_out/seq_tune_$v.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes --tune $K=$v
  echo "$K=$v $(grep -E 'k_gat_' gpurun_out/seq_tune_$v.txt | awk '{printf "%s %s; ", $2, $6}') busy $(grep 'GPU busy' gpurun_out/seq_tune_$v.txt | awk '{print $4}')"
done
