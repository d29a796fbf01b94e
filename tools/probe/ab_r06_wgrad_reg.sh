#!/bin/bash
# the register-streamed K = 128 weight-gradient form (FN_TUNE_WGRAD_DIRECT = 2) against the LDS ring (1) at several workgroup targets (FN_TUNE_WGRAD_BLOCKS),
# one gpurun call: step time (first loop, min, median of five loops), then k_wgrad_all / k_reduce_tasks from a trace of the replayed step
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "1 192" "2 192" "2 224" "2 256" "1 256" "2 160"; do
  set -- $cfg
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes --tune 8=$1 --tune 3=$2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[form $1, $2 workgroups] ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'])"
  bash tools/step_trace.sh gpurun_out/seq_w.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes --tune 8=$1 --tune 3=$2
  echo "[form $1, $2 workgroups] $(grep -E 'k_wgrad_all|k_reduce_tasks' gpurun_out/seq_w.txt | awk '{printf "%s %s wg %s us; ", $2, $4, $6}')"
done
