#!/bin/bash
# A/B in one gpurun call: default (out2 in every layer) | deferred in every layer (29=1) | mixed (29=2), interleaved twice; then the mixed form's step sequence
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for t in 0 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes --tune 29=$t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('29=$t  ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'], 'loss', d['final_loss'])"
done; done
bash tools/step_trace.sh gpurun_out/seq_mixed.txt --steps 20 --warmup 5 --epoch-batches 0 --no-round3-shapes --tune 29=2
cat gpurun_out/seq_mixed.txt
