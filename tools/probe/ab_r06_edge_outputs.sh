#!/bin/bash
# A/B in one gpurun call: the finetune models ask the encoder for its bond / fragment-bond outputs (FRAGNET_KEEP_EDGE_OUTPUTS=1: the behaviour before)
# or not (default) -- training step and forward-only sweep, interleaved twice
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for v in 1 0; do
  FRAGNET_KEEP_EDGE_OUTPUTS=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[keep edge outputs = $v] training ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'], 'loss', d['final_loss'])"
  FRAGNET_KEEP_EDGE_OUTPUTS=$v python bench.py --forward-sweep 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('[keep edge outputs = $v] forward only', ' '.join(f\"{d['per_gpu_batch']}: {d['ms_per_step']} ms\" for d in r))"
done; done
