#!/bin/bash
# A/B of fn_encoder.no_backward in one gpurun call: the forward-only sweep with evaluation passes that save what a backward pass would read
# (FRAGNET_EVAL_SAVES=1: the behaviour before) and that do not (default), interleaved twice
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for v in 1 0; do
  FRAGNET_EVAL_SAVES=$v python bench.py --forward-sweep 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('[eval saves = $v]', ' '.join(f\"{d['per_gpu_batch']}: {d['ms_per_step']} ms ({d['value']/1e6:.3f} M/s)\" for d in r))"
done; done
