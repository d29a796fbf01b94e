#!/bin/bash
# dev sweep on a GPU box: bash tools/probe/tune_sweep.sh "<key=value ...>" ...   one bench line (median of 5 x 30 steps) per argument; "" = defaults
for t in "$@"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes $args 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['ms_per_step_repeats']; print('%-24s median %.4f  min %.4f  max %.4f' % ('$t' or 'default', r['median'], r['min'], r['max']))"
done
