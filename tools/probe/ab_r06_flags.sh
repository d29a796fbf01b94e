#!/bin/bash
# A/B of a compile-time switch in one gpurun call: bash tools/probe/ab_r06_flags.sh "<flags A>" "<flags B>"   (interleaved twice; step time + the forward launches)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for f in "$1" "$2"; do
  export FRAGNET_EXTRA_HIPCC_FLAGS="$f"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f] ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'], 'loss', d['final_loss'])"
done; done
for f in "$1" "$2"; do
  export FRAGNET_EXTRA_HIPCC_FLAGS="$f"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  bash tools/step_trace.sh gpurun_out/seq_ab.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes
  echo "[$f] $(grep -E 'k_gat_fwd' gpurun_out/seq_ab.txt | awk '{printf "%s %s; ", $2, $6}') busy $(grep 'GPU busy' gpurun_out/seq_ab.txt | awk '{print $4}')"
done
