#!/bin/bash
# A/B of the product epilogue's early operand requests (FN_CU_EARLY rows at step group FN_CU_EARLY_AT) in one gpurun call:
#   bash tools/probe/ab_r06_cu_early.sh "0 5" "2 5" "2 2" "3 4"      -> step time, k_lin_rd_cu launches, GPU-busy per configuration
cd ${GRAFT_REPO_ROOT:-$(pwd)}
CFGS=("$@")
for rep in 1 2; do for cfg in "${CFGS[@]}"; do
  set -- $cfg
  export FRAGNET_EXTRA_HIPCC_FLAGS="-DFN_CU_EARLY=$1 -DFN_CU_EARLY_AT=$2"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg] ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'], 'loss', d['final_loss'])"
  if [ $rep = 2 ]; then
    bash tools/step_trace.sh gpurun_out/seq_ab.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes
    echo "[$cfg] k_lin_rd_cu $(grep -E 'k_lin_rd_cu' gpurun_out/seq_ab.txt | awk '{printf "%s ", $6}') busy $(grep 'GPU busy' gpurun_out/seq_ab.txt | awk '{print $4}')"
  fi
done; done
