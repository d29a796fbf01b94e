#!/bin/bash
# timing-only builds (results WRONG): which half of k_wgrad_all is its critical path?  bash tools/probe/ab_r06_wgrad.sh
# needs the two switches in k_wgrad_all (fragnet_hip.hip): `#ifdef FN_EXP_SKIP_W128  if ((int)blockIdx.x < n128) return;  #endif` and
# `#ifdef FN_EXP_SKIP_W0  if ((int)blockIdx.x >= n128) return;  #endif` in front of its first branch -- they were removed again after the run
# that produced profiles/r06_wgrad_split_timing.txt (the library's source digest is part of every committed table)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for f in "" "-DFN_EXP_SKIP_W0=1" "-DFN_EXP_SKIP_W128=1"; do
  export FRAGNET_EXTRA_HIPCC_FLAGS="$f"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for wb in 192 256; do
    bash tools/step_trace.sh gpurun_out/seq_wg.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes --tune 3=$wb
    echo "flags [$f] FN_TUNE_WGRAD_BLOCKS=$wb: $(grep -E 'k_wgrad_all|k_reduce_tasks' gpurun_out/seq_wg.txt | awk '{printf "%s %s wg %s us; ", $2, $4, $6}')"
  done
done
