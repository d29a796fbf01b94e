#!/bin/bash
# timing-only builds (results WRONG: -DFN_EXP_SKIP_W0=1) of the register-streamed K = 128 weight-gradient form (tools/probe/retired/wgrad_reg_r06.patch applied)
# at several prefetch distances FN_WR_D (own 4-row steps in flight per wave: D x 3 KB x 8 waves per CU), form 1 = the LDS ring for reference
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for d in 3 6 8 1; do
  export FRAGNET_EXTRA_HIPCC_FLAGS="-DFN_EXP_SKIP_W0=1 -DFN_WR_D=$d"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for cfg in "2 256" "2 192"; do
    set -- $cfg
    bash tools/step_trace.sh gpurun_out/seq_wg.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes --tune 8=$1 --tune 3=$2
    echo "K = 128 group alone, register form, D = $d, $2 workgroups: $(grep -E 'k_wgrad_all' gpurun_out/seq_wg.txt | awk '{printf "%s us", $6}')"
  done
done
