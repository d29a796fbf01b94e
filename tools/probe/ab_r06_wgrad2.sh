#!/bin/bash
# timing-only builds (results WRONG; switches FN_EXP_SKIP_W0 / FN_EXP_SKIP_W128 in k_wgrad_all, see ab_r06_wgrad.sh): the K = 128 group alone
# under both forms (FN_TUNE_WGRAD_DIRECT 1 = LDS ring, 2 = register-streamed) and several workgroup targets
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export FRAGNET_EXTRA_HIPCC_FLAGS="-DFN_EXP_SKIP_W0=1"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for cfg in "1 192" "2 192" "1 256" "2 256" "2 128" "1 128"; do
  set -- $cfg
  bash tools/step_trace.sh gpurun_out/seq_wg.txt --steps 12 --warmup 3 --epoch-batches 0 --no-round3-shapes --tune 8=$1 --tune 3=$2
  echo "K = 128 group alone, form $1, $2 workgroups: $(grep -E 'k_wgrad_all' gpurun_out/seq_wg.txt | awk '{printf "%s %s wg %s us; ", $2, $4, $6}')"
done
