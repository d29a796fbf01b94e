#!/bin/bash
# SQ / TA counters of the replayed step's kernels (two separate --pmc passes, --kernel-trace only), summarised on the box: every kernel of the step
#   bash tools/pmc_step_sq_ta.sh <tag>  ->  gpurun_out/<tag>/pmc_sq_ta_step.json (+ .md by tools/pmc_counters.py --md)
set -u
TAG=${1:-r06}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq /tmp/pmc_ta
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes > /dev/null 2>&1
timeout 600 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc_ta -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes > /dev/null 2>&1
python3 $R/tools/pmc_counters.py /tmp/pmc_sq,/tmp/pmc_ta k_ > $O/pmc_sq_ta_step.json 2> $O/pmc_err.txt
ls -la $O/pmc_sq_ta_step.json
