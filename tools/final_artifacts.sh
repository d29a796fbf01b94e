#!/bin/bash
# Regenerates the per-round evidence set on a GPU box:  bash tools/final_artifacts.sh <tag>   (writes gpurun_out/<tag>/, copy into profiles/)
set -u
TAG=${1:-r06}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -1 $O/pytest_gpu.txt
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
for r in 2 4 8; do timeout 300 python3 bench.py --no-cpu-baseline --no-roofline --shard-of $r 2>/dev/null | tail -1; done > $O/strong_shards.json
for v in gat2_lite gat2_edge; do timeout 300 python3 bench.py --no-cpu-baseline --no-roofline --model-version $v 2>/dev/null | tail -1; done > $O/variants.json
timeout 600 python3 bench.py --forward-sweep 2>/dev/null > $O/forward_sweep.json
timeout 900 python3 bench.py --forward-sweep --store 1048576 2>/dev/null > $O/store_sweep.json
timeout 300 python3 tools/tox21_bench.py > $O/tox21.txt 2>&1
timeout 300 python3 tools/pretrain_bench.py > $O/pretrain.txt 2>&1
timeout 300 python3 bench.py --gpus 1 --spawn --no-cpu-baseline --no-roofline > $O/bench_spawn_1rank.json 2> $O/bench_spawn_1rank.err
FRAGNET_BENCH_BACKEND=gloo timeout 300 python3 bench.py --gpus 2 --overlap off --no-cpu-baseline --no-roofline --steps 10 > $O/bench_2ranks_gloo_one_gpu.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd /tmp/pd2 /tmp/pmc_fetch /tmp/pmc_write
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pd -o d -- python3 $R/bench.py --no-cpu-baseline --no-roofline --epoch-batches 0 > /dev/null 2>&1
DB=$(ls /tmp/pd/*/*.db /tmp/pd/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/kernel_trace_summary.md 2>&1
python3 $R/tools/rocpd_summary.py $DB --json "rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-roofline --epoch-batches 0 ($TAG, whole-step hipGraph replays; k_gat_fwd_pair and the smallest k_gat_bwd_one3 grid are layer 0's launches: bond + fragment-bond levels and nothing else)" > $O/in_graph_kernels.json 2>/dev/null
python3 $R/tools/rocpd_sequence.py $DB > $O/step_sequence.txt 2>&1
# the default bench line again, now that roofline.in_graph can be read from a trace of THIS library (bench.py drops the figures
# when profiles/in_graph_kernels.json carries another source digest)
cp $O/in_graph_kernels.json $R/profiles/in_graph_kernels.json
(cd $R && timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err); cut -c1-200 $O/bench.json
bash $R/tools/step_trace.sh $O/step_sequence_shard_of_8.txt --shard-of 8 --steps 20 --warmup 5 --epoch-batches 0
cd /tmp
rm -rf /tmp/pdp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pdp -o d -- python3 $R/tools/pretrain_bench.py > /dev/null 2>&1
DBP=$(ls /tmp/pdp/*/*.db /tmp/pdp/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_sequence.py $DBP > $O/pretrain_step_sequence.txt 2>&1
python3 $R/tools/rocpd_summary.py $DBP > $O/pretrain_kernel_trace_summary.md 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pd2 -o d -- python3 $R/bench.py > $O/bench_under_rocprof.json 2>/dev/null
DB2=$(ls /tmp/pd2/*/*.db /tmp/pd2/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_summary.py $DB2 > $O/bench_full_kernel_trace_summary.md 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -o p -- python3 $R/bench.py --kernels-only > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -o p -- python3 $R/bench.py --kernels-only > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write > $O/pmc_per_launch.json 2>/dev/null
# whole-step traffic table: FETCH_SIZE x 2 / WRITE_SIZE of every kernel of one replayed step against the step's algorithmic bytes
rm -rf /tmp/ps_fetch /tmp/ps_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ps_fetch -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --epoch-batches 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ps_write -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --epoch-batches 0 > /dev/null 2>&1
ALGO=$(python3 -c "import json,sys; print(json.loads(open('$O/bench.json').read().strip().splitlines()[-1])['step_algorithmic_GB'])")
python3 $R/tools/pmc_step_traffic.py /tmp/ps_fetch /tmp/ps_write $ALGO $O/step_sequence.txt > $O/pmc_step.json 2> $O/pmc_step.err
rm -rf /tmp/ph
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ph -o d -- python3 $R/tools/probe/hbm_cold_probe.py > /dev/null 2>&1
python3 $R/tools/probe/hbm_cold_probe.py --summarise $(ls /tmp/ph/*/*.db /tmp/ph/*.db 2>/dev/null | head -1) > $O/hbm_cold_stream_table.md 2>&1
# A/B in the same call: the general two-pass backward (22=0) and the deferred form of the one-pass backward (29=1: no second forward output;
# the destination-side term is applied by the input-gradient GEMM and the weight gradient, HISTORY.md section 4h)
timeout 300 python3 $R/bench.py --no-cpu-baseline --no-roofline --epoch-batches 0 --tune 22=0 2>/dev/null | tail -1 > $O/bench_two_pass.json
timeout 300 python3 $R/bench.py --no-cpu-baseline --no-roofline --epoch-batches 0 --tune 29=1 2>/dev/null | tail -1 > $O/bench_deferred_form.json
bash $R/tools/step_trace.sh $O/step_sequence_deferred_form.txt --steps 20 --warmup 5 --epoch-batches 0 --tune 29=1
# the riders of round 4 off (last Linear / loss / its backward as three launches; one Adam launch over everything)
(cd $R && FRAGNET_FUSED_HEAD_LOSS=0 FRAGNET_ADAM_RIDER=0 timeout 300 python3 bench.py --no-cpu-baseline --no-roofline --epoch-batches 0 2>/dev/null | tail -1) > $O/bench_no_riders.json
# the default bench line once more, now that every table it reads (in-graph trace, per-launch and whole-step PMC) was collected with THIS
# library's sources (bench.py drops tables that carry another source digest): this is the line to commit as profiles/${TAG}_bench.json
cp $O/pmc_per_launch.json $R/profiles/pmc_per_launch.json
cp $O/pmc_step.json $R/profiles/${TAG}_pmc_step.json
(cd $R && timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err); cut -c1-200 $O/bench.json
# SQ / TA counters of every kernel of the replayed step (two more --pmc passes)
(cd $R && bash tools/pmc_step_sq_ta.sh $TAG)
ls -la $O
