#!/bin/bash
# Copies what `bash tools/final_artifacts.sh <tag>` left under gpurun_out/<tag>/ into profiles/ under the names profiles/README.md uses:
#   bash tools/adopt_artifacts.sh r06      (run in the build container after the gpurun call has merged gpurun_out/)
set -eu
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
P=$R/profiles
cp $O/bench.json $P/${TAG}_bench.json
cp $O/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json
cp $O/bench_full_kernel_trace_summary.md $P/${TAG}_bench_full_kernel_trace_summary.md
cp $O/kernel_trace_summary.md $P/${TAG}_kernel_trace_summary.md
cp $O/step_sequence.txt $P/${TAG}_step_sequence.txt
cp $O/step_sequence_shard_of_8.txt $P/${TAG}_step_sequence_shard_of_8.txt
cp $O/step_sequence_deferred_form.txt $P/${TAG}_step_sequence_deferred_form.txt
cp $O/in_graph_kernels.json $P/in_graph_kernels.json
cp $O/pmc_per_launch.json $P/pmc_per_launch.json
cp $O/pmc_step.json $P/${TAG}_pmc_step.json
cp $O/pmc_sq_ta_step.json $P/${TAG}_pmc_sq_ta_step.json
python3 $R/tools/pmc_sq_ta_md.py $P/${TAG}_pmc_sq_ta_step.json $TAG > $P/${TAG}_pmc_sq_ta_step.md
cp $O/hbm_cold_stream_table.md $P/${TAG}_hbm_cold_stream.md
for f in bench_two_pass bench_deferred_form bench_no_riders bench_spawn_1rank bench_2ranks_gloo_one_gpu forward_sweep store_sweep strong_shards variants; do cp $O/$f.json $P/${TAG}_$f.json; done
cp $O/pretrain_step_sequence.txt $P/${TAG}_pretrain_step_sequence.txt
cp $O/pretrain_kernel_trace_summary.md $P/${TAG}_pretrain_kernel_trace_summary.md
{ echo "# tools/tox21_bench.py"; grep -v amdgpu.ids $O/tox21.txt; echo; echo "# tools/pretrain_bench.py"; grep -v amdgpu.ids $O/pretrain.txt; echo
  echo "# __graft_entry__.smoke()"; tail -1 $O/smoke.txt; echo; echo "# pytest tests -m gpu"; tail -1 $O/pytest_gpu.txt; } > $P/${TAG}_other_configs.txt
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
from fragnet_amd import build
d = build.source_digest()
for f in ("in_graph_kernels.json", "pmc_per_launch.json", "${TAG}_pmc_step.json"):
    got = json.load(open("$P/" + f)).get("source_digest")
    print(f, "digest ok" if got == d else f"DIGEST MISMATCH ({str(got)[:12]} vs the tree's {d[:12]})")
PY
