#!/bin/bash
# dev A/B on a GPU box: bash tools/probe/ab_r05.sh <tag> "<extra hipcc flags>" [bench.py args...]  -> gpurun_out/seq_<tag>.txt + one bench line
tag=$1; export FRAGNET_EXTRA_HIPCC_FLAGS="$2"; shift; shift
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/step_trace.sh gpurun_out/seq_$tag.txt --steps 20 --warmup 5 --epoch-batches 0 --no-round3-shapes "$@"
echo "== $tag [$FRAGNET_EXTRA_HIPCC_FLAGS] $@"; grep -E "GPU busy" gpurun_out/seq_$tag.txt | cut -c1-78
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', d['ms_per_step'], d.get('ms_per_step_repeats'))"
