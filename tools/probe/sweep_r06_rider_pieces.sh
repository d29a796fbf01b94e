cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for v in 1 2 4 8; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --epoch-batches 0 --no-round3-shapes --tune 27=$v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('27=$v ms_per_step', d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'])"
done; done
