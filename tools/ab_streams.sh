# usage: bash tools/ab_streams.sh — headline workload, pair-kernel geometry x streams in flight (steps a multiple of streams)
for rep in 1 2; do
for cfg in "9 3 12" "18 3 12" "9 4 12" "18 4 12" "18 5 15" "9 6 12" "18 6 12"; do
  set -- $cfg
  printf "L=%s streams=%s steps=%s " $1 $2 $3
  python bench.py --no-cpu-baseline --limbs-per-lane $1 --wavefronts-per-group 1 --streams $2 --steps $3 --warmup $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
done
