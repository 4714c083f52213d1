# usage: bash tools/ab_queues.sh — wide pair kernel with 8 HIP hardware queues: streams in flight, repeated
export GPU_MAX_HW_QUEUES=8
for rep in 1 2 3; do
for cfg in "18 3 12" "18 4 12" "18 5 15" "18 6 12" "18 8 16" "9 6 12"; do
  set -- $cfg
  printf "L=%s streams=%s steps=%s " $1 $2 $3
  python bench.py --no-cpu-baseline --limbs-per-lane $1 --wavefronts-per-group 1 --streams $2 --steps $3 --warmup $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
done
