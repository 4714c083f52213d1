# usage: bash tools/ab_steps.sh — sustained rate of the default workload: steps in flight at 60 timed steps
for rep in 1 2; do
for cfg in "60 4 18" "60 5 18" "60 6 18" "60 3 9" "60 6 9"; do
  set -- $cfg
  printf "steps=%s streams=%s L=%s " $1 $2 $3
  python bench.py --no-cpu-baseline --steps $1 --warmup $2 --streams $2 --limbs-per-lane $3 --wavefronts-per-group 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],1))"
done
done
