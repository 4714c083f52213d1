# usage: bash tools/ab_geometry.sh [n] [streams] — headline workload, narrow vs wide pair kernel, n alternating runs each
for i in $(seq 1 ${1:-5}); do
  for l in 9 18; do
    printf "L=%s " $l
    python bench.py --no-cpu-baseline --limbs-per-lane $l --wavefronts-per-group 1 --streams ${2:-3} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
  done
done
