# usage: bash tools/sweep_geometry.sh  — headline workload over limbs-per-lane / streams / batch
for cfg in "9 3 10000" "18 3 10000" "18 4 10000" "18 6 10000" "18 3 40000" "9 3 40000"; do
  set -- $cfg
  echo "L=$1 streams=$2 batch=$3"
  python bench.py --no-cpu-baseline --limbs-per-lane $1 --wavefronts-per-group 1 --streams $2 --batch $3 --steps 9 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
