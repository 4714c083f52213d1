# usage: bash tools/sweep_keys.sh — pair kernel, narrow vs wide geometry over key sizes
for cfg in "1024 9 20000" "1024 18 20000" "3072 9 4000" "3072 18 4000" "4096 9 4000" "4096 18 4000" "4096 9 16000" "4096 18 16000" "2048 9 2000" "2048 18 2000"; do
  set -- $cfg
  echo "key=$1 L=$2 batch=$3"
  python bench.py --no-cpu-baseline --key-length $1 --limbs-per-lane $2 --streams 3 --batch $3 --steps 6 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
