# usage: bash tools/sweep_keys.sh — headline kernel over key sizes and batch sizes, default bench settings
for cfg in "1024 20000" "3072 4000" "4096 1000" "4096 4000" "4096 16000" "2048 2000" "2048 40000"; do
  set -- $cfg
  printf "key=%s batch=%s " $1 $2
  python bench.py --no-cpu-baseline --key-length $1 --batch $2 --steps 8 --warmup 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], d['config']['steps_in_flight'])"
done
python bench.py --no-cpu-baseline --generic-modulus 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], d['config']['steps_in_flight'])"
