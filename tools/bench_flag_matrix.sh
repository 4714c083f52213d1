# usage: bash tools/bench_flag_matrix.sh — bench.py under the flag combinations a driver may pass
for cfg in "--steps 5 --warmup 2" "--steps 1 --warmup 0" "--steps 7 --warmup 1" "--steps 10 --warmup 3" "--steps 20 --warmup 5" "--steps 3 --warmup 1" "--steps 11 --warmup 2"; do
  printf "%s -> " "$cfg"
  python bench.py --gpus 1 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), 'in flight', d['config']['steps_in_flight'], d['roofline']['kernel'])"
done
