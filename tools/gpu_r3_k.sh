# round 3, call K: double-buffered lanes (recombination beside the next launch): bench lines
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b1.json 2> $O/b1.err; tail -3 $O/b1.err
python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extras > $O/b2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streams 1 > $O/b3.json 2>/dev/null
MX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/b4.json 2> $O/b4.err; tail -3 $O/b4.err
python bench.py --workload c5 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/b5.json 2>/dev/null
python - <<'PY'
import json
for f in ('b1','b2','b3','b4','b5'):
    try:
        d=json.load(open(f'gpurun_out/r03k/{f}.json')); r=d['roofline']
        print(f, round(d['value']), round(d['ms_per_step'],2), 'kernel_ms', round(r['kernel_ms'],2), 'frac', round(r['frac'],3), round(r['frac_at_measured_clock'],3), d['config']['verified'][:40])
    except Exception as e: print(f, 'ERR', e)
PY
