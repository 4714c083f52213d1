# round 3, call M: CU-slice streams for small concurrent launches (C5 at 1024 x 4 in flight), with / without
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
for cs in 0 1; do
python bench.py --workload c5 --batch 1024 --steps 8 --warmup 4 --streams 4 --cu-slices $cs --no-cpu-baseline --no-extras > $O/c5_b1024_cs$cs.json 2> $O/c5_cs$cs.err
python bench.py --batch 2000 --steps 12 --warmup 4 --streams 4 --cu-slices $cs --no-cpu-baseline --no-extras > $O/c3_b2000_cs$cs.json 2>/dev/null
python bench.py --batch 1024 --steps 12 --warmup 4 --streams 4 --cu-slices $cs --no-cpu-baseline --no-extras > $O/c3_b1024_cs$cs.json 2>/dev/null
done
python bench.py --steps 20 --warmup 5 --cu-slices 1 --no-cpu-baseline --no-extras > $O/c3_main_cs1.json 2>/dev/null
tail -3 $O/c5_cs1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03m/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), d['config']['geometry_K_L_W_blocks'], d['config']['wavefronts_per_group'], d['config']['cu_slices'], 'kernel_ms', round(r['kernel_ms'],1))
    except Exception as e: print(f, 'ERR', e)
PY
