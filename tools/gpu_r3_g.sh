# round 3, call G: counter-based hand-over parity + sweep, bench line with leg timings
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_instances.py tests/test_gpu_powmod.py tests/test_gpu_ops.py -x -q ) > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 900 python tools/sweep_shapes.py 2048 > $O/sweep_shapes.txt 2>&1; tail -18 $O/sweep_shapes.txt
( time python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -5 $O/bench_driver_flags.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03g/bench_driver_flags.json'))
r=d['roofline']
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'geo',d['config']['geometry_K_L_W_blocks'],d['config']['wavefronts_per_group'],'frac',r.get('frac'),'frac@clk',r.get('frac_at_measured_clock'),'clk',r.get('shader_clock_mhz_measured'),'kernel_ms',r.get('kernel_ms'))
sb=d.get('single_batch'); print('single_batch',sb.get('value'),sb.get('ms_per_step'),sb.get('geometry_K_L_W_blocks'),sb.get('shader_clock_mhz_measured'))
print('latency',{k:(v.get('ms') if isinstance(v,dict) else v) for k,v in d.get('latency',{}).items() if k!='note' and k!='unit'})
e=d.get('end_to_end',{}); print('e2e', {k:(round(v['partial_decrypt_rate']), round(v['partial_decrypt_vs_tensor_level'],2)) for k,v in e.items() if isinstance(v,dict) and 'partial_decrypt_rate' in v} if 'error' not in e else e)
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
kg=d.get('end_to_end_keygen',{}); print('keygen', {k:{a:round(b,4) if isinstance(b,float) else b for a,b in v.items()} for k,v in kg.get('rounds',{}).items()}, kg.get('error'))
print('leg_seconds', d.get('leg_seconds'))
PY
