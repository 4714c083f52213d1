# round 3, call W: last check — smoke, full GPU suite, the driver's bench command
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-200
( time python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03w/bench_driver_flags.json'))
r=d['roofline']
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'frac',r.get('frac'),'frac@clk',r.get('frac_at_measured_clock'),'clk',r.get('shader_clock_mhz_measured'),'traffic',r.get('traffic'))
sb=d.get('single_batch'); print('single_batch',sb.get('value'),sb.get('ms_per_step'))
print('latency',{k:(round(v['ms'],2) if isinstance(v,dict) else v) for k,v in d.get('latency',{}).items() if k not in ('note','unit')})
e=d.get('end_to_end',{}); print('e2e', {k:(round(v['partial_decrypt_rate']), round(v['partial_decrypt_vs_tensor_level'],2)) for k,v in e.items() if isinstance(v,dict) and 'partial_decrypt_rate' in v} if 'error' not in e else e)
for k,v in d.get('extra',{}).items(): print(k, v.get('value') and round(v['value']), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('config',{}).get('cu_slices'), v.get('error'))
kg=d.get('end_to_end_keygen',{}); print('keygen', {k:round(v['candidates_per_s']) for k,v in kg.get('rounds',{}).items()}, kg.get('error'))
print('leg_seconds', d.get('leg_seconds'))
PY
