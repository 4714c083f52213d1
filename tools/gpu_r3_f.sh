# round 3, call F: instruction-model calibration (SQ_INSTS_VALU), first bench line with the new legs, full GPU suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
cd /tmp
( time rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/pmc_cal -- python3 $R/tools/calibrate_instr.py run $R/$O/cal_configs.json ) > $R/$O/cal_run.log 2>&1
cd $R
tail -3 $O/cal_run.log
python tools/calibrate_instr.py fit $O/cal_configs.json $O/pmc_cal profiles/r03_instr_model.json > $O/cal_fit.log 2>&1; tail -12 $O/cal_fit.log
cp profiles/r03_instr_model.json $O/
f=$(find $O/pmc_cal -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $O/cal_counter_collection.csv.gz
rm -rf $O/pmc_cal
( time python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -5 $O/bench_driver_flags.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03f/bench_driver_flags.json'))
r=d['roofline']
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'frac',r.get('frac'),'frac@clk',r.get('frac_at_measured_clock'),'clk',r.get('shader_clock_mhz_measured'),'guide',r.get('frac_vs_guide_vector_peak'))
print('single_batch',d.get('single_batch'))
print('latency',json.dumps(d.get('latency'))[:900])
e=d.get('end_to_end',{}); print('e2e', {k:(round(v['partial_decrypt_rate']), round(v['partial_decrypt_vs_tensor_level'],2)) for k,v in e.items() if isinstance(v,dict) and 'partial_decrypt_rate' in v} if 'error' not in e else e)
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
print('keygen', json.dumps(d.get('end_to_end_keygen'))[:2500])
PY
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
