# round 3, call H: full GPU suite, instruction-model calibration of the final kernels, biprime rate vs candidates per GPU, bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
cd /tmp
( time rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/pmc_cal -- python3 $R/tools/calibrate_instr.py run $R/$O/cal_configs.json ) > $R/$O/cal_run.log 2>&1
cd $R
python tools/calibrate_instr.py fit $O/cal_configs.json $O/pmc_cal profiles/r03_instr_model.json > $O/cal_fit.log 2>&1; head -3 $O/cal_fit.log
cp profiles/r03_instr_model.json $O/
f=$(find $O/pmc_cal -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $O/cal_counter_collection.csv.gz
rm -rf $O/pmc_cal
for c in 512 1024 2048 4096; do python bench.py --workload biprime --batch $c --steps 8 --warmup 2 --no-cpu-baseline > $O/biprime_c$c.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/biprime_c$c.json')); print($c, round(d['value']), round(d['ms_per_step'],2), d['config']['geometry_K_L_W_blocks'], d['roofline'].get('frac'))"; done
( time python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03h/bench_driver_flags.json'))
r=d['roofline']
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'frac',r.get('frac'),'frac@clk',r.get('frac_at_measured_clock'),'clk',r.get('shader_clock_mhz_measured'),'mac',r.get('frac_macs_vs_multiply_issue_peak'),'guide',r.get('frac_vs_guide_vector_peak'))
sb=d.get('single_batch'); print('single_batch',sb.get('value'),sb.get('ms_per_step'))
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('frac_at_measured_clock'), v.get('error'))
PY
