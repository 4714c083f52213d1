export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/calib -- python3 $R/tools/calibrate_instr.py run $R/$O/calib_configs.json > $R/$O/calib_run.log 2>&1
cd $R
python tools/calibrate_instr.py fit $O/calib_configs.json $O/calib profiles/r02_instr_model.json > $O/calib_fit.log 2>&1; tail -15 $O/calib_fit.log
cp profiles/r02_instr_model.json $O/
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; tail -2 $O/bench_20.err; cut -c1-300 $O/bench_20.json
python bench.py --workload biprime --streams 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_biprime_s1.json 2> /dev/null; cut -c1-300 $O/bench_biprime_s1.json
python bench.py --workload biprime --streams 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_biprime_s2.json 2> /dev/null; cut -c1-300 $O/bench_biprime_s2.json
python bench.py --workload biprime --streams 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_biprime_s4.json 2> /dev/null; cut -c1-300 $O/bench_biprime_s4.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_biprime -- python3 $R/bench.py --workload biprime --streams 2 --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/trace_biprime_summary.txt $O/trace_biprime > /dev/null; head -14 $O/trace_biprime_summary.txt
