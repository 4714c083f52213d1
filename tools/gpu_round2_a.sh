# first GPU pass of round 2: tests, smoke, bench (driver flags and defaults), kernel trace
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cat $O/bench_20.json | cut -c1-400
python bench.py --no-cpu-baseline > $O/bench_48.json 2> $O/bench_48.err; cat $O/bench_48.json | cut -c1-300
python bench.py --no-cpu-baseline --streams 1 --steps 12 --warmup 4 > $O/bench_single.json 2>/dev/null; cat $O/bench_single.json | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_20 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/trace_20_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/trace_20_summary.txt $O/trace_20 > /dev/null; head -30 $O/trace_20_summary.txt
