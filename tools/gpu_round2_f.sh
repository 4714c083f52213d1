export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
( time python -m pytest tests -m gpu -x -q -k "jacobi or biprime or keygen_flow or two_phase or c2_c4 or golden_biprime or smoke or sieve or verdict" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python tools/jacobi_timing.py > $O/jacobi_timing.txt 2>&1; tail -12 $O/jacobi_timing.txt
python bench.py --workload biprime --no-cpu-baseline > $O/bench_biprime.json 2> /dev/null; python -c "import json;d=json.load(open('$O/bench_biprime.json'));print('biprime',round(d['value']),d['ms_per_step'],d['stages'])"
python bench.py --workload biprime --no-cpu-baseline --key-length 1024 --batch 8192 > $O/bench_biprime_k1024.json 2> /dev/null; python -c "import json;d=json.load(open('$O/bench_biprime_k1024.json'));print('biprime1024',round(d['value']),d['ms_per_step'],d['stages'])"
