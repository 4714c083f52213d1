export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
( time python -m pytest tests -m gpu -x -q -k "segments or two_phase or biprime or keygen_flow or jacobi or nsquare or c3 or c5 or instance" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for seg in 1 2 4 8; do
  for st in 20 48; do
    python bench.py --no-cpu-baseline --no-extras --steps $st --warmup 5 --segments $seg > $O/bench_seg${seg}_s${st}.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$O/bench_seg${seg}_s${st}.json")); print("segments $seg steps $st:", round(d["value"]), round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), round(d["roofline"]["kernel_ms"],1))
PY
  done
done
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --segments 4 --streams 5 > $O/bench_seg4_s20_st5.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_seg4_s20_st5.json'));print('seg4 20 steps 5 streams',round(d['value']))"
python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --streams 1 --segments 1 > $O/bench_single_seg1.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_single_seg1.json'));print('single seg1',round(d['value']))"
python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --streams 1 --segments 4 > $O/bench_single_seg4.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_single_seg4.json'));print('single seg4',round(d['value']))"
python bench.py --workload biprime --streams 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_biprime.json 2> /dev/null; python -c "import json;d=json.load(open('$O/bench_biprime.json'));print('biprime',round(d['value']),d['ms_per_step'],d['stages'])"
