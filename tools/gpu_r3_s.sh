# round 3, call S: the alignment pass with e32->e64 re-encoding (dynamic programme) — parity suite, A/B against no pass
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
V=protocols/distributed_keygen_amd/build/variants
for round in 1 2; do
for lib in default $(ls $V/*.so); do
  if [ "$lib" = default ]; then unset MX_LIBRARY; else export MX_LIBRARY=$PWD/$lib; fi
  timeout 300 python tools/variant_probe.py 2>/dev/null | tail -1
done
done > $O/variants.txt
cat $O/variants.txt
unset MX_LIBRARY
for lib in default $V/noalign.so; do
  if [ "$lib" = default ]; then unset MX_LIBRARY; else export MX_LIBRARY=$PWD/$lib; fi
  for w in "" "--workload biprime" "--workload c5 --no-extras"; do
  printf "%s [%s] " "$lib" "$w"
  python bench.py --no-cpu-baseline --no-extras $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), (d.get('latency') or {}).get('value'))"
  done
done > $O/bench_ab.txt 2>&1
cat $O/bench_ab.txt
