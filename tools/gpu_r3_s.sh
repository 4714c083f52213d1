export TMPDIR=/tmp
V=protocols/distributed_keygen_amd/build/variants
for round in 1 2; do
for lib in default $(ls $V/*.so); do
  if [ "$lib" = default ]; then unset MX_LIBRARY; else export MX_LIBRARY=$PWD/$lib; fi
  timeout 300 python tools/variant_probe.py 2>/dev/null | tail -1 | python3 -c "import sys; p=sys.stdin.read().split(); print(p[0], [x for x in ' '.join(p).split('|') if 'L3' in x])"
done
done
