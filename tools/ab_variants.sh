# usage: bash tools/ab_variants.sh [bench args]  — A/B of kernel builds in build/variants (developer tool)
V=protocols/distributed_keygen_amd/build/variants
for round in 1 2; do
for lib in default $(ls $V/*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset MX_LIBRARY; else export MX_LIBRARY=$PWD/$lib; fi
  printf "%s " "$lib"
  python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
done
