# round 3, call S: A/B of library variants on the fixed launch points
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
V=protocols/distributed_keygen_amd/build/variants
for round in 1 2; do
for lib in $(ls $V/*.so); do
  export MX_LIBRARY=$PWD/$lib
  timeout 300 python tools/ts_r_probe.py 2>/dev/null | tail -1
done
done > $O/variants.txt
cat $O/variants.txt
