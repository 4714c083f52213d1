# round 3, call S: A/B of the alignment pass's run length (incl. the time-sliced instances)
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
V=protocols/distributed_keygen_amd/build/variants
for round in 1 2; do
for lib in default $(ls $V/*.so); do
  if [ "$lib" = default ]; then unset MX_LIBRARY; else export MX_LIBRARY=$PWD/$lib; fi
  timeout 300 python tools/variant_probe.py 2>/dev/null | tail -1
done
done > $O/variants.txt
cat $O/variants.txt
