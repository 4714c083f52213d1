# round 3, call J: A/B of library builds on one box (hand-over by counters vs workgroup barrier in the wide split kernel)
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/ab.txt
for v in "" tools/variants/lib_v1_barrier_wide.so; do
  echo "=== library: ${v:-current}" >> $O/ab.txt
  MX_LIBRARY=$v timeout 600 python tools/sweep_shapes.py 4096 2>&1 | grep -E "batch|^ +(1|64|2048|4096|8192) " >> $O/ab.txt
  MX_LIBRARY=$v timeout 600 python tools/sweep_shapes.py 2048 2>&1 | grep -E "^ +(1|64|8192|10000|16384) " >> $O/ab.txt
done
cat $O/ab.txt
