# round 3, call U: profile round with the aligned build
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
cat gpurun_out/prof_r03/sweep_shapes.txt
