# round 3, call A: VALU issue-peak microbenchmark (wall clock + counters) and the GPU test suite as it stands
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
tools/ubench/valu_peak > $O/valu_peak.txt 2>&1; tail -3 $O/valu_peak.txt
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_a -- $R/tools/ubench/valu_peak > $R/$O/valu_peak_pmc_a.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/$O/pmc_b -- $R/tools/ubench/valu_peak > $R/$O/valu_peak_pmc_b.txt 2>&1
cd $R
find $O/pmc_a $O/pmc_b -name "*.csv" | head; 
for d in pmc_a pmc_b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_counters.csv; f=$(find $O/$d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_trace.csv; done
rm -rf $O/pmc_a $O/pmc_b
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
