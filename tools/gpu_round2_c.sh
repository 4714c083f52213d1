# calibration re-run + PMC passes of the bench legs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/calib -- python3 $R/tools/calibrate_instr.py run $R/$O/calib_configs.json > $R/$O/calib_run.log 2>&1
cd $R
python tools/calibrate_instr.py fit $O/calib_configs.json $O/calib profiles/r02_instr_model.json > $O/calib_fit.log 2>&1; tail -15 $O/calib_fit.log
cp profiles/r02_instr_model.json $O/
# validation of the model on the headline launch + measured HBM traffic (single stream so that counters are per launch)
S="python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --steps 6 --warmup 2"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_c3_wide_sq -- $S > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c3_wide_fetch -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c3_wide_write -- $S > /dev/null 2>&1
B="python3 $R/bench.py --workload biprime --no-cpu-baseline --streams 1 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_write -- $B > /dev/null 2>&1
C="python3 $R/bench.py --workload c5 --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_c5_sq -- $C > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c5_fetch -- $C > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c5_write -- $C > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py n2_k2048_b10000_L18 powmod_n2_kernel $O/pmc_c3_wide_fetch $O/pmc_c3_wide_write 6
python tools/hbm_traffic.py biprime_b2053_c4096_L18 "mx::powmod_kernel" $O/pmc_biprime_fetch $O/pmc_biprime_write 3
python tools/hbm_traffic.py n2_k4096_b4096_L18 powmod_n2_kernel $O/pmc_c5_fetch $O/pmc_c5_write 3
cp profiles/r02_hbm_traffic.json $O/
python tools/prof_summary.py $O/pmc_c3_wide_summary.txt $O/pmc_c3_wide_sq $O/pmc_c3_wide_sq > /dev/null
python tools/prof_summary.py $O/pmc_biprime_summary.txt $O/pmc_biprime_sq $O/pmc_biprime_sq > /dev/null
python tools/prof_summary.py $O/pmc_c5_summary.txt $O/pmc_c5_sq $O/pmc_c5_sq > /dev/null
grep -h "SQ_INSTS_VALU\|SQ_WAVES " $O/pmc_*_summary.txt
# the full default line with the model in place, and its kernel trace
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cut -c1-300 $O/bench_20.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_20 -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $R/$O/trace_20_bench.json 2>/dev/null
cd $R
python tools/prof_summary.py $O/trace_20_summary.txt $O/trace_20 > /dev/null; head -12 $O/trace_20_summary.txt
