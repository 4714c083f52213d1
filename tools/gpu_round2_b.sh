# second GPU pass of round 2: box facts, instruction-model calibration, bench with all legs, PMC passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
( nproc; python3 -c "import os;print('affinity',len(os.sched_getaffinity(0)),'cpu_count',os.cpu_count())"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; rocm-smi --showclocks 2>/dev/null | head -20 ) > $O/box.txt 2>&1
python -m pytest tests -m gpu -x -q -k "instances or nsquare or one_engine" > $O/pytest_quick.log 2>&1; tail -3 $O/pytest_quick.log
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/calib -- python3 $R/tools/calibrate_instr.py run $R/$O/calib_configs.json > $R/$O/calib_run.log 2>&1
cd $R
python tools/calibrate_instr.py fit $O/calib_configs.json $O/calib profiles/r02_instr_model.json > $O/calib_fit.log 2>&1; tail -15 $O/calib_fit.log
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; tail -3 $O/bench_20.err; cut -c1-600 $O/bench_20.json
python bench.py --workload biprime --steps 8 --warmup 2 > $O/bench_biprime.json 2> $O/bench_biprime.err; tail -3 $O/bench_biprime.err; cut -c1-400 $O/bench_biprime.json
python bench.py --workload c5 --steps 8 --warmup 2 --no-extras > $O/bench_c5.json 2> $O/bench_c5.err; tail -3 $O/bench_c5.err; cut -c1-300 $O/bench_c5.json
MX_BENCH_FORCE_DIST=1 python bench.py --workload biprime --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_biprime_rccl1.json 2> $O/bench_biprime_rccl1.err; tail -2 $O/bench_biprime_rccl1.err; cut -c1-200 $O/bench_biprime_rccl1.json
