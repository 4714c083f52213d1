# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Runs the bench lines and rocprofv3 passes whose summaries are copied into profiles/ (tools/prof_summary.py,
# tools/hbm_traffic.py, tools/calibrate_instr.py).
tag=${1:-r02}
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT
O=gpurun_out/prof_$tag; mkdir -p $O
# ---- the bench lines (driver flags, defaults, variants)
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --no-extras --segments 1 > $O/bench_unsegmented.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --streams 3 --limbs-per-lane 9 > $O/bench_narrow_3inflight.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --streams 1 --steps 12 --warmup 4 > $O/bench_single_stream.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --batch 40000 --steps 12 --warmup 3 --streams 3 > $O/bench_40k.json 2>/dev/null
python bench.py --workload biprime > $O/bench_biprime.json 2>/dev/null
python bench.py --workload biprime --key-length 1024 --batch 8192 --no-cpu-baseline > $O/bench_biprime_k1024.json 2>/dev/null
python bench.py --workload c5 --no-extras > $O/bench_c5_b4096.json 2>/dev/null
python bench.py --workload c5 --no-extras --no-cpu-baseline --batch 1024 --steps 12 --warmup 4 > $O/bench_c5_b1024.json 2>/dev/null
python bench.py --workload c5 --no-extras --no-cpu-baseline --batch 16384 --steps 8 --warmup 4 > $O/bench_c5_b16384.json 2>/dev/null
MX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_rccl_single_rank.json 2> $O/bench_rccl_single_rank.err
MX_BENCH_FORCE_DIST=1 python bench.py --workload biprime --no-cpu-baseline > $O/bench_biprime_rccl_single_rank.json 2>/dev/null
# ---- kernel traces of the same commands
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_driver_flags -- $B > $R/$O/trace_driver_flags_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_biprime -- python3 $R/bench.py --workload biprime --no-cpu-baseline > $R/$O/trace_biprime_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -- python3 $R/bench.py --workload c5 --no-extras --no-cpu-baseline > $R/$O/trace_c5_bench.json 2>/dev/null
# ---- counters (single stream so that a dispatch's counters are its own)
S="python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --segments 1 --steps 6 --warmup 2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_c3_sq -- $S > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c3_fetch -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c3_write -- $S > /dev/null 2>&1
BP="python3 $R/bench.py --workload biprime --no-cpu-baseline --streams 1 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_sq -- $BP > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_fetch -- $BP > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_write -- $BP > /dev/null 2>&1
C5="python3 $R/bench.py --workload c5 --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --segments 1 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_c5_sq -- $C5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c5_fetch -- $C5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c5_write -- $C5 > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py n2_k2048_b10000_L18 powmod_n2_kernel $O/pmc_c3_fetch $O/pmc_c3_write 6 > /dev/null
python tools/hbm_traffic.py biprime_b2053_c4096_L18 "mx::powmod_kernel" $O/pmc_biprime_fetch $O/pmc_biprime_write 3 > /dev/null
python tools/hbm_traffic.py n2_k4096_b4096_L18 powmod_n2_kernel $O/pmc_c5_fetch $O/pmc_c5_write 3 > /dev/null
cp profiles/r02_hbm_traffic.json $O/
python tools/prof_summary.py $O/summary_driver_flags.txt $O/trace_driver_flags > /dev/null
python tools/prof_summary.py $O/summary_biprime.txt $O/trace_biprime $O/pmc_biprime_sq $O/pmc_biprime_fetch $O/pmc_biprime_write > /dev/null
python tools/prof_summary.py $O/summary_c5.txt $O/trace_c5 $O/pmc_c5_sq $O/pmc_c5_fetch $O/pmc_c5_write > /dev/null
python tools/prof_summary.py $O/summary_c3_single_stream_counters.txt $O/pmc_c3_sq $O/pmc_c3_sq $O/pmc_c3_fetch $O/pmc_c3_write > /dev/null
for f in trace_driver_flags trace_biprime trace_c5; do cp $(find $O/$f -name "*_kernel_stats.csv" | head -1) $O/${f}_kernel_stats.csv; done
rm -rf $O/trace_*/ $O/pmc_*/
ls $O | wc -l
