# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Runs the bench lines and rocprofv3 passes whose summaries are copied into profiles/ (tools/prof_summary.py,
# tools/hbm_traffic.py, tools/calibrate_instr.py, tools/sweep_shapes.py).  The library reads no environment; the
# bench opts into 16 HIP hardware queues itself (protocols.distributed_keygen_amd.configure_hw_queues).
tag=${1:-r06}
export PROFILE_TAG=$tag
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/prof_$tag; mkdir -p $O
# ---- instruction model of the kernels as built (SQ_INSTS_VALU), first: the bench lines below read it
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/pmc_cal -- python3 $R/tools/calibrate_instr.py run $R/$O/cal_configs.json > $R/$O/cal_run.log 2>&1
cd $R
python tools/calibrate_instr.py fit $O/cal_configs.json $O/pmc_cal profiles/${tag}_instr_model.json > $O/cal_fit.log 2>&1
cp profiles/${tag}_instr_model.json $O/
f=$(find $O/pmc_cal -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $O/instr_model_raw_counters.csv.gz
rm -rf $O/pmc_cal
# ---- the bench lines (driver flags, defaults, variants)
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --no-extras --streams 1 --steps 12 --warmup 4 > $O/bench_single_stream.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --batch 40000 --steps 12 --warmup 3 --streams 3 > $O/bench_40k.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --batch 8192 --streams 1 --steps 12 --warmup 4 > $O/bench_single_stream_8192.json 2>/dev/null
python bench.py --workload biprime > $O/bench_biprime.json 2>/dev/null
python bench.py --workload biprime --key-length 1024 --batch 8192 --no-cpu-baseline > $O/bench_biprime_k1024.json 2>/dev/null
for c in 512 1024 2048; do python bench.py --workload biprime --batch $c --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_biprime_c$c.json 2>/dev/null; done
python bench.py --workload c5 --no-extras > $O/bench_c5_b4096.json 2>/dev/null
MX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_rccl_single_rank.json 2> $O/bench_rccl_single_rank.err
MX_BENCH_FORCE_DIST=1 python bench.py --workload biprime --no-cpu-baseline > $O/bench_biprime_rccl_single_rank.json 2>/dev/null
MX_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.err
python tools/sweep_shapes.py 2048 4096 1024 > $O/sweep_shapes.txt 2>&1
python tools/sweep_generic.py 1024 2048 > $O/sweep_generic.txt 2>&1
python tools/bi_pivot_sweep.py 1024 2048 4096 > $O/bi_pivot_sweep.txt 2>&1
python tools/latency_probe.py > $O/small_batch_latency.txt 2>&1
TS_SEGS=2,8,12 TS_SIZES=8192,8448,8704,9216,10000,10240,10752,11264,11776,12288,13312,16384 python tools/ts_probe.py 2048 > $O/ts_probe_2048.txt 2>&1
TS_SEGS=2,8,12 TS_SIZES=4096,4352,4608,5000,5120,5632,6144,7000 python tools/ts_probe.py 4096 > $O/ts_probe_4096.txt 2>&1
python tools/lone_call_probe.py 10000 0 3 30 > $O/lone_call_probe.txt 2>&1
python tools/keygen_round_profile.py 65536 > $O/keygen_round_host_profile.txt 2>&1
# ---- kernel traces of the same commands
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_driver_flags -- $B > $R/$O/trace_driver_flags_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_single_batch -- python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --steps 8 --warmup 2 > $R/$O/trace_single_batch_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_biprime -- python3 $R/bench.py --workload biprime --no-cpu-baseline > $R/$O/trace_biprime_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -- python3 $R/bench.py --workload c5 --no-extras --no-cpu-baseline > $R/$O/trace_c5_bench.json 2>/dev/null
# ---- counters (single stream so that a dispatch's counters are its own)
S="python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --wavefronts-per-group 1 --segments 1 --steps 6 --warmup 2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_c3_sq -- $S > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c3_fetch -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c3_write -- $S > /dev/null 2>&1
# ---- where the issue slots of a SATURATED launch go (32 768 ciphertexts = two wavefronts per SIMD of the headline kernel):
# instruction-issue and wait counters in two passes
SAT="python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --wavefronts-per-group 1 --segments 1 --batch 32768 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$O/pmc_sat_a -- $SAT > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/pmc_sat_b -- $SAT > /dev/null 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $R/$O/pmc_sat_c -- $SAT > /dev/null 2>&1
S2="python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --segments 1 --steps 6 --warmup 2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_split_sq -- $S2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_split_fetch -- $S2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_split_write -- $S2 > /dev/null 2>&1
BP="python3 $R/bench.py --workload biprime --no-cpu-baseline --streams 1 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_sq -- $BP > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_fetch -- $BP > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_biprime_write -- $BP > /dev/null 2>&1
C5="python3 $R/bench.py --workload c5 --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --wavefronts-per-group 1 --segments 1 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_c5_sq -- $C5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c5_fetch -- $C5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_c5_write -- $C5 > /dev/null 2>&1
# ---- the short kernels: instruction counts (one counter pass) and durations (one trace pass) of bench.short_kernel_cases
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/$O/sk_pmc -- python3 $R/tools/short_kernels.py run > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/sk_trace -- python3 $R/tools/short_kernels.py run > /dev/null 2>&1
cd $R
python tools/short_kernels.py fit $O/sk_pmc $O/sk_trace profiles/${tag}_short_kernels.json > $O/short_kernels.txt 2>&1
cp profiles/${tag}_short_kernels.json $O/
python tools/hbm_traffic.py n2_k2048_b10000_L18 "powmod_n2_kernel" $O/pmc_c3_fetch $O/pmc_c3_write 6 2 > /dev/null
python tools/hbm_traffic.py n2_k2048_b10000_L18x2 "powmod_n2_split_kernel" $O/pmc_split_fetch $O/pmc_split_write 6 > /dev/null
python tools/hbm_traffic.py biprime_b2053_c4096_L18 "mx::powmod_kernel" $O/pmc_biprime_fetch $O/pmc_biprime_write 3 > /dev/null
python tools/hbm_traffic.py n2_k4096_b4096_L18 "powmod_n2_kernel" $O/pmc_c5_fetch $O/pmc_c5_write 3 2 > /dev/null
cp profiles/${tag}_hbm_traffic.json $O/
python tools/prof_summary.py $O/summary_driver_flags.txt $O/trace_driver_flags > /dev/null
python tools/prof_summary.py $O/summary_single_batch.txt $O/trace_single_batch $O/pmc_split_sq $O/pmc_split_fetch $O/pmc_split_write > /dev/null
python tools/prof_summary.py $O/summary_biprime.txt $O/trace_biprime $O/pmc_biprime_sq $O/pmc_biprime_fetch $O/pmc_biprime_write > /dev/null
python tools/prof_summary.py $O/summary_c5.txt $O/trace_c5 $O/pmc_c5_sq $O/pmc_c5_fetch $O/pmc_c5_write > /dev/null
python tools/prof_summary.py $O/summary_c3_single_stream_counters.txt $O/pmc_c3_sq $O/pmc_c3_sq $O/pmc_c3_fetch $O/pmc_c3_write > /dev/null
python tools/prof_summary.py $O/summary_c3_saturated_issue_counters.txt $O/pmc_sat_a $O/pmc_sat_a $O/pmc_sat_b $O/pmc_sat_c > /dev/null
for f in trace_driver_flags trace_single_batch trace_biprime trace_c5; do cp $(find $O/$f -name "*_kernel_stats.csv" | head -1) $O/${f}_kernel_stats.csv; done
for f in pmc_c3_sq pmc_split_sq pmc_sat_a pmc_sat_b pmc_sat_c; do g=$(find $O/$f -name "*counter_collection.csv" | head -1); [ -n "$g" ] && gzip -c $g > $O/${f}_raw_counters.csv.gz; done
rm -rf $O/trace_*/ $O/pmc_*/ $O/sk_pmc $O/sk_trace
ls $O | wc -l
