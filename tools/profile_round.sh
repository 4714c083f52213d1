# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Runs the bench lines and rocprofv3 passes whose summaries are copied into profiles/ (tools/prof_summary.py).
tag=${1:-r01}
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/prof_$tag; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline > $O/bench_default_rep2.json 2>/dev/null
python bench.py --no-cpu-baseline --streams 3 --limbs-per-lane 9 > $O/bench_narrow_3inflight.json 2>/dev/null
python bench.py --no-cpu-baseline --streams 1 --limbs-per-lane 9 --steps 12 --warmup 4 > $O/bench_single_narrow.json 2>/dev/null
python bench.py --no-cpu-baseline --streams 1 --limbs-per-lane 18 --steps 12 --warmup 4 > $O/bench_single_wide.json 2>/dev/null
python bench.py --no-cpu-baseline --batch 40000 --steps 12 --warmup 3 --streams 3 > $O/bench_40k.json 2>/dev/null
B="python3 bench.py --no-cpu-baseline"
S="python3 bench.py --no-cpu-baseline --streams 1 --limbs-per-lane 18 --steps 12 --warmup 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- $B > $O/trace_default_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_single_wide -- $S > $O/trace_single_wide_bench.json 2>/dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_single_wide_sq -- $S > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_single_wide_fetch -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_single_wide_write -- $S > /dev/null 2>&1
find $O -name "*.csv" | wc -l
