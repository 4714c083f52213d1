# usage (in the build container, after `gpurun -- bash tools/gpu_session.sh <tag> profile` merged gpurun_out/prof_<tag>):
#   bash tools/collect_profiles.sh <tag>
# Copies the outputs of tools/profile_round.sh into profiles/ under the names profiles/README.md lists.
tag=${1:-r06}
O=gpurun_out/prof_$tag; P=profiles
[ -d $O ] || { echo "no $O"; exit 1; }
for f in $O/bench_*.json; do cp $f $P/${tag}_$(basename $f); done
cp $O/${tag}_instr_model.json $O/${tag}_hbm_traffic.json $P/
cp $O/instr_model_raw_counters.csv.gz $P/${tag}_instr_model_raw_counters.csv.gz
for f in $O/pmc_*_raw_counters.csv.gz; do cp $f $P/${tag}_$(basename $f); done
cp $O/trace_driver_flags_bench.json $P/${tag}_bench_driver_flags_under_rocprof.json
for t in driver_flags single_batch biprime c5; do cp $O/trace_${t}_kernel_stats.csv $P/${tag}_trace_${t}_kernel_stats.csv; done
cp $O/summary_driver_flags.txt $P/${tag}_bench_driver_flags_rocprof_summary.txt
cp $O/summary_biprime.txt $P/${tag}_bench_biprime_rocprof_summary.txt
cp $O/summary_c5.txt $P/${tag}_bench_c5_rocprof_summary.txt
cp $O/summary_single_batch.txt $P/${tag}_single_batch_split_kernel_rocprof_summary.txt
cp $O/summary_c3_single_stream_counters.txt $P/${tag}_c3_single_stream_counters_summary.txt
cp $O/summary_c3_saturated_issue_counters.txt $P/${tag}_c3_saturated_issue_counters_summary.txt
cp $O/sweep_shapes.txt $P/${tag}_sweep_shapes.txt
cp $O/small_batch_latency.txt $P/${tag}_small_batch_latency.txt
cp $O/sweep_generic.txt $P/${tag}_sweep_generic.txt
cp $O/bi_pivot_sweep.txt $P/${tag}_bi_pivot_sweep.txt
for f in ts_probe_2048 ts_probe_4096 lone_call_probe keygen_round_host_profile; do [ -f $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > $P/${tag}_$f.txt; done
cp $O/short_kernels.txt $P/${tag}_short_kernels.txt; cp $O/${tag}_short_kernels.json $P/
ls $P | grep -c "^${tag}_"
