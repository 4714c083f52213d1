#!/bin/bash
# One GPU session = one gpurun call:   gpurun --timeout S -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/ (merged back into the build container); the summaries that are cited are
# copied to profiles/ by hand.  Steps (tools/README.md has the table):
#   census        wrong-row census of overlapping launches for the shipped library and the pad variants
#                 (build/variants/pad*.so, tools/build_variant.py padN -DMX_DEV_PRIVATE_PAD_WORDS=N), over hardware queues,
#                 segments, one stream vs four, and the runtime's scratch knobs
#   exec_half     tools/ubench/exec_half: issue cost of VALU instructions with half of EXEC disabled
#   tests         pytest -m gpu (whole suite, log + durations)
#   stress        only the four-stream tests
#   bench         bench.py with the driver's flags and with the defaults (+ bench_extras.json)
#   biprime_small bench.py --workload biprime at the literal sizes of configs[1] and of an 8-GPU shard
#   prio_ab       short kernels at raised wave priority against the noprio variant build
#   lanes_fine    biprime steps in flight x lane geometry at the small shard sizes
#   lanes_queues  the same with / without companion streams and 16 / 24 / 32 hardware queues
#   ts_probe      time-sliced launches (9 and 18 limbs per lane, resident workgroups, units) against the plain shapes
#   decrypt_lanes c3 / c5 steps in flight beyond four
#   bench_queues  the whole default bench with 16 / 24 / 32 hardware queues
#   profile       tools/profile_round.sh <tag> (calibration, bench lines, rocprofv3 traces and counter passes)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
V=protocols/distributed_keygen_amd/build/variants
census() { python tools/concurrency_census.py "$@" >> $O/census.txt 2>> $O/census.err; }
for step in "$@"; do
  echo "== step $step $(date +%T)"
  case $step in
  census)
    : > $O/census.txt
    census --queues 16 --label shipped
    census --queues 16 --label shipped-timesliced --shape 9,2 --timeslice 2
    for v in pad64 pad256 pad1024; do
      [ -f $V/$v.so ] || continue
      MX_LIBRARY=$V/$v.so census --queues 16 --label $v
      MX_LIBRARY=$V/$v.so census --queues 16 --label $v --segments 1
      MX_LIBRARY=$V/$v.so census --queues 4 --label $v
      MX_LIBRARY=$V/$v.so census --queues 16 --label $v --one-stream
    done
    for v in pad256 pad1024; do
      [ -f $V/$v.so ] || continue
      HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 MX_LIBRARY=$V/$v.so census --queues 16 --label "$v HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0"
      HSA_SCRATCH_SINGLE_LIMIT=4294967296 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4294967296 MX_LIBRARY=$V/$v.so census --queues 16 --label "$v HSA_SCRATCH_SINGLE_LIMIT=4G"
      HSA_SCRATCH_SINGLE_LIMIT=1048576 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=1048576 MX_LIBRARY=$V/$v.so census --queues 16 --label "$v HSA_SCRATCH_SINGLE_LIMIT=1M"
      MX_LIBRARY=$V/$v.so census --queues 16 --streams 2 --label "$v two streams"
      MX_LIBRARY=$V/$v.so census --queues 16 --rows 4096 --label "$v 4x4096 rows (1024 wavefronts in flight)"
    done
    grep -E "^==|WRONG|pad faults [a-z ]*[0-9]*: [1-9]" $O/census.txt | tail -60
    ;;
  census_fr)
    # the friendly-modulus instances of the one-wavefront wide kernel (round 3's dropped variant, rebuilt): four streams,
    # segments 4 / 1 / 7, against the plain instances (knob) — and the time-sliced launches' first overlapping round
    : > $O/census.txt
    census --queues 16 --label "friendly 1w"
    census --queues 16 --label "friendly 1w" --segments 1
    census --queues 16 --label "friendly 1w" --segments 7
    census --queues 4 --label "friendly 1w"
    census --queues 16 --label "plain 1w (knob)" --knob n2_friendly_1w=1
    census --queues 16 --label "time-sliced" --shape 9,2 --timeslice 2 --reps 4
    census --queues 16 --label "time-sliced, 2 streams" --shape 9,2 --timeslice 2 --reps 3 --streams 2
    [ -f $V/r03.so ] && MX_LIBRARY=$V/r03.so census --queues 16 --label "round-3 library time-sliced" --shape 9,2 --timeslice 2 --reps 4
    [ -f $V/r03.so ] && MX_LIBRARY=$V/r03.so census --queues 16 --label "round-3 library 18x1w"
    grep -E "^==|WRONG|  rep " $O/census.txt | tail -70
    ;;
  ts_first)
    # why the FIRST round of four overlapping time-sliced launches took seconds (r04a/r04b): per-launch start/end times for
    # the shipped library and for builds with three workgroups per CU (round 3's register budget) / without wavefront A's
    # agent-scope release
    : > $O/census.txt
    census --queues 16 --label "shipped time-sliced" --shape 9,2 --timeslice 2 --reps 3
    for v in ts_regs3 ts_nofence; do
      [ -f $V/$v.so ] && MX_LIBRARY=$V/$v.so census --queues 16 --label "$v time-sliced" --shape 9,2 --timeslice 2 --reps 3
    done
    census --queues 16 --label "shipped time-sliced 3 streams" --shape 9,2 --timeslice 2 --reps 2 --streams 3
    census --queues 16 --label "shipped time-sliced 8192 rows" --shape 9,2 --timeslice 2 --reps 2 --rows 8192
    census --queues 16 --label "shipped time-sliced, second process" --shape 9,2 --timeslice 2 --reps 2
    grep -E "^==|WRONG|  rep " $O/census.txt | tail -70
    ;;
  dist_paths)
    # the N > 1 code paths on one GPU: two ranks over gloo (slices, gathers, the distributed biprime leg), one rank over RCCL
    MX_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.err; tail -c 1200 $O/bench_two_ranks_gloo_one_gpu.json; echo
    MX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_rccl_single_rank.json 2> $O/bench_rccl_single_rank.err; tail -c 600 $O/bench_rccl_single_rank.json; echo
    MX_BENCH_FORCE_DIST=1 python bench.py --workload biprime --batch 512 --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_biprime_c512_rccl_single_rank.json 2>/dev/null; tail -c 500 $O/bench_biprime_c512_rccl_single_rank.json; echo
    ;;
  smoke)
    ( time python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
    ;;
  graph_probe)
    python tools/graph_probe.py 1024 256 > $O/graph_probe.txt 2>&1; python tools/graph_probe.py 2048 100 >> $O/graph_probe.txt 2>&1; grep -v amdgpu.ids $O/graph_probe.txt | tail -30
    ;;
  biprime_lanes)
    for spec in "2048 512" "2048 1024" "1024 256"; do set -- $spec
      for st in 1 2 3 4 6; do
        python bench.py --workload biprime --key-length $1 --batch $2 --streams $st --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('biprime k$1 c$2 streams $st:', round(d['value']), 'modexps/s', round(d['ms_per_step'],2), 'ms/step frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'), d['config'].get('geometry_K_L_W_blocks'))"
      done
    done | tee $O/biprime_lanes.txt
    ;;
  keygen_profile)
    python tools/keygen_round_profile.py 65536 > $O/keygen_round_profile.txt 2>&1; head -60 $O/keygen_round_profile.txt
    ;;
  soak)
    for seed in 4 5; do python tools/soak_round4.py $seed 150; done > $O/soak.txt 2>&1; tail -4 $O/soak.txt
    ;;
  sweep_split)
    python tools/sweep_split.py 2048 4096 > $O/sweep_split.txt 2>&1; tail -40 $O/sweep_split.txt
    ;;
  sweep_generic)
    python tools/sweep_generic.py 1024 2048 > $O/sweep_generic.txt 2>&1; cat $O/sweep_generic.txt | tail -80
    ;;
  keygen)
    ( time python -m pytest tests/test_gpu_keygen_flow.py tests/test_gpu_standin.py -m gpu -x -q ) > $O/pytest_keygen.log 2>&1; tail -5 $O/pytest_keygen.log
    ;;
  counters_avail)
    ( cd /tmp && rocprofv3 --list-avail 2>&1 ) > $O/counters_avail.txt; grep -c . $O/counters_avail.txt; grep -oE "\bSQ_[A-Z0-9_]+" $O/counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
    ;;
  instances)
    ( time python -m pytest tests/test_gpu_instances.py -m gpu -x -q --durations=5 ) > $O/pytest_instances.log 2>&1; tail -12 $O/pytest_instances.log
    ;;
  bench_ab)
    for rep in 1 2; do
      for kn in "" "--knob n2_friendly_1w=1"; do
        python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 $kn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 driver flags [$kn]', round(d['value']), 'modexps/s', round(d['ms_per_step'],3), 'ms/step kernel_ms', round(d['roofline']['kernel_ms'],2), 'clock', d['roofline'].get('shader_clock_mhz_measured'))"
        python bench.py --workload c5 --batch 16384 --streams 2 --steps 4 --warmup 2 --no-extras --no-cpu-baseline $kn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 16384 x 2 in flight [$kn]', round(d['value']), 'modexps/s', round(d['ms_per_step'],2), 'ms/step')"
      done
    done | tee $O/bench_ab.txt
    ;;
  exec_half)
    tools/ubench/exec_half > $O/exec_half.txt 2>&1; tail -40 $O/exec_half.txt
    ;;
  tests)
    ( time python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
    ;;
  stress)
    ( time python -m pytest tests/test_gpu_stress.py -m gpu -x -q -k "four_streams" --durations=8 ) > $O/pytest_stress.log 2>&1; tail -15 $O/pytest_stress.log
    ;;
  bench)
    python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; cp bench_extras.json $O/bench_driver_flags_extras.json
    tail -c 2500 $O/bench_driver_flags.json; wc -c $O/bench_driver_flags.json
    ;;
  biprime_small)
    for spec in "1024 256" "2048 512" "2048 100" "2048 25" "2048 1024" "2048 2048" "2048 4096"; do set -- $spec
      steps=16; [ "$2" -le 256 ] && steps=48
      python bench.py --workload biprime --key-length $1 --batch $2 --steps $steps --warmup 4 --no-cpu-baseline > $O/bench_biprime_k$1_c$2.json 2>/dev/null
      python -c "import json,sys; d=json.loads(open('$O/bench_biprime_k$1_c$2.json').read().strip().splitlines()[-1]); print('biprime k$1 c$2', round(d['value']), 'modexps/s', round(d['ms_per_step'],2), 'ms/step frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'))"
    done
    ;;
  prio_ab)
    # short kernels at raised wave priority (csrc/mx_prio.hpp) against a build without (build_variant.py noprio -DMX_DEV_AUX_WAVE_PRIO=0)
    line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms/step frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'))"; }
    for rep in 1 2; do for lib in shipped $V/noprio.so; do
      [ $lib = shipped ] && unset MX_LIBRARY || export MX_LIBRARY=$R/$lib
      for st in 6 8 12; do python bench.py --workload biprime --key-length 1024 --batch 256 --streams $st --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | line "$lib k1024 c256 lanes $st:"; done
      for st in 4 8; do python bench.py --workload biprime --key-length 2048 --batch 512 --streams $st --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | line "$lib k2048 c512 lanes $st:"; done
      python bench.py --workload biprime --key-length 2048 --batch 100 --streams 6 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | line "$lib k2048 c100 lanes 6:"
      python bench.py --workload biprime --no-cpu-baseline 2>/dev/null | line "$lib k2048 c4096:"
      python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | line "$lib c3 driver flags:"
    done; done | tee $O/prio_ab.txt
    unset MX_LIBRARY
    ;;
  lanes_fine)
    # steps in flight x lane geometry for the small biprime shards (after csrc/mx_prio.hpp)
    line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms/step frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'), d['config'].get('geometry_K_L_W_blocks'))"; }
    for spec in "1024 256 48" "2048 100 48" "2048 25 48" "2048 256 48" "2048 512 48" "2048 1024 24"; do set -- $spec
      for lpl in -1 9 18; do for st in 4 6 8 12 16 24; do
        [ $((${3} % st)) -ne 0 ] && continue
        python bench.py --workload biprime --key-length $1 --batch $2 --streams $st --limbs-per-lane $lpl --steps $3 --warmup $st --no-cpu-baseline 2>/dev/null | line "k$1 c$2 lpl $lpl lanes $st:"
      done; done
    done | tee $O/lanes_fine.txt
    ;;
  lanes_queues)
    # do the lanes need their high-priority companion streams now that the short kernels raise their wave priority, and
    # is the optimum of 8 lanes the 16 hardware queues (8 lanes + 8 companions)?
    line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms/step frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'), d['config'].get('geometry_K_L_W_blocks'))"; }
    for spec in "1024 256 9" "2048 100 18" "2048 25 9" "2048 512 18"; do set -- $spec
      for q in 16 24 32; do for aux in 1 0; do for st in 4 8 12 16; do
        python bench.py --workload biprime --key-length $1 --batch $2 --streams $st --limbs-per-lane $3 --hw-queues $q --priority-aux $aux --steps 48 --warmup $st --no-cpu-baseline 2>/dev/null | line "k$1 c$2 lpl $3 queues $q aux $aux lanes $st:"
      done; done; done
    done | tee $O/lanes_queues.txt
    ;;
  ts_probe)
    python tools/ts_probe.py 2048 > $O/ts_probe_2048.txt 2>&1; cat $O/ts_probe_2048.txt
    python tools/ts_probe.py 4096 > $O/ts_probe_4096.txt 2>&1; cat $O/ts_probe_4096.txt
    ;;
  decrypt_lanes)
    # partial decryption + recombination with more steps in flight than the default four (the recombination kernel raises
    # its wave priority; beyond 8 lanes it runs on the lane's own stream)
    line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms/step frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'), d['config'].get('geometry_K_L_W_blocks'), d['config'].get('wavefronts_per_group'))"; }
    for spec in "c5 1024 48" "c5 2048 48" "c5 4096 24" "c3 2048 48" "c3 4096 48" "c3 10000 24"; do set -- $spec
      for st in 4 6 8 12 16; do
        [ $((${3} % st)) -ne 0 ] && continue
        python bench.py --workload $1 --batch $2 --streams $st --steps $3 --warmup $st --no-cpu-baseline --no-extras 2>/dev/null | line "$1 b$2 lanes $st:"
      done
    done | tee $O/decrypt_lanes.txt
    ;;
  bench_queues)
    # the whole default bench (all legs in one process) with 16 / 24 / 32 hardware queues
    for q in 16 24 32; do
      python bench.py --steps 20 --warmup 5 --hw-queues $q --cpu-seconds 1 > $O/bench_q$q.json 2> $O/bench_q$q.err; cp bench_extras.json $O/bench_q${q}_extras.json
      python - $O/bench_q${q}_extras.json $q <<'PY'
import json, sys
e = json.load(open(sys.argv[1]))
print("queues", sys.argv[2], "headline", round(e["value"]), "single_batch", round(e["single_batch"]["value"]), "end_to_end", round(e["end_to_end"]["value"]),
      " ".join(f"{k} {round(v['value'])}" for k, v in e["extra"].items() if isinstance(v, dict) and "value" in v), "keygen", round(e["end_to_end_keygen"]["value"]))
PY
    done | tee $O/bench_queues.txt
    ;;
  profile)
    bash tools/profile_round.sh $tag
    ;;
  *) echo "unknown step $step";;
  esac
done
echo "== done $(date +%T)"
