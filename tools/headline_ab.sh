#!/bin/bash
# Same-box A/B of the headline: the round-4 and round-5 trees (each with its own bench.py and its
# own library, copied to .ab/r04 and .ab/r05 by hand from `git worktree` checkouts built with their own build.py) and the
# current tree, three alternating runs each of  bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline.
# usage (GPU box): bash tools/headline_ab.sh <outdir>         record: profiles/r06_headline_ab.txt
out=${1:-gpurun_out/headline_ab}
mkdir -p "$out"
root=$PWD
run() {   # <label> <dir>
  (cd "$2" && timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 |
     python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$1', 'value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', r.get('kernel_ms'), 'clock_mhz', r.get('shader_clock_mhz_measured'))")
}
{
  for rep in 1 2 3; do
    [ -d .ab/r04 ] && run r04 "$root/.ab/r04"
    [ -d .ab/r05 ] && run r05 "$root/.ab/r05"
    run r06 "$root"
  done
} 2>&1 | tee "$out/headline_ab.txt"
