#!/bin/bash
# Proves that the driver-run suite would catch the hand-over race of rounds 3-4 again (VERDICT r05 "Next round" 1c).
#   1. the shipped library passes tests/test_gpu_handover.py and the disassembly test of tests/test_instances.py;
#   2. a library built with -DMX_DEV_TS_COMPILER_RELEASE (csrc/mx_dev.hpp: the publish sequence that lost groups) FAILS both.
# usage (GPU box, one gpurun call):  bash tools/prove_handover_guard.sh <outdir>      record: profiles/r06_handover_guard.txt
# The variant is built here when it is not in the tree already (tools/build_variant.py; hipcc cross-compiles anywhere).
out=${1:-gpurun_out/handover_guard}
mkdir -p "$out"
V=protocols/distributed_keygen_amd/build/variants/compiler_release.so
[ -f "$V" ] || python tools/build_variant.py compiler_release -DMX_DEV_TS_COMPILER_RELEASE
{
  echo "== 1. library as shipped: must PASS"
  timeout 900 python -m pytest tests/test_gpu_handover.py tests/test_instances.py -q -p no:cacheprovider -k "handovers or write_back" 2>&1 | tail -8
  shipped=${PIPESTATUS[0]}
  echo "rc=$shipped"
  echo "== 2. -DMX_DEV_TS_COMPILER_RELEASE (the entry published with __hip_atomic_store(..., __ATOMIC_RELEASE, agent)): must FAIL"
  MX_LIBRARY=$PWD/$V timeout 900 python -m pytest tests/test_gpu_handover.py tests/test_instances.py -q -p no:cacheprovider -k "handovers or write_back" 2>&1 \
    | grep -E "^(FAILED|PASSED|ERROR)|passed|failed|wrong rows, " | cut -c1-400 | tail -40
  broken=${PIPESTATUS[0]}
  echo "rc=$broken"
  if [ "$shipped" = 0 ] && [ "$broken" = 1 ]; then echo "GUARD PROVEN: the suite passes on the shipped library and fails on the broken publish sequence"
  else echo "GUARD NOT PROVEN (shipped rc=$shipped, broken-variant rc=$broken)"; fi
} 2>&1 | tee "$out/handover_guard.txt"
grep -q "^GUARD PROVEN" "$out/handover_guard.txt"
