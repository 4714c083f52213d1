#!/bin/bash
# A/B of the LDS stride between the groups of a wavefront (lever b): the shipped library (152 words
# per 18-limb group of 4 lanes: the 16 groups' broadcast reads of a multiplier limb fall on 4 banks) against a build with
# -DMX_DEV_LDS_PAD_WORDS=1 (153 words: 16 banks).  usage (GPU box): bash tools/lds_stride_ab.sh <outdir>
# needs protocols/distributed_keygen_amd/build/variants/lds_pad1.so (tools/build_variant.py lds_pad1 -DMX_DEV_LDS_PAD_WORDS=1)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/lds_ab}; mkdir -p $O
V=$R/protocols/distributed_keygen_amd/build/variants/lds_pad1.so
val() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; }
{
echo "== headline (4 steps in flight, driver flags): value ms_per_step kernel_ms"
for round in 1 2 3; do
  for lib in shipped lds_pad1; do
    if [ $lib = shipped ]; then unset MX_LIBRARY; else export MX_LIBRARY=$V; fi
    printf "%s " $lib; python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | val
  done
done
echo "== saturated single stream (32768 ciphertexts, 18 limbs x 1 wavefront, 1 segment)"
for round in 1 2; do
  for lib in shipped lds_pad1; do
    if [ $lib = shipped ]; then unset MX_LIBRARY; else export MX_LIBRARY=$V; fi
    printf "%s " $lib; python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --wavefronts-per-group 1 --segments 1 --batch 32768 --steps 4 --warmup 1 2>/dev/null | val
  done
done
echo "== lone 10000 batch (library's shape)"
for lib in shipped lds_pad1; do
  if [ $lib = shipped ]; then unset MX_LIBRARY; else export MX_LIBRARY=$V; fi
  printf "%s " $lib; python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --steps 6 --warmup 2 2>/dev/null | val
done
} > $O/lds_stride_ab.txt 2>&1
cd /tmp
SAT="python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --limbs-per-lane 18 --wavefronts-per-group 1 --segments 1 --batch 32768 --steps 3 --warmup 1"
for lib in shipped lds_pad1; do
  if [ $lib = shipped ]; then unset MX_LIBRARY; else export MX_LIBRARY=$V; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_$lib -- $SAT > /dev/null 2>&1
  python3 $R/tools/prof_summary.py $O/pmc_${lib}_summary.txt $O/pmc_$lib $O/pmc_$lib > /dev/null
  rm -rf $O/pmc_$lib
done
unset MX_LIBRARY
cat $O/lds_stride_ab.txt
grep -h "powmod_n2_kernel" $O/pmc_*_summary.txt | grep -E "LDS|WAVE_CYCLES|WAIT_INST_ANY" 
