#!/bin/bash
# Where the issue slots of the generic fixed-window modexp kernel go at key_length 1024 (configs[1]):
# saturated launches (8192 candidates x 40 modexps, one stream so that a dispatch's counters are its own) of both lane
# geometries that fit a 1031-bit modulus — 9 limbs per lane on groups of 4 lanes (the library's choice) and 18 on groups of 2
# — under a kernel trace and three counter passes (issue / wait, LDS / scalar, memory / branches), plus the instruction mix
# of the two instances from the disassembly.   usage (GPU box): bash tools/k1024_counters.sh <outdir>
out=${1:-gpurun_out/k1024}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/$out; mkdir -p $O
cd /tmp
for lpl in 9 18; do
  B="python3 $R/bench.py --workload biprime --key-length 1024 --batch 8192 --no-cpu-baseline --streams 1 --steps 3 --warmup 1 --limbs-per-lane $lpl"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_l$lpl -- $B > $O/bench_l$lpl.json 2>/dev/null
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_a_l$lpl -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_b_l$lpl -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $O/pmc_c_l$lpl -- $B > /dev/null 2>&1
done
cd $R
for lpl in 9 18; do
  python tools/prof_summary.py $O/summary_l$lpl.txt $O/trace_l$lpl $O/pmc_a_l$lpl $O/pmc_b_l$lpl $O/pmc_c_l$lpl 2>&1 | grep -E "powmod_kernel|## " > $O/summary_l${lpl}_powmod.txt
done
rm -rf $O/trace_l* $O/pmc_*
ls $O
