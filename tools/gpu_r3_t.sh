# round 3, call T: instruction-cache counters of the two-wavefront instances (is <8,18> missing more than <4,18>?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
cd /tmp
rocprofv3 -L > $R/$O/counters_list.txt 2>&1
grep -i "icache\|SQC_" $R/$O/counters_list.txt | head -40
P="python3 $R/tools/icache_probe.py 4096 1024 18 2 2048 2048 18 2 4096 1024 9 2 2048 2048 9 2 2048 8192 18 1"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace --output-format csv -d $R/$O/pmc_icache -- $P > $R/$O/run1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_IFETCH --kernel-trace --output-format csv -d $R/$O/pmc_sq -- $P > $R/$O/run2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('pmc_icache','pmc_sq'):
    for f in glob.glob(f'gpurun_out/r03t/{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'powmod_n2' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(d, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
tail -3 $O/run1.log
