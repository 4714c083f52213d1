# round 3, call X: RCCL channels of the one-rank process-group path (what the all-gather costs the modexp kernels)
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
for ch in default 1 2 4 8; do
  if [ "$ch" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1; fi
  for rep in 1 2; do
  printf "channels %s: " "$ch"
  MX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
  done
done > $O/rccl_channels.txt 2>&1
unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS
printf "no process group: " >> $O/rccl_channels.txt
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))" >> $O/rccl_channels.txt
cat $O/rccl_channels.txt
