# usage: bash tools/ab_biprime.sh — biprimality-test modexps/s, narrow vs wide lane geometry
for kl in "2048 1024" "2048 4096" "1024 2048" "4096 256"; do
  set -- $kl
  for l in 9 18; do
    printf "key=%s cands=%s L=%s " $1 $2 $l
    python tools/bench_biprime.py --limbs-per-lane $l --key-length $1 --cands $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['modexps_per_s']), round(d['powmod_ms'],2), d['geometry'])"
  done
done
