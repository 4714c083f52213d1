export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# two ranks on the one GPU over gloo: the world > 1 code path of bench.py (c3 + the distributed biprime leg)
MX_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -3 $O/bench_gloo2.err; cut -c1-400 $O/bench_gloo2.json
MX_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --workload biprime --steps 3 --warmup 1 > $O/bench_biprime_gloo2.json 2> $O/bench_biprime_gloo2.err; tail -3 $O/bench_biprime_gloo2.err; cut -c1-300 $O/bench_biprime_gloo2.json
bash tools/profile_round.sh r02 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
