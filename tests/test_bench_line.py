"""bench.py's last stdout line must stay parseable by the driver (it keeps the last 8 KB of stdout): a compact JSON
object with the contract fields, `roofline` and `cpu_baseline`, whatever the legs put into the full result."""
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
CANNED = sorted((ROOT / "profiles").glob("r0[345]_bench_*.json"))


@pytest.fixture(scope="module")
def bench():
    import bench as module          # import only: no GPU call, the helper it runs sets one environment variable

    return module


@pytest.mark.parametrize("path", CANNED, ids=lambda p: p.stem)
def test_result_line_is_compact_and_complete(bench, path):
    full = json.loads(path.read_text().strip().splitlines()[-1])
    if "metric" not in full:
        pytest.skip("not a bench.py result")
    text = bench.result_line(full)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT and len(text.encode()) < 8192
    line = json.loads(text)
    for key in CONTRACT:
        assert key in line, key
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert "workload" in line["config"]
    roof = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert key in roof, key
    if full["roofline"].get("frac") is not None:
        assert roof["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
        assert roof["achieved"] / roof["peak"] == pytest.approx(roof["frac"], rel=1e-3)
    if full.get("cpu_baseline"):
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
    if full.get("distributed"):
        assert line["distributed"]["world_size"] == full["distributed"]["world_size"]


def test_result_line_survives_oversized_legs(bench):
    """legs that grow (or fail with long messages) cost summaries, never the contract fields"""
    full = json.loads(CANNED[0].read_text().strip().splitlines()[-1]) if CANNED else None
    if full is None or "metric" not in full:
        full = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": "w"}, "roofline": {"bound": "b", "kernel": "k", "achieved": 1.0, "peak": 2.0, "unit": "x", "frac": 0.5,
                                                          "traffic": None, "kernel_ms": 1.0}, "cpu_baseline": None}
    full = dict(full)
    full["extra"] = {f"leg{k}": {"value": 1.0 * k, "roofline": {"frac": 0.5}, "blob": "x" * 5000} for k in range(400)}
    full["config"] = dict(full["config"], workload="w" * 3000, verified="v" * 3000)
    full["cpu_baseline"] = {"value": 1.0, "unit": "u", "cores": 1, "kind": "port", "sample": "s" * 9000}
    text = bench.result_line(full)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for key in CONTRACT:
        assert key in line, key


def test_emit_writes_one_stdout_line_and_the_details_file(bench, tmp_path, monkeypatch):
    import os

    monkeypatch.setattr(bench, "EXTRAS_FILE", tmp_path / "bench_extras.json")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    full = {"metric": "m", "value": 2.5, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "w"}, "roofline": None, "cpu_baseline": None, "single_batch": {"value": 3.0, "note": "n" * 100}}
    r, w = os.pipe()
    bench.emit_result(full, w)
    os.close(w)
    got = os.read(r, 1 << 16).decode()
    os.close(r)
    assert got.endswith("\n") and got.count("\n") == 1
    assert json.loads(got)["extra_summary"]["single_batch"] == {"value": 3.0}
    assert json.loads((tmp_path / "bench_extras.json").read_text()) == full


def test_committed_instruction_model_describes_the_kernels_as_they_are(bench):
    """bench.py prices the roofline with an instruction-count model fitted to SQ_INSTS_VALU of a particular version of
    the device code and refuses it (roofline.frac null) when the kernel sources have changed since.  A kernel edit
    without a recalibration (tools/gpu_session.sh <tag> profile) must fail HERE, not show up as a null in the driver's
    record."""
    model, reason = bench.instr_model()
    assert model is not None, reason
    for kind, L, nblk in (("n2", 18, 4), ("n2", 18, 8), ("n2split", 9, 8), ("n2split", 3, 24), ("generic", 18, 4), ("generic", 9, 4), ("generic", 3, 13)):
        assert bench.instr_per_wave(kind, L, nblk, 4192, 592) is not None, (kind, L, nblk)


def test_steps_in_flight_stay_within_the_stream_budget(bench):
    """bench.biprime_lanes / priority_aux_for: at most MAX_LANES lanes, companions for up to four of them — a process
    that has used more than ~24 streams is time-sliced by the queue scheduler from then on and every later leg pays
    (profiles/r04_bench_queue_budget.txt)."""
    for cands in (25, 100, 256, 512, 1024, 4096):
        for steps in (1, 6, 8, 12, 20, 48):
            lanes = bench.biprime_lanes(cands, steps)
            assert steps % lanes == 0 and lanes <= bench.MAX_LANES
            streams = lanes * (2 if bench.priority_aux_for(lanes) else 1)
            assert streams <= 8
    assert bench.biprime_lanes(256, 48) == 8 and not bench.priority_aux_for(8)
    assert bench.biprime_lanes(512, 16) == 4 and bench.priority_aux_for(4)
    assert not bench.priority_aux_for(1)


# --- `python bench.py --gpus N` started without a launcher ---------------------------------------
_RANK_SCRIPT = """
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
mode = sys.argv[sys.argv.index("--mode") + 1]
print("banner of rank", rank)                     # ranks other than 0: must not reach the parent's stdout
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(60)                                # a rank waiting in a collective its peer will never join
if rank == 0:
    print(json.dumps({"world": world, "argv": sys.argv[1:]}))
"""


def test_rank_command_and_environment(bench):
    cmd = bench.rank_command(["--gpus", "4", "--steps", "8"])
    assert cmd[0] == sys.executable and Path(cmd[1]) == ROOT / "bench.py" and cmd[2:] == ["--gpus", "4", "--steps", "8"]
    env = bench.rank_environment(3, 4, 29999, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == ("3", "3", "4", "127.0.0.1", "29999")
    assert env["PATH"] == "/bin"


def test_spawn_ranks_relays_rank0_line(bench, tmp_path, capfd):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = bench.spawn_ranks(["--gpus", "3", "--mode", "ok"], 3, script=script)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0
    assert json.loads(out[-1]) == {"world": 3, "argv": ["--gpus", "3", "--mode", "ok"]}
    assert sum(1 for ln in out if ln.startswith("{")) == 1


def test_spawn_ranks_propagates_a_failing_rank(bench, tmp_path, capfd):
    import time

    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    t0 = time.perf_counter()
    rc = bench.spawn_ranks(["--gpus", "2", "--mode", "fail"], 2, script=script)
    assert rc == 7 and time.perf_counter() - t0 < 30          # rank 0 was stopped, not waited for
    assert not [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]


def test_main_becomes_the_launcher_before_any_gpu_call(bench, monkeypatch):
    """--gpus N > 1 without WORLD_SIZE: main() hands over to spawn_ranks before torch is imported by bench.main"""
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "4", "--warmup", "1"])
    monkeypatch.setattr(bench, "spawn_ranks", lambda argv, world: seen.update(argv=argv, world=world) or 0)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0 and seen == {"argv": ["--gpus", "8", "--steps", "4", "--warmup", "1"], "world": 8}


def test_spawn_ranks_retries_on_a_taken_port_and_refuses_under_a_profiler(bench, tmp_path, capfd, monkeypatch):
    """the rendezvous port is released before the ranks bind it — a rank 0 that reports "address already in
    use" gets the whole job started again on another port; and a parent with a GPU profiler preloaded must not start
    rank processes at all (the profiler has initialised the GPU: an exec from there takes the machine down on this pool)."""
    marker = tmp_path / "first_attempt_done"
    script = tmp_path / "rank.py"
    script.write_text(f"""
import json, os, sys
rank = int(os.environ["RANK"])
marker = {str(marker)!r}
if not os.path.exists(marker):
    if rank == 0:
        open(marker, "w").write(os.environ["MASTER_PORT"])
        sys.stderr.write("RuntimeError: The server socket has failed to listen on any local network address. EADDRINUSE: address already in use\\n")
        sys.exit(1)
    import time; time.sleep(60)
if rank == 0:
    print(json.dumps({{"port": os.environ["MASTER_PORT"], "first": open(marker).read()}}))
""")
    rc = bench.spawn_ranks(["--gpus", "2"], 2, script=script)
    cap = capfd.readouterr()
    assert rc == 0, cap.err[-500:]
    res = json.loads(cap.out.strip().splitlines()[-1])
    assert res["port"] != res["first"] and "starting them again on another port" in cap.err
    # a failure that is not about the port is not retried
    marker.unlink()
    script.write_text("import os, sys\nsys.exit(5 if os.environ['RANK'] == '0' else 0)\n")
    assert bench.spawn_ranks(["--gpus", "2"], 2, script=script) == 5
    # profiler preload: refused before anything is started
    assert bench.profiler_preload({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"}) == "LD_PRELOAD"
    assert bench.profiler_preload({"ROCP_TOOL_LIBRARIES": "x.so"}) == "ROCP_TOOL_LIBRARIES"
    assert bench.profiler_preload({"PATH": "/bin", "LD_PRELOAD": "libjemalloc.so"}) == ""
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    script.write_text("open(%r, 'w').write('started')\n" % str(tmp_path / "started"))
    assert bench.spawn_ranks(["--gpus", "2"], 2, script=script) == 2
    assert not (tmp_path / "started").exists() and "profiler is preloaded" in capfd.readouterr().err


def test_roofline_frac_is_multiply_accumulates_against_the_multiply_issue_peak(bench):
    """VERDICT r05 item 4 / SURVEY 8(d): roofline.frac = multiply-accumulate wave-instructions / time / multiply issue
    peak; the kernel's bookkeeping instructions are not credited.  The mix-based figure stays as frac_issue_slots."""
    roof = bench.valu_roofline("k", instr_per_launch=1.8e10, launches=20, elapsed=0.66, kernel_ms=30.0, concurrent=4, mac_share=0.805, clock_mhz=2300.0)
    macs = 1.8e10 * 20 / 0.66 * 0.805
    peak = bench.SIMDS * bench.NOMINAL_HZ / bench.MAC_CYCLES
    assert roof["frac"] == pytest.approx(macs / peak) and roof["achieved"] == pytest.approx(macs / 1e9) and roof["peak"] == pytest.approx(peak / 1e9)
    assert roof["achieved"] / roof["peak"] == pytest.approx(roof["frac"])
    assert "multiply-accumulate" in roof["unit"]
    mix = bench.SIMDS * bench.NOMINAL_HZ / (0.805 * bench.MAC_CYCLES + 0.195 * bench.OTHER_CYCLES)
    assert roof["frac_issue_slots"] == pytest.approx(1.8e10 * 20 / 0.66 / mix) == pytest.approx(roof["issue_slots"]["frac"])
    assert roof["frac"] < roof["frac_issue_slots"] < roof["frac_at_measured_clock"]
    compact = bench.compact_roofline(roof)
    assert compact["frac"] == pytest.approx(roof["frac"], rel=1e-4) and compact["frac_issue_slots"] == pytest.approx(roof["frac_issue_slots"], rel=1e-4)
    assert "frac_macs_vs_multiply_issue_peak" not in compact
    none = bench.valu_roofline("k", None, 20, 0.66, 30.0, 4)
    assert none["frac"] is None and none["frac_issue_slots"] is None


def test_multi_rank_line_carries_per_rank_times_and_the_cpu_baseline(bench):
    """an N > 1 line has every rank's ms_per_step (min / max / list), the world size as the backend
    reports it, and a cpu_baseline (rank 0 times it after the timed region)."""
    bench.RANK_TIMES["elapsed_s"] = [0.60, 0.66, 0.63, 0.61]
    blk = bench.per_rank_block(20)
    assert blk["min"] == pytest.approx(30.0) and blk["max"] == pytest.approx(33.0) and len(blk["ranks"]) == 4
    full = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 4, "steps": 20, "warmup": 5, "ms_per_step": 33.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "w"},
            "roofline": {"bound": "b", "kernel": "k", "achieved": 1.0, "peak": 2.0, "unit": "x", "frac": 0.5, "traffic": None, "kernel_ms": 1.0},
            "cpu_baseline": {"value": 1200.0, "unit": "modexps/s", "cores": 16, "kind": "reference", "sample": "s"},
            "distributed": {"backend": "nccl", "world_size": 4, "ranks_expected": 4, "rccl_version": "2.22.3", "ms_per_step_per_rank": blk, "exchange": "x"}}
    line = json.loads(bench.result_line(full))
    assert line["distributed"]["world_size"] == 4 and line["distributed"]["ms_per_step_per_rank"]["max"] == pytest.approx(33.0)
    assert line["cpu_baseline"]["value"] == 1200.0 and line["cpu_baseline"]["cores"] == 16
    # the waiting ranks' barrier helper: rank 0 runs the function, the others do not
    assert bench.cpu_baseline_on_rank0(None, 0, lambda: 7) == 7 and bench.cpu_baseline_on_rank0(None, 3, lambda: 7) is None


def test_a_failing_leg_leaves_the_launch_shape_as_it_found_it(bench):
    """leg_keygen_round / leg_end_to_end / leg_single_batch / leg_latency reset the engine to automatic
    shapes; an exception inside must not leave that in place for the legs behind them."""
    class Eng:
        _lpl, _wpg = 18, 1

        def set_limbs_per_lane(self, v):
            self._lpl = v

        def set_wavefronts_per_group(self, v):
            self._wpg = v

    @bench.restores_launch_shape
    def leg(eng, fail):
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)
        if fail:
            raise AssertionError("inside the leg")
        return "ok"

    e = Eng()
    assert leg(e, False) == "ok" and (e._lpl, e._wpg) == (18, 1)
    with pytest.raises(AssertionError):
        leg(e, True)
    assert (e._lpl, e._wpg) == (18, 1)
    for name in ("leg_single_batch", "leg_latency", "leg_end_to_end", "leg_keygen_round"):
        assert hasattr(getattr(bench, name), "__wrapped__"), name
