#!/usr/bin/env python3
"""Headline benchmark: modexps/sec (2048-bit N, mod N^2) — BASELINE.json's metric.

Workloads:
  c3 (default, BASELINE.json configs[2])  one 3-party threshold-Paillier key, key_length 2048, t = 1; a
      step = one party's pass over a batch of 10 000 ciphertexts:
        partial decryption   c^exp_i mod N^2   (paillier_shared_key.py:92 looped at distributed_keygen.py:463-466)
        share recombination  of the 3 partials (paillier_shared_key.py:95-127 looped at distributed_keygen.py:510-515)
      With N GPUs every rank processes its own batch (weak scaling, batches are independent) and the
      partial-decryption rows are all-gathered over RCCL — the only exchange step the path has.
  biprime (configs[3])  5-party key_length 2048, t = 2: a step = one party's biprimality-test pass over
      a batch of candidate moduli that survived the sieve: 160 Jacobi symbols, selection of the first
      40 generators with symbol 1 and 40 modexps g^e mod N per candidate (distributed_keygen.py:1084-1099
      looped at :1313-1329), then the verdict of every candidate from all parties' v values
      (:1110-1175 looped at :1339-1360).  With N GPUs the candidates are sharded contiguously; the v
      rows and the verdict bytes are all-gathered (the biprimality vote, :1331-1360).
  c5 (configs[4])  c3 at key_length 4096 (8200-bit modulus), batch 4096.

  python bench.py --gpus N --steps K --warmup W      (N > 1: under torch.distributed.run, or started directly — it then
                                                      spawns the N ranks itself, spawn_ranks below)

Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0) whose `roofline`
is the VALU instruction-issue roof (the path is integer-only: no MFMA, HBM three orders of
magnitude away) with the HBM figures as a sub-block, plus `cpu_baseline` (the
reference's gmpy2 engine on the host cores) and, on one GPU, the legs `single_batch`, `latency`,
`end_to_end`, `end_to_end_keygen`, `extra.biprime_k2048`, `extra.biprime_k1024_c256/_c8192` and
`extra.c5_k4096[_b1024/_b16384]`.  DESIGN.md §6 defines every field.
"""

from __future__ import annotations

import argparse
import json
import os
import random
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# The HIP runtime multiplexes streams onto 4 hardware queues by default; two streams that share a queue
# serialise (measured: 4 streams 203 k modexps/s with 4 queues, 246-272 k with 8; tools/ab_queues.sh; the
# chunks of a 40 000-ciphertext int-level call after other streams have been used: 170-199 k/s with 8
# queues, 273 k/s with 16; profiles/r02_hw_queue_collisions.txt).  Must be set before the runtime initialises.
from protocols.distributed_keygen_amd import configure_hw_queues  # noqa: E402  (no GPU call: sets GPU_MAX_HW_QUEUES)

HW_QUEUES = int(sys.argv[sys.argv.index("--hw-queues") + 1]) if "--hw-queues" in sys.argv[:-1] else 16
configure_hw_queues(HW_QUEUES)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU issue on MI355X, measured with fully independent instruction streams, wall clock AND counters
# (tools/ubench/valu_peak.hip, profiles/r03_ubench_valu_peak.txt; settles VERDICT r02 "weak" 2):
#   simple 32-bit VALU (v_fma_f32, v_add_u32)        2.25-2.30 cycles per wave64 instruction per SIMD — the SIMD-32
#                                                     issue MI355X_MICROARCH.md states (2 cycles);
#   every integer multiply (v_mul_lo_u32, v_mad_u64_u32) and v_pk_fma_f32   4.14-4.21 cycles (half rate);
#   shader clock under load: 2.08-2.17 GHz with every SIMD streaming v_mad_u64_u32, not the nominal 2.4 GHz.
# The modexp kernels are ~80 % v_mad_u64_u32, so the roof that binds them is the MULTIPLY issue rate; the guide's
# vector peak (any instruction at 2 cycles) is reported beside it.
SIMDS, NOMINAL_HZ = 1024, 2.4e9
MAC_CYCLES, OTHER_CYCLES = 4.19, 2.28       # v_mad_u64_u32 (accumulator form) / plain VALU, >= 2 wavefronts per SIMD
GUIDE_VECTOR_PEAK = SIMDS * NOMINAL_HZ / 2  # wave-instructions per second if every instruction issued in 2 cycles
MAC_ISSUE_PEAK = SIMDS * NOMINAL_HZ / MAC_CYCLES
INSTR_MODEL = ROOT / "profiles" / "r06_instr_model.json"       # tools/calibrate_instr.py (SQ_INSTS_VALU fits) + digest of the kernel sources
def _latest_profile(name: str) -> Path:
    """profiles/r0N_<name> of the latest round that committed one (the per-round measurement files a later round did not
    repeat stay valid as long as the kernels they describe are the same machine code — see INSTR_MODEL's digest)."""
    found = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{name}"))
    return found[-1] if found else ROOT / "profiles" / f"r06_{name}"


HBM_MEASURED = _latest_profile("hbm_traffic.json")             # tools/hbm_traffic.py (FETCH_SIZE / WRITE_SIZE passes)


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 48 timed steps: with 4 steps in flight the last round drains a partly empty machine, which costs
    # ~3 % of a 20-step run and ~1 % of a 48-step one
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=("c3", "biprime", "c5"), default="c3")
    ap.add_argument("--batch", type=int, default=0,
                    help="units per step per GPU: ciphertexts (c3: 10000, c5: 4096) or candidate moduli (biprime: 4096/N)")
    ap.add_argument("--key-length", type=int, default=0, help="default 2048 (c3, biprime), 4096 (c5)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="biprime workload on N GPUs: strong = --batch (default 4096) candidates sharded over the ranks (configs[3]), "
                         "weak = that many per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single_batch / end_to_end / extra legs")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--check", type=int, default=6, help="elements verified against CPython pow after timing")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent steps kept in flight, one HIP stream each; 1 = strictly one batch at a time; "
                         "0 = automatic: the first of 4, 5, 6, 7, 8, 3 that divides --steps, else 4")
    ap.add_argument("--limbs-per-lane", type=int, default=-1,
                    help="lane geometry 3|9|18; 0 / default = the library's choice for the launches in flight taken together")
    ap.add_argument("--wavefronts-per-group", type=int, default=0,
                    help="c3/c5 pair kernel: 1 | 2 wavefronts per group of elements, 0 = the library's choice")
    ap.add_argument("--cu-slices", type=int, default=-1,
                    help="c3/c5: 1 = every step in flight on a stream confined to its own slice of the CUs, 0 = ordinary streams, "
                         "-1 = slices when the launches in flight fit the chip side by side at one wavefront per SIMD")
    ap.add_argument("--segments", type=int, default=0,
                    help="c3/c5: launches per exponentiation (mx_powmod_nsquare_run), 0 = the library's choice")
    ap.add_argument("--hw-queues", type=int, default=16, help="developer: HIP hardware queues of this process (read before the runtime starts)")
    ap.add_argument("--priority-aux", type=int, default=-1,
                    help="developer, biprime: 1 / 0 = short kernels on a high-priority companion stream or on the lane's own stream; -1 = companion when several steps are in flight")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE",
                    help="developer override of the library for A/B runs (Engine.debug_knob), e.g. n2_friendly_1w=1")
    ap.add_argument("--generic-modulus", action="store_true",
                    help="c3/c5: time mx_powmod_shared on the modulus N^2 instead of the N-adic pair kernel")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the reference's engine (gmpy2.powmod -> libgmp mpz_powm) on the host cores
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(mod: int, exp: int, bases, seconds: float, what: str) -> dict:
    job = {"mod": hex(mod), "exp": hex(exp), "bases": [hex(c) for c in bases], "nprocs": 0, "seconds": seconds}
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(job, f)
        path = f.name
    script = str(ROOT / "oracle" / "cpu_baseline.py")
    tried = []
    for py in ("/opt/conda/bin/python3.9", sys.executable):
        if not os.path.exists(py):
            continue
        try:
            r = subprocess.run([py, script, path], capture_output=True, text=True, timeout=seconds * 6 + 120)
            if r.returncode == 0 and r.stdout.strip():
                res = json.loads(r.stdout.strip().splitlines()[-1])
                if py != sys.executable and res["engine"] != "gmpy2":
                    tried.append(f"{py}: no gmpy2")
                    continue
                os.unlink(path)
                ci = res["core_info"]
                return {
                    "value": res["rate_all_cores"], "unit": "modexps/s", "cores": res["cores"],
                    "kind": "reference", "engine": res["engine_desc"],
                    "single_core_value": res["rate_single_core"],
                    "parallel_efficiency": res["parallel_efficiency"],
                    "effective_cores": res["rate_all_cores"] / res["rate_single_core"],
                    "cores_basis": (f"processes = usable cores: sched_getaffinity {ci['affinity']}, cgroup quota "
                                    f"{ci['cgroup_quota']}, os.cpu_count {ci['host_cpu_count']}"),
                    "sample": (f"{res['modexps_timed']} modexps in {res['wall_s']:.1f} s wall on {res['cores']} processes; "
                               f"{what}; engine = the routine the reference's pow_mod dispatches to "
                               "(gmpy2.powmod -> libgmp mpz_powm)"),
                }
            tried.append(f"{py}: rc={r.returncode} {r.stderr[-200:]}")
        except Exception as exc:  # pragma: no cover - measurement plumbing
            tried.append(f"{py}: {exc}")
    os.unlink(path)
    return {"value": None, "unit": "modexps/s", "cores": 0, "kind": "reference", "sample": "failed: " + "; ".join(tried)}


# ---------------------------------------------------------------------------------------------------
# roofline helpers
# ---------------------------------------------------------------------------------------------------
def _load_json(path: Path):
    try:
        return json.loads(path.read_text())
    except Exception:
        return None


_MODEL_STATE = {}


def instr_model():
    """(model, reason): the committed instruction-count model, or (None, why) if it does not describe the kernels
    this run launches — it was fitted to counters of a particular version of the device code, and a kernel edit
    silently invalidates it (VERDICT r02 "weak" 10): the digest of the kernel sources must match."""
    if "m" not in _MODEL_STATE:
        model, reason = _load_json(INSTR_MODEL), None
        if model is None:
            reason = f"{INSTR_MODEL.name} not found"
        else:
            from tools.calibrate_instr import kernel_code_digest, kernel_sources_digest

            if model.get("kernel_code_sha256"):         # round 6: the machine code of the modelled kernels in the built library
                try:
                    same = model["kernel_code_sha256"] == kernel_code_digest()
                except Exception as exc:
                    same, reason = False, f"{INSTR_MODEL.name}: the library's kernel code could not be read ({exc})"
            else:                                       # older models: the source headers
                same = model.get("kernel_sources_sha256") == kernel_sources_digest()
            if not same:
                model, reason = None, reason or (f"{INSTR_MODEL.name} was fitted to other kernel code (digest mismatch): re-run "
                                                 "tools/calibrate_instr.py under rocprofv3 --pmc SQ_INSTS_VALU")
        _MODEL_STATE["m"] = (model, reason)
    return _MODEL_STATE["m"]


def instr_per_wave(kind: str, L: int, nblk: int, n_sqr: int, n_mul: int):
    """VALU wave-instructions one wavefront (kind "n2split": one PAIR of wavefronts) of a modexp launch executes."""
    model, _ = instr_model()
    try:
        i_sqr, i_mul, fixed = model[kind][str(L)][str(nblk)]
    except Exception:
        return None
    return n_sqr * i_sqr + n_mul * i_mul + fixed


def valu_roofline(kernel: str, instr_per_launch, launches: int, elapsed: float, kernel_ms: float, concurrent: int,
                  mac_share=None, clock_mhz=None) -> dict:
    """Roofline of an integer-VALU kernel as SURVEY.md §8(d) defines it: the binding roof is the integer multiply issue
    rate, so `achieved` = multiply-accumulate wave-instructions of all timed launches / wall time, `peak` = the rate at
    which the chip issues them (SIMDs x nominal clock / 4.19 cycles, measured), `frac` = achieved / peak.  The kernel's
    other VALU instructions (carries, quotient digits, lane exchanges: mac_share < 1 of the mix) are NOT credited as
    achieved work.  Beside it, under `issue_slots` / `frac_issue_slots`: all VALU wave-instructions
    against what THIS instruction mix could issue (mac_share at 4.19 cycles, the rest at 2.28) — how full the VALU issue
    slots are, which is what says whether scheduling or only fewer instructions can still help —, the same at the shader
    clock measured during the run, and everything against the guide's 2-cycle vector peak."""
    out = {
        "bound": "valu-multiply-issue", "kernel": kernel, "unit": "G multiply-accumulate wave-instructions/s",
        "kernel_ms": kernel_ms, "concurrent_launches": concurrent,
        "instructions_per_launch": instr_per_launch,
        "instructions_basis": "n_waves x (n_sqr x I_sqr + n_mul x I_mul + F): squarings/multiplications from the "
                              "exponent's tape, per-instance constants fitted to SQ_INSTS_VALU (profiles/r06_instr_model.json, "
                              "refused when the digest of the kernel sources it records no longer matches); x mac_share "
                              "(v_mad_u64_u32 + v_mul_* share of the kernel's VALU instructions, tools/isa_mix.py) = multiply-accumulates",
        "peak_basis": (f"{SIMDS} SIMDs x {NOMINAL_HZ / 1e9} GHz / {MAC_CYCLES} cycles per integer multiply(-accumulate) wave-instruction: the issue "
                       "cost measured on this chip with independent streams, wall clock and SQ_INSTS_VALU (profiles/r03_ubench_valu_peak.txt; "
                       f"plain VALU instructions: {OTHER_CYCLES} cycles)"),
        "shader_clock_mhz_measured": clock_mhz,
        "shader_clock_basis": "mx_clock_probe wavefronts running beside the timed steps (s_memtime / s_memrealtime x 100 MHz)",
    }
    if instr_per_launch is None:
        out.update({"achieved": None, "peak": None, "frac": None, "frac_issue_slots": None})
        return out
    all_valu = instr_per_launch * launches / elapsed
    share = 1.0 if mac_share is None else mac_share
    mix_cycles = share * MAC_CYCLES + (1 - share) * OTHER_CYCLES
    mix_peak = SIMDS * NOMINAL_HZ / mix_cycles
    out["achieved"] = all_valu * share / 1e9
    out["peak"] = MAC_ISSUE_PEAK / 1e9
    out["frac"] = all_valu * share / MAC_ISSUE_PEAK
    out["mac_share"] = mac_share
    out["issue_slots"] = {"achieved": all_valu / 1e9, "peak": mix_peak / 1e9, "unit": "G VALU wave-instructions/s", "frac": all_valu / mix_peak,
                          "peak_basis": f"{SIMDS} SIMDs x {NOMINAL_HZ / 1e9} GHz / (mac_share x {MAC_CYCLES} + (1 - mac_share) x {OTHER_CYCLES}) cycles"}
    out["frac_issue_slots"] = all_valu / mix_peak
    out["frac_at_measured_clock"] = (all_valu / (SIMDS * clock_mhz * 1e6 / mix_cycles)) if clock_mhz else None
    # above 1 the instruction model, the issue costs or — for the short kernels of a biprime step, where a 0.3 ms probe can
    # catch a clock the governor has already lowered — the clock samples are off: the raw value stays, flagged
    out["clock_sample_inconsistent"] = bool(out["frac_at_measured_clock"] is not None and out["frac_at_measured_clock"] > 1.0)
    out["frac_vs_guide_vector_peak"] = all_valu / GUIDE_VECTOR_PEAK
    out["guide_vector_peak"] = GUIDE_VECTOR_PEAK / 1e9
    out["note"] = ("frac = multiply-accumulates against the multiply issue rate at the NOMINAL clock (SURVEY 8d).  frac_issue_slots prices "
                   "the kernel's whole instruction mix at the measured issue costs; the chip sustains ~2.1-2.3 GHz under this load, so "
                   "frac_at_measured_clock is the fraction of the VALU issue slots the kernel fills.  frac_vs_guide_vector_peak is what a "
                   "stream of 2-cycle instructions could reach — no 32x32-bit multiply form issues at that rate (v_mul_lo_u32, "
                   "v_mad_u64_u32, v_mad_u32_u24 all 4.1-4.3)")
    return out


def hbm_block(alg_bytes: float, kernel_ms: float, traffic_model, traffic_key: str) -> dict:
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    measured = (_load_json(HBM_MEASURED) or {}).get(traffic_key)
    blk = {
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": alg_bytes,
        "traffic": measured["bytes"] if measured else None,
        "traffic_source": (measured["source"] if measured else
                           f"no rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass committed for '{traffic_key}' ({HBM_MEASURED.name})"),
        "traffic_model": traffic_model,
        "traffic_model_basis": "window-table accesses of the tape x 2 x limbs_per_lane words x lanes, plus the I/O rows "
                               "(an upper bound: L2 hits are not subtracted)",
    }
    t = blk["traffic"] or traffic_model
    if t:
        blk["traffic_over_algorithmic"] = t / alg_bytes
        blk["traffic_GBs"] = t / (kernel_ms * 1e-3) / 1e9
    return blk


# ---------------------------------------------------------------------------------------------------
# decryption workload (c3 / c5): partial decryption + share recombination
# ---------------------------------------------------------------------------------------------------
_LANE_STREAMS = []
_SMALL_STREAMS = []


def small_stream(torch, k: int):
    """High-priority streams for the small kernels of lane k (created once per process, like the lane streams)."""
    while len(_SMALL_STREAMS) <= k:
        _SMALL_STREAMS.append(torch.cuda.Stream(priority=-1))
    return _SMALL_STREAMS[k]



def lane_stream(torch, k: int, nstreams: int):
    """Streams of the steps in flight, created once per process and reused by every leg: the runtime maps
    a stream to a hardware queue, and streams created late in a process that has used many may share a
    queue with each other (two such streams serialise their launches: a c5 leg once ran at 26 k instead
    of 38 k modexps/s that way)."""
    if nstreams == 1:
        return torch.cuda.current_stream()
    while len(_LANE_STREAMS) <= k:
        _LANE_STREAMS.append(torch.cuda.Stream())
    return _LANE_STREAMS[k]


class DecryptWorkload:
    """Inputs of one party's decryption pass, resident on the device."""

    def __init__(self, eng, key_length: int, batch: int, rank: int, generic: bool):
        import torch

        from protocols.distributed_keygen_amd import limbs as L, synthetic

        self.eng, self.torch, self.L = eng, torch, L
        self.key = key = synthetic.make_key(key_length, 3, 1)
        self.n, self.n2 = key.n, key.n_square
        self.batch, self.generic = batch, generic
        self.parties = list(range(1, key.degree + 2))
        self.exps = {i: key.exponent(i) for i in self.parties}
        self.own = next((i for i in self.parties if self.exps[i] >= 0), self.parties[0])
        self.own_exp = abs(self.exps[self.own])
        self.own_slot = self.parties.index(self.own)
        self.limbs2 = L.limbs_for(self.n2)
        self.limbs = L.limbs_for(self.n)
        self.cts = synthetic.random_ciphertexts(key, batch, seed=synthetic.SEED + 17 * rank)
        self.c_t = eng.to_device(L.pack(self.cts, self.limbs2))
        # the other parties' partial decryptions, as they would arrive over the wire (untimed setup)
        self.partials_t = torch.empty((len(self.parties), batch, self.limbs2), dtype=torch.int32, device=eng.device)
        for k, i in enumerate(self.parties):
            src = self.c_t if self.exps[i] >= 0 else eng.modinv_t(self.c_t, self.n2)     # PSK:89-91
            eng.powmod_nsquare_t(src, self.n, abs(self.exps[i]), out_t=self.partials_t[k])
        self.own_in_t = self.c_t if self.exps[self.own] >= 0 else eng.modinv_t(self.c_t, self.n2)
        self.theta_inv = key.theta_inv
        self.lanes = []

    def make_lanes(self, nstreams: int, dist, world: int, cu_slices: bool = False) -> None:
        """One lane per step in flight (with `cu_slices` each lane's stream is confined to its own slice of the compute
        units, Engine.cu_slice_streams: for launches so small that several of them fit the chip side by side).  A lane has TWO sets of buffers: the recombination of
        step k reads set A while the modexp of step k+1 already writes set B (a caller that pipelines decryptions
        double-buffers the same way).  With one lane the recombination has a high-priority stream of its own."""
        torch, eng = self.torch, self.eng
        self.lanes = []
        for k in range(nstreams):
            bufs = []
            for b in range(2):
                bufs.append({
                    "partials": self.partials_t if (k == 0 and b == 0) else self.partials_t.clone(),
                    "msg": torch.empty((self.batch, self.limbs), dtype=torch.int32, device=eng.device),
                    "status": torch.zeros(self.batch, dtype=torch.uint8, device=eng.device),
                    "gathered": torch.empty((world, self.batch, self.limbs2), dtype=torch.int32, device=eng.device) if dist is not None else None,
                    "work": None, "done": None,
                })
            stream = eng.cu_slice_streams(nstreams)[k] if cu_slices else lane_stream(torch, k, nstreams)
            # Several lanes: the recombination runs on the lane's own stream — combine_kernel raises its wave priority
            # (csrc/mx_prio.hpp), and with that a high-priority companion per lane measured 0.9 % SLOWER on the headline
            # (304.1 vs 307.4 k/s, three alternating runs on one box) besides costing four hardware queues.  One lane: the
            # companion lets the recombination of step k overlap the launch of step k + 1.
            hp = small_stream(torch, k) if nstreams == 1 else stream
            self.lanes.append({"stream": stream, "hp": hp, "bufs": bufs, "turn": 0})
        torch.cuda.synchronize()

    def step(self, k: int, dist) -> None:
        ln = self.lanes[k % len(self.lanes)]
        buf = ln["bufs"][ln["turn"]]
        ln["turn"] ^= 1
        eng, torch = self.eng, self.torch
        with torch.cuda.stream(ln["stream"]):
            if buf["done"] is not None:          # the recombination that read this set two steps ago
                ln["stream"].wait_event(buf["done"])
            if buf["work"] is not None:          # the all-gather that read this set's rows
                buf["work"].wait()
                buf["work"] = None
            if self.generic:
                eng.powmod_shared_t(self.own_in_t, self.n2, self.own_exp, out_t=buf["partials"][self.own_slot])
            else:
                eng.powmod_nsquare_t(self.own_in_t, self.n, self.own_exp, out_t=buf["partials"][self.own_slot])
            if dist is not None:
                # the one exchange step of the path, on RCCL's own stream: the recombination needs only this rank's
                # rows, so it and the next step's launches run beside the gather (waited for when the set is reused)
                buf["work"] = dist.all_gather_into_tensor(buf["gathered"].view(-1), buf["partials"][self.own_slot].reshape(-1), async_op=True)
            ready = torch.cuda.Event()
            ready.record(ln["stream"])
        with torch.cuda.stream(ln["hp"]):
            ln["hp"].wait_event(ready)
            eng.combine_t(buf["partials"], self.n, self.theta_inv, out_t=buf["msg"], status_t=buf["status"])
            buf["done"] = torch.cuda.Event()
            buf["done"].record(ln["hp"])

    def verify(self, check: int, rank: int, dist) -> str:
        L, eng = self.L, self.eng
        for ln in self.lanes:
            for buf in ln["bufs"]:
                assert int(buf["status"].sum().item()) == 0, "share recombination flagged ciphertexts as inconsistent"
        if dist is not None:
            assert self.torch.equal(self.lanes[0]["bufs"][0]["gathered"][rank], self.partials_t[self.own_slot]), "all-gather shard mismatch"
        if check <= 0:
            return "skipped"
        # spot check with CPython big-int arithmetic (the definition of the reference's pow_mod /
        # PaillierSharedKey.decrypt, paillier_shared_key.py:92 and :115-125)
        batch, n, n2 = self.batch, self.n, self.n2
        idx = [0, batch - 1] + [(k * 7919) % batch for k in range(1, max(1, check - 1))]
        rows = eng.to_host(self.partials_t[self.own_slot][idx])
        msgs = L.unpack(eng.to_host(self.lanes[0]["bufs"][0]["msg"][idx]))
        allp = [L.unpack(eng.to_host(self.partials_t[k][idx])) for k in range(len(self.parties))]
        for j, e in enumerate(idx):
            base = self.cts[e] if self.exps[self.own] >= 0 else pow(self.cts[e], -1, n2)
            assert L.unpack(rows[j : j + 1])[0] == pow(base, self.own_exp, n2), f"partial decryption {e} differs from pow()"
            x = 1
            for k in range(len(self.parties)):
                x = x * allp[k][j] % n2
            assert (x - 1) % n == 0 and msgs[j] == (x - 1) // n * self.theta_inv % n, f"plaintext {e} differs"
        return f"{len(idx)} elements bit-exact vs CPython pow; all {batch} combines divisible by N"


def pick_decrypt_shape(eng, args, n_bits: int, batch: int, nstreams: int):
    """(limbs per lane, wavefronts per group) of the timed launches.  The library chooses the shape for ONE launch on
    an idle GPU; a caller that keeps `nstreams` launches in flight asks for the shape that suits their sum."""
    lpl = args.limbs_per_lane if args.limbs_per_lane >= 0 else 0
    wpg = args.wavefronts_per_group
    if args.generic_modulus:
        return (lpl or (18 if nstreams >= 3 else 0)), 0
    if lpl and wpg:
        return lpl, wpg
    eng.set_limbs_per_lane(lpl)
    eng.set_wavefronts_per_group(wpg)
    if nstreams >= 3:
        return eng.saturating_shape(n_bits, batch * nstreams)
    _, l, _, _, w = eng.nsquare_launch_shape(n_bits, batch * nstreams)
    return (lpl or l), (wpg or w)


def time_steps(eng, torch, dist, step_fn, steps: int, warmup: int, nstreams: int):
    """W warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier + device synchronisation;
    returns (elapsed seconds [max over ranks], mean modexp-kernel ms, kernel launches)."""

    def barrier() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # one untimed pass per lane allocates that lane's workspace, so that no allocation (a device
    # synchronisation) can fall into the timed region even when --warmup < steps in flight
    for k in range(nstreams):
        step_fn(k)
    barrier()
    for k in range(warmup):
        step_fn(k)
    barrier()
    # HIP events around the modexp kernel itself, recorded by the library on the stream it launches
    # on (mx_profile): the same interval rocprofv3 --kernel-trace reports for that kernel
    eng.profile(True)
    probes = []
    t0 = time.perf_counter()
    for k in range(steps):
        step_fn(k)
        if k % max(1, steps // 4) == 0 and k >= nstreams - 1:
            probes.append(eng.clock_probe_start(300))       # a one-wavefront probe beside the steps in flight
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile(False)
    kernel_total_ms, launches = eng.profile_collect()
    clocks = [eng.clock_probe_mhz(h) for h in probes]
    # (a few samples of 0.3 ms: good to a few per cent for the long launches of the decryption workloads; beside the short
    # kernels of a biprime step a sample can catch a clock the governor has already lowered — valu_roofline flags a
    # frac_at_measured_clock above 1 as clock_sample_inconsistent)
    CLOCK["mhz"] = sum(clocks) / len(clocks) if clocks else None
    RANK_TIMES["elapsed_s"] = [elapsed]
    if dist is not None:
        # every rank's own clock around the same barrier-to-barrier region: the job's time is the slowest rank's, and the
        # spread says whether a straggler (a slower GPU, a rank that shares its host cores) set it
        world = dist.get_world_size()
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        every = torch.zeros(world, dtype=torch.float64, device=eng.device)
        dist.all_gather_into_tensor(every, t)
        RANK_TIMES["elapsed_s"] = [float(x) for x in every.cpu().tolist()]
        elapsed = max(RANK_TIMES["elapsed_s"])
    return elapsed, (kernel_total_ms / launches if launches else 0.0), launches


CLOCK = {"mhz": None}       # shader clock measured during the last time_steps call
QUIET = {"group": None}     # gloo process group of an N > 1 run (main): barriers that block in a socket instead of spinning


def cpu_baseline_on_rank0(dist, rank: int, fn):
    """The CPU baseline of an N > 1 run: rank 0 times it AFTER the timed region, on all usable host
    cores, while the other ranks wait in a gloo barrier (no GPU work, no spinning host thread beside the timed processes)."""
    res = fn() if rank == 0 else None
    if dist is not None and QUIET["group"] is not None:
        dist.barrier(group=QUIET["group"])
    return res
RANK_TIMES = {"elapsed_s": []}      # every rank's elapsed time of the last time_steps call (rank order)


def per_rank_block(steps: int) -> dict:
    """ms_per_step of every rank of the last timed region (the line's ms_per_step is the max)."""
    ms = [e / steps * 1e3 for e in RANK_TIMES["elapsed_s"]]
    return {"min": min(ms), "max": max(ms), "ranks": [float(f"{m:.5g}") for m in ms]} if ms else None


def decrypt_roofline(eng, wl: DecryptWorkload, steps: int, elapsed: float, kernel_ms: float, nstreams: int, key_length: int) -> dict:
    n_bits, batch = wl.n.bit_length(), wl.batch
    if wl.generic:
        K, Lw, _, nblk = eng.geometry(wl.n2.bit_length(), batch, 1)
        name = f"mx::powmod_kernel<{K},{Lw},29,true>"
        roof = valu_roofline(name, None, steps, elapsed, kernel_ms, nstreams, clock_mhz=CLOCK["mhz"])
        roof["instructions_basis"] = "no model for the sliding-window generic kernel (not the product path)"
        traffic_model = None
    else:
        K, Lw, _, nblk, wv = eng.nsquare_launch_shape(n_bits, batch)
        name = f"mx::powmod_n2_kernel<{K},{Lw},29>" if wv == 1 else f"mx::powmod_n2_split_kernel<{K},{Lw},29>"
        plan = eng.nsquare_plan(wl.n, wl.own_exp).desc
        groups = -(-batch // (64 // K))                      # groups of elements = wavefronts (wv = 1) or wavefront pairs (wv = 2)
        if wv == 2:
            groups += groups & 1                             # two pairs per workgroup: an odd tail pair runs on a duplicate
        per_group = instr_per_wave("n2" if wv == 1 else "n2split", Lw, nblk, plan.n_sqr, plan.n_mul)
        # multiply-accumulates of the stream: a pair squaring is a symmetric pass (L/2+1 product + L
        # reduction MACs per limb step) + a full pass (L + L); a pair multiplication a full pass + a
        # two-row pass (2L + L); nblk * L limb steps per pass
        steps_per_pass = nblk * Lw
        macs = plan.n_sqr * steps_per_pass * ((Lw // 2 + 1 + Lw) + 2 * Lw) + plan.n_mul * steps_per_pass * (2 * Lw + 3 * Lw)
        roof = valu_roofline(name, None if per_group is None else per_group * groups, steps, elapsed, kernel_ms, nstreams,
                             mac_share=None if per_group is None else macs / per_group, clock_mhz=CLOCK["mhz"])
        if per_group is None:
            roof["instructions_basis"] = instr_model()[1] or f"no entry for this instance in {INSTR_MODEL.name}"
        roof["tape"] = {"pair_squarings": plan.n_sqr, "pair_multiplications": plan.n_mul, "window": plan.window}
        roof["wavefronts_per_group"] = wv
        traffic_model = groups * 64 * 4 * 2 * Lw * (plan.n_slot_reads + plan.n_slot_writes) + 2 * batch * 4 * wl.limbs2
    e_bits = wl.own_exp.bit_length()
    alg_bytes = batch * (2 * 4 * wl.limbs2) + 4 * wl.limbs2 + (e_bits + 7) // 8
    roof["traffic"] = None
    traffic_key = "generic" if wl.generic else f"n2_k{key_length}_b{batch}_L{Lw}" + ("x2" if roof.get("wavefronts_per_group") == 2 else "")
    roof["hbm"] = hbm_block(alg_bytes, kernel_ms, traffic_model, traffic_key)
    roof["traffic"] = roof["hbm"]["traffic"]
    roof["hbm"]["note"] = ("bytes of ONE launch over its own duration; the path is integer-VALU bound (north_star: no MFMA) and "
                           "the window table of odd powers lives in HBM on purpose: its traffic is ~1 % of the HBM roof")
    return roof


def run_decrypt_main(args, eng, torch, dist, rank: int, world: int, key_length: int, batch: int, label: str) -> dict:
    nstreams = args.streams if args.streams > 0 else next(
        (d for d in (4, 5, 6, 7, 8, 3) if args.steps % d == 0), min(4, max(1, args.steps)))
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    eng.set_segments(args.segments)
    wl = DecryptWorkload(eng, key_length, batch, rank, args.generic_modulus)     # untimed set-up with the library's own choices
    lpl, wpg = pick_decrypt_shape(eng, args, wl.n.bit_length(), batch, nstreams)
    eng.set_limbs_per_lane(lpl)
    eng.set_wavefronts_per_group(wpg)
    eng.set_priority_aux(False)                 # the lanes place their recombinations themselves (make_lanes)
    # Launches that fit a 1/nstreams slice of the chip at one wavefront per SIMD: the dispatcher would stack them on the
    # same CUs of every XCD, so every lane gets its own CUs (same range in every XCD) instead.
    cu_slices = False
    if not args.generic_modulus and 2 <= nstreams <= 8 and args.cu_slices != 0:
        k_, _, _, _, w_ = eng.nsquare_launch_shape(wl.n.bit_length(), batch)
        waves = -(-batch // (64 // k_)) * w_
        cu_slices = args.cu_slices == 1 or waves * nstreams <= 1024
    wl.make_lanes(nstreams, dist, world, cu_slices=cu_slices)
    elapsed, kernel_ms, launches = time_steps(eng, torch, dist, lambda k: wl.step(k, dist), args.steps, args.warmup, nstreams)
    assert launches == args.steps, (launches, args.steps)
    note = wl.verify(args.check if rank == 0 else 0, rank, dist)
    if rank != 0:
        return {}
    geo = eng.geometry(wl.n2.bit_length(), batch, 1) if args.generic_modulus else eng.nsquare_launch_shape(wl.n.bit_length(), batch)
    out = {
        "metric": "modexps/sec (2048-bit N, mod N^2)" if key_length == 2048 else f"modexps/sec ({key_length}-bit N, mod N^2)",
        "value": world * batch * args.steps / elapsed,
        "unit": "modexps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "distributed": dist_info(torch, dist, world, args.steps),
        "config": {
            "workload": f"{label}: 3-party key_length={key_length} t=1, {batch} ciphertexts/GPU/step: "
                        "partial-decrypt c^exp mod N^2 + share-combine",
            "batch_per_gpu": batch, "mod_bits": wl.n2.bit_length(), "exp_bits": wl.own_exp.bit_length(),
            "limbs_u32": wl.limbs2, "party": wl.own, "parallelism": f"dp{world}", "steps_in_flight": nstreams,
            "geometry_K_L_W_blocks": list(geo[:4]), "wavefronts_per_group": geo[4] if len(geo) > 4 else 1,
            "cu_slices": cu_slices,
            "algorithm": "Montgomery modulo N^2" if args.generic_modulus else "N-adic pairs, two Montgomery passes modulo N per product",
            "call_path": "per-key plan (mx_powmod_nsquare_prepare once) + mx_powmod_nsquare_run + mx_combine_run per step",
            "segments_per_exponentiation": args.segments or "library default (4 for long exponents on large batches)",
            "verified": note,
        },
        "roofline": decrypt_roofline(eng, wl, args.steps, elapsed, kernel_ms, nstreams, key_length),
    }
    out["_wl"] = wl
    return out


def restores_launch_shape(fn):
    """A leg that changes the engine's launch shape (limbs per lane, wavefronts per group) leaves it as it found it,
    also when it raises: main()'s guarded() swallows a leg's exception, and every later leg would silently run with the
    wrong shape and report numbers that no longer mean what their labels say."""
    import functools

    @functools.wraps(fn)
    def wrapper(eng, *a, **k):
        saved = (eng._lpl, eng._wpg)
        try:
            return fn(eng, *a, **k)
        finally:
            eng.set_limbs_per_lane(saved[0])
            eng.set_wavefronts_per_group(saved[1])

    return wrapper


@restores_launch_shape
def leg_single_batch(eng, torch, wl: DecryptWorkload, key_length: int) -> dict:
    """The same step with ONE launch in flight (what a caller without its own streams gets): the library picks the
    launch shape for a lone launch of this size."""
    saved = (eng._lpl, eng._wpg)
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    wl.make_lanes(1, None, 1)
    steps = 6
    elapsed, kernel_ms, _ = time_steps(eng, torch, None, lambda k: wl.step(k, None), steps, 1, 1)
    geo = eng.nsquare_launch_shape(wl.n.bit_length(), wl.batch)
    sliced = eng.nsquare_launch_timesliced(wl.n.bit_length(), wl.batch)
    eng.set_limbs_per_lane(saved[0])
    eng.set_wavefronts_per_group(saved[1])
    waves = -(-wl.batch // (64 // geo[0])) * geo[4]
    return {"value": wl.batch * steps / elapsed, "unit": "modexps/s", "steps": steps, "ms_per_step": elapsed / steps * 1e3,
            "kernel_ms": kernel_ms, "geometry_K_L_W_blocks": list(geo[:4]), "wavefronts_per_group": geo[4],
            "time_sliced": {"resident_workgroups_per_cu": sliced[0], "units_per_group": sliced[1]} if sliced[0] else None,
            "shader_clock_mhz_measured": CLOCK["mhz"],
            "note": f"one {wl.batch}-ciphertext launch at a time: {waves} wavefronts for 1024 SIMDs"
                    + (f", time-sliced over {sliced[0]} resident workgroups per CU" if sliced[0] else "")}


@restores_launch_shape
def leg_latency(eng, torch, wl: DecryptWorkload, single_core_rate) -> dict:
    """The lone call the reference's API produces (DistributedPaillier.decrypt of ONE ciphertext,
    distributed_keygen.py:345-349 -> paillier_shared_key.py:92): Python int in, Python int out through
    GpuPaillierSharedKey.partial_decrypt, beside one gmpy2.powmod on one host core.  Also a keygen-sized batch."""
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    key = wl.key
    gk = GpuPaillierSharedKey(key.n, key.t, wl.own, ShareView(dict(key.shares), key.degree, key.n_fac), key.theta, engine=eng)
    saved = (eng._lpl, eng._wpg)
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    out = {"unit": "ms per call, Python ints to Python ints (pack, H2D, modexp, D2H, unpack)"}
    for count in (1, 64, 1024):
        times = []
        for rep in range(5):
            cts = [PlainCiphertext(c, key.n) for c in wl.cts[:count]]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = gk.partial_decrypt_batch(cts) if count > 1 else [gk.partial_decrypt(cts[0])]
            times.append(time.perf_counter() - t0)
        base = wl.cts[0] if wl.exps[wl.own] >= 0 else pow(wl.cts[0], -1, wl.n2)
        assert got[0] == pow(base, wl.own_exp, wl.n2)
        times.sort()
        geo = eng.nsquare_launch_shape(wl.n.bit_length(), count)
        out[f"n{count}"] = {"ms": times[len(times) // 2] * 1e3, "best_ms": times[0] * 1e3, "geometry_K_L_W_blocks": list(geo[:4]),
                            "wavefronts_per_group": geo[4], "ciphertexts_per_s": count / times[len(times) // 2]}
    # the same lone call by a party whose Lagrange exponent is negative (two of three parties at t = 1): PSK:89-91 inverts
    # the ciphertext modulo N^2 first — one more kernel, a host look at its status byte, then the exponentiation
    neg = next((i for i in wl.parties if wl.exps[i] < 0), None)
    if neg is not None:
        gk2 = GpuPaillierSharedKey(key.n, key.t, neg, ShareView(dict(key.shares), key.degree, key.n_fac), key.theta, engine=eng)
        ct = PlainCiphertext(wl.cts[0], key.n)
        times = []
        for rep in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got2 = gk2.partial_decrypt(ct)
            times.append(time.perf_counter() - t0)
        assert got2 == pow(pow(wl.cts[0], -1, wl.n2), -wl.exps[neg], wl.n2)
        times = sorted(times[1:])
        out["n1_negative_exponent"] = {"ms": times[len(times) // 2] * 1e3, "best_ms": times[0] * 1e3, "party": neg}
    eng.set_limbs_per_lane(saved[0])
    eng.set_wavefronts_per_group(saved[1])
    out["value"] = out["n1"]["ms"]
    if single_core_rate:
        out["gmpy2_one_core_ms"] = 1e3 / single_core_rate
        out["vs_gmpy2_one_core"] = (1e3 / single_core_rate) / out["n1"]["ms"]
        out["note"] = ("vs_gmpy2_one_core > 1 means the GPU answers a single decrypt() faster than the reference's own "
                       "scalar path; below 1 install(scalars=False) keeps the reference's path for lone calls")
    return out


@restores_launch_shape
def leg_end_to_end(eng, torch, wl: DecryptWorkload, tensor_rate: float) -> dict:
    """Python ints in -> Python ints out through the drop-in classes: GpuPaillierSharedKey
    .partial_decrypt_batch (pack, H2D, modexp, D2H, unpack) and .decrypt_batch (the same around the
    recombination; `.decrypt_columns` is the form the patched _decrypt_sequence_raw uses: the own partials
    stay on the device, received ones arrive as per-player lists) — the loops distributed_keygen.py:463-466
    and :510-515 as the patch runs them, for one batch and for a 4x longer sequence (which the engine cuts
    into chunks on several streams)."""
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    key, L = wl.key, wl.L
    share = ShareView(dict(key.shares), key.degree, key.n_fac)
    gk = GpuPaillierSharedKey(key.n, key.t, wl.own, share, key.theta, engine=eng)
    saved = (eng._lpl, eng._wpg)
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    others1 = {i: L.unpack(eng.to_host(wl.partials_t[k])) for k, i in enumerate(wl.parties) if i != wl.own}
    out = {"unit": "ciphertexts/s, Python ints to Python ints",
           "note": "one call each on the default stream; includes Python int <-> limb rows (C codec), PCIe both ways, "
                   "ciphertext.get_value() per element and the reference's type/key checks"}
    for mult in (1, 4):
        count = wl.batch * mult
        others = {i: v * mult for i, v in others1.items()}
        runs = []
        for rep in range(4):                                # the first pass warms allocations and pinned buffers
            cts = [PlainCiphertext(c, key.n) for c in wl.cts * mult]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            partials, own_column = gk.partial_decrypt_batch(cts, keep_rows=True)
            t1 = time.perf_counter()
            tm = eng.last_timing
            dicts = [{wl.own: p, **{i: others[i][k] for i in others}} for k, p in enumerate(partials)]
            t2 = time.perf_counter()
            msgs = gk.decrypt_batch(dicts)                  # per-ciphertext dictionaries, as the reference's loop has them
            t3 = time.perf_counter()
            msgs_c = gk.decrypt_columns({wl.own: own_column, **others}, count)   # per-player columns, as the patch has them
            t4 = time.perf_counter()
            assert msgs_c == msgs
            if rep:
                runs.append({"partial_decrypt_s": t1 - t0, "combine_s": t3 - t2, "columns_s": t4 - t3, "tm": tm})
        # median of the three timed calls (a call that finds the GPU in a low power state after the host-only
        # phase in between pays a wake-up of tens of milliseconds; all three are listed)
        runs.sort(key=lambda r: r["partial_decrypt_s"])
        tm = runs[1]["tm"]
        res = {f: sorted(r[f] for r in runs)[1] for f in ("partial_decrypt_s", "combine_s", "columns_s")}       # every field its own median
        spread = [round(count / r["partial_decrypt_s"]) for r in runs]
        k = 7
        x = 1
        for i in wl.parties:
            x = x * (partials[k] if i == wl.own else others[i][k]) % wl.n2
        assert msgs[k] == (x - 1) // wl.n * wl.theta_inv % wl.n and len(msgs) == count
        out[f"n{count}"] = {
            "partial_decrypt_rate": count / res["partial_decrypt_s"],
            "partial_decrypt_rate_of_each_call": spread,
            "partial_decrypt_vs_tensor_level": count / res["partial_decrypt_s"] / tensor_rate,
            "combine_rate": count / res["combine_s"],
            "combine_columns_rate": count / res["columns_s"],
            "both_rate": count / (res["partial_decrypt_s"] + res["columns_s"]),
            "partial_decrypt_breakdown": {k2: (round(v, 5) if isinstance(v, float) else v) for k2, v in (tm or {}).items()},
        }
    out["value"] = out[f"n{wl.batch * 4}"]["partial_decrypt_rate"]
    eng.set_limbs_per_lane(saved[0])
    eng.set_wavefronts_per_group(saved[1])
    return out


# ---------------------------------------------------------------------------------------------------
# biprimality-test workload (configs[3]; configs[1] with --key-length 1024)
# ---------------------------------------------------------------------------------------------------
class BiprimeWorkload:
    """`cands` candidate moduli of the distributed_keygen.py:855-876 shape that pass the small-prime
    sieve, 160 jointly random generators each, this party's exponent per candidate; all on the device."""

    GENS, KEEP = 160, 40          # 4 x correct_param_biprime generators (DK:1028), 40 tests (DK:1086)

    def __init__(self, eng, key_length: int, n_parties: int, cands: int, seed: int, index: int = 1):
        import sympy
        import torch

        from protocols.distributed_keygen_amd import limbs as L, synthetic

        self.eng, self.torch, self.L = eng, torch, L
        rng = random.Random(seed)
        half = key_length // 2
        self.primes = [int(p) for p in sympy.primerange(3, 2001)]          # prime_threshold 2000 (DK:85)
        self.shares, self.mods = [], []
        self.sieved = 0
        while len(self.mods) < cands:                                     # survivors of DK:1288-1292
            cand = [synthetic.candidate_shares(rng, n_parties, half) for _ in range(max(256, cands * 4))]
            cm = [sum(p) * sum(q) for p, q in cand]
            bad = eng.sieve_batch(cm, self.primes)
            self.sieved += len(cm)
            for sh, m, b in zip(cand, cm, bad):
                if not b and len(self.mods) < cands:
                    self.shares.append(sh)
                    self.mods.append(m)
        self.cands, self.n_parties, self.index = cands, n_parties, index
        self.mod_bits = max(m.bit_length() for m in self.mods)
        self.limbs = L.limbs_for_bits(self.mod_bits)
        self.exps_by_party = {
            i: [((m - p[0] - q[0] + 1) // 4) if i == 1 else ((p[i - 1] + q[i - 1]) // 4) for m, (p, q) in zip(self.mods, self.shares)]
            for i in range(1, n_parties + 1)
        }
        self.exps = self.exps_by_party[index]
        self.exp_bits = max(e.bit_length() for e in self.exps)
        nb = (self.mod_bits + 7) // 8 + 8
        g_all = [int.from_bytes(rng.randbytes(nb), "little") % m for m in self.mods for _ in range(self.GENS)]
        self.g_sample = g_all[: self.GENS]
        self.g_t = eng.to_device(L.pack(g_all, self.limbs))
        self.mods_op = (eng.to_device(L.pack(self.mods, self.limbs)), self.mod_bits)
        self.exps_op = (eng.to_device(L.pack(self.exps, L.limbs_for_bits(self.exp_bits))), self.exp_bits)
        # the other parties' v rows, as they would arrive (untimed): same generators, their exponents
        self.v_all = torch.zeros((n_parties, cands, self.KEEP, self.limbs), dtype=torch.int32, device=eng.device)
        for i in range(1, n_parties + 1):
            ex = self.exps_by_party[i]
            ex_op = (eng.to_device(L.pack(ex, L.limbs_for_bits(max(e.bit_length() for e in ex)))), max(e.bit_length() for e in ex))
            v_t, cnt_t = eng.biprime_v_t(self.g_t, self.mods_op, ex_op, self.GENS, self.KEEP)
            self.v_all[i - 1] = v_t.view(cands, self.KEEP, self.limbs)
            if i == index:
                self.counts = cnt_t.clone()
        self.lanes = []

    def make_lanes(self, nstreams: int, dist, world: int) -> None:
        torch, eng = self.torch, self.eng
        self.lanes = []
        for k in range(nstreams):
            self.lanes.append({
                "stream": lane_stream(torch, k, nstreams),
                "v_all": self.v_all if k == 0 else self.v_all.clone(),
                "verdict": torch.empty((self.cands, self.KEEP), dtype=torch.uint8, device=eng.device),
                "v_gather": torch.empty((world, self.cands * self.KEEP, self.limbs), dtype=torch.int32, device=eng.device) if dist is not None else None,
                "vote_gather": torch.empty((world, self.cands, self.KEEP), dtype=torch.uint8, device=eng.device) if dist is not None else None,
            })
        torch.cuda.synchronize()

    def step(self, k: int, dist) -> None:
        ln = self.lanes[k % len(self.lanes)]
        eng = self.eng
        with self.torch.cuda.stream(ln["stream"]):
            for w in ln.pop("works", []):        # the gathers of this lane's previous step must have read its buffers
                w.wait()
            # DK:1084-1099 over this rank's candidates: Jacobi filter -> first 40 -> 40 modexps each
            v_t, _ = eng.biprime_v_t(self.g_t, self.mods_op, self.exps_op, self.GENS, self.KEEP)
            ln["v_all"][self.index - 1].view(-1, self.limbs).copy_(v_t)
            works = []
            if dist is not None:                                          # v rows to every rank (DK:1331-1337), beside the verdict
                ln["v_keep"] = v_t
                works.append(dist.all_gather_into_tensor(ln["v_gather"].view(-1), v_t.reshape(-1), async_op=True))
            # DK:1147-1158 for every (candidate, test slot) of this rank's candidates
            eng.biprime_verdict_t(ln["v_all"], self.mods_op, pass_t=ln["verdict"])
            if dist is not None:                                          # the vote: verdict bytes of all ranks
                works.append(dist.all_gather_into_tensor(ln["vote_gather"].view(-1), ln["verdict"].reshape(-1), async_op=True))
            ln["works"] = works

    def verify(self, check: int) -> str:
        import sympy

        L, eng = self.L, self.eng
        v_rows = L.unpack(eng.to_host(self.lanes[0]["v_all"][self.index - 1][0]))
        m, e = self.mods[0], self.exps[0]
        keep = [g for g in self.g_sample if sympy.jacobi_symbol(g, m) == 1][: self.KEEP]
        assert v_rows[: len(keep)] == [pow(g, e, m) for g in keep], "v values differ from pow()"
        assert int(self.counts[0].item()) == len(keep)
        # composites fail a slot almost surely; the verdict bytes must equal the host computation for candidate 0
        want = []
        allv = [L.unpack(eng.to_host(self.lanes[0]["v_all"][i][0])) for i in range(self.n_parties)]
        for s in range(self.KEEP):
            prod = 1
            for i in range(1, self.n_parties):
                prod *= allv[i][s]
            want.append(1 if (allv[0][s] % m == prod % m or allv[0][s] % m == (-prod) % m) else 0)
        got = [int(x) for x in self.lanes[0]["verdict"][0].cpu().numpy()]
        assert got == want, "verdict bytes differ from DK:1147-1158 on the host"
        return f"candidate 0: {len(keep)} v values bit-exact vs CPython pow, Jacobi selection vs sympy, {self.KEEP} verdict bytes vs host"


def biprime_roofline(eng, wl: BiprimeWorkload, steps: int, elapsed: float, kernel_ms: float, nstreams: int) -> dict:
    from tools.calibrate_instr import generic_counts

    batch = wl.cands * wl.KEEP
    K, Lw, _, nblk = eng.geometry(wl.mod_bits, batch, wl.cands)
    elimbs = wl.exps_op[0].shape[1]
    n_sqr, n_mul = generic_counts(wl.exp_bits, elimbs)
    per_wave = instr_per_wave("generic", Lw, nblk, n_sqr, n_mul)
    nwaves = -(-batch // (64 // K))
    steps_per_pass = nblk * Lw
    macs = n_sqr * steps_per_pass * (Lw // 2 + 1 + Lw) + n_mul * steps_per_pass * 2 * Lw
    roof = valu_roofline(f"mx::powmod_kernel<{K},{Lw},29,false>", None if per_wave is None else per_wave * nwaves, steps,
                         elapsed, kernel_ms, nstreams, mac_share=None if per_wave is None else macs / per_wave, clock_mhz=CLOCK["mhz"])
    if per_wave is None:
        roof["instructions_basis"] = instr_model()[1] or f"no entry for this instance in {INSTR_MODEL.name}"
    roof["exponentiation"] = {"squarings": n_sqr, "multiplications": n_mul, "fixed_window": True}
    s = wl.limbs
    alg_bytes = batch * 2 * 4 * s + wl.cands * 4 * (s + elimbs)
    # window table: 2^win entries written once, one entry read per exponent digit (= n_mul + 3 accesses), L words per lane
    traffic_model = nwaves * 64 * 4 * Lw * (n_mul + 3) + 2 * batch * 4 * s
    roof["hbm"] = hbm_block(alg_bytes, kernel_ms, traffic_model, f"biprime_b{wl.mod_bits}_c{wl.cands}_L{Lw}")
    roof["traffic"] = roof["hbm"]["traffic"]
    return roof


MAX_LANES = 8


def biprime_lanes(cands_per_gpu: int, steps: int) -> int:
    """Steps kept in flight for the biprimality workload: two for launches that fill the machine on their own, four for
    the small shards (a rank of an 8-GPU run gets 512 candidates = 2560 wavefronts for 1024 SIMDs: two in flight leave the
    machine waiting on the short kernels of a step), eight for a keygen round's worth of candidates (256 at key_length
    1024 = a 2.4 ms kernel behind ~1 ms of dependent short kernels).  Measured with the short kernels at raised wave
    priority (csrc/mx_prio.hpp), profiles/r04_biprime_lanes_queues.txt: key_length 1024 x 256 candidates 5.6 / 7.3 / 7.4 M
    modexps/s with 4 / 8 / 12 in flight, key_length 2048 x 100 candidates 0.64 / 1.08 / 1.19 M/s; 512 candidates 1.28 M/s
    with 4 or more.  The host enqueues a step in 0.2 ms: not launch-bound, HIP graphs buy nothing
    (profiles/r04_graph_probe.txt).
    Not more than MAX_LANES: every stream a process has used owns a hardware queue for good, and a process that has
    used more than ~24 of them is time-sliced by the queue scheduler from then on — twelve lanes for this leg were
    0.1 M/s faster on their own and cost every LATER leg of the same process 30-45 % (key_length 4096: 39 -> 23 k/s;
    profiles/r04_bench_queue_budget.txt)."""
    for want in ((8, 6, 4, 2) if cands_per_gpu <= 256 else (4, 2) if cands_per_gpu <= 1024 else (2,)):
        if steps % want == 0 and want <= MAX_LANES:
            return want
    return 1


def priority_aux_for(nstreams: int) -> bool:
    """Whether the short kernels of a step go to a high-priority companion of the lane's stream (Engine.set_priority_aux,
    DecryptWorkload.make_lanes): for up to four lanes.  Beyond that the companions buy nothing now that the short kernels
    raise their own wave priority (profiles/r04_biprime_lanes_queues.txt) and would be eight more hardware queues."""
    return 1 < nstreams <= 4


def dist_info(torch, dist, world: int, steps: int = 0):
    """What the process group actually is: world size as the backend reports it, the RCCL version, and every rank's own
    ms_per_step of the timed region just measured (straggler visibility)."""
    if dist is None:
        return None
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_expected": world}
    if steps:
        info["ms_per_step_per_rank"] = per_rank_block(steps)
    try:
        info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # pragma: no cover
        info["rccl_version"] = None
    info["exchange"] = "all_gather_into_tensor, async on RCCL's stream, overlapped with the recombination / verdict and the next step's launches"
    return info


def run_biprime(args, eng, torch, dist, rank: int, world: int, key_length: int, total_cands: int, steps: int, warmup: int,
                nstreams: int, n_parties: int = 5, weak: bool = False) -> dict:
    cands = total_cands if weak else -(-total_cands // world)
    total_cands = cands * world if weak else total_cands
    eng.set_limbs_per_lane(args.limbs_per_lane if args.limbs_per_lane >= 0 else 0)
    wl = BiprimeWorkload(eng, key_length, n_parties, cands, seed=0xD15C0 + 3 + 101 * rank)
    if args.limbs_per_lane <= 0 and nstreams > 1:
        # the library picks the lane geometry for ONE launch on an idle GPU; with `nstreams` steps in flight the launches
        # fill the machine between them, so the shape is the one that suits their sum (as pick_decrypt_shape does for c3)
        eng.set_limbs_per_lane(eng.geometry(wl.mod_bits, cands * wl.KEEP * nstreams, cands * nstreams)[1])
    eng.set_priority_aux(priority_aux_for(nstreams) if args.priority_aux < 0 else bool(args.priority_aux))          # Jacobi filter, selection and verdict do not queue behind the other lane's modexps
    wl.make_lanes(nstreams, dist, world)
    elapsed, kernel_ms, launches = time_steps(eng, torch, dist, lambda k: wl.step(k, dist), steps, warmup, nstreams)
    assert launches == steps
    note = wl.verify(args.check) if rank == 0 else ""
    if dist is not None:
        assert torch.equal(wl.lanes[0]["vote_gather"][rank], wl.lanes[0]["verdict"]), "vote all-gather shard mismatch"
    if rank != 0:
        return {}
    geo = eng.geometry(wl.mod_bits, cands * wl.KEEP, cands)
    # the stages on their own (one launch each, untimed region above excluded)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    j_t = eng.jacobi_t(wl.g_t, wl.mods_op, wl.GENS)
    torch.cuda.synchronize()
    jac_s = time.perf_counter() - t0
    big = eng.to_device(wl.L.pack((wl.mods * (65536 // len(wl.mods) + 1))[:65536], wl.limbs))
    eng.sieve_t(big, wl.primes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sieve_t(big, wl.primes)
    torch.cuda.synchronize()
    sieve_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.biprime_verdict_t(wl.lanes[0]["v_all"], wl.mods_op)
    torch.cuda.synchronize()
    verdict_s = time.perf_counter() - t0
    del j_t, big
    return {
        "metric": f"biprimality-test modexps/sec ({key_length}-bit N)",
        "value": world * cands * wl.KEEP * steps / elapsed,
        "unit": "modexps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "distributed": dist_info(torch, dist, world, steps),
        "config": {
            "workload": f"C4: {n_parties}-party key_length={key_length} t=2, {total_cands} sieve-surviving candidate moduli "
                        f"sharded over {world} GPU(s) ({cands}/GPU/step): 160 Jacobi symbols + first-40 selection + 40 modexps "
                        "g^e mod N per candidate (party 1's exponent (N-p1-q1+1)/4), verdict of all parties' v values"
                        + ("; all-gather of v rows and verdict bytes" if dist is not None else ""),
            "candidates_per_gpu": cands, "mod_bits": wl.mod_bits, "exp_bits": wl.exp_bits, "limbs_u32": wl.limbs,
            "parallelism": f"dp{world}", "steps_in_flight": nstreams, "geometry_K_L_W_blocks": list(geo),
            "verified": note,
        },
        "stages": {
            "jacobi_symbols_per_s": cands * wl.GENS / jac_s, "jacobi_ms": jac_s * 1e3,
            "sieve_candidates_per_s": 65536 / sieve_s, "sieve_ms_65536": sieve_s * 1e3, "sieve_primes": len(wl.primes),
            "verdict_slots_per_s": cands * wl.KEEP / verdict_s, "verdict_ms": verdict_s * 1e3,
            "sieve_survival": cands / wl.sieved,
        },
        "roofline": biprime_roofline(eng, wl, steps, elapsed, kernel_ms, nstreams),
        "_wl": wl,
    }


# ---------------------------------------------------------------------------------------------------
# one key-generation round, Python ints in -> verdicts out (distributed_keygen.py:1284-1360)
# ---------------------------------------------------------------------------------------------------
@restores_launch_shape
def leg_keygen_round(eng, torch, args, key_length: int = 2048, n_parties: int = 5, t: int = 2,
                     batch_sizes=(100, 1024, 16384, 65536)) -> dict:
    """What patch.compute_modulus does per round (biprime.BiprimeRound), timed from Python ints to Python verdicts for
    `batch_size` candidates (100 = the reference's default, distributed_keygen.py:102: ~2 survivors, whose 80 modexps run the
    bipartite latency form): every party's Shamir shares of the candidate moduli -> reconstruct + sieve (one device
    pass; the survivors' moduli stay on the device) -> survivors' v values (Jacobi filter, selection, 40 modexps each;
    this party's rows stay on the device) -> verdicts (the peers' columns packed with one codec call per party).  The exchange rounds in between (shares,
    jointly random generators, the other parties' v values) are network traffic in the reference and are prepared
    untimed.  Beside it: the same four steps as the reference computes them (oracle/cpu_keygen_round.py, one core —
    the reference runs a round sequentially on its asyncio thread), timed on a sample and scaled to the round."""
    import sympy

    from protocols.distributed_keygen_amd import biprime, shamir, synthetic

    # launch shapes: the library's own choice per launch, as a caller of patch.install() gets them (the headline leg leaves
    # its explicit 18-limb shape on the engine: a small round then ran the WIDE generic kernel, 12.7 instead of 4.3 ms for
    # its few dozen modexps — rounds 3 and 4 reported b1024 that way)
    saved_shape = (eng._lpl, eng._wpg)
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    rng = random.Random(0xD15C0 + 77)
    half = key_length // 2
    degree = 2 * t
    prime = synthetic.random_prime(rng, 2 * (half + 4) + 44, mod4=1)        # the Shamir field: larger than any candidate modulus
    prime_list = [int(q) for q in sympy.primerange(3, 2001)]
    total = max(batch_sizes)
    shares = [synthetic.candidate_shares(rng, n_parties, half) for _ in range(total)]
    mods = [sum(p) * sum(q) for p, q in shares]
    points = list(range(1, n_parties + 1))
    columns = {i: [] for i in points}
    for m in mods:                                                           # degree-2t sharing of every modulus (DK:1274-1281)
        coeffs = [m] + [rng.getrandbits(prime.bit_length() + 8) % prime for _ in range(degree)]
        for i in points:
            acc = 0
            for c in reversed(coeffs):
                acc = (acc * i + c) % prime
            columns[i].append(acc)
    out = {"unit": "candidate moduli per second through one round (Python ints -> verdicts), one GPU",
           "config": f"{n_parties}-party key_length={key_length} t={t}: Shamir field of {prime.bit_length()} bits, {len(prime_list)} sieve primes, "
                     "160 generators and 40 test slots per surviving candidate",
           "rounds": {}}
    sample_job = None
    for B in batch_sizes:
        by_party = {i: columns[i][:B] for i in points}
        best = None
        for rep in range(2 if B > 1024 else 4):                              # the first pass warms allocations (small rounds: a few more)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rnd = biprime.BiprimeRound(eng)                                      # what patch.compute_modulus runs per round
            surviving = rnd.reconstruct_and_sieve(by_party, prime, degree, prime_list, points=points)
            has_div = rnd.has_divisor
            t1 = time.perf_counter()
            surv = sorted(surviving)
            moduli = [surviving[k] for k in surv]
            assert all(moduli[j] == mods[k] for j, k in enumerate(surv))
            g_rng = random.Random(B)
            g_values = [[g_rng.getrandbits(key_length + 64) % m for _ in range(160)] for m in moduli]      # untimed: a communication round
            t2 = time.perf_counter()
            v1 = rnd.v_calculation(g_values, 1, [shares[k][0][0] for k in surv], [shares[k][1][0] for k in surv], 40)
            t3 = time.perf_counter()
            v_by = [{1: v} for v in v1]                                          # untimed: the other parties' v values arrive
            for i in range(2, n_parties + 1):
                vi = biprime.biprime_test_v_calculation_batch(g_values, i, moduli, [shares[k][0][i - 1] for k in surv],
                                                              [shares[k][1][i - 1] for k in surv], 40, eng)
                for d, v in zip(v_by, vi):
                    d[i] = v
            t4 = time.perf_counter()
            verdicts = rnd.verdicts(v_by, 40, errors="return")
            t5 = time.perf_counter()
            cur = {"reconstruct_sieve_s": t1 - t0, "v_calculation_s": t3 - t2, "verdict_s": t5 - t4}
            cur["total_s"] = sum(cur.values())
            if best is None or cur["total_s"] < best["total_s"]:
                best = cur
        if not surv:                                                          # a small round may leave no survivor
            best.update({"batch_size": B, "survivors": 0, "candidates_per_s": B / best["total_s"], "modexps": 0, "biprimes_found": 0})
            out["rounds"][f"b{B}"] = best
            continue
        # spot checks against the definitions (CPython)
        k0 = surv[0]
        assert has_div[k0] is False and sum(1 for b in has_div if not b) == len(surv)
        assert bool(has_div[1]) == any(mods[1] % q == 0 for q in prime_list)
        e0 = (moduli[0] - shares[k0][0][0] - shares[k0][1][0] + 1) // 4
        keep = [g for g in g_values[0] if sympy.jacobi_symbol(g, moduli[0]) == 1][:40]
        assert v1[0] == [pow(g, e0, moduli[0]) for g in keep]
        want = len(v1[0]) >= 40                                               # DK:1147-1172 on the host for survivor 0
        for slot in range(min(40, len(v1[0]))):
            prod = 1
            for i in range(2, n_parties + 1):
                prod *= v_by[0][i][slot]
            if v1[0][slot] % moduli[0] not in (prod % moduli[0], (-prod) % moduli[0]):
                want = False
                break
        assert verdicts[0] is want
        best.update({"batch_size": B, "survivors": len(surv), "candidates_per_s": B / best["total_s"],
                     "modexps": 40 * len(surv), "biprimes_found": sum(1 for v in verdicts if v is True)})
        out["rounds"][f"b{B}"] = best
        if sample_job is None and B >= 1024:
            ns, nc = 3, 256
            sample_job = {"prime": hex(prime), "points": points, "columns": {str(i): [hex(v) for v in columns[i][:nc]] for i in points},
                          "prime_list": prime_list, "moduli_check": [hex(m) for m in mods[:8]],
                          "survivors": [{"modulus": hex(moduli[j]), "exponent": hex((moduli[j] - shares[surv[j]][0][0] - shares[surv[j]][1][0] + 1) // 4),
                                         "g": [hex(g) for g in g_values[j]], "v_check": [hex(v) for v in v1[j]],
                                         "v_others": [[hex(v) for v in v_by[j][i]] for i in range(2, n_parties + 1)]} for j in range(min(ns, len(surv)))]}
    out["value"] = out["rounds"][f"b{max(batch_sizes)}"]["candidates_per_s"]
    if not args.no_cpu_baseline and sample_job is not None:
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump(sample_job, f)
            path = f.name
        script = str(ROOT / "oracle" / "cpu_keygen_round.py")
        res = None
        for py in ("/opt/conda/bin/python3.9", sys.executable):
            if os.path.exists(py):
                try:
                    r = subprocess.run([py, script, path], capture_output=True, text=True, timeout=75)
                    if r.returncode == 0 and r.stdout.strip():
                        res = json.loads(r.stdout.strip().splitlines()[-1])
                        break
                except Exception:  # pragma: no cover - measurement plumbing
                    pass
        os.unlink(path)
        if res is not None:
            base = {"kind": "port", "cores": 1, "engine": res["engine"], "unit": "seconds per unit, one core",
                    "per_unit": {k: res[k] for k in res if k.endswith("_candidate") or k.endswith("_survivor")},
                    "sample": f"{res['sample_candidates']} candidates (reconstruct, sieve) and {res['sample_survivors']} survivors (v values, verdict) "
                              "of the first round, one core: the reference runs a round sequentially on its asyncio thread; scaled to each round size",
                    "rounds": {}}
            for name, rd in out["rounds"].items():
                cpu_s = rd["batch_size"] * (res["reconstruct_s_per_candidate"] + res["sieve_s_per_candidate"]) + rd["survivors"] * (
                    res["v_calculation_s_per_survivor"] + res["verdict_s_per_survivor"])
                base["rounds"][name] = {"cpu_round_s": cpu_s, "candidates_per_s": rd["batch_size"] / cpu_s, "gpu_speedup": cpu_s / rd["total_s"]}
            out["cpu_baseline"] = base
    eng.set_limbs_per_lane(saved_shape[0])
    eng.set_wavefronts_per_group(saved_shape[1])
    return out



# ---------------------------------------------------------------------------------------------------
# The short kernels of the path: rate and issue fraction, each at the shape it has in a full-size
# step.  Instruction counts per launch come from a counter pass of THIS function (tools/short_kernels.py:
# rocprofv3 --pmc SQ_INSTS_VALU), committed as profiles/r05_short_kernels.json; the durations are measured live.
# ---------------------------------------------------------------------------------------------------
SHORT_KERNELS = _latest_profile("short_kernels.json")
PLAIN_ISSUE_PEAK = SIMDS * NOMINAL_HZ / OTHER_CYCLES


def short_kernel_cases(eng, torch):
    """([(name, kernel-name substring, units, unit, callable)], operands to keep alive); every operand is resident on
    the device before the first call."""
    import sympy

    from protocols.distributed_keygen_amd import limbs as L

    bp = BiprimeWorkload(eng, 2048, 5, 4096, seed=0xD15C0 + 3)
    rng = random.Random(0x5EED)
    limbs = bp.limbs
    primes = [int(q) for q in sympy.primerange(3, 2001)]            # prime_threshold 2000 (DK:85)
    cands = [(rng.getrandbits(2053) | (1 << 2052) | 1) for _ in range(65536)]
    cands_t = eng.to_device(L.pack(cands, limbs))
    sieve_out = torch.empty(len(cands), dtype=torch.uint8, device=eng.device)
    jac_out = torch.empty(bp.cands * bp.GENS, dtype=torch.int8, device=eng.device)
    verdict = torch.empty((bp.cands, bp.KEEP), dtype=torch.uint8, device=eng.device)
    dw = DecryptWorkload(eng, 2048, 10000, 0, False)
    msg = torch.empty((dw.batch, dw.limbs), dtype=torch.int32, device=eng.device)
    status = torch.zeros(dw.batch, dtype=torch.uint8, device=eng.device)
    prime_p = int(sympy.nextprime(1 << (2 * (1024 + 3))))               # the Shamir prime of a 5-party key_length-2048 keygen (DK:647-651)
    sh_limbs = L.limbs_for(prime_p)
    base_col = [rng.randrange(prime_p) for _ in range(4096)]
    share_cols = torch.stack([eng.to_device(L.pack(base_col[k:] + base_col[:k], sh_limbs)).repeat(16, 1) for k in range(5)])
    coeffs = [rng.randrange(prime_p) for _ in range(5)]
    cases = [
        ("jacobi", "jacobi_kernel", bp.cands * bp.GENS, "symbols", lambda: eng.jacobi_t(bp.g_t, bp.mods_op, bp.GENS, out_t=jac_out)),
        ("sieve", "sieve_kernel", len(cands), "candidates", lambda: eng.sieve_t(cands_t, primes, out_t=sieve_out)),
        ("combine", "combine_kernel", dw.batch, "ciphertexts",
         lambda: eng.combine_t(dw.partials_t, dw.n, dw.theta_inv, out_t=msg, status_t=status)),
        ("verdict", "verdict_kernel", bp.cands * bp.KEEP, "slots", lambda: eng.biprime_verdict_t(bp.v_all, bp.mods_op, pass_t=verdict)),
        ("lincomb", "lincomb_kernel", int(share_cols.shape[1]), "candidates", lambda: eng.shamir_lincomb_t(share_cols, coeffs, prime_p)),
    ]
    return cases, (bp, dw, cands_t, share_cols, sieve_out, jac_out, verdict, msg, status)


def leg_short_kernels(eng, torch, reps: int = 8) -> dict:
    counts = _load_json(SHORT_KERNELS) or {}
    cases, keep_alive = short_kernel_cases(eng, torch)
    eng.set_priority_aux(False)
    out = {}
    for name, kernel, units, unit, fn in cases:
        fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            fn()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / reps
        row = {"kernel": kernel, "ms_per_launch": ms, "value": units / (ms * 1e-3), "unit": f"{unit}/s", "units_per_launch": units}
        c = counts.get(name)
        if c and c.get("units_per_launch") == units:
            instr = c["valu_instructions_per_launch"]
            row["roofline"] = {
                "bound": "valu-issue", "instructions_per_launch": instr, "waves_per_launch": c.get("waves_per_launch"),
                "achieved": instr / (ms * 1e-3) / 1e9, "unit": "G VALU wave-instructions/s",
                # the instruction mix of these kernels is not calibrated per class: the plain-VALU issue rate (2.28 cycles per
                # instruction and SIMD) is the roof of a multiply-free stream, the multiply issue rate (4.19) that of a pure
                # multiply stream; the roof of the real mix lies between the two
                "peak": PLAIN_ISSUE_PEAK / 1e9, "frac": instr / (ms * 1e-3) / PLAIN_ISSUE_PEAK,
                "frac_if_all_multiplies": instr / (ms * 1e-3) / MAC_ISSUE_PEAK,
                "kernel_ms_in_counter_pass": c.get("kernel_ms"), "source": SHORT_KERNELS.name,
            }
        out[name] = row
    del keep_alive
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------
# The result: ONE compact JSON line on stdout (the driver keeps the last 8 KB of stdout and parses the last line;
# round 3's 31.6 KB line left BENCH_r03.parsed null), every detail in bench_extras.json beside this file.
# ---------------------------------------------------------------------------------------------------
EXTRAS_FILE = ROOT / "bench_extras.json"
LINE_TARGET, LINE_LIMIT = 4096, 8192


def _num(x, digits: int = 5):
    """floats to `digits` significant digits (the line is for parsing, the full precision is in bench_extras.json)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _pick(d, keys, digits: int = 5):
    return {k: _num(d[k], digits) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_roofline(roof):
    """Numbers + kernel + one-tag bases; the prose of the full block is DESIGN.md §6."""
    if not isinstance(roof, dict):
        return None
    r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "concurrent_launches", "mac_share",
                     "frac_issue_slots", "frac_at_measured_clock", "clock_sample_inconsistent", "shader_clock_mhz_measured",
                     "frac_vs_guide_vector_peak", "instructions_per_launch"))
    r["traffic"] = _num(roof.get("traffic"))
    if roof.get("frac") is None:
        r["frac"] = None
        r["why_null"] = str(roof.get("instructions_basis", ""))[:160]
    r["peak_basis"] = (f"{SIMDS} SIMDs x {NOMINAL_HZ / 1e9} GHz / {MAC_CYCLES} cycles per integer multiply wave-instruction (measured; SURVEY 8d); "
                       f"frac_issue_slots: whole mix at mac_share x {MAC_CYCLES} + (1 - mac_share) x {OTHER_CYCLES} cycles (DESIGN.md 6)")
    hbm = roof.get("hbm")
    if isinstance(hbm, dict):
        r["hbm"] = _pick(hbm, ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "traffic", "traffic_model",
                               "traffic_over_algorithmic"))
    return r


def compact_cpu(cb):
    if not isinstance(cb, dict):
        return None
    c = _pick(cb, ("value", "unit", "cores", "kind", "engine", "single_core_value", "parallel_efficiency"))
    c.setdefault("value", None)
    c["sample"] = str(cb.get("sample", ""))[:200]
    return c


def _leg_summary(leg):
    """one number (+ the roofline fraction where the leg has one) per extra leg"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": str(leg["error"])[:80]}
    s = {"value": _num(leg.get("value"), 4)}
    roof = leg.get("roofline")
    if isinstance(roof, dict) and roof.get("frac") is not None:
        s["frac"] = _num(roof["frac"], 3)
    return s


def compact_result(out: dict) -> dict:
    """The driver's contract fields + roofline + cpu_baseline (+ distributed when N > 1) and one-number summaries of
    the extra legs; everything else stays in bench_extras.json."""
    line = {k: _num(out.get(k), 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                             "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "batch_per_gpu", "candidates_per_gpu", "mod_bits", "exp_bits", "parallelism",
                                          "steps_in_flight", "geometry_K_L_W_blocks", "wavefronts_per_group", "cu_slices")
                      if k in cfg}
    if "verified" in cfg:
        line["config"]["verified"] = str(cfg["verified"])[:120]
    line["roofline"] = compact_roofline(out.get("roofline"))
    line["cpu_baseline"] = compact_cpu(out.get("cpu_baseline"))
    if out.get("distributed"):
        line["distributed"] = {k: (v if not isinstance(v, str) else v[:100]) for k, v in out["distributed"].items()}
    summary = {}
    for name in ("single_batch", "latency", "end_to_end", "end_to_end_keygen"):
        if name in out:
            summary[name] = _leg_summary(out[name])
    for name, leg in (out.get("extra") or {}).items():
        summary[name] = _leg_summary(leg)
        if isinstance(leg, dict) and leg.get("n_gpus", 1) > 1:
            summary[name]["n_gpus"] = leg["n_gpus"]
        if isinstance(leg, dict) and isinstance(leg.get("latency"), dict) and "n1" in leg["latency"]:
            summary[name]["latency_ms"] = _num(leg["latency"]["n1"].get("ms"), 4)       # one decrypt() at this key length
    sk = out.get("short_kernels")
    if isinstance(sk, dict):
        summary["short_kernels"] = ({"error": str(sk["error"])[:80]} if "error" in sk else
                                    {name: _leg_summary(row) for name, row in sk.items()})
    if summary:
        line["extra_summary"] = summary
    # north_star's second target (biprimality-test modexps/s >= 10x gmpy2) as flat top-level scalars: the driver's record
    # keeps top-level fields, not the nested summaries
    bp = (out.get("extra") or {}).get("biprime_k2048")
    if isinstance(bp, dict) and "error" not in bp:
        line["biprime_modexps_per_s"] = _num(bp.get("value"), 5)
        line["biprime_roofline_frac"] = _num((bp.get("roofline") or {}).get("frac"), 4)
        line["biprime_cpu_baseline_modexps_per_s"] = _num((bp.get("cpu_baseline") or {}).get("value"), 5)
    line["details"] = EXTRAS_FILE.name
    return line


def result_line(out: dict) -> str:
    """The last stdout line: compact_result, shrunk further (summaries, then the optional roofline fields) if it would
    not fit — it never exceeds LINE_LIMIT."""
    line = compact_result(out)
    text = json.dumps(line, separators=(",", ":"))
    for drop in (("extra_summary",), ("distributed",), ("biprime_modexps_per_s", "biprime_roofline_frac", "biprime_cpu_baseline_modexps_per_s")):
        if len(text) <= LINE_TARGET:
            break
        if drop != ("extra_summary",) and len(text) <= LINE_LIMIT:     # the summaries go first; the rest only if it must
            break
        for k in drop:
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:          # cannot happen with the fields above; keep the contract anyway
        roof = line.get("roofline") or {}
        line["roofline"] = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")}
        line["config"] = {"workload": str((line.get("config") or {}).get("workload", ""))[:200]}
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def emit_result(out: dict, result_fd: int) -> None:
    full = json.dumps(out)
    try:
        EXTRAS_FILE.write_text(full + "\n")
        scratch = ROOT / "gpurun_out"
        if scratch.is_dir():                       # on the GPU box: the copy that travels back
            (scratch / EXTRAS_FILE.name).write_text(full + "\n")
    except OSError as exc:  # pragma: no cover - read-only checkout
        sys.stderr.write(f"bench.py: could not write {EXTRAS_FILE}: {exc}\n")
    sys.stderr.write("bench.py details: " + full + "\n")
    os.write(result_fd, (result_line(out) + "\n").encode())


# ---------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` without a launcher: this process starts the N ranks itself
# ---------------------------------------------------------------------------------------------------
def rank_environment(rank: int, world: int, port: int, base=None) -> dict:
    """What torch.distributed.run would export for local rank `rank` of a one-node job."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MX_BENCH_SPAWNED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: the only form this pool's host driver supports
    return env


def rank_command(argv, script=None) -> list:
    return [sys.executable, str(script or Path(__file__).resolve()), *argv]


def profiler_preload(env=None) -> str:
    """Name of the variable that says a GPU profiler is preloaded into this process (rocprofv3 exports these for its
    target), or ''.  Such a process has initialised the GPU before Python started, and starting rank processes from it is
    the exec-after-GPU-init this pool forbids."""
    env = os.environ if env is None else env
    if any(tok in env.get("LD_PRELOAD", "") for tok in ("rocprof", "rocprofiler", "roctracer", "rocsys")):
        return "LD_PRELOAD"
    return next((k for k in env if k.startswith(("ROCP_TOOL_LIB", "ROCPROFILER_", "ROCPROF_", "ROCTRACER_")) and env[k]), "")


def spawn_ranks(argv, world: int, script=None, poll_s: float = 0.2, attempts: int = 3) -> int:
    """One FRESH child process per GPU (this parent has made no GPU call and makes none), rendezvous on 127.0.0.1.
    Rank 0's single stdout line is relayed as this process's single stdout line; the other ranks' stdout goes to
    stderr.  Returns 0, or the exit code of the first rank that failed — the remaining ranks (exactly the PIDs started
    here) are then terminated instead of being left in a collective nobody will complete.  The rendezvous port is
    picked by binding port 0 and released before the ranks start; if another process takes it in between, rank 0 fails
    with "address already in use" and the ranks are started again on another port (at most `attempts` times)."""
    import socket
    import threading

    why = profiler_preload()
    if why:
        sys.stderr.write(f"bench.py --gpus {world}: a GPU profiler is preloaded into this process ({why}); it has initialised the GPU, and "
                         "starting the rank processes from here is refused.  Profile ONE rank instead: rocprofv3 ... -- python3 bench.py --gpus 1 ..., "
                         "or start the ranks with torch.distributed.run and put the profiler in front of a single rank's command.\n")
        return 2
    for attempt in range(attempts):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(world):
            procs.append(subprocess.Popen(rank_command(argv, script), env=rank_environment(r, world, port),
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(),
                                          stderr=subprocess.PIPE if r == 0 else None, text=(r == 0)))
        captured, port_taken = [], []

        def relay_stderr():
            for ln in procs[0].stderr:
                if "address already in use" in ln.lower() or "EADDRINUSE" in ln:
                    port_taken.append(ln)
                sys.stderr.write(ln)

        reader = threading.Thread(target=lambda: captured.extend(procs[0].stdout), daemon=True)
        err_reader = threading.Thread(target=relay_stderr, daemon=True)
        reader.start()
        err_reader.start()
        failed = 0
        try:
            while any(p.poll() is None for p in procs):
                bad = next((p for p in procs if p.poll() not in (None, 0)), None)
                if bad is not None:
                    failed = bad.returncode
                    sys.stderr.write(f"bench.py: rank {procs.index(bad)} exited with {failed}; stopping the other ranks\n")
                    break
                time.sleep(poll_s)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        reader.join(timeout=5)
        err_reader.join(timeout=5)
        failed = failed or next((p.returncode for p in procs if p.returncode != 0), 0)
        if failed and port_taken and attempt + 1 < attempts:
            sys.stderr.write(f"bench.py: port {port} was taken before the ranks met; starting them again on another port\n")
            continue
        break
    lines = [ln.strip() for ln in captured if ln.strip()]
    if failed == 0 and not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        failed = 1
    if failed == 0:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return failed if failed > 0 else (128 - failed if failed < 0 else 0)


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way a one-GPU run is started: become the launcher (before torch or the HIP runtime is loaded)
        raise SystemExit(spawn_ranks(sys.argv[1:], args.gpus))
    # stdout carries exactly ONE line, the JSON result.  Libraries write banners to the C-level stdout
    # (RCCL prints its version block when the first communicator is created, and stdio would flush it
    # after Python's own output), so file descriptor 1 is pointed at stderr for the whole run and the
    # result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py directly (it spawns one process per GPU) "
                         "or under torch.distributed.run with --nproc-per-node equal to --gpus")
    # one process per GPU; MX_BENCH_BACKEND=gloo lets several ranks share one GPU (single-GPU smoke
    # test of the multi-rank code path; RCCL refuses two ranks on one device)
    backend = os.environ.get("MX_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and 0 < n_dev < world:
        raise SystemExit(f"--gpus {world} on a node with {n_dev} GPU(s): RCCL needs one device per rank "
                         "(MX_BENCH_BACKEND=gloo runs the multi-rank code path with ranks sharing a device)")
    local_rank = local_rank % max(1, n_dev)
    torch.cuda.set_device(local_rank)
    dist = None
    # MX_BENCH_FORCE_DIST=1 runs the process-group code path (RCCL init, all-gather, barrier) with a
    # single rank: the only way to exercise it on a one-GPU box
    force_dist = world == 1 and os.environ.get("MX_BENCH_FORCE_DIST") == "1"
    if force_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or force_dist:
        import torch.distributed as dist  # type: ignore

        if backend == "nccl":
            # the all-gather kernels must find wavefront slots on a GPU that the modexp launches keep
            # full: give RCCL's stream the high-priority queue (falls back if the option is unavailable)
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                dist.init_process_group("nccl", pg_options=opts, device_id=torch.device("cuda", local_rank))
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # a second, socket-based group for waits that must not occupy host cores: while rank 0 times the CPU baseline
        # on ALL usable cores, the other ranks block in a gloo barrier (an RCCL barrier is a device kernel plus a host
        # thread that may spin on the stream)
        QUIET["group"] = dist.new_group(backend="gloo") if world > 1 else None

    from protocols.distributed_keygen_amd import build as _build

    if _build.needs_build():         # missing or stale library: compile it (local rank 0), never a CPU path
        if local_rank == 0:
            _build.build(force=False)
        if dist is not None:
            dist.barrier()
    from protocols.distributed_keygen_amd import Engine

    eng = Engine(local_rank)
    for kv in args.knob:
        name, _, value = kv.partition("=")
        eng.debug_knob(name, int(value))
    extras = world == 1 and not args.no_extras and not args.generic_modulus
    if args.workload in ("c3", "c5"):
        key_length = args.key_length or (2048 if args.workload == "c3" else 4096)
        batch = args.batch or (10000 if args.workload == "c3" else 4096)
        label = "C3 (BASELINE.json configs[2])" if args.workload == "c3" else "C5 (BASELINE.json configs[4])"
        out = run_decrypt_main(args, eng, torch, dist, rank, world, key_length, batch, label)
        cb = None
        if not args.no_cpu_baseline:
            def _cpu():
                w_ = out["_wl"]
                bases = [c if w_.exps[w_.own] >= 0 else pow(c, -1, w_.n2) for c in w_.cts[:64]]
                # N > 1: a shorter sample — the other ranks wait for it, and the driver runs four such jobs back to back
                secs = args.cpu_seconds if world == 1 else min(args.cpu_seconds, 5.0)
                res = cpu_baseline(w_.n2, w_.own_exp, bases, secs, "same modulus/exponent, first 64 ciphertexts of the batch cycled")
                if world > 1:
                    res["sample"] = (res.get("sample", "") + f"; timed by rank 0 after the timed region while the other {world - 1} rank(s) wait in a gloo barrier")[:400]
                return res

            cb = cpu_baseline_on_rank0(dist, rank, _cpu)
        if rank == 0:
            wl = out.pop("_wl")
            out["cpu_baseline"] = cb
            if not extras:
                del wl
            if extras:
                # the extra legs must never cost the headline line: a failing leg is reported in its field
                leg_seconds = out.setdefault("leg_seconds", {})

                def guarded(name, fn):
                    t_leg = time.perf_counter()
                    try:
                        return fn()
                    except Exception as exc:  # pragma: no cover - measurement plumbing
                        import traceback

                        sys.stderr.write(f"bench.py: leg {name} failed:\n{traceback.format_exc()}\n")
                        return {"error": f"{type(exc).__name__}: {exc}"}
                    finally:
                        leg_seconds[name] = round(time.perf_counter() - t_leg, 2)

                out["single_batch"] = guarded("single_batch", lambda: leg_single_batch(eng, torch, wl, key_length))
                out["latency"] = guarded("latency", lambda: leg_latency(eng, torch, wl, (out.get("cpu_baseline") or {}).get("single_core_value")))
                out["end_to_end"] = guarded("end_to_end", lambda: leg_end_to_end(eng, torch, wl, out["value"]))
                if args.workload == "c3" and key_length == 2048:
                    # (before the legs that keep up to eight steps in flight: a process that has used dozens of streams is
                    # time-sliced by the queue scheduler, and the small rounds of this leg — a 6 ms launch — then measure
                    # that, 13.6 instead of 5.4 ms per v-calculation, not the engine: profiles/r05_keygen_round_small.txt)
                    out["end_to_end_keygen"] = guarded("end_to_end_keygen", lambda: leg_keygen_round(eng, torch, args))
                del wl
                torch.cuda.empty_cache()
                out["extra"] = {}
                if args.workload == "c3" and key_length == 2048:
                    keep_bp = ("metric", "value", "unit", "steps", "ms_per_step", "config", "stages", "roofline", "cpu_baseline")

                    def biprime_leg(klen, cands, steps, cpu):
                        bp = run_biprime(args, eng, torch, None, 0, 1, klen, cands, steps=steps, warmup=2, nstreams=biprime_lanes(cands, steps))
                        bwl = bp.pop("_wl")
                        if cpu and not args.no_cpu_baseline:
                            bp["cpu_baseline"] = cpu_baseline(bwl.mods[0], bwl.exps[0], bwl.g_sample[:40], min(args.cpu_seconds, 3.0),
                                                              "candidate 0's modulus and party-1 exponent, its first 40 generators cycled")
                        del bwl
                        torch.cuda.empty_cache()
                        return {k: bp[k] for k in keep_bp if k in bp}

                    def c5_leg(batch, steps, streams, cpu):
                        a5 = argparse.Namespace(**vars(args))
                        a5.steps, a5.warmup, a5.streams, a5.limbs_per_lane, a5.wavefronts_per_group, a5.check = steps, 2, streams, -1, 0, 3
                        c5 = run_decrypt_main(a5, eng, torch, None, 0, 1, 4096, batch, "C5 (BASELINE.json configs[4])")
                        c5wl = c5.pop("_wl")
                        if cpu and not args.no_cpu_baseline:
                            bases = [c if c5wl.exps[c5wl.own] >= 0 else pow(c, -1, c5wl.n2) for c in c5wl.cts[:32]]
                            c5["cpu_baseline"] = cpu_baseline(c5wl.n2, c5wl.own_exp, bases, min(args.cpu_seconds, 3.0),
                                                              "same modulus/exponent, first 32 ciphertexts cycled")
                        if cpu:
                            # the lone call at this key length (decrypt() of ONE ciphertext, 64, 1024), as leg `latency` at 2048
                            lat = leg_latency(eng, torch, c5wl, (c5.get("cpu_baseline") or {}).get("single_core_value"))
                            c5["latency"] = {k: lat[k] for k in ("unit", "n1", "n1_negative_exponent", "n64", "n1024", "gmpy2_one_core_ms", "vs_gmpy2_one_core") if k in lat}
                        del c5wl
                        torch.cuda.empty_cache()
                        return {k: c5[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline", "cpu_baseline", "latency") if k in c5}

                    out["extra"]["biprime_k2048"] = guarded("biprime_k2048", lambda: biprime_leg(2048, 4096, 8, True))
                    # configs[1]: key_length 1024, at the size of a keygen round's survivors and at a saturating size
                    out["extra"]["biprime_k1024_c256"] = guarded("biprime_k1024_c256", lambda: biprime_leg(1024, 256, 48, False))
                    out["extra"]["biprime_k1024_c8192"] = guarded("biprime_k1024_c8192", lambda: biprime_leg(1024, 8192, 6, False))
                    # configs[4]: the sweep points of key_length 4096
                    out["extra"]["c5_k4096"] = guarded("c5_k4096", lambda: c5_leg(4096, 8, 4, True))
                    # 1024 per step: eight in flight (one launch is 256 wavefronts; 4 / 8 / 12 in flight: 32.6 / 36.8 / 37.2 k/s,
                    # profiles/r04_decrypt_lanes.txt — past four lanes the recombination shares the lane's stream)
                    out["extra"]["c5_k4096_b1024"] = guarded("c5_k4096_b1024", lambda: c5_leg(1024, 24, 8, False))
                    out["extra"]["c5_k4096_b16384"] = guarded("c5_k4096_b16384", lambda: c5_leg(16384, 4, 2, False))
                    out["short_kernels"] = guarded("short_kernels", lambda: leg_short_kernels(eng, torch))
        if world > 1 and not args.no_extras and args.workload == "c3" and not args.generic_modulus:
            # configs[3] on N GPUs inside the driver's scaling run: 4096 candidates sharded over the ranks,
            # all-gather of the v rows and of the verdict bytes (the biprimality vote, DK:1331-1360)
            if rank == 0:
                out.pop("_wl", None)
            torch.cuda.empty_cache()
            # (16 steps: a rank's shard is a 16-64 ms step, and a run of two rounds of four lanes is mostly ramp and drain —
            # 512 candidates per rank: 1.19 M modexps/s over 8 steps, 1.28-1.31 over 16-48)
            bp = run_biprime(args, eng, torch, dist, rank, world, 2048, 4096, steps=16, warmup=4, nstreams=biprime_lanes(-(-4096 // world), 16))
            if rank == 0:
                bp.pop("_wl", None)
                out["extra"] = {"biprime_k2048": {k: bp[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "distributed", "config", "stages", "roofline") if k in bp}}
    else:
        key_length = args.key_length or 2048
        weak = args.scaling == "weak"
        total = (args.batch or 4096) if weak else (args.batch * world if args.batch else 4096)
        nstreams = args.streams if args.streams > 0 else biprime_lanes(total if weak else -(-total // world), args.steps)
        out = run_biprime(args, eng, torch, dist, rank, world, key_length, total, args.steps, args.warmup, nstreams, weak=weak)
        cb = None
        if not args.no_cpu_baseline:
            def _cpu_bp():
                w_ = out["_wl"]
                secs = args.cpu_seconds if world == 1 else min(args.cpu_seconds, 5.0)
                return cpu_baseline(w_.mods[0], w_.exps[0], w_.g_sample[:40], secs,
                                    "candidate 0's modulus and party-1 exponent, its first 40 generators cycled"
                                    + ("" if world == 1 else f"; timed by rank 0 after the timed region while the other {world - 1} rank(s) wait in a gloo barrier"))

            cb = cpu_baseline_on_rank0(dist, rank, _cpu_bp)
        if rank == 0:
            wl = out.pop("_wl")
            out["cpu_baseline"] = cb
    if rank == 0:
        emit_result(out, result_fd)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
