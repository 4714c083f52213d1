"""Drift guard: the HIP engine is only ever composed with the builder-written stand-in package
(tests/standin/keygen_standin, on the GPU box) and the real reference only with the test double of the engine (here).
What ties the two together is that patch.py touches the SAME names with the SAME call signatures in both.  This test
imports the unmodified reference (the recipe of tests/golden/make_golden.py; build container only) and asserts that
every callable patch.py rebinds, calls or instantiates has an identical signature in the stand-in, and that every
attribute it reads exists on both.

Reference lines: distributed_keygen.py:314 (_decrypt_raw), :430 (_decrypt_sequence_raw), :1056 / :1110 / :1197 (the
name-mangled class-methods), :1211 (compute_modulus), :1015 (__biprime_test_g_generation), :771 (_generate_pq);
paillier_shared_key.py:30-50 (constructor and attributes), :52, :95."""

from __future__ import annotations

import inspect
import re
import sys
from pathlib import Path

import pytest

REF = Path("/root/reference/src/tno/mpc/protocols/distributed_keygen")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference sources only exist in the build container")

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
ROOT = Path(__file__).resolve().parent.parent
M = "_DistributedPaillier__"

# (module, dotted name) of every callable patch.py replaces, calls or instantiates
CALLABLES = [
    ("psk", "PaillierSharedKey.__init__"), ("psk", "PaillierSharedKey.partial_decrypt"), ("psk", "PaillierSharedKey.decrypt"),
    ("dk", "DistributedPaillier._decrypt_raw"), ("dk", "DistributedPaillier._decrypt_sequence_raw"),
    ("dk", "DistributedPaillier.compute_modulus"), ("dk", "DistributedPaillier._generate_pq"),
    ("dk", f"DistributedPaillier.{M}small_prime_divisors_test"), ("dk", f"DistributedPaillier.{M}biprime_test_v_calculation"),
    ("dk", f"DistributedPaillier.{M}biprime_test_with_v_i"), ("dk", f"DistributedPaillier.{M}biprime_test_g_generation"),
    ("dk", "exchange_reconstruct"),
]
# call shapes patch.py uses on objects of the package (keyword names must exist on both sides)
CALL_SHAPES = [
    ("dk", "Batched.__init__", ("batch_size",)), ("dk", "AdditiveVariable.__init__", ("label", "modulus")),
    ("dk", "Batched.set_share", ()), ("dk", "AdditiveVariable.get_share", ()), ("dk", "ShamirVariable.get_shares", ()),
    ("dk", "EncodedPlaintext.__init__", ("scheme",)),
]
# (the reference's distributed_keygen module binds pow_mod only, DK:35; patch.install(leaf=True) rebinds mod_inv there
# only `if hasattr(...)`)
ATTRIBUTES = [
    ("psk", "pow_mod"), ("psk", "mod_inv"), ("psk", "PaillierCiphertext"),
    ("dk", "pow_mod"), ("dk", "logger"), ("dk", "Shares"), ("dk", "Shares.P"), ("dk", "Shares.Q"),
    ("dk", "Batched"), ("dk", "AdditiveVariable"), ("dk", "EncodedPlaintext"),
]


@pytest.fixture(scope="module")
def pairs():
    import make_golden

    import standin_harness as sh

    ref_psk, ref_dk = make_golden.load_reference()
    st_psk, st_dk = sh.modules()
    return {"psk": (ref_psk, st_psk), "dk": (ref_dk, st_dk)}


def _get(mod, dotted):
    obj = mod
    for part in dotted.split("."):
        obj = inspect.getattr_static(obj, part) if inspect.isclass(obj) else getattr(obj, part)
    return obj


def _sig(obj):
    kind = "plain"
    if isinstance(obj, classmethod):
        kind, obj = "classmethod", obj.__func__
    elif isinstance(obj, staticmethod):
        kind, obj = "staticmethod", obj.__func__
    sig = inspect.signature(obj)
    params = [(p.name, p.kind, p.default is not inspect.Parameter.empty, None if p.default is inspect.Parameter.empty else repr(p.default))
              for p in sig.parameters.values()]
    return kind, inspect.iscoroutinefunction(obj), params


@pytest.mark.parametrize("which,name", CALLABLES, ids=[n for _, n in CALLABLES])
def test_same_signature(pairs, which, name):
    ref_mod, st_mod = pairs[which]
    ref, st = _sig(_get(ref_mod, name)), _sig(_get(st_mod, name))
    if name == "PaillierSharedKey.__init__":
        # the reference's constructor passes further keyword arguments on to SecretKey (none are used by the patch)
        ref = (ref[0], ref[1], [p for p in ref[2] if p[1] not in (inspect.Parameter.VAR_KEYWORD,)])
        st = (st[0], st[1], [p for p in st[2] if p[1] not in (inspect.Parameter.VAR_KEYWORD,)])
    assert st == ref, f"{name}: stand-in {st} != reference {ref}"


@pytest.mark.parametrize("which,name,keywords", CALL_SHAPES, ids=[n for _, n, _ in CALL_SHAPES])
def test_call_shapes_exist_on_both(pairs, which, name, keywords):
    for mod in pairs[which]:
        params = inspect.signature(_get(mod, name)).parameters
        for kw in keywords:
            assert kw in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()), (mod.__name__, name, kw)


@pytest.mark.parametrize("which,name", ATTRIBUTES, ids=[n for _, n in ATTRIBUTES])
def test_same_names_exist(pairs, which, name):
    for mod in pairs[which]:
        assert _get(mod, name) is not None


def test_patch_touches_nothing_unlisted():
    """Every attribute patch.py reads off the two modules is on the lists above — a new dependency of the patch on the
    reference's surface has to be added here (and to the stand-in) to pass."""
    src = (ROOT / "protocols" / "distributed_keygen_amd" / "patch.py").read_text()
    listed = {n.split(".")[0] for _, n in CALLABLES + ATTRIBUTES} | {n.split(".")[0] for _, n, _ in CALL_SHAPES}
    listed |= {n.split(".")[-1] for _, n in CALLABLES} | {"PaillierSharedKey", "DistributedPaillier"}
    used = set(re.findall(r"\b(?:dk_mod|psk_mod)\.([A-Za-z_]\w*)", src))
    used |= {M + n for n in re.findall(r'mangled \+ "(\w+)"', src)}
    used |= set(re.findall(r'_save\(DP, "(\w+)"\)', src)) | set(re.findall(r"cls\.(_generate_pq)\b", src))
    missing = {u for u in used if u not in listed}
    assert not missing, missing


def test_instance_attributes_the_patch_reads(pairs):
    """PaillierSharedKey attributes (PSK:30-50) read by shared_key.GpuPaillierSharedKey.from_reference / patch._gpu_key."""
    import random

    import make_golden
    import standin_harness as sh
    from protocols.distributed_keygen_amd import synthetic

    ref_psk, _ = pairs["psk"]
    shamir = sys.modules["tno.mpc.encryption_schemes.shamir"]
    k = make_golden.synth_key(random.Random(1), 64, 3, 1)
    ref_key = ref_psk.PaillierSharedKey(n=k["n"], t=1, player_id=1, theta=k["theta"],
                                        share=shamir.IntegerShares(shamir._Scheme(3), {1: k["shares"][1]}, k["degree"], k["n_fac"] ** 2))
    st_key = sh.parties_for_key(synthetic.make_key(64, 3, 1, kappa=20))[0].secret_key
    for attr in ("n", "n_square", "t", "player_id", "theta", "theta_inv", "share"):
        assert hasattr(ref_key, attr) and hasattr(st_key, attr), attr
    for attr in ("shares", "degree", "n_fac"):
        assert hasattr(ref_key.share, attr) and hasattr(st_key.share, attr), attr
