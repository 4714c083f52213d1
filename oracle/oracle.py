"""Python big-int restatement of the reference's hot-path arithmetic (TEST INFRASTRUCTURE).

Reference = TNO-MPC/protocols.distributed_keygen v4.2.2.  File abbreviations:
  DK  = src/tno/mpc/protocols/distributed_keygen/distributed_keygen.py
  PSK = src/tno/mpc/protocols/distributed_keygen/paillier_shared_key.py
  UT  = src/tno/mpc/protocols/distributed_keygen/utils.py

The arithmetic leaf (``pow_mod`` / ``mod_inv``) lives in the un-vendored dependency
``tno.mpc.encryption_schemes.utils ~=0.10`` (pyproject.toml:40; imported DK:35, PSK:20), which
dispatches to ``gmpy2.powmod`` / ``gmpy2.invert`` when gmpy2 is installed and to CPython ``pow``
otherwise.  Both return the canonical residue in ``[0, modulus)``, so CPython ``pow`` is a
bit-exact restatement.

Every function below is a plain function of integers; protocol containers (Batched,
AdditiveVariable, IntegerShares, PaillierCiphertext) are replaced by the integers they carry.
"""

from __future__ import annotations

from math import factorial
from typing import Dict, List, Sequence


# --------------------------------------------------------------------------- leaf arithmetic
def pow_mod(value: int, exponent: int, modulus: int) -> int:
    """tno.mpc.encryption_schemes.utils.pow_mod as used at DK:1094, DK:1097, PSK:92."""
    return pow(value, exponent, modulus)


def mod_inv(value: int, modulus: int) -> int:
    """tno.mpc.encryption_schemes.utils.mod_inv as used at PSK:50, PSK:90."""
    return pow(value, -1, modulus)


def mult_list(list_: Sequence[int], modulus: int | None = None) -> int:
    """UT:23-38 — left-to-right product, reduced each step when a modulus is given."""
    out = 1
    if modulus is None:
        for element in list_:
            out = out * element
    else:
        for element in list_:
            out = out * element % modulus
    return out


def jacobi_symbol(m: int, n: int) -> int:
    """sympy.jacobi_symbol(m, n) as called at DK:1089 (n odd, positive).

    Binary algorithm (quadratic reciprocity + the (2/n) supplement); checked against
    sympy.jacobi_symbol in tests/test_oracle_golden.py.
    """
    if n <= 0 or n % 2 == 0:
        raise ValueError("n should be an odd positive integer")
    m %= n
    result = 1
    while m != 0:
        while m % 2 == 0:
            m //= 2
            if n % 8 in (3, 5):
                result = -result
        m, n = n, m
        if m % 4 == 3 and n % 4 == 3:
            result = -result
        m %= n
    return result if n == 1 else 0


# --------------------------------------------------------------------------- sieve
def small_prime_list(prime_threshold: int) -> List[int]:
    """DK:552-554 — ``list(sympy.primerange(3, prime_threshold + 1))`` (odd primes ≤ threshold)."""
    if prime_threshold < 3:
        return []
    flags = bytearray([1]) * (prime_threshold + 1)
    flags[0:2] = b"\x00\x00"
    i = 2
    while i * i <= prime_threshold:
        if flags[i]:
            flags[i * i :: i] = bytearray(len(range(i * i, prime_threshold + 1, i)))
        i += 1
    return [p for p in range(3, prime_threshold + 1) if flags[p]]


def small_prime_divisors_test(prime_list: Sequence[int], modulus: int) -> bool:
    """DK:1197-1209 — True iff some prime of the list divides the modulus."""
    for prime in prime_list:
        if modulus % prime == 0:
            return True
    return False


# --------------------------------------------------------------------------- biprimality test (a1, a2)
def biprime_exponent(index: int, modulus: int, p_i: int, q_i: int) -> int:
    """Exponent used by party ``index``: DK:1094 (party 1) / DK:1097 (others)."""
    if index == 1:
        return (modulus - p_i - q_i + 1) // 4
    return (p_i + q_i) // 4


def biprime_test_v_calculation(
    g_values: Sequence[int],
    index: int,
    modulus: int,
    p_i: int,
    q_i: int,
    correct_param_biprime: int,
) -> List[int]:
    """DK:1056-1108 — the v values of one party for one candidate modulus.

    Walks g_values in order, skips g with Jacobi(g/N) != 1 (DK:1089), stops after
    ``correct_param_biprime`` values (DK:1086).  Returns the list the reference stores in the
    Batched[AdditiveVariable] under this party's index (DK:1103-1108).
    """
    v_values: List[int] = []
    for g in g_values:
        if len(v_values) == correct_param_biprime:
            break
        if jacobi_symbol(g, modulus) != 1:
            continue
        v_values.append(int(pow_mod(g, biprime_exponent(index, modulus, p_i, q_i), modulus)))
    return v_values


def biprime_test_with_v_i(
    v_by_party: Dict[int, Sequence[int]],
    modulus: int,
    correct_param_biprime: int,
) -> bool:
    """DK:1110-1175 — verdict for one candidate given every party's v list.

    ``v_by_party[i][k]`` is party i's share in test slot k.  Per slot: product of the shares of
    all parties but party 1, un-reduced (DK:1147-1151); pass iff v_1 ≡ ±product (mod N)
    (DK:1156-1158); False on the first failing slot (DK:1160-1164); True after
    ``correct_param_biprime`` passes (DK:1168-1172).  A slot without a share raises KeyError in
    the reference (``AdditiveVariable.get_share`` UT:368-377); mirrored here.
    """
    successful = 0
    for slot in range(correct_param_biprime):
        sharing = {}
        for i, values in v_by_party.items():
            if slot >= len(values):
                raise KeyError(i)
            sharing[i] = values[slot]
        product = 1
        for key, value in sharing.items():
            if key != 1:
                product *= value
        value1 = sharing[1]
        success = ((value1 % modulus) == (product % modulus)) or (
            (value1 % modulus) == (-product % modulus)
        )
        if not success:
            return False
        successful += 1
        if successful >= correct_param_biprime:
            return True
    return False


# --------------------------------------------------------------------------- threshold decryption (a4, a5)
def partial_decrypt_exponent(player_id: int, degree: int, n_fac: int, share: int) -> int:
    """PSK:70-85 — Lagrange-folded exponent of one player (may be negative)."""
    other_honest_players = [i + 1 for i in range(degree + 1) if i + 1 != player_id]
    enumerator = mult_list(other_honest_players)
    denominator = mult_list([(j - player_id) for j in other_honest_players])
    return (n_fac * enumerator * share) // denominator


def partial_decrypt(
    ciphertext_value: int, n: int, player_id: int, degree: int, n_fac: int, share: int
) -> int:
    """PSK:52-93 on the integers the objects carry (ciphertext value, share of this player)."""
    n_square = n * n
    exp = partial_decrypt_exponent(player_id, degree, n_fac, share)
    if exp < 0:  # PSK:89-91
        ciphertext_value = mod_inv(ciphertext_value, n_square)
        exp = -exp
    return pow_mod(ciphertext_value, exp, n_square)


def decrypt_combine(partial_dict: Dict[int, int], n: int, degree: int, theta_inv: int) -> int:
    """PSK:95-127 — recombine partial decryptions of players 1..degree+1."""
    n_square = n * n
    partial_decryptions = [partial_dict[i + 1] for i in range(degree + 1)]  # KeyError if absent
    if len(partial_decryptions) < degree + 1:
        raise ValueError("Not enough shares.")
    combined = mult_list(partial_decryptions[: degree + 1]) % n_square
    if (combined - 1) % n != 0:
        raise ValueError(
            "Combined decryption minus one is not divisible by N. This might be caused by the "
            "fact that the ciphertext that is being decrypted, differs between the parties."
        )
    return ((combined - 1) // n * theta_inv) % n


# --------------------------------------------------------------------------- input shapes
def prime_candidate_from_bits(index: int, prime_length: int, random_bits: int) -> int:
    """DK:855-876 with the ``secrets.randbits(prime_length - 3)`` draw passed in."""
    mod4 = 3 if index == 1 else 0
    return 2 ** (prime_length - 1) + (random_bits << 2) + mod4


def n_factorial(number_of_parties: int) -> int:
    """IntegerShares.n_fac of the un-vendored shamir package (used PSK:70)."""
    return factorial(number_of_parties)


# ---------------------------------------------------------------------------------------------------
# Shamir field of the key generation (DK:1274-1284 through utils.py:205-270 / 404-471).  The field
# arithmetic itself lives in the un-vendored tno.mpc.encryption_schemes.shamir (pinned ~=... by the
# reference's pyproject); it is the textbook prime-field Shamir scheme: share-wise product / sum
# modulo P, and Lagrange interpolation at 0 over degree+1 shares.
# ---------------------------------------------------------------------------------------------------
def shamir_mul_add(p_share: int, q_share: int, zero_share: int, prime: int) -> int:
    """One party's share of a candidate modulus: `p * q` (DK:1274, UT:229-250) then `+= zero` (DK:1277, UT:205-227)."""
    return (p_share * q_share % prime + zero_share) % prime


def shamir_share(secret: int, prime: int, number_of_parties: int, degree: int, rng) -> dict:
    """`ShamirSecretSharingScheme.share_secret` of the un-vendored package as `_generate_pq` uses it (DK:785-823 through
    utils.py:252-260): a random polynomial of `degree` with constant term `secret`, evaluated at 1..n.  `rng` supplies
    the coefficients (test-side randomness; the scheme's own is `secrets`)."""
    coeffs = [secret % prime] + [rng.randrange(prime) for _ in range(degree)]
    return {i: sum(c * pow(i, k, prime) for k, c in enumerate(coeffs)) % prime for i in range(1, number_of_parties + 1)}


def shamir_reconstruct(shares: dict, prime: int, degree: int) -> int:
    """`candidate_n.reconstruct()` (DK:1284, UT:263-270): value at 0 of the polynomial through the
    first degree+1 shares in party order."""
    pts = sorted(shares.items())[: degree + 1]
    if len(pts) < degree + 1:
        raise ValueError("not enough shares")
    total = 0
    for i, y in pts:
        num = den = 1
        for j, _ in pts:
            if j != i:
                num = num * j % prime
                den = den * (j - i) % prime
        total = (total + y * num * pow(den, -1, prime)) % prime
    return total
