"""Stand-in for the reference's paillier_shared_key module (see the package docstring): a threshold
Paillier secret-key share with the scalar partial decryption and recombination, on CPython integers."""

from __future__ import annotations

from math import factorial
from typing import Dict


def pow_mod(value: int, exponent: int, modulus: int) -> int:
    return pow(value, exponent, modulus)


def mod_inv(value: int, modulus: int) -> int:
    return pow(value, -1, modulus)


class IntegerShares:
    """The members of the un-vendored integer Shamir share object that the key reads."""

    def __init__(self, n_parties: int, shares: Dict[int, int], degree: int) -> None:
        self.shares, self.degree, self.n_fac = shares, degree, factorial(n_parties)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, IntegerShares) and (self.shares, self.degree, self.n_fac) == (other.shares, other.degree, other.n_fac)


class PaillierCiphertext:
    def __init__(self, value: int, scheme) -> None:
        self._value, self.scheme, self.fresh = value, scheme, True

    def get_value(self) -> int:
        self.fresh = False
        return self._value

    def peek_value(self) -> int:
        return self._value


class PaillierSharedKey:
    def __init__(self, n: int, t: int, player_id: int, share: IntegerShares, theta: int) -> None:
        self.share, self.n, self.n_square, self.t, self.player_id, self.theta = share, n, n * n, t, player_id, theta
        self.theta_inv = mod_inv(theta, n)

    def partial_decrypt(self, ciphertext: PaillierCiphertext) -> int:
        if not isinstance(ciphertext, PaillierCiphertext):
            raise TypeError(f"not a ciphertext: {type(ciphertext)}")
        if ciphertext.scheme.public_key.n != self.n:
            raise ValueError("encrypted against a different key!")
        value = ciphertext.get_value()
        others = [j for j in range(1, self.share.degree + 2) if j != self.player_id]
        num = den = 1
        for j in others:
            num *= j
            den *= j - self.player_id
        exponent = (self.share.n_fac * num * self.share.shares[self.player_id]) // den
        if exponent < 0:
            value, exponent = mod_inv(value, self.n_square), -exponent
        return pow_mod(value, exponent, self.n_square)

    def decrypt(self, partial_dict: Dict[int, int]) -> int:
        product = 1
        for player in range(1, self.share.degree + 2):
            product = product * partial_dict[player] % self.n_square
        if (product - 1) % self.n != 0:
            raise ValueError("Combined decryption minus one is not divisible by N.")
        return (product - 1) // self.n * self.theta_inv % self.n
