"""MI355X-native engine for the compute hot path of TNO-MPC/protocols.distributed_keygen.

Batched modular exponentiation (biprimality-test v-values, partial decryptions), small-prime
sieve, share recombination and biprimality verdict as hand-written HIP kernels for gfx950 behind
a C ABI (include/mxpaillier.h), with a Python layer that mirrors the reference's operator
surface.  See DESIGN.md and INTEGRATION.md.
"""

from . import limbs  # noqa: F401
from .engine import Engine, default_engine  # noqa: F401

__version__ = "0.1.0"
