"""MI355X-native engine for the compute hot path of TNO-MPC/protocols.distributed_keygen.

Batched modular exponentiation (biprimality-test v-values, partial decryptions), small-prime
sieve, share recombination and biprimality verdict as hand-written HIP kernels for gfx950 behind
a C ABI (include/mxpaillier.h), with a Python layer that mirrors the reference's operator
surface.  See DESIGN.md and INTEGRATION.md.
"""

import os as _os

# The HIP runtime multiplexes streams onto 4 hardware queues by default, and two streams that share a
# queue serialise their kernels.  A launch of this engine occupies the GPU for tens of milliseconds, so
# a collision between two of the streams that keep the machine full (the chunks of a long int-level
# batch, a caller's batches in flight) costs 25-40 % (profiles/r02_hw_queue_collisions.txt).  The
# variable is read when the runtime initialises, so it only takes effect if this package is imported
# before the first HIP call of the process; an explicit setting of the user is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import limbs  # noqa: F401,E402
from .engine import Engine, default_engine  # noqa: F401,E402
from .operators import mod_inv, mod_inv_batch, pow_mod, pow_mod_batch, pow_mod_batch_multi  # noqa: F401,E402

__version__ = "0.1.0"
