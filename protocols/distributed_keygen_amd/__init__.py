"""MI355X-native engine for the compute hot path of TNO-MPC/protocols.distributed_keygen.

Batched modular exponentiation (biprimality-test v-values, partial decryptions), small-prime
sieve, share recombination and biprimality verdict as hand-written HIP kernels for gfx950 behind
a C ABI (include/mxpaillier.h), with a Python layer that mirrors the reference's operator
surface.  See DESIGN.md and INTEGRATION.md.
"""

import os as _os


def configure_hw_queues(queues: int = 16) -> bool:
    """Opt-in process set-up for callers that keep several launches of this engine in flight (bench.py
    calls it; importing the package does NOT).  The HIP runtime multiplexes streams onto 4 hardware queues
    by default and two streams that share a queue serialise their kernels; a launch of this engine occupies
    the GPU for tens of milliseconds, so a collision between two of the streams that are meant to fill the
    machine together costs 25-40 % (profiles/r02_hw_queue_collisions.txt).  ``GPU_MAX_HW_QUEUES`` is read
    once, when the runtime initialises: this sets it (an explicit setting of the user wins) and returns
    True if the process has not touched the GPU yet, False — without changing anything — if it is too
    late.  Either way ``Engine`` measures what it actually got (``Engine.stream_concurrency``) and cuts long
    batches into no more chunks than run concurrently."""
    import sys

    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return False
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(queues)))
    return True


from . import limbs  # noqa: F401
from .engine import Engine, default_engine  # noqa: F401
from .operators import mod_inv, mod_inv_batch, pow_mod, pow_mod_batch, pow_mod_batch_multi  # noqa: F401

__version__ = "0.1.0"
