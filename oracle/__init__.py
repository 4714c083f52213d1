"""CPU oracle for the hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package.  The product (``protocols.distributed_keygen_amd``) never
does, and fails loudly when its HIP library is missing.

Pinning status: **pinned** against outputs of the reference itself.  The reference's own tests
hold no known-answer vectors for this path, so ``tests/golden/make_golden.py``
shim-imports the unmodified reference modules from ``/root/reference`` (in the build container
only) and records what ``PaillierSharedKey.partial_decrypt`` / ``.decrypt`` and the
``DistributedPaillier`` biprimality / sieve class-methods return on seeded inputs, including the
reference's 24 stored key fixtures.  ``tests/test_oracle_golden.py`` checks every function here
against those vectors.
"""
