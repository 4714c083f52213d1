"""A builder-written STAND-IN for the reference package ``tno.mpc.protocols.distributed_keygen`` — test
infrastructure, not product code and not reference text.

The reference cannot travel to the GPU box and its dependencies are not installed anywhere, so the drop-in
patch (protocols/distributed_keygen_amd/patch.py) could only ever be driven by a CPU test double there
(VERDICT r02 "weak" 3).  This package re-creates, from the interface description in SURVEY.md §8b and the
attribute accesses patch.py makes, the SURFACE the patch binds to — module names ``paillier_shared_key`` /
``distributed_keygen``, the classes ``PaillierSharedKey`` / ``DistributedPaillier`` with the methods and
name-mangled class-methods the patch replaces, the share containers ``Batched`` / ``AdditiveVariable`` /
``ShamirVariable``, ``exchange_reconstruct``, ``Shares``, ``EncodedPlaintext``, and the leaf names
``pow_mod`` / ``mod_inv`` — with plain textbook implementations (CPython big integers) behind it.  The
protocol logic is deliberately minimal (three in-process parties over an in-memory pool); what matters is
that ``patch.install(package="keygen_standin")`` finds everything it rebinds, and that the unpatched
stand-in is an independent CPU computation of the same results to compare the patched GPU run against.
"""
