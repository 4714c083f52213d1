"""pytest configuration: registers the ``gpu`` marker and exposes the golden vectors."""

from __future__ import annotations

import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config: pytest.Config) -> None:
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The GPU suite runs like a process that keeps several launches in flight: 16 HIP hardware queues instead of the
    # runtime's default of 4, chosen before anything touches the GPU (it cannot be changed afterwards).  With 4, the
    # chunked int-level path (Engine._pipelined: up to 8 side streams) and every test with more than four streams only
    # ever exercised four-way concurrency (GPUTEST r03 warning).  No effect on the CPU tests.
    from protocols.distributed_keygen_amd import configure_hw_queues

    configure_hw_queues(16)


@pytest.fixture(scope="session", autouse=True)
def _library_present() -> None:
    """The HIP library is built in-tree by __graft_entry__.build(); if a checkout arrives without it,
    build it once (hipcc cross-compiles without a GPU).  Never a CPU fallback: a failed build fails
    the tests that need the library."""
    from protocols.distributed_keygen_amd import build

    build.build(force=False)      # also rebuilds a library older than its sources


def unhex(s: str) -> int:
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


@pytest.fixture(scope="session")
def golden_ref_keys() -> dict:
    return json.loads((GOLDEN / "ref_keys.json").read_text())


@pytest.fixture(scope="session")
def golden_decrypt_synth() -> dict:
    return json.loads((GOLDEN / "decrypt_synth.json").read_text())


@pytest.fixture(scope="session")
def golden_biprime() -> dict:
    return json.loads((GOLDEN / "biprime.json").read_text())


@pytest.fixture(scope="session")
def golden_reconstruct() -> dict:
    return json.loads((GOLDEN / "reconstruct.json").read_text())
